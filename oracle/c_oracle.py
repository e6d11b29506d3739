"""ctypes front-end of oracle/_build/libmzoracle.so (plain-C CPU oracle) -- TEST INFRASTRUCTURE ONLY.

Mirrors the method names of muzero_amd.tree_pool.TreePool on numpy arrays so parity tests can drive the HIP
pool and the oracle with the same code.  Never imported by muzero_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libmzoracle.so')
_lib = None


class _Config(ctypes.Structure):
    _fields_ = [('num_trees', ctypes.c_int32), ('num_actions', ctypes.c_int32), ('max_simulations', ctypes.c_int32),
                ('num_players', ctypes.c_int32), ('has_known_bounds', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('discount', ctypes.c_double), ('default_value', ctypes.c_double), ('known_min', ctypes.c_double),
                ('known_max', ctypes.c_double), ('log_table', ctypes.c_void_p)]


def build():
    subprocess.run(['make', '-s', '-C', _HERE], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.mzo_numpy_sum_f32.restype = ctypes.c_float
        _lib.mzo_numpy_sum_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def numpy_sum_f32(a):
    a = np.ascontiguousarray(a, np.float32)
    return np.float32(lib().mzo_numpy_sum_f32(_ptr(a), a.size))


def masked_policy_from_exp(exp_logits, actions, A):
    e = np.ascontiguousarray(exp_logits, np.float32)
    acts = np.ascontiguousarray(actions, np.int32)
    out = np.empty(A, np.float32)
    lib().mzo_masked_policy_from_exp(_ptr(e), _ptr(acts), ctypes.c_int32(len(acts)), ctypes.c_int32(A), _ptr(out))
    return out


class OraclePool:
    def __init__(self, num_trees, num_actions, max_simulations, num_players=1, discount=1.0, default_value=0.0,
                 known_bounds=None, log_table=None):
        self.T, self.A, self.N = int(num_trees), int(num_actions), int(max_simulations)
        cfg = _Config(self.T, self.A, self.N, int(num_players), int(known_bounds is not None), 0, float(discount),
                      float(default_value), float(known_bounds[0]) if known_bounds else 0.0,
                      float(known_bounds[1]) if known_bounds else 0.0, None)
        self._log = None
        if log_table is not None:
            self._log = np.ascontiguousarray(log_table, np.float64)
            cfg.log_table = self._log.ctypes.data
        self._h = ctypes.c_void_p()
        rc = lib().mzo_pool_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != 0:
            raise MemoryError('mzo_pool_create failed')

    def __del__(self):
        if getattr(self, '_h', None):
            lib().mzo_pool_destroy(self._h)
            self._h = None

    def expand_roots(self, priors, legal=None, n_legal=None):
        priors = np.ascontiguousarray(priors, np.float32).reshape(self.T, self.A)
        legal = None if legal is None else np.ascontiguousarray(legal, np.int32).reshape(self.T, self.A)
        n_legal = None if n_legal is None else np.ascontiguousarray(n_legal, np.int32).reshape(self.T)
        lib().mzo_expand_roots(self._h, _ptr(priors), _ptr(legal), _ptr(n_legal))

    def add_root_noise(self, noise, frac):
        noise = np.ascontiguousarray(noise, np.float64).reshape(self.T, self.A)
        lib().mzo_add_root_noise(self._h, _ptr(noise), ctypes.c_double(frac))

    def select(self):
        parent = np.empty(self.T, np.int32)
        action = np.empty(self.T, np.int32)
        lib().mzo_select(self._h, _ptr(parent), _ptr(action))
        return parent, action

    def expand_backprop(self, priors, reward, value):
        priors = np.ascontiguousarray(priors, np.float32).reshape(self.T, self.A)
        reward = np.ascontiguousarray(reward, np.float32).reshape(self.T)
        value = np.ascontiguousarray(value, np.float32).reshape(self.T)
        lib().mzo_expand_backprop(self._h, _ptr(priors), _ptr(reward), _ptr(value))

    def root_stats(self):
        visits = np.empty((self.T, self.A), np.int32)
        root_value = np.empty(self.T, np.float64)
        minmax = np.empty((self.T, 2), np.float64)
        policy = np.empty((self.T, self.A), np.float32)
        lib().mzo_root_stats(self._h, _ptr(visits), _ptr(root_value), _ptr(minmax), _ptr(policy))
        return visits, root_value, minmax, policy

    def select_action(self, u, temperature):
        u = np.ascontiguousarray(u, np.float64).reshape(self.T)
        temperature = np.ascontiguousarray(np.broadcast_to(temperature, (self.T,)), np.float64)
        out = np.empty(self.T, np.int32)
        lib().mzo_select_action(self._h, _ptr(u), _ptr(temperature), _ptr(out))
        return out

    def counters(self):
        out = np.zeros(4, np.int64)
        lib().mzo_pool_counters(self._h, _ptr(out))
        return out

    def read_block(self, tree, block):
        A = self.A
        n = ctypes.c_int32(0)
        actions, count, child = (np.full(A, -1, np.int32) for _ in range(3))
        vsum, prior = np.zeros(A, np.float64), np.zeros(A, np.float64)
        reward = np.zeros(A, np.float32)
        rc = lib().mzo_pool_read_block(self._h, ctypes.c_int32(tree), ctypes.c_int32(block), ctypes.byref(n),
                                       _ptr(actions), _ptr(count), _ptr(vsum), _ptr(prior), _ptr(reward), _ptr(child))
        if rc != 0:
            raise IndexError(f'block {block} of tree {tree} does not exist')
        k = n.value
        return dict(actions=actions[:k], count=count[:k], value_sum=vsum[:k], prior=prior[:k], reward=reward[:k],
                    child=child[:k])
