"""TreePool: the GPU-resident SoA tree pool behind MCTSAgent (host side of include/mzhip.h's mz_pool_*).

One pool steps T independent search trees in lockstep; every method is one HIP launch on torch's current
stream.  Arguments and results are torch tensors on the pool's device (plumbing only: device memory and
streams).  Reference semantics: agent.py:154-232, utils.py:37-76, config.py:159-160.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .exceptions import MuZeroError


def _check_tensor(name, t, dtype, shape, device):
    if not isinstance(t, torch.Tensor):
        raise MuZeroError(f'{name} must be a torch.Tensor on {device}')
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or t.device != device or not t.is_contiguous():
        raise MuZeroError(f'{name}: expected contiguous {dtype} tensor of shape {tuple(shape)} on {device}, got '
                          f'{t.dtype} {tuple(t.shape)} on {t.device} (contiguous={t.is_contiguous()})')
    return t.data_ptr()


class TreePool:
    def __init__(self, num_trees, num_actions, max_simulations, num_players=1, discount=1.0, default_value=0.0,
                 known_bounds=None, device='cuda', lanes_per_tree=0):
        self._h = None
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise MuZeroError('TreePool needs a HIP device (torch.cuda.is_available() is False); '
                              'muzero_amd has no CPU fallback')
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.T, self.A, self.N = int(num_trees), int(num_actions), int(max_simulations)
        self.known_bounds = None if known_bounds is None else (float(known_bounds[0]), float(known_bounds[1]))
        # config.py:159-160 tabulated with the host's math.log (trap T6); parent visit count <= N
        self._log_table = np.array([2 * math.log(n + 1) for n in range(self.N + 2)], np.float64)
        cfg = _lib.PoolConfig(self.T, self.A, self.N, int(num_players), int(known_bounds is not None),
                              int(lanes_per_tree),
                              float(discount), float(default_value),
                              self.known_bounds[0] if known_bounds is not None else 0.0,
                              self.known_bounds[1] if known_bounds is not None else 0.0,
                              self._log_table.ctypes.data)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.mz_pool_create(ctypes.byref(cfg), ctypes.c_void_p(self._stream), ctypes.byref(handle)))
        self._h = handle
        self._last_temperature = None
        i32 = dict(dtype=torch.int32, device=self.device)
        self._parent = torch.empty(self.T, **i32)
        self._action = torch.empty(self.T, **i32)

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if self._h is not None:
            self._lib.mz_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _sync_stream(self):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if stream != self._stream:
            _lib.check(self._lib.mz_pool_set_stream(self._h, ctypes.c_void_p(stream)))
            self._stream = stream

    def _f32(self, name, t, shape):
        return _check_tensor(name, t, torch.float32, shape, self.device)

    @property
    def num_blocks(self):
        return self._lib.mz_pool_num_blocks(self._h)

    @property
    def bytes(self):
        return self._lib.mz_pool_bytes(self._h)

    # ------------------------------------------------------------------ search
    def reset(self):
        """agent.py:220-221: fresh roots + MinMaxStats for the next move."""
        _lib.check(self._lib.mz_pool_reset(self._h))

    def expand_roots(self, policy, legal=None, n_legal=None, logits=False):
        """Root expand_node (agent.py:154-160,223-225). policy [T,A] f32 = masked_policy(legal) (or raw logits)."""
        self._sync_stream()
        p = self._f32('policy', policy, (self.T, self.A))
        lg = None if legal is None else _check_tensor('legal', legal, torch.int32, (self.T, self.A), self.device)
        nl = None if n_legal is None else _check_tensor('n_legal', n_legal, torch.int32, (self.T,), self.device)
        fn = self._lib.mz_expand_roots_logits if logits else self._lib.mz_expand_roots
        _lib.check(fn(self._h, p, lg, nl))

    def add_root_noise(self, noise, frac):
        """agent.py:170-175; noise [T,A] f64 (row t holds the Dirichlet draw over its legal actions)."""
        self._sync_stream()
        n = _check_tensor('noise', noise, torch.float64, (self.T, self.A), self.device)
        _lib.check(self._lib.mz_add_root_noise(self._h, n, float(frac)))

    def select(self):
        """select_leaf for every tree (agent.py:177-190) -> (parent_block [T] i32, action [T] i32)."""
        self._sync_stream()
        _lib.check(self._lib.mz_select(self._h, self._parent.data_ptr(), self._action.data_ptr()))
        return self._parent, self._action

    def set_selection(self, tree, blocks, actions):
        """The selection path of one tree given from outside (mz_pool_set_selection): blocks[i] / actions[i] = the record
        and the action taken in it, root first, the last child unvisited -> (parent_block [T], action [T])."""
        self._sync_stream()
        b = np.ascontiguousarray(blocks, np.int32)
        a = np.ascontiguousarray(actions, np.int32)
        if b.ndim != 1 or b.shape != a.shape or b.size < 1:
            raise MuZeroError('blocks and actions must be equally long, non-empty 1-D sequences')
        _lib.check(self._lib.mz_pool_set_selection(self._h, int(tree), b.ctypes.data, a.ctypes.data, int(b.size),
                                                   self._parent.data_ptr(), self._action.data_ptr()))
        return self._parent, self._action

    def expand_backprop(self, policy, reward, value, logits=False):
        """expand_node + backpropagate for the last select (agent.py:208-211)."""
        self._sync_stream()
        p = self._f32('policy', policy, (self.T, self.A))
        r = self._f32('reward', reward, (self.T,))
        v = self._f32('value', value, (self.T,))
        fn = self._lib.mz_expand_backprop_logits if logits else self._lib.mz_expand_backprop
        _lib.check(fn(self._h, p, r, v))

    def step(self, policy, reward, value, logits=False):
        """expand_backprop + the next select in one launch -> (parent_block, action)."""
        self._sync_stream()
        p = self._f32('policy', policy, (self.T, self.A))
        r = self._f32('reward', reward, (self.T,))
        v = self._f32('value', value, (self.T,))
        fn = self._lib.mz_step_logits if logits else self._lib.mz_step
        _lib.check(fn(self._h, p, r, v, self._parent.data_ptr(), self._action.data_ptr()))
        return self._parent, self._action

    def search_injected(self, policy, reward, value, noise=None, frac=0.25, legal=None, n_legal=None, logits=False,
                        num_simulations=None, events=None):
        """run_mcts (agent.py:198-211) against given network outputs, enqueued by one native call
        (mz_search_move_injected): policy [N+1,T,A] f32 (row 0 = the root's masked policy), reward/value [N+1,T] f32.
        events: optional ctypes array of 2*(N-1) hipEvent_t recorded around the fused steps (measurement)."""
        self._sync_stream()
        N = self.N if num_simulations is None else int(num_simulations)
        p = _check_tensor('policy', policy, torch.float32, (N + 1, self.T, self.A), self.device)
        r = _check_tensor('reward', reward, torch.float32, (N + 1, self.T), self.device)
        v = _check_tensor('value', value, torch.float32, (N + 1, self.T), self.device)
        nz = None if noise is None else _check_tensor('noise', noise, torch.float64, (self.T, self.A), self.device)
        lg = None if legal is None else _check_tensor('legal', legal, torch.int32, (self.T, self.A), self.device)
        nl = None if n_legal is None else _check_tensor('n_legal', n_legal, torch.int32, (self.T,), self.device)
        _lib.check(self._lib.mz_search_move_injected(self._h, N, p, int(bool(logits)), r, v, lg, nl, nz, float(frac),
                                                     self._parent.data_ptr(), self._action.data_ptr(), events))
        return self._parent, self._action

    # ------------------------------------------------------------------ results
    def root_stats(self):
        """-> visits [T,A] i32, root_value [T] f64, minmax [T,2] f64, policy [T,A] f32 (agent.py:229-231)."""
        self._sync_stream()
        visits = torch.empty(self.T, self.A, dtype=torch.int32, device=self.device)
        root_value = torch.empty(self.T, dtype=torch.float64, device=self.device)
        minmax = torch.empty(self.T, 2, dtype=torch.float64, device=self.device)
        policy = torch.empty(self.T, self.A, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.mz_root_stats(self._h, visits.data_ptr(), root_value.data_ptr(), minmax.data_ptr(),
                                           policy.data_ptr()))
        return visits, root_value, minmax, policy

    def set_temperature(self, temperature):
        """Registers the exact sampling weights count ** (1/t), c = 0..N, evaluated HERE by CPython like the reference
        does (agent.py:167: int ** float = the host libm's pow), for a temperature about to be used
        (mz_pool_set_temperature_table).  Cheap to call again with the temperature of the previous call."""
        t = float(temperature)
        if t == self._last_temperature:
            return
        self._last_temperature = t
        if t == 0.0 or 1 / t in (1.0, 2.0, 4.0):          # arg-max / exact products on the device: no table needed
            return
        self._sync_stream()
        table = np.array([c ** (1 / t) for c in range(self.N + 1)], np.float64)
        _lib.check(self._lib.mz_pool_set_temperature_table(self._h, t, table.ctypes.data))

    def select_action(self, u, temperature):
        """softmax_sample (agent.py:162-168) with injected uniforms u [T] f64 and temperature: a number, or [T] f64
        (then register every distinct value with set_temperature first)."""
        self._sync_stream()
        up = _check_tensor('u', u, torch.float64, (self.T,), self.device)
        if not isinstance(temperature, torch.Tensor):
            self.set_temperature(temperature)
            temperature = torch.full((self.T,), float(temperature), dtype=torch.float64, device=self.device)
        tp = _check_tensor('temperature', temperature, torch.float64, (self.T,), self.device)
        out = torch.empty(self.T, dtype=torch.int32, device=self.device)
        _lib.check(self._lib.mz_select_action(self._h, up, tp, out.data_ptr()))
        return out

    def counters(self):
        """[simulations, sum of path lengths d, select launches, backprop launches] (synchronises)."""
        self._sync_stream()
        out = np.zeros(4, np.int64)
        _lib.check(self._lib.mz_pool_counters(self._h, out.ctypes.data))
        return out

    def read_root_priors(self, tree):
        """The root's float32 priors before the noise mix, one per listed child (host copy; synchronises)."""
        self._sync_stream()
        n = ctypes.c_int32(0)
        out = np.zeros(self.A, np.float32)
        _lib.check(self._lib.mz_pool_read_root_priors(self._h, int(tree), ctypes.addressof(n), out.ctypes.data))
        return out[:n.value]

    def read_block(self, tree, block):
        """Host copy of one block (Node views, tests); synchronises."""
        self._sync_stream()
        A = self.A
        n = ctypes.c_int32(0)
        actions, count, child = (np.full(A, -1, np.int32) for _ in range(3))
        vsum, prior = np.zeros(A, np.float64), np.zeros(A, np.float64)
        reward = np.zeros(A, np.float32)
        _lib.check(self._lib.mz_pool_read_block(self._h, int(tree), int(block), ctypes.addressof(n),
                                                actions.ctypes.data, count.ctypes.data, vsum.ctypes.data,
                                                prior.ctypes.data, reward.ctypes.data, child.ctypes.data))
        k = n.value
        return dict(actions=actions[:k], count=count[:k], value_sum=vsum[:k], prior=prior[:k], reward=reward[:k],
                    child=child[:k])
