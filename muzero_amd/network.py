"""Network API of the reference (network.py:124-279) on top of the fused HIP inference kernels.

`Network.initial_inference(observation)` and `Network.recurrent_inference(hidden_state, action)` return a
`BatchNetworkOutput` with the reference's field names; tensors are torch tensors on the network's device
(anything with `.numpy()`/indexing satisfies the reference's callers).  The value/reward scalarisation of
`split_batch` (network.py:165-174) is fused into the kernels; the un-scalarised value head is still returned
in `batch_value` so `value_to_scalar(batch_value)` gives the same numbers.

Weights are one flat float32 buffer in Keras layer order (kernel, bias per layer; layout in DESIGN.md), the
format `save`/`load` use (reference network.py:256-260 writes a TF checkpoint; TensorFlow is not available).
"""
import ctypes
import math
from typing import Callable, List, Optional

import numpy as np
import torch

from . import _lib
from .exceptions import MuZeroError

MZ_NET_TICTACTOE, MZ_NET_CARTPOLE, MZ_NET_ATARI_SHAPED = 1, 2, 3


def _to_numpy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class NetworkOutput:
    """One game's network output (network.py:124-146)."""

    def __init__(self, hidden_state, reward: Optional[float], value: float, policy_logits) -> None:
        self.hidden_state = hidden_state
        self.reward = reward
        self.value = value
        self.policy_logits = policy_logits

    def __repr__(self) -> str:
        return (f'NetworkOutput(hidden_state={self.hidden_state}, reward={self.reward}, value={self.value}, '
                f'policy_logits={self.policy_logits})')

    def masked_policy(self, actions: List[int]) -> np.ndarray:
        """exp(logits) restricted to `actions`, normalised; float32, full length A (network.py:142-146)."""
        logits = _to_numpy(self.policy_logits).astype(np.float32)
        mask = np.zeros_like(logits)
        mask[list(actions)] = 1
        weights = np.exp(logits) * mask
        return weights / weights.sum()


class BatchNetworkOutput:
    """network.py:149-174.  `batch_value_scalar` / `batch_reward_scalar` hold the kernel-fused to_scalar results."""

    def __init__(self, batch_hidden_state, batch_reward, batch_value, batch_policy_logits,
                 reward_to_scalar: Callable, value_to_scalar: Callable,
                 batch_value_scalar=None, batch_reward_scalar=None) -> None:
        self.batch_hidden_state = batch_hidden_state
        self.batch_reward = batch_reward
        self.batch_value = batch_value
        self.batch_policy_logits = batch_policy_logits
        self.reward_to_scalar = reward_to_scalar
        self.value_to_scalar = value_to_scalar
        self.batch_value_scalar = batch_value_scalar
        self.batch_reward_scalar = batch_reward_scalar

    def value_scalars(self):
        if self.batch_value_scalar is not None:
            return self.batch_value_scalar
        v = self.value_to_scalar(self.batch_value)
        return v.reshape(v.shape[0])

    def reward_scalars(self):
        if self.batch_reward is None and self.batch_reward_scalar is None:
            return None
        if self.batch_reward_scalar is not None:
            return self.batch_reward_scalar
        r = self.reward_to_scalar(self.batch_reward)
        return r.reshape(r.shape[0])

    def split_batch(self) -> List[NetworkOutput]:
        values = _to_numpy(self.value_scalars()).reshape(-1)
        rewards = self.reward_scalars()
        rewards = None if rewards is None else _to_numpy(rewards).reshape(-1)
        return [NetworkOutput(hidden_state=self.batch_hidden_state[i],
                              reward=None if rewards is None else float(rewards[i]),
                              value=float(values[i]),
                              policy_logits=self.batch_policy_logits[i])
                for i in range(len(values))]


# ---- state-action encoders (network.py:14-116) as tensor ops, for networks composed outside the built-in kernels
# (the three built-in networks fold their encoder into a bias table or vector).  Tensors stay on
# the device they are on.
def one_hot_tensor_encoder(hidden_state: torch.Tensor, action: torch.Tensor, action_space_size: int) -> torch.Tensor:
    """network.py:14-40: [B, ..., C] -> [B, ..., C + A]; channel C + action[b] is 1 at every position of sample b."""
    planes = torch.zeros(hidden_state.shape[:-1] + (int(action_space_size),), dtype=hidden_state.dtype,
                         device=hidden_state.device)
    idx = action.to(device=hidden_state.device, dtype=torch.long).reshape((-1,) + (1,) * (hidden_state.dim() - 1))
    planes.scatter_(-1, idx.expand(hidden_state.shape[:-1] + (1,)), 1.0)
    return torch.cat([hidden_state, planes], dim=-1)


def tensor_product_encoder(hidden_state: torch.Tensor, action: torch.Tensor, action_space_size: int) -> torch.Tensor:
    """network.py:43-75: output[b, action[b], ...] = hidden_state[b, ...], zeros elsewhere ([B, A, ...])."""
    B = hidden_state.shape[0]
    out = torch.zeros((B, int(action_space_size)) + tuple(hidden_state.shape[1:]), dtype=hidden_state.dtype,
                      device=hidden_state.device)
    out[torch.arange(B, device=hidden_state.device), action.to(device=hidden_state.device, dtype=torch.long)] = hidden_state
    return out


def binary_plane_encoder(hidden_state: torch.Tensor, action: torch.Tensor, axis: int = 0) -> torch.Tensor:
    """network.py:88-104: one more plane along `axis` (of a sample; +1 for the batch axis), all 0 or all 1 = action[b]."""
    shape = list(hidden_state.shape)
    shape[axis + 1] = 1
    plane = action.to(device=hidden_state.device, dtype=hidden_state.dtype).reshape((-1,) + (1,) * (hidden_state.dim() - 1))
    return torch.cat([hidden_state, plane.expand(shape)], dim=axis + 1)


def glorot_uniform_weights(spec, seed: int) -> np.ndarray:
    """Keras defaults (glorot-uniform kernels, zero biases) for a [(name, shape)] spec -> flat float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    for name, shape in spec:
        if name.endswith('_k'):
            receptive = int(np.prod(shape[:-2]))
            limit = math.sqrt(6.0 / (receptive * shape[-2] + receptive * shape[-1]))
            parts.append(rng.uniform(-limit, limit, size=shape).astype(np.float32).ravel())
        else:
            parts.append(np.zeros(int(np.prod(shape)), np.float32))
    return np.concatenate(parts)


class Network:
    """Base class of the HIP-backed networks (reference network.py:177-279).

    Sub-classes set KIND, WEIGHT_SPEC, OBSERVATION_SHAPE, HIDDEN_SHAPE, ACTION_SPACE_SIZE, VALUE_RAW.
    """
    KIND = 0
    WEIGHT_SPEC = []
    OBSERVATION_SHAPE = ()
    HIDDEN_SHAPE = ()
    ACTION_SPACE_SIZE = 0
    VALUE_RAW = 1          # width of the un-scalarised value head

    # the layer sizes the HIP kernels implement, under the names the reference's network classes take them by
    # (games/tictactoe.py:156-161, games/cartpole.py:68-74); None = the reference class has no such parameter
    SIZES = {'hidden_state_size': 128, 'hidden_size': 128, 'support_size': None}

    def __init__(self, config, weights: Optional[np.ndarray] = None, seed: int = 0, training_steps: int = 0,
                 device='cuda', regularizer=None, **size_parameters) -> None:
        # `regularizer` only matters to training (reference network classes hand it to Keras); the sizes select the
        # architecture, and the kernels implement exactly one: anything else is refused, never silently replaced
        for name, value in size_parameters.items():
            if name not in self.SIZES:
                raise MuZeroError(f'{type(self).__name__}: unknown network parameter {name!r}')
            if value != self.SIZES[name]:
                raise MuZeroError(f'{type(self).__name__}: {name}={value!r} is not implemented by the HIP kernels '
                                  f'(they are built for {name}={self.SIZES[name]!r})')
        self.config = config
        self._h = None
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise MuZeroError('Network needs a HIP device (torch.cuda.is_available() is False); '
                              'muzero_amd has no CPU fallback')
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.hidden_size = int(np.prod(self.HIDDEN_SHAPE))
        self.observation_size = int(np.prod(self.OBSERVATION_SHAPE))
        self.steps = int(training_steps)
        self._stream = None
        self.set_weights(glorot_uniform_weights(self.WEIGHT_SPEC, seed) if weights is None else weights,
                         training_steps)

    # ------------------------------------------------------------------ weights
    @classmethod
    def num_weights(cls) -> int:
        return sum(int(np.prod(shape)) for _, shape in cls.WEIGHT_SPEC)

    def set_weights(self, flat: np.ndarray, training_steps: Optional[int] = None) -> None:
        """Upload a new flat weight buffer (the weight hand-off of training.py:62-68 / network.py:256-266)."""
        flat = np.ascontiguousarray(flat, np.float32).ravel()
        if flat.size != self.num_weights():
            raise MuZeroError(f'{type(self).__name__} takes {self.num_weights()} weights, got {flat.size}')
        if training_steps is not None:
            self.steps = int(training_steps)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.mz_net_create(self.KIND, flat.ctypes.data, flat.size, self.steps,
                                               ctypes.c_void_p(self._stream), ctypes.byref(handle)))
        self.close()
        self._h = handle
        self._weights = flat.copy()

    def get_weights(self) -> np.ndarray:
        return self._weights.copy()

    def training_steps(self) -> int:
        return self.steps

    def save(self, checkpoint_prefix: str) -> None:
        np.savez(checkpoint_prefix + '.npz', weights=self._weights, steps=np.int64(self.steps),
                 kind=np.int32(self.KIND))

    def load(self, checkpoint_prefix: str) -> None:
        with np.load(checkpoint_prefix + '.npz', allow_pickle=False) as data:
            if int(data['kind']) != self.KIND:
                raise MuZeroError('checkpoint is for a different network kind')
            self.set_weights(data['weights'], int(data['steps']))

    # ------------------------------------------------------------------ weight hand-off
    WEIGHTS_FILE = 'weights.npz'

    def save_version(self, models_path: str) -> str:
        """Publish the current weights as version `training_steps()`: `<models_path>/<steps>/weights.npz`, the
        versioned-directory convention of Network.save_tfx_models (network.py:262-266, called from training.py:62-63)
        that TF-Serving watches.  Written to a temporary directory and renamed, so a reader never sees half a file."""
        import os
        import tempfile
        os.makedirs(models_path, exist_ok=True)
        final = os.path.join(models_path, str(self.steps))
        tmp = tempfile.mkdtemp(prefix='.tmp_', dir=models_path)
        np.savez(os.path.join(tmp, self.WEIGHTS_FILE), weights=self._weights, steps=np.int64(self.steps),
                 kind=np.int32(self.KIND))
        if os.path.isdir(final):                      # same version published twice: keep the first
            import shutil
            shutil.rmtree(tmp)
        else:
            os.rename(tmp, final)
        return final

    @classmethod
    def available_versions(cls, models_path: str) -> List[int]:
        import os
        if not os.path.isdir(models_path):
            return []
        return sorted(int(d) for d in os.listdir(models_path)
                      if d.isdigit() and os.path.isfile(os.path.join(models_path, d, cls.WEIGHTS_FILE)))

    def load_latest(self, models_path: str) -> bool:
        """Hot-reload: upload the highest published version if it is newer than `training_steps()` (what a
        RemoteNetwork observes as response.model_spec.version, network_services.py:57).  -> True if weights changed.
        Call it between moves or games: the upload is ordered on the network's stream."""
        import os
        versions = self.available_versions(models_path)
        if not versions or versions[-1] <= self.steps:
            return False
        with np.load(os.path.join(models_path, str(versions[-1]), self.WEIGHTS_FILE), allow_pickle=False) as data:
            if int(data['kind']) != self.KIND:
                raise MuZeroError('published weights are for a different network kind')
            self.set_weights(data['weights'], int(data['steps']))
        return True

    def status(self) -> int:
        """mz_net_status: bit 0 = an activation left the kernels' numeric domain (|v| > 65504) since the last call and was
        saturated.  Clears the flag; synchronises."""
        self._sync_stream()
        flags = ctypes.c_int32(0)
        _lib.check(self._lib.mz_net_status(self._h, ctypes.addressof(flags)))
        return int(flags.value)

    def set_run_mode(self, mode: int) -> None:
        """mz_debug_set_run_mode (tests, measurement): how mz_infer_recurrent_run / mz_search_move run a TicTacToe move's
        simulations -- 0 one launch (default), 1 one launch with agent-scope fences at every meeting, 2 a launch per
        simulation.  Same results in every mode."""
        _lib.check(self._lib.mz_debug_set_run_mode(self._h, int(mode)))

    def set_job_rows(self, rows: int) -> None:
        """mz_debug_set_job_rows (tests, measurement): the TicTacToe job kernel's games per workgroup, 0 = by batch size."""
        _lib.check(self._lib.mz_debug_set_job_rows(self._h, int(rows)))

    def run_info(self):
        """mz_debug_run_info -> (row-block runs that took the fence path, row-block runs); synchronises."""
        return self.run_info_all()[:2]

    def run_info_all(self):
        """-> (fence-path row-block runs, row-block runs, 100 MHz ticks spent inside jobs, jobs); synchronises."""
        out = np.zeros(4, np.int64)
        _lib.check(self._lib.mz_debug_run_info(self._h, out.ctypes.data))
        return tuple(int(x) for x in out)

    def check_range(self) -> None:
        """Raises MuZeroError if an inference since the last check saturated an activation (see include/mzhip.h,
        mz_net_create: the numeric domain of the split-f16 kernels)."""
        if self.status() & 1:
            raise MuZeroError(f'{type(self).__name__}: an activation exceeded the inference kernels\' numeric domain '
                              '(|v| > 65504) and was saturated; outputs since the last check are unreliable')

    def close(self):
        if self._h is not None:
            self._lib.mz_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self) -> str:
        return f'{type(self).__name__}(parameters={self.num_weights()}, training_steps={self.steps})'

    # ------------------------------------------------------------------ inference
    def _sync_stream(self):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if stream != self._stream:
            _lib.check(self._lib.mz_net_set_stream(self._h, ctypes.c_void_p(stream)))
            self._stream = stream

    def _as_f32(self, x, width):
        t = torch.as_tensor(x if isinstance(x, torch.Tensor) else np.asarray(x), device=self.device)
        t = t.to(torch.float32).reshape(-1, width).contiguous()
        return t

    def _outputs(self, batch, with_reward):
        f32 = dict(dtype=torch.float32, device=self.device)
        return dict(hidden=torch.empty(batch, self.hidden_size, **f32),
                    reward=torch.empty(batch, **f32) if with_reward else None,
                    value=torch.empty(batch, **f32), logits=torch.empty(batch, self.ACTION_SPACE_SIZE, **f32),
                    raw=torch.empty(batch, self.VALUE_RAW, **f32))

    def _wrap(self, out):
        return BatchNetworkOutput(
            batch_hidden_state=out['hidden'].view(-1, *self.HIDDEN_SHAPE),
            batch_reward=None if out['reward'] is None else out['reward'].view(-1, 1),
            batch_value=out['raw'], batch_policy_logits=out['logits'],
            reward_to_scalar=self.config.reward_config.to_scalar if self.config is not None else (lambda t: t),
            value_to_scalar=self.config.value_config.to_scalar if self.config is not None else (lambda t: t),
            batch_value_scalar=out['value'], batch_reward_scalar=out['reward'])

    def initial_inference(self, observation) -> BatchNetworkOutput:
        """network.py:221-229.  observation: [B, *OBSERVATION_SHAPE]."""
        self._sync_stream()
        obs = self._as_f32(observation, self.observation_size)
        out = self._outputs(obs.shape[0], with_reward=False)
        _lib.check(self._lib.mz_infer_initial(self._h, obs.shape[0], obs.data_ptr(), out['hidden'].data_ptr(),
                                              out['value'].data_ptr(), out['logits'].data_ptr(),
                                              out['raw'].data_ptr()))
        return self._wrap(out)

    def recurrent_inference(self, hidden_state, action) -> BatchNetworkOutput:
        """network.py:246-254.  hidden_state: [B, *HIDDEN_SHAPE]; action: [B] ints."""
        self._sync_stream()
        hid = self._as_f32(hidden_state, self.hidden_size)
        act = torch.as_tensor(action if isinstance(action, torch.Tensor) else np.asarray(action),
                              device=self.device).to(torch.int32).reshape(-1).contiguous()
        if act.shape[0] != hid.shape[0]:
            raise MuZeroError(f'batch mismatch: {hid.shape[0]} hidden states, {act.shape[0]} actions')
        out = self._outputs(hid.shape[0], with_reward=True)
        _lib.check(self._lib.mz_infer_recurrent(self._h, hid.shape[0], hid.data_ptr(), None, 0, act.data_ptr(),
                                                out['hidden'].data_ptr(), out['reward'].data_ptr(),
                                                out['value'].data_ptr(), out['logits'].data_ptr(),
                                                out['raw'].data_ptr()))
        return self._wrap(out)

    def recurrent_inference_pool(self, hidden_pool, parent_block, action, out_block, reward, value, logits):
        """The lockstep fast path: gather leaf.parent.hidden_state rows of `hidden_pool` [N+1,T,H] by
        `parent_block` inside the kernel (agent.py:204) and write the new hidden states straight into
        hidden_pool[out_block]; reward/value/logits are caller-owned [T]/[T]/[T,A] f32 outputs."""
        self._sync_stream()
        n_blocks, T, H = hidden_pool.shape
        if H != self.hidden_size or not hidden_pool.is_contiguous() or hidden_pool.dtype != torch.float32:
            raise MuZeroError(f'hidden_pool must be a contiguous float32 [blocks, T, {self.hidden_size}] tensor')
        if not 0 <= out_block < n_blocks:
            raise MuZeroError(f'out_block {out_block} outside [0, {n_blocks})')
        for name, t, dt, shape in (('parent_block', parent_block, torch.int32, (T,)), ('action', action, torch.int32, (T,)),
                                   ('reward', reward, torch.float32, (T,)), ('value', value, torch.float32, (T,)),
                                   ('logits', logits, torch.float32, (T, self.ACTION_SPACE_SIZE))):
            if t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous() or t.device != hidden_pool.device:
                raise MuZeroError(f'{name}: expected contiguous {dt} {shape} on {hidden_pool.device}')
        _lib.check(self._lib.mz_infer_recurrent(self._h, T, hidden_pool.data_ptr(), parent_block.data_ptr(), T * H,
                                                action.data_ptr(), hidden_pool[out_block].data_ptr(),
                                                reward.data_ptr(), value.data_ptr(), logits.data_ptr(), None))

    def run_simulations_pool(self, pool, num_simulations, hidden_pool, reward, value, logits, final_select=False,
                             trace=None):
        """run_mcts (reference agent.py:198-211) on a TreePool that holds a pending selection: `num_simulations` times
        { recurrent inference of the selected leaves' parents, expand_node, backpropagate, select_leaf of the next
        simulation } -- mz_infer_recurrent_run: ONE launch for the CartPole network, a simulation at a time otherwise.
        The pool's parent/action tensors carry the selection in and (with `final_select`) the next one out; reward /
        value / logits [T] / [T] / [T,A] receive the last simulation's outputs, or with `trace` = dict of
        reward/value [n,T], logits [n,T,A], parent/action [n,T] tensors every simulation's."""
        self._sync_stream()
        pool._sync_stream()
        n_blocks, T, H = hidden_pool.shape
        if H != self.hidden_size or not hidden_pool.is_contiguous() or hidden_pool.dtype != torch.float32 or T != pool.T:
            raise MuZeroError(f'hidden_pool must be a contiguous float32 [blocks, {pool.T}, {self.hidden_size}] tensor')
        ptr = lambda t: None if t is None else t.data_ptr()
        out = trace if trace is not None else dict(reward=reward, value=value, logits=logits)
        _lib.check(self._lib.mz_infer_recurrent_run(
            self._h, pool._h, int(num_simulations), int(bool(final_select)), hidden_pool.data_ptr(),
            pool._parent.data_ptr(), T * H, pool._action.data_ptr(), ptr(out['reward']), ptr(out['value']),
            ptr(out['logits']), T if trace is not None else 0, ptr(out.get('parent')), ptr(out.get('action'))))
        return pool._parent, pool._action

    def initial_inference_pool(self, observation, hidden_pool, value, logits):
        """initial_inference writing the root hidden states into hidden_pool[0]."""
        self._sync_stream()
        obs = self._as_f32(observation, self.observation_size)
        if obs.shape[0] != hidden_pool.shape[1]:
            raise MuZeroError('observation batch does not match the pool')
        _lib.check(self._lib.mz_infer_initial(self._h, obs.shape[0], obs.data_ptr(), hidden_pool[0].data_ptr(),
                                              value.data_ptr(), logits.data_ptr(), None))


def _spec(kind):
    # kept in one place: the same tables the oracle uses are restated here so the product never imports oracle/
    if kind == 'tictactoe':      # reference games/tictactoe.py:179-213
        return [('rep_conv_k', (3, 3, 3, 128)), ('rep_conv_b', (128,)), ('rep_dense_k', (128, 128)), ('rep_dense_b', (128,)),
                ('dyn_conv_k', (3, 3, 137, 128)), ('dyn_conv_b', (128,)), ('dyn_dense_k', (128, 128)), ('dyn_dense_b', (128,)),
                ('dyn_reward_k', (1152, 1)), ('dyn_reward_b', (1,)),
                ('pred_conv_k', (3, 3, 128, 128)), ('pred_conv_b', (128,)), ('pred_dense_k', (128, 128)), ('pred_dense_b', (128,)),
                ('pred_value_k', (128, 1)), ('pred_value_b', (1,)), ('pred_policy_k', (128, 9)), ('pred_policy_b', (9,))]
    if kind == 'cartpole':       # reference games/cartpole.py:92-132
        return [('rep_conv_k', (4, 101, 128)), ('rep_conv_b', (128,)), ('rep_dense_k', (128, 128)), ('rep_dense_b', (128,)),
                ('dyn_conv_k', (2, 128, 128)), ('dyn_conv_b', (128,)), ('dyn_dense_k', (128, 128)), ('dyn_dense_b', (128,)),
                ('dyn_rdense_k', (128, 128)), ('dyn_rdense_b', (128,)), ('dyn_reward_k', (128, 1)), ('dyn_reward_b', (1,)),
                ('pred_dense1_k', (128, 128)), ('pred_dense1_b', (128,)), ('pred_dense2_k', (128, 128)), ('pred_dense2_b', (128,)),
                ('pred_value_k', (128, 101)), ('pred_value_b', (101,)), ('pred_policy_k', (128, 2)), ('pred_policy_b', (2,))]
    if kind == 'atari':          # the synthetic network of BASELINE.json configs[3] (include/mzhip.h MZ_NET_ATARI_SHAPED)
        res = [(f'dyn_res{b}{ab}_{kb}', (3, 128, 128) if kb == 'k' else (128,))
               for b in (1, 2) for ab in 'ab' for kb in 'kb']
        return ([('rep_conv_k', (3, 128, 128)), ('rep_conv_b', (128,)), ('rep_dense_k', (128, 128)), ('rep_dense_b', (128,)),
                 ('dyn_conv_k', (3, 146, 128)), ('dyn_conv_b', (128,))] + res +
                [('dyn_reward_k', (256, 1)), ('dyn_reward_b', (1,)),
                 ('pred_conv_k', (2, 128, 128)), ('pred_conv_b', (128,)), ('pred_dense_k', (128, 128)), ('pred_dense_b', (128,)),
                 ('pred_value_k', (128, 1)), ('pred_value_b', (1,)), ('pred_policy_k', (128, 18)), ('pred_policy_b', (18,))])
    raise MuZeroError(f'unknown network kind {kind}')


class TicTacToeNetwork(Network):
    """games/tictactoe.py:151-220: observation (3,3,3), hidden (3,3,128), tanh value/reward, 9 logits."""
    KIND = MZ_NET_TICTACTOE
    WEIGHT_SPEC = _spec('tictactoe')
    OBSERVATION_SHAPE = (3, 3, 3)
    HIDDEN_SHAPE = (3, 3, 128)
    ACTION_SPACE_SIZE = 9
    VALUE_RAW = 1


class CartPoleNetwork(Network):
    """games/cartpole.py:63-139: observation (4,), hidden (1,128), 101-bin value support, sigmoid reward."""
    KIND = MZ_NET_CARTPOLE
    WEIGHT_SPEC = _spec('cartpole')
    OBSERVATION_SHAPE = (4,)
    HIDDEN_SHAPE = (1, 128)
    ACTION_SPACE_SIZE = 2
    VALUE_RAW = 101
    SIZES = {'hidden_state_size': 128, 'hidden_size': 128, 'support_size': 100}


class AtariShapedNetwork(Network):
    """The synthetic Atari-shaped network of BASELINE.json configs[3] (no reference class; built from the reference's
    blocks, see include/mzhip.h MZ_NET_ATARI_SHAPED): observation (2,128), hidden (2,128) = 256 floats, conv-residual
    dynamics with one-hot action planes, tanh value/reward, 18 logits."""
    KIND = MZ_NET_ATARI_SHAPED
    WEIGHT_SPEC = _spec('atari')
    OBSERVATION_SHAPE = (2, 128)
    HIDDEN_SHAPE = (2, 128)
    ACTION_SPACE_SIZE = 18
    VALUE_RAW = 1
