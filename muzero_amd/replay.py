"""Replay-side training targets on the device: reference ReplayBuffer.preprocess_history and
compute_target_value (replay_buffer.py:66-99) for a whole BatchedHistory in one launch (csrc/mz_replay.hip)."""
import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .exceptions import MuZeroError


def effective_discount(config) -> float:
    g = config.game_config
    return -g.discount if g.num_players == 2 else g.discount            # replay_buffer.py:26-28


def _support(scalar_config):
    s = scalar_config.support_size
    lo, hi = scalar_config.known_bounds.endpoints() if scalar_config.known_bounds is not None else (0.0, 1.0)
    return (int(s) if s else 0), float(lo), float(hi)


def compute_targets(history, config, random_actions: Optional[torch.Tensor] = None, generator=None) -> Dict[str, torch.Tensor]:
    """history: selfplay.BatchedHistory on a HIP device.  -> dict of padded move-major tensors
    extended_actions [M+K,G] i32, target_rewards [M+K,G,Dr] f32, target_values [M+K+1,G,Dv] f32,
    target_policies [M+K+1,G,A] f32, total_value [G] f64 (reference fields of the same names, replay_buffer.py:79-98).
    random_actions [G,K] i32 are the actions appended past the terminal state; drawn on the device when absent."""
    dev = history.lengths.device
    if dev.type != 'cuda':
        raise MuZeroError('compute_targets needs the history on a HIP device; muzero_amd has no CPU fallback')
    M, G, A = history.M, history.G, history.A
    K, td = int(config.training_config.num_unroll_steps), int(config.training_config.td_steps)
    disc = effective_discount(config)
    vs, vlo, vhi = _support(config.value_config)
    rs, rlo, rhi = _support(config.reward_config)
    if random_actions is None:
        random_actions = torch.randint(0, A, (G, max(K, 1)), device=dev, dtype=torch.int32, generator=generator)[:, :K]
    random_actions = random_actions.to(device=dev, dtype=torch.int32).contiguous()
    if tuple(random_actions.shape) != (G, K):
        raise MuZeroError(f'random_actions must have shape ({G}, {K})')
    powers = torch.tensor(np.array([disc ** i for i in range(max(td, M) + 1)], dtype=np.float64), device=dev)
    out = {
        'extended_actions': torch.empty(M + K, G, dtype=torch.int32, device=dev),
        'target_rewards': torch.empty(M + K, G, rs + 1 if rs else 1, dtype=torch.float32, device=dev),
        'target_values': torch.empty(M + K + 1, G, vs + 1 if vs else 1, dtype=torch.float32, device=dev),
        'target_policies': torch.empty(M + K + 1, G, A, dtype=torch.float32, device=dev),
        'total_value': torch.empty(G, dtype=torch.float64, device=dev),
    }
    cfg = _lib.ReplayConfig(G, M, A, K, td, vs, rs, 0, vlo, vhi, rlo, rhi)
    for name, dtype in (('lengths', torch.int32), ('actions', torch.int32), ('rewards', torch.float32),
                        ('root_values', torch.float64), ('policies', torch.float32)):
        t = getattr(history, name)
        if t.dtype != dtype or not t.is_contiguous() or t.device != dev:
            raise MuZeroError(f'history.{name} must be a contiguous {dtype} tensor on {dev}')
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.load().mz_replay_targets(
            stream, ctypes.byref(cfg), history.lengths.data_ptr(), history.actions.data_ptr(), history.rewards.data_ptr(),
            history.root_values.data_ptr(), history.policies.data_ptr(), random_actions.data_ptr(), powers.data_ptr(),
            out['extended_actions'].data_ptr(), out['target_rewards'].data_ptr(), out['target_values'].data_ptr(),
            out['target_policies'].data_ptr(), out['total_value'].data_ptr()))
    out['random_actions'] = random_actions
    out['_keepalive'] = powers
    return out
