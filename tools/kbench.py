#!/usr/bin/env python3
"""Kernel A/B harness (profiling aid): times mz_infer_recurrent of several builds of libmzhip.so in ONE process, rounds
interleaved, so that box-to-box and run-to-run variance (several per cent on this pool) cancels out of the comparison.

    python tools/kbench.py --kind atari --batch 16384 --libs muzero_amd/libmzhip.so tools/micro/libmzhip_x.so
    python tools/kbench.py --kind tree --actions 18 --sims 50 --batch 16384 --libs ...      (tree-only injected move)

Prints median / min microseconds per call (network kinds: one recurrent inference over the batch with the pool gather;
tree: one whole injected move divided by its simulations)."""
import argparse
import ctypes
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muzero_amd import _lib as L  # noqa: E402  (struct definitions and prototypes only)
from oracle import net_oracle  # noqa: E402  (random weights: measurement infrastructure)

KINDS = {'tictactoe': 1, 'cartpole': 2, 'atari': 3}


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in L.PROTOTYPES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
    return lib


def check(lib, status):
    assert status == 0, lib.mz_last_error()


class NetCase:
    def __init__(self, lib, kind, B, seed=5):
        self.lib, self.B = lib, B
        k = KINDS[kind]
        flat = net_oracle.random_weights(kind, 7)
        self.h = ctypes.c_void_p()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib, lib.mz_net_create(k, flat.ctypes.data, flat.size, 0, stream, ctypes.byref(self.h)))
        H, A = lib.mz_net_hidden_size(k), {1: 9, 2: 2, 3: 18}[k]
        g = torch.Generator(device='cuda').manual_seed(seed)
        self.pool = torch.rand(4, B, H, generator=g, device='cuda')
        self.parent = torch.randint(0, 3, (B,), generator=g, device='cuda', dtype=torch.int32)
        self.action = torch.randint(0, A, (B,), generator=g, device='cuda', dtype=torch.int32)
        self.reward, self.value = torch.empty(B, device='cuda'), torch.empty(B, device='cuda')
        self.logits = torch.empty(B, A, device='cuda')
        self.stride = B * H

    def call(self):
        check(self.lib, self.lib.mz_infer_recurrent(self.h, self.B, self.pool.data_ptr(), self.parent.data_ptr(), self.stride,
                                                    self.action.data_ptr(), self.pool[3].data_ptr(), self.reward.data_ptr(),
                                                    self.value.data_ptr(), self.logits.data_ptr(), None))

    def outputs(self):
        return torch.cat([self.pool[3].flatten(), self.reward, self.value, self.logits.flatten()])

    units = 1


class TreeCase:
    def __init__(self, lib, T, A, N, bounds, players, discount, seed=5, lanes=0):
        self.lib, self.T, self.A, self.N = lib, T, A, N
        table = np.array([2 * np.log(n + 1) for n in range(N + 2)], np.float64)
        self._table = table
        cfg = L.PoolConfig(T, A, N, players, int(bounds is not None), lanes, discount, 0.0,
                           bounds[0] if bounds else 0.0, bounds[1] if bounds else 0.0, table.ctypes.data)
        self.h = ctypes.c_void_p()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib, lib.mz_pool_create(ctypes.byref(cfg), stream, ctypes.byref(self.h)))
        g = torch.Generator(device='cuda').manual_seed(seed)
        self.logits = torch.randn(N + 1, T, A, generator=g, device='cuda')
        self.values = torch.tanh(torch.randn(N + 1, T, generator=g, device='cuda'))
        self.rewards = torch.rand(N + 1, T, generator=g, device='cuda')
        gam = torch._standard_gamma(torch.ones(T, A, dtype=torch.float64, device='cuda'), generator=g)
        self.noise = (gam / gam.sum(1, keepdim=True)).contiguous()
        self.parent = torch.empty(T, dtype=torch.int32, device='cuda')
        self.action = torch.empty(T, dtype=torch.int32, device='cuda')
        self.visits = torch.empty(T, A, dtype=torch.int32, device='cuda')
        self.units = N

    def call(self):
        check(self.lib, self.lib.mz_search_move_injected(self.h, self.N, self.logits.data_ptr(), 1, self.rewards.data_ptr(),
                                                         self.values.data_ptr(), None, None, self.noise.data_ptr(), 0.25,
                                                         self.parent.data_ptr(), self.action.data_ptr(), None))

    def outputs(self):
        check(self.lib, self.lib.mz_root_stats(self.h, self.visits.data_ptr(), None, None, None))
        return self.visits.flatten().float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kind', default='atari', choices=['tictactoe', 'cartpole', 'atari', 'tree'])
    ap.add_argument('--batch', type=int, default=16384)
    ap.add_argument('--actions', type=int, default=18)
    ap.add_argument('--sims', type=int, default=50)
    ap.add_argument('--bounds', type=float, nargs=2, default=None)
    ap.add_argument('--players', type=int, default=1)
    ap.add_argument('--discount', type=float, default=0.997)
    ap.add_argument('--libs', nargs='+', default=[os.path.join(ROOT, 'muzero_amd', 'libmzhip.so')])
    ap.add_argument('--lanes', type=int, nargs='+', default=[0], help='tree: lanes per tree, one case per value')
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--iters', type=int, default=40)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    cases = []
    for path in a.libs:
        lib = bind(path)
        if a.kind == 'tree':
            for lanes in a.lanes:
                cases.append((f'{os.path.basename(path)} lanes={lanes}',
                              TreeCase(lib, a.batch, a.actions, a.sims, a.bounds, a.players, a.discount, lanes=lanes)))
        else:
            cases.append((os.path.basename(path), NetCase(lib, a.kind, a.batch)))
    for _, c in cases:
        for _ in range(3):
            c.call()
    torch.cuda.synchronize()
    ref = cases[0][1].outputs().clone()
    times = {name: [] for name, _ in cases}
    for _ in range(a.rounds):
        for name, c in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                c.call()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 1e3 / a.iters / c.units)
    for name, c in cases:
        same = bool(torch.equal(c.outputs(), ref))
        t = times[name]
        print(f'{name:40s} median {statistics.median(t):8.2f} us  min {min(t):8.2f} us  '
              f'{"outputs == first lib" if same else "OUTPUTS DIFFER from the first lib"}')


if __name__ == '__main__':
    main()
