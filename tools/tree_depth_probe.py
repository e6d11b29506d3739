#!/usr/bin/env python3
"""Profiling aid: fused-step time and mean path length per simulation index (injected outputs)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from muzero_amd.tree_pool import TreePool

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'atari_16384x50'
    wl = dict(bench.WORKLOADS[name])
    if len(sys.argv) > 2: wl['N'] = int(sys.argv[2])
    T, A, N = wl['T'], wl['A'], wl['N']
    dev = torch.device('cuda', 0)
    pool = TreePool(T, A, N, wl['players'], wl['discount'], wl['default'], wl['bounds'], dev)
    torch.manual_seed(0)
    logits, values, rewards, noise, u = bench.synthetic_outputs(T, A, N, 1234, wl['reward'], dev)
    pol = torch.softmax(logits, -1).contiguous()
    times = np.zeros(N - 1); ds = np.zeros(N - 1)
    for m in range(3):
        pool.reset(); pool.expand_roots(pol[0]); pool.add_root_noise(noise, 0.25); pool.select()
        evs = []
        prev = pool.counters()
        for s in range(N - 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); pool.step(pol[s + 1], rewards[s + 1], values[s + 1]); e1.record()
            evs.append((e0, e1))
            if m == 2:
                c = pool.counters(); ds[s] = (c[1] - prev[1]) / T; prev = c
        pool.expand_backprop(pol[N], rewards[N], values[N])
        torch.cuda.synchronize()
        if m == 1:
            times = np.array([a.elapsed_time(b) * 1e3 for a, b in evs])
    idx = sorted(set(list(range(0, N - 1, max(1, (N - 1) // 16))) + [N - 2]))
    for s in idx:
        print(f'sim {s+1:4d}: step {times[s]:7.1f} us   mean backed-up path len {ds[s]:.2f}')
    print('mean', times.mean())
main()
