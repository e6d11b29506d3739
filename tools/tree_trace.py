#!/usr/bin/env python3
"""Profiling aid: phase times inside the tree kernels from an experiment build of the library.

    hipcc ... -DMZ_TREE_TRACE -> tools/micro/libmzhip_trace.so (tools/micro/build_trace.sh), then
    MZ_LIB=tools/micro/libmzhip_trace.so python tools/tree_trace.py <workload> [lanes] [run]
"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from muzero_amd import _lib
from muzero_amd.tree_pool import TreePool

NAMES = {1: 'backup: prefetch', 2: 'backup: expand', 3: 'backup: chain+slots', 4: 'backup: root+minmax',
         8: 'select: root stats', 9: 'select: root level', 10: 'select: other level'}

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'atari_16384x50'
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    wl = dict(bench.WORKLOADS[name])
    T, A, N = wl['T'], wl['A'], wl['N']
    dev = torch.device('cuda', 0)
    lib = _lib.load()
    pool = TreePool(T, A, N, wl['players'], wl['discount'], wl['default'], wl['bounds'], dev, lanes_per_tree=lanes)
    logits, values, rewards, noise, u = bench.synthetic_outputs(T, A, N, 1234, wl['reward'], dev)
    pol = torch.softmax(logits, -1).contiguous()
    W, PH = 1 << 15, 12
    out = np.zeros((W, 2 * PH), np.uint32)
    read = lambda: lib.mz_debug_tree_trace(ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(out.nbytes))
    run = len(sys.argv) > 3 and sys.argv[3] == 'run'      # the one-launch run (run_injected_kernel), logits like bench.py
    for m in range(2 if not run else 0):
        pool.reset(); pool.expand_roots(pol[0]); pool.add_root_noise(noise, 0.25); pool.select()
        read()
        for s in range(N - 1):
            pool.step(pol[s + 1], rewards[s + 1], values[s + 1])
        read()
        pool.expand_backprop(pol[N], rewards[N], values[N])
    launches = N - 1
    if run:
        pool.search_injected(logits, rewards, values, noise=noise, frac=0.25, logits=True); torch.cuda.synchronize(); read()
        pool.search_injected(logits, rewards, values, noise=noise, frac=0.25, logits=True); torch.cuda.synchronize(); read()
        launches = N
    used = out[:, PH + 1] > 0
    nw = int(used.sum())
    print(f'{name} lanes={lanes}: {nw} wavefronts, {launches} launches; shader clocks per wavefront and launch')
    clk, cnt = out[used, :PH].astype(np.float64), out[used, PH:].astype(np.float64)
    for ph, nm in NAMES.items():
        if cnt[:, ph].sum():
            per_wave = clk[:, ph] / launches
            print(f'  {nm:22s} mean {per_wave.mean():8.0f}  max {per_wave.max():8.0f}   {clk[:, ph].sum() / cnt[:, ph].sum():7.0f} clk each, '
                  f'x{cnt[:, ph].mean() / launches:5.2f} (max {cnt[:, ph].max() / launches:5.2f})')
    tot = clk.sum(1) / launches
    print(f'  total                  mean {tot.mean():8.0f}  max {tot.max():8.0f}')
main()
