#!/usr/bin/env python3
"""Profiling aid: where a wavefront of the CartPole run kernel (cartpole16_run_kernel: a move's simulations in one launch)
spends its clocks (experiment build with -DMZ_TREE_TRACE).
    MZ_LIB=tools/micro/libmzhip_trace.so python tools/run_trace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from muzero_amd import _lib
NAMES = ['prologue', 'expand + backup', 'backup: expand', 'backup: chain+slots', 'backup: root+minmax',
         'setup + gather', 'conv, dense, reward dense, pred 1', 'pred 2, value head', 'select: root stats',
         'select', 'select: other level', 'softmax + outputs']
wl = dict(bench.WORKLOADS['cartpole_1024x50']); wl['T'] = int(os.environ.get('TRACE_T', wl['T']))
dev = torch.device('cuda', 0)
move = bench.NetworkMove(wl, dev, 1234, 1)
lib = _lib.load()
W, PH = 1 << 15, 12
out = np.zeros((W, 2 * PH), np.uint32)
read = lambda: lib.mz_debug_net_trace(ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(out.nbytes))
move.run(); torch.cuda.synchronize(); read()
move.run(); torch.cuda.synchronize(); read()
used = out[:, PH:].sum(1) > 0
clk, cnt = out[used, :PH].astype(np.float64), out[used, PH:].astype(np.float64)
w0 = np.arange(out.shape[0])[used] % 4 == 0          # wavefront 0 of each workgroup runs the tree step
print(int(used.sum()), 'wavefronts; shader clocks per wavefront and MOVE (%d simulations)' % wl['N'])
for sel, nm in ((w0, 'wave 0 (network + trees)'), (~w0, 'waves 1-3 (network)')):
    print(nm)
    for i, n in enumerate(NAMES):
        c, k = clk[sel, i], cnt[sel, i]
        if k.sum() == 0: continue
        print(f'  {n:36s} mean {c.mean():9.0f}  x{k.mean():6.1f}  -> {c.sum() / k.sum():7.0f} each')
    print('  total %9.0f' % clk[sel].sum(1).mean())
