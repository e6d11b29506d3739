#!/usr/bin/env python3
"""Profiling aid: where a wavefront of the TicTacToe job kernel spends its clocks (experiment build with -DMZ_TREE_TRACE).
    MZ_LIB=tools/micro/libmzhip_trace.so python tools/job_trace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from muzero_amd import _lib
TREE = os.environ.get('TRACE_TREE')      # build with -DMZ_TREE_TRACE only: stamps of the fused head + tree-step kernel
NAMES = ['head', 'backup: prefetch', 'backup: expand', 'backup: chain+slots', 'backup: root+minmax', '', '', '',
         'select: root stats', 'select: root level', 'select: other level'] if TREE else ['prologue', 'stage set-up', 'MFMA section', 'commit / stores', 'barrier', 'after barrier', 'epilogue math']
NAMES = [n for n in NAMES]
wl = dict(bench.WORKLOADS[os.environ.get('TRACE_WL', 'tictactoe_4096x25')]); wl['T'] = int(os.environ.get('TRACE_T', wl['T']))
dev = torch.device('cuda', 0)
move = bench.NetworkMove(wl, dev, 1234, 1)
lib = _lib.load()
W, PH = 1 << 15, 12
out = np.zeros((W, 2 * PH), np.uint32)
read = lambda: lib.mz_debug_net_trace(ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(out.nbytes))
move.run(); torch.cuda.synchronize(); read()
move.run(); torch.cuda.synchronize(); read()
launches = wl['N']
used = out[:, PH:].sum(1) > 0
clk, cnt = out[used, :PH].astype(np.float64) / launches, out[used, PH:].astype(np.float64) / launches
print(int(used.sum()), 'wavefronts; shader clocks per wavefront and launch')
for i, nm in enumerate(NAMES):
    if not nm: continue
    print(f'  {nm:14s} mean {clk[:, i].mean():8.0f}  max {clk[:, i].max():8.0f}   x{cnt[:, i].mean():5.1f}  -> {clk[:, i].sum() / max(cnt[:, i].sum(), 1):7.0f} each')
print('  total          mean %8.0f  max %8.0f' % (clk.sum(1).mean(), clk.sum(1).max()))
if not TREE:
    # per job (blockIdx = 8 * (4 * rbhi + job) + xcd lane; 8 wavefronts per workgroup): which job is the long pole
    idx = np.nonzero(used)[0]
    job = ((idx // 8) >> 3) & 3
    tot = clk.sum(1)
    for j in range(4):
        sel = job == j
        if sel.any():
            print(f'  job {j}: {int(sel.sum())} wavefronts, total mean {tot[sel].mean():8.0f}  max {tot[sel].max():8.0f}; '
                  + ' '.join(f'{NAMES[i][:8]}={clk[sel, i].mean():.0f}' for i in range(6)))
