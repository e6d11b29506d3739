#!/usr/bin/env python3
"""Profiling aid: where a wavefront of atari_kernel spends its clocks (experiment build with -DMZ_TREE_TRACE -DMZ_JOB_TRACE).
    MZ_LIB=tools/micro/libmzhip_trace.so python tools/atari_trace.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kbench
lib = kbench.bind(os.environ['MZ_LIB'])
lib.mz_debug_net_trace.restype = ctypes.c_int
lib.mz_debug_net_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
torch.cuda.set_device(0)
B = int(os.environ.get('TRACE_B', 16384))
case = kbench.NetCase(lib, 'atari', B)
NAMES = ['prologue', 'two stages', 'epilogue', 'barrier', 'hidden out', 'pred conv+dense', 'heads+outputs']
W, PH = 1 << 15, 12
out = np.zeros((W, 2 * PH), np.uint32)
read = lambda: lib.mz_debug_net_trace(ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(out.nbytes))
case.call(); torch.cuda.synchronize(); read()
L = 10
for _ in range(L): case.call()
torch.cuda.synchronize(); read()
used = out[:, PH] > 0
clk, cnt = out[used, :PH].astype(np.float64) / L, out[used, PH:].astype(np.float64) / L
print(int(used.sum()), 'wavefronts; shader clocks per wavefront and launch')
for i, nm in enumerate(NAMES):
    print(f'  {nm:16s} mean {clk[:, i].mean():8.0f}  max {clk[:, i].max():8.0f}   x{cnt[:, i].mean():5.1f}  -> {clk[:, i].sum() / max(cnt[:, i].sum(), 1):7.0f} each')
print('  total            mean %8.0f  max %8.0f' % (clk.sum(1).mean(), clk.sum(1).max()))
