#!/usr/bin/env python3
"""Profiling aid: where the workgroups of ttt_run_kernel (a TicTacToe move's simulations in one launch) spend their time,
per phase of a simulation and per job (experiment build with -DMZ_RUN_TRACE):
    tools/micro/build_exp.sh runtrace -DMZ_RUN_TRACE      # in the container
    MZ_LIB=tools/micro/libmzhip_runtrace.so python tools/ttt_run_trace.py [run_mode]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from muzero_amd import _lib
NAMES = ['job', 'meet A (wait for the row block\'s jobs)', 'head', 'expand + backup', 'select', 'meet B (wait for the selections)']
wl = dict(bench.WORKLOADS['tictactoe_4096x25']); wl['T'] = int(os.environ.get('TRACE_T', wl['T']))
dev = torch.device('cuda', 0)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
move = bench.NetworkMove(wl, dev, 1234, 1, run_mode=mode)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = np.zeros((2048, 8), np.uint64)
read = lambda: lib.mz_debug_run_trace(ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(out.nbytes))
for _ in range(3):
    move.run()
torch.cuda.synchronize(); read()
K = 10
for _ in range(K):
    move.run()
torch.cuda.synchronize(); assert read() == 0
nb = 32 * (((wl['T'] + 63) // 64 + 7) // 8)
t = out[:nb, :6].astype(np.float64) * 10.0 / (K * wl['N'])        # ns per simulation
job = (np.arange(nb) >> 3) & 3
print(f"T={wl['T']} N={wl['N']} mode={mode}: ns per simulation and workgroup (mean over {K} moves)")
for j in range(4):
    sel = job == j
    print(f' job {j}: ' + '  '.join(f'{n.split(" (")[0]} {t[sel, i].mean():7.0f}' for i, n in enumerate(NAMES)) +
          f'   total {t[sel].sum(1).mean():7.0f}')
print(' all  : ' + '  '.join(f'{n.split(" (")[0]} {t[:, i].mean():7.0f}' for i, n in enumerate(NAMES)) +
      f'   total {t.sum(1).mean():7.0f}')
