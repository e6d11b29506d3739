#!/usr/bin/env python3
"""Profiling aid: step time of the headline workload over the first seconds of the FIRST GPU process on a fresh box
(the bench line there has come out ~1.7x slow now and then while kernel durations were normal).
    python tools/first_process_probe.py [seconds]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
every = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # record HIP events on every n-th step (0 = never)
wl = dict(bench.WORKLOADS['tictactoe_4096x25'])
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
move = bench.NetworkMove(wl, dev, 1234, 1)
t_start = time.perf_counter()
group = 10
while time.perf_counter() - t_start < seconds:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(group):
        move.run(timed_events=bool(every) and i % every == 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f't={t0 - t_start:6.3f}s  {group} steps: enqueue {1e3 * (t1 - t0) / group:.3f} ms/step, done {1e3 * (t2 - t0) / group:.3f} ms/step',
          flush=True)
