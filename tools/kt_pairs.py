#!/usr/bin/env python3
"""Profiling aid: from a rocprofv3 kernel trace (out_kernel_trace.csv), the idle time between consecutive kernels by
(previous kernel -> next kernel) pair: what a launch boundary costs on the stream.
    python tools/kt_pairs.py <out_kernel_trace.csv>"""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
gaps = collections.defaultdict(list)
rows = rows[len(rows) // 3:]
for a, b in zip(rows, rows[1:]):
    gaps[(name(a), name(b))].append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
    print(f'{k[0]:40s} -> {k[1]:40s} n={len(v):5d} median gap {statistics.median(v):6.2f} us  min {min(v):6.2f}')
