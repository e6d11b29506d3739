#!/usr/bin/env python3
"""Profiling aid: from a rocprofv3 kernel trace (out_kernel_trace.csv) of a self-play run, the time between the end of one
move's search (root_finish_kernel) and the start of the next (the initial-inference kernel), and what ran in between.
    python tools/kt_gaps.py <out_kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
gaps, between = [], collections.Counter()
busy = collections.Counter()
last_end, inside = None, False
for r in rows:
    n = name(r)
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'root_finish_kernel' in n:
        last_end, inside, cur = e, True, []
        continue
    if inside and ('initial_kernel' in n or 'cartpole16_kernel<true' in n or 'atari_kernel<true' in n):
        gaps.append((s - last_end, cur))
        inside = False
        continue
    if inside:
        cur.append((n, e - s))
gaps = gaps[len(gaps) // 4:]                      # steady state
if gaps:
    import statistics
    print(f'{len(gaps)} turnovers: median gap {statistics.median(g for g, _ in gaps) / 1e3:.1f} us, mean {statistics.mean(g for g, _ in gaps) / 1e3:.1f} us')
    agg = collections.defaultdict(list)
    for _, cur in gaps:
        for n, d in cur:
            agg[n].append(d)
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f'  {n:60s} x{len(v) / len(gaps):4.1f} per turnover, {statistics.mean(v) / 1e3:6.1f} us each')
