#!/usr/bin/env python3
"""Profiling aid: mean/min/max duration per kernel from a rocprofv3 kernel-trace CSV (name filter optional)."""
import collections, csv, sys
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'anonymous')
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if pat in r['Kernel_Name'] and 'at::' not in r['Kernel_Name']:
        agg[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(agg.items()):
    print('%-44s n=%5d mean %8.1f us  min %7.1f  max %7.1f' % (k[:44], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
