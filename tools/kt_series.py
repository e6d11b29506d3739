#!/usr/bin/env python3
"""Profiling aid: from a rocprofv3 kernel trace CSV print the durations of the LAST move's launches of a kernel, sampled."""
import csv, sys
path, pat, per_move = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(path)) if pat in r['Kernel_Name']]
rows.sort()
d = [x[1] for x in rows][-per_move:]
step = max(1, len(d) // 16)
print(pat, 'launches', len(rows), 'last move:', ' '.join(f'{i+1}:{d[i]/1e3:.1f}' for i in range(0, len(d), step)), 'mean %.1f us' % (sum(d) / len(d) / 1e3))
