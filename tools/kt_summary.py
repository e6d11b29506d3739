#!/usr/bin/env python3
"""profiles/*_kernel_medians_<workload>.json from a rocprofv3 --kernel-trace CSV: calls, median / mean / min duration per
kernel, plus the ms_per_step that the profiled bench run itself printed (so kernel times and step time come from one run).

    python tools/kt_summary.py <workload> <out_kernel_trace.csv> <bench stdout of the profiled run> <out.json> "<command>"
"""
import collections, csv, json, re, statistics, sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0]


def main():
    workload, trace, log, out, command = sys.argv[1:6]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if 'at::' in r['Kernel_Name'] or '__amd' in r['Kernel_Name']:
            continue
        agg[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    ms = None
    for line in open(log):
        if line.startswith('{'):
            try:
                ms = json.loads(line).get('ms_per_step')
            except ValueError:
                pass
    kernels = {k: {'calls': len(v), 'median_us': statistics.median(v), 'mean_us': sum(v) / len(v), 'min_us': min(v),
                   'max_us': max(v)} for k, v in sorted(agg.items())}
    json.dump({'workload': workload, 'command': command, 'ms_per_step': ms, 'kernels': kernels}, open(out, 'w'), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['calls'] * kv[1]['mean_us'])[:8]:
        print('%-44s n=%5d median %8.1f us  mean %8.1f  min %7.1f' % (k[:44], v['calls'], v['median_us'], v['mean_us'], v['min_us']))


main()
