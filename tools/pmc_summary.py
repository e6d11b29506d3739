"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel (name filter optional)."""
import collections, csv, sys
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if pat in r['Kernel_Name']:
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print('   %-36s %16.1f  (n=%d)' % (c, sum(x) / len(x), len(x)))
