#!/bin/bash
# A/B of library builds on any bench.py command line in one GPU call: tools/micro/ab_any.sh "<bench flags>" <lib>...
FLAGS=$1; shift
for pass in 1 2 3; do
  for lib in "$@"; do
    MZ_LIB=$lib timeout -k 10 120 python bench.py $FLAGS --no-cpu-baseline 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-36s %7.2f M/s  frac %.4f  %s us a simulation' % ('$lib', d['value']/1e6, r['frac'], r.get('us_per_simulation')))" || exit 1
  done
done
