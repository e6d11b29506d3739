#!/bin/bash
# A/B of library builds on the Go-shaped tree-only workload in one GPU call: tools/micro/ab_go.sh <lib>...
for pass in 1 2 3; do
  for lib in "$@"; do
    MZ_LIB=$lib timeout -k 10 120 python bench.py --workload ${WL:-go_2048x800} --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-36s %7.2f M/s  frac %.4f  %.2f us a simulation' % ('$lib', d['value']/1e6, r['frac'], r['us_per_simulation']))" || exit 1
  done
done
