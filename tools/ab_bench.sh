#!/bin/bash
# A/B of library builds on the bench workload, in one GPU call: tools/ab_bench.sh <workload> <lib>...   (profiling aid)
# prints the step time and the dominant kernel's average launch time per library, two passes.
WL=$1; shift
for pass in 1 2; do
  for lib in "$@"; do
    MZ_LIB=$lib timeout -k 10 120 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-44s %8.1f M/s  step %.4f ms  kernel %.2f us' % ('$lib', d['value']/1e6, d['ms_per_step'], r['avg_launch_us']))" || exit 1
  done
done
