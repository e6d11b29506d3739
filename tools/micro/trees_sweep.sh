#!/bin/bash
# job-kernel launch time against the number of games (profiling aid): 2048 games = one workgroup per CU
for t in 1024 2048 4096 8192; do
  timeout -k 10 120 python bench.py --trees $t --no-cpu-baseline > gpurun_out/b_t$t.log 2>&1 || exit 1
  tail -1 gpurun_out/b_t$t.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($t, d['value'], d['roofline']['avg_launch_us'], d['head_and_tree_step_kernel_us'])"
done
