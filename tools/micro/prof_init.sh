#!/bin/bash
# per-kernel stats of the headline bench for a list of experiment libraries: tools/micro/prof_init.sh <name>...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for n in "$@"; do
  export MZ_LIB=$GRAFT_REPO_ROOT/tools/micro/libmzhip_$n.so
  [ "$n" = head ] && unset MZ_LIB
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$n -o out --output-format csv -- python3 bench.py --no-extra-legs --no-cpu-baseline --steps 30 ${BENCH_ARGS} > gpurun_out/ab/prof_$n.log 2>&1 || exit 1
  echo "== $n"
  python3 - gpurun_out/prof_$n/out_kernel_stats.csv <<PY
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if r[0]!="Name" and ("initial" in r[0] or "head_kernel" in r[0] or "expand_roots" in r[0] or "root_finish" in r[0]): print(r[0][40:90], r[1], r[3], r[5], r[6])
PY
done
