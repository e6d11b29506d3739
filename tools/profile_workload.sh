#!/bin/bash
# On the GPU box: bench line + rocprofv3 kernel stats + the two PMC passes for one workload, written to
# gpurun_out/prof_<workload>/ (copy what should be judged into profiles/).
#   tools/profile_workload.sh <workload> [extra bench.py flags...]
set -e -o pipefail
W=$1; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_$W
mkdir -p $O
BENCH="bench.py --workload $W $*"
timeout -k 10 300 python3 $BENCH > $O/bench.json 2> $O/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o out --output-format csv -- python3 $BENCH --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o out --output-format csv -- python3 $BENCH --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o out --output-format csv -- python3 $BENCH --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
cp $(find $O/kt -name out_kernel_stats.csv) $O/kernel_stats.csv
python3 tools/kt_summary.py $W $(find $O/kt -name out_kernel_trace.csv) $O/kt.log $O/kernel_medians.json \
    "rocprofv3 --kernel-trace --stats -- python3 $BENCH --steps 5 --warmup 2 --no-cpu-baseline"
python3 tools/pmc_to_json.py $W $(find $O/pmc_f -name out_counter_collection.csv) $(find $O/pmc_w -name out_counter_collection.csv) \
    $O/hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 $BENCH --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $O/pmc_f $O/pmc_w $O/kt
tail -c 600 $O/bench.json
