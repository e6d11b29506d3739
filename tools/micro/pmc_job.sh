#!/bin/bash
# SQ / TCP / TA counters of the TicTacToe job kernel (profiling aid), one pass per run
set -e -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_job; mkdir -p $O
B="bench.py ${BENCH_ARGS} --steps 2 --warmup 1 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o out --output-format csv -- python3 $B > $O/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq2 -o out --output-format csv -- python3 $B > $O/sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum -d $O/tcp -o out --output-format csv -- python3 $B > $O/tcp.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
for d in ('sq','sq2','tcp'):
    f = glob.glob(f'gpurun_out/pmc_job/{d}/**/out_counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if os.environ.get("KERNEL", "ttt_job_kernel") in r["Kernel_Name"]:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in acc.items(): print(d, k, v / n)
PY
rm -rf $O/sq $O/sq2 $O/tcp
