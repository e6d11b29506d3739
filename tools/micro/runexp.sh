for v in nosync noepi nocommit now; do
  MZ_LIB=$PWD/tools/micro/libmzhip_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline > gpurun_out/b_$v.log 2>&1 || exit 1
  tail -1 gpurun_out/b_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['avg_launch_us'], d['head_and_tree_step_kernel_us'])"
done
