for v in samejob noepi nocommit now; do
  MZ_LIB=$PWD/tools/micro/libmzhip_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline > gpurun_out/b_$v.log 2>&1 || exit 1
  tail -1 gpurun_out/b_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['avg_launch_us'], d['head_and_tree_step_kernel_us'])"
done
timeout -k 10 300 python -u -m pytest tests/test_net_gpu.py tests/test_selfplay_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_split.log 2>&1 || { tail -20 gpurun_out/t_split.log; exit 1; }
tail -1 gpurun_out/t_split.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/b_split.log 2>&1
tail -1 gpurun_out/b_split.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('product', d['value'], d['roofline']['avg_launch_us'], d['head_and_tree_step_kernel_us'])"
