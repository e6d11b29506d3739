"""Two ranks on ONE GPU (a rehearsal box): do asynchronous all_gather_into_tensor calls of device tensors, waited for one
iteration later with device work in between (the pattern of selfplay.TrajectoryGather), complete under gloo?
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/micro/dist_probe.py gloo [iterations]
(nccl refuses two ranks on one device.)"""
import faulthandler, sys, torch, torch.distributed as dist
faulthandler.dump_traceback_later(40, exit=True)
backend, iters = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.cuda.set_device(0)
dist.init_process_group(backend)
rank, world = dist.get_rank(), dist.get_world_size()
send = torch.zeros(4096, 11, device='cuda')
recv = torch.empty(world, 4096, 11, device='cuda')
a = torch.randn(2048, 2048, device='cuda')
work = None
for i in range(iters):
    b = a @ a                                        # device work between the calls
    if work is not None:
        work.wait()
    send.fill_(float(rank + 10 * i))
    work = dist.all_gather_into_tensor(recv.view(-1), send.view(-1), async_op=True)
    if i % 10 == 0:
        print(backend, 'issued', i, 'rank', rank, flush=True)
work.wait()
torch.cuda.synchronize()
print(backend, 'done', rank, recv[:, 0, 0].tolist(), flush=True)
dist.barrier()
dist.destroy_process_group()
