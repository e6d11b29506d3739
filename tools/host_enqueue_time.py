import sys, time, torch
sys.path.insert(0, '.')
import bench
wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'tictactoe_4096x25'])
dev = torch.device('cuda', 0)
move = bench.NetworkMove(wl, dev, 1234, 1)
for _ in range(5): move.run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); move.run(); ts.append(time.perf_counter() - t0)
t1 = time.perf_counter(); torch.cuda.synchronize(); drain = time.perf_counter() - t1
print('host enqueue per move: median %.0f us min %.0f max %.0f; drain after 20 moves %.1f ms' % (sorted(ts)[10]*1e6, min(ts)*1e6, max(ts)*1e6, drain*1e3))
# one move at a time (GPU idle at start): enqueue + wait
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); move.run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('one move, synchronised: median %.0f us' % (sorted(ts)[5]*1e6))
