#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations/s of the self-play hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2], the config the metric is quoted on -- "4096 concurrent games"):
TicTacToe-shaped search, T=4096 concurrent games PER GPU (weak scaling: games shard across ranks with no
data-path collective), A=9, 25 simulations/move, 2 players, value bounds (-1,1).  One "step" is one move of
all T games: root expansion + Dirichlet noise + 25 x (select -> recurrent_inference -> expand -> backprop) +
root statistics + action sampling, i.e. T*25 simulations; everything is resident in HBM when the clock starts.

The JSON line carries `roofline` for the dominant kernel (HIP-event timed inside the timed region) and
`cpu_baseline` (the CPU oracle -- a port of the reference algorithm -- timed on this box's host cores on a
bounded sample).  Other BASELINE configs are available with --workload for profiling, not as bench lines.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: T per GPU, A, N, players, discount, default value, known bounds, network kind
    'tictactoe_4096x25': dict(T=4096, A=9, N=25, players=2, discount=1.0, default=0.0, bounds=(-1.0, 1.0),
                              net='tictactoe', reward='tanh'),
    'cartpole_1024x50': dict(T=1024, A=2, N=50, players=1, discount=0.99, default=50.0, bounds=(0.0, 100.0),
                             net='cartpole', reward='uniform'),
    'atari_16384x50': dict(T=16384, A=18, N=50, players=1, discount=0.997, default=0.0, bounds=None, net=None,
                           reward='uniform'),
    'go_2048x800': dict(T=2048, A=362, N=800, players=2, discount=1.0, default=0.0, bounds=(-1.0, 1.0), net=None,
                        reward='zero'),
}


def bytes_per_simulation(A, d, a_root):
    """algorithmic bytes of one simulation with d expanded nodes on its path."""
    return 20 * A * (d + 1) + 28 * d + 32 + 4 * a_root


def synthetic_outputs(T, A, N, seed, reward, device):
    """Injected network outputs: logits~N(0,1), values~tanh(N(0,1)), rewards by mode; generated up
    front on the device so that only the tree kernels are inside the timed region."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    logits = torch.randn(N + 1, T, A, generator=g, device=device, dtype=torch.float32)
    values = torch.tanh(torch.randn(N + 1, T, generator=g, device=device, dtype=torch.float32))
    if reward == 'uniform':
        rewards = torch.rand(N + 1, T, generator=g, device=device, dtype=torch.float32)
    elif reward == 'tanh':
        rewards = torch.tanh(torch.randn(N + 1, T, generator=g, device=device, dtype=torch.float32))
    else:
        rewards = torch.zeros(N + 1, T, device=device, dtype=torch.float32)
    noise = torch.distributions.Dirichlet(torch.ones(A, dtype=torch.float64)).sample((T,)).to(device)
    u = torch.rand(T, generator=g, device=device, dtype=torch.float64)
    return logits, values, rewards, noise.contiguous(), u


class SyntheticTreeMove:
    """One move of T searches with injected outputs (tree kernels only)."""

    def __init__(self, wl, device, seed):
        from muzero_amd.tree_pool import TreePool
        self.wl = wl
        self.pool = TreePool(num_trees=wl['T'], num_actions=wl['A'], max_simulations=wl['N'],
                             num_players=wl['players'], discount=wl['discount'], default_value=wl['default'],
                             known_bounds=wl['bounds'], device=device)
        self.logits, self.values, self.rewards, self.noise, self.u = synthetic_outputs(
            wl['T'], wl['A'], wl['N'], seed, wl['reward'], device)
        self.temperature = torch.ones(wl['T'], dtype=torch.float64, device=device)
        self.events = []

    dominant = 'step_kernel'

    def run(self, timed_events=False):
        pool, N = self.pool, self.wl['N']
        pool.reset()
        pool.expand_roots(self.logits[0], logits=True)
        pool.add_root_noise(self.noise, 0.25)
        pool.select()
        for s in range(N - 1):
            if timed_events:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            pool.step(self.logits[s + 1], self.rewards[s + 1], self.values[s + 1], logits=True)
            if timed_events:
                e1.record()
                self.events.append((e0, e1))
        pool.expand_backprop(self.logits[N], self.rewards[N], self.values[N], logits=True)
        stats = pool.root_stats()
        action = pool.select_action(self.u, self.temperature)
        return stats, action

    def flops_per_launch(self):
        return None


def cpu_baseline_tree(wl, seconds):
    """The CPU oracle (oracle/mcts_oracle.py, a port of the reference's Python MCTS) on ONE host core, same
    shapes and synthetic outputs, for about `seconds` of CPU work.  Tree-only: an upper bound on the
    reference, whose real path adds a batch-1 TensorFlow call per simulation."""
    from oracle import mcts_oracle
    A, N = wl['A'], wl['N']
    rng = np.random.Generator(np.random.PCG64(99))
    sims, moves = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        logits = rng.standard_normal((N + 1, A)).astype(np.float32)
        values = np.tanh(rng.standard_normal(N + 1)).astype(np.float32)
        rewards = rng.random(N + 1).astype(np.float32)
        tree = mcts_oracle.TreeOracle(A, N, wl['players'], wl['discount'], wl['default'], wl['bounds'])
        tree.expand_root(mcts_oracle.masked_policy_f32(logits[0], range(A)), range(A))
        tree.add_root_noise(rng.dirichlet([1.0] * A), 0.25)
        for s in range(N):
            tree.select()
            tree.expand_backprop(mcts_oracle.masked_policy_f32(logits[s + 1], range(A)), rewards[s + 1], values[s + 1])
        tree.select_action(float(rng.random()), 1.0)
        sims += N
        moves += 1
    dt = time.perf_counter() - t0
    return dict(value=sims / dt, unit='simulations/s', cores=1, kind='port',
                sample=f'{moves} single-tree moves x {N} simulations, A={A}, python oracle (tree only), {dt:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='tictactoe_4096x25', choices=sorted(WORKLOADS))
    ap.add_argument('--trees', type=int, default=None, help='override trees per GPU (profiling only)')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
    device = torch.device('cuda', torch.cuda.current_device())

    wl = dict(WORKLOADS[args.workload])
    if args.trees:
        wl['T'] = args.trees
    move = SyntheticTreeMove(wl, device, seed=1234 + rank)      # per-rank seeds: ranks search different games

    for _ in range(args.warmup):
        move.run()
    torch.cuda.synchronize()
    c0 = move.pool.counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        move.run(timed_events=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    c1 = move.pool.counters()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    sims_rank = int(c1[0] - c0[0])
    sum_d = int(c1[1] - c0[1])
    assert sims_rank == args.steps * wl['T'] * wl['N'], (sims_rank, args.steps, wl['T'], wl['N'])
    total_sims = sims_rank * world
    kernel_ms = [a.elapsed_time(b) for a, b in move.events]
    avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
    # algorithmic bytes of one dominant-kernel launch: T simulations with the measured mean path length
    mean_d = sum_d / sims_rank
    bytes_launch = wl['T'] * bytes_per_simulation(wl['A'], mean_d, wl['A'])
    achieved = bytes_launch / avg_kernel_s / 1e9

    if rank == 0:
        out = {
            'metric': 'mcts_simulations_per_sec', 'value': total_sims / elapsed, 'unit': 'simulations/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': args.workload, 'trees_per_gpu': wl['T'], 'actions': wl['A'],
                       'simulations_per_move': wl['N'], 'network': 'injected outputs (tree kernels only)',
                       'sharding': f'{world} x independent game shards, no collective'},
            'roofline': {'bound': 'hbm', 'kernel': move.dominant, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                         'avg_launch_us': avg_kernel_s * 1e6, 'bytes_per_launch': bytes_launch,
                         'mean_path_len': mean_d},
        }
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_tree(wl, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
