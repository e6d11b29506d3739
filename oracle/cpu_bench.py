#!/usr/bin/env python3
"""One worker of bench.py's cpu_baseline leg -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

Runs the CPU port of the reference's per-game self-play search on ONE core for about --seconds and prints one JSON line:
the Python tree oracle (oracle/mcts_oracle.py, the restatement of agent.py:154-232 pinned by the reference fixtures) calling
the numpy float32 network oracle (oracle/net_oracle.py) at batch 1 per simulation like agent.py:204-207 -- or the tree
alone against random outputs for --net none.  bench.py starts one of these per host core (no torch import here, so a worker
starts in a fraction of a second and the processes share nothing)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mcts_oracle, net_oracle  # noqa: E402


def run(A, N, players, discount, default, bounds, net_kind, seconds, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    net = None
    if net_kind != 'none':
        net = net_oracle.make_oracle(net_kind, net_oracle.random_weights(net_kind, 7), dtype=np.float32)
    sims, moves = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        tree = mcts_oracle.TreeOracle(A, N, players, discount, default, bounds)
        if net is not None:
            if net_kind == 'tictactoe':
                obs = rng.integers(0, 2, size=(1, 27)).astype(np.float32)
            elif net_kind == 'atari':
                obs = rng.standard_normal((1, 256)).astype(np.float32)
            else:
                obs = (rng.uniform(-1, 1, size=(1, 4)) * [2.4, 3.0, 0.21, 3.0]).astype(np.float32)
            hidden, _, logits, _ = net.initial_inference(obs)
            hiddens = [hidden]
            tree.expand_root(mcts_oracle.masked_policy_f32(logits[0], range(A)), range(A))
        else:
            logits = rng.standard_normal((N + 1, A)).astype(np.float32)
            values = np.tanh(rng.standard_normal(N + 1)).astype(np.float32)
            rewards = rng.random(N + 1).astype(np.float32)
            tree.expand_root(mcts_oracle.masked_policy_f32(logits[0], range(A)), range(A))
        tree.add_root_noise(rng.dirichlet([1.0] * A), 0.25)
        for s in range(N):
            parent, action = tree.select()
            if net is not None:
                h, r, v, lg, _ = net.recurrent_inference(hiddens[parent], np.array([action]))
                hiddens.append(h)
                tree.expand_backprop(mcts_oracle.masked_policy_f32(lg[0], range(A)), r[0], v[0])
            else:
                tree.expand_backprop(mcts_oracle.masked_policy_f32(logits[s + 1], range(A)), rewards[s + 1], values[s + 1])
        tree.select_action(float(rng.random()), 1.0)
        sims += N
        moves += 1
    return sims, moves, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--actions', type=int, required=True)
    ap.add_argument('--simulations', type=int, required=True)
    ap.add_argument('--players', type=int, default=1)
    ap.add_argument('--discount', type=float, default=1.0)
    ap.add_argument('--default-value', type=float, default=0.0)
    ap.add_argument('--bounds', type=float, nargs=2, default=None)
    ap.add_argument('--net', default='none', choices=['none', 'tictactoe', 'cartpole', 'atari'])
    ap.add_argument('--seconds', type=float, default=10.0)
    ap.add_argument('--seed', type=int, default=99)
    ap.add_argument('--wait-for-go', action='store_true',
                    help='start measuring only when a line arrives on stdin (bench.py starts its workers before it touches '
                         'the GPU and releases them after the GPU timed region)')
    a = ap.parse_args()
    if a.wait_for_go:
        print('ready', flush=True)
        if not sys.stdin.readline():
            return                                   # the parent went away
    sims, moves, dt = run(a.actions, a.simulations, a.players, a.discount, a.default_value,
                          None if a.bounds is None else tuple(a.bounds), a.net, a.seconds, a.seed)
    print(json.dumps(dict(simulations=sims, moves=moves, seconds=dt)))


if __name__ == '__main__':
    main()
