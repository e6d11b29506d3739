#!/usr/bin/env python3
"""Profiling aid: continuous-refill self-play (SelfPlayEngine.play_continuous) at the headline size, per-move wall time
next to the search alone.  python tools/refill_probe.py [games] [check_every]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd.games import tictactoe
from muzero_amd.selfplay import SelfPlayEngine

G, N = 4096, 25
total = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * G
every = int(sys.argv[2]) if len(sys.argv) > 2 else 32
config = tictactoe.make_config(num_simulations=N)
net = config.network_config.network_class(config, seed=5)
engine = SelfPlayEngine(config, net, G, seed=11)
engine.play_continuous(G, check_every=every)
engine.env.reset()
obs, legal, n_legal = engine.env.observe()
noise = engine._dirichlet(n_legal)
u = torch.rand(G, dtype=torch.float64, device=engine.device)
temp = torch.ones(G, dtype=torch.float64, device=engine.device)
for _ in range(3):
    engine.search.run(obs, noise, u, temp, legal, n_legal)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40):
    engine.search.run(obs, noise, u, temp, legal, n_legal)
torch.cuda.synchronize(); per_move = (time.perf_counter() - t0) / 40
print(f'search alone: {per_move * 1e3:.4f} ms per move', flush=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hist = engine.play_continuous(total, check_every=every)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m = hist.metadata['moves_played']
    print(f'refill: {m} moves, {dt / m * 1e3:.4f} ms per move = {per_move / (dt / m):.3f} of the per-move rate', flush=True)
