#!/usr/bin/env python3
"""Soak run (not a bench line): whole self-play games at the benchmark sizes through SelfPlayEngine, several batches in a
row; prints games/s and checks the histories' basic invariants.  python tools/soak_selfplay.py [batches]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd.games import cartpole, tictactoe
from muzero_amd.selfplay import SelfPlayEngine

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, mod, G, N, max_moves in (('tictactoe', tictactoe, 4096, 25, None), ('cartpole', cartpole, 1024, 50, 60)):
    config = mod.make_config(num_simulations=N)
    net = config.network_config.network_class(config, seed=5)
    engine = SelfPlayEngine(config, net, G, seed=11)
    for b in range(batches):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hist = engine.play_games() if max_moves is None else engine.play_games(max_moves=max_moves)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        lengths = hist.lengths.cpu().numpy()
        moves = int(lengths.sum())
        pol, rv = hist.policies.cpu().numpy(), hist.root_values.cpu().numpy()      # [moves, games, ...]
        played = np.arange(pol.shape[0])[:, None] < lengths[None, :]                 # padding past a game's end aside
        assert np.isfinite(pol[played]).all() and np.isfinite(rv[played]).all()
        assert np.allclose(pol[played].sum(-1), 1.0, atol=1e-5)
        assert (lengths >= 1).all()
        print(f'{name}: batch {b}: {G} games, {moves} moves, {moves * N / dt / 1e6:.1f} M simulations/s end to end, '
              f'{G / dt:.0f} games/s, lengths {lengths.min()}..{lengths.max()}', flush=True)
