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
    # the per-move rate of the search alone on this box (what bench.py measures), for the ratio below
    engine.env.reset()
    obs, legal, n_legal = engine.env.observe()
    noise = engine._dirichlet(n_legal)
    u = torch.rand(G, dtype=torch.float64, device=engine.device)
    temp = torch.ones(G, dtype=torch.float64, device=engine.device)
    for _ in range(3):
        engine.search.run(obs, noise, u, temp, legal, n_legal)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        engine.search.run(obs, noise, u, temp, legal, n_legal)
    torch.cuda.synchronize(); per_move = 20 * G * N / (time.perf_counter() - t0)
    print(f'{name}: search alone: {per_move / 1e6:.1f} M simulations/s per move', flush=True)
    # continuous refill: finished slots restart at once, every move searches a full batch (agent_services.py:47-54)
    total = 8 * G
    for b in range(batches):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hist = engine.play_continuous(total, max_moves=max_moves)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        searched = hist.metadata['moves_played'] * G * N
        stored = int(hist.lengths.sum().item())
        print(f'{name}: refill {b}: {total} games stored ({hist.metadata["games_started"]} started) in '
              f'{hist.metadata["moves_played"]} moves, {searched / dt / 1e6:.1f} M simulations/s searched, '
              f'{stored * N / dt / 1e6:.1f} M simulations/s in stored games, {total / dt:.0f} games/s; end to end = '
              f'{searched / dt / per_move:.3f} of the per-move rate', flush=True)
