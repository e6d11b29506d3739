"""The N>1 path on CPU: two gloo ranks shard the games, search their shard independently (C oracle standing in for
the HIP pool -- no GPU here) and exchange per-move search statistics with TrajectoryGather.  The gathered result must
equal the single-process search over all games."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

T_TOTAL, A, N = 24, 9, 12


def search(inp, lo, hi):
    from helpers import NumpyAdapter, run_synthetic_on_pool
    from oracle import c_oracle
    shard = {k: (v[:, lo:hi] if k in ('logits', 'priors', 'values', 'rewards') else v[lo:hi]) if isinstance(v, np.ndarray) else v
             for k, v in inp.items()}
    pool = c_oracle.OraclePool(hi - lo, A, N, 2, 1.0, 0.0, (-1.0, 1.0))
    return run_synthetic_on_pool(pool, shard, NumpyAdapter)


def worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from helpers import synthetic_inputs
    from muzero_amd.selfplay import TrajectoryGather, shard_bounds
    inp = synthetic_inputs(T_TOTAL, A, N, seed=5, reward_mode='tanh', legal_mode='ragged')
    lo, hi = shard_bounds(T_TOTAL, rank, world)
    res = search(inp, lo, hi)
    gather = TrajectoryGather(hi - lo, A, torch.device('cpu'))
    gather.start(torch.from_numpy(res['policy']), torch.from_numpy(res['root_value']), torch.from_numpy(res['action']))
    got = gather.wait().numpy().copy()
    # a second move re-uses the buffers (async handle waited one move later)
    gather.start(torch.from_numpy(res['policy']) * 0 + rank, torch.from_numpy(res['root_value']), torch.from_numpy(res['action']))
    second = gather.wait().numpy().copy()
    np.save(os.path.join(out_dir, f'rank{rank}.npy'), got)
    np.save(os.path.join(out_dir, f'second{rank}.npy'), second)
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path):
    from helpers import synthetic_inputs
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    inp = synthetic_inputs(T_TOTAL, A, N, seed=5, reward_mode='tanh', legal_mode='ragged')
    whole = search(inp, 0, T_TOTAL)
    for rank in range(world):
        got = np.load(tmp_path / f'rank{rank}.npy')            # [world, T/world, A+2]: every rank holds everything
        flat = got.reshape(T_TOTAL, A + 2)
        np.testing.assert_array_equal(flat[:, :A], whole['policy'])
        np.testing.assert_array_equal(flat[:, A + 1].astype(np.int32), whole['action'])
        live = ~np.isnan(whole['root_value'])
        np.testing.assert_array_equal(flat[live, A], whole['root_value'][live].astype(np.float32))
        second = np.load(tmp_path / f'second{rank}.npy')
        assert (second[0, :, :A] == 0).all() and (second[1, :, :A] == 1).all()


def test_shard_bounds_cover_all_games():
    from muzero_amd.selfplay import shard_bounds
    for total in (1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def history_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from muzero_amd.selfplay import BatchedHistory
    h = make_history(rank)
    full = h.all_gather()
    torch.save({f: getattr(full, f) for f in BatchedHistory.FIELDS}, os.path.join(out_dir, f'hist{rank}.pt'))
    dist.destroy_process_group()


def make_history(rank, M=5, G=3, A=4):
    from muzero_amd.selfplay import BatchedHistory
    g = torch.Generator().manual_seed(100 + rank)
    h = BatchedHistory(M, G, (2, 3), A, torch.device('cpu'))
    h.observations.copy_(torch.rand(h.observations.shape, generator=g))
    h.to_plays.copy_(torch.randint(0, 2, h.to_plays.shape, generator=g))
    h.actions.copy_(torch.randint(0, A, h.actions.shape, generator=g))
    h.rewards.copy_(torch.rand(h.rewards.shape, generator=g))
    h.root_values.copy_(torch.rand(h.root_values.shape, generator=g, dtype=torch.float64))
    h.policies.copy_(torch.rand(h.policies.shape, generator=g))
    h.lengths.copy_(torch.randint(1, M + 1, h.lengths.shape, generator=g))
    return h


def test_finished_games_all_gather_two_ranks(tmp_path):
    """The one exchange of the path: every rank ends up with every rank's games, in rank order."""
    world, port = 2, 31500 + os.getpid() % 2000
    mp.spawn(history_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [make_history(r) for r in range(world)]
    G = parts[0].G
    for rank in range(world):
        got = torch.load(tmp_path / f'hist{rank}.pt', weights_only=True)
        for r, p in enumerate(parts):
            assert torch.equal(got['lengths'][r * G:(r + 1) * G], p.lengths)
            for f in ('observations', 'to_plays', 'actions', 'rewards', 'root_values', 'policies'):
                assert torch.equal(got[f][:, r * G:(r + 1) * G], getattr(p, f)), f
    # the gathered block converts to the reference's GameHistory field set
    from muzero_amd.selfplay import BatchedHistory
    whole = BatchedHistory(parts[0].M, G * world, (2, 3), parts[0].A, torch.device('cpu'))
    for f in BatchedHistory.FIELDS:
        getattr(whole, f).copy_(got[f])
    games = whole.to_game_histories()
    assert len(games) == G * world
    for i, h in enumerate(games):
        n = int(got['lengths'][i])
        assert len(h) == n and len(h.observations) == n + 1 and h.observations[0].shape == (2, 3)
        assert h.actions == got['actions'][:n, i].tolist() and len(h.policies) == n == len(h.root_values)
