"""BASELINE.json's configurations at their FULL sizes on the GPU, held to the same bar as the small cases.

* the native move (mz_search_move: the fused kernels of the bench line) at 4096 x 25 and 512 x 25 TicTacToe (512 = one
  rank's shard of BASELINE configs[2] at 8 GPUs), 1024 x 50 CartPole and 16384 x 50 Atari-shaped games with the network
  in the loop: its per-simulation trace (mz_move_io_t.trace_*) of EVERY live tree (Atari shape: 1024 of them) is
  replayed through the plain-C oracle -- selections, visit counts, root values, policies and chosen
  actions bit for bit, ZERO mismatches -- the network outputs in the trace are checked against the fp64 network oracle
  (1e-5), and size-independent properties are checked on every tree;
* the tree kernels alone at 16384 x 18 x 50 (unknown bounds) and 2048 x 362 x 800 (Go-shaped: 801 records per tree, an
  802-entry exploration table, deep paths) via mz_search_move_injected against the C oracle on a sample of trees.
Reference rule held: agent.py:198-232."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _decode_hidden(game, hidden):
    """The hidden workspace of a native move as float64 rows [blocks, T, H] (TicTacToe keeps split images there: per
    (game, cell) row [128 h | 128 l] halfs, value = h + l / 2048; mz_net.hip)."""
    import torch
    if game != 'tictactoe':
        return hidden.double().cpu().numpy()
    img = hidden.view(torch.float16).reshape(-1, 2, 128).double()
    return (img[:, 0] + img[:, 1] / 2048.0).reshape(hidden.shape).cpu().numpy()


# replayed: how many live trees go through the C oracle (None = every one of them; the Atari shape reads 50 records of
# a tree back one call at a time, so it keeps a sample)
@pytest.mark.parametrize('game,T,N,replayed', [('tictactoe', 4096, 25, None), ('tictactoe', 512, 25, None), ('tictactoe', 2048, 25, None),
                                               ('cartpole', 1024, 50, None), ('atari', 16384, 50, 1024)])
def test_native_move_full_size_against_the_oracles(game, T, N, replayed):
    import torch
    from muzero_amd.games import atari_shaped, cartpole, tictactoe
    from muzero_amd.selfplay import LockstepSearch
    from oracle import c_oracle, net_oracle
    mod = {'tictactoe': tictactoe, 'cartpole': cartpole, 'atari': atari_shaped}[game]
    config = mod.make_config(num_simulations=N)
    flat = net_oracle.random_weights(game, 31)
    net = config.network_config.network_class(config, weights=flat)
    ref = net_oracle.make_oracle(game, flat)
    A = ref.A
    rng = np.random.default_rng(17)
    if game == 'tictactoe':
        obs = rng.integers(0, 2, size=(T, 27)).astype(np.float32)
    elif game == 'atari':
        obs = rng.standard_normal((T, 256)).astype(np.float32)
    else:
        obs = (rng.uniform(-1, 1, size=(T, 4)) * [2.4, 3, 0.2, 3]).astype(np.float32)
    n_legal = rng.integers(0, A + 1, size=T).astype(np.int32)            # ragged, some games finished (inactive)
    n_legal[:8] = A
    legal = np.zeros((T, A), np.int32)
    noise = np.zeros((T, A))
    for t in range(T):
        legal[t, :n_legal[t]] = np.sort(rng.permutation(A)[:n_legal[t]])
        if n_legal[t]:
            noise[t, :n_legal[t]] = rng.dirichlet([1.0] * int(n_legal[t]))
    u = rng.random(T)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    search = LockstepSearch(config, net, num_trees=T)
    tr = {}
    visits, root_value, minmax, policy, chosen = (x.cpu().numpy() for x in search.run_native(
        dev(obs), dev(noise), dev(u), dev(np.ones(T)), dev(legal), dev(n_legal), trace=tr))
    tr = {k: v.cpu().numpy() for k, v in tr.items()}
    live = n_legal > 0

    # ---- properties on every tree
    assert (visits[live].sum(1) == N).all() and (visits[~live] == 0).all()
    np.testing.assert_array_equal(policy[live], (visits[live].astype(np.float64) / N).astype(np.float32))
    assert np.isfinite(root_value[live]).all()
    rows = np.nonzero(live)[0]
    assert (visits[rows, chosen[rows]] > 0).all()                        # a sampled action was visited, hence legal
    assert (chosen[~live] == -1).all()
    assert (tr['parent'][:, ~live] == -1).all() and (tr['parent'][0, live] == 0).all()
    for s in range(N):
        assert (tr['parent'][s, live] >= 0).all() and (tr['parent'][s, live] <= s).all()
        assert (tr['action'][s, live] >= 0).all() and (tr['action'][s, live] < A).all()
    kb = config.value_config.known_bounds
    if kb is None:
        assert (minmax[live, 0] <= minmax[live, 1]).all()

    # ---- the trees replayed through the C oracle with the device's own outputs: zero mismatches
    sample = rows if replayed is None else np.concatenate([rows[:8], rng.choice(rows[8:], size=replayed - 8, replace=False)])
    S = len(sample)
    opool = c_oracle.OraclePool(S, A, N, config.game_config.num_players, config.game_config.discount,
                                config.mcts_config.default_value, None if kb is None else kb.endpoints())
    root = np.zeros((S, A), np.float32)
    priors = np.zeros((N + 1, S, A), np.float32)
    for i, t in enumerate(sample):
        p = search.pool.read_root_priors(int(t))
        root[i, :len(p)] = p
        for e in range(1, N + 1):
            priors[e, i] = search.pool.read_block(int(t), e)['prior']
    opool.expand_roots(root, legal[sample], n_legal[sample])
    opool.add_root_noise(noise[sample], config.mcts_config.root_exploration_fraction)
    for s in range(N):
        p, a = opool.select()
        np.testing.assert_array_equal(p, tr['parent'][s, sample], err_msg=f'parent block, simulation {s}')
        np.testing.assert_array_equal(a, tr['action'][s, sample], err_msg=f'action, simulation {s}')
        opool.expand_backprop(priors[s + 1], tr['reward'][s, sample], tr['value'][s, sample])
    v2, rv2, mm2, pol2 = opool.root_stats()
    np.testing.assert_array_equal(visits[sample], v2)
    np.testing.assert_array_equal(root_value[sample], rv2)
    np.testing.assert_array_equal(policy[sample], pol2)
    np.testing.assert_array_equal(minmax[sample], mm2)
    np.testing.assert_array_equal(chosen[sample], opool.select_action(u[sample], 1.0))

    # ---- the network outputs of the trace against the fp64 network oracle, from the workspace's own parent rows
    sample = sample[:64] if len(sample) <= 64 else np.concatenate([sample[:8], rng.choice(sample[8:], size=56, replace=False)])
    hidden = _decode_hidden(game, search.hidden)
    vs = 100.0 if game == 'cartpole' else 1.0
    h0, _, l0, _ = ref.initial_inference(obs[sample])
    np.testing.assert_allclose(hidden[0][sample], h0, rtol=1e-5, atol=1e-5)
    for s in sorted({0, 1, N // 2, N - 1}):
        pb, act = tr['parent'][s, sample], tr['action'][s, sample]
        h2, r2, v2n, l2, _ = ref.recurrent_inference(hidden[pb, sample], act)
        np.testing.assert_allclose(hidden[s + 1][sample], h2, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tr['reward'][s, sample], r2, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tr['value'][s, sample], v2n, rtol=1e-5, atol=1e-5 * vs)
        np.testing.assert_allclose(tr['logits'][s, sample], l2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('T,A,N,players,discount,bounds,S', [
    (16384, 18, 50, 1, 0.997, None, 64),                 # BASELINE configs[3], tree half: live MinMaxStats
    (2048, 362, 800, 2, 1.0, (-1.0, 1.0), 10),           # BASELINE configs[4]: Go-shaped, 800 simulations
], ids=['atari_16384x18x50', 'go_2048x362x800'])
def test_injected_search_full_size_against_the_c_oracle(T, A, N, players, discount, bounds, S):
    import torch
    from muzero_amd.tree_pool import TreePool
    from oracle import c_oracle
    g = torch.Generator(device='cuda').manual_seed(99)
    # injected network outputs, generated on the device: priors = an fp32 softmax of N(0,1) logits
    priors = torch.softmax(torch.randn(N + 1, T, A, generator=g, device='cuda'), dim=-1).contiguous()
    values = torch.tanh(torch.randn(N + 1, T, generator=g, device='cuda')).contiguous()
    rewards = (torch.rand(N + 1, T, generator=g, device='cuda') if players == 1 else
               torch.zeros(N + 1, T, device='cuda')).contiguous()
    noise = torch.distributions.Dirichlet(torch.ones(A, dtype=torch.float64)).sample((T,)).cuda().contiguous()
    u = torch.rand(T, generator=g, device='cuda', dtype=torch.float64)
    kw = dict(num_actions=A, max_simulations=N, num_players=players, discount=discount, default_value=0.0,
              known_bounds=bounds)
    pool = TreePool(num_trees=T, **kw)
    pool.search_injected(priors, rewards, values, noise=noise, frac=0.25)
    visits, root_value, minmax, policy = (x.cpu().numpy() for x in pool.root_stats())
    chosen = pool.select_action(u, 1.0).cpu().numpy()
    counters = pool.counters()
    assert counters[0] == T * N
    assert (visits.sum(1) == N).all()
    np.testing.assert_array_equal(policy, (visits.astype(np.float64) / N).astype(np.float32))
    assert (visits[np.arange(T), chosen] > 0).all()

    rng = np.random.default_rng(3)
    sample = np.sort(rng.choice(T, size=S, replace=False))
    idx = torch.from_numpy(sample).cuda()
    opool = c_oracle.OraclePool(num_trees=S, **kw)
    sp = priors[:, idx].cpu().numpy()
    sr, sv = rewards[:, idx].cpu().numpy(), values[:, idx].cpu().numpy()
    opool.expand_roots(sp[0])
    opool.add_root_noise(noise[idx].cpu().numpy(), 0.25)
    for s in range(N):
        opool.select()
        opool.expand_backprop(sp[s + 1], sr[s + 1], sv[s + 1])
    v2, rv2, mm2, pol2 = opool.root_stats()
    np.testing.assert_array_equal(visits[sample], v2)
    np.testing.assert_array_equal(root_value[sample], rv2)
    np.testing.assert_array_equal(minmax[sample], mm2)
    np.testing.assert_array_equal(policy[sample], pol2)
    np.testing.assert_array_equal(chosen[sample], opool.select_action(u[idx].cpu().numpy(), 1.0))
    # whole trees: every record of two sampled trees, block by block (children, counts, value sums, rewards, links)
    for i in (0, S - 1):
        for e in list(range(0, N + 1, max(1, N // 40))) + [N]:
            a, b = pool.read_block(int(sample[i]), e), opool.read_block(i, e)
            for key in ('actions', 'count', 'value_sum', 'prior', 'child'):
                np.testing.assert_array_equal(a[key], b[key], err_msg=f'tree {sample[i]} block {e} {key}')
            seen = b['count'] > 0
            np.testing.assert_array_equal(a['reward'][seen], b['reward'][seen])
    pool.close()
