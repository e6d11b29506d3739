"""Shared drivers: run the same search on the HIP TreePool and on the CPU oracles."""
import numpy as np


class NumpyAdapter:
    """For oracle.c_oracle.OraclePool (numpy in, numpy out)."""
    @staticmethod
    def put(a):
        return a

    @staticmethod
    def get(a):
        return np.asarray(a)


class TorchAdapter:
    """For muzero_amd.tree_pool.TreePool (device tensors in, device tensors out)."""
    def __init__(self, device='cuda'):
        import torch
        self.torch = torch
        self.device = device

    def put(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    @staticmethod
    def get(a):
        return a.detach().cpu().numpy()


def fixture_kwargs(fx, num_trees):
    kb = (float(fx['known_min']), float(fx['known_max'])) if int(fx['has_bounds']) else None
    return dict(num_trees=num_trees, num_actions=int(fx['A']), max_simulations=int(fx['N']),
                num_players=int(fx['num_players']), discount=float(fx['discount']),
                default_value=float(fx['default_value']), known_bounds=kb)


def fixture_temperature(fx):
    return float(fx['temperature']) if int(fx['num_moves']) < int(fx['freezing_moves']) else 0.0


def run_fixture_on_pool(pool, fx, ad, fused=False):
    """Replays one golden fixture on every tree of `pool`; returns traces and root statistics (numpy)."""
    T, A, N = pool.T, pool.A, pool.N
    legal = np.zeros((T, A), np.int32)
    nl = len(fx['legal'])
    legal[:, :nl] = fx['legal']
    n_legal = np.full(T, nl, np.int32)
    rep = lambda row: np.ascontiguousarray(np.broadcast_to(row, (T,) + np.shape(row)))
    pool.expand_roots(ad.put(rep(fx['priors'][0])), ad.put(legal), ad.put(n_legal))
    pool.add_root_noise(ad.put(rep(fx['noise'])), float(fx['frac']))
    trace_parent, trace_action = [], []
    parent, action = pool.select()
    for s in range(N):
        trace_parent.append(ad.get(parent).copy())
        trace_action.append(ad.get(action).copy())
        args = (ad.put(rep(fx['priors'][s + 1])), ad.put(rep(fx['rewards'][s + 1])), ad.put(rep(fx['values'][s + 1])))
        if s == N - 1:
            pool.expand_backprop(*args)
        elif fused:
            parent, action = pool.step(*args)
        else:
            pool.expand_backprop(*args)
            parent, action = pool.select()
    visits, root_value, minmax, policy = (ad.get(x) for x in pool.root_stats())
    if hasattr(pool, 'set_temperature'):      # HIP pool: the host tabulates count ** (1/t) (mz_pool_set_temperature_table)
        pool.set_temperature(fixture_temperature(fx))
    chosen = ad.get(pool.select_action(ad.put(np.full(T, float(fx['u']), np.float64)),
                                       ad.put(np.full(T, fixture_temperature(fx), np.float64))))
    return dict(trace_parent=np.stack(trace_parent, 1), trace_action=np.stack(trace_action, 1), visits=visits,
                root_value=root_value, minmax=minmax, policy=policy, action=chosen)


def assert_matches_fixture(res, fx):
    T = res['visits'].shape[0]
    for t in range(T):
        np.testing.assert_array_equal(res['trace_parent'][t], fx['trace_parent'])
        np.testing.assert_array_equal(res['trace_action'][t], fx['trace_action'])
        np.testing.assert_array_equal(res['visits'][t], fx['root_visits'])
        assert res['root_value'][t] == float(fx['root_value'])          # bit-exact fp64
        assert res['minmax'][t, 0] == float(fx['mm_min']) and res['minmax'][t, 1] == float(fx['mm_max'])
        np.testing.assert_array_equal(res['policy'][t], fx['policy'])
        assert res['action'][t] == int(fx['action'])


def assert_blocks_match_fixture(pool, fx, tree=0):
    if 'tree_count' not in fx:
        return
    for e in range(int(fx['N']) + 1):
        n = int(fx['tree_nchild'][e])
        b = pool.read_block(tree, e)
        assert len(b['count']) == n
        np.testing.assert_array_equal(b['actions'], fx['tree_actions'][e, :n])
        np.testing.assert_array_equal(b['count'], fx['tree_count'][e, :n])
        np.testing.assert_array_equal(b['value_sum'], fx['tree_vsum'][e, :n])
        np.testing.assert_array_equal(b['prior'], fx['tree_prior'][e, :n])
        np.testing.assert_array_equal(b['child'], fx['tree_child'][e, :n])
        visited = fx['tree_count'][e, :n] > 0
        np.testing.assert_array_equal(b['reward'][visited].astype(np.float64), fx['tree_reward'][e, :n][visited])


def synthetic_inputs(T, A, N, seed, reward_mode='uniform', legal_mode='all', value_scale=1.0):
    """SURVEY.md section 8(d) synthetic tree inputs: logits~N(0,1) -> fp32 softmax priors, values~tanh(N(0,1)),
    rewards~U(0,1) (or tanh for 2-player), Dirichlet(1) root noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    logits = rng.standard_normal((N + 1, T, A)).astype(np.float32)
    e = np.exp(logits)
    priors = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    values = (np.tanh(rng.standard_normal((N + 1, T))) * value_scale).astype(np.float32)
    if reward_mode == 'uniform':
        rewards = rng.random((N + 1, T)).astype(np.float32)
    elif reward_mode == 'tanh':
        rewards = np.tanh(rng.standard_normal((N + 1, T))).astype(np.float32)
    else:
        rewards = np.zeros((N + 1, T), np.float32)
    legal = np.zeros((T, A), np.int32)
    n_legal = np.zeros(T, np.int32)
    for t in range(T):
        if legal_mode == 'all':
            acts = np.arange(A)
        elif legal_mode == 'ragged':      # includes empty (inactive) trees and permuted action orders
            k = int(rng.integers(0, A + 1))
            acts = rng.permutation(A)[:k]
            if t % 2 == 0:
                acts = np.sort(acts)
        n_legal[t] = len(acts)
        legal[t, :len(acts)] = acts
    # root masked policy, positional (trap T5): full-length, zero at illegal actions
    root = np.zeros((T, A), np.float32)
    for t in range(T):
        acts = legal[t, :n_legal[t]]
        if len(acts):
            m = np.zeros(A, np.float32)
            m[acts] = e[0, t, acts]
            root[t] = m / m.sum()
    priors[0] = root
    noise = np.zeros((T, A), np.float64)
    for t in range(T):
        if n_legal[t]:
            noise[t, :n_legal[t]] = rng.dirichlet([1.0] * int(n_legal[t]))
    return dict(logits=logits, priors=priors, values=values, rewards=rewards, legal=legal, n_legal=n_legal, noise=noise,
                u=rng.random(T), seed=seed)


def run_synthetic_on_pool(pool, inp, ad, frac=0.25, fused=False, temperature=1.0, num_sims=None, use_logits=False):
    T, A = pool.T, pool.A
    N = pool.N if num_sims is None else num_sims
    key = 'logits' if use_logits else 'priors'
    kw = dict(logits=True) if use_logits else {}
    pool.expand_roots(ad.put(inp[key][0]), ad.put(inp['legal']), ad.put(inp['n_legal']), **kw)
    pool.add_root_noise(ad.put(inp['noise']), frac)
    trace_parent, trace_action = [], []
    parent, action = pool.select()
    for s in range(N):
        trace_parent.append(ad.get(parent).copy())
        trace_action.append(ad.get(action).copy())
        args = (ad.put(inp[key][s + 1]), ad.put(inp['rewards'][s + 1]), ad.put(inp['values'][s + 1]))
        if s == N - 1:
            pool.expand_backprop(*args, **kw)
        elif fused:
            parent, action = pool.step(*args, **kw)
        else:
            pool.expand_backprop(*args, **kw)
            parent, action = pool.select()
    visits, root_value, minmax, policy = (ad.get(x) for x in pool.root_stats())
    if hasattr(pool, 'set_temperature'):
        pool.set_temperature(temperature)
    chosen = ad.get(pool.select_action(ad.put(inp['u']), ad.put(np.full(T, temperature, np.float64))))
    return dict(trace_parent=np.stack(trace_parent, 1), trace_action=np.stack(trace_action, 1), visits=visits,
                root_value=root_value, minmax=minmax, policy=policy, action=chosen, counters=pool.counters())


def assert_same_search(a, b):
    for key in ('trace_parent', 'trace_action', 'visits', 'policy', 'action'):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    np.testing.assert_array_equal(a['root_value'], b['root_value'])      # NaN == NaN positions included
    np.testing.assert_array_equal(a['minmax'], b['minmax'])
    np.testing.assert_array_equal(a['counters'][:2], b['counters'][:2])


def reference_sample(fx, u):
    """softmax_sample (agent.py:162-168) over the fixture's root with the weights the REFERENCE computed
    (fx['choice_weights'] = count ** (1/t) per child in insertion order, recorded from its random.choices call):
    CPython's random.choices = population[bisect_right(accumulate(weights), u * total, 0, n - 1)]."""
    import bisect
    import itertools
    w = [float(x) for x in fx['choice_weights']]
    cum = list(itertools.accumulate(w))
    total = cum[-1] + 0.0
    return int(fx['legal'][bisect.bisect_right(cum, u * total, 0, len(w) - 1)])
