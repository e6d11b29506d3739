"""HIP tree pool (through the C ABI) against the reference fixtures and the plain-C oracle.  Bit-exact."""
import numpy as np
import pytest

from conftest import TREE_FIXTURES, load_golden
from helpers import (NumpyAdapter, TorchAdapter, assert_blocks_match_fixture, assert_matches_fixture,
                     assert_same_search, fixture_kwargs, run_fixture_on_pool, run_synthetic_on_pool,
                     synthetic_inputs)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ad():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return TorchAdapter('cuda')


def hip_pool(**kw):
    from muzero_amd.tree_pool import TreePool
    return TreePool(**kw)


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('name', TREE_FIXTURES)
def test_hip_pool_matches_reference_fixture(name, fused, ad):
    fx = load_golden(name)
    pool = hip_pool(**fixture_kwargs(fx, num_trees=5))
    res = run_fixture_on_pool(pool, fx, ad, fused=fused)
    assert_matches_fixture(res, fx)
    assert_blocks_match_fixture(pool, fx, tree=4)
    pool.close()


@pytest.mark.parametrize('name', [n for n in TREE_FIXTURES if 'choice_weights' in load_golden(n)])
def test_hip_pool_samples_with_the_reference_weights(name, ad):
    """softmax_sample at temperatures whose 1/t is no small integer: the pool looks count ** (1/t) up in the table the host
    registered (mz_pool_set_temperature_table) and draws, for every uniform, the action the reference's random.choices
    arithmetic draws from the weights the reference itself computed (fixture 'choice_weights')."""
    from helpers import fixture_temperature, reference_sample
    fx = load_golden(name)
    t = fixture_temperature(fx)
    us = np.concatenate([np.linspace(0.0, 1.0, 400, endpoint=False), [float(fx['u']), np.nextafter(1.0, 0.0)]])
    T = len(us)
    pool = hip_pool(**fixture_kwargs(fx, num_trees=T))
    run_fixture_on_pool(pool, fx, ad, fused=True)            # registers t through TreePool.set_temperature
    got = ad.get(pool.select_action(ad.put(us), ad.put(np.full(T, t))))
    want = np.array([reference_sample(fx, float(u)) for u in us])
    np.testing.assert_array_equal(got, want)
    # a ninth temperature evicts the oldest table; registering t again brings it back
    for other in (0.3, 0.6, 0.9, 1.1, 1.3, 1.7, 1.9, 2.3):
        pool.set_temperature(other)
    pool.set_temperature(t)
    np.testing.assert_array_equal(ad.get(pool.select_action(ad.put(us), ad.put(np.full(T, t)))), want)
    pool.close()


CASES = [
    # T, A, N, players, discount, default, bounds, reward_mode, legal_mode      (BASELINE.json configs, reduced T)
    dict(T=64, A=2, N=50, players=1, discount=0.99, default=50.0, bounds=(0.0, 100.0), reward='uniform', legal='all'),
    dict(T=200, A=9, N=25, players=2, discount=1.0, default=0.0, bounds=(-1.0, 1.0), reward='tanh', legal='ragged'),
    dict(T=300, A=18, N=50, players=1, discount=0.997, default=0.0, bounds=None, reward='uniform', legal='all'),
    dict(T=33, A=18, N=50, players=1, discount=0.997, default=0.0, bounds=None, reward='uniform', legal='ragged'),
    dict(T=24, A=362, N=120, players=2, discount=1.0, default=0.0, bounds=(-1.0, 1.0), reward='zero', legal='all'),
    dict(T=7, A=362, N=60, players=2, discount=1.0, default=0.0, bounds=None, reward='tanh', legal='ragged'),
    dict(T=5, A=1, N=10, players=1, discount=0.9, default=0.0, bounds=None, reward='uniform', legal='all'),
    dict(T=9, A=65, N=40, players=1, discount=0.9, default=0.3, bounds=(-2.0, 3.0), reward='tanh', legal='ragged'),
    dict(T=3, A=700, N=20, players=1, discount=0.9, default=0.0, bounds=None, reward='uniform', legal='ragged'),
]


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('case', CASES, ids=lambda c: f"T{c['T']}_A{c['A']}_N{c['N']}_{c['legal']}")
def test_hip_pool_matches_c_oracle(case, fused, ad):
    from oracle import c_oracle
    inp = synthetic_inputs(case['T'], case['A'], case['N'], seed=1234 + case['A'], reward_mode=case['reward'],
                           legal_mode=case['legal'], value_scale=100.0 if case['default'] == 50.0 else 1.0)
    kw = dict(num_trees=case['T'], num_actions=case['A'], max_simulations=case['N'], num_players=case['players'],
              discount=case['discount'], default_value=case['default'], known_bounds=case['bounds'])
    want = run_synthetic_on_pool(c_oracle.OraclePool(**kw), inp, NumpyAdapter, temperature=1.0)
    pool = hip_pool(**kw)
    got = run_synthetic_on_pool(pool, inp, ad, fused=fused, temperature=1.0)
    assert_same_search(got, want)
    # temperature 0 (arg-max with ties to the largest action) on the same finished trees
    opool = c_oracle.OraclePool(**kw)
    run_synthetic_on_pool(opool, inp, NumpyAdapter)
    np.testing.assert_array_equal(ad.get(pool.select_action(ad.put(inp['u']), 0.0)),
                                  opool.select_action(inp['u'], 0.0))
    # a few whole trees, block by block
    for t in range(0, case['T'], max(1, case['T'] // 3)):
        if inp['n_legal'][t] == 0:
            continue
        for e in range(0, case['N'] + 1, 7):
            a, b = pool.read_block(t, e), opool.read_block(t, e)
            for key in ('actions', 'count', 'value_sum', 'prior', 'child'):
                np.testing.assert_array_equal(a[key], b[key], err_msg=f'tree {t} block {e} {key}')
            vis = a['count'] > 0
            np.testing.assert_array_equal(a['reward'][vis], b['reward'][vis])
    pool.close()


LANE_CASES = [
    # every group width against the C oracle: narrow groups take several children per lane, keep several path
    # positions per lane and stage the softmax terms and the backup chain in LDS; long paths overflow those
    dict(T=70, A=2, N=60, players=1, discount=0.99, default=50.0, bounds=(0.0, 100.0), reward='uniform', legal='all'),
    dict(T=130, A=9, N=25, players=2, discount=1.0, default=0.0, bounds=(-1.0, 1.0), reward='tanh', legal='ragged'),
    dict(T=90, A=18, N=50, players=1, discount=0.997, default=0.0, bounds=None, reward='uniform', legal='ragged'),
    dict(T=10, A=362, N=40, players=2, discount=1.0, default=0.0, bounds=None, reward='tanh', legal='all'),
]


@pytest.mark.parametrize('lanes', [2, 4, 8, 16, 32, 64])
@pytest.mark.parametrize('case', LANE_CASES, ids=lambda c: f"A{c['A']}_N{c['N']}")
def test_lanes_per_tree_does_not_change_results(case, lanes, ad):
    from oracle import c_oracle
    inp = synthetic_inputs(case['T'], case['A'], case['N'], seed=4321 + case['A'], reward_mode=case['reward'],
                           legal_mode=case['legal'], value_scale=100.0 if case['default'] == 50.0 else 1.0)
    kw = dict(num_trees=case['T'], num_actions=case['A'], max_simulations=case['N'], num_players=case['players'],
              discount=case['discount'], default_value=case['default'], known_bounds=case['bounds'])
    opool = c_oracle.OraclePool(**kw)
    want = run_synthetic_on_pool(opool, inp, NumpyAdapter, temperature=1.0)
    for fused in (False, True):
        pool = hip_pool(lanes_per_tree=lanes, **kw)
        got = run_synthetic_on_pool(pool, inp, ad, fused=fused, temperature=1.0)
        assert_same_search(got, want)
        for t in range(0, case['T'], max(1, case['T'] // 3)):
            if inp['n_legal'][t] == 0:
                continue
            for e in range(0, case['N'] + 1, 9):
                a, b = pool.read_block(t, e), opool.read_block(t, e)
                for key in ('actions', 'count', 'value_sum', 'prior', 'child'):
                    np.testing.assert_array_equal(a[key], b[key], err_msg=f'tree {t} block {e} {key}')
        pool.close()


@pytest.mark.parametrize('lanes', [2, 8, 64])
def test_logits_any_lanes_within_exp_tolerance(lanes, ad):
    """The device-side softmax (LDS-staged for action axes wider than the group) for every group width."""
    for A in (2, 9, 18, 65, 362):
        T = 6
        inp = synthetic_inputs(T, A, 2, seed=900 + A)
        pool = hip_pool(num_trees=T, num_actions=A, max_simulations=2, known_bounds=None, lanes_per_tree=lanes)
        pool.expand_roots(ad.put(inp['logits'][0]), logits=True)
        pool.select()
        pool.step(ad.put(inp['logits'][1]), ad.put(inp['rewards'][1]), ad.put(inp['values'][1]), logits=True)
        for t in range(T):
            np.testing.assert_allclose(pool.read_block(t, 1)['prior'], inp['priors'][1, t].astype(np.float64),
                                       rtol=5e-7, err_msg=f'A={A} lanes={lanes}')
        pool.close()


def _tied_priors(inp, mode, rng):
    """Replaces the non-root priors of synthetic inputs by ones that make score ties (or irregular values) common."""
    pri = inp['priors']
    N1, T, A = pri.shape
    if mode == 'equal':                 # every child of a node scores the same until visited: largest action wins
        pri[1:] = np.float32(1.0 / A)
    elif mode == 'levels':              # three distinct values: long runs of equal priors
        lv = np.array([0.5, 0.25, 0.125], np.float32)
        pri[1:] = lv[rng.integers(0, 3, size=(N1 - 1, T, A))] / A
    elif mode == 'absorbed':            # a few real priors; the rest so small that prior * 2 log(n+1) vanishes in the sum
        tiny = (10.0 ** rng.uniform(-38, -20, size=(N1 - 1, T, A))).astype(np.float32)
        big = rng.random((N1 - 1, T, A)) < 2.0 / A
        pri[1:] = np.where(big, rng.random((N1 - 1, T, A)).astype(np.float32) * 0.5, tiny)
    elif mode == 'irregular':           # values the prefix argument does not cover: the node must be read whole
        bad = rng.random((N1 - 1, T, A)) < 0.03
        vals = np.array([np.inf, -0.25, -np.inf], np.float32)[rng.integers(0, 3, size=(N1 - 1, T, A))]
        pri[1:] = np.where(bad, vals, pri[1:])
    return inp


@pytest.mark.parametrize('mode', ['equal', 'levels', 'absorbed', 'irregular'])
@pytest.mark.parametrize('shape', [(60, 9, 30, 16), (60, 18, 50, 16), (60, 18, 50, 4), (12, 65, 60, 64), (12, 100, 80, 32)],
                         ids=lambda x: 'T%d_A%d_N%d_G%d' % x)
def test_prefix_select_with_tied_priors(shape, mode, ad):
    """Nodes keep their children by descending prior and select reads a prefix (mz_tree_dev.h): ties at the end of the
    prefix, priors that vanish in the score, and non-finite / negative priors must all end where a full scan ends."""
    from oracle import c_oracle
    T, A, N, lanes = shape
    rng = np.random.Generator(np.random.PCG64(7 + A + len(mode)))
    for bounds, default, players in (((-1.0, 1.0), 0.0, 2), (None, 0.25, 1)):
        inp = _tied_priors(synthetic_inputs(T, A, N, seed=31 + A, reward_mode='tanh', legal_mode='ragged'), mode, rng)
        kw = dict(num_trees=T, num_actions=A, max_simulations=N, num_players=players, discount=0.97,
                  default_value=default, known_bounds=bounds)
        opool = c_oracle.OraclePool(**kw)
        want = run_synthetic_on_pool(opool, inp, NumpyAdapter, temperature=1.0)
        pool = hip_pool(lanes_per_tree=lanes, **kw)
        got = run_synthetic_on_pool(pool, inp, ad, fused=True, temperature=1.0)
        assert_same_search(got, want)
        for t in range(0, T, 7):
            if inp['n_legal'][t] == 0:
                continue
            for e in range(0, N + 1, 5):
                a, b = pool.read_block(t, e), opool.read_block(t, e)
                for key in ('actions', 'count', 'value_sum', 'prior', 'child'):
                    np.testing.assert_array_equal(a[key], b[key], err_msg=f'{mode} tree {t} block {e} {key}')
        pool.close()


@pytest.mark.parametrize('use_logits', [False, True])
def test_native_injected_move_equals_stepwise_calls(use_logits, ad):
    """mz_search_move_injected enqueues root expansion, noise and N simulations from native code: same trees as the same
    sequence of single calls (and therefore as the C oracle, which the stepwise path is held to above)."""
    import torch
    T, A, N = 96, 18, 30
    inp = synthetic_inputs(T, A, N, seed=606, reward_mode='uniform', legal_mode='ragged')
    kw = dict(num_trees=T, num_actions=A, max_simulations=N, num_players=1, discount=0.997, default_value=0.0,
              known_bounds=None)
    step = hip_pool(**kw)
    want = run_synthetic_on_pool(step, inp, ad, fused=True, temperature=1.0, use_logits=use_logits)
    native = hip_pool(**kw)
    key = 'logits' if use_logits else 'priors'
    parent, action = native.search_injected(ad.put(inp[key]), ad.put(inp['rewards']), ad.put(inp['values']),
                                            noise=ad.put(inp['noise']), frac=0.25, legal=ad.put(inp['legal']),
                                            n_legal=ad.put(inp['n_legal']), logits=use_logits)
    visits, root_value, minmax, policy = (ad.get(x) for x in native.root_stats())
    np.testing.assert_array_equal(visits, want['visits'])
    np.testing.assert_array_equal(root_value, want['root_value'])
    np.testing.assert_array_equal(minmax, want['minmax'])
    np.testing.assert_array_equal(policy, want['policy'])
    chosen = ad.get(native.select_action(ad.put(inp['u']), ad.put(np.full(T, 1.0, np.float64))))
    np.testing.assert_array_equal(chosen, want['action'])
    np.testing.assert_array_equal(native.counters()[:2], want['counters'][:2])
    for t in range(0, T, 17):
        if inp['n_legal'][t]:
            for e in (0, 1, N // 2, N):
                a, b = native.read_block(t, e), step.read_block(t, e)
                for k in ('actions', 'count', 'value_sum', 'prior', 'child'):
                    np.testing.assert_array_equal(a[k], b[k])
    step.close()
    native.close()


def test_pool_is_reusable_across_moves(ad):
    """reset() + expand_roots() starts a fresh search in the same memory (agent.py:220-221: no tree reuse)."""
    from oracle import c_oracle
    kw = dict(num_trees=40, num_actions=9, max_simulations=25, num_players=2, discount=1.0, default_value=0.0,
              known_bounds=None)
    pool = hip_pool(**kw)
    for move in range(3):
        inp = synthetic_inputs(40, 9, 25, seed=50 + move, reward_mode='tanh', legal_mode='ragged')
        pool.reset()
        got = run_synthetic_on_pool(pool, inp, ad, fused=bool(move % 2))
        want = run_synthetic_on_pool(c_oracle.OraclePool(**kw), inp, NumpyAdapter)
        got['counters'][:2] -= move * 0       # counters are cumulative on the HIP pool
        for key in ('trace_parent', 'trace_action', 'visits', 'policy', 'action', 'root_value', 'minmax'):
            np.testing.assert_array_equal(got[key], want[key], err_msg=f'move {move} {key}')
    pool.close()


def test_logits_entry_points_within_exp_tolerance(ad):
    """mz_*_logits computes masked_policy on the device; expf vs NumPy's SIMD exp differ by <=2 ulp (trap T4),
    so priors are compared to 4 ulp (stated tolerance: rtol 5e-7) instead of bit for bit."""
    T, A, N = 50, 18, 3
    inp = synthetic_inputs(T, A, N, seed=77, legal_mode='ragged')
    pool = hip_pool(num_trees=T, num_actions=A, max_simulations=N, known_bounds=None)
    pool.expand_roots(ad.put(inp['logits'][0]), ad.put(inp['legal']), ad.put(inp['n_legal']), logits=True)
    for t in range(T):
        nl = int(inp['n_legal'][t])
        if nl == 0:
            continue
        b = pool.read_block(t, 0)
        np.testing.assert_allclose(b['prior'], inp['priors'][0, t, :nl].astype(np.float64), rtol=5e-7, atol=0)
    pool.select()
    pool.expand_backprop(ad.put(inp['logits'][1]), ad.put(inp['rewards'][1]), ad.put(inp['values'][1]), logits=True)
    for t in range(T):
        if inp['n_legal'][t] == 0:
            continue
        b = pool.read_block(t, 1)
        np.testing.assert_allclose(b['prior'], inp['priors'][1, t].astype(np.float64), rtol=5e-7, atol=0)
    pool.close()


def test_logits_wide_action_space(ad):
    T, A = 4, 362
    inp = synthetic_inputs(T, A, 2, seed=78)
    pool = hip_pool(num_trees=T, num_actions=A, max_simulations=2, known_bounds=(-1.0, 1.0), num_players=2)
    pool.expand_roots(ad.put(inp['logits'][0]), logits=True)
    for t in range(T):
        np.testing.assert_allclose(pool.read_block(t, 0)['prior'], inp['priors'][0, t].astype(np.float64), rtol=5e-7)
    pool.select()
    pool.expand_backprop(ad.put(inp['logits'][1]), ad.put(inp['rewards'][1]), ad.put(inp['values'][1]), logits=True)
    for t in range(T):
        np.testing.assert_allclose(pool.read_block(t, 1)['prior'], inp['priors'][1, t].astype(np.float64), rtol=5e-7)
    pool.close()


def test_call_sequence_errors(ad):
    import torch
    from muzero_amd.exceptions import MuZeroError, MuZeroHipError
    pool = hip_pool(num_trees=4, num_actions=3, max_simulations=2)
    z = lambda *s: torch.zeros(*s, device='cuda')
    with pytest.raises(MuZeroHipError):
        pool.select()                                    # before expand_roots
    pool.expand_roots(torch.full((4, 3), 1 / 3, device='cuda'))
    with pytest.raises(MuZeroHipError):
        pool.expand_backprop(z(4, 3), z(4), z(4))        # no select pending
    pool.select()
    with pytest.raises(MuZeroHipError):
        pool.select()                                    # select twice
    with pytest.raises(MuZeroError):
        pool.expand_backprop(z(4, 2), z(4), z(4))        # wrong shape is rejected on the host
    with pytest.raises(MuZeroError):
        pool.expand_backprop(z(4, 3).double(), z(4), z(4))
    pool.step(torch.full((4, 3), 1 / 3, device='cuda'), z(4), z(4))
    with pytest.raises(MuZeroHipError):
        pool.step(z(4, 3), z(4), z(4))                   # step on the last simulation
    pool.expand_backprop(torch.full((4, 3), 1 / 3, device='cuda'), z(4), z(4))
    with pytest.raises(MuZeroHipError):
        pool.select()                                    # pool full (N = 2)
    assert int(pool.root_stats()[0].sum()) == 8
    with pytest.raises(MuZeroHipError):
        hip_pool(num_trees=0, num_actions=3, max_simulations=2)
    with pytest.raises(MuZeroHipError):
        hip_pool(num_trees=1, num_actions=5000, max_simulations=2)
    pool.close()
