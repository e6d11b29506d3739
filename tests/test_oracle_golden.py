"""The CPU oracle against the reference's own outputs (tests/golden/*.npz, made by make_golden.py)."""
import numpy as np
import pytest

from conftest import TREE_FIXTURES, load_golden
from oracle import mcts_oracle


@pytest.mark.parametrize('name', TREE_FIXTURES)
def test_python_oracle_matches_reference(name):
    fx = load_golden(name)
    tree, tp, ta, action = mcts_oracle.run_fixture(fx)
    np.testing.assert_array_equal(np.array(tp, np.int32), fx['trace_parent'])
    np.testing.assert_array_equal(np.array(ta, np.int32), fx['trace_action'])
    np.testing.assert_array_equal(tree.root_visits(), fx['root_visits'])
    assert tree.root_count == int(fx['root_count'])
    assert tree.root_vsum == float(fx['root_vsum'])            # bit-exact fp64
    assert tree.root_value() == float(fx['root_value'])
    assert tree.mm_min == float(fx['mm_min']) and tree.mm_max == float(fx['mm_max'])
    np.testing.assert_array_equal(tree.policy_target(), fx['policy'])
    assert action == int(fx['action'])
    if 'tree_count' in fx:
        for e in range(int(fx['N']) + 1):
            n = int(fx['tree_nchild'][e])
            if n == 0:
                assert e >= len(tree.actions) or tree.count[e] == [0] * len(tree.count[e])
                continue
            np.testing.assert_array_equal(np.array(tree.actions[e], np.int32), fx['tree_actions'][e, :n])
            np.testing.assert_array_equal(np.array(tree.count[e], np.int32), fx['tree_count'][e, :n])
            np.testing.assert_array_equal(np.array(tree.vsum[e]), fx['tree_vsum'][e, :n])
            np.testing.assert_array_equal(np.array(tree.prior[e]), fx['tree_prior'][e, :n])
            np.testing.assert_array_equal(np.array(tree.child[e], np.int32), fx['tree_child'][e, :n])
            visited = fx['tree_count'][e, :n] > 0
            np.testing.assert_array_equal(np.array(tree.reward[e])[visited], fx['tree_reward'][e, :n][visited])


# ------------------------------------------------------------------------------- plain-C oracle
from helpers import (NumpyAdapter, assert_blocks_match_fixture, assert_matches_fixture, fixture_kwargs,
                     run_fixture_on_pool, run_synthetic_on_pool, synthetic_inputs)
from oracle import c_oracle


@pytest.mark.parametrize('name', TREE_FIXTURES)
def test_c_oracle_matches_reference(name):
    fx = load_golden(name)
    pool = c_oracle.OraclePool(**fixture_kwargs(fx, num_trees=3))
    res = run_fixture_on_pool(pool, fx, NumpyAdapter)
    assert_matches_fixture(res, fx)
    assert_blocks_match_fixture(pool, fx, tree=2)


SAMPLED_FIXTURES = [n for n in TREE_FIXTURES if 'choice_weights' in load_golden(n)]


@pytest.mark.parametrize('name', SAMPLED_FIXTURES)
def test_oracles_sample_with_the_reference_weights(name):
    """count ** (1/t) for temperatures whose 1/t is no small integer (agent.py:167): both oracles reproduce the weights the
    reference passed to random.choices, and the action it would draw for any uniform."""
    from helpers import fixture_temperature, reference_sample
    fx = load_golden(name)
    t = fixture_temperature(fx)
    tree, _, _, _ = mcts_oracle.run_fixture(fx)
    counts = [int(fx['root_visits'][a]) for a in fx['legal']]
    np.testing.assert_array_equal(np.array([c ** (1 / t) for c in counts]), fx['choice_weights'])
    us = np.concatenate([np.linspace(0.0, 1.0, 300, endpoint=False), [float(fx['u']), np.nextafter(1.0, 0.0)]])
    T = len(us)
    pool = c_oracle.OraclePool(**fixture_kwargs(fx, num_trees=T))
    run_fixture_on_pool(pool, fx, NumpyAdapter)
    got = pool.select_action(us, t)
    for i, u in enumerate(us):
        want = reference_sample(fx, float(u))
        assert tree.select_action(float(u), t) == want
        assert got[i] == want, (i, u)


def test_numpy_pairwise_sum_model():
    """Trap T4: the oracle's fp32 summation equals ndarray.sum bit for bit, sizes 1..700."""
    rng = np.random.Generator(np.random.PCG64(7))
    for n in list(range(1, 300)) + [361, 362, 363, 511, 512, 513, 700, 4096]:
        a = np.exp(rng.standard_normal(n).astype(np.float32))
        assert c_oracle.numpy_sum_f32(a) == a.sum(), n


def test_masked_policy_model():
    """network.py:142-146 restated on top of NumPy's own exp: identical to the numpy expression."""
    rng = np.random.Generator(np.random.PCG64(8))
    for A, legal in ((9, [1, 3, 4, 7]), (18, list(range(18))), (362, list(range(0, 362, 3))), (2, [1])):
        logits = rng.standard_normal(A).astype(np.float32)
        want = mcts_oracle.masked_policy_f32(logits, legal)
        got = c_oracle.masked_policy_from_exp(np.exp(logits), legal, A)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('A,N,players,bounds', [(2, 50, 1, (0.0, 100.0)), (9, 25, 2, (-1.0, 1.0)),
                                                 (18, 50, 1, None), (40, 30, 2, None)])
def test_python_and_c_oracles_agree_on_ragged_synthetic_trees(A, N, players, bounds):
    """Both restatements on ragged legal sets (empty, permuted, partial) -- the inputs the GPU tests use."""
    T = 12
    inp = synthetic_inputs(T, A, N, seed=100 + A, reward_mode='tanh' if players == 2 else 'uniform',
                           legal_mode='ragged')
    pool = c_oracle.OraclePool(T, A, N, players, 0.997, 0.0, bounds)
    res = run_synthetic_on_pool(pool, inp, NumpyAdapter)
    for t in range(T):
        nl = int(inp['n_legal'][t])
        if nl == 0:
            assert (res['trace_parent'][t] == -1).all() and res['action'][t] == -1
            continue
        tree = mcts_oracle.TreeOracle(A, N, players, 0.997, 0.0, bounds)
        tree.expand_root(inp['priors'][0, t], inp['legal'][t, :nl])
        tree.add_root_noise(inp['noise'][t], 0.25)
        tp, ta = tree.run(inp['values'][:, t], inp['rewards'][:, t], inp['priors'][:, t])
        np.testing.assert_array_equal(res['trace_parent'][t], tp)
        np.testing.assert_array_equal(res['trace_action'][t], ta)
        np.testing.assert_array_equal(res['visits'][t], tree.root_visits())
        assert res['root_value'][t] == tree.root_value()
        assert res['action'][t] == tree.select_action(float(inp['u'][t]), 1.0)
