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
