"""mz_replay_targets (csrc/mz_replay.hip) through the C ABI against the golden vectors generated from the reference's
ReplayBuffer.preprocess_history (replay_buffer.py:66-99) and against oracle/replay_oracle.py on larger ragged
batches.  Bit-exact: integer actions, fp64 value sums rounded once to float32, IEEE float32 two-hot arithmetic."""
import numpy as np
import pytest

from conftest import REPLAY_FIXTURES, load_golden

pytestmark = pytest.mark.gpu


def make_config(A, players, discount, unroll, td, value_support, value_bounds):
    from muzero_amd.config import GameConfig, MuZeroConfig, ReplayBufferConfig, ScalarConfig, TrainingConfig
    from muzero_amd.utils import KnownBounds
    lo, hi = value_bounds
    return MuZeroConfig(
        game_config=GameConfig(name='g', environment_class=None, environment_parameters={}, action_space_size=A,
                               num_players=players, discount=discount),
        replay_buffer_config=ReplayBufferConfig(), training_config=TrainingConfig(num_unroll_steps=unroll, td_steps=td),
        mcts_config=None, network_config=None,
        reward_config=ScalarConfig(known_bounds=KnownBounds(0.0, 1.0), support_size=None, loss_decay=1.0),
        value_config=ScalarConfig(known_bounds=KnownBounds(None if np.isnan(lo) else float(lo), float(hi)),
                                  support_size=value_support, loss_decay=1.0))


def history_from(arrays, obs_shape=(1,)):
    import torch
    from muzero_amd.selfplay import BatchedHistory
    M, G, A = arrays['policies'].shape
    h = BatchedHistory(M, G, obs_shape, A, torch.device('cuda'))
    h.actions.copy_(torch.from_numpy(arrays['actions'].astype(np.int32)))
    h.rewards.copy_(torch.from_numpy(arrays['rewards'].astype(np.float32)))
    h.root_values.copy_(torch.from_numpy(arrays['root_values'].astype(np.float64)))
    h.policies.copy_(torch.from_numpy(arrays['policies'].astype(np.float32)))
    h.lengths.copy_(torch.from_numpy(np.asarray(arrays['lengths']).astype(np.int32)))
    return h


@pytest.mark.parametrize('name', REPLAY_FIXTURES)
def test_replay_targets_equal_reference_golden(name):
    import torch
    from muzero_amd.replay import compute_targets
    fx = load_golden(name)
    support = int(fx['value_support'])
    cfg = make_config(int(fx['A']), int(fx['players']), float(fx['discount']), int(fx['unroll']), int(fx['td']),
                      support if support > 0 else None, tuple(fx['value_bounds']))
    assert (fx['rewards'].astype(np.float32).astype(np.float64) == fx['rewards']).all()
    got = compute_targets(history_from(fx), cfg, random_actions=torch.from_numpy(fx['random_actions']).cuda())
    np.testing.assert_array_equal(got['extended_actions'].cpu().numpy(), fx['expected_extended_actions'])
    np.testing.assert_array_equal(got['target_rewards'].cpu().numpy()[..., 0], fx['expected_target_rewards'])
    np.testing.assert_array_equal(got['target_values'].cpu().numpy(), fx['expected_target_values'])
    np.testing.assert_array_equal(got['target_policies'].cpu().numpy(), fx['expected_target_policies'])
    np.testing.assert_array_equal(got['total_value'].cpu().numpy(), fx['expected_total_value'])


@pytest.mark.parametrize('case', [dict(A=9, players=2, discount=1.0, unroll=2, td=9, support=None, bounds=(np.nan, 1.0), M=9),
                                  dict(A=2, players=1, discount=0.99, unroll=2, td=100, support=100, bounds=(0.0, 100.0), M=160),
                                  dict(A=18, players=1, discount=0.997, unroll=5, td=10, support=None, bounds=(np.nan, 10.0), M=50)])
def test_replay_targets_equal_oracle_on_ragged_batches(case):
    import torch
    from muzero_amd.replay import compute_targets
    from oracle import replay_oracle
    rng = np.random.default_rng(5)
    G, M, A, K = 700, case['M'], case['A'], case['unroll']
    lengths = rng.integers(1, M + 1, G).astype(np.int32)
    lengths[:3] = [1, M, M]
    arrays = dict(actions=rng.integers(0, A, (M, G)).astype(np.int32),
                  rewards=(rng.random((M, G)) < 0.7).astype(np.float32) if case['support'] else rng.normal(size=(M, G)).astype(np.float32),
                  root_values=rng.uniform(0 if case['support'] else -case['bounds'][1], case['bounds'][1], (M, G)),
                  policies=rng.dirichlet(np.ones(A), (M, G)).astype(np.float32), lengths=lengths)
    rand = rng.integers(0, A, (G, K)).astype(np.int32)
    cfg = make_config(A, case['players'], case['discount'], K, case['td'], case['support'], case['bounds'])
    got = {k: v.cpu().numpy() for k, v in compute_targets(history_from(arrays), cfg,
                                                          random_actions=torch.from_numpy(rand).cuda()).items()
           if not k.startswith('_')}
    disc = replay_oracle.effective_discount(case['discount'], case['players'])
    for g in range(0, G, 7):
        L = int(lengths[g])
        want = replay_oracle.preprocess(arrays['actions'][:L, g], arrays['rewards'][:L, g], arrays['root_values'][:L, g],
                                        arrays['policies'][:L, g], rand[g], A, K, case['td'], disc, case['support'],
                                        case['bounds'])
        np.testing.assert_array_equal(got['extended_actions'][:L + K, g], want['extended_actions'])
        np.testing.assert_array_equal(got['target_rewards'][:L + K, g, 0], want['target_rewards'])
        np.testing.assert_array_equal(got['target_values'][:L + K + 1, g], want['target_values'])
        np.testing.assert_array_equal(got['target_policies'][:L + K + 1, g], want['target_policies'])
        assert got['total_value'][g] == want['total_value']
        assert (got['target_values'][L + K + 1:, g] == 0).all() and (got['extended_actions'][L + K:, g] == 0).all()


def test_selfplay_history_feeds_replay_targets():
    """End of the pipeline: games played by the lockstep engine -> targets, checked per game against the oracle."""
    from muzero_amd.games import tictactoe
    from muzero_amd.replay import compute_targets, effective_discount
    from muzero_amd.selfplay import SelfPlayEngine
    from oracle import replay_oracle
    config = tictactoe.make_config(num_simulations=5)
    net = config.network_config.network_class(config, seed=2)
    hist = SelfPlayEngine(config, net, 64, seed=9).play_games()
    out = compute_targets(hist, config)
    K, td = config.training_config.num_unroll_steps, config.training_config.td_steps
    host = {k: getattr(hist, k).cpu().numpy() for k in hist.FIELDS}
    rand = out['random_actions'].cpu().numpy()
    assert ((rand >= 0) & (rand < 9)).all()
    for g in range(hist.G):
        L = int(host['lengths'][g])
        want = replay_oracle.preprocess(host['actions'][:L, g], host['rewards'][:L, g], host['root_values'][:L, g],
                                        host['policies'][:L, g], rand[g], 9, K, td, effective_discount(config))
        np.testing.assert_array_equal(out['target_values'].cpu().numpy()[:L + K + 1, g], want['target_values'])
        np.testing.assert_array_equal(out['extended_actions'].cpu().numpy()[:L + K, g], want['extended_actions'])
        assert out['total_value'][g].item() == want['total_value']
