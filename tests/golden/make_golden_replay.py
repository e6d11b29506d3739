#!/usr/bin/env python3
"""Golden vectors for the replay-side targets from the UNMODIFIED reference
`ReplayBuffer.preprocess_history` / `compute_target_value` (replay_buffer.py:66-99).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_replay.py

Harness (nothing from the reference is edited or copied): the stub `tensorflow` of make_golden.py plus numpy-backed
`tf.concat`, `tf.math.sqrt/floor/ceil`, `tf.ones_like`, `tf.cast`, `tf.one_hot` in float32, and an empty `game_services`
module (the reference's needs generated protobuf code that is not in its tree).  `random.randrange` draws (the random
actions appended past the terminal state) are recorded.  The scalar (support_size None) targets involve no TensorFlow
arithmetic and pin the oracle exactly; the support (two-hot) targets go through the harness' float32 numpy stand-ins for
TF kernels, so they pin the restatement's formula, not TensorFlow's rounding ("parity unpinned" for that branch).
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  installs the stubs and puts /root/reference on sys.path

tf = sys.modules['tensorflow']
T = mg.FakeTensor
tf.concat = lambda xs, axis=0: np.concatenate([np.asarray(x) for x in xs], axis=axis).view(T)
tf.ones_like = lambda x: np.ones_like(np.asarray(x)).view(T)
tf.cast = lambda x, dtype: np.asarray(x).astype(dtype).view(T)
tf.one_hot = lambda idx, depth: np.eye(depth, dtype=np.float32)[np.asarray(idx)].view(T)
tf.sqrt = lambda x: np.sqrt(np.asarray(x, dtype=np.float32)).view(T)
tf.math.sqrt = tf.sqrt
tf.math.floor = lambda x: np.floor(np.asarray(x)).view(T)
tf.math.ceil = lambda x: np.ceil(np.asarray(x)).view(T)
gs = types.ModuleType('game_services')
gs.save_games = gs.load_games = lambda *a, **k: None
sys.modules['game_services'] = gs

import replay_buffer as ref_replay            # noqa: E402  (reference, unmodified)
import config as ref_config                   # noqa: E402
import game as ref_game                       # noqa: E402
import utils as ref_utils                     # noqa: E402


def make_cfg(A, players, discount, unroll, td, value_support, value_bounds, reward_bounds=(0.0, 1.0)):
    kb = ref_utils.KnownBounds
    return ref_config.MuZeroConfig(
        game_config=ref_config.GameConfig(name='g', environment_class=None, environment_parameters={},
                                          action_space_size=A, num_players=players, discount=discount),
        replay_buffer_config=ref_config.ReplayBufferConfig(window_size=10, prefetch_buffer_size=1),
        mcts_config=None,
        training_config=ref_config.TrainingConfig(optimizer=None, batch_size=1, training_steps=1, checkpoint_interval=1,
                                                  replay_buffer_loginterval=1, num_unroll_steps=unroll, td_steps=td,
                                                  steps_per_execution=1),
        network_config=None,
        value_config=ref_config.ScalarConfig(known_bounds=kb(*value_bounds), support_size=value_support, loss_decay=1.0),
        reward_config=ref_config.ScalarConfig(known_bounds=kb(*reward_bounds), support_size=None, loss_decay=1.0))


def run_case(name, seed, lengths, A, players, discount, unroll, td, value_support, value_bounds, reward_mode):
    rng = np.random.default_rng(seed)
    cfg = make_cfg(A, players, discount, unroll, td, value_support, value_bounds)
    buf = ref_replay.ReplayBuffer(cfg)
    out = dict(A=A, players=players, discount=discount, unroll=unroll, td=td,
               value_support=-1 if value_support is None else value_support,
               value_bounds=np.array([np.nan if b is None else b for b in value_bounds]), lengths=np.array(lengths))
    M = max(lengths)
    G = len(lengths)
    actions = np.zeros((M, G), np.int32)
    rewards = np.zeros((M, G), np.float64)
    root_values = np.zeros((M, G), np.float64)
    policies = np.zeros((M, G, A), np.float32)
    ext_rand = np.zeros((G, unroll), np.int32)
    exp = dict(extended_actions=np.zeros((M + unroll, G), np.int32), target_rewards=np.zeros((M + unroll, G), np.float32),
               target_policies=np.zeros((M + unroll + 1, G, A), np.float32), total_value=np.zeros(G, np.float64))
    vdim = 1 if value_support is None else value_support + 1
    exp['target_values'] = np.zeros((M + unroll + 1, G, vdim), np.float32)
    for g, L in enumerate(lengths):
        h = ref_game.GameHistory()
        h.actions = [int(a) for a in rng.integers(0, A, L)]
        if reward_mode == 'win':        # tic-tac-toe: 1.0 on the last move of some games
            h.rewards = [0.0] * (L - 1) + [1.0 if rng.random() < 0.6 else 0.0]
        elif reward_mode == 'one':
            h.rewards = [1.0] * L
        else:
            h.rewards = [float(np.float32(x)) for x in rng.normal(size=L)]     # float32-representable, like env rewards
        lo = value_bounds[0] if value_bounds[0] is not None else -value_bounds[1]
        h.root_values = [float(x) for x in rng.uniform(lo, value_bounds[1], L)]
        pol = rng.dirichlet(np.ones(A), L).astype(np.float32)
        h.policies = [tf.constant(p) for p in pol]
        random.seed(seed * 1000 + g)
        state = random.getstate()
        buf.preprocess_history(h)
        random.setstate(state)
        ext_rand[g] = [random.randrange(A) for _ in range(unroll)]
        actions[:L, g], rewards[:L, g], root_values[:L, g], policies[:L, g] = h.actions, h.rewards, h.root_values, pol
        exp['extended_actions'][:L + unroll, g] = np.asarray(h.extended_actions)
        exp['target_rewards'][:L + unroll, g] = np.asarray(h.target_rewards)[:, 0]
        exp['target_values'][:L + unroll + 1, g] = np.asarray(h.target_values).reshape(L + unroll + 1, vdim)
        exp['target_policies'][:L + unroll + 1, g] = np.asarray(h.target_policies)
        exp['total_value'][g] = h.total_value
        assert np.asarray(h.extended_actions).dtype == np.int32 and np.asarray(h.target_values).dtype == np.float32
        assert list(np.asarray(h.extended_actions)[L:]) == list(ext_rand[g])
    out.update(actions=actions, rewards=rewards, root_values=root_values, policies=policies, random_actions=ext_rand)
    out.update({'expected_' + k: v for k, v in exp.items()})
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'games', G, 'max len', M)


def main():
    run_case('replay_ttt_s31', 31, [5, 6, 7, 8, 9, 9, 5, 7], A=9, players=2, discount=1.0, unroll=2, td=9,
             value_support=None, value_bounds=(None, 1.0), reward_mode='win')
    run_case('replay_short_td_s32', 32, [1, 2, 3, 12, 20, 4], A=4, players=2, discount=0.9, unroll=3, td=3,
             value_support=None, value_bounds=(-2.0, 2.0), reward_mode='normal')
    run_case('replay_cartpole_s33', 33, [9, 40, 130, 260], A=2, players=1, discount=0.99, unroll=2, td=100,
             value_support=100, value_bounds=(0.0, 100.0), reward_mode='one')
    run_case('replay_1p_td5_s34', 34, [6, 6, 30], A=18, players=1, discount=0.997, unroll=5, td=5,
             value_support=None, value_bounds=(None, 10.0), reward_mode='normal')


if __name__ == '__main__':
    main()
