"""The single-game host TicTacToeEnvironment against every transition of the UNMODIFIED reference environment
(games/tictactoe.py:71-148): tests/golden/ttt_env_transitions.npz, made by tests/golden/make_golden_env.py -- every
reachable position x every legal move, plus a sample of the moves the reference refuses."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ttt_env_transitions.npz')


def test_fixture_covers_every_way_a_game_ends():
    fx = dict(np.load(GOLDEN))          # decompress every array once
    K = len(fx['action'])
    assert K == 16167 and fx['board'].shape == (K, 9)
    lines = ((0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8), (2, 4, 6))
    won = fx['reward'] == 1.0
    mover = fx['to_play']
    for line in lines:                       # each of the eight win lines ends games of both players
        full = (fx['next_board'][:, line] == mover[:, None]).all(axis=1)
        assert (full <= won).all()
        for player in (0, 1):
            assert (full & (mover == player)).any(), (line, player)
    assert ((fx['ended'] == 1) & ~won).sum() == 80 and (fx['next_steps'][(fx['ended'] == 1) & ~won] == 9).all()
    assert (fx['next_n_legal'][fx['ended'] == 1] == 0).all()


def test_host_environment_matches_every_reference_transition():
    from muzero_amd.environment import TicTacToeEnvironment
    fx = dict(np.load(GOLDEN))          # decompress every array once
    env = TicTacToeEnvironment()
    s0 = env.reset()
    np.testing.assert_array_equal(np.asarray(s0.observation, np.float32).reshape(27), fx['reset_observation'])
    assert list(s0.legal_actions) == [a for a in fx['reset_legal'] if a >= 0] and s0.to_play == 0
    for k in range(len(fx['action'])):
        env.reset()
        env.board[:] = fx['board'][k]
        env.to_play, env.steps = int(fx['to_play'][k]), int(fx['steps'][k])
        st, reward, done, info = env.step(int(fx['action'][k]))
        assert reward == fx['reward'][k] and bool(done) == bool(fx['ended'][k]) and info == {}
        assert st.to_play == fx['next_to_play'][k] and env.steps == fx['next_steps'][k]
        np.testing.assert_array_equal(env.board, fx['next_board'][k])
        np.testing.assert_array_equal(np.asarray(st.observation, np.float32).reshape(27), fx['next_observation'][k])
        n = int(fx['next_n_legal'][k])
        assert list(st.legal_actions) == list(fx['next_legal'][k, :n])


def test_host_environment_refuses_what_the_reference_refuses():
    from muzero_amd.environment import TicTacToeEnvironment
    from muzero_amd.exceptions import MuZeroEnvironmentError
    fx = dict(np.load(GOLDEN))          # decompress every array once
    env = TicTacToeEnvironment()
    for k in range(0, len(fx['illegal_action']), 5):
        env.reset()
        env.board[:] = fx['illegal_board'][k]
        env.to_play = int(fx['illegal_to_play'][k])
        env.steps = int((fx['illegal_board'][k] >= 0).sum())
        with pytest.raises(MuZeroEnvironmentError):
            env.step(int(fx['illegal_action'][k]))
