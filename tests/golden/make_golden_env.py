#!/usr/bin/env python3
"""Generate tests/golden/ttt_env_transitions.npz from the UNMODIFIED reference TicTacToeEnvironment.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden_env.py

Every position reachable from `reset()` through legal moves x every legal move in it, stepped by the reference's own
`TicTacToeEnvironment.step` (games/tictactoe.py:121-141) under the stub harness of make_golden.py (TensorFlow and gym are
absent; `tf.stack` / `tf.where` / `tf.ones` are numpy stand-ins, `np.int` is aliased).  A position is reached by
replaying one recorded move sequence on a fresh reference environment, so nothing but `reset()` and `step()` ever
touches its state.  The fixture holds data only:

    board[K,9] i8, to_play[K], steps[K]            the position before the move (never ended)
    action[K]                                      the legal move played
    next_board[K,9], next_to_play[K], next_steps[K], reward[K] f32, ended[K]
    next_observation[K,27] f32                     State.observation after the move, (3,3,3) flattened
    next_legal[K,9] i32 (-1 padded), next_n_legal[K]
    reset_observation[27], reset_legal[9]
    illegal_board[M,9], illegal_to_play[M], illegal_action[M]   moves the reference refuses with MuZeroEnvironmentError

It pins muzero_amd.environment.TicTacToeEnvironment (CPU test) and the device kernels mz_env_ttt_step / _observe
(GPU test) directly against the reference, for every win line, draw and move count there is.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden  # noqa: E402,F401  (installs the stubs, puts /root/reference on sys.path)

import games.tictactoe as ref_ttt          # noqa: E402  (reference, unmodified)
from exceptions import MuZeroEnvironmentError as RefEnvError   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ttt_env_transitions.npz')


def replay(path):
    env = ref_ttt.TicTacToeEnvironment()
    state = env.reset()
    for a in path:
        state, _, _, _ = env.step(a)
    return env, state


def main():
    env, state0 = replay(())
    key0 = (tuple(int(v) for v in env.board.reshape(-1)), int(env.to_play))
    paths = {key0: ()}                      # position -> one move sequence that reaches it
    frontier = [key0]
    rows = []
    illegal = []
    while frontier:
        nxt = []
        for key in frontier:
            path = paths[key]
            env, state = replay(path)
            assert not env.ended
            board = np.array(key[0], np.int8)
            legal = list(state.legal_actions)
            assert legal == [a for a in range(9) if board[a] == -1]
            for a in range(-1, 10):         # the moves the reference refuses in this position
                if a not in legal:
                    e2, _ = replay(path)
                    try:
                        e2.step(a)
                    except RefEnvError:
                        illegal.append((board.copy(), key[1], a))
                    else:
                        raise AssertionError(f'the reference accepted illegal move {a} at {key}')
            for a in legal:
                e2, _ = replay(path)
                st, reward, done, info = e2.step(a)
                nb = tuple(int(v) for v in e2.board.reshape(-1))
                obs = np.asarray(st.observation, np.float32)
                assert obs.shape == (3, 3, 3) and info == {}
                nl = np.full(9, -1, np.int32)
                nl[:len(st.legal_actions)] = st.legal_actions
                rows.append(dict(board=board, to_play=key[1], steps=len(path), action=a, next_board=np.array(nb, np.int8),
                                 next_to_play=int(st.to_play), next_steps=int(e2.steps), reward=np.float32(reward),
                                 ended=int(bool(done)), next_observation=obs.reshape(27), next_legal=nl,
                                 next_n_legal=len(st.legal_actions)))
                assert int(st.to_play) == int(e2.to_play) and bool(done) == bool(e2.ended)
                k2 = (nb, int(e2.to_play))
                if not done and k2 not in paths:
                    paths[k2] = path + (a,)
                    nxt.append(k2)
        frontier = nxt
    stack = lambda name, dtype: np.array([r[name] for r in rows], dtype)
    # a sample of the refused moves is enough (they are all "occupied cell", -1 or 9): every 7th
    illegal = illegal[::7]
    legal0 = np.full(9, -1, np.int32)
    legal0[:len(state0.legal_actions)] = state0.legal_actions
    np.savez_compressed(
        OUT, board=stack('board', np.int8), to_play=stack('to_play', np.int32), steps=stack('steps', np.int32),
        action=stack('action', np.int32), next_board=stack('next_board', np.int8),
        next_to_play=stack('next_to_play', np.int32), next_steps=stack('next_steps', np.int32),
        reward=stack('reward', np.float32), ended=stack('ended', np.int32),
        next_observation=stack('next_observation', np.float32), next_legal=stack('next_legal', np.int32),
        next_n_legal=stack('next_n_legal', np.int32),
        reset_observation=np.asarray(state0.observation, np.float32).reshape(27), reset_legal=legal0,
        illegal_board=np.array([b for b, _, _ in illegal], np.int8),
        illegal_to_play=np.array([t for _, t, _ in illegal], np.int32),
        illegal_action=np.array([a for _, _, a in illegal], np.int32))
    wins = int(sum(r['reward'] == 1.0 for r in rows))
    draws = int(sum(r['ended'] and r['reward'] == 0.0 for r in rows))
    print(f'{len(paths)} non-terminal positions, {len(rows)} transitions ({wins} winning moves, {draws} draws), '
          f'{len(illegal)} refused moves kept -> {OUT} ({os.path.getsize(OUT)} bytes)')


if __name__ == '__main__':
    main()
