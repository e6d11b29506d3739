"""Environment API of the reference (environment.py:12-63) and the two single-game host environments.

`Environment.step(action) -> (State, reward, done, info)` and `reset() -> State` keep the reference contract.
TicTacToeEnvironment follows games/tictactoe.py:71-148.  CartPoleEnvironment restates the CartPole-v1 physics the
reference obtains from the third-party `gym` package (environment.py:44-63; gym is absent and unpinned, so this
is "parity unpinned": the classic Barto-Sutton-Anderson equations with Euler integration).
The batched device-side environments used for many concurrent games live in vector_env.py.
"""
import math
from typing import List, Optional, Tuple

import numpy as np

from .exceptions import MuZeroEnvironmentError, MuZeroImplementationError
from .muzero_types import State


class Environment:
    def __init__(self, action_space_size: int, num_players: int) -> None:
        self.action_space_size = action_space_size
        self.num_players = num_players

    def step(self, action: int) -> Tuple[State, float, bool, dict]:
        raise MuZeroImplementationError('step', 'Environment')

    def reset(self) -> State:
        raise MuZeroImplementationError('reset', 'Environment')


WIN_LINES = ((0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8), (2, 4, 6))


class TicTacToeEnvironment(Environment):
    """3x3 board, cells -1 (empty) / 0 / 1; observation planes [mine-as-player-0, player-1, to_play] (3,3,3);
    reward 1.0 to the mover on a winning move, draw after 9 moves (games/tictactoe.py:107-148)."""
    NO_REWARD = 0.0
    WINNING_MOVE = 1.0

    def __init__(self) -> None:
        super().__init__(action_space_size=9, num_players=2)
        self.board: Optional[np.ndarray] = None
        self.to_play: Optional[int] = None
        self.steps: Optional[int] = None
        self.ended: Optional[bool] = None

    def __repr__(self) -> str:
        if self.board is None:
            return ''
        symbols = {-1: '.', 0: 'X', 1: 'O'}
        return ''.join(''.join(symbols[int(v)] for v in row) + '\n' for row in self.board.reshape(3, 3))

    def _legal_actions(self) -> List[int]:
        if self.ended:
            return []
        return [a for a in range(9) if self.board[a] == -1]

    def _get_state(self) -> State:
        grid = self.board.reshape(3, 3)
        observation = np.stack([(grid == 0), (grid == 1), np.full((3, 3), self.to_play)], axis=-1).astype(np.float32)
        return State(observation, self.to_play, self._legal_actions())

    def step(self, action: int) -> Tuple[State, float, bool, dict]:
        if self.ended or action not in range(9) or self.board[action] != -1:
            raise MuZeroEnvironmentError(f'cannot perform action {action}')
        self.board[action] = self.to_play
        won = any(all(self.board[i] == self.to_play for i in line) for line in WIN_LINES)
        if won:
            self.ended = True
        reward = self.WINNING_MOVE if won else self.NO_REWARD
        self.to_play = 1 - self.to_play
        self.steps += 1
        if self.steps == 9:
            self.ended = True
        return self._get_state(), reward, self.ended, {}

    def reset(self) -> State:
        self.board = -np.ones(9, dtype=np.int64)
        self.steps = 0
        self.ended = False
        self.to_play = 0
        return self._get_state()


class CartPoleEnvironment(Environment):
    """CartPole-v1: 2 actions, observation (x, x_dot, theta, theta_dot) float32, reward 1 per step, episode ends
    when |x| > 2.4, |theta| > 12 degrees or after 500 steps."""
    GRAVITY, MASSCART, MASSPOLE, LENGTH, FORCE_MAG, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    THETA_THRESHOLD = 12 * 2 * math.pi / 360
    X_THRESHOLD = 2.4
    MAX_STEPS = 500

    def __init__(self, seed: Optional[int] = None) -> None:
        super().__init__(action_space_size=2, num_players=1)
        self.rng = np.random.default_rng(seed)
        self.state = None
        self.steps = 0

    def _state(self) -> State:
        return State(np.asarray(self.state, np.float32), 0, [0, 1])

    def reset(self) -> State:
        self.state = self.rng.uniform(-0.05, 0.05, size=4)
        self.steps = 0
        return self._state()

    def step(self, action: int) -> Tuple[State, float, bool, dict]:
        if action not in (0, 1):
            raise MuZeroEnvironmentError(f'cannot perform action {action}')
        x, x_dot, theta, theta_dot = self.state
        force = self.FORCE_MAG if action == 1 else -self.FORCE_MAG
        total_mass = self.MASSPOLE + self.MASSCART
        polemass_length = self.MASSPOLE * self.LENGTH
        costheta, sintheta = math.cos(theta), math.sin(theta)
        temp = (force + polemass_length * theta_dot ** 2 * sintheta) / total_mass
        thetaacc = (self.GRAVITY * sintheta - costheta * temp) / (
            self.LENGTH * (4.0 / 3.0 - self.MASSPOLE * costheta ** 2 / total_mass))
        xacc = temp - polemass_length * thetaacc * costheta / total_mass
        x, x_dot = x + self.TAU * x_dot, x_dot + self.TAU * xacc
        theta, theta_dot = theta + self.TAU * theta_dot, theta_dot + self.TAU * thetaacc
        self.state = np.array([x, x_dot, theta, theta_dot])
        self.steps += 1
        done = bool(abs(x) > self.X_THRESHOLD or abs(theta) > self.THETA_THRESHOLD or self.steps >= self.MAX_STEPS)
        return self._state(), 1.0, done, {}
