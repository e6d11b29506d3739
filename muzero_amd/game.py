"""Episode book-keeping with the reference's field set (game.py:8-75)."""
from typing import Any, Dict, List

from .environment import Environment
from .muzero_types import State


class GameHistory:
    """What a finished game hands to the replay buffer (game.py:8-41; protos/replay_buffer.proto:30-42)."""

    def __init__(self) -> None:
        self.observations: List[Any] = []
        self.to_plays: List[int] = []
        self.actions: List[int] = []
        self.rewards: List[float] = []
        self.root_values: List[float] = []
        self.policies: List[Any] = []
        self.metadata: Dict[str, Any] = {}
        # filled by a replay buffer, never by self-play
        self.extended_actions = None
        self.target_rewards = None
        self.target_values = None
        self.target_policies = None
        self.total_value = float('nan')

    def make_image(self, index: int = -1):
        return self.observations[index]

    def __len__(self) -> int:
        return len(self.actions)

    def __repr__(self) -> str:
        return 'Game({})'.format(', '.join(str(a) for a in self.actions))


class Game:
    """One episode against an Environment (game.py:43-75)."""

    def __init__(self, environment: Environment) -> None:
        self.environment = environment
        self.history = GameHistory()
        self.state: State = environment.reset()
        self.history.observations.append(self.state.observation)
        self.history.to_plays.append(self.state.to_play)
        self.ended = False

    def to_play(self) -> int:
        return self.state.to_play

    def legal_actions(self) -> List[int]:
        return self.state.legal_actions

    def terminal(self) -> bool:
        return self.ended

    def apply(self, action: int) -> None:
        self.state, reward, self.ended, _info = self.environment.step(action)
        h = self.history
        h.observations.append(self.state.observation)
        h.to_plays.append(self.state.to_play)
        h.actions.append(action)
        h.rewards.append(reward)

    def store_search_statistics(self, value: float, policy) -> None:
        self.history.root_values.append(value)
        self.history.policies.append(policy)
