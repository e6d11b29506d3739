"""Configuration objects with the reference's names and fields (config.py:15-173), minus TensorFlow.

Only what the self-play hot path reads is functional: GameConfig, MCTSConfig, ScalarConfig (value/reward
scalarisation), NetworkConfig and MuZeroConfig.{action_space, visit_softmax_temperature_fn,
exploration_function, make_uniform_network, create_environment}.  ReplayBufferConfig / TrainingConfig are plain
records kept so that a reference-style make_config() can be written unchanged (training is out of scope).
"""
from dataclasses import dataclass, field
from math import log
from typing import Any, Callable, Dict, Optional

from .utils import KnownBounds, scalar_to_support, support_to_scalar


class ScalarConfig:
    """How a value/reward head is (de)scalarised (config.py:15-40)."""

    def __init__(self, known_bounds: Optional[KnownBounds], support_size: Optional[int], loss_decay: float = 0.0):
        self.known_bounds = known_bounds
        self.support_size = support_size
        self.loss_decay = loss_decay

    def _range(self):
        return self.known_bounds.endpoints() if self.known_bounds is not None else (0.0, 1.0)

    def to_scalar(self, tensor):
        if self.support_size is None:
            return tensor
        a, b = self._range()
        return a + (b - a) * support_to_scalar(tensor, support_size=self.support_size)

    def inv_to_scalar(self, tensor):
        if self.support_size is None:
            return tensor.unsqueeze(-1)
        a, b = self._range()
        return scalar_to_support((tensor - a) / (b - a), support_size=self.support_size)


@dataclass
class GameConfig:
    name: str
    environment_class: Optional[Callable[..., Any]]
    environment_parameters: Dict[str, Any]
    action_space_size: int
    num_players: int
    discount: float


@dataclass
class ReplayBufferConfig:
    window_size: int = 0
    prefetch_buffer_size: int = 0


@dataclass
class TrainingConfig:
    optimizer: Any = None
    batch_size: int = 0
    training_steps: int = 0
    checkpoint_interval: int = 0
    replay_buffer_loginterval: int = 0
    num_unroll_steps: int = 0
    td_steps: int = 0
    steps_per_execution: int = 1


@dataclass
class MCTSConfig:
    max_moves: int
    root_dirichlet_alpha: float
    root_exploration_fraction: float
    num_simulations: int
    temperature: float
    freezing_moves: int
    default_value: float


class NetworkConfig:
    REPRESENTATION = 'representation'
    DYNAMICS = 'dynamics'
    PREDICTION = 'prediction'
    INITIAL_INFERENCE = 'initial_inference'
    RECURRENT_INFERENCE = 'recurrent_inference'
    OBSERVATION = 'observation'
    ACTION = 'action'
    HIDDEN_STATE = 'hidden_state'

    def __init__(self, network_class: Callable, **network_parameters) -> None:
        self.network_class = network_class
        self.network_parameters = network_parameters


class MuZeroConfig:
    def __init__(self, game_config: GameConfig, replay_buffer_config, training_config, mcts_config: MCTSConfig,
                 network_config: Optional[NetworkConfig], reward_config: ScalarConfig, value_config: ScalarConfig):
        self.game_config = game_config
        self.replay_buffer_config = replay_buffer_config
        self.training_config = training_config
        self.mcts_config = mcts_config
        self.network_config = network_config
        self.reward_config = reward_config
        self.value_config = value_config

    def action_space(self):
        return list(range(self.game_config.action_space_size))

    def create_environment(self):
        return self.game_config.environment_class(**self.game_config.environment_parameters)

    def make_uniform_network(self):
        return self.network_config.network_class(self, **self.network_config.network_parameters)

    def visit_softmax_temperature_fn(self, num_moves: int, training_steps: int) -> float:
        """config.py:153-157"""
        return self.mcts_config.temperature if num_moves < self.mcts_config.freezing_moves else 0.0

    def exploration_function(self, parent_visit_count: int, child_visit_count: int) -> float:
        """config.py:159-160 (tabulated on the device as log_table[parent] / (child + 1))"""
        return 2 * log(parent_visit_count + 1) / (child_visit_count + 1)
