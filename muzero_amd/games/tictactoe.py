"""Tic-tac-toe: configuration (reference games/tictactoe.py:16-68) and network class (:151-220)."""
from ..config import (GameConfig, MCTSConfig, MuZeroConfig, NetworkConfig, ReplayBufferConfig, ScalarConfig,
                      TrainingConfig)
from ..network import TicTacToeNetwork
from ..utils import KnownBounds

__all__ = ['make_config', 'TicTacToeNetwork']


def make_config(num_simulations: int = 20, environment_class=None) -> MuZeroConfig:
    """Reference defaults; BASELINE.json's configs use num_simulations=25."""
    if environment_class is None:
        from ..environment import TicTacToeEnvironment
        environment_class = TicTacToeEnvironment
    return MuZeroConfig(
        game_config=GameConfig(name='TicTacToe', environment_class=environment_class, environment_parameters={},
                               action_space_size=9, num_players=2, discount=1.0),
        replay_buffer_config=ReplayBufferConfig(window_size=int(1e4), prefetch_buffer_size=10),
        training_config=TrainingConfig(batch_size=128, training_steps=int(5e4), checkpoint_interval=int(5e2),
                                       replay_buffer_loginterval=50, num_unroll_steps=2, td_steps=9),
        mcts_config=MCTSConfig(max_moves=9, root_dirichlet_alpha=1.0, root_exploration_fraction=0.25,
                               num_simulations=num_simulations, temperature=1.0, freezing_moves=9, default_value=0.0),
        network_config=NetworkConfig(network_class=TicTacToeNetwork, hidden_state_size=128, hidden_size=128),
        reward_config=ScalarConfig(known_bounds=KnownBounds(minv=0.0, maxv=1.0), support_size=None, loss_decay=0.0),
        value_config=ScalarConfig(known_bounds=KnownBounds(minv=None, maxv=1.0), support_size=None, loss_decay=4.0))
