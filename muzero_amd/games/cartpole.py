"""CartPole: configuration (reference games/cartpole.py:12-60) and network class (:63-139)."""
from ..config import (GameConfig, MCTSConfig, MuZeroConfig, NetworkConfig, ReplayBufferConfig, ScalarConfig,
                      TrainingConfig)
from ..network import CartPoleNetwork
from ..utils import KnownBounds

__all__ = ['make_config', 'CartPoleNetwork']


def make_config(num_simulations: int = 4, environment_class=None) -> MuZeroConfig:
    """Reference defaults (4 simulations/move); BASELINE.json's CartPole config uses 50."""
    if environment_class is None:
        from ..environment import CartPoleEnvironment
        environment_class = CartPoleEnvironment
    return MuZeroConfig(
        game_config=GameConfig(name='CartPole', environment_class=environment_class, environment_parameters={},
                               action_space_size=2, num_players=1, discount=0.99),
        replay_buffer_config=ReplayBufferConfig(window_size=int(1e3), prefetch_buffer_size=10),
        training_config=TrainingConfig(batch_size=128, training_steps=int(2e5), checkpoint_interval=int(1e3),
                                       replay_buffer_loginterval=100, num_unroll_steps=2, td_steps=100),
        mcts_config=MCTSConfig(max_moves=500, root_dirichlet_alpha=1.0, root_exploration_fraction=0.25,
                               num_simulations=num_simulations, temperature=1.0, freezing_moves=50, default_value=50.0),
        network_config=NetworkConfig(network_class=CartPoleNetwork, hidden_state_size=128, hidden_size=128,
                                     support_size=100),
        reward_config=ScalarConfig(known_bounds=KnownBounds(minv=0.0, maxv=1.0), support_size=None, loss_decay=1.0),
        value_config=ScalarConfig(known_bounds=KnownBounds(minv=0.0, maxv=100.0), support_size=100, loss_decay=0.1))
