"""Synthetic Atari-shaped game of BASELINE.json configs[3]: 18 actions, 256-d hidden state, conv-residual dynamics,
unknown value bounds (live MinMaxStats), discount 0.997, 50 simulations per move.

The reference has no such game (its two are games/tictactoe.py and games/cartpole.py); this module follows their layout
-- make_config() returning a MuZeroConfig whose network_class is the HIP-backed network -- so that the lockstep search
(selfplay.LockstepSearch) runs it like the others.  There is no environment: observations are synthetic (2,128) tensors."""
from ..config import (GameConfig, MCTSConfig, MuZeroConfig, NetworkConfig, ReplayBufferConfig, ScalarConfig,
                      TrainingConfig)
from ..network import AtariShapedNetwork

__all__ = ['make_config', 'AtariShapedNetwork']


def make_config(num_simulations: int = 50) -> MuZeroConfig:
    return MuZeroConfig(
        game_config=GameConfig(name='AtariShaped', environment_class=None, environment_parameters={},
                               action_space_size=18, num_players=1, discount=0.997),
        replay_buffer_config=ReplayBufferConfig(window_size=int(1e4), prefetch_buffer_size=10),
        training_config=TrainingConfig(batch_size=128, training_steps=int(5e4), checkpoint_interval=int(5e2),
                                       replay_buffer_loginterval=50, num_unroll_steps=5, td_steps=10),
        mcts_config=MCTSConfig(max_moves=500, root_dirichlet_alpha=0.25, root_exploration_fraction=0.25,
                               num_simulations=num_simulations, temperature=1.0, freezing_moves=500, default_value=0.0),
        network_config=NetworkConfig(network_class=AtariShapedNetwork, hidden_state_size=128, hidden_size=128),
        reward_config=ScalarConfig(known_bounds=None, support_size=None, loss_decay=0.0),
        value_config=ScalarConfig(known_bounds=None, support_size=None, loss_decay=0.0))
