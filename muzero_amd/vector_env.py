"""Batched environments on the device: G games advance one move per launch.

The single-game host classes in environment.py keep the reference's `Environment.step/reset` contract
(environment.py:12-63, games/tictactoe.py:71-148); these are their many-games form for the lockstep self-play
loop (`selfplay.SelfPlayEngine`): state lives in device tensors, `observe()` produces the network input and the
legal-action lists the search takes, `step(action)` plays one move in every unfinished game.  A finished game is
frozen and reports `n_legal == 0`, which the search treats as inactive.  The kernels are in csrc/mz_env.hip.
"""
from typing import Optional, Tuple

import torch

from . import _lib
from .exceptions import MuZeroEnvironmentError, MuZeroError


def _stream_ptr(device):
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class VectorEnvironment:
    """G concurrent episodes of one game.  Sub-classes set action_space_size, num_players, observation_size,
    max_moves."""
    action_space_size = 0
    num_players = 0
    observation_size = 0
    supports_fused = False         # as_struct() gives an mz_env_t (the two built-in environments)

    def __init__(self, num_games: int, device='cuda') -> None:
        if not torch.cuda.is_available():
            raise MuZeroError('VectorEnvironment needs a HIP device (torch.cuda.is_available() is False); '
                              'muzero_amd has no CPU fallback')
        self._lib = _lib.load()
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.G = int(num_games)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.steps = torch.zeros(self.G, **i32)
        self.ended = torch.ones(self.G, **i32)             # nothing to play before reset()
        self.to_play = torch.zeros(self.G, **i32)
        self.observation = torch.zeros(self.G, self.observation_size, dtype=torch.float32, device=self.device)
        self.legal = torch.zeros(self.G, self.action_space_size, **i32)
        self.n_legal = torch.zeros(self.G, **i32)
        self.reward = torch.zeros(self.G, dtype=torch.float32, device=self.device)
        self.status = torch.zeros(self.G, **i32)
        self._illegal = torch.zeros(1, **i32)

    def reset(self) -> None:
        raise NotImplementedError

    def reset_where(self, where: torch.Tensor) -> None:
        """Restarts the games with where[g] != 0 (int32 [G]); the others are left as they are."""
        raise NotImplementedError

    def observe(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (observation [G, obs] f32, legal [G, A] i32, n_legal [G] i32); the tensors are reused by the next call."""
        raise NotImplementedError

    def step(self, action: torch.Tensor) -> torch.Tensor:
        """Plays action[g] (i32) in every unfinished game -> reward [G] f32 (reused by the next call)."""
        raise NotImplementedError

    def all_ended(self) -> bool:
        """Synchronises."""
        return bool(self.ended.all().item())

    def check_illegal(self) -> None:
        """Raises like the reference's step would have (games/tictactoe.py:122-123) if any step so far was handed
        an illegal action.  Synchronises."""
        if int(self._illegal.item()) != 0:
            self._illegal.zero_()
            raise MuZeroEnvironmentError('a batched step was handed an illegal action (status == 2 marks the games)')

    def as_struct(self, restart_state: Optional[torch.Tensor] = None):
        """mz_env_t over this environment's state (for the fused self-play calls); None when the environment has none."""
        return None

    def draw_restart_states(self, count: int = 1) -> Optional[torch.Tensor]:
        """The states restarted games begin in, drawn ahead of time for `count` moves ([count, G, ...]; CartPole); None
        when a restart is deterministic."""
        return None

    def _action_ptr(self, action):
        if (not isinstance(action, torch.Tensor) or action.dtype != torch.int32 or tuple(action.shape) != (self.G,)
                or action.device != self.device or not action.is_contiguous()):
            raise MuZeroError(f'action must be a contiguous int32 tensor of shape ({self.G},) on {self.device}')
        return action.data_ptr()


class TicTacToeVectorEnv(VectorEnvironment):
    """games/tictactoe.py:71-148 for G boards."""
    action_space_size = 9
    num_players = 2
    observation_size = 27
    max_moves = 9
    supports_fused = True

    def __init__(self, num_games: int, device='cuda') -> None:
        super().__init__(num_games, device)
        self.board = torch.full((self.G, 9), -1, dtype=torch.int8, device=self.device)

    def as_struct(self, restart_state=None):
        return _lib.Env(_lib.MZ_ENV_TICTACTOE, self.G, self.board.data_ptr(), self.to_play.data_ptr(),
                        self.steps.data_ptr(), self.ended.data_ptr(), None)

    def reset(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mz_env_ttt_reset(_stream_ptr(self.device), self.G, self.board.data_ptr(),
                                                  self.to_play.data_ptr(), self.steps.data_ptr(),
                                                  self.ended.data_ptr()))

    def reset_where(self, where):
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mz_env_ttt_reset_where(_stream_ptr(self.device), self.G, self.board.data_ptr(),
                                                        self.to_play.data_ptr(), self.steps.data_ptr(),
                                                        self.ended.data_ptr(), self._action_ptr(where)))

    def observe(self):
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mz_env_ttt_observe(_stream_ptr(self.device), self.G, self.board.data_ptr(),
                                                    self.to_play.data_ptr(), self.ended.data_ptr(),
                                                    self.observation.data_ptr(), self.legal.data_ptr(),
                                                    self.n_legal.data_ptr()))
        return self.observation, self.legal, self.n_legal

    def step(self, action):
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mz_env_ttt_step(_stream_ptr(self.device), self.G, self.board.data_ptr(),
                                                 self.to_play.data_ptr(), self.steps.data_ptr(), self.ended.data_ptr(),
                                                 self._action_ptr(action), self.reward.data_ptr(),
                                                 self.status.data_ptr(), self._illegal.data_ptr()))
        return self.reward


class CartPoleVectorEnv(VectorEnvironment):
    """CartPole-v1 (environment.py:44-63 around gym; equations restated, see environment.CartPoleEnvironment)."""
    action_space_size = 2
    num_players = 1
    observation_size = 4
    max_moves = 500
    supports_fused = True

    def __init__(self, num_games: int, device='cuda', seed: Optional[int] = None) -> None:
        super().__init__(num_games, device)
        self.state = torch.zeros(self.G, 4, dtype=torch.float64, device=self.device)
        self.generator = torch.Generator(device=self.device)
        if seed is not None:
            self.generator.manual_seed(seed)

    def draw_restart_states(self, count: int = 1):
        return (torch.rand(count, self.G, 4, dtype=torch.float64, device=self.device, generator=self.generator) - 0.5) * 0.1

    def as_struct(self, restart_state=None):
        if restart_state is None:
            restart_state = self.draw_restart_states()[0]
        self._init_keepalive = restart_state
        return _lib.Env(_lib.MZ_ENV_CARTPOLE, self.G, self.state.data_ptr(), None, self.steps.data_ptr(),
                        self.ended.data_ptr(), restart_state.data_ptr())

    def reset(self, initial_state: Optional[torch.Tensor] = None) -> None:
        """initial_state [G,4] f64; default U(-0.05, 0.05) like gym's reset."""
        if initial_state is None:
            initial_state = (torch.rand(self.G, 4, dtype=torch.float64, device=self.device,
                                        generator=self.generator) - 0.5) * 0.1
        init = initial_state.to(device=self.device, dtype=torch.float64).contiguous()
        if tuple(init.shape) != (self.G, 4):
            raise MuZeroError(f'initial_state must have shape ({self.G}, 4)')
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mz_env_cartpole_reset(_stream_ptr(self.device), self.G, self.state.data_ptr(),
                                                       self.steps.data_ptr(), self.ended.data_ptr(), init.data_ptr()))
        self._init_keepalive = init

    def reset_where(self, where, initial_state: Optional[torch.Tensor] = None):
        if initial_state is None:
            initial_state = (torch.rand(self.G, 4, dtype=torch.float64, device=self.device,
                                        generator=self.generator) - 0.5) * 0.1
        init = initial_state.to(device=self.device, dtype=torch.float64).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mz_env_cartpole_reset_where(_stream_ptr(self.device), self.G, self.state.data_ptr(),
                                                             self.steps.data_ptr(), self.ended.data_ptr(),
                                                             init.data_ptr(), self._action_ptr(where)))
        self._init_keepalive = init

    def observe(self):
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mz_env_cartpole_observe(_stream_ptr(self.device), self.G, self.state.data_ptr(),
                                                         self.ended.data_ptr(), self.observation.data_ptr(),
                                                         self.legal.data_ptr(), self.n_legal.data_ptr()))
        return self.observation, self.legal, self.n_legal

    def step(self, action):
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mz_env_cartpole_step(_stream_ptr(self.device), self.G, self.state.data_ptr(),
                                                      self.steps.data_ptr(), self.ended.data_ptr(),
                                                      self._action_ptr(action), self.reward.data_ptr(),
                                                      self.status.data_ptr(), self._illegal.data_ptr()))
        return self.reward
