"""Lockstep search over T concurrent games: the batched form of MCTSAgent.make_move (agent.py:219-232).

One `LockstepSearch.run()` is one move of every game:

    initial_inference -> expand roots -> Dirichlet noise -> N x (select -> recurrent_inference -> expand ->
    backprop) -> root statistics -> action sampling

Every stage is a HIP launch on one stream; nothing returns to the host inside a move.  Hidden states live in
one [N+1][T][H] tensor: simulation s writes block s+1 of every tree (lockstep), and the inference kernel
gathers leaf.parent.hidden_state rows by the parent block that select reports, so there is no gather or
scatter copy (agent.py:204 is an operand load).

Games shard over GPUs as independent ranks (one process per GPU); the only exchange is `TrajectoryGather`,
an all-gather of each move's search statistics (root visit distribution, root value, chosen action) -- what the
reference's agents send to the replay buffer (agent.py:231, agent_services.py:54).
"""
import ctypes
from typing import Optional

import torch

from . import _lib
from .exceptions import MuZeroError
from .network import Network
from .tree_pool import TreePool, _check_tensor


def shard_bounds(total_games: int, rank: int, world: int):
    """Contiguous block partition of the games over the ranks: games are independent (fresh root and MinMaxStats
    per move, agent.py:220-221), so a shard needs nothing from the others during a search."""
    base, extra = divmod(total_games, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class DuckNetwork:
    """Any object with the reference's network interface -- initial_inference(observation) and
    recurrent_inference(hidden_state, action) returning an object with batch_hidden_state / batch_reward / batch_value /
    batch_policy_logits (+ the to_scalar functions), agent.py:141, network.py:149-174 -- behind the two *_pool methods
    the batched search calls.  The network is handed torch tensors on the search's device (observation [T, *shape] f32,
    hidden_state [T, ...], action [T] i64) and may return torch tensors, numpy arrays or objects with .numpy(); hidden
    states of any shape are kept in a [N+1, T, ...] tensor created at the first call.  The tree still runs on the GPU
    pool; only the kernels of muzero_amd.network.Network are bypassed."""

    def __init__(self, network, config, device):
        self.net, self.config, self.device = network, config, torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.ACTION_SPACE_SIZE = config.game_config.action_space_size
        self.observation_shape = getattr(network, 'OBSERVATION_SHAPE', None)
        self.pool_hidden = None

    def training_steps(self):
        return self.net.training_steps()

    def _tensor(self, x, dtype=torch.float32):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x.numpy() if hasattr(x, 'numpy') else x)
        return x.to(device=self.device, dtype=dtype)

    def _scalars(self, out, which):
        fused = getattr(out, f'batch_{which}_scalar', None)
        raw = fused if fused is not None else getattr(out, f'{which}_to_scalar')(getattr(out, f'batch_{which}'))
        return self._tensor(raw).reshape(-1)

    def initial_inference_pool(self, observation, num_blocks, value, logits):
        T = observation.shape[0]
        obs = observation if self.observation_shape is None else observation.reshape(T, *self.observation_shape)
        out = self.net.initial_inference(obs)
        h = self._tensor(out.batch_hidden_state)
        if self.pool_hidden is None or tuple(self.pool_hidden.shape[1:]) != tuple(h.shape) or self.pool_hidden.shape[0] != num_blocks:
            self.pool_hidden = torch.empty(num_blocks, *h.shape, dtype=torch.float32, device=self.device)
        self.pool_hidden[0].copy_(h)
        value.copy_(self._scalars(out, 'value'))
        logits.copy_(self._tensor(out.batch_policy_logits).reshape(T, self.ACTION_SPACE_SIZE))

    def recurrent_inference_pool(self, parent_block, action, out_block, reward, value, logits):
        T = parent_block.shape[0]
        rows = torch.arange(T, device=self.device)
        h = self.pool_hidden[parent_block.long().clamp(min=0), rows]          # leaf.parent.hidden_state (agent.py:204)
        out = self.net.recurrent_inference(h, action.long().clamp(min=0))
        self.pool_hidden[out_block].copy_(self._tensor(out.batch_hidden_state))
        reward.copy_(self._scalars(out, 'reward'))
        value.copy_(self._scalars(out, 'value'))
        logits.copy_(self._tensor(out.batch_policy_logits).reshape(T, self.ACTION_SPACE_SIZE))


class LockstepSearch:
    def __init__(self, config, network, num_trees: int, device=None, out=None, lanes_per_tree: int = 0):
        """`out`: optional (visits, root_value, minmax, policy, action) tensors to write the results to (views of a
        larger batch, see PipelinedSearch); allocated here when absent.  `lanes_per_tree`: TreePool's (0 = its choice)."""
        self.config = config
        self.network = network
        g, m = config.game_config, config.mcts_config
        kb = config.value_config.known_bounds
        self.native = isinstance(network, Network)
        if not self.native:              # a duck-typed network (agent.py:141): the tree on the GPU, the network its own
            if device is None and not hasattr(network, 'device'):
                device = torch.device('cuda', torch.cuda.current_device())
            self.duck = DuckNetwork(network, config, device if device is not None else network.device)
        self.device = torch.device(device) if device is not None else network.device
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.T, self.A, self.N = int(num_trees), g.action_space_size, m.num_simulations
        if self.native and network.ACTION_SPACE_SIZE != self.A:
            raise MuZeroError(f'network has {network.ACTION_SPACE_SIZE} actions, the game {self.A}')
        self.pool = TreePool(num_trees=self.T, num_actions=self.A, max_simulations=self.N,
                             num_players=g.num_players, discount=g.discount, default_value=m.default_value,
                             known_bounds=None if kb is None else kb.endpoints(), device=self.device,
                             lanes_per_tree=lanes_per_tree)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.hidden = torch.empty(self.N + 1, self.T, network.hidden_size, **f32) if self.native else None
        self.root_value_net = torch.empty(self.T, **f32)        # initial-inference value (unused by the search)
        self.logits = torch.empty(self.T, self.A, **f32)
        self.reward = torch.empty(self.T, **f32)
        self.value = torch.empty(self.T, **f32)
        self.frac = m.root_exploration_fraction
        i32 = dict(dtype=torch.int32, device=self.device)
        if out is None:
            out = (torch.empty(self.T, self.A, **i32), torch.empty(self.T, dtype=torch.float64, device=self.device),
                   torch.empty(self.T, 2, dtype=torch.float64, device=self.device),
                   torch.empty(self.T, self.A, **f32), torch.empty(self.T, **i32))
        self.out_visits, self.out_root_value, self.out_minmax, self.out_policy, self.out_action = out
        _check_tensor('out visits', self.out_visits, torch.int32, (self.T, self.A), self.device)
        _check_tensor('out root_value', self.out_root_value, torch.float64, (self.T,), self.device)
        _check_tensor('out minmax', self.out_minmax, torch.float64, (self.T, 2), self.device)
        _check_tensor('out policy', self.out_policy, torch.float32, (self.T, self.A), self.device)
        _check_tensor('out action', self.out_action, torch.int32, (self.T,), self.device)

    def run_native(self, observation, noise, u, temperature, legal=None, n_legal=None, events=None, trace=None):
        """The whole move enqueued by one C call (mz_search_move): same launches as run(), no interpreter time
        between them.  `events`: optional ctypes array of 4*N hipEvent_t (measurement).  `trace`: optional dict that
        receives the per-simulation network outputs and selections (reward/value [N,T], logits [N,T,A], parent/action
        [N,T]; mz_move_io_t.trace_*).  Results are written to the search's own out_* tensors, which are returned."""
        if not self.native:
            raise MuZeroError('run_native needs a HIP-backed muzero_amd.network.Network')
        pool, net, dev = self.pool, self.network, self.device
        pool._sync_stream()
        net._sync_stream()
        obs = _check_tensor('observation', observation, torch.float32, (self.T, net.observation_size), dev)
        io = _lib.MoveIO(
            num_trees=self.T, num_simulations=self.N, hidden_size=net.hidden_size, stream=pool._stream,
            observation=obs,
            legal=None if legal is None else _check_tensor('legal', legal, torch.int32, (self.T, self.A), dev),
            n_legal=None if n_legal is None else _check_tensor('n_legal', n_legal, torch.int32, (self.T,), dev),
            noise=_check_tensor('noise', noise, torch.float64, (self.T, self.A), dev), noise_frac=self.frac,
            u=_check_tensor('u', u, torch.float64, (self.T,), dev),
            temperature=_check_tensor('temperature', temperature, torch.float64, (self.T,), dev),
            hidden=self.hidden.data_ptr(), logits=self.logits.data_ptr(), reward=self.reward.data_ptr(),
            value=self.value.data_ptr(), parent_block=pool._parent.data_ptr(), action=pool._action.data_ptr(),
            out_visits=self.out_visits.data_ptr(), out_root_value=self.out_root_value.data_ptr(),
            out_minmax=self.out_minmax.data_ptr(), out_policy=self.out_policy.data_ptr(),
            out_action=self.out_action.data_ptr(),
            events=None if events is None else ctypes.cast(events, ctypes.c_void_p))
        if trace is not None:
            f32 = dict(dtype=torch.float32, device=dev)
            i32 = dict(dtype=torch.int32, device=dev)
            trace.update(reward=torch.zeros(self.N, self.T, **f32), value=torch.zeros(self.N, self.T, **f32),
                         logits=torch.zeros(self.N, self.T, self.A, **f32), parent=torch.zeros(self.N, self.T, **i32),
                         action=torch.zeros(self.N, self.T, **i32))
            io.trace_reward, io.trace_value, io.trace_logits = (trace[k].data_ptr() for k in ('reward', 'value', 'logits'))
            io.trace_parent, io.trace_action = trace['parent'].data_ptr(), trace['action'].data_ptr()
        _lib.check(_lib.load().mz_search_move(pool._h, net._h, ctypes.byref(io)))
        return self.out_visits, self.out_root_value, self.out_minmax, self.out_policy, self.out_action

    def run(self, observation, noise, u, temperature, legal=None, n_legal=None, hooks=None):
        """One move for all T games.  observation [T, obs] f32; noise [T,A] f64 (Dirichlet draws, injected like
        np.random.dirichlet in agent.py:172); u [T] f64 uniforms; temperature [T] f64.
        Returns (visits [T,A] i32, root_value [T] f64, minmax [T,2] f64, policy [T,A] f32, action [T] i32)."""
        if hooks is None and self.native:
            return self.run_native(observation, noise, u, temperature, legal, n_legal)
        pool, net, N = self.pool, self.network, self.N
        pool.reset()
        if not self.native:
            return self._run_duck(observation, noise, u, temperature, legal, n_legal, hooks)
        net.initial_inference_pool(observation, self.hidden, self.root_value_net, self.logits)
        pool.expand_roots(self.logits, legal, n_legal, logits=True)
        pool.add_root_noise(noise, self.frac)
        parent, action = pool.select()
        for s in range(N):
            if hooks is not None:
                hooks.before_inference(s)
            net.recurrent_inference_pool(self.hidden, parent, action, s + 1, self.reward, self.value, self.logits)
            if hooks is not None:
                hooks.after_inference(s)
            if s + 1 < N:
                parent, action = pool.step(self.logits, self.reward, self.value, logits=True)
            else:
                pool.expand_backprop(self.logits, self.reward, self.value, logits=True)
            if hooks is not None:
                hooks.after_tree(s)
        visits, root_value, minmax, policy = pool.root_stats()
        chosen = pool.select_action(u, temperature)
        return visits, root_value, minmax, policy, chosen


    def _run_duck(self, observation, noise, u, temperature, legal, n_legal, hooks):
        pool, net, N = self.pool, self.duck, self.N
        net.initial_inference_pool(observation, N + 1, self.root_value_net, self.logits)
        pool.expand_roots(self.logits, legal, n_legal, logits=True)
        pool.add_root_noise(noise, self.frac)
        parent, action = pool.select()
        for s in range(N):
            if hooks is not None:
                hooks.before_inference(s)
            net.recurrent_inference_pool(parent, action, s + 1, self.reward, self.value, self.logits)
            if hooks is not None:
                hooks.after_inference(s)
            if s + 1 < N:
                parent, action = pool.step(self.logits, self.reward, self.value, logits=True)
            else:
                pool.expand_backprop(self.logits, self.reward, self.value, logits=True)
            if hooks is not None:
                hooks.after_tree(s)
        visits, root_value, minmax, policy = pool.root_stats()
        return visits, root_value, minmax, policy, pool.select_action(u, temperature)


class PipelinedSearch:
    """T games searched as S independent lockstep lanes of T/S games, each lane on its own HIP stream.

    Inside one lane a simulation is a strict chain (inference -> tree step -> inference ...), and the tree step and the
    inference's head kernel are short latency-bound launches that leave the matrix cores idle.  With S > 1 lanes in
    flight the hardware runs one lane's tree/head kernels underneath another lane's matrix kernel.  Lanes share
    nothing but the weights (each has its own handle, scratch and tree pool), so the results are those of
    LockstepSearch on the same games, bit for bit.  Same run()/run_native() signature."""

    def __init__(self, config, network, num_trees: int, lanes: int = 2, device=None):
        if not isinstance(network, Network):
            raise MuZeroError('PipelinedSearch needs a HIP-backed muzero_amd.network.Network')
        self.device = torch.device(device) if device is not None else network.device
        self.T, self.A = int(num_trees), config.game_config.action_space_size
        lanes = max(1, min(int(lanes), self.T))
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.out_visits = torch.empty(self.T, self.A, **i32)
        self.out_root_value = torch.empty(self.T, dtype=torch.float64, device=self.device)
        self.out_minmax = torch.empty(self.T, 2, dtype=torch.float64, device=self.device)
        self.out_policy = torch.empty(self.T, self.A, **f32)
        self.out_action = torch.empty(self.T, **i32)
        self.bounds = [shard_bounds(self.T, i, lanes) for i in range(lanes)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(lanes)]
        self.networks = [network] + [type(network)(config, weights=network.get_weights(),
                                                   training_steps=network.training_steps(), device=self.device)
                                     for _ in range(lanes - 1)]
        self.lanes = []
        for (lo, hi), net in zip(self.bounds, self.networks):
            out = (self.out_visits[lo:hi], self.out_root_value[lo:hi], self.out_minmax[lo:hi],
                   self.out_policy[lo:hi], self.out_action[lo:hi])
            self.lanes.append(LockstepSearch(config, net, hi - lo, device=self.device, out=out))
        self.pool = self.lanes[0].pool

    def set_weights(self, flat, training_steps=None):
        for net in self.networks:
            net.set_weights(flat, training_steps)

    def counters(self):
        """(simulations, summed path lengths) over all lanes"""
        per_lane = [lane.pool.counters() for lane in self.lanes]
        return tuple(sum(c[i] for c in per_lane) for i in range(2))

    def run_native(self, observation, noise, u, temperature, legal=None, n_legal=None, events=None):
        """`events` (measurement) are recorded on lane 0's stream."""
        cur = torch.cuda.current_stream(self.device)
        for i, ((lo, hi), lane, stream) in enumerate(zip(self.bounds, self.lanes, self.streams)):
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                lane.run_native(observation[lo:hi], noise[lo:hi], u[lo:hi], temperature[lo:hi],
                                None if legal is None else legal[lo:hi],
                                None if n_legal is None else n_legal[lo:hi], events=events if i == 0 else None)
        for stream in self.streams:
            cur.wait_stream(stream)
        return self.out_visits, self.out_root_value, self.out_minmax, self.out_policy, self.out_action

    run = run_native


class BatchedHistory:
    """GameHistory (game.py:8-41) of G lockstep games as padded device tensors, M = move capacity:

        observations [M+1, G, obs] f32 | to_plays [M+1, G] i32 | actions [M, G] i32 | rewards [M, G] f32 |
        root_values [M, G] f64 | policies [M, G, A] f32 | lengths [G] i32

    Entries at move index >= lengths[g] are padding.  This is also the block the ranks exchange when games finish
    (`all_gather`): fixed-size, so one all-gather per field and no size negotiation."""
    FIELDS = ('observations', 'to_plays', 'actions', 'rewards', 'root_values', 'policies', 'lengths')

    def __init__(self, max_moves: int, num_games: int, observation_shape, num_actions: int, device) -> None:
        self.M, self.G, self.A = int(max_moves), int(num_games), int(num_actions)
        self.observation_shape = tuple(observation_shape)
        obs = 1
        for d in self.observation_shape:
            obs *= int(d)
        z = lambda shape, dtype: torch.zeros(*shape, dtype=dtype, device=device)
        self.observations = z((self.M + 1, self.G, obs), torch.float32)
        self.to_plays = z((self.M + 1, self.G), torch.int32)
        self.actions = z((self.M, self.G), torch.int32)
        self.rewards = z((self.M, self.G), torch.float32)
        self.root_values = z((self.M, self.G), torch.float64)
        self.policies = z((self.M, self.G, self.A), torch.float32)
        self.lengths = z((self.G,), torch.int32)
        self.metadata = {}

    def as_struct(self):
        """mz_history_t over this history's tensors (keep `self` alive while the struct is in use)."""
        obs = self.observations.shape[2]
        return _lib.History(self.G, self.M, obs, self.A, self.observations.data_ptr(), self.to_plays.data_ptr(),
                            self.actions.data_ptr(), self.rewards.data_ptr(), self.root_values.data_ptr(),
                            self.policies.data_ptr(), self.lengths.data_ptr())

    def to_game_histories(self):
        """-> list of G `game.GameHistory` with the reference's fields (host copies; synchronises)."""
        from .game import GameHistory
        host = {f: getattr(self, f).cpu().numpy() for f in self.FIELDS}
        games = []
        for g in range(self.G):
            n = int(host['lengths'][g])
            h = GameHistory()
            h.observations = [host['observations'][m, g].reshape(self.observation_shape) for m in range(n + 1)]
            h.to_plays = [int(x) for x in host['to_plays'][:n + 1, g]]
            h.actions = [int(x) for x in host['actions'][:n, g]]
            h.rewards = [float(x) for x in host['rewards'][:n, g]]
            h.root_values = [float(x) for x in host['root_values'][:n, g]]
            h.policies = [host['policies'][m, g].copy() for m in range(n)]
            h.metadata.update(self.metadata)
            games.append(h)
        return games

    def save_games(self, filepath: str) -> None:
        """Writes the games as the reference's serialised GameHistoryList (game_services.py:36-42), the file its
        ReplayBuffer.load_games reads (interchange.py)."""
        from . import interchange
        interchange.save_games(self.to_game_histories(), filepath)

    def all_gather(self, group=None):
        """Every rank's finished games, concatenated along the game axis in rank order (RCCL over xGMI on GPUs,
        gloo in the CPU tests).  All ranks must hold the same M and G."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        out = BatchedHistory(self.M, self.G * world, self.observation_shape, self.A, self.lengths.device)
        out.metadata = dict(self.metadata)
        for f in self.FIELDS:
            mine = getattr(self, f).contiguous()
            if world == 1:
                getattr(out, f).copy_(mine)
                continue
            parts = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(parts.view(-1), mine.view(-1), group=group)
            if f == 'lengths':
                out.lengths.copy_(parts.reshape(-1))
            else:       # [world, M(+1), G, ...] -> [M(+1), world*G, ...]
                moved = parts.movedim(0, 1)
                getattr(out, f).copy_(moved.reshape(moved.shape[0], world * self.G, *moved.shape[3:]))
        return out


class SelfPlayEngine:
    """Agent.play_game (agent.py:31-44) for G games at once: every move is one `observe` launch, one lockstep search
    (`mz_search_move`) and one `step` launch; observations, search statistics and rewards are recorded into a
    BatchedHistory on the device.  Randomness (Dirichlet noise, action sampling) is drawn on the device from a
    torch generator -- the single-game MCTSAgent is the path that reproduces the reference's host RNG stream."""

    def __init__(self, config, network, num_games: int, vector_env=None, device=None, lanes: int = 1,
                 seed: Optional[int] = None, models_path: Optional[str] = None):
        """models_path: directory of published weight versions (Network.save_version); the newest one is hot-loaded
        before every batch of games, like an agent picking up TF-Serving's latest model version."""
        from . import vector_env as venv
        self.config, self.network, self.models_path = config, network, models_path
        self.device = torch.device(device) if device is not None else network.device
        self.G = int(num_games)
        if vector_env is None:
            classes = {'TicTacToe': venv.TicTacToeVectorEnv, 'CartPole': venv.CartPoleVectorEnv}
            name = config.game_config.name
            if name not in classes:
                raise MuZeroError(f'no batched environment for game {name!r}: pass vector_env=')
            vector_env = classes[name](self.G, device=self.device)
        self.env = vector_env
        if self.env.G != self.G or self.env.action_space_size != config.game_config.action_space_size:
            raise MuZeroError('vector_env does not match num_games / the action space of the config')
        if lanes > 1:
            self.search = PipelinedSearch(config, network, self.G, lanes=lanes, device=self.device)
        else:
            self.search = LockstepSearch(config, network, self.G, device=self.device)
        self.A = config.game_config.action_space_size
        self.generator = torch.Generator(device=self.device)
        if seed is not None:
            self.generator.manual_seed(seed)
        self.alpha = torch.full((self.G, self.A), float(config.mcts_config.root_dirichlet_alpha), dtype=torch.float64,
                                device=self.device)
        self.slot = torch.arange(self.A, device=self.device, dtype=torch.int32).unsqueeze(0)
        self.temperature = torch.empty(self.G, dtype=torch.float64, device=self.device)
        self._noise = torch.empty(self.G, self.A, dtype=torch.float64, device=self.device)

    def _dirichlet(self, n_legal, gamma=None):
        """noise[g, :n_legal[g]] ~ Dirichlet(alpha) (agent.py:172), zeros elsewhere: normalised Gamma(alpha, 1) draws
        (`gamma`: draws made ahead of time, else drawn here)."""
        if gamma is None:
            gamma = torch._standard_gamma(self.alpha, generator=self.generator)
        _lib.check(_lib.load().mz_dirichlet_from_gamma(
            ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), self.G, self.A, gamma.data_ptr(),
            n_legal.data_ptr(), self._noise.data_ptr()))
        return self._noise

    def play_games(self, max_moves: Optional[int] = None, check_every: int = 1) -> BatchedHistory:
        env, cfg = self.env, self.config
        M = int(max_moves if max_moves is not None else cfg.mcts_config.max_moves)
        hist = BatchedHistory(M, self.G, self.network.OBSERVATION_SHAPE, self.A, self.device)
        if self.models_path and self.network.load_latest(self.models_path) and isinstance(self.search, PipelinedSearch):
            for replica in self.search.networks[1:]:
                replica.set_weights(self.network.get_weights(), self.network.training_steps())
        env.reset()
        steps = self.network.training_steps()
        played = 0
        for m in range(M):
            obs, legal, n_legal = env.observe()
            hist.observations[m].copy_(obs)
            hist.to_plays[m].copy_(env.to_play)
            noise = self._dirichlet(n_legal)
            u = torch.rand(self.G, dtype=torch.float64, device=self.device, generator=self.generator)
            t = float(cfg.visit_softmax_temperature_fn(num_moves=m, training_steps=steps))
            self.temperature.fill_(t)
            for lane in getattr(self.search, 'lanes', [self.search]):
                lane.pool.set_temperature(t)         # exact count ** (1/t) sampling weights (agent.py:167)
            _visits, root_value, _minmax, policy, action = self.search.run(obs, noise, u, self.temperature, legal,
                                                                           n_legal)
            hist.actions[m].copy_(action)
            hist.root_values[m].copy_(root_value)
            hist.policies[m].copy_(policy)
            hist.rewards[m].copy_(env.step(action))
            played = m + 1
            if played % check_every == 0 and env.all_ended():
                break
        obs, _, _ = env.observe()
        hist.observations[played].copy_(obs)
        hist.to_plays[played].copy_(env.to_play)
        hist.lengths.copy_(env.steps)
        self._check_devices()
        hist.metadata = {'network_id': str(steps), 'moves_played': played}
        return hist

    def _check_devices(self):
        """The device-side validation flags, read once per batch: an illegal action handed to the environment
        (games/tictactoe.py:122-123 raises) or a malformed legal-action list handed to the search."""
        self.env.check_illegal()
        for lane in getattr(self.search, 'lanes', [self.search]):
            lane.pool.counters()                     # raises MuZeroHipError when the pool's err_flag is set
            if isinstance(lane.network, Network):
                lane.network.check_range()           # an activation outside the kernels' numeric domain

    def play_continuous(self, num_games: int, max_moves: Optional[int] = None, check_every: int = 32,
                        fused: bool = True) -> BatchedHistory:
        """The reference agent's outer loop (agent_services.py:47-54: play a game, hand its history over, start the next
        one at once) for G slots: a slot whose game ends is flushed to the finished store and restarted before the next
        search, so every move searches a full batch of live games.  Returns the first `num_games` games started (in
        start order: slot order within a wave of restarts) as one BatchedHistory; games still in flight at the end are
        dropped.  Slots are at different moves of their games, so the temperature is per slot (config.py:153-157).
        check_every: the longest run of moves between two looks at the finished-games counter (a look synchronises; the
        fused loop spaces its looks by the rate at which games have been finishing and looks every move near the end).
        fused=False takes the one-call-per-step entry points instead of the two fused ones (same games for the same random
        draws; the draws are made in a different order, so a seed gives different games)."""
        env, cfg, dev, G = self.env, self.config, self.device, self.G
        lib = _lib.load()
        M = int(max_moves if max_moves is not None else cfg.mcts_config.max_moves)
        live = BatchedHistory(M, G, self.network.OBSERVATION_SHAPE, self.A, dev)
        done = BatchedHistory(M, int(num_games), self.network.OBSERVATION_SHAPE, self.A, dev)
        live_s, done_s = live.as_struct(), done.as_struct()
        if self.models_path and self.network.load_latest(self.models_path) and isinstance(self.search, PipelinedSearch):
            for replica in self.search.networks[1:]:
                replica.set_weights(self.network.get_weights(), self.network.training_steps())
        steps_trained = self.network.training_steps()
        i32 = dict(dtype=torch.int32, device=dev)
        serial = torch.arange(G, **i32)
        counters = torch.tensor([G, 0], dtype=torch.int64, device=dev)       # next serial, games stored
        flush_dst, reset_mask = torch.empty(G, **i32), torch.empty(G, **i32)
        t_cfg = float(cfg.mcts_config.temperature)
        for lane in getattr(self.search, 'lanes', [self.search]):
            lane.pool.set_temperature(t_cfg)
        freezing = int(cfg.mcts_config.freezing_moves)
        t_on = torch.full((G,), t_cfg, dtype=torch.float64, device=dev)
        t_off = torch.zeros(G, dtype=torch.float64, device=dev)
        stream = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        env.reset()
        moves = 0
        fused = bool(fused and getattr(env, 'supports_fused', False))
        if fused:
            # Three small launches a move instead of fourteen: the bookkeeping between two searches is one call
            # (mz_selfplay_turnover: two kernels), the environment step and the move record another.
            # The random draws (root-noise gammas, the uniforms of the action sampling, CartPole restart states) are made
            # for `block` moves in one launch each.
            block, next_check = 16, 0
            alpha = self.alpha.unsqueeze(0).expand(block, G, self.A).contiguous()
            alpha_is_one = float(cfg.mcts_config.root_dirichlet_alpha) == 1.0
            while True:
                k = moves % block
                if k == 0:
                    if alpha_is_one:         # Gamma(1, 1) is the standard exponential (numpy's own path for shape 1)
                        gammas = torch.empty_like(alpha).exponential_(generator=self.generator)
                    else:
                        gammas = torch._standard_gamma(alpha, generator=self.generator)
                    us = torch.rand(block, G, dtype=torch.float64, device=dev, generator=self.generator)
                    restarts = env.draw_restart_states(block)
                gamma, u = gammas[k], us[k]
                env_s = env.as_struct(None if restarts is None else restarts[k])
                _lib.check(lib.mz_selfplay_turnover(
                    stream(), ctypes.byref(env_s), ctypes.byref(live_s), ctypes.byref(done_s), serial.data_ptr(),
                    counters.data_ptr(), flush_dst.data_ptr(), reset_mask.data_ptr(), gamma.data_ptr(), freezing, t_cfg,
                    env.observation.data_ptr(), env.legal.data_ptr(), env.n_legal.data_ptr(), self._noise.data_ptr(),
                    self.temperature.data_ptr()))
                if moves >= next_check:
                    stored = int(counters[1].item())                                        # synchronises
                    if stored >= num_games:
                        break
                    # the next look: three quarters of the way to the expected end at the rate seen so far (a
                    # synchronisation drains the launch queue, ~0.2 ms), every move once the end is near
                    ahead = check_every
                    if stored > 0 and moves > 0:
                        ahead = min(check_every, int(0.75 * (num_games - stored) * moves / stored))
                    next_check = moves + max(1, ahead)
                _visits, root_value, _minmax, policy, action = self.search.run(env.observation, self._noise, u,
                                                                               self.temperature, env.legal, env.n_legal)
                _lib.check(lib.mz_selfplay_step_record(
                    stream(), ctypes.byref(env_s), ctypes.byref(live_s), action.data_ptr(), root_value.data_ptr(),
                    policy.data_ptr(), env.reward.data_ptr(), env.status.data_ptr(), env._illegal.data_ptr()))
                moves += 1
        while not fused:
            obs, legal, n_legal = env.observe()
            _lib.check(lib.mz_history_record_observation(stream(), ctypes.byref(live_s), env.steps.data_ptr(),
                                                         obs.data_ptr(), env.to_play.data_ptr()))
            # finished games -> the store, their slots restarted (first iteration: nothing is over yet)
            _lib.check(lib.mz_history_flush(stream(), ctypes.byref(live_s), ctypes.byref(done_s), env.steps.data_ptr(),
                                            env.ended.data_ptr(), serial.data_ptr(), counters.data_ptr(),
                                            flush_dst.data_ptr(), reset_mask.data_ptr()))
            if moves % check_every == 0 and int(counters[1].item()) >= num_games:      # synchronises
                break
            env.reset_where(reset_mask)
            obs, legal, n_legal = env.observe()
            _lib.check(lib.mz_history_record_observation(stream(), ctypes.byref(live_s), env.steps.data_ptr(),
                                                         obs.data_ptr(), env.to_play.data_ptr()))
            noise = self._dirichlet(n_legal)
            u = torch.rand(G, dtype=torch.float64, device=dev, generator=self.generator)
            torch.where(env.steps < freezing, t_on, t_off, out=self.temperature)
            _visits, root_value, _minmax, policy, action = self.search.run(obs, noise, u, self.temperature, legal,
                                                                           n_legal)
            reward = env.step(action)
            _lib.check(lib.mz_history_record_move(stream(), ctypes.byref(live_s), env.steps.data_ptr(),
                                                  env.status.data_ptr(), action.data_ptr(), reward.data_ptr(),
                                                  root_value.data_ptr(), policy.data_ptr()))
            moves += 1
        self._check_devices()
        done.metadata = {'network_id': str(steps_trained), 'moves_played': moves, 'slots': G,
                         'games_started': int(counters[0].item())}
        return done


class TrajectoryGather:
    """All-gather of per-move search statistics over the ranks (RCCL over xGMI on GPUs, gloo in CPU tests).

    Record per game: policy[A] f32 | root_value f32 | action f32 -> [T, A+2] f32 per rank, gathered into
    [world, T, A+2].  The payload is KBs per move, so the call is latency-bound; it is issued asynchronously and
    waited for one move later."""

    def __init__(self, num_trees: int, num_actions: int, device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.T, self.A = num_trees, num_actions
        self.send = torch.empty(num_trees, num_actions + 2, dtype=torch.float32, device=device)
        self.recv = torch.empty(self.world, num_trees, num_actions + 2, dtype=torch.float32, device=device)
        self.work = None

    def start(self, policy, root_value, action):
        self.wait()
        self.send[:, :self.A] = policy
        self.send[:, self.A] = root_value.to(torch.float32)
        self.send[:, self.A + 1] = action.to(torch.float32)
        if self.world == 1:
            self.recv[0].copy_(self.send)
            return
        self.work = self.dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1), group=self.group,
                                                     async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.recv
