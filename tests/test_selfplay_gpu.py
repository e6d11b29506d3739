"""End-to-end self-play path on the GPU: lockstep search with the HIP network in the loop, and the drop-in
single-game MCTSAgent replaying a full game of the reference agent (tests/golden/ttt_game_*.npz)."""
import random

import numpy as np
import pytest

from conftest import GAME_FIXTURES, load_golden

pytestmark = pytest.mark.gpu


class Recorder:
    """Copies the per-simulation network outputs and the select results out of a LockstepSearch run."""

    def __init__(self, search):
        self.s = search
        self.parent, self.action, self.reward, self.value, self.logits = [], [], [], [], []

    def before_inference(self, s):
        self.parent.append(self.s.pool._parent.clone())
        self.action.append(self.s.pool._action.clone())

    def after_inference(self, s):
        self.reward.append(self.s.reward.clone())
        self.value.append(self.s.value.clone())
        self.logits.append(self.s.logits.clone())

    def after_tree(self, s):
        pass


@pytest.mark.parametrize('game', ['tictactoe', 'cartpole', 'atari'])
def test_lockstep_search_with_network_in_the_loop(game):
    import torch
    from muzero_amd.games import atari_shaped, cartpole, tictactoe
    from muzero_amd.selfplay import LockstepSearch
    from oracle import c_oracle, net_oracle
    mod = {'tictactoe': tictactoe, 'cartpole': cartpole, 'atari': atari_shaped}[game]
    N, T = 12, 70
    config = mod.make_config(num_simulations=N)
    flat = net_oracle.random_weights(game, 21)
    net = config.network_config.network_class(config, weights=flat)
    ref = net_oracle.make_oracle(game, flat)
    A, H = ref.A, ref.H
    rng = np.random.default_rng(5)
    if game == 'tictactoe':
        obs = rng.integers(0, 2, size=(T, 27)).astype(np.float32)
    elif game == 'atari':
        obs = rng.standard_normal((T, 256)).astype(np.float32)
    else:
        obs = (rng.uniform(-1, 1, size=(T, 4)) * [2.4, 3, 0.2, 3]).astype(np.float32)
    n_legal = rng.integers(0, A + 1, size=T).astype(np.int32)            # ragged, some trees inactive
    n_legal[0] = A
    legal = np.zeros((T, A), np.int32)
    for t in range(T):
        legal[t, :n_legal[t]] = np.sort(rng.permutation(A)[:n_legal[t]])
    noise = np.zeros((T, A))
    for t in range(T):
        if n_legal[t]:
            noise[t, :n_legal[t]] = rng.dirichlet([1.0] * int(n_legal[t]))
    u = rng.random(T)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    search = LockstepSearch(config, net, num_trees=T)
    rec = Recorder(search)
    visits, root_value, minmax, policy, chosen = search.run(dev(obs), dev(noise), dev(u), dev(np.ones(T)), dev(legal),
                                                            dev(n_legal), hooks=rec)
    live = n_legal > 0
    assert (visits.cpu().numpy()[live].sum(1) == N).all() and (visits.cpu().numpy()[~live] == 0).all()
    # (1) wiring: every simulation's hidden state / outputs are the oracle network applied to the parent's row
    hidden = search.hidden.cpu().numpy()
    vscale = 100.0 if game == 'cartpole' else 1.0
    h0, _, l0, _ = ref.initial_inference(obs)
    np.testing.assert_allclose(hidden[0], h0, rtol=1e-5, atol=1e-5)
    for s in range(N):
        pb, act = rec.parent[s].cpu().numpy(), rec.action[s].cpu().numpy()
        assert (pb[live] >= 0).all() and (pb[live] <= s).all() and (pb[~live] == -1).all()
        h2, r2, v2, l2, _ = ref.recurrent_inference(hidden[pb[live], np.nonzero(live)[0]], act[live])
        np.testing.assert_allclose(hidden[s + 1][live], h2, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(rec.reward[s].cpu().numpy()[live], r2, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(rec.value[s].cpu().numpy()[live], v2, rtol=1e-5, atol=1e-5 * vscale)
        np.testing.assert_allclose(rec.logits[s].cpu().numpy()[live], l2, rtol=1e-5, atol=1e-5)
    # (3) tree statistics: replay the device's own outputs (and the priors it derived from the logits) through the
    # C oracle -> identical selections, visit counts, root values and chosen actions
    kb = config.value_config.known_bounds
    opool = c_oracle.OraclePool(T, A, N, config.game_config.num_players, config.game_config.discount,
                                config.mcts_config.default_value, None if kb is None else kb.endpoints())
    priors = np.zeros((N + 1, T, A), np.float32)
    for t in np.nonzero(live)[0]:
        for e in range(N + 1):
            p = search.pool.read_block(int(t), e)['prior']
            priors[e, t, :len(p)] = p if e else 0
    # the root's fp32 priors as the device derived them from the logits, before the noise mix (its expf is <= 2 ulp from
    # NumPy's exp, trap T4, so the oracle is fed the device's own priors: the comparison is then exact)
    root = np.zeros((T, A), np.float32)
    for t in np.nonzero(live)[0]:
        p = search.pool.read_root_priors(int(t))
        assert len(p) == n_legal[t]
        root[t, :len(p)] = p
    # ... which are masked_policy of the oracle's root logits to within the exp difference
    for t in np.nonzero(live)[0]:
        acts = legal[t, :n_legal[t]]
        m = np.zeros(A, np.float32)
        m[acts] = np.exp(l0.astype(np.float32)[t, acts])
        nl = int(n_legal[t])              # child k <- policy[k], positionally (trap T5)
        np.testing.assert_allclose(root[t, :nl], (m / m.sum())[:nl], rtol=2e-5, atol=1e-7)
    opool.expand_roots(root, legal, n_legal)
    opool.add_root_noise(noise, config.mcts_config.root_exploration_fraction)
    for s in range(N):
        p, a = opool.select()
        np.testing.assert_array_equal(p, rec.parent[s].cpu().numpy(), err_msg=f'parent block, simulation {s}')
        np.testing.assert_array_equal(a, rec.action[s].cpu().numpy(), err_msg=f'action, simulation {s}')
        opool.expand_backprop(priors[s + 1], rec.reward[s].cpu().numpy(), rec.value[s].cpu().numpy())
    v2, rv2, mm2, pol2 = opool.root_stats()
    np.testing.assert_array_equal(visits.cpu().numpy(), v2)
    np.testing.assert_array_equal(root_value.cpu().numpy()[live], rv2[live])
    np.testing.assert_array_equal(policy.cpu().numpy(), pol2)
    np.testing.assert_array_equal(chosen.cpu().numpy(), opool.select_action(u, 1.0))


def test_lockstep_search_with_a_duck_typed_network():
    """LockstepSearch with any object that has the reference's network interface (agent.py:141; here the numpy oracle
    behind initial_inference / recurrent_inference): the tree runs on the GPU pool, the network is called once per
    simulation with the gathered parent hidden states.  The outputs the tree consumed are the network's own, and the
    search replays exactly through the C oracle."""
    import torch
    from muzero_amd.games import tictactoe
    from muzero_amd.network import BatchNetworkOutput
    from muzero_amd.selfplay import LockstepSearch
    from oracle import c_oracle, net_oracle
    N, T, A = 10, 50, 9
    config = tictactoe.make_config(num_simulations=N)
    ref = net_oracle.make_oracle('tictactoe', net_oracle.random_weights('tictactoe', 4))
    ident = lambda x: x

    class OracleNetwork:                       # knows nothing of muzero_amd.network.Network
        calls = 0

        def training_steps(self):
            return 0

        def initial_inference(self, observation):
            h, v, logits, _ = ref.initial_inference(observation.cpu().numpy().reshape(-1, 27))
            return BatchNetworkOutput(h.astype(np.float32), None, v.astype(np.float32), logits.astype(np.float32), ident, ident)

        def recurrent_inference(self, hidden_state, action):
            OracleNetwork.calls += 1
            h, r, v, logits, _ = ref.recurrent_inference(hidden_state.cpu().numpy().astype(np.float64), action.cpu().numpy())
            return BatchNetworkOutput(h.astype(np.float32), r.astype(np.float32), v.astype(np.float32),
                                      logits.astype(np.float32), ident, ident)

    rng = np.random.default_rng(8)
    obs = rng.integers(0, 2, size=(T, 27)).astype(np.float32)
    n_legal = rng.integers(0, A + 1, size=T).astype(np.int32)
    n_legal[0] = A
    legal = np.zeros((T, A), np.int32)
    noise = np.zeros((T, A))
    for t in range(T):
        legal[t, :n_legal[t]] = np.sort(rng.permutation(A)[:n_legal[t]])
        if n_legal[t]:
            noise[t, :n_legal[t]] = rng.dirichlet([1.0] * int(n_legal[t]))
    u = rng.random(T)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    search = LockstepSearch(config, OracleNetwork(), num_trees=T, device='cuda')
    rec = Recorder(search)
    visits, root_value, minmax, policy, chosen = search.run(dev(obs), dev(noise), dev(u), dev(np.ones(T)), dev(legal),
                                                            dev(n_legal), hooks=rec)
    assert OracleNetwork.calls == N
    live = n_legal > 0
    rows = np.nonzero(live)[0]
    hidden = search.duck.pool_hidden.cpu().numpy()
    for s in range(N):                         # the network saw leaf.parent.hidden_state and its outputs reached the tree
        pb, act = rec.parent[s].cpu().numpy(), rec.action[s].cpu().numpy()
        h2, r2, v2, l2, _ = ref.recurrent_inference(hidden[pb[live], rows].astype(np.float64), act[live])
        np.testing.assert_array_equal(hidden[s + 1][live], h2.astype(np.float32))
        np.testing.assert_array_equal(rec.reward[s].cpu().numpy()[live], r2.astype(np.float32))
        np.testing.assert_array_equal(rec.logits[s].cpu().numpy()[live], l2.astype(np.float32))
    kb = config.value_config.known_bounds
    opool = c_oracle.OraclePool(T, A, N, config.game_config.num_players, config.game_config.discount,
                                config.mcts_config.default_value, None if kb is None else kb.endpoints())
    priors = np.zeros((N + 1, T, A), np.float32)
    root = np.zeros((T, A), np.float32)
    for t in rows:
        p = search.pool.read_root_priors(int(t))
        root[t, :len(p)] = p
        for e in range(1, N + 1):
            priors[e, t] = search.pool.read_block(int(t), e)['prior']
    opool.expand_roots(root, legal, n_legal)
    opool.add_root_noise(noise, config.mcts_config.root_exploration_fraction)
    for s in range(N):
        p, a = opool.select()
        np.testing.assert_array_equal(p, rec.parent[s].cpu().numpy(), err_msg=f'parent block, simulation {s}')
        np.testing.assert_array_equal(a, rec.action[s].cpu().numpy(), err_msg=f'action, simulation {s}')
        opool.expand_backprop(priors[s + 1], rec.reward[s].cpu().numpy(), rec.value[s].cpu().numpy())
    v2, rv2, mm2, pol2 = opool.root_stats()
    np.testing.assert_array_equal(visits.cpu().numpy(), v2)
    np.testing.assert_array_equal(root_value.cpu().numpy()[live], rv2[live])
    np.testing.assert_array_equal(chosen.cpu().numpy(), opool.select_action(u, 1.0))


def test_search_move_rejects_io_that_does_not_match_the_handles():
    """mz_search_move checks mz_move_io_t against the pool and the network before anything is launched."""
    import ctypes
    import torch
    from muzero_amd import _lib
    from muzero_amd.exceptions import MuZeroError
    from muzero_amd.games import cartpole, tictactoe
    from muzero_amd.selfplay import LockstepSearch
    T, N = 40, 6
    config = tictactoe.make_config(num_simulations=N)
    net = config.network_config.network_class(config, seed=1)
    search = LockstepSearch(config, net, num_trees=T)
    obs = torch.zeros(T, 27, device='cuda')
    noise = torch.full((T, 9), 1 / 9, dtype=torch.float64, device='cuda')
    u = torch.zeros(T, dtype=torch.float64, device='cuda')
    temp = torch.ones(T, dtype=torch.float64, device='cuda')
    search.run_native(obs, noise, u, temp)                   # the matching call is accepted

    def call(**override):
        pool = search.pool
        io = _lib.MoveIO(num_trees=T, num_simulations=N, hidden_size=net.hidden_size, stream=pool._stream,
                         observation=obs.data_ptr(), noise=noise.data_ptr(), noise_frac=0.25, u=u.data_ptr(),
                         temperature=temp.data_ptr(), hidden=search.hidden.data_ptr(), logits=search.logits.data_ptr(),
                         reward=search.reward.data_ptr(), value=search.value.data_ptr(),
                         parent_block=pool._parent.data_ptr(), action=pool._action.data_ptr(),
                         out_action=search.out_action.data_ptr())
        for k, v in override.items():
            setattr(io, k, v)
        return _lib.load().mz_search_move(pool._h, net._h, ctypes.byref(io))

    assert call() == 0
    assert call(num_trees=T + 1) == -1 and b'num_trees' in _lib.load().mz_last_error()
    assert call(num_simulations=N + 1) == -1 and b'num_simulations' in _lib.load().mz_last_error()
    assert call(hidden_size=128) == -1 and b'hidden_size' in _lib.load().mz_last_error()
    other = torch.cuda.Stream()
    assert call(stream=other.cuda_stream) == -1 and b'stream' in _lib.load().mz_last_error()
    cart = cartpole.make_config(num_simulations=N)
    cnet = cart.network_config.network_class(cart, seed=1)
    pool = search.pool
    io = _lib.MoveIO(num_trees=T, num_simulations=N, hidden_size=cnet.hidden_size, stream=pool._stream,
                     observation=obs.data_ptr(), noise=noise.data_ptr(), noise_frac=0.25, u=u.data_ptr(),
                     temperature=temp.data_ptr(), hidden=search.hidden.data_ptr(), logits=search.logits.data_ptr(),
                     reward=search.reward.data_ptr(), value=search.value.data_ptr(),
                     parent_block=pool._parent.data_ptr(), action=pool._action.data_ptr(),
                     out_action=search.out_action.data_ptr())
    assert _lib.load().mz_search_move(pool._h, cnet._h, ctypes.byref(io)) == -1       # 2 actions vs the pool's 9
    assert b'actions' in _lib.load().mz_last_error()
    with pytest.raises(MuZeroError):
        LockstepSearch(config, cnet, num_trees=T)


class FixtureOutput:
    """NetworkOutput whose masked_policy returns the priors the reference computed for this call (trap T4)."""

    def __init__(self, hidden_state, reward, value, policy_logits, priors):
        self.hidden_state, self.reward, self.value, self.policy_logits, self._priors = (
            hidden_state, reward, value, policy_logits, priors)

    def masked_policy(self, actions):
        return self._priors


class FixtureBatch:
    def __init__(self, out):
        self._out = out

    def split_batch(self):
        return [self._out]


class ReplayNetwork:
    """Duck-typed network (agent.py:141) replaying the outputs the reference agent was given."""

    def __init__(self, fx):
        self.fx, self.calls, self.trace = fx, 0, []

    def training_steps(self):
        return 0

    def _next(self, with_reward):
        i = self.calls
        self.calls += 1
        fx = self.fx
        return FixtureBatch(FixtureOutput(np.full((1,), float(i), np.float32),
                                          float(fx['rewards'][i]) if with_reward else None, float(fx['values'][i]),
                                          fx['logits'][i], fx['priors'][i]))

    def initial_inference(self, observation):
        self.trace.append(np.asarray(observation)[0].copy())
        return self._next(False)

    def recurrent_inference(self, hidden_state, action):
        return self._next(True)


@pytest.mark.parametrize('name', GAME_FIXTURES)
def test_dropin_agent_replays_a_reference_game(name):
    """MCTSAgent.play_game on the TicTacToe environment reproduces the reference agent's whole game: same host
    RNG seeds, same network outputs -> same actions, root values (bit-exact fp64) and policy targets."""
    from muzero_amd.agent import MCTSAgent
    from muzero_amd.environment import TicTacToeEnvironment
    from muzero_amd.games import tictactoe
    fx = load_golden(name)
    seed = int(name.split('_s')[-1])
    config = tictactoe.make_config(num_simulations=int(fx['N']))
    net = ReplayNetwork(fx)
    agent = MCTSAgent(config=config, network=net, agent_id='golden', device='cuda')
    random.seed(seed)
    np.random.seed(seed)
    history = agent.play_game(TicTacToeEnvironment())
    assert len(history) == int(fx['moves']) and net.calls == int(fx['calls'])
    np.testing.assert_array_equal(np.array(history.actions), fx['actions'])
    np.testing.assert_array_equal(np.array(history.rewards, np.float64), fx['env_rewards'])
    np.testing.assert_array_equal(np.array(history.root_values, np.float64), fx['root_values'])
    np.testing.assert_array_equal(np.stack([np.asarray(p) for p in history.policies]), fx['policies'])
    np.testing.assert_array_equal(np.stack([np.asarray(o) for o in history.observations]), fx['observations'])
    np.testing.assert_array_equal(np.array(history.to_plays), fx['to_plays'])
    np.testing.assert_array_equal(np.stack(net.trace), fx['initial_observations'])
    assert history.metadata['agent_id'] == 'golden' and history.metadata['network_id'] == '0'


def test_dropin_agent_stepwise_api():
    """The reference's step-by-step methods (select_leaf / expand_node / backpropagate, ucb_score, Node views)."""
    from muzero_amd.agent import MCTSAgent
    from muzero_amd.games import tictactoe
    from muzero_amd.utils import MinMaxStats
    fx = load_golden('ttt_legal_s2')
    config = tictactoe.make_config(num_simulations=int(fx['N']))
    agent = MCTSAgent(config=config, network=ReplayNetwork(fx), device='cuda')
    net = agent.network
    np.random.seed(0)
    root = agent.new_root()
    mm = MinMaxStats(config.value_config.known_bounds)
    legal = [int(a) for a in fx['legal']]
    agent.expand_node(root, legal, net.initial_inference(np.zeros((1, 27), np.float32)).split_batch()[0])
    assert root.expanded() and list(root.children) == legal
    from muzero_amd.tree_pool import torch
    agent._pool.add_root_noise(torch.from_numpy(fx['noise'].reshape(1, 9)).cuda(), float(fx['frac']))
    for s in range(int(fx['N'])):
        host_action, _ = agent.select_child(root, mm)          # host re-evaluation of the same rule
        action, leaf = agent.select_leaf(root, mm)
        assert action == int(fx['trace_action'][s]) and leaf.parent._block == int(fx['trace_parent'][s])
        if leaf.parent is root:
            assert host_action == action
        assert not leaf.expanded() and leaf.visit_count == 0 and np.isnan(leaf.value)
        out = net.recurrent_inference(None, None).split_batch()[0]
        agent.expand_node(leaf, config.action_space(), out)
        agent.backpropagate(leaf, out.value, mm)
        assert leaf.expanded() and leaf.visit_count == 1 and leaf.reward == float(np.float32(out.reward))
    assert root.visit_count == int(fx['N']) and root.value == float(fx['root_value'])
    np.testing.assert_array_equal([root.children[a].visit_count for a in legal], fx['root_visits'][legal])


def test_native_move_equals_python_driven_move():
    """mz_search_move (one C call per move) enqueues exactly the launches of the Python-driven LockstepSearch.run."""
    import torch
    from muzero_amd.games import tictactoe
    from muzero_amd.selfplay import LockstepSearch
    from oracle import net_oracle
    T, N = 130, 9
    config = tictactoe.make_config(num_simulations=N)
    net = config.network_config.network_class(config, weights=net_oracle.random_weights('tictactoe', 4))
    g = torch.Generator(device='cuda').manual_seed(1)
    obs = torch.randint(0, 2, (T, 27), generator=g, device='cuda').float()
    noise = torch.distributions.Dirichlet(torch.ones(9, dtype=torch.float64)).sample((T,)).cuda()
    u = torch.rand(T, generator=g, device='cuda', dtype=torch.float64)
    temp = torch.ones(T, dtype=torch.float64, device='cuda')
    search = LockstepSearch(config, net, num_trees=T)

    class NoHooks:
        before_inference = after_inference = after_tree = staticmethod(lambda s: None)

    a = [x.clone() for x in search.run(obs, noise, u, temp, hooks=NoHooks())]       # Python-driven
    hidden_a = search.hidden.clone()
    b = [x.clone() for x in search.run(obs, noise, u, temp)]                         # native
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # The hidden workspace of a native move belongs to the network: for TicTacToe every (game, cell) row of 128 floats
    # holds its split image, [128 h | 128 l] halfs with value = h + l / 2048 (21 bits; mz_net.hip, SPLIT_IO).
    img = search.hidden.view(torch.float16).reshape(-1, 2, 128).double()
    decoded = (img[:, 0] + img[:, 1] / 2048.0).reshape(hidden_a.shape)
    assert torch.allclose(decoded, hidden_a.double(), rtol=2.0 ** -20, atol=1e-9)


@pytest.mark.parametrize('game', ['tictactoe', 'cartpole'])
def test_native_move_with_unfused_tree_step(game):
    """A pool whose lane count rules out the fused network + tree-step kernels: the native move then runs the network and
    the tree step as separate launches -- on the same hidden workspace (TicTacToe: split images) -- and returns what the
    default pool returns, bit for bit (lanes per tree do not change results, test_tree_gpu.py)."""
    import torch
    from muzero_amd.games import cartpole, tictactoe
    from muzero_amd.selfplay import LockstepSearch
    from oracle import net_oracle
    mod = {'tictactoe': tictactoe, 'cartpole': cartpole}[game]
    T, N = 150, 10
    config = mod.make_config(num_simulations=N)
    net = config.network_config.network_class(config, weights=net_oracle.random_weights(game, 9))
    A = config.game_config.action_space_size
    g = torch.Generator(device='cuda').manual_seed(2)
    if game == 'tictactoe':
        obs = torch.randint(0, 2, (T, 27), generator=g, device='cuda').float()
    else:
        obs = (torch.rand(T, 4, generator=g, device='cuda') * 2 - 1) * torch.tensor([2.4, 3.0, 0.2, 3.0], device='cuda')
    noise = torch.distributions.Dirichlet(torch.ones(A, dtype=torch.float64)).sample((T,)).cuda()
    u = torch.rand(T, generator=g, device='cuda', dtype=torch.float64)
    temp = torch.ones(T, dtype=torch.float64, device='cuda')
    a = [x.clone() for x in LockstepSearch(config, net, num_trees=T).run(obs, noise, u, temp)]
    other = {'tictactoe': 8, 'cartpole': 4}[game]
    b = [x.clone() for x in LockstepSearch(config, net, num_trees=T, lanes_per_tree=other).run(obs, noise, u, temp)]
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_atari_fused_step_equals_separate_launches():
    """The Atari-shaped native move (network and tree step of 64 games in one launch, eight lanes per tree) against the
    Python-driven move (mz_infer_recurrent, then mz_step_logits with the pool's own lanes per tree): bit for bit, some
    games finished, some roots with few legal actions."""
    import torch
    from muzero_amd.games import atari_shaped
    from muzero_amd.selfplay import LockstepSearch
    from oracle import net_oracle
    T, N = 150, 14
    config = atari_shaped.make_config(num_simulations=N)
    net = config.network_config.network_class(config, weights=net_oracle.random_weights('atari', 6))
    A = config.game_config.action_space_size
    g = torch.Generator(device='cuda').manual_seed(4)
    obs = torch.randn(T, 256, generator=g, device='cuda')
    noise = torch.distributions.Dirichlet(torch.ones(A, dtype=torch.float64)).sample((T,)).cuda()
    u = torch.rand(T, generator=g, device='cuda', dtype=torch.float64)
    temp = torch.ones(T, dtype=torch.float64, device='cuda')
    n_legal = torch.full((T,), A, dtype=torch.int32, device='cuda')
    n_legal[::9] = 0
    n_legal[4::13] = 3
    legal = torch.arange(A, dtype=torch.int32, device='cuda').repeat(T, 1).contiguous()
    search = LockstepSearch(config, net, num_trees=T)

    class NoHooks:
        before_inference = after_inference = after_tree = staticmethod(lambda s: None)

    a = [x.clone() for x in search.run(obs, noise, u, temp, legal, n_legal, hooks=NoHooks())]     # separate launches
    blocks_a = [[search.pool.read_block(t, e) for e in range(N + 1)] for t in (1, 4, T - 1)]
    b = [x.clone() for x in search.run(obs, noise, u, temp, legal, n_legal)]                       # native, fused
    active = n_legal > 0
    for x, y in zip(a, b):
        assert torch.equal(x[active], y[active])
    for t, ref in zip((1, 4, T - 1), blocks_a):
        for e in range(N + 1):
            got = search.pool.read_block(t, e)
            for k in ref[e]:
                np.testing.assert_array_equal(ref[e][k], got[k], err_msg=f'tree {t} block {e} {k}')


@pytest.mark.parametrize('bounded', [True, False])
def test_cartpole_run_kernel_equals_the_launch_per_simulation(bounded):
    """The CartPole move whose simulations are one launch (trees in LDS, the two-action quad select / backup of
    cartpole16_run_kernel) against the same move as one launch per simulation (a pool with four lanes per tree takes
    mz_tree_dev.h's general tree code): bit for bit, with and without known value bounds (MinMaxStats live: first-
    simulation NaN scores, renormalised exploitation terms, min/max updates), with roots of one legal action, roots
    whose legal list is [1, 0], and finished games."""
    import copy
    import torch
    from muzero_amd.games import cartpole
    from muzero_amd.selfplay import LockstepSearch
    from oracle import net_oracle
    T, N = 203, 24
    config = cartpole.make_config(num_simulations=N)
    net = config.network_config.network_class(config, weights=net_oracle.random_weights('cartpole', 21))
    if not bounded:
        config = copy.copy(config)
        config.value_config = copy.copy(config.value_config)
        config.value_config.known_bounds = None
    g = torch.Generator(device='cuda').manual_seed(8)
    obs = (torch.rand(T, 4, generator=g, device='cuda') * 2 - 1) * torch.tensor([2.4, 3.0, 0.2, 3.0], device='cuda')
    noise = torch.distributions.Dirichlet(torch.ones(2, dtype=torch.float64)).sample((T,)).cuda()
    u = torch.rand(T, generator=g, device='cuda', dtype=torch.float64)
    temp = torch.ones(T, dtype=torch.float64, device='cuda')
    n_legal = torch.full((T,), 2, dtype=torch.int32, device='cuda')
    legal = torch.arange(2, dtype=torch.int32, device='cuda').repeat(T, 1).contiguous()
    n_legal[::11] = 0                                   # finished games
    n_legal[5::17] = 1                                  # one legal action ...
    legal[5::34, 0] = 1                                 # ... which is action 1 for half of them
    legal[3::19] = torch.tensor([1, 0], dtype=torch.int32, device='cuda')       # both, listed in reverse
    fast = LockstepSearch(config, net, num_trees=T)
    slow = LockstepSearch(config, net, num_trees=T, lanes_per_tree=4)
    ta, tb = {}, {}
    a = [x.clone() for x in fast.run_native(obs, noise, u, temp, legal, n_legal, trace=ta)]
    b = [x.clone() for x in slow.run_native(obs, noise, u, temp, legal, n_legal, trace=tb)]
    active = n_legal > 0
    for x, y in zip(a, b):
        assert torch.equal(x[active], y[active])
    for k in ('reward', 'value', 'logits', 'parent', 'action'):
        assert torch.equal(ta[k][:, active], tb[k][:, active]), k
    for t in (1, 3, 5, 39, T - 1):
        for e in range(N + 1):
            x, y = fast.pool.read_block(t, e), slow.pool.read_block(t, e)
            for k in x:
                np.testing.assert_array_equal(x[k], y[k], err_msg=f'tree {t} block {e} {k}')
    assert fast.pool.counters()[:2].tolist() == slow.pool.counters()[:2].tolist()


@pytest.mark.parametrize('game', ['cartpole', 'tictactoe', 'atari'])
def test_run_of_simulations_in_chunks(game):
    """mz_infer_recurrent_run (reference agent.py:198-211; CartPole and Atari-shaped: all simulations of the call in one
    launch, CartPole's trees in LDS meanwhile) called twice on one pool -- a first chunk that ends with a select, a second one that continues from
    records the first left in the pool -- gives the move of one call, bit for bit: statistics, sampled actions, every
    record of sampled trees, the traced selections and outputs."""
    import torch
    from muzero_amd.games import atari_shaped, cartpole, tictactoe
    from muzero_amd.selfplay import LockstepSearch
    from oracle import net_oracle
    mod = {'tictactoe': tictactoe, 'cartpole': cartpole, 'atari': atari_shaped}[game]
    T, N, first = 150, 12, 5
    config = mod.make_config(num_simulations=N)
    net = config.network_config.network_class(config, weights=net_oracle.random_weights(game, 11))
    A = config.game_config.action_space_size
    g = torch.Generator(device='cuda').manual_seed(5)
    if game == 'tictactoe':
        obs = torch.randint(0, 2, (T, 27), generator=g, device='cuda').float()
    elif game == 'atari':
        obs = torch.randn(T, 256, generator=g, device='cuda')
    else:
        obs = (torch.rand(T, 4, generator=g, device='cuda') * 2 - 1) * torch.tensor([2.4, 3.0, 0.2, 3.0], device='cuda')
    noise = torch.distributions.Dirichlet(torch.ones(A, dtype=torch.float64)).sample((T,)).cuda()
    u = torch.rand(T, generator=g, device='cuda', dtype=torch.float64)
    temp = torch.ones(T, dtype=torch.float64, device='cuda')
    n_legal = torch.full((T,), A, dtype=torch.int32, device='cuda')
    n_legal[::13] = 0                                                     # some finished games
    legal = torch.arange(A, dtype=torch.int32, device='cuda').repeat(T, 1).contiguous()
    search = LockstepSearch(config, net, num_trees=T)
    pool = search.pool
    trace_a = {}
    a = [x.clone() for x in search.run_native(obs, noise, u, temp, legal, n_legal, trace=trace_a)]
    trees = [1, 2, 77, T - 1]
    blocks_a = [[pool.read_block(t, e) for e in range(N + 1)] for t in trees]

    pool.reset()
    net.initial_inference_pool(obs, search.hidden, search.root_value_net, search.logits)
    pool.expand_roots(search.logits, legal, n_legal, logits=True)
    pool.add_root_noise(noise, search.frac)
    pool.select()
    mk = lambda n: dict(reward=torch.zeros(n, T, device='cuda'), value=torch.zeros(n, T, device='cuda'),
                        logits=torch.zeros(n, T, A, device='cuda'),
                        parent=torch.zeros(n, T, dtype=torch.int32, device='cuda'),
                        action=torch.zeros(n, T, dtype=torch.int32, device='cuda'))
    t1, t2 = mk(first), mk(N - first)
    net.run_simulations_pool(pool, first, search.hidden, None, None, None, final_select=True, trace=t1)
    assert pool.num_blocks == first + 1
    net.run_simulations_pool(pool, N - first, search.hidden, None, None, None, final_select=False, trace=t2)
    assert pool.num_blocks == N + 1
    visits, root_value, minmax, policy = pool.root_stats()
    chosen = pool.select_action(u, temp)
    active = n_legal > 0
    for x, y in zip(a, (visits, root_value, minmax, policy, chosen)):
        assert torch.equal(x[active], y[active])
    for k in ('reward', 'value', 'logits', 'parent', 'action'):
        both = torch.cat([t1[k], t2[k]])
        assert torch.equal(both[:, active], trace_a[k][:, active]), k
    for t, ref in zip(trees, blocks_a):
        for e in range(N + 1):
            got = pool.read_block(t, e)
            for k in ref[e]:
                np.testing.assert_array_equal(ref[e][k], got[k], err_msg=f'tree {t} block {e} {k}')
    from muzero_amd.exceptions import MuZeroError
    with pytest.raises(MuZeroError):                                      # the pool is full
        net.run_simulations_pool(pool, 1, search.hidden, search.reward, search.value, search.logits)


@pytest.mark.parametrize('game', ['cartpole', 'tictactoe', 'atari'])
def test_native_move_ends_like_root_stats_and_select_action(game):
    """The last launch of a native move (root_finish_kernel: for action axes below 64 one wavefront per tree that scatters
    the counts with a lane permute and sums the sampling weights over readlane) against mz_root_stats + mz_select_action
    on the trees the move left in the pool (agent.py:162-168, :229-231): arg-max (t = 0), the exact-product temperatures,
    registered pow tables (0.7, 3.0), an unregistered temperature, ragged legal lists and finished games."""
    import torch
    from muzero_amd.games import atari_shaped, cartpole, tictactoe
    from muzero_amd.selfplay import LockstepSearch
    from oracle import net_oracle
    mod = {'tictactoe': tictactoe, 'cartpole': cartpole, 'atari': atari_shaped}[game]
    T, N = 140, 20
    config = mod.make_config(num_simulations=N)
    net = config.network_config.network_class(config, weights=net_oracle.random_weights(game, 11))
    A = config.game_config.action_space_size
    g = torch.Generator(device='cuda').manual_seed(9)
    if game == 'tictactoe':
        obs = torch.randint(0, 2, (T, 27), generator=g, device='cuda').float()
    elif game == 'atari':
        obs = torch.randn(T, 256, generator=g, device='cuda')
    else:
        obs = (torch.rand(T, 4, generator=g, device='cuda') * 2 - 1) * torch.tensor([2.4, 3.0, 0.2, 3.0], device='cuda')
    noise = torch.distributions.Dirichlet(torch.ones(A, dtype=torch.float64)).sample((T,)).cuda()
    u = torch.rand(T, generator=g, device='cuda', dtype=torch.float64)
    temps = [0.0, 0.7, 3.0, 0.5, 0.25, 1.0, 0.9]
    temp = torch.tensor([temps[i % len(temps)] for i in range(T)], dtype=torch.float64, device='cuda')
    rng = np.random.default_rng(3)
    n_legal_h = rng.integers(1, A + 1, size=T).astype(np.int32)
    n_legal_h[::11] = 0                                                   # finished games
    legal_h = np.zeros((T, A), np.int32)
    for t in range(T):                                                    # a random subset, in random order
        legal_h[t, :n_legal_h[t]] = rng.permutation(A)[:n_legal_h[t]]
    legal, n_legal = torch.tensor(legal_h, device='cuda'), torch.tensor(n_legal_h, device='cuda')
    search = LockstepSearch(config, net, num_trees=T)
    pool = search.pool
    pool.set_temperature(0.7)
    pool.set_temperature(3.0)
    got = [x.clone() for x in search.run_native(obs, noise, u, temp, legal, n_legal)]
    visits, root_value, minmax, policy = pool.root_stats()
    chosen = pool.select_action(u, temp)
    active = n_legal > 0
    for name, x, y in zip(('visits', 'root_value', 'minmax', 'policy', 'action'), got,
                          (visits, root_value, minmax, policy, chosen)):
        assert torch.equal(x[active], y[active]), name
    assert torch.equal(got[0][~active], torch.zeros_like(got[0][~active]))           # no children: zeros, action -1
    assert torch.all(got[4][~active] == -1)
    assert int(got[0][active].sum()) == int(active.sum()) * N
    # the chosen action is one of the game's legal actions
    ga = got[4].cpu().numpy()
    for t in np.nonzero(n_legal_h > 0)[0]:
        assert ga[t] in legal_h[t, :n_legal_h[t]]


@pytest.mark.parametrize('lanes', [2, 3])
def test_pipelined_lanes_equal_one_lockstep_search(lanes):
    """PipelinedSearch (games split into lanes on separate HIP streams) returns what one LockstepSearch over the same
    games returns, bit for bit: lanes share nothing but the weights."""
    import torch
    from muzero_amd.games import tictactoe
    from muzero_amd.selfplay import LockstepSearch, PipelinedSearch
    from oracle import net_oracle
    T, N = 203, 11
    config = tictactoe.make_config(num_simulations=N)
    net = config.network_config.network_class(config, weights=net_oracle.random_weights('tictactoe', 5))
    g = torch.Generator(device='cuda').manual_seed(3)
    obs = torch.randint(0, 2, (T, 27), generator=g, device='cuda').float()
    noise = torch.distributions.Dirichlet(torch.ones(9, dtype=torch.float64)).sample((T,)).cuda()
    u = torch.rand(T, generator=g, device='cuda', dtype=torch.float64)
    temp = torch.ones(T, dtype=torch.float64, device='cuda')
    n_legal = torch.full((T,), 9, dtype=torch.int32, device='cuda')
    n_legal[::17] = 0                                                     # some finished games
    legal = torch.arange(9, dtype=torch.int32, device='cuda').repeat(T, 1).contiguous()
    one = LockstepSearch(config, net, num_trees=T)
    a = [x.clone() for x in one.run(obs, noise, u, temp, legal, n_legal)]
    piped = PipelinedSearch(config, net, num_trees=T, lanes=lanes)
    for _ in range(2):
        b = piped.run(obs, noise, u, temp, legal, n_legal)
        torch.cuda.synchronize()
        active = (n_legal > 0)
        for x, y in zip(a, b):
            assert torch.equal(x[active], y[active])
    assert piped.counters()[0] == 2 * int(active.sum()) * N


def test_integration_section_1_verbatim_with_the_hip_network():
    """INTEGRATION.md section 1 as written: MCTSAgent(config, config.make_uniform_network(), ...).play_game(...) -- one
    game, 25 simulations per move, the HIP network called at batch 1 like the reference calls its network
    (agent.py:204-207, BASELINE configs[0]'s shape on the product path).  Every network call is recorded; each move's
    tree is then replayed through the C oracle with those outputs (selections, policy, root value and sampled action
    bit for bit) and the outputs themselves are held to the fp64 network oracle at 1e-5."""
    import random
    import torch
    from muzero_amd.agent import MCTSAgent
    from muzero_amd.games.tictactoe import make_config
    from oracle import c_oracle, net_oracle
    N = 25
    config = make_config(num_simulations=N)
    network = config.make_uniform_network()           # HIP-backed TicTacToeNetwork
    agent = MCTSAgent(config, network, 'a0')
    calls = []
    initial, recurrent = network.initial_inference, network.recurrent_inference

    def rec_initial(observation):
        out = initial(observation)
        calls.append(('initial', np.asarray(torch.as_tensor(observation).cpu(), np.float32).reshape(1, 27), None, out))
        return out

    def rec_recurrent(hidden_state, action):
        out = recurrent(hidden_state, action)
        calls.append(('recurrent', torch.as_tensor(hidden_state).cpu().numpy().reshape(1, 1152).copy(),
                      int(np.asarray(action).reshape(-1)[0]), out))
        return out

    network.initial_inference, network.recurrent_inference = rec_initial, rec_recurrent
    np.random.seed(1234)
    random.seed(1234)
    history = agent.play_game(config.create_environment())
    moves = len(history)
    assert 5 <= moves <= 9 and len(calls) == moves * (N + 1)
    assert history.metadata['agent_id'] == 'a0' and history.metadata['network_id'] == '0'

    ref = net_oracle.make_oracle('tictactoe', network.get_weights())
    env = config.create_environment()
    state = env.reset()
    np.random.seed(1234)
    random.seed(1234)
    kb = config.value_config.known_bounds.endpoints()
    for m in range(moves):
        mc = calls[m * (N + 1):(m + 1) * (N + 1)]
        assert mc[0][0] == 'initial' and all(c[0] == 'recurrent' for c in mc[1:])
        np.testing.assert_array_equal(mc[0][1].reshape(3, 3, 3), np.asarray(state.observation, np.float32))
        outs = [c[3].split_batch()[0] for c in mc]
        hidden = [o.hidden_state.cpu().numpy().reshape(1, 1152) for o in outs]
        # -- the network against the fp64 oracle
        h0, v0, l0, _ = ref.initial_inference(mc[0][1])
        np.testing.assert_allclose(hidden[0], h0, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(outs[0].value, v0[0], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(outs[0].policy_logits.cpu().numpy(), l0[0], rtol=1e-5, atol=1e-5)
        # -- the tree through the C oracle, fed the device's outputs
        legal = list(state.legal_actions)
        opool = c_oracle.OraclePool(1, 9, N, 2, config.game_config.discount, config.mcts_config.default_value, kb)
        lg = np.zeros((1, 9), np.int32)
        lg[0, :len(legal)] = legal
        opool.expand_roots(outs[0].masked_policy(legal).reshape(1, 9), lg, np.array([len(legal)], np.int32))
        noise = np.zeros((1, 9))
        noise[0, :len(legal)] = np.random.dirichlet([config.mcts_config.root_dirichlet_alpha] * len(legal))
        opool.add_root_noise(noise, config.mcts_config.root_exploration_fraction)
        for s in range(N):
            pb, a = opool.select()
            kind, hid_in, act_in, _ = mc[s + 1]
            assert act_in == int(a[0]), f'move {m} simulation {s}: action'
            np.testing.assert_array_equal(hid_in, hidden[int(pb[0])], err_msg=f'move {m} simulation {s}: parent state')
            o = outs[s + 1]
            h2, r2, v2, l2, _ = ref.recurrent_inference(hid_in, np.array([act_in]))
            np.testing.assert_allclose(hidden[s + 1], h2, rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose([o.reward, o.value], [r2[0], v2[0]], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(o.policy_logits.cpu().numpy(), l2[0], rtol=1e-5, atol=1e-5)
            opool.expand_backprop(o.masked_policy(list(range(9))).reshape(1, 9), np.array([o.reward], np.float32),
                                  np.array([o.value], np.float32))
        _, root_value, _, policy = opool.root_stats()
        assert history.root_values[m] == root_value[0]
        np.testing.assert_array_equal(np.asarray(history.policies[m], np.float32), policy[0])
        u = random.random()                               # random.choices consumes one uniform per sampled action
        assert history.actions[m] == int(opool.select_action(np.array([u]), 1.0)[0])
        state, reward, done, _ = env.step(int(history.actions[m]))
        assert reward == history.rewards[m]
    assert done


@pytest.mark.parametrize('T', [64, 300, 1100])
def test_tictactoe_one_launch_move_equals_the_launch_per_simulation(T):
    """ttt_run_kernel (all simulations of a move in one launch; a row block's four workgroups meet through their XCD's L2)
    against the launch-per-simulation kernels it replaces, and against itself with agent-scope fences at every meeting
    (the path it takes when the workgroups are not on one XCD): results, traces and hidden workspace bit for bit.
    Ragged legal lists, finished games, a batch that is not a multiple of 64."""
    import torch
    from muzero_amd.games import tictactoe
    from muzero_amd.selfplay import LockstepSearch
    N, A = 25, 9
    config = tictactoe.make_config(num_simulations=N)
    net = config.network_config.network_class(config, seed=11)
    net.set_job_rows(64)          # the one-launch move is the 64-game jobs' (by default only batches > 2560 take them)
    rng = np.random.default_rng(T)
    obs = rng.integers(0, 2, size=(T, 27)).astype(np.float32)
    n_legal = rng.integers(0, A + 1, size=T).astype(np.int32)
    n_legal[:4] = A
    legal = np.zeros((T, A), np.int32)
    noise = np.zeros((T, A))
    for t in range(T):
        legal[t, :n_legal[t]] = np.sort(rng.permutation(A)[:n_legal[t]])
        if n_legal[t]:
            noise[t, :n_legal[t]] = rng.dirichlet([1.0] * int(n_legal[t]))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (dev(obs), dev(noise), dev(rng.random(T)), dev(np.ones(T)), dev(legal), dev(n_legal))
    results = {}
    modes = (2, 0, 1)
    for mode in modes:
        net.set_run_mode(mode)
        fenced0, runs0 = net.run_info()
        search = LockstepSearch(config, net, num_trees=T)
        search.hidden.zero_()
        tr = {}
        out = [x.cpu().numpy().copy() for x in search.run_native(*args, trace=tr)]
        search.pool.counters()                                    # raises if the run gave up or the pool is inconsistent
        fenced1, runs1 = net.run_info()
        row_blocks = (T + 63) // 64
        assert runs1 - runs0 == (0 if mode == 2 else row_blocks)
        if mode == 1:
            assert fenced1 - fenced0 == row_blocks
        results[mode] = (out, {k: v.cpu().numpy().copy() for k, v in tr.items()}, search.hidden.cpu().numpy().copy(),
                         [search.pool.read_block(int(t), e) for t in np.nonzero(n_legal > 0)[0][[0, -1]] for e in (0, 1, N)])
        if mode == 0:
            results['fenced_in_default_mode'] = fenced1 - fenced0
    net.set_run_mode(0)
    live = n_legal > 0
    for mode in modes[1:]:
        for a, b in zip(results[mode][0], results[2][0]):
            np.testing.assert_array_equal(a, b)
        for k in results[2][1]:
            np.testing.assert_array_equal(results[mode][1][k], results[2][1][k], err_msg=f'trace {k}, mode {mode}')
        np.testing.assert_array_equal(results[mode][2][:, live].view(np.uint32), results[2][2][:, live].view(np.uint32))
        for x, y in zip(results[mode][3], results[2][3]):
            for key in x:
                np.testing.assert_array_equal(x[key], y[key])
    assert results[2][0][0][live].sum() == live.sum() * N
    # on this part the four workgroups of a row block are dealt to one XCD: the L2 path is the one that ran
    assert results['fenced_in_default_mode'] == 0


def test_tictactoe_single_cell_job_shapes_are_bit_identical():
    """The 16-game and the 32-game single-cell jobs (ttt_ring_kernel<., 1> / <., 2>, chosen by batch size) compute every row
    with the same instructions in the same order and hand the head the same nine per-cell slices: a whole move -- outputs,
    traces, hidden workspace -- must agree bit for bit.  A batch that is a multiple of neither 16 nor 32, ragged legal
    lists, finished games."""
    import torch
    from muzero_amd.games import tictactoe
    from muzero_amd.selfplay import LockstepSearch
    T, N, A = 210, 25, 9
    config = tictactoe.make_config(num_simulations=N)
    net = config.network_config.network_class(config, seed=5)
    rng = np.random.default_rng(77)
    obs = rng.integers(0, 2, size=(T, 27)).astype(np.float32)
    n_legal = rng.integers(0, A + 1, size=T).astype(np.int32)
    n_legal[:4] = A
    legal = np.zeros((T, A), np.int32)
    noise = np.zeros((T, A))
    for t in range(T):
        legal[t, :n_legal[t]] = np.sort(rng.permutation(A)[:n_legal[t]])
        if n_legal[t]:
            noise[t, :n_legal[t]] = rng.dirichlet([1.0] * int(n_legal[t]))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (dev(obs), dev(noise), dev(rng.random(T)), dev(np.ones(T)), dev(legal), dev(n_legal))
    results = {}
    for rows in (16, 32):
        net.set_job_rows(rows)
        search = LockstepSearch(config, net, num_trees=T)
        search.hidden.zero_()
        tr = {}
        out = [x.cpu().numpy().copy() for x in search.run_native(*args, trace=tr)]
        search.pool.counters()
        results[rows] = (out, {k: v.cpu().numpy().copy() for k, v in tr.items()}, search.hidden.cpu().numpy().copy())
    net.set_job_rows(0)
    live = n_legal > 0
    for a, b in zip(results[16][0], results[32][0]):
        np.testing.assert_array_equal(a, b)
    for k in results[16][1]:
        np.testing.assert_array_equal(results[16][1][k], results[32][1][k], err_msg=f'trace {k}')
    np.testing.assert_array_equal(results[16][2][:, live].view(np.uint32), results[32][2][:, live].view(np.uint32))
    assert results[16][0][0][live].sum() == live.sum() * N


@pytest.mark.parametrize('name', ['ttt_legal_s2', 'atari_s9', 'asym_bounds_s15'])
def test_select_leaf_from_a_node_below_the_root(name):
    """The reference's select_leaf takes ANY node (agent.py:177-181).  Here: whenever the root's own choice is an expanded
    child, the descent is started at THAT child (host arithmetic of select_child / ucb_score, path adopted by the pool via
    mz_pool_set_selection) -- it must end at the leaf the reference reached from the root in the same simulation, and the
    search statistics after all simulations must be the reference's, bit for bit (known and live value bounds, a root
    with a non-monotone legal list)."""
    import torch
    from muzero_amd.agent import MCTSAgent
    from muzero_amd.config import GameConfig, MCTSConfig, MuZeroConfig, NetworkConfig, ScalarConfig
    from muzero_amd.exceptions import MuZeroError
    from muzero_amd.utils import KnownBounds, MinMaxStats
    fx = load_golden(name)
    A, N = int(fx['A']), int(fx['N'])
    kb = KnownBounds(minv=float(fx['known_min']), maxv=float(fx['known_max'])) if int(fx['has_bounds']) else None
    config = MuZeroConfig(
        game_config=GameConfig(name='synthetic', environment_class=None, environment_parameters={}, action_space_size=A,
                               num_players=int(fx['num_players']), discount=float(fx['discount'])),
        replay_buffer_config=None, training_config=None,
        mcts_config=MCTSConfig(max_moves=10, root_dirichlet_alpha=1.0, root_exploration_fraction=float(fx['frac']),
                               num_simulations=N, temperature=1.0, freezing_moves=10,
                               default_value=float(fx['default_value'])),
        network_config=NetworkConfig(network_class=None),
        reward_config=ScalarConfig(known_bounds=None, support_size=None, loss_decay=0.0),
        value_config=ScalarConfig(known_bounds=kb, support_size=None, loss_decay=0.0))
    net = ReplayNetwork(fx)
    agent = MCTSAgent(config=config, network=net, device='cuda')
    root = agent.new_root()
    mm = MinMaxStats(config.value_config.known_bounds)
    legal = [int(a) for a in fx['legal']]
    agent.expand_node(root, legal, net.initial_inference(np.zeros((1, 1), np.float32)).split_batch()[0])
    noise = np.zeros((1, A))
    noise[0] = fx['noise']
    agent._pool.add_root_noise(torch.from_numpy(noise).cuda(), float(fx['frac']))
    below = 0
    for s in range(N):
        if s == 3:                                                  # a chain that is not one is refused, nothing changes
            with pytest.raises(MuZeroError):
                agent._pool.set_selection(0, [0, N], [legal[0], 0])
        first_action, first = agent.select_child(root, mm)
        if first.expanded():
            action, leaf = agent.select_leaf(first, mm)             # started below the root
            below += 1
            with pytest.raises(MuZeroError):
                agent.select_leaf(first, mm)                        # a leaf is pending
        else:
            action, leaf = agent.select_leaf(root, mm)
        assert action == int(fx['trace_action'][s]), f'simulation {s}'
        parent_block = leaf.parent._block if leaf.parent.parent is None else leaf.parent._own_block()
        assert parent_block == int(fx['trace_parent'][s]), f'simulation {s}'
        assert not leaf.expanded()
        out = net.recurrent_inference(None, None).split_batch()[0]
        agent.expand_node(leaf, config.action_space(), out)
        agent.backpropagate(leaf, out.value, mm)
    assert below > N // 3
    assert root.visit_count == N and root.value == float(fx['root_value'])
    np.testing.assert_array_equal([root.children[a].visit_count for a in legal], fx['root_visits'][legal])
    if kb is None:
        assert (mm.minimum, mm.maximum) == (float(fx['mm_min']), float(fx['mm_max']))
