"""Batched device environments and the lockstep self-play loop against the single-game host
classes, which follow the reference's environment semantics (games/tictactoe.py:71-148, environment.py:44-63)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tictactoe_vector_env_matches_host_env():
    import torch
    from muzero_amd.environment import TicTacToeEnvironment
    from muzero_amd.vector_env import TicTacToeVectorEnv
    G = 300
    rng = np.random.default_rng(11)
    venv = TicTacToeVectorEnv(G)
    assert venv.all_ended()                                   # nothing to play before reset
    venv.reset()
    hosts = [TicTacToeEnvironment() for _ in range(G)]
    states = [h.reset() for h in hosts]
    done = [False] * G
    for move in range(10):
        obs, legal, n_legal = (x.cpu().numpy().copy() for x in venv.observe())
        to_play = venv.to_play.cpu().numpy()
        for g in range(G):
            np.testing.assert_array_equal(obs[g].reshape(3, 3, 3), states[g].observation)
            assert to_play[g] == states[g].to_play
            want = states[g].legal_actions
            assert n_legal[g] == len(want) and list(legal[g, :len(want)]) == want and (legal[g, len(want):] == -1).all()
        if all(done):
            break
        action = np.full(G, -1, np.int32)
        for g in range(G):
            if not done[g]:
                action[g] = rng.choice(states[g].legal_actions)
        reward = venv.step(torch.from_numpy(action).cuda()).cpu().numpy().copy()
        status, ended = venv.status.cpu().numpy(), venv.ended.cpu().numpy()
        for g in range(G):
            if done[g]:
                assert status[g] == 1 and reward[g] == 0.0
                continue
            states[g], r, d, _ = hosts[g].step(int(action[g]))
            done[g] = d
            assert status[g] == 0 and reward[g] == np.float32(r) and bool(ended[g]) == d
    assert all(done) and venv.all_ended()
    venv.check_illegal()                                      # no illegal action was played
    np.testing.assert_array_equal(venv.steps.cpu().numpy(), [h.steps for h in hosts])


def test_tictactoe_vector_env_rejects_illegal_actions():
    import torch
    from muzero_amd.exceptions import MuZeroEnvironmentError
    from muzero_amd.vector_env import TicTacToeVectorEnv
    venv = TicTacToeVectorEnv(4)
    venv.reset()
    venv.step(torch.tensor([0, 4, 8, 2], dtype=torch.int32, device='cuda'))
    before = venv.board.clone()
    venv.step(torch.tensor([0, 9, -1, 3], dtype=torch.int32, device='cuda'))     # occupied, out of range x2, legal
    assert venv.status.tolist() == [2, 2, 2, 0]
    assert torch.equal(venv.board[:3], before[:3]) and venv.steps.tolist() == [1, 1, 1, 2]
    with pytest.raises(MuZeroEnvironmentError):
        venv.check_illegal()


def test_cartpole_vector_env_matches_host_restatement():
    import torch
    from muzero_amd.environment import CartPoleEnvironment
    from muzero_amd.vector_env import CartPoleVectorEnv
    G, rng = 64, np.random.default_rng(3)
    venv = CartPoleVectorEnv(G, seed=5)
    venv.reset()
    init = venv.state.cpu().numpy().copy()
    assert (np.abs(init) <= 0.05).all() and len(np.unique(init)) > G
    hosts = [CartPoleEnvironment() for _ in range(G)]
    for g, h in enumerate(hosts):
        h.reset()
        h.state = init[g].copy()
    done = np.zeros(G, bool)
    for move in range(200):
        obs, legal, n_legal = (x.cpu().numpy().copy() for x in venv.observe())
        assert (n_legal == np.where(done, 0, 2)).all() and (legal == [0, 1]).all()
        if done.all():
            break
        # a biased policy so that episodes end at different times
        action = (rng.random(G) < 0.8).astype(np.int32)
        reward = venv.step(torch.from_numpy(action).cuda()).cpu().numpy().copy()
        state, ended = venv.state.cpu().numpy(), venv.ended.cpu().numpy().astype(bool)
        for g in range(G):
            if done[g]:
                assert reward[g] == 0.0
                continue
            st, r, d, _ = hosts[g].step(int(action[g]))
            # fp64 state: same operation order; device sin/cos differ from glibc by a few ulp at most
            np.testing.assert_allclose(state[g], hosts[g].state, rtol=1e-12, atol=1e-13)
            assert reward[g] == 1.0 and ended[g] == d
            done[g] = d
    assert done.all()


@pytest.mark.parametrize('lanes', [1, 2])
def test_selfplay_engine_plays_legal_complete_tictactoe_games(lanes):
    """The batched play_game loop: every recorded game replays on the host environment to the same rewards,
    players, observations and length; search statistics are well-formed."""
    import torch
    from muzero_amd.environment import TicTacToeEnvironment
    from muzero_amd.games import tictactoe
    from muzero_amd.selfplay import SelfPlayEngine
    G, N = 150, 6
    config = tictactoe.make_config(num_simulations=N)
    net = config.network_config.network_class(config, seed=3)
    engine = SelfPlayEngine(config, net, G, seed=17, lanes=lanes)
    hist = engine.play_games()
    games = hist.to_game_histories()
    assert len(games) == G
    lengths = hist.lengths.cpu().numpy()
    assert lengths.min() >= 5 and lengths.max() <= 9 and len(set(lengths.tolist())) > 1
    for g, h in enumerate(games):
        env = TicTacToeEnvironment()
        state = env.reset()
        assert len(h) == lengths[g] and len(h.observations) == len(h) + 1 == len(h.to_plays)
        for m, a in enumerate(h.actions):
            np.testing.assert_array_equal(h.observations[m], state.observation)
            assert h.to_plays[m] == state.to_play and a in state.legal_actions
            pol = h.policies[m]
            assert pol.shape == (9,) and abs(float(pol.sum()) - 1.0) < 1e-6
            assert (pol[[x for x in range(9) if x not in state.legal_actions]] == 0).all()
            assert (np.round(pol * N) == pol * N).all() or np.allclose(np.round(pol * N), pol * N, atol=1e-5)
            assert -1.0 <= h.root_values[m] <= 1.0
            state, r, done, _ = env.step(a)
            assert h.rewards[m] == r
        assert done
        np.testing.assert_array_equal(h.observations[-1], state.observation)
        assert h.to_plays[-1] == state.to_play
        assert h.metadata['network_id'] == '0'
    # a second batch re-uses the engine; a different random stream gives different games
    again = engine.play_games()
    assert not torch.equal(again.actions, hist.actions)


@pytest.mark.parametrize('lanes,fused', [(1, True), (1, False), (2, True)])
def test_selfplay_engine_refills_finished_slots(lanes, fused):
    """play_continuous: a slot whose game ends is flushed and restarted before the next search (the reference agent's
    loop, agent_services.py:47-54).  Every stored game -- including the refilled ones -- replays on the host environment
    to the same observations, players, rewards and length; the store holds exactly the first `num_games` games started."""
    import torch
    from muzero_amd.environment import TicTacToeEnvironment
    from muzero_amd.games import tictactoe
    from muzero_amd.selfplay import SelfPlayEngine
    G, N, total = 48, 5, 200
    config = tictactoe.make_config(num_simulations=N)
    net = config.network_config.network_class(config, seed=3)
    engine = SelfPlayEngine(config, net, G, seed=23, lanes=lanes)
    hist = engine.play_continuous(total, check_every=3, fused=fused)
    assert hist.G == total and hist.metadata['slots'] == G and hist.metadata['games_started'] >= total
    games = hist.to_game_histories()
    lengths = hist.lengths.cpu().numpy()
    assert lengths.min() >= 5 and lengths.max() <= 9 and len(set(lengths.tolist())) > 1
    # more than four waves of games went through the 48 slots, and every move searched a full batch:
    # moves played x slots >= the moves of the stored games
    assert hist.metadata['moves_played'] * G >= int(lengths.sum()) and total > 4 * G
    for g, h in enumerate(games):
        env = TicTacToeEnvironment()
        state = env.reset()
        assert len(h) == lengths[g] and len(h.observations) == len(h) + 1 == len(h.to_plays)
        done = False
        for m, a in enumerate(h.actions):
            np.testing.assert_array_equal(h.observations[m], state.observation)
            assert h.to_plays[m] == state.to_play and a in state.legal_actions
            pol = h.policies[m]
            assert abs(float(pol.sum()) - 1.0) < 1e-6
            assert (pol[[x for x in range(9) if x not in state.legal_actions]] == 0).all()
            assert -1.0 <= h.root_values[m] <= 1.0
            state, r, done, _ = env.step(a)
            assert h.rewards[m] == r
        assert done
        np.testing.assert_array_equal(h.observations[-1], state.observation)
        assert h.to_plays[-1] == state.to_play
    # same seed, same games; the first G games are the ones a plain batch would start with
    again = SelfPlayEngine(config, net, G, seed=23, lanes=lanes).play_continuous(total, check_every=5, fused=fused)
    for f in hist.FIELDS:
        assert torch.equal(getattr(hist, f), getattr(again, f)), f


@pytest.mark.parametrize('fused', [True, False])
def test_selfplay_engine_refills_cartpole_with_a_move_cap(fused):
    """Slots that reach the move capacity without the episode ending are flushed as truncated games (agent.py:34's
    `len(history) < max_moves`) and restarted like finished ones."""
    from muzero_amd.games import cartpole
    from muzero_amd.selfplay import SelfPlayEngine
    G, N, M, total = 32, 4, 10, 100
    config = cartpole.make_config(num_simulations=N)
    net = config.network_config.network_class(config, seed=1)
    hist = SelfPlayEngine(config, net, G, seed=2).play_continuous(total, max_moves=M, check_every=4, fused=fused)
    lengths = hist.lengths.cpu().numpy()
    assert hist.G == total and (lengths >= 1).all() and (lengths <= M).all()
    rew, acts = hist.rewards.cpu().numpy(), hist.actions.cpu().numpy()
    obs = hist.observations.cpu().numpy()
    for g in range(total):
        assert set(acts[:lengths[g], g].tolist()) <= {0, 1}
        assert (rew[:lengths[g], g] == 1.0).all() and (rew[lengths[g]:, g] == 0.0).all()
        assert (np.abs(obs[0, g]) <= 0.05).all()                 # every game starts from a fresh reset state
    np.testing.assert_allclose(hist.policies.cpu().numpy()[0].sum(-1), 1.0, atol=1e-6)


@pytest.mark.parametrize('game', ['tictactoe', 'cartpole'])
def test_fused_selfplay_calls_equal_the_separate_ones(game):
    """mz_selfplay_turnover / mz_selfplay_step_record against the calls they fuse (observe, record_observation, flush,
    reset_where, observe, record_observation, dirichlet_from_gamma; step, record_move) on the same random mid-game
    states, several rounds: environment state, both histories, serials, counters and the search inputs bit for bit."""
    import ctypes
    import torch
    from muzero_amd import _lib
    from muzero_amd.selfplay import BatchedHistory
    from muzero_amd.vector_env import CartPoleVectorEnv, TicTacToeVectorEnv
    lib = _lib.load()
    G, M, F = 300, 6 if game == 'tictactoe' else 5, 900
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(4)
    mk = (lambda: TicTacToeVectorEnv(G, dev)) if game == 'tictactoe' else (lambda: CartPoleVectorEnv(G, dev, seed=1))
    envs = [mk(), mk()]
    A, O = envs[0].action_space_size, envs[0].observation_size
    init = (torch.rand(G, 4, dtype=torch.float64, device=dev, generator=g) - 0.5) * 0.4
    for e in envs:
        e.reset(init) if game == 'cartpole' else e.reset()
    hists = [(BatchedHistory(M, G, (O,), A, dev), BatchedHistory(M, F, (O,), A, dev)) for _ in range(2)]
    i32 = dict(dtype=torch.int32, device=dev)
    book = [dict(serial=torch.arange(G, **i32), counters=torch.tensor([G, 0], dtype=torch.int64, device=dev),
                 dst=torch.empty(G, **i32), mask=torch.empty(G, **i32), noise=torch.zeros(G, A, dtype=torch.float64, device=dev),
                 temp=torch.zeros(G, dtype=torch.float64, device=dev)) for _ in range(2)]
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    freezing, t_cfg = 3, 0.7
    for rnd in range(14):
        gamma = torch._standard_gamma(torch.ones(G, A, dtype=torch.float64, device=dev), generator=g)
        restart = (torch.rand(G, 4, dtype=torch.float64, device=dev, generator=g) - 0.5) * 0.1
        # ---- separate calls
        e, (live, done), b = envs[0], hists[0], book[0]
        ls, ds = live.as_struct(), done.as_struct()
        obs, legal, n_legal = e.observe()
        _lib.check(lib.mz_history_record_observation(stream(), ctypes.byref(ls), e.steps.data_ptr(), obs.data_ptr(),
                                                     e.to_play.data_ptr()))
        _lib.check(lib.mz_history_flush(stream(), ctypes.byref(ls), ctypes.byref(ds), e.steps.data_ptr(), e.ended.data_ptr(),
                                        b['serial'].data_ptr(), b['counters'].data_ptr(), b['dst'].data_ptr(),
                                        b['mask'].data_ptr()))
        e.reset_where(b['mask'], restart) if game == 'cartpole' else e.reset_where(b['mask'])
        obs, legal, n_legal = e.observe()
        _lib.check(lib.mz_history_record_observation(stream(), ctypes.byref(ls), e.steps.data_ptr(), obs.data_ptr(),
                                                     e.to_play.data_ptr()))
        _lib.check(lib.mz_dirichlet_from_gamma(stream(), G, A, gamma.data_ptr(), n_legal.data_ptr(), b['noise'].data_ptr()))
        b['temp'].copy_(torch.where(e.steps < freezing, torch.tensor(t_cfg, dtype=torch.float64, device=dev),
                                    torch.tensor(0.0, dtype=torch.float64, device=dev)))
        # ---- fused
        e2, (live2, done2), b2 = envs[1], hists[1], book[1]
        ls2, ds2 = live2.as_struct(), done2.as_struct()
        es = e2.as_struct(restart)
        _lib.check(lib.mz_selfplay_turnover(stream(), ctypes.byref(es), ctypes.byref(ls2), ctypes.byref(ds2),
                                            b2['serial'].data_ptr(), b2['counters'].data_ptr(), b2['dst'].data_ptr(),
                                            b2['mask'].data_ptr(), gamma.data_ptr(), freezing, t_cfg,
                                            e2.observation.data_ptr(), e2.legal.data_ptr(), e2.n_legal.data_ptr(),
                                            b2['noise'].data_ptr(), b2['temp'].data_ptr()))
        for name in ('observation', 'legal', 'n_legal', 'steps', 'ended', 'to_play'):
            assert torch.equal(getattr(e, name), getattr(e2, name)), (rnd, name)
        for k in b:
            assert torch.equal(b[k], b2[k]), (rnd, k)
        # ---- a move: random legal actions (some slots get an illegal one in TicTacToe: status 2, nothing recorded)
        pick = torch.randint(0, A, (G,), generator=g, device=dev, dtype=torch.int32)
        action = torch.where(n_legal > 0, legal.gather(1, (pick % n_legal.clamp(min=1)).long().unsqueeze(1)).squeeze(1), pick)
        if game == 'tictactoe':
            action[::17] = pick[::17]
        action = action.to(torch.int32).contiguous()
        root_value = torch.rand(G, dtype=torch.float64, device=dev, generator=g)
        policy = torch.rand(G, A, device=dev, generator=g)
        reward = e.step(action)
        _lib.check(lib.mz_history_record_move(stream(), ctypes.byref(ls), e.steps.data_ptr(), e.status.data_ptr(),
                                              action.data_ptr(), reward.data_ptr(), root_value.data_ptr(), policy.data_ptr()))
        _lib.check(lib.mz_selfplay_step_record(stream(), ctypes.byref(es), ctypes.byref(ls2), action.data_ptr(),
                                               root_value.data_ptr(), policy.data_ptr(), e2.reward.data_ptr(),
                                               e2.status.data_ptr(), e2._illegal.data_ptr()))
        for name in ('reward', 'status', 'steps', 'ended', 'to_play', '_illegal'):
            assert torch.equal(getattr(e, name), getattr(e2, name)), (rnd, name)
        state = 'board' if game == 'tictactoe' else 'state'
        assert torch.equal(getattr(e, state), getattr(e2, state)), rnd
        for h, h2 in zip(hists[0], hists[1]):
            for f in h.FIELDS:
                assert torch.equal(getattr(h, f), getattr(h2, f)), (rnd, f)
    assert int(book[0]['counters'][1].item()) > G          # several waves of games were flushed
    envs[0]._illegal.zero_(); envs[1]._illegal.zero_()


def test_selfplay_engine_cartpole_short_episodes():
    import torch
    from muzero_amd.games import cartpole
    from muzero_amd.selfplay import SelfPlayEngine
    G, N, M = 40, 5, 12
    config = cartpole.make_config(num_simulations=N)
    net = config.network_config.network_class(config, seed=1)
    engine = SelfPlayEngine(config, net, G, seed=2)
    hist = engine.play_games(max_moves=M)
    lengths = hist.lengths.cpu().numpy()
    assert (lengths >= 1).all() and (lengths <= M).all()
    acts = hist.actions.cpu().numpy()
    rew = hist.rewards.cpu().numpy()
    for g in range(G):
        assert set(acts[:lengths[g], g].tolist()) <= {0, 1}
        assert (rew[:lengths[g], g] == 1.0).all() and (rew[lengths[g]:, g] == 0.0).all()
    pol = hist.policies.cpu().numpy()
    np.testing.assert_allclose(pol[0].sum(-1), 1.0, atol=1e-6)


def test_selfplay_engine_hot_reloads_published_weights(tmp_path):
    from muzero_amd.games import tictactoe
    from muzero_amd.selfplay import SelfPlayEngine
    config = tictactoe.make_config(num_simulations=4)
    cls = config.network_config.network_class
    net = cls(config, seed=1)
    models = str(tmp_path / 'models')
    engine = SelfPlayEngine(config, net, 32, seed=4, lanes=2, models_path=models)
    assert engine.play_games().metadata['network_id'] == '0'
    cls(config, seed=2, training_steps=250).save_version(models)
    hist = engine.play_games()
    assert hist.metadata['network_id'] == '250' and net.training_steps() == 250
    assert all(n.training_steps() == 250 for n in engine.search.networks)
    import numpy as np
    np.testing.assert_array_equal(engine.search.networks[1].get_weights(), cls(config, seed=2).get_weights())


def test_device_tictactoe_env_matches_every_reference_transition():
    """mz_env_ttt_step / mz_env_ttt_observe against the UNMODIFIED reference environment (games/tictactoe.py:71-148):
    every reachable position x every legal move (tests/golden/ttt_env_transitions.npz, made by make_golden_env.py from
    the reference itself), one game slot per transition, stepped directly through the C-ABI -- no host class in between."""
    import os
    import torch
    from muzero_amd.exceptions import MuZeroEnvironmentError
    from muzero_amd.vector_env import TicTacToeVectorEnv
    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ttt_env_transitions.npz')))
    K = len(fx['action'])
    venv = TicTacToeVectorEnv(K)
    venv.reset()
    obs, legal, n_legal = venv.observe()
    np.testing.assert_array_equal(obs.cpu().numpy(), np.broadcast_to(fx['reset_observation'], (K, 27)))
    np.testing.assert_array_equal(legal.cpu().numpy(), np.broadcast_to(fx['reset_legal'], (K, 9)))
    dev = venv.device
    venv.board.copy_(torch.from_numpy(fx['board']).to(dev))
    venv.to_play.copy_(torch.from_numpy(fx['to_play']).to(dev))
    venv.steps.copy_(torch.from_numpy(fx['steps']).to(dev))
    venv.ended.zero_()
    reward = venv.step(torch.from_numpy(fx['action']).to(dev)).cpu().numpy().copy()
    venv.check_illegal()
    assert (venv.status.cpu().numpy() == 0).all()
    np.testing.assert_array_equal(reward, fx['reward'])
    np.testing.assert_array_equal(venv.board.cpu().numpy(), fx['next_board'])
    np.testing.assert_array_equal(venv.to_play.cpu().numpy(), fx['next_to_play'])
    np.testing.assert_array_equal(venv.steps.cpu().numpy(), fx['next_steps'])
    np.testing.assert_array_equal(venv.ended.cpu().numpy() != 0, fx['ended'] != 0)
    obs, legal, n_legal = (x.cpu().numpy() for x in venv.observe())
    np.testing.assert_array_equal(obs, fx['next_observation'])
    np.testing.assert_array_equal(n_legal, fx['next_n_legal'])
    np.testing.assert_array_equal(legal, fx['next_legal'])
    # the moves the reference refuses (MuZeroEnvironmentError, games/tictactoe.py:122-123): status 2, state untouched
    M = len(fx['illegal_action'])
    venv = TicTacToeVectorEnv(M)
    venv.reset()
    venv.board.copy_(torch.from_numpy(fx['illegal_board']).to(dev))
    venv.to_play.copy_(torch.from_numpy(fx['illegal_to_play']).to(dev))
    venv.steps.copy_(torch.from_numpy((fx['illegal_board'] >= 0).sum(axis=1).astype(np.int32)).to(dev))
    venv.step(torch.from_numpy(fx['illegal_action']).to(dev))
    assert (venv.status.cpu().numpy() == 2).all()
    np.testing.assert_array_equal(venv.board.cpu().numpy(), fx['illegal_board'])
    with pytest.raises(MuZeroEnvironmentError):
        venv.check_illegal()
