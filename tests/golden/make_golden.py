#!/usr/bin/env python3
"""Generate the golden MCTS fixtures under tests/golden/ from the UNMODIFIED reference agent.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What it does:
  * TensorFlow / gym / tensorboard / tensorflow_serving are not installed, so permissive stub modules
    are injected into ``sys.modules`` and the reference's own ``agent.py``, ``network.py`` (only the
    ``NetworkOutput`` / ``BatchNetworkOutput`` classes are exercised), ``utils.py``, ``config.py``,
    ``game.py`` and ``games/tictactoe.py`` are imported from /root/reference as they are.
  * A fake network hands the reference agent pre-generated (value, reward, logits) per call and records
    which (parent expansion index, action) every ``recurrent_inference`` call was made for.
  * ``np.random.dirichlet`` is wrapped so the drawn root noise is recorded; ``random.seed`` is fixed and
    the uniform draw that ``random.choices`` consumes is recorded.
  * Era semantics (trap T3): the reference was written for NumPy 1.x where ``np.float32 * python-float``
    promotes to float64.  Under NumPy 2 the same code would compute non-root scores in float32.  The
    harness therefore gives the agent a ``NetworkOutput`` subclass whose ``masked_policy`` returns the
    reference's own float32 result widened to float64 (values unchanged) so every score is float64, as
    it was for the reference's authors.
  * Outputs are small ``.npz`` files holding inputs and expected outputs only (no reference source).

The fixtures are the pin for oracle/mcts_oracle.py, oracle/mcts_oracle.c and the HIP tree kernels.
"""
import os
import random
import sys
import types

import numpy as np

REFERENCE = '/root/reference'
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------------
# Stub modules (harness code; nothing from the reference is edited)
# ----------------------------------------------------------------------------------------------------
class _Anything:
    """Subclassable, callable, attribute-permissive dummy."""

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *args, **kwargs):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything


class _PermissiveModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        sub = _PermissiveModule(self.__name__ + '.' + name)
        sub.__class__ = _PermissiveModuleClassLike
        setattr(self, name, sub)
        return sub


class _PermissiveModuleClassLike(_PermissiveModule):
    """Attribute chains such as tf.keras.metrics.Metric must end in something subclassable."""

    def __call__(self, *args, **kwargs):
        return _Anything()

    def __mro_entries__(self, bases):
        return (_Anything,)


class FakeTensor(np.ndarray):
    def numpy(self):
        return np.asarray(self)


def _as_tensor(value, dtype=None):
    arr = np.asarray(value)
    if dtype is None:
        # tf.constant: python floats -> float32, python ints -> int32, numpy arrays keep their dtype
        if not isinstance(value, np.ndarray) and not isinstance(value, np.generic):
            if arr.dtype == np.float64:
                arr = arr.astype(np.float32)
            elif arr.dtype == np.int64:
                arr = arr.astype(np.int32)
    else:
        arr = arr.astype(dtype)
    return arr.view(FakeTensor)


def install_stubs():
    tf = _PermissiveModule('tensorflow')
    tf.Tensor = FakeTensor
    tf.float32 = np.float32
    tf.int32 = np.int32
    tf.constant = lambda value, dtype=None, **kw: _as_tensor(value, dtype)
    tf.expand_dims = lambda x, axis: np.expand_dims(np.asarray(x), axis).view(FakeTensor)
    tf.stack = lambda xs, axis=0: np.stack([np.asarray(x) for x in xs], axis=axis).view(FakeTensor)
    tf.where = lambda c, a, b: np.where(c, np.float32(a), np.float32(b)).astype(np.float32).view(FakeTensor)
    tf.ones = lambda shape, dtype=np.float32: np.ones(shape, dtype=dtype).view(FakeTensor)
    sys.modules['tensorflow'] = tf
    for name in ('tensorboard', 'tensorboard.plugins', 'tensorboard.plugins.hparams',
                 'tensorboard.plugins.hparams.api', 'tensorflow_serving', 'tensorflow_serving.apis',
                 'tensorflow_serving.apis.predict_pb2', 'tensorflow_serving.apis.prediction_service_pb2_grpc',
                 'gym'):
        sys.modules[name] = _PermissiveModule(name)
    sys.modules['tensorboard.plugins.hparams'].api = sys.modules['tensorboard.plugins.hparams.api']
    sys.modules['tensorflow_serving.apis'].predict_pb2 = sys.modules['tensorflow_serving.apis.predict_pb2']
    sys.modules['tensorflow_serving.apis'].prediction_service_pb2_grpc = \
        sys.modules['tensorflow_serving.apis.prediction_service_pb2_grpc']
    if not hasattr(np, 'int'):
        np.int = int  # games/tictactoe.py:144 uses the NumPy<1.24 alias
    sys.path.insert(0, REFERENCE)


install_stubs()

import agent as ref_agent            # noqa: E402  (reference, unmodified)
import network as ref_network        # noqa: E402
import config as ref_config          # noqa: E402
import utils as ref_utils            # noqa: E402
import game as ref_game              # noqa: E402
import environment as ref_environment  # noqa: E402


class EraNetworkOutput(ref_network.NetworkOutput):
    """Trap T3: widen the reference's float32 priors to float64 so NumPy 2 reproduces NumPy 1 scores."""
    recorded_priors = None

    def masked_policy(self, actions):
        p32 = super().masked_policy(actions)
        assert p32.dtype == np.float32
        if EraNetworkOutput.recorded_priors is not None:
            EraNetworkOutput.recorded_priors.append(p32.copy())
        return p32.astype(np.float64)


ref_network.NetworkOutput = EraNetworkOutput   # split_batch looks the class up by module global


class FakeNetwork:
    """Duck-typed network (agent.py:141): returns injected outputs, records the request trace."""

    def __init__(self, config, values, rewards, logits):
        self.config = config
        self.values = values      # [N+1] f32   (call 0 = initial inference)
        self.rewards = rewards    # [N+1] f32   (rewards[0] unused)
        self.logits = logits      # [N+1, A] f32
        self.calls = 0
        self.trace_parent = []
        self.trace_action = []

    def training_steps(self):
        return 0

    def _output(self, with_reward):
        i = self.calls
        self.calls += 1
        hidden = np.full((1, 1), float(i), dtype=np.float32).view(FakeTensor)   # tag = expansion index
        reward = self.rewards[i:i + 1].view(FakeTensor) if with_reward else None
        return ref_network.BatchNetworkOutput(
            batch_hidden_state=hidden, batch_reward=reward,
            batch_value=self.values[i:i + 1].view(FakeTensor),
            batch_policy_logits=self.logits[i:i + 1].view(FakeTensor),
            reward_to_scalar=self.config.reward_config.to_scalar,
            value_to_scalar=self.config.value_config.to_scalar)

    def initial_inference(self, observation):
        assert self.calls == 0
        return self._output(with_reward=False)

    def recurrent_inference(self, hidden_state, action):
        self.trace_parent.append(int(np.asarray(hidden_state).reshape(-1)[0]))
        self.trace_action.append(int(np.asarray(action).reshape(-1)[0]))
        return self._output(with_reward=True)


class FixedEnvironment(ref_environment.Environment):
    """One-state environment: exposes a fixed legal-action list to Game (game.py:48-61)."""

    def __init__(self, action_space_size, num_players, legal_actions):
        super().__init__(action_space_size=action_space_size, num_players=num_players)
        self._legal = list(legal_actions)

    def reset(self):
        from muzero_types import State
        return State(np.zeros((1,), np.float32).view(FakeTensor), 0, list(self._legal))


def make_config(A, num_players, discount, num_simulations, default_value, known_bounds,
                temperature=1.0, freezing_moves=10, alpha=1.0, frac=0.25):
    kb = None if known_bounds is None else ref_utils.KnownBounds(minv=known_bounds[0], maxv=known_bounds[1])
    return ref_config.MuZeroConfig(
        game_config=ref_config.GameConfig(name='synthetic', environment_class=None, environment_parameters={},
                                          action_space_size=A, num_players=num_players, discount=discount),
        replay_buffer_config=None, training_config=None,
        mcts_config=ref_config.MCTSConfig(max_moves=10, root_dirichlet_alpha=alpha, root_exploration_fraction=frac,
                                          num_simulations=num_simulations, temperature=temperature,
                                          freezing_moves=freezing_moves, default_value=default_value),
        network_config=None,
        reward_config=ref_config.ScalarConfig(known_bounds=None, support_size=None, loss_decay=0.0),
        value_config=ref_config.ScalarConfig(known_bounds=kb, support_size=None, loss_decay=0.0))


def dump_tree(root, A, N):
    """Breadth-less dump in EXPANSION order: block e = node expanded by network call e (hidden tag)."""
    blocks = {}

    def visit(node):
        if not node.expanded():
            return
        e = int(np.asarray(node.hidden_state).reshape(-1)[0])
        acts = list(node.children.keys())
        blocks[e] = dict(
            actions=np.array(acts, np.int32),
            count=np.array([node.children[a].visit_count for a in acts], np.int32),
            vsum=np.array([node.children[a].value_sum for a in acts], np.float64),
            prior=np.array([float(node.children[a].prior) for a in acts], np.float64),
            child=np.array([int(np.asarray(node.children[a].hidden_state).reshape(-1)[0])
                            if node.children[a].expanded() else -1 for a in acts], np.int32),
            reward=np.array([node.children[a].reward if node.children[a].reward is not None else np.nan
                             for a in acts], np.float64))
        for a in acts:
            visit(node.children[a])

    sys.setrecursionlimit(10000)
    visit(root)
    nb = N + 1
    out = dict(tree_nchild=np.zeros(nb, np.int32), tree_actions=np.full((nb, A), -1, np.int32),
               tree_count=np.zeros((nb, A), np.int32), tree_vsum=np.zeros((nb, A), np.float64),
               tree_prior=np.zeros((nb, A), np.float64), tree_child=np.full((nb, A), -1, np.int32),
               tree_reward=np.full((nb, A), np.nan, np.float64))
    for e, b in blocks.items():
        n = len(b['actions'])
        out['tree_nchild'][e] = n
        out['tree_actions'][e, :n] = b['actions']
        out['tree_count'][e, :n] = b['count']
        out['tree_vsum'][e, :n] = b['vsum']
        out['tree_prior'][e, :n] = b['prior']
        out['tree_child'][e, :n] = b['child']
        out['tree_reward'][e, :n] = b['reward']
    return out


def _run_case(name, A, N, num_players, discount, default_value, known_bounds, legal, seed,
             temperature=1.0, num_moves=0, freezing_moves=10, frac=0.25, alpha=1.0,
             value_mode='tanh', reward_mode='uniform', logit_scale=1.0, full_tree=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    logits = (rng.standard_normal((N + 1, A)) * logit_scale).astype(np.float32)
    if value_mode == 'tanh':
        values = np.tanh(rng.standard_normal(N + 1)).astype(np.float32)
    elif value_mode == 'const':
        values = np.full(N + 1, 0.5, np.float32)
    elif value_mode == 'cartpole':
        values = (rng.random(N + 1) * 100.0).astype(np.float32)
    else:
        raise ValueError(value_mode)
    if reward_mode == 'uniform':
        rewards = rng.random(N + 1).astype(np.float32)
    elif reward_mode == 'tanh':
        rewards = np.tanh(rng.standard_normal(N + 1)).astype(np.float32)
    elif reward_mode == 'zero':
        rewards = np.zeros(N + 1, np.float32)
    elif reward_mode == 'one':
        rewards = np.ones(N + 1, np.float32)
    else:
        raise ValueError(reward_mode)

    config = make_config(A, num_players, discount, N, default_value, known_bounds,
                         temperature=temperature, freezing_moves=freezing_moves, alpha=alpha, frac=frac)
    net = FakeNetwork(config, values, rewards, logits)
    mcts = ref_agent.MCTSAgent(config=config, network=net, agent_id='golden')

    drawn = {}
    real_dirichlet = np.random.dirichlet

    def recording_dirichlet(alpha_vec, size=None):
        noise = real_dirichlet(alpha_vec, size)
        drawn['noise'] = np.array(noise, np.float64)
        return noise

    real_choices = random.choices

    def recording_choices(population, weights=None, **kw):      # softmax_sample's weights, as the reference computed them
        drawn['choice_weights'] = np.array(weights, np.float64)
        return real_choices(population, weights=weights, **kw)

    np.random.seed(seed)
    np.random.dirichlet = recording_dirichlet
    random.choices = recording_choices
    EraNetworkOutput.recorded_priors = []
    try:
        # -- the body of MCTSAgent.make_move (agent.py:219-232), step by step so the root stays inspectable
        root = ref_agent.Node()
        mm = ref_utils.MinMaxStats(known_bounds=config.value_config.known_bounds)
        game = ref_game.Game(environment=FixedEnvironment(A, num_players, legal))
        observation = np.expand_dims(np.asarray(game.history.make_image()), 0).view(FakeTensor)
        mcts.expand_node(node=root, actions=game.legal_actions(),
                         network_output=net.initial_inference(observation).split_batch()[0])
        mcts.add_exploration_noise(root)
        mcts.run_mcts(root, mm)
        action_space = config.action_space()
        policy = [root.children[a].visit_count / root.visit_count if a in root.children else 0 for a in action_space]
        policy32 = np.asarray(_as_tensor(policy)).astype(np.float32)
        # select_action (agent.py:213-217): record the uniform that random.choices will consume
        for _ in range(num_moves):
            game.history.actions.append(0)
        random.seed(seed)
        u = random.random()
        random.seed(seed)
        action = mcts.select_action(root, len(game.history))
    finally:
        np.random.dirichlet = real_dirichlet
        random.choices = real_choices
        priors = EraNetworkOutput.recorded_priors
        EraNetworkOutput.recorded_priors = None

    assert len(priors) == N + 1 and net.calls == N + 1
    priors32 = np.stack(priors).astype(np.float32)        # [N+1, A]; row 0 = root masked policy (full length A)
    noise = np.zeros(A, np.float64)
    noise[:len(legal)] = drawn['noise']
    visits = np.zeros(A, np.int32)
    for a, child in root.children.items():
        visits[a] = child.visit_count
    out = dict(
        A=np.int32(A), N=np.int32(N), num_players=np.int32(num_players), discount=np.float64(discount),
        default_value=np.float64(default_value),
        has_bounds=np.int32(known_bounds is not None),
        known_min=np.float64(config.value_config.known_bounds.endpoints()[0] if known_bounds is not None else 0.0),
        known_max=np.float64(config.value_config.known_bounds.endpoints()[1] if known_bounds is not None else 0.0),
        frac=np.float64(frac), temperature=np.float64(temperature), num_moves=np.int32(num_moves),
        freezing_moves=np.int32(freezing_moves),
        legal=np.array(legal, np.int32), values=values, rewards=rewards, priors=priors32, noise=noise, u=np.float64(u),
        trace_parent=np.array(net.trace_parent, np.int32), trace_action=np.array(net.trace_action, np.int32),
        root_visits=visits, root_count=np.int32(root.visit_count), root_vsum=np.float64(root.value_sum),
        root_value=np.float64(root.value), mm_min=np.float64(mm.minimum), mm_max=np.float64(mm.maximum),
        policy=policy32, action=np.int32(action))
    if 'choice_weights' in drawn:         # count ** (1 / t) per root child in insertion order (agent.py:167)
        out['choice_weights'] = drawn['choice_weights']
    if A <= 32:
        out['logits'] = logits
    if full_tree:
        out.update(dump_tree(root, A, N))
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print(f'{name}: A={A} N={N} root N={root.visit_count} action={action} depth_max='
          f'{max_depth(root)} mm=({mm.minimum:.4g},{mm.maximum:.4g})')


def max_depth(node):
    best, stack = 0, [(node, 0)]
    while stack:
        n, d = stack.pop()
        best = max(best, d)
        for c in n.children.values():
            if c.expanded():
                stack.append((c, d + 1))
    return best


def _tictactoe_game(name, seed, N):
    """A full reference self-play game (agent.py:31-44) on the reference TicTacToeEnvironment."""
    import games.tictactoe as ttt
    config = ttt.make_config.__globals__['MuZeroConfig'](
        game_config=ref_config.GameConfig(name='TicTacToe', environment_class=ttt.TicTacToeEnvironment,
                                          environment_parameters={}, action_space_size=9, num_players=2, discount=1.0),
        replay_buffer_config=None, training_config=None,
        mcts_config=ref_config.MCTSConfig(max_moves=9, root_dirichlet_alpha=1.0, root_exploration_fraction=0.25,
                                          num_simulations=N, temperature=1.0, freezing_moves=9, default_value=0.0),
        network_config=None,
        reward_config=ref_config.ScalarConfig(known_bounds=ref_utils.KnownBounds(minv=0.0, maxv=1.0),
                                              support_size=None, loss_decay=0.0),
        value_config=ref_config.ScalarConfig(known_bounds=ref_utils.KnownBounds(minv=None, maxv=1.0),
                                             support_size=None, loss_decay=4.0))
    rng = np.random.Generator(np.random.PCG64(seed))
    max_calls = 9 * (N + 1)
    logits = rng.standard_normal((max_calls, 9)).astype(np.float32)
    values = np.tanh(rng.standard_normal(max_calls)).astype(np.float32)
    rewards = np.tanh(rng.standard_normal(max_calls)).astype(np.float32)

    class StreamNetwork(FakeNetwork):
        def initial_inference(self, observation):
            self.obs.append(np.asarray(observation)[0].copy())
            return self._output(with_reward=False)

    net = StreamNetwork(config, values, rewards, logits)
    net.obs = []
    mcts = ref_agent.MCTSAgent(config=config, network=net, agent_id='golden')
    noises, us, priors = [], [], []
    real_dirichlet = np.random.dirichlet

    def recording_dirichlet(alpha_vec, size=None):
        noise = real_dirichlet(alpha_vec, size)
        full = np.zeros(9, np.float64)
        full[:len(noise)] = noise
        noises.append(full)
        return noise

    # random.choices consumes exactly one random.random() per move; pre-draw the stream and replay it
    random.seed(seed)
    state = random.getstate()
    us = [random.random() for _ in range(9)]
    random.setstate(state)
    np.random.seed(seed)
    np.random.dirichlet = recording_dirichlet
    EraNetworkOutput.recorded_priors = priors
    try:
        history = mcts.play_game(ttt.TicTacToeEnvironment())
    finally:
        np.random.dirichlet = real_dirichlet
        EraNetworkOutput.recorded_priors = None
    moves = len(history)
    calls = net.calls
    np.savez_compressed(
        os.path.join(OUT_DIR, name + '.npz'),
        N=np.int32(N), moves=np.int32(moves), calls=np.int32(calls),
        values=values[:calls], rewards=rewards[:calls], logits=logits[:calls],
        priors=np.stack(priors).astype(np.float32), noise=np.stack(noises), u=np.array(us[:moves], np.float64),
        observations=np.stack([np.asarray(o, np.float32) for o in history.observations]),
        initial_observations=np.stack(net.obs).astype(np.float32),
        to_plays=np.array(history.to_plays, np.int32), actions=np.array(history.actions, np.int32),
        env_rewards=np.array(history.rewards, np.float64), root_values=np.array(history.root_values, np.float64),
        policies=np.stack([np.asarray(p, np.float32) for p in history.policies]))
    print(f'{name}: {moves} moves, actions={list(history.actions)}, rewards={list(history.rewards)}')


def main():
    only = set(sys.argv[1:])              # optional: generate only the named fixtures

    def run_case(name, **kw):
        if not only or name in only:
            _run_case(name, **kw)

    def tictactoe_game(name, **kw):
        if not only or name in only:
            _tictactoe_game(name, **kw)

    # TicTacToe-shaped (C1/C3): 2 players, known bounds (-1,1) via KnownBounds(None, 1.0), discount 1 -> -1
    run_case('ttt_full_s1', A=9, N=25, num_players=2, discount=1.0, default_value=0.0, known_bounds=(None, 1.0),
             legal=list(range(9)), seed=1, reward_mode='tanh')
    run_case('ttt_legal_s2', A=9, N=25, num_players=2, discount=1.0, default_value=0.0, known_bounds=(None, 1.0),
             legal=[1, 3, 4, 7], seed=2, reward_mode='tanh')          # trap T5: positional priors
    run_case('ttt_legal_s3', A=9, N=20, num_players=2, discount=1.0, default_value=0.0, known_bounds=(None, 1.0),
             legal=[0, 2, 8], seed=3, reward_mode='zero', temperature=0.0)
    run_case('ttt_onelegal_s4', A=9, N=12, num_players=2, discount=1.0, default_value=0.0, known_bounds=(None, 1.0),
             legal=[5], seed=4, reward_mode='tanh')
    run_case('ttt_frozen_s5', A=9, N=25, num_players=2, discount=1.0, default_value=0.0, known_bounds=(None, 1.0),
             legal=[0, 1, 2, 3, 5, 6, 8], seed=5, reward_mode='tanh', num_moves=9, freezing_moves=9)  # t -> 0
    # CartPole-shaped (C2): 1 player, bounds (0,100), default 50, discount 0.99
    run_case('cartpole_s6', A=2, N=50, num_players=1, discount=0.99, default_value=50.0, known_bounds=(0.0, 100.0),
             legal=[0, 1], seed=6, value_mode='cartpole', reward_mode='one')
    run_case('cartpole_s7', A=2, N=4, num_players=1, discount=0.99, default_value=50.0, known_bounds=(0.0, 100.0),
             legal=[0, 1], seed=7, value_mode='cartpole', reward_mode='uniform')   # reference default N=4
    # Atari-shaped (C4): unknown bounds -> live MinMaxStats (traps T2)
    run_case('atari_s8', A=18, N=50, num_players=1, discount=0.997, default_value=0.0, known_bounds=None,
             legal=list(range(18)), seed=8)
    run_case('atari_s9', A=18, N=50, num_players=1, discount=0.997, default_value=0.0, known_bounds=None,
             legal=[0, 3, 4, 9, 10, 17], seed=9, reward_mode='tanh')
    run_case('atari_constvalue_s10', A=18, N=30, num_players=1, discount=1.0, default_value=0.5, known_bounds=None,
             legal=list(range(18)), seed=10, value_mode='const', reward_mode='zero')   # max == min stays un-normalised
    run_case('atari_2p_unknown_s11', A=18, N=40, num_players=2, discount=0.9, default_value=0.0, known_bounds=None,
             legal=list(range(18)), seed=11, reward_mode='tanh', temperature=0.5)
    # ties: all logits equal, zero reward, constant value -> tie-break on the largest action (trap T1)
    run_case('ties_s12', A=9, N=25, num_players=1, discount=1.0, default_value=0.0, known_bounds=(0.0, 1.0),
             legal=list(range(9)), seed=12, value_mode='const', reward_mode='zero', logit_scale=0.0, frac=0.0)
    # Go-shaped (C5), reduced N so the fixture stays small
    run_case('go_s13', A=362, N=200, num_players=2, discount=1.0, default_value=0.0, known_bounds=(None, 1.0),
             legal=list(range(362)), seed=13, reward_mode='zero', full_tree=False)
    run_case('go_legal_s14', A=362, N=120, num_players=2, discount=1.0, default_value=0.0, known_bounds=(None, 1.0),
             legal=list(range(0, 362, 3)), seed=14, reward_mode='zero', full_tree=False, temperature=0.25)
    # symmetric bounds given explicitly, discount<1, asymmetric range
    run_case('asym_bounds_s15', A=6, N=64, num_players=1, discount=0.5, default_value=-1.0, known_bounds=(-2.0, 3.0),
             legal=[5, 0, 3], seed=15, reward_mode='tanh')   # non-monotone legal list: insertion order != action order
    # temperatures whose 1/t is not a small integer: count ** (1/t) goes through the host libm's pow (agent.py:167)
    run_case('temp07_s16', A=9, N=25, num_players=2, discount=1.0, default_value=0.0, known_bounds=(None, 1.0),
             legal=[0, 1, 2, 4, 5, 7, 8], seed=16, reward_mode='tanh', temperature=0.7)
    run_case('temp3_s17', A=18, N=50, num_players=1, discount=0.997, default_value=0.0, known_bounds=None,
             legal=list(range(18)), seed=17, temperature=3.0)
    tictactoe_game('ttt_game_s21', seed=21, N=25)
    tictactoe_game('ttt_game_s22', seed=22, N=20)


if __name__ == '__main__':
    main()
