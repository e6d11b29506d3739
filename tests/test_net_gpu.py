"""Fused HIP inference kernels (through the C ABI) against the fp64 numpy oracle.

Tolerance (BASELINE.json north_star): 1e-5 relative for fp32.  Stated here as rtol=1e-5 plus atol=1e-5 x the
head's output scale (1 for tanh/sigmoid/logit heads and hidden activations, 100 for CartPole's value range).
The oracle itself is "parity unpinned" against TensorFlow (oracle/net_oracle.py header)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import net_oracle as no

RTOL, ATOL = 1e-5, 1e-5


def make(kind, seed=11):
    from muzero_amd.network import AtariShapedNetwork, CartPoleNetwork, TicTacToeNetwork
    flat = no.random_weights(kind, seed)
    cls = {'tictactoe': TicTacToeNetwork, 'cartpole': CartPoleNetwork, 'atari': AtariShapedNetwork}[kind]
    return cls(config=None, weights=flat), no.make_oracle(kind, flat)


def observations(kind, B, rng):
    if kind == 'tictactoe':
        board = rng.integers(-1, 2, size=(B, 3, 3))
        to_play = rng.integers(0, 2, size=(B, 1, 1))
        obs = np.stack([(board == 0), (board == 1), np.broadcast_to(to_play, (B, 3, 3))], -1).astype(np.float32)
        return obs.reshape(B, 27)
    if kind == 'atari':
        return rng.standard_normal((B, 256)).astype(np.float32)
    span = np.array([2.4, 3.0, 0.21, 3.0], np.float32)
    return (rng.uniform(-1, 1, size=(B, 4)) * span).astype(np.float32)


def close(got, want, scale=1.0):
    np.testing.assert_allclose(got.detach().cpu().numpy().reshape(np.shape(want)), want, rtol=RTOL, atol=ATOL * scale)


@pytest.mark.parametrize('B', [1, 63, 64, 65, 200])
@pytest.mark.parametrize('kind', ['tictactoe', 'cartpole', 'atari'])
def test_initial_and_recurrent_inference(kind, B):
    net, ref = make(kind)
    rng = np.random.default_rng(B)
    vscale = 100.0 if kind == 'cartpole' else 1.0
    obs = observations(kind, B, rng)
    out = net.initial_inference(obs)
    h, v, logits, raw = ref.initial_inference(obs)
    close(out.batch_hidden_state, h)
    close(out.batch_value_scalar, v, vscale)
    close(out.batch_policy_logits, logits)
    close(out.batch_value, raw)
    assert out.batch_reward is None
    # value_to_scalar(batch_value) on the host gives the same scalars as the fused epilogue
    action = rng.integers(0, ref.A, size=B)
    out2 = net.recurrent_inference(out.batch_hidden_state, action)
    h2, r2, v2, l2, raw2 = ref.recurrent_inference(h, action)
    close(out2.batch_hidden_state, h2)
    close(out2.batch_reward_scalar, r2)
    close(out2.batch_value_scalar, v2, vscale)
    close(out2.batch_policy_logits, l2)
    close(out2.batch_value, raw2)
    # a second unrolled step, from the kernel's own hidden state
    action3 = rng.integers(0, ref.A, size=B)
    out3 = net.recurrent_inference(out2.batch_hidden_state, action3)
    h3, r3, v3, l3, _ = ref.recurrent_inference(h2, action3)
    close(out3.batch_hidden_state, h3)
    close(out3.batch_reward_scalar, r3)
    close(out3.batch_value_scalar, v3, vscale)
    close(out3.batch_policy_logits, l3)
    outs = out2.split_batch()
    assert len(outs) == B and isinstance(outs[0].value, float) and isinstance(outs[0].reward, float)
    assert outs[B - 1].value == float(out2.batch_value_scalar[B - 1])


@pytest.mark.parametrize('B', [1, 63, 65, 200, 2600])
def test_tictactoe_initial_inference_of_the_64_game_jobs(B):
    """Batches beyond 2560 games take ttt_initial_kernel_w (64 games x a group of cells per workgroup, the 3 -> 128 conv
    as one MFMA k-step over the observation row); set_job_rows(64) selects it for any batch.  Ragged batches, general
    (non 0/1) observation values, and the recurrent step that reads the hidden rows it wrote."""
    net, ref = make('tictactoe')
    rng = np.random.default_rng(1000 + B)
    if B <= 2560:
        net.set_job_rows(64)
    obs = observations('tictactoe', B, rng)
    if B == 65:      # the planes are 0/1 in the game; the kernel splits any f32 value
        obs = (obs + rng.standard_normal(obs.shape) * 0.37).astype(np.float32)
    out = net.initial_inference(obs)
    h, v, logits, raw = ref.initial_inference(obs)
    close(out.batch_hidden_state, h)
    close(out.batch_value_scalar, v)
    close(out.batch_policy_logits, logits)
    action = rng.integers(0, 9, size=B)
    out2 = net.recurrent_inference(out.batch_hidden_state, action)
    h2, r2, v2, l2, _ = ref.recurrent_inference(h, action)
    close(out2.batch_hidden_state, h2)
    close(out2.batch_reward_scalar, r2)
    close(out2.batch_policy_logits, l2)
    net.set_job_rows(0)


def _scaled_close(got, want, what):
    """1e-5 relative to the tensor's scale: rtol 1e-5 plus atol 1e-5 x max|want|."""
    want = np.asarray(want)
    np.testing.assert_allclose(got.detach().cpu().numpy().reshape(want.shape), want, rtol=RTOL,
                               atol=ATOL * max(float(np.abs(want).max()), 1e-30), err_msg=what)


@pytest.mark.parametrize('scale', [8.0, 1.0 / 64.0])
@pytest.mark.parametrize('kind', ['tictactoe', 'cartpole', 'atari'])
def test_split_f16_numeric_domain(kind, scale):
    """The kernels carry every f32 operand as two f16 halfs (include/mzhip.h, mz_net_create: |v| <= 65504).  With every
    weight scaled by 8 the activations grow ~64x per layer pair, with 1/64 they shrink towards f16's subnormals: inside
    the domain the outputs stay within 1e-5 of the fp64 oracle relative to each tensor's scale and the range flag stays
    down; once an activation leaves the domain the handle says so (mz_net_status) instead of returning clamped numbers
    silently."""
    from muzero_amd.network import AtariShapedNetwork, CartPoleNetwork, TicTacToeNetwork
    cls = {'tictactoe': TicTacToeNetwork, 'cartpole': CartPoleNetwork, 'atari': AtariShapedNetwork}[kind]
    flat = (no.random_weights(kind, 23) * np.float32(scale)).astype(np.float32)
    net, ref = cls(config=None, weights=flat), no.make_oracle(kind, flat)
    rng = np.random.default_rng(4)
    B = 150
    obs = observations(kind, B, rng)
    out = net.initial_inference(obs)
    h, v, logits, raw = ref.initial_inference(obs)
    if not (kind == 'cartpole' and scale > 1):
        # (CartPole's two-hot preprocessing runs in f32 like the reference's TF graph, utils.py:138-143: the bin weights
        # carry ~1e-5 of absolute error, which the x8 layers amplify 64-fold -- not a property of the split arithmetic)
        _scaled_close(out.batch_hidden_state, h, 'initial hidden')
        _scaled_close(out.batch_policy_logits, logits, 'initial logits')
        _scaled_close(out.batch_value_scalar, v, 'initial value')
    assert net.status() == 0
    saturated = False
    hidden_in = h.astype(np.float32)                     # the recurrent steps start from the oracle's hidden state
    h = hidden_in.astype(np.float64)
    for step in range(3):
        action = rng.integers(0, ref.A, size=B)
        out = net.recurrent_inference(hidden_in, action)
        h, r, v, logits, _ = ref.recurrent_inference(h, action)
        flags = net.status()
        hmax = float(np.abs(h).max())
        if hmax > 2.0e5:                                  # clearly outside: the flag must be up
            assert flags & 1, f'step {step}: activations up to {hmax:.3g} went unreported'
        if flags & 1:
            # raised by some layer's output (the conv feeding the dense is ~8x the hidden state it produces): never
            # inside a comfortably small network
            assert scale > 1 and hmax > 1.0e3, f'step {step}: range flag raised with max |hidden| = {hmax:.3g}'
            saturated = True
            break
        _scaled_close(out.batch_hidden_state, h, f'hidden, step {step}')
        _scaled_close(out.batch_reward_scalar, r, f'reward, step {step}')
        _scaled_close(out.batch_policy_logits, logits, f'logits, step {step}')
        _scaled_close(out.batch_value_scalar, v, f'value, step {step}')
        hidden_in = out.batch_hidden_state.cpu().numpy().reshape(h.shape)              # continue from the kernel's state
        h = hidden_in.astype(np.float64)
    if scale > 1 and kind != 'cartpole':
        assert saturated, 'the x8 network was expected to leave the domain within three unrolled steps'
    with pytest.raises(Exception):
        net2 = cls(config=None, weights=flat)
        big = np.full((B, ref.H), 1.0e6, np.float32)
        net2.recurrent_inference(big, rng.integers(0, ref.A, size=B))
        net2.check_range()


def test_weights_outside_the_domain_are_rejected():
    from muzero_amd.exceptions import MuZeroError
    from muzero_amd.network import TicTacToeNetwork
    flat = no.random_weights('tictactoe', 1)
    for bad in (np.inf, np.nan, 7.0e4, -1.0e5):
        w = flat.copy()
        w[12345] = bad
        with pytest.raises(MuZeroError):
            TicTacToeNetwork(config=None, weights=w)
    w = flat.copy()
    w[12345] = 6.0e4                                        # inside the range
    TicTacToeNetwork(config=None, weights=w)


def test_cartpole_large_batch_takes_the_row_tile_kernel():
    """Batches up to 8192 games run 16 per workgroup (cartpole16_kernel); beyond that the 32/64-row tiles do."""
    net, ref = make('cartpole')
    B = 8300
    rng = np.random.default_rng(5)
    obs = observations('cartpole', B, rng)
    out = net.initial_inference(obs)
    h, v, logits, raw = ref.initial_inference(obs)
    close(out.batch_hidden_state, h)
    close(out.batch_value_scalar, v, 100.0)
    close(out.batch_policy_logits, logits)
    action = rng.integers(0, 2, size=B)
    out2 = net.recurrent_inference(out.batch_hidden_state, action)
    h2, r2, v2, l2, raw2 = ref.recurrent_inference(h, action)
    close(out2.batch_hidden_state, h2)
    close(out2.batch_reward_scalar, r2)
    close(out2.batch_value_scalar, v2, 100.0)
    close(out2.batch_policy_logits, l2)
    close(out2.batch_value, raw2)


@pytest.mark.parametrize('kind', ['tictactoe', 'cartpole', 'atari'])
def test_pool_gather_path_and_inactive_rows(kind):
    """recurrent_inference reading leaf.parent.hidden_state rows by parent block from the [N+1][T][H] tensor and
    writing block 3 in place; rows with parent_block < 0 are left untouched."""
    import torch
    net, ref = make(kind, seed=5)
    T, H, A = 150, ref.H, ref.A
    rng = np.random.default_rng(3)
    pool_np = np.abs(rng.standard_normal((4, T, H))).astype(np.float32)
    pool_np[3] = -7.0                                      # sentinel in the output block
    parent = rng.integers(-1, 3, size=T).astype(np.int32)
    action = rng.integers(0, A, size=T).astype(np.int32)
    pool = torch.from_numpy(pool_np).cuda()
    reward = torch.full((T,), -7.0, device='cuda')
    value = torch.full((T,), -7.0, device='cuda')
    logits = torch.full((T, A), -7.0, device='cuda')
    net.recurrent_inference_pool(pool, torch.from_numpy(parent).cuda(), torch.from_numpy(action).cuda(), 3, reward,
                                 value, logits)
    live = parent >= 0
    gathered = pool_np[np.maximum(parent, 0), np.arange(T)]
    h2, r2, v2, l2, _ = ref.recurrent_inference(gathered[live], action[live])
    got = pool[3].cpu().numpy()
    np.testing.assert_allclose(got[live], h2, rtol=RTOL, atol=ATOL)
    assert (got[~live] == -7.0).all()
    np.testing.assert_allclose(reward.cpu().numpy()[live], r2, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(value.cpu().numpy()[live], v2, rtol=RTOL, atol=ATOL * (100 if kind == 'cartpole' else 1))
    np.testing.assert_allclose(logits.cpu().numpy()[live], l2, rtol=RTOL, atol=ATOL)
    assert (reward.cpu().numpy()[~live] == -7.0).all() and (logits.cpu().numpy()[~live] == -7.0).all()
    np.testing.assert_array_equal(pool[:3].cpu().numpy(), pool_np[:3])          # inputs untouched


def test_host_to_scalar_matches_fused_epilogue():
    """ScalarConfig.to_scalar (config.py:25-40) applied on the host to the raw 101-bin head equals the fused one."""
    from muzero_amd.config import ScalarConfig
    from muzero_amd.utils import KnownBounds
    net, _ = make('cartpole')
    out = net.initial_inference(observations('cartpole', 40, np.random.default_rng(0)))
    cfg = ScalarConfig(known_bounds=KnownBounds(0.0, 100.0), support_size=100)
    np.testing.assert_allclose(cfg.to_scalar(out.batch_value).cpu().numpy(), out.batch_value_scalar.cpu().numpy(),
                               rtol=1e-5, atol=1e-3)


def test_weight_count_and_checkpoint_round_trip(tmp_path):
    from muzero_amd.exceptions import MuZeroError
    from muzero_amd.network import AtariShapedNetwork, CartPoleNetwork, TicTacToeNetwork
    assert TicTacToeNetwork.num_weights() == no.num_weights('tictactoe')
    assert CartPoleNetwork.num_weights() == no.num_weights('cartpole')
    assert AtariShapedNetwork.num_weights() == no.num_weights('atari')
    net, _ = make('tictactoe')
    obs = observations('tictactoe', 8, np.random.default_rng(1))
    before = net.initial_inference(obs).batch_policy_logits.cpu().numpy()
    net.steps = 1234
    net.save(str(tmp_path / 'ckpt'))
    other = TicTacToeNetwork(config=None, seed=99)
    assert other.training_steps() == 0
    other.load(str(tmp_path / 'ckpt'))
    assert other.training_steps() == 1234
    np.testing.assert_array_equal(other.initial_inference(obs).batch_policy_logits.cpu().numpy(), before)
    with pytest.raises(MuZeroError):
        TicTacToeNetwork(config=None, weights=np.zeros(10, np.float32))


def test_versioned_weight_handoff(tmp_path):
    """save_version / load_latest: the versioned-directory hand-off of network.py:262-266 + training.py:62-63; the
    reader picks the highest complete version newer than its own and its outputs change with it."""
    import torch
    from muzero_amd.games import cartpole
    from oracle import net_oracle
    config = cartpole.make_config(num_simulations=4)
    cls = config.network_config.network_class
    trainer = cls(config, weights=net_oracle.random_weights('cartpole', 1), training_steps=0)
    actor = cls(config, weights=net_oracle.random_weights('cartpole', 1), training_steps=0)
    obs = torch.tensor([[0.01, -0.02, 0.03, 0.04]], device='cuda')
    before = actor.initial_inference(obs).batch_policy_logits.clone()
    models = str(tmp_path / 'models')
    assert cls.available_versions(models) == [] and not actor.load_latest(models)
    trainer.save_version(models)                                        # version 0: not newer
    assert not actor.load_latest(models)
    trainer.set_weights(net_oracle.random_weights('cartpole', 2), training_steps=500)
    trainer.save_version(models)
    trainer.set_weights(net_oracle.random_weights('cartpole', 3), training_steps=1000)
    trainer.save_version(models)
    (tmp_path / 'models' / '2000').mkdir()                              # incomplete version: no weights file yet
    (tmp_path / 'models' / 'notes').mkdir()
    assert cls.available_versions(models) == [0, 500, 1000]
    assert actor.load_latest(models) and actor.training_steps() == 1000
    after = actor.initial_inference(obs).batch_policy_logits
    want = trainer.initial_inference(obs).batch_policy_logits
    assert torch.equal(after, want) and not torch.equal(after, before)
    assert not actor.load_latest(models)                                # already current
    other = cartpole.make_config().network_config.network_class
    from muzero_amd.games import tictactoe
    ttt = tictactoe.make_config().network_config.network_class(tictactoe.make_config(), seed=0)
    with pytest.raises(Exception):
        ttt.load_latest(models)
