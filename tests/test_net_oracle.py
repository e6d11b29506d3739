"""The numpy network oracle: the reference's one known-answer test plus cross-checks against torch's own
conv/linear (an independent implementation of the same Keras-documented semantics)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import net_oracle as no


def test_tensor_product_encoder_known_answer():
    """network.py:77-85 (the only test the reference ships): 4 states k*I2, actions [0,2,1,2], A=3."""
    states = np.stack([np.eye(2), 2 * np.eye(2), 3 * np.eye(2), 4 * np.eye(2)]).astype(np.float32)
    actions = np.array([0, 2, 1, 2])
    result = no.tensor_product_encoder(states, actions, 3)
    i, z = np.eye(2, dtype=np.float32), np.zeros((2, 2), np.float32)
    answer = np.array([[i, z, z], [z, z, 2 * i], [z, 3 * i, z], [z, z, 4 * i]], np.float32)
    assert np.linalg.norm(answer - result) < 1e-7


def test_encoders():
    h = np.arange(2 * 3 * 3 * 4, dtype=np.float64).reshape(2, 3, 3, 4)
    enc = no.one_hot_tensor_encoder(h, [1, 2], 3)
    assert enc.shape == (2, 3, 3, 7)
    np.testing.assert_array_equal(enc[..., :4], h)
    np.testing.assert_array_equal(enc[0, :, :, 4:], np.broadcast_to([0, 1, 0], (3, 3, 3)))
    np.testing.assert_array_equal(enc[1, :, :, 4:], np.broadcast_to([0, 0, 1], (3, 3, 3)))
    h = np.ones((3, 1, 5))
    enc = no.binary_plane_encoder(h, [0, 1, 1], axis=0)
    assert enc.shape == (3, 2, 5)
    np.testing.assert_array_equal(enc[:, 1, :], [[0] * 5, [1] * 5, [1] * 5])


def test_scalar_support_round_trip():
    x = np.linspace(0.0, 1.0, 37)
    s = no.scalar_to_support(x, 100)
    assert s.shape == (37, 101)
    np.testing.assert_allclose(s.sum(-1), 1.0, atol=1e-12)
    assert ((s > 0).sum(-1) <= 2).all()
    np.testing.assert_allclose(no.support_to_scalar(s, 100), x, atol=2e-3)   # piecewise-linear interpolation error
    np.testing.assert_allclose(no.support_to_scalar(no.scalar_to_support(np.array([0.0, 1.0]), 100), 100), [0, 1],
                               atol=1e-12)


def test_conv_and_dense_against_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 3, 3, 7))
    k = rng.standard_normal((3, 3, 7, 6))
    b = rng.standard_normal(6)
    tx = torch.from_numpy(x).permute(0, 3, 1, 2)
    tk = torch.from_numpy(k).permute(3, 2, 0, 1)
    for padding, pad in (('same', 1), ('valid', 0)):
        want = F.conv2d(tx, tk, torch.from_numpy(b), padding=pad).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(no.conv2d(x, k, b, padding), want, rtol=1e-12, atol=1e-12)
    x1 = rng.standard_normal((4, 4, 9))
    k1 = rng.standard_normal((4, 9, 5))
    want = F.conv1d(torch.from_numpy(x1).permute(0, 2, 1), torch.from_numpy(k1).permute(2, 1, 0)).permute(0, 2, 1).numpy()
    np.testing.assert_allclose(no.conv1d(x1, k1, np.zeros(5)), want, rtol=1e-12, atol=1e-12)


def test_network_shapes_and_parameter_counts():
    """SURVEY.md a18/a19 parameter counts."""
    assert no.num_weights('tictactoe') == (3584 + 16512) + (157952 + 16512 + 1153) + (147584 + 16512 + 129 + 1161)
    assert no.num_weights('cartpole') == (51840 + 16512) + (32896 + 16512 + 16512 + 129) + (2 * 16512 + 13029 + 258)
    for kind, obs in (('tictactoe', np.zeros((3, 27), np.float32)), ('cartpole', np.zeros((3, 4), np.float32))):
        net = no.make_oracle(kind, no.random_weights(kind, 1))
        h, v, logits, raw = net.initial_inference(obs)
        assert h.shape == (3, net.H) and v.shape == (3,) and logits.shape == (3, net.A)
        h2, r, v2, l2, raw2 = net.recurrent_inference(h, np.array([0, 1, net.A - 1]))
        assert h2.shape == (3, net.H) and r.shape == (3,) and v2.shape == (3,) and l2.shape == (3, net.A)
    net = no.make_oracle('cartpole', no.random_weights('cartpole', 1))
    _, v, _, raw = net.initial_inference(np.array([[0.1, -0.2, 0.05, 0.3]], np.float32))
    assert raw.shape == (1, 101) and abs(raw.sum() - 1) < 1e-12 and 0 <= v[0] <= 100


def test_product_encoders_match_the_reference_known_answer_and_the_oracle():
    """muzero_amd.network's tensor-op encoders: the reference's known-answer test (network.py:77-85) and the numpy oracle
    on random inputs (CPU tensors here; the ops are device-agnostic)."""
    from muzero_amd import network as mn
    states = torch.stack([k * torch.eye(2) for k in (1, 2, 3, 4)])
    actions = torch.tensor([0, 2, 1, 2])
    i, z = torch.eye(2), torch.zeros(2, 2)
    answer = torch.stack([torch.stack(r) for r in ([i, z, z], [z, z, 2 * i], [z, 3 * i, z], [z, z, 4 * i])])
    assert torch.norm(answer - mn.tensor_product_encoder(states, actions, 3)) < 1e-7
    rng = np.random.default_rng(1)
    h = rng.standard_normal((5, 3, 3, 4)).astype(np.float32)
    a = rng.integers(0, 6, size=5)
    np.testing.assert_array_equal(mn.one_hot_tensor_encoder(torch.from_numpy(h), torch.from_numpy(a), 6).numpy(),
                                  no.one_hot_tensor_encoder(h, a, 6))
    np.testing.assert_array_equal(mn.tensor_product_encoder(torch.from_numpy(h), torch.from_numpy(a), 6).numpy(),
                                  no.tensor_product_encoder(h, a, 6))
    b = rng.integers(0, 2, size=5)
    for axis in (0, 1, 2):
        np.testing.assert_array_equal(mn.binary_plane_encoder(torch.from_numpy(h), torch.from_numpy(b), axis).numpy(),
                                      no.binary_plane_encoder(h, b, axis=axis))
