"""The arithmetic of the split-f16 job kernel (muzero_amd/csrc/mz_net.hip, ttt_ring_kernel), modelled in NumPy and held
to the fp64 network oracle: every f32 operand v is carried as h = f16(v), l = f16((v - h) * 2048), a product is
ah*bh + (ah*bl + al*bh)/2048 accumulated in f32.  CPU only; the kernel itself is tested in test_net_gpu.py."""
import numpy as np

from oracle import net_oracle as no


def split(a):
    a = a.astype(np.float32)
    h = a.astype(np.float16)
    l = ((a - h.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32)


def split_matmul(x, k):
    xh, xl = split(x)
    kh, kl = split(k)
    return (xh @ kh + (xh @ kl + xl @ kh) * np.float32(1 / 2048)).astype(np.float32)


def test_split_reconstructs_to_21_bits():
    """h + l/2048 gives back v to 2^-21 relative; below the f16 normal range (|v| < ~1e-4) the error is absolute
    instead, under 1e-10 -- five orders below the tolerance the outputs are held to."""
    rng = np.random.default_rng(0)
    v = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-6, 2, 100000)).astype(np.float32)
    h, l = split(v)
    err = np.abs((h.astype(np.float64) + l.astype(np.float64) / 2048) - v)
    assert (err <= np.maximum(2.0 ** -21 * np.abs(v), 1e-10)).all()


def test_split_gemm_chain_stays_inside_the_f32_tolerance():
    """dynamics conv + Dense + prediction conv of the TicTacToe network (the three contractions of the job kernel), five
    unrolled steps, against the fp64 oracle: rtol = atol = 1e-5 is the kernel's bar; the model stays 10x inside it."""
    ref = no.make_oracle('tictactoe', no.random_weights('tictactoe', 7))
    apx = no.make_oracle('tictactoe', no.random_weights('tictactoe', 7), dtype=np.float32)

    def dense(x, k, b):
        return split_matmul(x.reshape(-1, x.shape[-1]), k).reshape(x.shape[:-1] + (k.shape[-1],)) + b.astype(np.float32)

    def conv2d(x, k, b, padding):
        kh, kw, cin, cout = k.shape
        if padding == 'same':
            x = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
        oh, ow = x.shape[1] - kh + 1, x.shape[2] - kw + 1
        out = np.zeros((x.shape[0], oh, ow, cout), np.float32)
        for i in range(kh):
            for j in range(kw):
                xs = x[:, i:i + oh, j:j + ow, :]
                out += split_matmul(xs.reshape(-1, cin), k[i, j]).reshape(xs.shape[:-1] + (cout,))
        return out + b.astype(np.float32)

    B = 64
    rng = np.random.default_rng(1)
    cells = rng.integers(0, 3, (B, 9))
    tp = rng.integers(0, 2, (B, 1)).repeat(9, 1)
    obs = np.stack([cells == 0, cells == 1, tp == 1], -1).astype(np.float32).reshape(B, 3, 3, 3)
    h = ref.initial_inference(obs)[0]
    ha = h.astype(np.float32)
    saved = (no.dense, no.conv2d)
    worst = 0.0
    for step in range(5):
        a = rng.integers(0, 9, B)
        want = ref.recurrent_inference(h, a)
        try:
            no.dense, no.conv2d = dense, conv2d
            got = apx.recurrent_inference(ha, a)
        finally:
            no.dense, no.conv2d = saved
        for x, y in zip(got[:4], want[:4]):
            worst = max(worst, float((np.abs(x - y) / (1e-5 + 1e-5 * np.abs(y))).max()))
        h, ha = want[0], got[0]
    assert worst < 0.1, worst
