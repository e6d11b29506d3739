"""mz_infer_recurrent_noalloc (include/mzhip.h): the allocation-free variant of mz_infer_recurrent, whose kernels take
their scratch from a static device array inside the library.  Called through the C ABI and compared with the existing
entry point bit for bit (same kernels, so equality is exact -- the bar for this path is 1e-5 against the fp64 oracle,
checked as well), on two calls in a row, the second with a smaller batch; oversize batches are rejected."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _call(lib, fn, net, pool, parent, action, block):
    import torch
    T = pool.shape[1]
    dev = pool.device
    out = dict(hidden=torch.full((T, pool.shape[2]), -7.0, device=dev), reward=torch.full((T,), -7.0, device=dev),
               value=torch.full((T,), -7.0, device=dev), logits=torch.full((T, net.ACTION_SPACE_SIZE), -7.0, device=dev),
               raw=torch.full((T, net.VALUE_RAW), -7.0, device=dev))
    status = fn(net._h, T, pool.data_ptr(), parent.data_ptr(), T * pool.shape[2], action.data_ptr(),
                out['hidden'].data_ptr(), out['reward'].data_ptr(), out['value'].data_ptr(), out['logits'].data_ptr(),
                out['raw'].data_ptr())
    torch.cuda.synchronize()
    return status, {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize('kind', ['tictactoe', 'cartpole', 'atari'])
def test_noalloc_equals_recurrent_inference_two_calls_in_a_row(kind):
    import torch
    from muzero_amd import _lib
    from muzero_amd.network import AtariShapedNetwork, CartPoleNetwork, TicTacToeNetwork
    from oracle import net_oracle as no
    lib = _lib.load()
    flat = no.random_weights(kind, 17)
    net = {'tictactoe': TicTacToeNetwork, 'cartpole': CartPoleNetwork, 'atari': AtariShapedNetwork}[kind](config=None,
                                                                                                      weights=flat)
    ref = no.make_oracle(kind, flat)
    rng = np.random.default_rng(8)
    for T in (700, 130):                         # the second call is smaller: stale scratch rows lie beyond it
        pool_np = np.abs(rng.standard_normal((3, T, ref.H))).astype(np.float32)
        parent = rng.integers(-1, 3, size=T).astype(np.int32)             # some rows skipped (inactive trees)
        action = rng.integers(0, ref.A, size=T).astype(np.int32)
        pool, par, act = (torch.from_numpy(x).cuda() for x in (pool_np, parent, action))
        net._sync_stream()
        s_new, got = _call(lib, lib.mz_infer_recurrent_noalloc, net, pool, par, act, 0)
        s_old, want = _call(lib, lib.mz_infer_recurrent, net, pool, par, act, 0)
        assert s_new == 0 and s_old == 0, lib.mz_last_error()
        for key in want:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f'{kind} T={T} {key}')
        live = parent >= 0
        h2, r2, v2, l2, _ = ref.recurrent_inference(pool_np[np.maximum(parent, 0), np.arange(T)][live], action[live])
        vs = 100.0 if kind == 'cartpole' else 1.0
        np.testing.assert_allclose(got['hidden'][live], h2, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(got['reward'][live], r2, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(got['value'][live], v2, rtol=1e-5, atol=1e-5 * vs)
        np.testing.assert_allclose(got['logits'][live], l2, rtol=1e-5, atol=1e-5)
        assert (got['hidden'][~live] == -7.0).all() and (got['reward'][~live] == -7.0).all()


def test_noalloc_rejects_a_batch_beyond_the_static_scratch():
    import torch
    from muzero_amd import _lib
    from muzero_amd.network import TicTacToeNetwork
    lib = _lib.load()
    net = TicTacToeNetwork(config=None, seed=1)
    T = 1025                                      # MZ_NOALLOC_MAX_BATCH + 1
    pool = torch.zeros(1, T, 1152, device='cuda')
    par = torch.zeros(T, dtype=torch.int32, device='cuda')
    status, out = _call(lib, lib.mz_infer_recurrent_noalloc, net, pool, par, par, 0)
    assert status == -1                           # MZ_ERR_INVALID_ARG
    assert b'static scratch' in lib.mz_last_error()
    assert (out['hidden'] == -7.0).all()          # nothing was launched
    status, _ = _call(lib, lib.mz_infer_recurrent_noalloc, net, pool[:, :1024].contiguous(), par[:1024].contiguous(),
                      par[:1024].contiguous(), 0)
    assert status == 0
