"""Host-side logic that needs no GPU: parameter validation of the network classes, bench.py's CPU-baseline plumbing."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_network_classes_refuse_sizes_the_kernels_do_not_implement():
    """The reference's network classes take hidden_state_size / hidden_size / support_size (games/tictactoe.py:156-161,
    games/cartpole.py:68-74); the HIP kernels implement 128 / 128 / 100 only and say so instead of running another
    network than the one configured."""
    from muzero_amd.exceptions import MuZeroError
    from muzero_amd.network import CartPoleNetwork, TicTacToeNetwork
    for cls, kw in ((TicTacToeNetwork, dict(hidden_size=256)), (TicTacToeNetwork, dict(hidden_state_size=64)),
                    (TicTacToeNetwork, dict(support_size=100)), (CartPoleNetwork, dict(support_size=300)),
                    (CartPoleNetwork, dict(hidden_size=128, hidden_state_size=256)),
                    (CartPoleNetwork, dict(no_such_parameter=1))):
        with pytest.raises(MuZeroError, match='not implemented by the HIP kernels|unknown network parameter'):
            cls(None, **kw)


def test_reference_style_network_config_passes_the_size_check():
    """make_config()'s own parameters (the reference's defaults) pass; what fails next without a GPU is the device check."""
    import torch
    from muzero_amd.exceptions import MuZeroError
    from muzero_amd.games import cartpole, tictactoe
    if torch.cuda.is_available():
        pytest.skip('exercises the no-GPU failure path')
    for mod in (tictactoe, cartpole):
        config = mod.make_config()
        with pytest.raises(MuZeroError, match='needs a HIP device'):
            config.make_uniform_network()


def test_cpu_baseline_workers_start_idle_and_run_when_released():
    sys.path.insert(0, ROOT)
    import bench
    wl = dict(bench.WORKLOADS['tictactoe_4096x25'], net=None)
    cpu = bench.CpuBaseline(wl, seconds=0.3, cores=2)
    out = cpu.collect()
    assert out['cores'] == 2 and out['kind'] == 'port' and out['value'] > 0 and out['unit'] == 'simulations/s'
    cpu = bench.CpuBaseline(wl, seconds=0.3, cores=1)
    cpu.abandon()
    assert cpu.procs[0].returncode == 0
