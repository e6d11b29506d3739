"""oracle/replay_oracle.py against the golden vectors generated from the reference's ReplayBuffer.preprocess_history
(tests/golden/make_golden_replay.py; reference replay_buffer.py:66-99).  Bit-exact."""
import numpy as np
import pytest

from conftest import REPLAY_FIXTURES, load_golden


@pytest.mark.parametrize('name', REPLAY_FIXTURES)
def test_replay_oracle_reproduces_reference_targets(name):
    from oracle import replay_oracle
    fx = load_golden(name)
    got = replay_oracle.run_fixture(fx)
    for key, value in got.items():
        np.testing.assert_array_equal(value, fx['expected_' + key], err_msg=key)


def test_two_hot_rows_sum_to_one_and_invert():
    from oracle import replay_oracle
    v = np.linspace(0, 100, 257).astype(np.float32)
    rows = replay_oracle.two_hot(v, 100, 0.0, 100.0)
    np.testing.assert_allclose(rows.sum(-1), 1.0, atol=1e-6)
    # support_to_scalar (utils.py:146-148) inverts it
    y = (rows * np.arange(101, dtype=np.float32)).sum(-1) / 100
    back = 100 * (((np.sqrt(np.float32(2)) - 1) * y + 1) ** 2 - 1)
    np.testing.assert_allclose(back, v, rtol=1e-4, atol=1e-3)
