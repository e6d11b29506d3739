"""CPU oracle for the replay-side training targets.  TEST INFRASTRUCTURE ONLY: imported by tests/
(never by muzero_amd/), pinned by tests/golden/replay_*.npz, which tests/golden/make_golden_replay.py generated from
the unmodified reference.

Restates, per game, reference replay_buffer.py:66-99:
  compute_target_value (:66-73)  value = root_values[i+td] * disc**td  (0 past the end)  +  sum_k rewards[i+k] * disc**k,
                                 Python floats (fp64), `sum()` left to right starting from int 0;
  preprocess_history   (:75-99)  actions + num_unroll_steps random actions; rewards + zeros; values + zeros;
                                 policies + uniform rows; inv_to_scalar (config.py:15-36, utils.py:138-143) = a trailing
                                 axis of 1 (support_size None) or the float32 two-hot support; total_value.
The two-hot branch follows the formula in float32 like TensorFlow would; TensorFlow itself is absent, so that branch is
"parity unpinned" against TF's kernels (sqrt/floor/ceil/divide are IEEE-exact operations, so no difference is expected).
"""
import numpy as np


def effective_discount(discount, num_players):
    return -discount if num_players == 2 else discount          # replay_buffer.py:26-28


def discount_powers(disc, n):
    """disc**i for i in 0..n with Python's float pow, as the reference evaluates it (:69,:72,:98)."""
    return np.array([disc ** i for i in range(n + 1)], dtype=np.float64)


def compute_target_value(rewards, root_values, index, td, disc):
    b = index + td
    value = root_values[b] * disc ** td if b < len(rewards) else 0
    value += sum(r * disc ** i for i, r in enumerate(rewards[index:b]))
    return value


def two_hot(values_f32, support, lo, hi):
    """config.py:34-36 + utils.py:138-143 in float32."""
    f = np.float32
    x = (values_f32.astype(f) - f(lo)) / f(hi - lo)
    y = f(support) * (np.sqrt(x + f(1)) - f(1)) / (np.sqrt(f(2.0)) - f(1))
    low = np.floor(y)
    high = np.ceil(y).astype(np.int32)
    p = (y - low)[..., None]
    eye = lambda idx: (np.arange(support + 1)[None, :] == idx[..., None]).astype(f)
    return (p * eye(high) + (f(1) - p) * eye(low.astype(np.int32))).astype(f)


def inv_to_scalar(values, support, bounds):
    v = np.asarray(values, dtype=np.float32)                      # tf.constant(list, dtype=float32)
    if support is None or support <= 0:
        return v[..., None]
    lo, hi = bounds
    if np.isnan(lo):
        lo = -hi
    return two_hot(v, support, lo, hi)


def preprocess(actions, rewards, root_values, policies, random_actions, A, unroll, td, disc, value_support=None,
               value_bounds=(0.0, 1.0)):
    """One game.  -> dict of extended_actions [L+K] i32, target_rewards [L+K] f32, target_values [L+K+1, Dv] f32,
    target_policies [L+K+1, A] f32, total_value f64."""
    rewards = [float(r) for r in rewards]
    root_values = [float(v) for v in root_values]
    L = len(actions)
    ext = np.array(list(actions) + list(random_actions), dtype=np.int32)
    trew = np.array(rewards + [0] * unroll, dtype=np.float32)
    tval = [compute_target_value(rewards, root_values, i, td, disc) for i in range(L)] + [0] * (unroll + 1)
    tval = inv_to_scalar(tval, value_support, value_bounds)
    uniform = np.ones((unroll + 1, A), np.float32) / A
    tpol = np.concatenate([np.asarray(policies, np.float32).reshape(L, A), uniform], axis=0)
    total = sum(r * disc ** i for i, r in enumerate(rewards))
    return dict(extended_actions=ext, target_rewards=trew, target_values=tval, target_policies=tpol,
                total_value=np.float64(total))


def run_fixture(fx):
    """All games of a replay fixture -> padded arrays shaped like the fixture's expected_* entries."""
    A, K, td = int(fx['A']), int(fx['unroll']), int(fx['td'])
    disc = effective_discount(float(fx['discount']), int(fx['players']))
    support = int(fx['value_support'])
    out = {k[len('expected_'):]: np.zeros_like(fx[k]) for k in fx if k.startswith('expected_')}
    for g, L in enumerate(fx['lengths']):
        r = preprocess(fx['actions'][:L, g], fx['rewards'][:L, g], fx['root_values'][:L, g], fx['policies'][:L, g],
                       fx['random_actions'][g], A, K, td, disc, support if support > 0 else None,
                       tuple(fx['value_bounds']))
        out['extended_actions'][:L + K, g] = r['extended_actions']
        out['target_rewards'][:L + K, g] = r['target_rewards']
        out['target_values'][:L + K + 1, g] = r['target_values']
        out['target_policies'][:L + K + 1, g] = r['target_policies']
        out['total_value'][g] = r['total_value']
    return out
