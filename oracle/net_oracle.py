"""CPU oracle for the network half of the hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED against TensorFlow: the reference's arithmetic for this half lives in the un-vendored,
un-pinned third-party `tensorflow` package (inferred ~2.4; no requirements file), which is not installed here
and cannot be fetched.  The only reference known-answer test near this path is
test_tensor_product_encoder (network.py:77-85); it is reproduced in tests/test_net_oracle.py.  Everything
else below restates the documented Keras layer semantics in numpy (float64 by default, so it serves as the
independent reference for the fp32 HIP kernels; tolerance 1e-5 relative, see tests/test_net_gpu.py):

  * Conv2D / Conv1D: NHWC / NWC, kernel (k..., cin, cout), cross-correlation, `same` = zero pad 1 for k=3,
    `valid` = no pad; Dense acts on the last axis; Flatten is row-major over (h, w, c).
  * one_hot_tensor_encoder    network.py:14-40      binary_plane_encoder   network.py:88-104
  * tensor_product_encoder    network.py:43-74      scalar_to_support(_model)  network.py:107-116, utils.py:138-143
  * support_to_scalar         utils.py:146-148      ScalarConfig.to_scalar     config.py:25-40
  * TicTacToeNetwork          games/tictactoe.py:151-220
  * CartPoleNetwork           games/cartpole.py:63-139
  * initial / recurrent inference graphs   network.py:209-254; split_batch network.py:165-174
  * AtariShapedOracle: the build's OWN synthetic network for BASELINE.json configs[3] ("18 actions, 256-d hidden
    state, conv-residual dynamics") -- the reference has no such network; it is assembled from the reference's
    building blocks (one_hot_tensor_encoder, the conv + dense stacks of games/tictactoe.py:188-213) with Keras
    semantics, and is the definition the HIP kernels are held to.

Weights are a flat float32 buffer in the order of WEIGHT_SPECS (Keras layer order, kernel then bias), the same
buffer libmzhip's mz_net_create takes.
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------------ specs
TICTACTOE_SPEC = [
    ('rep_conv_k', (3, 3, 3, 128)), ('rep_conv_b', (128,)),
    ('rep_dense_k', (128, 128)), ('rep_dense_b', (128,)),
    ('dyn_conv_k', (3, 3, 137, 128)), ('dyn_conv_b', (128,)),
    ('dyn_dense_k', (128, 128)), ('dyn_dense_b', (128,)),
    ('dyn_reward_k', (1152, 1)), ('dyn_reward_b', (1,)),
    ('pred_conv_k', (3, 3, 128, 128)), ('pred_conv_b', (128,)),
    ('pred_dense_k', (128, 128)), ('pred_dense_b', (128,)),
    ('pred_value_k', (128, 1)), ('pred_value_b', (1,)),
    ('pred_policy_k', (128, 9)), ('pred_policy_b', (9,)),
]
CARTPOLE_SPEC = [
    ('rep_conv_k', (4, 101, 128)), ('rep_conv_b', (128,)),
    ('rep_dense_k', (128, 128)), ('rep_dense_b', (128,)),
    ('dyn_conv_k', (2, 128, 128)), ('dyn_conv_b', (128,)),
    ('dyn_dense_k', (128, 128)), ('dyn_dense_b', (128,)),
    ('dyn_rdense_k', (128, 128)), ('dyn_rdense_b', (128,)),
    ('dyn_reward_k', (128, 1)), ('dyn_reward_b', (1,)),
    ('pred_dense1_k', (128, 128)), ('pred_dense1_b', (128,)),
    ('pred_dense2_k', (128, 128)), ('pred_dense2_b', (128,)),
    ('pred_value_k', (128, 101)), ('pred_value_b', (101,)),
    ('pred_policy_k', (128, 2)), ('pred_policy_b', (2,)),
]
ATARI_SPEC = [      # hidden state (2, 128): two positions x 128 channels = 256 floats; 18 actions
    ('rep_conv_k', (3, 128, 128)), ('rep_conv_b', (128,)),
    ('rep_dense_k', (128, 128)), ('rep_dense_b', (128,)),
    ('dyn_conv_k', (3, 146, 128)), ('dyn_conv_b', (128,)),
    ('dyn_res1a_k', (3, 128, 128)), ('dyn_res1a_b', (128,)), ('dyn_res1b_k', (3, 128, 128)), ('dyn_res1b_b', (128,)),
    ('dyn_res2a_k', (3, 128, 128)), ('dyn_res2a_b', (128,)), ('dyn_res2b_k', (3, 128, 128)), ('dyn_res2b_b', (128,)),
    ('dyn_reward_k', (256, 1)), ('dyn_reward_b', (1,)),
    ('pred_conv_k', (2, 128, 128)), ('pred_conv_b', (128,)),
    ('pred_dense_k', (128, 128)), ('pred_dense_b', (128,)),
    ('pred_value_k', (128, 1)), ('pred_value_b', (1,)), ('pred_policy_k', (128, 18)), ('pred_policy_b', (18,)),
]
WEIGHT_SPECS = {'tictactoe': TICTACTOE_SPEC, 'cartpole': CARTPOLE_SPEC, 'atari': ATARI_SPEC}
CARTPOLE_SCALAR_MIN = np.array([-4.8, -5., -0.418, -5.], np.float32)     # games/cartpole.py:93
CARTPOLE_SCALAR_MAX = np.array([4.8, 5., 0.418, 5.], np.float32)        # games/cartpole.py:94
CARTPOLE_SUPPORT = 100                                                   # games/cartpole.py:35


def num_weights(kind):
    return sum(int(np.prod(shape)) for _, shape in WEIGHT_SPECS[kind])


def unpack(kind, flat, dtype=np.float64):
    flat = np.asarray(flat, np.float32)
    assert flat.size == num_weights(kind)
    out, off = {}, 0
    for name, shape in WEIGHT_SPECS[kind]:
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape).astype(dtype)
        off += n
    return out


def random_weights(kind, seed, bias_scale=0.05):
    """Keras defaults are glorot-uniform kernels and zero biases; biases are drawn non-zero here so tests
    exercise the bias path.  Returns the flat float32 buffer."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    for name, shape in WEIGHT_SPECS[kind]:
        if name.endswith('_k'):
            receptive = int(np.prod(shape[:-2]))
            fan_in, fan_out = receptive * shape[-2], receptive * shape[-1]
            limit = math.sqrt(6.0 / (fan_in + fan_out))
            parts.append(rng.uniform(-limit, limit, size=shape).astype(np.float32).ravel())
        else:
            parts.append((rng.standard_normal(shape) * bias_scale).astype(np.float32).ravel())
    return np.concatenate(parts)


# ------------------------------------------------------------------------------------------------ layers
def relu(x):
    return np.maximum(x, 0)


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


def softmax(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def dense(x, k, b):
    return x @ k + b


def conv2d(x, k, b, padding):
    """x [B,H,W,Cin], k [kh,kw,Cin,Cout]; stride 1; cross-correlation like Keras."""
    kh, kw = k.shape[:2]
    if padding == 'same':
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        x = np.pad(x, ((0, 0), (ph, kh - 1 - ph), (pw, kw - 1 - pw), (0, 0)))
    B, H, W, _ = x.shape
    oh, ow = H - kh + 1, W - kw + 1
    out = np.zeros((B, oh, ow, k.shape[-1]), x.dtype)
    for i in range(kh):
        for j in range(kw):
            out += np.einsum('bhwc,co->bhwo', x[:, i:i + oh, j:j + ow, :], k[i, j])
    return out + b


def conv1d(x, k, b, padding='valid'):
    """x [B,L,Cin], k [kw,Cin,Cout]; `valid`, or `same` (zero pad (kw-1)//2 left, the rest right, like Keras)."""
    kw = k.shape[0]
    if padding == 'same':
        pl = (kw - 1) // 2
        x = np.pad(x, ((0, 0), (pl, kw - 1 - pl), (0, 0)))
    ol = x.shape[1] - kw + 1
    out = np.zeros((x.shape[0], ol, k.shape[-1]), x.dtype)
    for j in range(kw):
        out += np.einsum('blc,co->blo', x[:, j:j + ol, :], k[j])
    return out + b


# ------------------------------------------------------------------------------------------------ encoders
def one_hot_tensor_encoder(hidden, action, action_space_size):
    """network.py:14-40: append A channels; channel C+action is 1 at every position."""
    B = hidden.shape[0]
    planes = np.zeros(hidden.shape[:-1] + (action_space_size,), hidden.dtype)
    planes[np.arange(B), ..., np.asarray(action)] = 1
    return np.concatenate([hidden, planes], axis=-1)


def binary_plane_encoder(hidden, action, axis=0):
    """network.py:88-104: append one all-0 / all-1 plane along `axis` (+1 for the batch axis)."""
    shape = list(hidden.shape)
    shape[axis + 1] = 1
    plane = np.broadcast_to(np.asarray(action, hidden.dtype).reshape((-1,) + (1,) * (hidden.ndim - 1)), shape)
    return np.concatenate([hidden, plane], axis=axis + 1)


def tensor_product_encoder(hidden, action, action_space_size):
    """network.py:43-74: out[b, action[b], ...] = hidden[b, ...], zero elsewhere."""
    out = np.zeros((hidden.shape[0], action_space_size) + hidden.shape[1:], hidden.dtype)
    out[np.arange(hidden.shape[0]), np.asarray(action)] = hidden
    return out


def scalar_to_support(x, support_size):
    """utils.py:138-143 (sqrt transform, two-hot between floor and ceil)."""
    y = support_size * (np.sqrt(x + 1) - 1) / (math.sqrt(2.) - 1)
    low = np.floor(y)
    high = np.ceil(y).astype(np.int64)
    p = (y - low)[..., None]
    eye = np.eye(support_size + 1, dtype=x.dtype)
    return p * eye[high] + (1 - p) * eye[low.astype(np.int64)]


def support_to_scalar(x, support_size):
    """utils.py:146-148."""
    y = (x * np.arange(support_size + 1, dtype=x.dtype)).sum(-1) / support_size
    return ((math.sqrt(2.) - 1) * y + 1) ** 2 - 1


def to_scalar(x, support_size, known_bounds):
    """ScalarConfig.to_scalar (config.py:25-40)."""
    if support_size is None:
        return x
    a, b = known_bounds if known_bounds is not None else (0.0, 1.0)
    return a + (b - a) * support_to_scalar(x, support_size)


# ------------------------------------------------------------------------------------------------ networks
class TicTacToeOracle:
    """games/tictactoe.py:151-220.  hidden [B,3,3,128]; value/reward scalar heads (tanh)."""
    kind, A, H = 'tictactoe', 9, 1152

    def __init__(self, flat_weights, dtype=np.float64):
        self.w = unpack(self.kind, flat_weights, dtype)
        self.dtype = dtype

    def representation(self, obs):
        w = self.w
        x = relu(conv2d(np.asarray(obs, self.dtype).reshape(-1, 3, 3, 3), w['rep_conv_k'], w['rep_conv_b'], 'same'))
        return relu(dense(x, w['rep_dense_k'], w['rep_dense_b']))

    def dynamics(self, hidden, action):
        w = self.w
        enc = one_hot_tensor_encoder(np.asarray(hidden, self.dtype).reshape(-1, 3, 3, 128), action, 9)
        x = relu(conv2d(enc, w['dyn_conv_k'], w['dyn_conv_b'], 'same'))
        new_hidden = relu(dense(x, w['dyn_dense_k'], w['dyn_dense_b']))
        reward = np.tanh(dense(new_hidden.reshape(-1, 1152), w['dyn_reward_k'], w['dyn_reward_b']))
        return new_hidden, reward

    def prediction(self, hidden):
        w = self.w
        x = relu(conv2d(hidden, w['pred_conv_k'], w['pred_conv_b'], 'valid')).reshape(-1, 128)
        x = relu(dense(x, w['pred_dense_k'], w['pred_dense_b']))
        value = np.tanh(dense(x, w['pred_value_k'], w['pred_value_b']))
        logits = dense(x, w['pred_policy_k'], w['pred_policy_b'])
        return value, logits

    def initial_inference(self, obs):
        """-> hidden [B,1152], value [B] (scalar), logits [B,9], raw value [B,1]"""
        hidden = self.representation(obs)
        value, logits = self.prediction(hidden)
        return hidden.reshape(-1, 1152), value[:, 0], logits, value

    def recurrent_inference(self, hidden, action):
        """-> hidden' [B,1152], reward [B], value [B], logits [B,9], raw value [B,1]"""
        new_hidden, reward = self.dynamics(hidden, action)
        value, logits = self.prediction(new_hidden)
        return new_hidden.reshape(-1, 1152), reward[:, 0], value[:, 0], logits, value


class CartPoleOracle:
    """games/cartpole.py:63-139.  hidden [B,1,128]; 101-bin value support scalarised over (0,100)."""
    kind, A, H = 'cartpole', 2, 128
    value_bounds = (0.0, 100.0)      # games/cartpole.py:50-52

    def __init__(self, flat_weights, dtype=np.float64):
        self.w = unpack(self.kind, flat_weights, dtype)
        self.dtype = dtype

    def preprocess(self, obs):
        obs = np.asarray(obs, self.dtype).reshape(-1, 4)
        lo, hi = CARTPOLE_SCALAR_MIN.astype(self.dtype), CARTPOLE_SCALAR_MAX.astype(self.dtype)
        return scalar_to_support((obs - lo) / (hi - lo), CARTPOLE_SUPPORT)       # [B,4,101]

    def representation(self, obs):
        w = self.w
        x = relu(conv1d(self.preprocess(obs), w['rep_conv_k'], w['rep_conv_b']))    # [B,1,128]
        return relu(dense(x, w['rep_dense_k'], w['rep_dense_b']))

    def dynamics(self, hidden, action):
        w = self.w
        enc = binary_plane_encoder(np.asarray(hidden, self.dtype).reshape(-1, 1, 128), action, axis=0)   # [B,2,128]
        x = relu(conv1d(enc, w['dyn_conv_k'], w['dyn_conv_b']))
        new_hidden = relu(dense(x, w['dyn_dense_k'], w['dyn_dense_b']))             # [B,1,128]
        x = relu(dense(new_hidden.reshape(-1, 128), w['dyn_rdense_k'], w['dyn_rdense_b']))
        reward = sigmoid(dense(x, w['dyn_reward_k'], w['dyn_reward_b']))
        return new_hidden, reward

    def prediction(self, hidden):
        w = self.w
        x = relu(dense(hidden.reshape(-1, 128), w['pred_dense1_k'], w['pred_dense1_b']))
        x = relu(dense(x, w['pred_dense2_k'], w['pred_dense2_b']))
        value = softmax(dense(x, w['pred_value_k'], w['pred_value_b']))
        logits = dense(x, w['pred_policy_k'], w['pred_policy_b'])
        return value, logits

    def initial_inference(self, obs):
        hidden = self.representation(obs)
        value, logits = self.prediction(hidden)
        return hidden.reshape(-1, 128), to_scalar(value, CARTPOLE_SUPPORT, self.value_bounds), logits, value

    def recurrent_inference(self, hidden, action):
        new_hidden, reward = self.dynamics(hidden, action)
        value, logits = self.prediction(new_hidden)
        return (new_hidden.reshape(-1, 128), reward[:, 0], to_scalar(value, CARTPOLE_SUPPORT, self.value_bounds),
                logits, value)


class AtariShapedOracle:
    """The synthetic Atari-shaped network of BASELINE.json configs[3] (no counterpart class in the reference; built
    from its blocks).  observation (2,128); hidden (2,128) = 256 floats; 18 actions; tanh value / reward scalars.

      representation  Conv1D k=3 same 128->128 relu; Dense 128->128 relu (per position)
      dynamics        one_hot_tensor_encoder(state_shape=(2,), 18) (network.py:14-40) -> (2,146);
                      Conv1D k=3 same 146->128 relu = x;  two residual blocks  x = relu(x + Conv1D(relu(Conv1D(x))))
                      (k=3 same 128->128, biases on both) = hidden';  Flatten(256) -> Dense 1 tanh = reward
      prediction      Conv1D k=2 valid 128->128 relu -> (1,128); Flatten; Dense 128->128 relu; Dense 1 tanh = value;
                      Dense 18 = logits                         (the shape of games/tictactoe.py:200-213)
    FLOPs of one recurrent_inference (multiply-adds x 2, only the taps that touch real positions):
      2 * (4*146*128 + 16*128*128 + 256 + 2*128*128 + 128*128 + 128*19) = 777,472"""
    kind, A, H = 'atari', 18, 256
    FLOPS_RECURRENT = 2 * (4 * 146 * 128 + 16 * 128 * 128 + 256 + 2 * 128 * 128 + 128 * 128 + 128 * 19)

    def __init__(self, flat_weights, dtype=np.float64):
        self.w = unpack(self.kind, flat_weights, dtype)
        self.dtype = dtype

    def representation(self, obs):
        w = self.w
        x = relu(conv1d(np.asarray(obs, self.dtype).reshape(-1, 2, 128), w['rep_conv_k'], w['rep_conv_b'], 'same'))
        return relu(dense(x, w['rep_dense_k'], w['rep_dense_b']))

    def dynamics(self, hidden, action):
        w = self.w
        enc = one_hot_tensor_encoder(np.asarray(hidden, self.dtype).reshape(-1, 2, 128), action, 18)
        x = relu(conv1d(enc, w['dyn_conv_k'], w['dyn_conv_b'], 'same'))
        for blk in ('dyn_res1', 'dyn_res2'):
            y = relu(conv1d(x, w[blk + 'a_k'], w[blk + 'a_b'], 'same'))
            x = relu(x + conv1d(y, w[blk + 'b_k'], w[blk + 'b_b'], 'same'))
        reward = np.tanh(dense(x.reshape(-1, 256), w['dyn_reward_k'], w['dyn_reward_b']))
        return x, reward

    def prediction(self, hidden):
        w = self.w
        x = relu(conv1d(hidden, w['pred_conv_k'], w['pred_conv_b'])).reshape(-1, 128)
        x = relu(dense(x, w['pred_dense_k'], w['pred_dense_b']))
        value = np.tanh(dense(x, w['pred_value_k'], w['pred_value_b']))
        logits = dense(x, w['pred_policy_k'], w['pred_policy_b'])
        return value, logits

    def initial_inference(self, obs):
        """-> hidden [B,256], value [B] (scalar), logits [B,18], raw value [B,1]"""
        hidden = self.representation(obs)
        value, logits = self.prediction(hidden)
        return hidden.reshape(-1, 256), value[:, 0], logits, value

    def recurrent_inference(self, hidden, action):
        """-> hidden' [B,256], reward [B], value [B], logits [B,18], raw value [B,1]"""
        new_hidden, reward = self.dynamics(hidden, action)
        value, logits = self.prediction(new_hidden)
        return new_hidden.reshape(-1, 256), reward[:, 0], value[:, 0], logits, value


def make_oracle(kind, flat_weights, dtype=np.float64):
    return {'tictactoe': TicTacToeOracle, 'cartpole': CartPoleOracle, 'atari': AtariShapedOracle}[kind](flat_weights, dtype)
