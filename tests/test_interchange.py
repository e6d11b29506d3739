"""muzero_amd.interchange against the protobuf runtime: the message types of the reference's
protos/replay_buffer.proto:30-46 (and the TensorProto / TensorShapeProto fields tf.make_tensor_proto fills) are built
as dynamic descriptors, then bytes written by the hand-rolled encoder must parse with the runtime, bytes serialised by
the runtime must decode with the hand-rolled decoder, and both serialisations must be identical."""
import pickle

import numpy as np
import pytest

pb = pytest.importorskip('google.protobuf')
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory     # noqa: E402


def build_messages():
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='mz_test_replay.proto', package='tensorflow.muprover_test', syntax='proto3')

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, number, ftype, label
        if type_name:
            f.type_name = type_name
        return f

    pkg = '.tensorflow.muprover_test.'
    dim = msg('Dim')
    field(dim, 'size', 1, F.TYPE_INT64)
    field(dim, 'name', 2, F.TYPE_STRING)
    shape = msg('TensorShapeProto')
    field(shape, 'dim', 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, pkg + 'Dim')
    field(shape, 'unknown_rank', 3, F.TYPE_BOOL)
    tensor = msg('TensorProto')
    field(tensor, 'dtype', 1, F.TYPE_INT32)                       # enum DataType on the wire is a varint
    field(tensor, 'tensor_shape', 2, F.TYPE_MESSAGE, type_name=pkg + 'TensorShapeProto')
    field(tensor, 'version_number', 3, F.TYPE_INT32)
    field(tensor, 'tensor_content', 4, F.TYPE_BYTES)
    field(tensor, 'float_val', 5, F.TYPE_FLOAT, F.LABEL_REPEATED)
    field(tensor, 'int_val', 7, F.TYPE_INT32, F.LABEL_REPEATED)
    entry = msg('MetadataEntry')
    entry.options.map_entry = True
    field(entry, 'key', 1, F.TYPE_STRING)
    field(entry, 'value', 2, F.TYPE_BYTES)
    gh = msg('GameHistory')
    field(gh, 'observations', 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, pkg + 'TensorProto')
    field(gh, 'to_plays', 2, F.TYPE_INT32, F.LABEL_REPEATED)
    field(gh, 'actions', 3, F.TYPE_INT32, F.LABEL_REPEATED)
    field(gh, 'rewards', 4, F.TYPE_FLOAT, F.LABEL_REPEATED)
    field(gh, 'root_values', 5, F.TYPE_FLOAT, F.LABEL_REPEATED)
    field(gh, 'policies', 6, F.TYPE_MESSAGE, F.LABEL_REPEATED, pkg + 'TensorProto')
    field(gh, 'metadata', 7, F.TYPE_MESSAGE, F.LABEL_REPEATED, pkg + 'MetadataEntry')
    ghl = msg('GameHistoryList')
    field(ghl, 'histories', 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, pkg + 'GameHistory')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.muprover_test.' + n))
    return get('GameHistory'), get('GameHistoryList'), get('TensorProto')


def sample_history(seed, moves, obs_shape=(3, 3, 3), A=9):
    from muzero_amd.game import GameHistory
    rng = np.random.default_rng(seed)
    h = GameHistory()
    h.observations = [rng.integers(0, 2, obs_shape).astype(np.float32) for _ in range(moves + 1)]
    h.to_plays = [int(i % 2) for i in range(moves + 1)]
    h.actions = [int(a) for a in rng.integers(0, A, moves)]
    h.rewards = [float(np.float32(x)) for x in rng.normal(size=moves)]
    h.root_values = [float(np.float32(x)) for x in rng.uniform(-1, 1, moves)]
    h.policies = [rng.dirichlet(np.ones(A)).astype(np.float32) for _ in range(moves)]
    h.metadata = {'agent_id': f'agent_{seed}', 'network_id': '250', 'timing': 0.125}
    return h


def fill_runtime(message, h, Tensor):
    def tensor(arr):
        t = Tensor(dtype=1, tensor_content=np.asarray(arr, '<f4').tobytes())
        for d in np.asarray(arr).shape:
            t.tensor_shape.dim.add(size=d)
        return t
    message.observations.extend(tensor(o) for o in h.observations)
    message.to_plays.extend(h.to_plays)
    message.actions.extend(h.actions)
    message.rewards.extend(h.rewards)
    message.root_values.extend(h.root_values)
    message.policies.extend(tensor(p) for p in h.policies)
    for k, v in h.metadata.items():
        message.metadata[k] = pickle.dumps(v)
    return message


def assert_same_history(a, b):
    assert a.to_plays == b.to_plays and a.actions == b.actions
    assert a.rewards == b.rewards and a.root_values == b.root_values
    assert len(a.observations) == len(b.observations) and len(a.policies) == len(b.policies)
    for x, y in zip(a.observations + a.policies, b.observations + b.policies):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize('moves', [0, 1, 9])
def test_encoder_and_decoder_agree_with_protobuf_runtime(moves):
    from muzero_amd import interchange
    GameHistoryMsg, _, Tensor = build_messages()
    h = sample_history(7 + moves, moves)
    if moves:
        h.to_plays[-1] = -1                      # negative int32: ten-byte varint
    mine = interchange.history_to_bytes(h)
    parsed = GameHistoryMsg()
    parsed.ParseFromString(mine)
    assert list(parsed.to_plays) == h.to_plays and list(parsed.actions) == h.actions
    assert list(parsed.rewards) == h.rewards and list(parsed.root_values) == h.root_values
    assert len(parsed.observations) == moves + 1 and len(parsed.policies) == moves
    for t, want in zip(list(parsed.observations) + list(parsed.policies), h.observations + h.policies):
        assert t.dtype == 1 and [d.size for d in t.tensor_shape.dim] == list(want.shape)
        np.testing.assert_array_equal(np.frombuffer(t.tensor_content, '<f4').reshape(want.shape), want)
    assert {k: pickle.loads(v) for k, v in parsed.metadata.items()} == h.metadata      # our own bytes: safe to load
    theirs = fill_runtime(GameHistoryMsg(), h, Tensor).SerializeToString(deterministic=True)
    assert theirs == mine
    back = interchange.history_from_bytes(theirs)
    assert_same_history(back, h)
    assert back.metadata == {k: pickle.dumps(v) for k, v in h.metadata.items()}        # raw bytes, not unpickled


def test_game_list_file_round_trip(tmp_path):
    from muzero_amd import interchange
    _, GameHistoryListMsg, Tensor = build_messages()
    games = [sample_history(s, m) for s, m in ((1, 5), (2, 9), (3, 7))]
    path = str(tmp_path / 'games.pbuf')
    interchange.save_games(games, path)
    runtime = GameHistoryListMsg()
    with open(path, 'rb') as f:
        runtime.ParseFromString(f.read())
    assert len(runtime.histories) == 3 and list(runtime.histories[1].actions) == games[1].actions
    for got, want in zip(interchange.load_games(path), games):
        assert_same_history(got, want)


def test_tensor_scalar_forms_are_accepted():
    """TensorFlow encodes scalars / constant tensors through float_val / int_val instead of tensor_content."""
    from muzero_amd import interchange
    _, _, Tensor = build_messages()
    t = Tensor(dtype=1, float_val=[0.5])
    for d in (2, 3):
        t.tensor_shape.dim.add(size=d)
    np.testing.assert_array_equal(interchange.tensor_from_bytes(t.SerializeToString()), np.full((2, 3), 0.5, np.float32))
    t = Tensor(dtype=3, int_val=[4, -1, 7])
    t.tensor_shape.dim.add(size=3)
    np.testing.assert_array_equal(interchange.tensor_from_bytes(t.SerializeToString()), np.array([4, -1, 7], np.int32))
    with pytest.raises(ValueError):
        interchange.history_from_bytes(b'\x0a\x05abc')
