"""GameHistory interchange in the reference's wire format.

The reference ships finished games to its replay buffer as the protobuf message `tensorflow.muprover.GameHistory`
(protos/replay_buffer.proto:30-42) built by `history_to_protobuf` (game_services.py:13-23), and stores batches of
them as a serialised `GameHistoryList` (game_services.py:36-42).  Neither generated protobuf code nor TensorFlow's
`tensor.proto` is available here, so the wire format (proto3: varints, packed repeated scalars, length-delimited
sub-messages) is written and read directly; tests/test_interchange.py checks the bytes against the protobuf runtime
with descriptors built from the .proto definitions.

    GameHistory     observations=1 (repeated TensorProto), to_plays=2, actions=3 (packed int32), rewards=4,
                    root_values=5 (packed float), policies=6 (repeated TensorProto), metadata=7 (map<string, bytes>)
    TensorProto     dtype=1 (DT_FLOAT=1, DT_INT32=3), tensor_shape=2 {dim=2 {size=1}}, tensor_content=4 (raw
                    little-endian bytes, what tf.make_tensor_proto emits for arrays); float_val=5 / int_val=7 are
                    accepted on input (TensorFlow uses them for scalars)

Metadata values are pickled by the reference (utils.py:29-34).  They are WRITTEN the same way (str/float/int values
only) and READ BACK AS RAW BYTES: nothing is unpickled here.
"""
import pickle
import struct
from typing import Dict, Iterable, List, Tuple

import numpy as np

from .game import GameHistory

DT_FLOAT, DT_INT32 = 1, 3
_NP_OF_DT = {DT_FLOAT: np.dtype('<f4'), DT_INT32: np.dtype('<i4')}


# ---------------------------------------------------------------------------------------- wire primitives
def _varint(value: int) -> bytes:
    value &= (1 << 64) - 1                      # negative int32/int64 are sent as 10-byte two's complement
    out = bytearray()
    while True:
        byte = value & 0x7F
        value >>= 7
        if value:
            out.append(byte | 0x80)
        else:
            out.append(byte)
            return bytes(out)


def _tag(field: int, wire_type: int) -> bytes:
    return _varint((field << 3) | wire_type)


def _delimited(field: int, payload: bytes) -> bytes:
    return _tag(field, 2) + _varint(len(payload)) + payload


def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise ValueError('truncated varint')
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if not byte & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError('varint too long')


def _fields(buf: bytes) -> Iterable[Tuple[int, int, object]]:
    """Yields (field number, wire type, value) of one message; value = int (varint, fixed) or bytes (delimited)."""
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            value, pos = _read_varint(buf, pos)
        elif wire == 1:
            value, pos = buf[pos:pos + 8], pos + 8
        elif wire == 2:
            size, pos = _read_varint(buf, pos)
            value, pos = buf[pos:pos + size], pos + size
            if len(value) != size:
                raise ValueError('truncated field')
        elif wire == 5:
            value, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f'unsupported wire type {wire}')
        yield field, wire, value


def _signed32(value: int) -> int:
    value &= 0xFFFFFFFF
    return value - (1 << 32) if value & 0x80000000 else value


def _packed_int32(values) -> bytes:
    return b''.join(_varint(int(v)) for v in values)


def _unpack_int32(wire: int, value) -> List[int]:
    if wire == 0:
        return [_signed32(value)]
    out, pos = [], 0
    while pos < len(value):
        v, pos = _read_varint(value, pos)
        out.append(_signed32(v))
    return out


def _unpack_float(wire: int, value) -> List[float]:
    if wire == 5:
        return [struct.unpack('<f', value)[0]]
    return [float(x) for x in np.frombuffer(value, dtype='<f4')]


# ---------------------------------------------------------------------------------------- TensorProto
def tensor_to_bytes(array) -> bytes:
    arr = np.asarray(array)
    if arr.dtype.kind == 'f':
        dt, arr = DT_FLOAT, arr.astype('<f4')
    elif arr.dtype.kind in 'iub':
        dt, arr = DT_INT32, arr.astype('<i4')
    else:
        raise TypeError(f'unsupported tensor dtype {arr.dtype}')
    shape = b''.join(_delimited(2, _tag(1, 0) + _varint(int(d))) for d in arr.shape)
    return _tag(1, 0) + _varint(dt) + _delimited(2, shape) + _delimited(4, np.ascontiguousarray(arr).tobytes())


def tensor_from_bytes(buf: bytes) -> np.ndarray:
    dt, shape, content, float_val, int_val = DT_FLOAT, [], None, [], []
    for field, wire, value in _fields(buf):
        if field == 1:
            dt = value
        elif field == 2:
            for f2, _w2, dim in _fields(value):
                if f2 == 2:
                    size = 0
                    for f3, _w3, v3 in _fields(dim):
                        if f3 == 1:
                            size = v3
                    shape.append(size)
        elif field == 4:
            content = value
        elif field == 5:
            float_val += _unpack_float(wire, value)
        elif field == 7:
            int_val += _unpack_int32(wire, value)
    if dt not in _NP_OF_DT:
        raise ValueError(f'unsupported TensorProto dtype {dt}')
    count = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        arr = np.frombuffer(content, dtype=_NP_OF_DT[dt])
    else:
        vals = float_val if dt == DT_FLOAT else int_val
        if len(vals) == 1 and count > 1:             # TensorFlow's "all elements equal" compression
            vals = vals * count
        arr = np.asarray(vals, dtype=_NP_OF_DT[dt])
    if arr.size != count:
        raise ValueError(f'TensorProto holds {arr.size} elements for shape {shape}')
    return arr.reshape(shape).astype(_NP_OF_DT[dt].newbyteorder('='))


# ---------------------------------------------------------------------------------------- GameHistory
def history_to_bytes(history: GameHistory) -> bytes:
    """history_to_protobuf(...).SerializeToString() (game_services.py:13-23)."""
    out = bytearray()
    for obs in history.observations:
        out += _delimited(1, tensor_to_bytes(np.asarray(obs, dtype=np.float32)))
    if history.to_plays:
        out += _delimited(2, _packed_int32(history.to_plays))
    if history.actions:
        out += _delimited(3, _packed_int32(history.actions))
    if history.rewards:
        out += _delimited(4, np.asarray(history.rewards, dtype='<f4').tobytes())
    if history.root_values:
        out += _delimited(5, np.asarray(history.root_values, dtype='<f4').tobytes())
    for policy in history.policies:
        out += _delimited(6, tensor_to_bytes(np.asarray(policy, dtype=np.float32)))
    for key in sorted(history.metadata):
        value = history.metadata[key]
        if not isinstance(value, (str, int, float, bool)):
            raise TypeError(f'metadata[{key!r}] must be str/int/float/bool to be pickled for the reference')
        entry = _delimited(1, key.encode('utf-8')) + _delimited(2, pickle.dumps(value))        # utils.py:29-30
        out += _delimited(7, entry)
    return bytes(out)


def history_from_bytes(buf: bytes) -> GameHistory:
    """history_from_protobuf (game_services.py:26-34); metadata values stay raw bytes (never unpickled)."""
    h = GameHistory()
    for field, wire, value in _fields(buf):
        if field == 1:
            h.observations.append(tensor_from_bytes(value))
        elif field == 2:
            h.to_plays += _unpack_int32(wire, value)
        elif field == 3:
            h.actions += _unpack_int32(wire, value)
        elif field == 4:
            h.rewards += _unpack_float(wire, value)
        elif field == 5:
            h.root_values += _unpack_float(wire, value)
        elif field == 6:
            h.policies.append(tensor_from_bytes(value))
        elif field == 7:
            key, raw = '', b''
            for f2, _w2, v2 in _fields(value):
                if f2 == 1:
                    key = v2.decode('utf-8')
                elif f2 == 2:
                    raw = bytes(v2)
            h.metadata[key] = raw
    return h


def histories_to_bytes(histories: Iterable[GameHistory]) -> bytes:
    """GameHistoryList.SerializeToString() (game_services.py:36-41)."""
    return b''.join(_delimited(1, history_to_bytes(h)) for h in histories)


def histories_from_bytes(buf: bytes) -> List[GameHistory]:
    return [history_from_bytes(value) for field, _wire, value in _fields(buf) if field == 1]


def save_games(game_histories: Iterable[GameHistory], filepath: str) -> None:
    """game_services.save_games (:36-42): a file the reference's ReplayBuffer.load_games can read."""
    with open(filepath, 'wb') as f:
        f.write(histories_to_bytes(game_histories))


def load_games(filepath: str) -> List[GameHistory]:
    """game_services.load_games (:45-52)."""
    with open(filepath, 'rb') as f:
        return histories_from_bytes(f.read())
