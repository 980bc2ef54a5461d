"""TensorFlow checkpoint bundles without TensorFlow (imm_amd/utils/tf_checkpoint.py): byte-level known
answers assembled by hand from the published formats (LevelDB table format, tensor_bundle.proto, RFC 3720 CRC-32C
vectors), round trips, corruption detection and the variable-name map of the reference's graph.  PARITY UNPINNED: no
file written by TensorFlow is available here."""
import os
import struct

import numpy as np
import pytest

from imm_amd.utils import tf_checkpoint as T


def test_crc32c_known_answers():
    assert T.crc32c(b'123456789') == 0xe3069283                     # the classic check value
    assert T.crc32c(b'\x00' * 32) == 0x8a9136aa                      # RFC 3720 B.4
    assert T.crc32c(b'\xff' * 32) == 0x62a8ab43
    assert T.crc32c(bytes(range(32))) == 0x46dd794e
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
    assert T.crc32c(b'world', T.crc32c(b'hello ')) == T.crc32c(b'hello world')
    assert T.crc32c(b'') == 0
    for v in (0, 1, 0xe3069283, 0xffffffff):
        assert T.unmask_crc(T.mask_crc(v)) == v
    assert T.mask_crc(0) == 0xa282ead8 and T.mask_crc(0x00008000) == (1 + 0xa282ead8)


def test_protobuf_known_answers():
    # BundleEntryProto{dtype: DT_FLOAT, shape: [2, 3], size: 24, crc32c: 0x12345678}; zero offset / shard are omitted
    want = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03, 0x28, 0x18, 0x35, 0x78, 0x56, 0x34, 0x12])
    assert T.encode_entry(T.DT_FLOAT, (2, 3), 0, 0, 24, 0x12345678) == want
    e = T.decode_entry(want)
    assert (e['dtype'], e['shape'], e['shard_id'], e['offset'], e['size'], e['crc32c']) == (1, [2, 3], 0, 0, 24, 0x12345678)
    # offsets beyond one byte are varints; a scalar has an empty shape message
    msg = T.encode_entry(T.DT_INT64, (), 0, 300, 8, 1)
    assert msg == bytes([0x08, 0x09, 0x12, 0x00, 0x20, 0xac, 0x02, 0x28, 0x08, 0x35, 1, 0, 0, 0])
    assert T.decode_entry(msg)['shape'] == [] and T.decode_entry(msg)['offset'] == 300
    assert T.encode_header() == bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])


def test_table_bytes_of_a_two_entry_file(tmp_path):
    """One data block with keys '' and 'ab', written out by hand following the table format."""
    p = str(tmp_path / 't.index')
    T.write_table(p, [(b'', b'H'), (b'ab', b'xyz')])
    raw = open(p, 'rb').read()
    block = (bytes([0, 0, 1]) + b'H'                       # shared 0, non_shared 0, value_len 1
             + bytes([0, 2, 3]) + b'ab' + b'xyz'            # first block of a restart interval: full keys
             + struct.pack('<II', 0, 1))                    # restart offsets [0], count 1
    trailer = b'\x00' + struct.pack('<I', T.mask_crc(T.crc32c(block + b'\x00')))
    assert raw[:len(block) + 5] == block + trailer
    meta = struct.pack('<II', 0, 1)                         # empty metaindex block: restarts [0], count 1
    off_meta = len(block) + 5
    assert raw[off_meta:off_meta + 8] == meta
    off_index = off_meta + 8 + 5
    handle0 = bytes([0, len(block)])                        # BlockHandle{offset 0, size}
    index = bytes([0, 2, 2]) + b'ab' + handle0 + struct.pack('<II', 0, 1)
    assert raw[off_index:off_index + len(index)] == index
    footer = raw[-48:]
    assert footer[:4] == bytes([off_meta, 8, off_index, len(index)]) and footer[4:40] == b'\x00' * 36
    assert struct.unpack('<Q', footer[40:])[0] == 0xdb4775248b80fb57
    assert len(raw) == off_index + len(index) + 5 + 48
    assert list(T.read_table(p).items()) == [(b'', b'H'), (b'ab', b'xyz')]


def test_prefix_compression_restarts_and_many_blocks(tmp_path):
    p = str(tmp_path / 'big.index')
    items = [(('model/layer_%04d/batch_normalization/gamma' % i).encode(), os.urandom(1 + i % 40)) for i in range(700)]
    T.write_table(p, items)
    assert list(T.read_table(p).items()) == items
    assert os.path.getsize(p) < sum(len(k) + len(v) for k, v in items)          # shared prefixes are not repeated
    with pytest.raises(ValueError):
        T.write_table(str(tmp_path / 'bad'), [(b'b', b''), (b'a', b'')])


def test_bundle_round_trip_and_listing(tmp_path):
    rng = np.random.RandomState(0)
    tens = {'model/renderer/conv_1/conv_1/w': rng.randn(3, 3, 4, 8).astype(np.float32), 'global_step': np.asarray(7, np.float32),
            'beta1_power': np.asarray(0.9 ** 8, np.float32), 'ints': np.arange(5, dtype=np.int64), 'h': rng.randn(3).astype(np.float16),
            'empty': np.zeros((0, 4), np.float32)}
    prefix = str(tmp_path / 'logs' / 'model.ckpt-7')
    T.write_bundle(prefix, tens)
    assert sorted(os.listdir(tmp_path / 'logs')) == ['model.ckpt-7.data-00000-of-00001', 'model.ckpt-7.index']
    assert os.path.getsize(prefix + '.data-00000-of-00001') == sum(v.nbytes for v in tens.values())
    back = T.read_bundle(prefix)
    assert list(back) == sorted(tens)                                    # bytewise key order
    for k in tens:
        assert back[k].dtype == tens[k].dtype and back[k].shape == tens[k].shape
        np.testing.assert_array_equal(back[k], tens[k])
    listing = T.list_bundle(prefix)
    assert listing['global_step'] == (np.dtype(np.float32), ()) and listing['model/renderer/conv_1/conv_1/w'][1] == (3, 3, 4, 8)
    assert list(T.read_bundle(prefix, names={'ints'})) == ['ints']


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'model.ckpt-1')
    T.write_bundle(prefix, {'a': np.arange(64, dtype=np.float32), 'b': np.ones(3, np.float32)})
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[10] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    with pytest.raises(ValueError, match='checksum'):
        T.read_bundle(prefix)
    assert T.read_bundle(prefix, verify=False)['b'].tolist() == [1, 1, 1]
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[3] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='checksum'):
        T.read_table(prefix + '.index')
    open(prefix + '.index', 'wb').write(bytes(idx[:-1]) + b'\x00')
    with pytest.raises(ValueError, match='magic'):
        T.read_table(prefix + '.index')


def test_snappy_blocks_are_readable():
    # literal "abcd" then a copy of 8 bytes at offset 4 (overlapping) then a 2-byte-offset copy
    comp = bytes([16, (4 - 1) << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4]) + bytes([((4 - 1) << 2) | 2, 12, 0])
    assert T.snappy_decompress(comp) == b'abcd' + b'abcdabcd' + b'abcd'
    with pytest.raises(ValueError):
        T.snappy_decompress(bytes([3, 0]) + b'a')


def test_variable_name_map():
    f = T.tf_variable_name
    assert f('model/image_encoder/encoder/conv_1/w') == 'model/image_encoder/encoder/conv_1/conv_1/w'
    assert f('model/pose_encoder/conv_1/b') == 'model/pose_encoder/conv_1/conv_1/b'
    assert f('model/renderer/conv_7/gamma') == 'model/renderer/conv_7/batch_normalization/gamma'
    assert f('model/renderer/conv_7/moving_variance') == 'model/renderer/conv_7/batch_normalization/moving_variance'
    assert f('loss/conv3_2_agg') == 'SelfSupReconstructionLoss/conv3_2_agg'
    # every trainable / state variable of the default model maps to a distinct name
    from oracle import imm_oracle as O
    P, S = O.init_params(O.default_model_config(10), 128)
    names = [f(k) for k in list(P) + list(S)]
    assert len(set(names)) == len(names)
    assert 'model/pose_encoder/encoder/conv_6/batch_normalization/beta' in names


def test_snappy_against_vectors_of_the_real_library(tmp_path):
    """tests/golden/snappy_golden.npz was produced by libsnappy 1.1.8 (tests/golden/make_snappy_golden.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'snappy_golden.npz'))
    for k in ('text', 'random', 'zeros', 'mixed', 'empty', 'short', 'block'):
        assert T.snappy_decompress(g[k + '_snappy'].tobytes()) == g[k + '_raw'].tobytes(), k
    # a table whose only data block is stored snappy-compressed (type byte 1), assembled by hand around the real bytes
    comp, raw = g['block_snappy'].tobytes(), g['block_raw'].tobytes()
    entries = list(T._block_entries(raw))
    assert len(entries) == 40 and entries[7] == (b'model/renderer/conv_07/batch_normalization/gamma', bytes([7]) * 3)

    def framed(block, ctype):
        return block + bytes([ctype]) + struct.pack('<I', T.mask_crc(T.crc32c(block + bytes([ctype]))))
    meta_block = struct.pack('<II', 0, 1)
    ib = T._BlockBuilder(restart_interval=1)
    ib.add(entries[-1][0], T._put_varint(0) + T._put_varint(len(comp)))
    index_block = ib.finish()
    off_meta = len(comp) + 5
    off_index = off_meta + len(meta_block) + 5
    footer = T._put_varint(off_meta) + T._put_varint(len(meta_block)) + T._put_varint(off_index) + T._put_varint(len(index_block))
    p = tmp_path / 'snappy.index'
    p.write_bytes(framed(comp, 1) + framed(meta_block, 0) + framed(index_block, 0) + footer + b'\x00' * (40 - len(footer))
                  + struct.pack('<Q', T.TABLE_MAGIC))
    assert list(T.read_table(str(p)).items()) == entries
