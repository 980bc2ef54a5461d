"""Writes tests/golden/snappy_golden.npz with the REAL snappy library (libsnappy 1.1.8 found under /opt/conda/lib in the build
image; its C API snappy_compress): pairs (raw, compressed) that tests/test_tf_checkpoint_cpu.py feeds to the pure-Python
snappy_decompress of imm_amd/utils/tf_checkpoint.py (TensorFlow/LevelDB table blocks may be snappy-compressed).
Usage: python tests/golden/make_snappy_golden.py"""
import ctypes as C
import os

import numpy as np

lib = C.CDLL('/opt/conda/lib/libsnappy.so.1')
lib.snappy_max_compressed_length.restype = C.c_size_t
lib.snappy_max_compressed_length.argtypes = [C.c_size_t]
lib.snappy_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t)]


def compress(raw):
    n = C.c_size_t(lib.snappy_max_compressed_length(len(raw)))
    out = C.create_string_buffer(n.value)
    assert lib.snappy_compress(raw, len(raw), out, C.byref(n)) == 0
    return out.raw[:n.value]


rng = np.random.RandomState(0)
cases = {
    'text': b'model/image_encoder/encoder/conv_1/batch_normalization/moving_variance ' * 40,
    'random': rng.randint(0, 256, size=3000).astype(np.uint8).tobytes(),
    'zeros': b'\x00' * 70000,                                             # long overlapping copies, > 64 KiB (two snappy blocks)
    'mixed': b''.join((b'w/Adam_%d' % i) * (i % 7 + 1) + rng.randint(0, 256, size=i % 13).astype(np.uint8).tobytes() for i in range(400)),
    'empty': b'',
    'short': b'ab',
}
# a table data block as the index writer lays it out (prefix-compressed keys + restart array), to test a snappy-compressed
# block inside a table file
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imm_amd.utils.tf_checkpoint import _BlockBuilder      # noqa: E402
bb = _BlockBuilder()
for i in range(40):
    bb.add(('model/renderer/conv_%02d/batch_normalization/gamma' % i).encode(), bytes([i]) * (i % 5 + 1))
cases['block'] = bb.finish()
out = {}
for k, raw in cases.items():
    out[k + '_raw'] = np.frombuffer(raw, np.uint8)
    out[k + '_snappy'] = np.frombuffer(compress(raw), np.uint8)
    print(k, len(raw), '->', len(out[k + '_snappy']))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'snappy_golden.npz'), **out)
