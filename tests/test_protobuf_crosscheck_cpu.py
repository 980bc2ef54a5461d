"""The hand-written protobuf subset of imm_amd/utils/tf_checkpoint.py and tf_events.py against the REAL protobuf runtime
(google.protobuf, present in the image): the message types of tensor_bundle.proto / tensor_shape.proto / versions.proto /
event.proto / summary.proto are declared at run time from their published field numbers and types, and the bytes must agree
in both directions."""
import struct

import numpy as np
import pytest

pb = pytest.importorskip('google.protobuf')
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory   # noqa: E402

from imm_amd.utils import tf_checkpoint as T      # noqa: E402
from imm_amd.utils import tf_events as E          # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    return f


@pytest.fixture(scope='module')
def protos():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = 'imm_test_tf.proto', 'immtf', 'proto3'
    shape = fd.message_type.add(); shape.name = 'TensorShapeProto'
    dim = shape.nested_type.add(); dim.name = 'Dim'
    _field(dim, 'size', 1, F.TYPE_INT64); _field(dim, 'name', 2, F.TYPE_STRING)
    _field(shape, 'dim', 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.immtf.TensorShapeProto.Dim')
    _field(shape, 'unknown_rank', 3, F.TYPE_BOOL)
    ver = fd.message_type.add(); ver.name = 'VersionDef'
    _field(ver, 'producer', 1, F.TYPE_INT32); _field(ver, 'min_consumer', 2, F.TYPE_INT32)
    hdr = fd.message_type.add(); hdr.name = 'BundleHeaderProto'
    _field(hdr, 'num_shards', 1, F.TYPE_INT32); _field(hdr, 'endianness', 2, F.TYPE_INT32)
    _field(hdr, 'version', 3, F.TYPE_MESSAGE, type_name='.immtf.VersionDef')
    ent = fd.message_type.add(); ent.name = 'BundleEntryProto'
    _field(ent, 'dtype', 1, F.TYPE_INT32)                       # an enum upstream: same varint wire format
    _field(ent, 'shape', 2, F.TYPE_MESSAGE, type_name='.immtf.TensorShapeProto')
    _field(ent, 'shard_id', 3, F.TYPE_INT32); _field(ent, 'offset', 4, F.TYPE_INT64); _field(ent, 'size', 5, F.TYPE_INT64)
    _field(ent, 'crc32c', 6, F.TYPE_FIXED32)
    summ = fd.message_type.add(); summ.name = 'Summary'
    img = summ.nested_type.add(); img.name = 'Image'
    _field(img, 'height', 1, F.TYPE_INT32); _field(img, 'width', 2, F.TYPE_INT32); _field(img, 'colorspace', 3, F.TYPE_INT32)
    _field(img, 'encoded_image_string', 4, F.TYPE_BYTES)
    val = summ.nested_type.add(); val.name = 'Value'
    _field(val, 'tag', 1, F.TYPE_STRING); _field(val, 'simple_value', 2, F.TYPE_FLOAT)
    _field(val, 'image', 4, F.TYPE_MESSAGE, type_name='.immtf.Summary.Image')
    _field(summ, 'value', 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.immtf.Summary.Value')
    ev = fd.message_type.add(); ev.name = 'Event'
    _field(ev, 'wall_time', 1, F.TYPE_DOUBLE); _field(ev, 'step', 2, F.TYPE_INT64); _field(ev, 'file_version', 3, F.TYPE_STRING)
    _field(ev, 'summary', 5, F.TYPE_MESSAGE, type_name='.immtf.Summary')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('immtf.' + n))     # noqa: E731
    return {n: get(n) for n in ('BundleHeaderProto', 'BundleEntryProto', 'Event', 'Summary')}


def test_bundle_protos_agree_with_protobuf(protos):
    Entry, Header = protos['BundleEntryProto'], protos['BundleHeaderProto']
    for dtype, shape, shard, offset, size, crc in [(1, (2, 3), 0, 0, 24, 0x12345678), (9, (), 0, 300, 8, 1),
                                                   (1, (3, 3, 320, 256), 0, 123456789, 2949120, 0xfffffffe), (19, (0, 4), 0, 7, 0, 0x80000000)]:
        m = Entry(dtype=dtype, shard_id=shard, offset=offset, size=size, crc32c=crc)
        m.shape.SetInParent()
        for d in shape:
            m.shape.dim.add().size = d
        mine = T.encode_entry(dtype, shape, shard, offset, size, crc)
        if size:                                                  # proto3 omits zero scalars; the writer always emits size
            assert mine == m.SerializeToString(deterministic=True)
        back = Entry(); back.ParseFromString(mine)
        assert (back.dtype, [d.size for d in back.shape.dim], back.offset, back.size, back.crc32c) == (dtype, list(shape), offset, size, crc)
        e = T.decode_entry(m.SerializeToString())
        assert (e['dtype'], e['shape'], e['offset'], e['size']) == (dtype, list(shape), offset, size)
    h = Header(num_shards=1); h.version.producer = 1
    assert T.encode_header() == h.SerializeToString(deterministic=True)
    # a negative dimension (unknown size, -1) is a ten-byte varint on the wire
    m = Entry(dtype=1, size=4); m.shape.dim.add().size = -1
    assert T.decode_entry(m.SerializeToString())['shape'] == [-1]


def test_event_protos_agree_with_protobuf(protos):
    Event = protos['Event']
    ev = Event(wall_time=1234.5, step=70)
    v = ev.summary.value.add(); v.tag = 'train/loss'; v.simple_value = 0.125
    v = ev.summary.value.add(); v.tag = 'train/im'; v.image.height = 2; v.image.width = 3; v.image.colorspace = 3
    v.image.encoded_image_string = b'\x89PNG-not-really'
    mine = E.encode_event(1234.5, step=70, values=[E.scalar_value('train/loss', 0.125)])
    only_scalar = Event(wall_time=1234.5, step=70)
    w = only_scalar.summary.value.add(); w.tag = 'train/loss'; w.simple_value = 0.125
    assert mine == only_scalar.SerializeToString(deterministic=True)
    first = Event(wall_time=2.0, file_version='brain.Event:2')
    assert E.encode_event(2.0, file_version='brain.Event:2') == first.SerializeToString(deterministic=True)
    # image values: parse what the writer produces with the real runtime
    img = (np.arange(2 * 3 * 3) * 9 % 255).astype(np.uint8).reshape(2, 3, 3)
    data = E.encode_event(5.0, step=1, values=[E.image_value('train/im', img)])
    back = Event(); back.ParseFromString(data)
    got = back.summary.value[0]
    assert got.tag == 'train/im' and (got.image.height, got.image.width, got.image.colorspace) == (2, 3, 3)
    assert got.image.encoded_image_string[:8] == b'\x89PNG\r\n\x1a\n'
    # and the reader parses what the real runtime serialises
    rec = E.encode_record(ev.SerializeToString())
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'ev')
        open(p, 'wb').write(rec)
        (e,) = E.read_events(p)
    assert e['step'] == 70 and e['wall_time'] == 1234.5 and e['scalars'] == {'train/loss': 0.125}
    assert e['images']['train/im'] == (2, 3, 3, b'\x89PNG-not-really')
