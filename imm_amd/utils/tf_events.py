"""TensorBoard event files without TensorFlow — the on-disk form of the reference's `tf.summary.FileWriter`
(imm/train/cnn_train_multi.py:436,447-452,489,504-508; scalars of base_model.py:52-60 and scripts/train.py:111, images of
imm_model.py:456-468).

Format (tensorflow/core/lib/io/record_writer + util/event.proto, summary.proto), PARITY UNPINNED (no TensorFlow here;
checked by byte-level known answers and a reader of its own in tests/test_tf_events_cpu.py):
  file    events.out.tfevents.<unix time>.<host>          a sequence of TFRecords
  record  u64 length | u32 masked_crc32c(length bytes) | data | u32 masked_crc32c(data)
  data    Event { 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary }
          Summary { repeated 1: Value { 1: string tag, 2: float simple_value | 4: Image {1: height, 2: width,
                    3: colorspace, 4: bytes encoded_image_string (PNG)} } }
The first record carries file_version 'brain.Event:2'."""
import io
import os
import socket
import struct
import time

import numpy as np

from .tf_checkpoint import _get_varint, _pb_bytes, _pb_fields, _pb_varint, _put_varint, crc32c, mask_crc, unmask_crc


def _pb_double(field, v):
    return _put_varint((field << 3) | 1) + struct.pack('<d', float(v))


def _pb_float(field, v):
    return _put_varint((field << 3) | 5) + struct.pack('<f', float(v))


def encode_record(data):
    head = struct.pack('<Q', len(data))
    return head + struct.pack('<I', mask_crc(crc32c(head))) + data + struct.pack('<I', mask_crc(crc32c(data)))


def scalar_value(tag, value):
    return _pb_bytes(1, _pb_bytes(1, tag.encode()) + _pb_float(2, value))


def image_value(tag, image_u8):
    """image_u8: HxWx{1,3,4} uint8 -> Summary.Value with a PNG-encoded image."""
    from PIL import Image
    a = np.asarray(image_u8, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    buf = io.BytesIO()
    Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).save(buf, format='PNG')
    img = _pb_varint(1, a.shape[0]) + _pb_varint(2, a.shape[1]) + _pb_varint(3, a.shape[2]) + _pb_bytes(4, buf.getvalue())
    return _pb_bytes(1, _pb_bytes(1, tag.encode()) + _pb_bytes(4, img))


def encode_event(wall_time, step=None, file_version=None, values=None):
    ev = _pb_double(1, wall_time)
    if step is not None:
        ev += _pb_varint(2, int(step))
    if file_version is not None:
        ev += _pb_bytes(3, file_version.encode())
    if values:
        ev += _pb_bytes(5, b''.join(values))
    return ev


class EventFileWriter(object):
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'ab')
        self._f.write(encode_record(encode_event(time.time(), file_version='brain.Event:2')))

    def add_scalars(self, scalars, step, images=None):
        """scalars: {tag: float}; images: {tag: HxWxC uint8 array} (optional)."""
        values = [scalar_value(k, v) for k, v in scalars.items()]
        values += [image_value(k, v) for k, v in (images or {}).items()]
        self._f.write(encode_record(encode_event(time.time(), step=step, values=values)))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path, verify=True):
    """-> list of dicts {wall_time, step, file_version, scalars {tag: float}, images {tag: (h, w, c, png bytes)}}."""
    out = []
    with open(path, 'rb') as f:
        raw = f.read()
    pos = 0
    while pos < len(raw):
        (n,) = struct.unpack_from('<Q', raw, pos)
        if verify and unmask_crc(struct.unpack_from('<I', raw, pos + 8)[0]) != crc32c(raw[pos:pos + 8]):
            raise ValueError('event record at %d: length checksum mismatch' % pos)
        data = raw[pos + 12:pos + 12 + n]
        if verify and unmask_crc(struct.unpack_from('<I', raw, pos + 12 + n)[0]) != crc32c(data):
            raise ValueError('event record at %d: data checksum mismatch' % pos)
        pos += 12 + n + 4
        ev = {'wall_time': None, 'step': 0, 'file_version': None, 'scalars': {}, 'images': {}}
        for field, wt, v in _pb_fields(data):
            if field == 1 and wt == 1:
                ev['wall_time'] = struct.unpack('<d', struct.pack('<Q', v))[0]
            elif field == 2:
                ev['step'] = v
            elif field == 3:
                ev['file_version'] = v.decode()
            elif field == 5:
                for f2, _w2, val in _pb_fields(v):
                    if f2 != 1:
                        continue
                    tag, simple, image = None, None, None
                    for f3, w3, x in _pb_fields(val):
                        if f3 == 1:
                            tag = x.decode()
                        elif f3 == 2 and w3 == 5:
                            simple = struct.unpack('<f', struct.pack('<I', x))[0]
                        elif f3 == 4:
                            d = {f4: x4 for f4, _w4, x4 in _pb_fields(x)}
                            image = (d.get(1), d.get(2), d.get(3), d.get(4))
                    if simple is not None:
                        ev['scalars'][tag] = simple
                    if image is not None:
                        ev['images'][tag] = image
        out.append(ev)
    return out
