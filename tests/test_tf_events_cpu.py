"""TensorBoard event files (imm_amd/utils/tf_events.py): byte-level known answers from the published record / proto
formats, a write -> read round trip, corruption detection, and the train loop's SummaryWriter on top of it."""
import io
import struct

import numpy as np
import pytest

from imm_amd.utils import tf_events as E
from imm_amd.utils.tf_checkpoint import crc32c, mask_crc


def test_record_and_event_bytes():
    ev = E.encode_event(1.5, step=7, values=[E.scalar_value('loss', 0.25)])
    want = (bytes([0x09]) + struct.pack('<d', 1.5)                 # 1: double wall_time
            + bytes([0x10, 0x07])                                  # 2: step
            + bytes([0x2a, 0x0d,                                   # 5: Summary, 13 bytes
                     0x0a, 0x0b,                                   #   1: Value, 11 bytes
                     0x0a, 0x04]) + b'loss'                        #     1: tag
            + bytes([0x15]) + struct.pack('<f', 0.25))             #     2: float simple_value
    assert ev == want
    rec = E.encode_record(ev)
    head = struct.pack('<Q', len(ev))
    assert rec[:8] == head and rec[8:12] == struct.pack('<I', mask_crc(crc32c(head)))
    assert rec[12:-4] == ev and rec[-4:] == struct.pack('<I', mask_crc(crc32c(ev)))
    first = E.encode_event(2.0, file_version='brain.Event:2')
    assert first == bytes([0x09]) + struct.pack('<d', 2.0) + bytes([0x1a, 0x0d]) + b'brain.Event:2'


def test_writer_round_trip_with_images(tmp_path):
    w = E.EventFileWriter(str(tmp_path))
    img = (np.arange(6 * 5 * 3) % 255).astype(np.uint8).reshape(6, 5, 3)
    w.add_scalars({'train/loss': 12.5, 'train/lr': 1e-3}, 0, images={'train/future_im': img})
    w.add_scalars({'test/loss': 3.0}, 10)
    w.close()
    evs = E.read_events(w.path)
    assert evs[0]['file_version'] == 'brain.Event:2' and len(evs) == 3
    assert evs[1]['step'] == 0 and evs[1]['scalars']['train/loss'] == 12.5 and abs(evs[1]['scalars']['train/lr'] - 1e-3) < 1e-9
    h, wd, c, png = evs[1]['images']['train/future_im']
    assert (h, wd, c) == (6, 5, 3)
    from PIL import Image
    np.testing.assert_array_equal(np.asarray(Image.open(io.BytesIO(png))), img)
    assert evs[2]['step'] == 10 and evs[2]['scalars'] == {'test/loss': 3.0}
    raw = bytearray(open(w.path, 'rb').read())
    raw[-6] ^= 1
    open(w.path, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        E.read_events(w.path)


def test_summary_writer_writes_jsonl_and_events(tmp_path):
    import glob
    import json
    from imm_amd.train.cnn_train_multi import SummaryWriter
    sw = SummaryWriter(str(tmp_path))
    sw.add_summary({'tag': 'train', 'loss': 5.0, 'lr': 1e-3, 'loss_terms': [1.0, 2.0], 'examples_per_sec': 100.0}, 20)
    sw.add_summary({'tag': 'test', 'loss': 4.0, 'n_samples': 6}, 20)
    sw.flush(); sw.close()
    recs = [json.loads(l) for l in open(tmp_path / 'summaries.jsonl')]
    assert [r['tag'] for r in recs] == ['train', 'test'] and recs[0]['step'] == 20
    (path,) = glob.glob(str(tmp_path / 'events.out.tfevents.*'))
    evs = E.read_events(path)
    assert evs[1]['scalars'] == {'train/loss': 5.0, 'train/lr': np.float32(1e-3), 'train/loss_terms/0': 1.0,
                                 'train/loss_terms/1': 2.0, 'train/examples_per_sec': 100.0}
    assert evs[2]['scalars'] == {'test/loss': 4.0, 'test/n_samples': 6.0} and evs[2]['step'] == 20
