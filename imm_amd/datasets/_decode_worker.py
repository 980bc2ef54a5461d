"""Decode worker process of the input pipeline (started by imm_amd/datasets/impair_dataset.py:DecodeWorkers as
`python _decode_worker.py <ring file> <slot bytes>`; imports nothing of the package: PIL + numpy only, so start-up is
fast and there is no torch / HIP state in the children).

Protocol, one line each way per image (the Python-level work of PIL — header parsing, the chunked decode loop — holds
the GIL, so decode threads in ONE process do not scale; processes do):
    parent -> worker :  "<slot> <channels> <path>\\n"
    worker -> parent :  "<h> <w> <c>\\n"      pixels written u8 HWC at ring[slot * slot_bytes ...]
                        "big <h> <w> <c>\\n"  image does not fit a slot (the parent decodes it itself)
                        "err <message>\\n"
"""
import sys

import numpy as np


def main():
    ring_path, slot_bytes = sys.argv[1], int(sys.argv[2])
    from PIL import Image
    ring = np.memmap(ring_path, dtype=np.uint8, mode='r+')
    out = sys.stdout
    for line in sys.stdin:
        try:
            slot_s, ch_s, path = line.rstrip('\n').split(' ', 2)
            slot, channels = int(slot_s), int(ch_s)
            with Image.open(path) as im:
                im = im.convert('RGB' if channels == 3 else 'L')
                a = np.asarray(im, dtype=np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            h, w, c = a.shape
            if a.size > slot_bytes:
                out.write('big %d %d %d\n' % (h, w, c))
            else:
                ring[slot * slot_bytes: slot * slot_bytes + a.size] = a.reshape(-1)
                out.write('%d %d %d\n' % (h, w, c))
        except Exception as e:                      # reported to the parent, which raises
            out.write('err %s: %s\n' % (type(e).__name__, str(e).replace('\n', ' ')))
        out.flush()


if __name__ == '__main__':
    main()
