"""CPU restatement of the reference's image-ingest geometry (TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing on
the product path).

PARITY UNPINNED against TF1: the reference evaluates these ops inside tf.data (tf.image.resize_images,
tf.image.decode_jpeg) and TensorFlow cannot be installed here.  `resize_bilinear` restates the published TF1
`ResizeBilinear` CPU kernel with align_corners=True; tests/test_datasets_cpu.py cross-checks it against torch's
independent `F.interpolate(mode='bilinear', align_corners=True)` (same mapping, different operation order: agreement to
float rounding, not bitwise).

What is restated (reference file:line):
  resize_bilinear   imm/datasets/celeba_dataset.py:160-161, aflw_dataset.py:100-102 (tf.image.resize_images, BILINEAR,
                    align_corners=True on the float image)
  celeba_image      imm/datasets/celeba_dataset.py:136-174 (to_float, resize to round(size/0.8), central crop)
  aflw_image        imm/datasets/aflw_dataset.py:81-114     (to_float, resize to size)
  smooth_mask       imm/datasets/tps_dataset.py:47-67
  resize_points     imm/datasets/impair_dataset.py:116-123
"""
import numpy as np


def resize_bilinear(img, out_h, out_w):
    """img [H,W,C] (u8 or f32) -> [out_h,out_w,C] f32.  scale = (in-1)/(out-1) (in/out when out == 1); per output
    coordinate lo = floor(o*scale), hi = min(ceil(o*scale), in-1), t = o*scale - lo; top/bottom row lerp in x, then lerp
    in y; all in float32."""
    img = np.asarray(img).astype(np.float32)
    ih, iw = img.shape[:2]

    def taps(n_in, n_out):
        scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(n_in) / np.float32(n_out)
        s = np.arange(n_out, dtype=np.float32) * scale
        lo = np.floor(s).astype(np.int64)
        hi = np.minimum(np.ceil(s).astype(np.int64), n_in - 1)
        return lo, hi, (s - lo.astype(np.float32)).astype(np.float32)

    yl, yh, ty = taps(ih, out_h)
    xl, xh, tx = taps(iw, out_w)
    tx = tx[None, :, None]
    ty = ty[:, None, None]
    tl, tr = img[yl][:, xl], img[yl][:, xh]
    bl, br = img[yh][:, xl], img[yh][:, xh]
    top = tl + (tr - tl) * tx
    bot = bl + (br - bl) * tx
    return (top + (bot - top) * ty).astype(np.float32)


def celeba_geometry(final):
    resize = int(np.round(final / 0.8))
    return resize, int(np.round((resize - final) / 2.0))


def celeba_image(img_u8, final=128):
    resize, margin = celeba_geometry(final)
    return resize_bilinear(img_u8, resize, resize)[margin:margin + final, margin:margin + final]


def aflw_image(img_u8, final=128):
    return resize_bilinear(img_u8, final, final)


def smooth_mask(h, w, margin=10, step=20, b=0.4):
    def sstep(n, bb):
        x = np.linspace(-1.0, 1.0, n).astype(np.float32)
        return (0.5 + 0.5 * np.tanh(x / np.float32(bb))).astype(np.float32)

    def strip(size):
        return np.concatenate([np.zeros(margin, np.float32), sstep(step, b), np.ones(size - 2 * margin - 2 * step, np.float32),
                               sstep(step, -b), np.zeros(margin, np.float32)])
    return strip(h)[:, None] * strip(w)[None]


def resize_points(points, size, new_size):
    ratio = np.asarray(new_size, np.float32) / np.asarray(size, np.float32)
    return np.asarray(points, np.float32) * ratio[None]
