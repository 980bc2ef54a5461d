"""Landmark overlays without matplotlib (counterpart of imm/utils/plot_landmarks.py, which scatters markers on a
matplotlib axis): each landmark index gets a (colour, marker shape) style — 8 colours x 7 shapes like the reference's
style table — drawn with PIL onto an RGB image."""
import numpy as np

# eight well separated colours (the reference samples matplotlib's 'Dark2' map at 8 points)
COLORS = [(27, 158, 119), (217, 95, 2), (117, 112, 179), (231, 41, 138), (102, 166, 30), (230, 171, 2), (166, 118, 29),
          (102, 102, 102)]
MARKERS = ['v', 'o', 's', 'd', '^', 'x', '+']


def get_marker_style(i):
    """(rgb, marker) of landmark i: colour cycles fastest, the shape changes every 8 landmarks (plot_landmarks.py:9-18)."""
    max_i = len(COLORS) * len(MARKERS) - 1
    if i > max_i:
        raise ValueError('Exceeded maximum (' + str(max_i) + ') index for styles.')
    return COLORS[i % len(COLORS)], MARKERS[i // len(COLORS)]


def single_marker_style(color, marker):
    return lambda _: (color, marker)


def plot_landmark(draw, landmark, k, size=2.5, style_fn=None, scale=1.0):
    """landmark = (y, x) in pixels of the unscaled image; `draw` a PIL.ImageDraw of the (scale x) enlarged image."""
    c, m = get_marker_style(k) if style_fn is None else style_fn(k)
    y, x = float(landmark[0]) * scale, float(landmark[1]) * scale
    r = size * scale
    if m == 'o':
        draw.ellipse([x - r, y - r, x + r, y + r], fill=c, outline=(255, 255, 255))
    elif m == 's':
        draw.rectangle([x - r, y - r, x + r, y + r], fill=c, outline=(255, 255, 255))
    elif m == 'd':
        draw.polygon([(x, y - r), (x + r, y), (x, y + r), (x - r, y)], fill=c, outline=(255, 255, 255))
    elif m == 'v':
        draw.polygon([(x - r, y - r), (x + r, y - r), (x, y + r)], fill=c, outline=(255, 255, 255))
    elif m == '^':
        draw.polygon([(x - r, y + r), (x + r, y + r), (x, y - r)], fill=c, outline=(255, 255, 255))
    elif m == 'x':
        w = max(1, int(round(0.5 * scale)))
        draw.line([x - r, y - r, x + r, y + r], fill=c, width=w); draw.line([x - r, y + r, x + r, y - r], fill=c, width=w)
    else:
        w = max(1, int(round(0.5 * scale)))
        draw.line([x - r, y, x + r, y], fill=c, width=w); draw.line([x, y - r, x, y + r], fill=c, width=w)


def plot_landmarks(image, landmarks, size=2.5, style_fn=None, scale=3):
    """image: HxWx3 array (0..255); landmarks [K,2] (y, x) pixels.  Returns a PIL image `scale` times larger with the
    markers drawn."""
    from PIL import Image, ImageDraw
    im = Image.fromarray(np.clip(np.asarray(image), 0, 255).astype(np.uint8))
    if scale != 1:
        im = im.resize((im.width * scale, im.height * scale), Image.BILINEAR)
    draw = ImageDraw.Draw(im)
    for k, lm in enumerate(landmarks):
        plot_landmark(draw, lm, k, size=size, style_fn=style_fn, scale=scale)
    return im
