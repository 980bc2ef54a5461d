"""Independent fp64 numpy LOOP restatement of the TF1 primitives.  TEST INFRASTRUCTURE ONLY.

Purpose: cross-check oracle/imm_oracle.py (which leans on torch's conv / pooling / softmax kernels)
with code that shares nothing with it: explicit index loops written straight from the TF 1.10 op
definitions (S1..S9 in imm_oracle.py's header).  Small shapes only.  PARITY UNPINNED (see
imm_oracle.py): the reference repo has no tests or vectors, and TF cannot be imported here.
"""
import math

import numpy as np


def conv2d_same(x, w, b=None, stride=1):
    """S1.  x [B,H,W,Ci], w [kh,kw,Ci,Co] (tf.nn.conv2d, nn_utils.py:100)."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    Ho, Wo = -(-H // stride), -(-W // stride)
    pt = max((Ho - 1) * stride + kh - H, 0) // 2
    pl = max((Wo - 1) * stride + kw - W, 0) // 2
    y = np.zeros((B, Ho, Wo, Co), np.float64)
    for oy in range(Ho):
        for ox in range(Wo):
            for ky in range(kh):
                iy = oy * stride + ky - pt
                if iy < 0 or iy >= H:
                    continue
                for kx in range(kw):
                    ix = ox * stride + kx - pl
                    if ix < 0 or ix >= W:
                        continue
                    y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) @ w[ky, kx].astype(np.float64)
    if b is not None:
        y += b
    return y


def resize_bilinear(x, oh, ow, align_corners=False):
    """S2/S3 (tf.image.resize_images / resize_bilinear, imm_model.py:175,334)."""
    B, H, W, C = x.shape
    sy = (H - 1) / (oh - 1) if (align_corners and oh > 1) else H / oh
    sx = (W - 1) / (ow - 1) if (align_corners and ow > 1) else W / ow
    y = np.zeros((B, oh, ow, C), np.float64)
    for i in range(oh):
        fy = i * sy
        y0 = int(math.floor(fy)); y1 = min(y0 + 1, H - 1); ly = fy - y0
        for j in range(ow):
            fx = j * sx
            x0 = int(math.floor(fx)); x1 = min(x0 + 1, W - 1); lx = fx - x0
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * lx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * lx
            y[:, i, j] = top + (bot - top) * ly
    return y


def batch_norm_train(x, gamma, beta, eps=1e-3):
    """S4: returns y, mean, biased var, unbiased var."""
    xf = x.reshape(-1, x.shape[-1]).astype(np.float64)
    n = xf.shape[0]
    mean = xf.sum(0) / n
    var = ((xf - mean) ** 2).sum(0) / n
    y = (x - mean) / np.sqrt(var + eps) * gamma + beta
    return y, mean, var, var * n / (n - 1)


def max_pool2(x):
    B, H, W, C = x.shape
    y = np.zeros((B, H // 2, W // 2, C), x.dtype)
    for i in range(H // 2):
        for j in range(W // 2):
            y[:, i, j] = np.maximum(np.maximum(x[:, 2 * i, 2 * j], x[:, 2 * i, 2 * j + 1]),
                                    np.maximum(x[:, 2 * i + 1, 2 * j], x[:, 2 * i + 1, 2 * j + 1]))
    return y


def soft_argmax(heat):
    """imm_model.py:252-264."""
    B, H, W, K = heat.shape
    mu = np.zeros((B, K, 2)); py = np.zeros((B, H, K)); px = np.zeros((B, W, K))
    for b in range(B):
        for k in range(K):
            rows = np.array([heat[b, i, :, k].astype(np.float64).sum() / W for i in range(H)])
            cols = np.array([heat[b, :, j, k].astype(np.float64).sum() / H for j in range(W)])
            for v, n, dst, ax in ((rows, H, py, 0), (cols, W, px, 1)):
                e = np.exp(v - v.max()); p = e / e.sum()
                dst[b, :, k] = p
                mu[b, k, ax] = sum(p[i] * (-1.0 + 2.0 * i / (n - 1)) for i in range(n))
    return mu, py, px


def gaussian_maps(mu, s, inv_std, mode='rot'):
    """imm_model.py:34-78."""
    B, K, _ = mu.shape
    g = np.zeros((B, s, s, K))
    for b in range(B):
        for k in range(K):
            for i in range(s):
                yl = -1.0 + 2.0 * i / (s - 1)
                for j in range(s):
                    xl = -1.0 + 2.0 * j / (s - 1)
                    if mode == 'rot':
                        g[b, i, j, k] = math.exp(-((yl - mu[b, k, 0]) ** 2 + (xl - mu[b, k, 1]) ** 2) * inv_std ** 2)
                    elif mode == 'flat':
                        d = ((yl - mu[b, k, 0]) ** 2 + (xl - mu[b, k, 1]) ** 2) * inv_std ** 2
                        g[b, i, j, k] = math.exp(-(d + 1e-5) ** 0.25)
                    elif mode == 'ankush':
                        gy = math.exp(-math.sqrt(1e-4 + abs((mu[b, k, 0] - yl) * inv_std)))
                        gx = math.exp(-math.sqrt(1e-4 + abs((mu[b, k, 1] - xl) * inv_std)))
                        g[b, i, j, k] = gy * gx
                    else:
                        raise ValueError('Unknown mode: ' + str(mode))
    return g


def clip_by_norm(g, c):
    n = math.sqrt(float((g.astype(np.float64) ** 2).sum()))
    return g * c / max(n, c)


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """TF AdamOptimizer, t = 1-based step index."""
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return p - lr_t * m / (np.sqrt(v) + eps), m, v
