"""CPU restatement of the reference's thin-plate-spline augmentation (TEST INFRASTRUCTURE ONLY: imported by tests/,
tools/ benchmarks' baseline legs and nothing on the product path).

PARITY PINNED: unlike the TF1 training graph, this part of the reference is plain numpy/torch and imports in the build
container, so tests/golden/make_tps_golden.py ran the reference's own `TPSGridGen`, `sample_tps_w` and
`F.grid_sample` call (imm/utils/tps_sampler.py) and committed inputs/outputs as tests/golden/tps_golden.npz;
tests/test_tps_cpu.py checks every function below against those vectors.

What is restated (reference file:line):
  tps_basis        imm/utils/tps_sampler.py:106-139  (TPSGridGen.__init__: grid, control points, U = d*log d, [U | 1 | x y])
  tps_grid         imm/utils/tps_sampler.py:142-157  (TPSGridGen.forward: L @ W, reshaped to Ho x Wo x 2, last axis (x, y))
  sample_tps_w     imm/utils/tps_sampler.py:161-189  (random non-linear + affine parameters, (Hc*Wc+3) x 2)
  grid_sample      imm/utils/tps_sampler.py:89       (F.grid_sample bilinear / zero padding; the reference pins
                                                      torch 0.4.1 whose behaviour is today's align_corners=True)
  warp             imm/utils/tps_sampler.py:76-99    (TPSRandomSampler.forward with pad=False, as the dataset uses it)
  apply_pair       imm/datasets/tps_dataset.py:70-96 (mask||image -> target warp -> source warp -> split)
"""
import numpy as np

REAL_MIN = 1e-8     # tps_sampler.py:127 clip of the squared distance before the log


def _lin(n):
    return np.linspace(-1.0, 1.0, n)


def tps_basis(ho, wo, hc, wc):
    """[ho*wo, hc*wc + 3] float32: thin-plate kernel of every output grid point against every control point, then a
    column of ones and the point's (x, y).  Grid points and control points are regular grids on [-1, 1]^2, x fastest."""
    gx, gy = np.meshgrid(_lin(wo), _lin(ho))
    grid = np.stack([gx.ravel(), gy.ravel()], axis=1).astype(np.float32)           # N x 2 (x, y)
    cx, cy = np.meshgrid(_lin(wc), _lin(hc))
    ctrl = np.stack([cx.ravel(), cy.ravel()], axis=1).astype(np.float32)           # M x 2
    # squared euclidean distance in float64 of the float32 coordinates (scipy cdist promotes to double)
    d = ((grid[:, None, :].astype(np.float64) - ctrl[None, :, :].astype(np.float64)) ** 2).sum(axis=2)
    d = np.clip(d, REAL_MIN, None)
    u = np.log(d) * d
    basis = np.concatenate([u, np.ones((grid.shape[0], 1)), grid.astype(np.float64)], axis=1)
    return basis.astype(np.float32)


def tps_grid(basis, w, ho, wo):
    """basis [N, M+3] f32, w [B, M+3, 2] f32 -> sampling grid [B, ho, wo, 2] f32 ((x, y) in [-1, 1] coordinates)."""
    w = np.asarray(w, dtype=np.float32)
    g = np.einsum('nm,bmc->bnc', basis.astype(np.float32), w).astype(np.float32)
    return g.reshape(w.shape[0], ho, wo, 2)


def sample_tps_w(hc, wc, warpsd, rotsd, scalesd, transsd, rng=np.random):
    """Random TPS parameters [(hc*wc + 3), 2]: per control point a dense N(0, warpsd[0]) term plus a half-sparse
    N(0, warpsd[1]) term; then the affine rows: translation ~ N(0, transsd), scale 1 + N(0, scalesd), rotation
    N(0, rotsd) degrees.  Draw order follows the reference so that a shared numpy seed gives the same parameters."""
    nc = hc * wc
    keep = (rng.rand(nc, 2) > 0.5).astype(np.float32)
    w = warpsd[0] * rng.randn(nc, 2) + warpsd[1] * (keep * rng.randn(nc, 2))
    rot = np.deg2rad(rng.randn() * rotsd)
    sc = 1.0 + rng.randn() * scalesd
    aff = np.array([[transsd * rng.randn(), transsd * rng.randn()],
                    [sc * np.cos(rot), -sc * np.sin(rot)],
                    [sc * np.sin(rot), sc * np.cos(rot)]])
    return np.concatenate([w, aff], axis=0)


def grid_sample(img, grid, align_corners=True):
    """img [B, H, W, C] f32 (NHWC), grid [B, Ho, Wo, 2] (x, y) in [-1, 1] -> [B, Ho, Wo, C]: bilinear interpolation,
    corners outside the image contribute zero (F.grid_sample, padding_mode='zeros')."""
    img = np.asarray(img, dtype=np.float32)
    b, h, w, c = img.shape
    gx = grid[..., 0].astype(np.float32)
    gy = grid[..., 1].astype(np.float32)
    if align_corners:
        fx = (gx + 1.0) * np.float32(0.5 * (w - 1))
        fy = (gy + 1.0) * np.float32(0.5 * (h - 1))
    else:
        fx = ((gx + 1.0) * np.float32(w) - 1.0) * np.float32(0.5)
        fy = ((gy + 1.0) * np.float32(h) - 1.0) * np.float32(0.5)
    x0 = np.floor(fx); y0 = np.floor(fy)
    ax = (fx - x0).astype(np.float32); ay = (fy - y0).astype(np.float32)
    out = np.zeros((b,) + gx.shape[1:] + (c,), dtype=np.float32)
    bi = np.arange(b)[:, None, None]
    for dy, wy in ((0, 1.0 - ay), (1, ay)):
        for dx, wx in ((0, 1.0 - ax), (1, ax)):
            xi = (x0 + dx).astype(np.int64); yi = (y0 + dy).astype(np.int64)
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            v = img[bi, np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)]
            out += (wy * wx * ok)[..., None].astype(np.float32) * v
    return out


def warp(img, w_tps, hc=10, wc=10, basis=None):
    """One TPS warp of a batch (TPSRandomSampler.forward with pad=False): img [B,H,W,C], w_tps [B, hc*wc+3, 2]."""
    b, h, wd, _c = img.shape
    if basis is None:
        basis = tps_basis(h, wd, hc, wc)
    return grid_sample(img, tps_grid(basis, w_tps, h, wd), align_corners=True)


def warp_pad(img, w_tps, hc=10, wc=10):
    """TPSRandomSampler.forward with pad=True (tps_sampler.py:24-29,89-92), step by step as the reference does it on its
    NCHW tensor: h_pad = H//2, w_pad = W//2; F.pad(x, (h_pad, h_pad, w_pad, w_pad), 'replicate') pads the last dimension
    (columns) by h_pad and the rows by w_pad; the grid is (H + h_pad) x (W + w_pad); the negative pad crops h_pad columns and
    w_pad rows from each side.  img [B,H,W,C] -> [B, H + h_pad - 2 w_pad, W + w_pad - 2 h_pad, C]."""
    img = np.asarray(img, dtype=np.float32)
    _b, h, w, _c = img.shape
    h_pad, w_pad = h // 2, w // 2
    gh, gw = h + h_pad, w + w_pad
    padded = np.pad(img, ((0, 0), (w_pad, w_pad), (h_pad, h_pad), (0, 0)), mode='edge')
    out = grid_sample(padded, tps_grid(tps_basis(gh, gw, hc, wc), w_tps, gh, gw), align_corners=True)
    return out[:, w_pad:gh - w_pad, h_pad:gw - h_pad]


def apply_pair(image, mask, w_target, w_source, hc=10, wc=10):
    """tps_dataset.py:70-96: the mask rides as channel 0 through both warps; future = target(mask||image),
    image = source(future); returns image [B,H,W,3], future_image [B,H,W,3], mask (= the FUTURE image's mask) [B,H,W,1]."""
    x = np.concatenate([mask, image], axis=3).astype(np.float32)
    basis = tps_basis(x.shape[1], x.shape[2], hc, wc)
    future = warp(x, w_target, hc, wc, basis)
    src = warp(future, w_source, hc, wc, basis)
    return {'image': src[..., 1:], 'future_image': future[..., 1:], 'mask': future[..., 0:1]}
