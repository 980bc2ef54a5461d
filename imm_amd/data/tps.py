"""Thin-plate-spline augmentation on the GPU — the host side of imm_tps_warp (imm_amd/csrc/tps.hip).

Mirrors the reference interface (imm/utils/tps_sampler.py): `TPSRandomSampler(height, width, vertical_points,
horizontal_points, rotsd, scalesd, transsd, warpsd, cache_size, cache_evict_prob, pad, device)` with the same meaning
of every argument, and the dataset-level pairing of imm/datasets/tps_dataset.py:70-96 (`TPSPairAugmenter`).  Differences
that follow from running on the device: the cache holds TPS PARAMETERS (206 floats) instead of full sampling grids, the
grid is never materialised (the kernel forms it per pixel), and inputs/outputs are NHWC device tensors.  There is no CPU
fallback: without libimm_hip.so the calls raise ImmHipError."""
import random

import numpy as np
import torch

from .. import ops


def tps_basis_t(ho, wo, hc, wc):
    """TPSGridGen's basis matrix (tps_sampler.py:106-139), transposed to [hc*wc + 3, ho*wo] float32: thin-plate kernel
    d*log(d) of the squared distance grid point <-> control point (clipped at 1e-8), a row of ones, the x and the y of
    the grid point; both point sets are regular grids on [-1, 1]^2.  Computed in float64 and rounded once, like the
    reference."""
    gx, gy = np.meshgrid(np.linspace(-1, 1, wo), np.linspace(-1, 1, ho))
    grid = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32).astype(np.float64)
    cx, cy = np.meshgrid(np.linspace(-1, 1, wc), np.linspace(-1, 1, hc))
    ctrl = np.stack([cx.ravel(), cy.ravel()], 1).astype(np.float32).astype(np.float64)
    d = np.clip(((grid[None, :, :] - ctrl[:, None, :]) ** 2).sum(-1), 1e-8, None)      # [M, N]
    rows = np.concatenate([np.log(d) * d, np.ones((1, grid.shape[0])), grid.T], axis=0)
    return np.ascontiguousarray(rows.astype(np.float32))


def sample_tps_w(hc, wc, warpsd, rotsd, scalesd, transsd, rng=np.random):
    """Random TPS parameters [(hc*wc + 3), 2] (tps_sampler.py:161-189): control-point displacements
    N(0, warpsd[0]) + Bernoulli(1/2) * N(0, warpsd[1]); affine rows = translation N(0, transsd), then the rotation
    (N(0, rotsd) degrees) / scale (1 + N(0, scalesd)) matrix.  Same draw order as the reference for a shared seed."""
    n = hc * wc
    keep = (rng.rand(n, 2) > 0.5).astype(np.float32)
    w = warpsd[0] * rng.randn(n, 2) + warpsd[1] * (keep * rng.randn(n, 2))
    rot = np.deg2rad(rng.randn() * rotsd)
    sc = 1.0 + rng.randn() * scalesd
    aff = np.array([[transsd * rng.randn(), transsd * rng.randn()],
                    [sc * np.cos(rot), -sc * np.sin(rot)],
                    [sc * np.sin(rot), sc * np.cos(rot)]])
    return np.concatenate([w, aff], 0)


class TPSParamCache(object):
    """Host side of the reference's sampler (tps_sampler.py:12-74): random TPS parameters with its cache policy; no
    device state, so it can live in a loader process."""

    def __init__(self, vertical_points=10, horizontal_points=10, rotsd=0.0, scalesd=0.0, transsd=0.1,
                 warpsd=(0.001, 0.005), cache_size=1000, cache_evict_prob=0.01, rng=None):
        self.vertical_points, self.horizontal_points = int(vertical_points), int(horizontal_points)
        self.rotsd, self.scalesd, self.transsd, self.warpsd = rotsd, scalesd, transsd, tuple(warpsd)
        self.cache_size, self.cache_evict_prob = int(cache_size), float(cache_evict_prob)
        self.rng = rng
        self.cache = [None] * self.cache_size

    def _sample_w(self):
        return sample_tps_w(self.vertical_points, self.horizontal_points, self.warpsd, self.rotsd, self.scalesd,
                            self.transsd, self.rng if self.rng is not None else np.random).astype(np.float32)

    def sample(self, batch_size):
        ws = []
        for _ in range(batch_size):
            slot = random.randint(0, self.cache_size - 1)
            if self.cache[slot] is None or random.random() < self.cache_evict_prob:
                self.cache[slot] = self._sample_w()
            ws.append(self.cache[slot])
        return np.stack(ws)


class TPSRandomSampler(object):
    """Random TPS warps of NHWC float32 device batches.

    `pad=True` follows the reference's arithmetic to the letter (tps_sampler.py:24-29,89-92): h_pad = H//2, w_pad = W//2;
    the sampling grid is (H + h_pad) x (W + w_pad); `F.pad(input, (h_pad, h_pad, w_pad, w_pad), 'replicate')` on an NCHW
    tensor pads the COLUMNS by h_pad and the ROWS by w_pad; the negative pad afterwards crops the same amounts — so the
    result is (H + h_pad - 2 w_pad) x (W + w_pad - 2 h_pad), e.g. 64x64 from 128x128.  (The datasets construct their
    samplers with pad=False, imm/datasets/tps_dataset.py:35-41.)"""

    def __init__(self, height, width, vertical_points=10, horizontal_points=10, rotsd=0.0, scalesd=0.0, transsd=0.1,
                 warpsd=(0.001, 0.005), cache_size=1000, cache_evict_prob=0.01, pad=True, device='cuda:0', rng=None):
        self.input_height, self.input_width = int(height), int(width)
        self.pad = bool(pad)
        self.h_pad = self.input_height // 2 if pad else 0
        self.w_pad = self.input_width // 2 if pad else 0
        self.height, self.width = self.input_height + self.h_pad, self.input_width + self.w_pad      # the sampling grid
        self.out_height = self.height - 2 * self.w_pad
        self.out_width = self.width - 2 * self.h_pad
        if self.out_height < 1 or self.out_width < 1:
            raise ValueError('TPSRandomSampler(pad=True) on %dx%d leaves no output (the reference crops %d rows / %d columns '
                             'from a %dx%d grid)' % (height, width, 2 * self.w_pad, 2 * self.h_pad, self.height, self.width))
        self.vertical_points, self.horizontal_points = int(vertical_points), int(horizontal_points)
        self.rotsd, self.scalesd, self.transsd, self.warpsd = rotsd, scalesd, transsd, tuple(warpsd)
        self.cache_size, self.cache_evict_prob = int(cache_size), float(cache_evict_prob)
        self.device = torch.device(device)
        self.params = TPSParamCache(self.vertical_points, self.horizontal_points, rotsd, scalesd, transsd, self.warpsd,
                                    self.cache_size, self.cache_evict_prob, rng)
        self.m3 = self.vertical_points * self.horizontal_points + 3
        self.basis_t = ops.to_device_pinned(tps_basis_t(self.height, self.width, self.vertical_points, self.horizontal_points),
                                            self.device)

    def sample_params_host(self, batch_size):
        """[B, M+3, 2] float32 numpy.  Cache policy of the reference (tps_sampler.py:60-74): a random slot per sample,
        refilled when empty or with probability cache_evict_prob."""
        return self.params.sample(batch_size)

    def sample_params(self, batch_size):
        """The same on the device."""
        return ops.to_device_pinned(self.sample_params_host(batch_size), self.device)

    def warp(self, x, w_tps, dst=None, dst_c0=None, dst_rest=None):
        """x [B,H,W,C] float32 NHWC on the device, w_tps [B, M+3, 2].  Writes any of: dst (all channels), dst_c0
        (channel 0, [B,H,W]), dst_rest (channels 1.., [B,H,W,C-1]); allocates and returns dst when none is given."""
        assert x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == self.input_height and x.shape[2] == self.input_width
        assert tuple(w_tps.shape) == (x.shape[0], self.m3, 2), w_tps.shape
        if self.pad:
            assert dst_c0 is None and dst_rest is None, 'pad=True writes one tensor'
            if dst is None:
                dst = torch.empty(x.shape[0], self.out_height, self.out_width, x.shape[3], dtype=torch.float32, device=x.device)
            assert tuple(dst.shape[1:3]) == (self.out_height, self.out_width), dst.shape
            ops.tps_warp_pad(x, self.basis_t, w_tps, (self.w_pad, self.h_pad), (self.height, self.width),
                             (self.w_pad, self.h_pad), dst)
            return dst
        if dst is None and dst_c0 is None and dst_rest is None:
            dst = torch.empty_like(x)
        ops.tps_warp(x, self.basis_t, w_tps, dst, dst_c0, dst_rest)
        return dst

    def forward(self, x):
        return self.warp(x, self.sample_params(x.shape[0]))

    __call__ = forward


class TPSPairAugmenter(object):
    """imm/datasets/tps_dataset.py:70-96 on the device: future = target_warp(mask || image), image = source_warp(future);
    returns / fills `image`, `future_image`, `mask` (= the future image's warped mask) — the three inputs of the training
    step.  Defaults = the dataset's (rotsd [0, 5], scalesd [0, 0.1], transsd [0.1, 0.1], warpsd [.001,.005,.001,.01])."""

    def __init__(self, image_size=(128, 128), vertical_points=10, horizontal_points=10, rotsd=(0.0, 5.0),
                 scalesd=(0.0, 0.1), transsd=(0.1, 0.1), warpsd=(0.001, 0.005, 0.001, 0.01), device='cuda:0', rng=None):
        h, w = int(image_size[1]), int(image_size[0])
        kw = dict(vertical_points=vertical_points, horizontal_points=horizontal_points, pad=False, device=device, rng=rng)
        self.target = TPSRandomSampler(h, w, rotsd=rotsd[0], scalesd=scalesd[0], transsd=transsd[0], warpsd=warpsd[:2], **kw)
        self.source = TPSRandomSampler(h, w, rotsd=rotsd[1], scalesd=scalesd[1], transsd=transsd[1], warpsd=warpsd[2:], **kw)
        self._stack = None

    def stack(self, b):
        """The [B,H,W,4] float32 mask||image staging tensor of this augmenter (channel 0 = mask, 1..3 = image): producers
        (e.g. imm_resize_crop_u8 in imm_amd/datasets) may fill it directly and call warp_stack()."""
        h, w = self.target.height, self.target.width
        if self._stack is None or self._stack.shape[0] != b:
            self._stack = torch.empty(b, h, w, 4, dtype=torch.float32, device=self.target.device)
            self._future = torch.empty_like(self._stack)
        return self._stack

    def warp_stack(self, stack, out_image=None, out_future=None, out_mask=None, w_target=None, w_source=None):
        b, h, w, _ = stack.shape
        assert stack is self._stack, 'warp_stack() takes the tensor returned by stack()'
        wt = self.target.sample_params(b) if w_target is None else w_target
        ws = self.source.sample_params(b) if w_source is None else w_source
        out_image = torch.empty(b, h, w, 3, dtype=torch.float32, device=stack.device) if out_image is None else out_image
        out_future = torch.empty_like(out_image) if out_future is None else out_future
        out_mask = torch.empty(b, h, w, dtype=torch.float32, device=stack.device) if out_mask is None else out_mask
        self.target.warp(stack, wt, dst=self._future, dst_c0=out_mask, dst_rest=out_future)
        self.source.warp(self._future, ws, dst_rest=out_image)
        return {'image': out_image, 'future_image': out_future, 'mask': out_mask}

    def __call__(self, image, mask, out_image=None, out_future=None, out_mask=None, w_target=None, w_source=None):
        """image [B,H,W,3] float32 in [0,255], mask [B,H,W,1] or [B,H,W].  The optional out_* tensors (e.g. the engine's
        in_image / in_future / in_mask) are filled in place."""
        b, h, w, _ = image.shape
        stack = self.stack(b)
        stack[..., 0] = mask.reshape(b, h, w)
        stack[..., 1:] = image
        return self.warp_stack(stack, out_image, out_future, out_mask, w_target, w_source)
