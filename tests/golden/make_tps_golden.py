"""Generates tests/golden/tps_golden.npz by running the REFERENCE's own TPS code (imm/utils/tps_sampler.py, which is
plain numpy/torch and imports in the build container) on seeded inputs.  Run once in the build container:
    python tests/golden/make_tps_golden.py
The reference pins torch 0.4.1, whose F.grid_sample behaves like today's align_corners=True; the
script therefore wraps F.grid_sample to pass align_corners=True while the reference code runs.  Only inputs and
outputs are stored (data, not source)."""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

REF = '/root/reference'
sys.path.insert(0, REF)
warnings.simplefilter('ignore')
from imm.utils import tps_sampler as T   # noqa: E402

_orig = F.grid_sample


def _gs(inp, grid, *a, **k):
    k.setdefault('align_corners', True)
    return _orig(inp, grid, *a, **k)


def main():
    F.grid_sample = _gs
    out = {}
    # (1) basis matrix of a small configuration, complete
    g_small = T.TPSGridGen(12, 20, 3, 4)
    out['basis_12x20_3x4'] = g_small._L.numpy()
    # (2) the dataset configuration (128x128, 10x10 control points): parameters from the reference sampler under a numpy
    # seed, grid and warped output sub-sampled to keep the fixture small
    np.random.seed(123)
    w_t = np.stack([T.sample_tps_w(10, 10, (0.001, 0.005), 0.0, 0.0, 0.1) for _ in range(3)])
    w_s = np.stack([T.sample_tps_w(10, 10, (0.001, 0.01), 5.0, 0.1, 0.1) for _ in range(3)])
    out['w_target'] = w_t; out['w_source'] = w_s
    gen = T.TPSGridGen(128, 128, 10, 10)
    grid_t = gen(torch.from_numpy(w_t.astype(np.float32))).numpy()
    out['basis_128_rows'] = gen._L.numpy()[::997]               # 17 rows of the 16384 x 103 basis
    out['grid_target_sub'] = grid_t[:, ::9, ::7]
    rng = np.random.RandomState(7)
    img = (rng.rand(3, 128, 128, 4) * 255).astype(np.float32)
    img[..., 0] = rng.rand(3, 128, 128)                          # channel 0 plays the mask
    out['img_seed'] = np.array([7])
    x = torch.from_numpy(img).permute(0, 3, 1, 2)
    fut = F.grid_sample(x, torch.from_numpy(grid_t))
    grid_s = gen(torch.from_numpy(w_s.astype(np.float32)))
    src = F.grid_sample(fut, grid_s)
    out['future_sub'] = fut.permute(0, 2, 3, 1).numpy()[:, ::5, ::3]
    out['source_sub'] = src.permute(0, 2, 3, 1).numpy()[:, ::5, ::3]
    # (3) the sampler class end to end (pad=False as the dataset builds it), grids forced to the ones above
    smp = T.TPSRandomSampler(128, 128, rotsd=0.0, scalesd=0.0, transsd=0.1, warpsd=(0.001, 0.005), pad=False,
                             cache_size=1, cache_evict_prob=0.0)
    smp.cache[0] = torch.from_numpy(grid_t[:1])
    y = smp.forward_py(img[:1])
    out['sampler_forward_py_sub'] = y[:, ::5, ::3]
    # (3b) pad=True (the constructor default; not used by the datasets): non-square so that the reference's swapped
    # paddings show; 24x20 -> grid 36x30, output 16x6
    np.random.seed(11)
    w_p = np.stack([T.sample_tps_w(4, 5, (0.001, 0.01), 5.0, 0.1, 0.1) for _ in range(2)])
    smp = T.TPSRandomSampler(24, 20, vertical_points=4, horizontal_points=5, pad=True, cache_size=1, cache_evict_prob=0.0)
    img_p = (np.random.RandomState(8).rand(2, 24, 20, 3) * 255).astype(np.float32)
    ys = []
    for i in range(2):
        smp.cache[0] = smp.tps(torch.from_numpy(w_p[i:i + 1].astype(np.float32)))
        ys.append(smp.forward_py(img_p[i:i + 1]))
    out['pad_w'] = w_p; out['pad_img'] = img_p; out['pad_out'] = np.concatenate(ys)
    # (4) sample_tps_w under a fixed seed (draw order)
    np.random.seed(5)
    out['w_seed5'] = T.sample_tps_w(4, 3, (0.01, 0.02), 10.0, 0.2, 0.3)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tps_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items()}, os.path.getsize(path))


if __name__ == '__main__':
    main()
