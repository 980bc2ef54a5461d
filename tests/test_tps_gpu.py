"""TPS augmentation on the GPU (imm_tps_warp through the C-ABI) against the numpy oracle, which is itself pinned to
the reference's own outputs (tests/test_tps_cpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import tps_oracle as T

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tps_golden.npz'))


@pytest.fixture(scope='module')
def tps():
    from imm_amd.data import tps as M
    return M


def test_basis_is_the_reference_matrix(tps):
    np.testing.assert_array_equal(tps.tps_basis_t(12, 20, 3, 4).T, G['basis_12x20_3x4'])


def test_warp_matches_reference_vectors(tps):
    """Same inputs as the golden run of the reference's TPSGridGen + F.grid_sample."""
    rng = np.random.RandomState(int(G['img_seed'][0]))
    img = (rng.rand(3, 128, 128, 4) * 255).astype(np.float32)
    img[..., 0] = rng.rand(3, 128, 128)
    s = tps.TPSRandomSampler(128, 128, pad=False, device=DEV)
    x = torch.from_numpy(img).to(DEV)
    fut = s.warp(x, torch.from_numpy(G['w_target'].astype(np.float32)).to(DEV))
    src = s.warp(fut, torch.from_numpy(G['w_source'].astype(np.float32)).to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(fut.cpu().numpy()[:, ::5, ::3], G['future_sub'], rtol=0, atol=3e-3)   # values up to 255
    np.testing.assert_allclose(src.cpu().numpy()[:, ::5, ::3], G['source_sub'], rtol=0, atol=6e-3)


@pytest.mark.parametrize('B,H,W,C,hc,wc', [(1, 16, 16, 1, 3, 3), (5, 32, 48, 4, 4, 5), (11, 64, 64, 3, 10, 10), (32, 128, 128, 4, 10, 10)],
                         ids=['tiny', 'rect', 'b11_c3', 'dataset_shape'])
def test_warp_vs_oracle(tps, B, H, W, C, hc, wc):
    rng = np.random.RandomState(B)
    img = (rng.rand(B, H, W, C) * 255).astype(np.float32)
    # strong warps so that many samples leave the image (zero padding) and the affine part is exercised
    w = np.stack([T.sample_tps_w(hc, wc, (0.01, 0.05), 20.0, 0.3, 0.4, rng) for _ in range(B)]).astype(np.float32)
    ref = T.warp(img, w, hc, wc)
    s = tps.TPSRandomSampler(H, W, hc, wc, pad=False, device=DEV)
    x = torch.from_numpy(img).to(DEV)
    wd = torch.from_numpy(w).to(DEV)
    full = torch.full((B, H, W, C), float('nan'), device=DEV)
    c0 = torch.full((B, H, W), float('nan'), device=DEV)
    rest = torch.full((B, H, W, max(C - 1, 1)), float('nan'), device=DEV)
    s.warp(x, wd, dst=full, dst_c0=c0, dst_rest=rest if C > 1 else None)
    torch.cuda.synchronize()
    got = full.cpu().numpy()
    # bilinear weights magnify the f32 summation-order difference of the grid near pixel boundaries: a few 1e-3 on 0..255
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-2)
    assert float(np.mean(np.abs(got - ref))) < 2e-4
    assert float((ref == 0).mean()) > 0.01 or B == 1, 'test should exercise the zero padding'
    np.testing.assert_array_equal(c0.cpu().numpy(), got[..., 0])
    if C > 1:
        np.testing.assert_array_equal(rest.cpu().numpy(), got[..., 1:])


def test_identity_parameters_reproduce_the_image(tps):
    B, S = 2, 64
    s = tps.TPSRandomSampler(S, S, pad=False, device=DEV)
    w = np.zeros((B, s.m3, 2), np.float32)
    w[:, -2, 0] = 1.0; w[:, -1, 1] = 1.0          # x' = x, y' = y
    img = torch.rand(B, S, S, 4, device=DEV) * 255
    out = s.warp(img, torch.from_numpy(w).to(DEV))
    torch.cuda.synchronize()
    assert float((out - img).abs().max()) < 2e-3


def test_pair_augmenter_fills_engine_inputs(tps):
    """tps_dataset.py:70-96 pairing; outputs land in caller-provided buffers (the training step's inputs)."""
    from oracle import imm_oracle as O
    B, S = 4, 128
    rng = np.random.RandomState(3)
    image = (rng.rand(B, S, S, 3) * 255).astype(np.float32)
    mask = np.broadcast_to(O.smooth_mask(S, S).numpy().reshape(1, S, S, 1), (B, S, S, 1)).astype(np.float32)
    aug = tps.TPSPairAugmenter((S, S), device=DEV, rng=np.random.RandomState(11))
    wt = aug.target.sample_params(B); ws = aug.source.sample_params(B)
    bufs = [torch.full((B, S, S, 3), float('nan'), device=DEV), torch.full((B, S, S, 3), float('nan'), device=DEV),
            torch.full((B, S, S), float('nan'), device=DEV)]
    out = aug(torch.from_numpy(image).to(DEV), torch.from_numpy(mask).to(DEV), bufs[0], bufs[1], bufs[2], wt, ws)
    torch.cuda.synchronize()
    ref = T.apply_pair(image, mask, wt.cpu().numpy(), ws.cpu().numpy())
    for k, tol in (('future_image', 2e-2), ('image', 4e-2)):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=0, atol=tol)
    np.testing.assert_allclose(out['mask'].cpu().numpy(), ref['mask'][..., 0], rtol=0, atol=1e-4)
    assert out['image'] is bufs[0] and out['future_image'] is bufs[1] and out['mask'] is bufs[2]


def test_padded_warp_matches_reference_vectors(tps):
    """pad=True against the reference's own TPSRandomSampler(pad=True).forward_py (tests/golden/make_tps_golden.py 3b)."""
    s = tps.TPSRandomSampler(24, 20, 4, 5, pad=True, device=DEV)
    assert (s.out_height, s.out_width) == (16, 6)
    got = s.warp(torch.from_numpy(G['pad_img']).to(DEV), torch.from_numpy(G['pad_w'].astype(np.float32)).to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), G['pad_out'], rtol=0, atol=3e-3)


@pytest.mark.parametrize('B,H,W,C,hc,wc', [(3, 16, 16, 1, 3, 3), (9, 40, 32, 4, 4, 5), (32, 128, 128, 4, 10, 10)], ids=['tiny', 'rect', 'dataset_shape'])
def test_padded_warp_vs_oracle(tps, B, H, W, C, hc, wc):
    rng = np.random.RandomState(100 + B)
    img = (rng.rand(B, H, W, C) * 255).astype(np.float32)
    w = np.stack([T.sample_tps_w(hc, wc, (0.01, 0.05), 20.0, 0.3, 0.8, rng) for _ in range(B)]).astype(np.float32)
    ref = T.warp_pad(img, w, hc, wc)
    s = tps.TPSRandomSampler(H, W, hc, wc, pad=True, device=DEV)
    got = s.warp(torch.from_numpy(img).to(DEV), torch.from_numpy(w).to(DEV))
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == ref.shape == (B, H + H // 2 - 2 * (W // 2), W + W // 2 - 2 * (H // 2), C)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-2)
    assert float(np.mean(np.abs(got - ref))) < 3e-4


def test_sampler_errors_and_cache(tps):
    with pytest.raises(ValueError):
        tps.TPSRandomSampler(16, 64, pad=True, device=DEV)                  # 16 + 8 - 64 rows: nothing left
    s = tps.TPSRandomSampler(16, 16, pad=False, device=DEV, cache_size=2, cache_evict_prob=0.0, rng=np.random.RandomState(0))
    a = s.sample_params(64)
    assert len({a[i].cpu().numpy().tobytes() for i in range(64)}) <= 2       # only two cached parameter sets are reused


def test_cpu_restatement_rate(tps, capsys):
    """Reported, not asserted: the numpy restatement's rate beside the kernel's (tools/bench_tps.py times the kernel)."""
    import time
    S = 128
    rng = np.random.RandomState(0)
    img = (rng.rand(4, S, S, 3) * 255).astype(np.float32)
    msk = rng.rand(4, S, S, 1).astype(np.float32)
    w1 = np.stack([T.sample_tps_w(10, 10, (0.001, 0.005), 0.0, 0.0, 0.1, rng) for _ in range(4)]).astype(np.float32)
    w2 = np.stack([T.sample_tps_w(10, 10, (0.001, 0.01), 5.0, 0.1, 0.1, rng) for _ in range(4)]).astype(np.float32)
    t0 = time.time()
    T.apply_pair(img, msk, w1, w2)
    dt = time.time() - t0
    with capsys.disabled():
        print('\n[tps] numpy restatement: %.1f ms per 4 pairs = %.0f pairs/s' % (dt * 1e3, 4 / dt))
