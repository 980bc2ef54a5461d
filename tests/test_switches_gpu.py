"""Every IMM_* environment switch the product path still reads, exercised in both positions: the
kernel-dispatch ablation list IMM_CONV_DISABLE (each specialised kernel family off -> the layer falls back to the next more
general kernel, same numbers to accumulation order), IMM_NOL, IMM_CONV_FIRST, IMM_VGG_HEAD, IMM_SSE_ALL, IMM_BN_DIRECT_ROWS, IMM_HALO2X, IMM_TWO_STREAMS, IMM_VGG_SPLIT / IMM_GT_CUS / IMM_WG_CUS / IMM_WG_LANES / IMM_SSE_INPUT_LANE (lane experiments of round 6), IMM_DEBUG_STAMPS, IMM_DEBUG_SKIP_TAGS.  The switches are
read once per process (static dispatch tables / engine construction), hence the child processes (tests/_switch_probe.py).
IMM_DP_BUCKETS, IMM_RCCL_NATIVE and IMM_RCCL_GRAPH have their tests in test_dp_gpu.py / test_step_gpu.py; IMM_HIP_LIB and
IMM_HIPCC_FLAGS (A/B builds) in test_host_cpu.py."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def probe(**env):
    e = {k: v for k, v in os.environ.items() if not k.startswith('IMM_')}
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_switch_probe.py')], env=e, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=280)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith('PROBE ')][-1]
    return json.loads(line[6:])


@pytest.fixture(scope='module')
def base():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return probe()


def same(a, b, rel):
    # the landmarks of a freshly initialised model lie near 0 (their abs-sum is an ill-conditioned yardstick) and the two Adam
    # updates before them are sign-like (|dw| ~ lr whatever the gradient's size): 5e-3 there
    return all(abs(a[k] - b[k]) <= (5e-3 if k == 'mu_abs_sum' else rel) * abs(a[k])
               for k in ('loss0', 'loss1', 'mu_abs_sum', 'params_abs_sum'))


@pytest.mark.timeout(300)
@pytest.mark.parametrize('names', ['halo,halo2,hdeep,deepk', 'group,group32', 'wgrad_tr,wgrad_halo', 's2d', 's2d,s2d_halo,group,group32',
                                   'hdeep6', 's2f', 'hdeep6,s2f'])
def test_conv_disable_falls_back_to_the_general_kernels(base, names):
    """IMM_CONV_DISABLE: with the specialised kernels out of the dispatch every layer runs on the im2col / generic kernels —
    the same step to accumulation order (loss 1e-4, parameters after two updates 1e-4 of their abs-sum)."""
    got = probe(IMM_CONV_DISABLE=names)
    assert same(base, got, 2e-4), (names, base, got)


@pytest.mark.timeout(300)
@pytest.mark.parametrize('env', [{'IMM_VGG_HEAD': 0}, {'IMM_CONV_DISABLE': 'vgg_head'}], ids=['IMM_VGG_HEAD=0', 'IMM_CONV_DISABLE=vgg_head'])
def test_vgg_head_in_one_launch_is_the_same_step(base, env):
    """The default program runs conv1_1 + conv1_2 of the frozen VGG16 as one launch (imm_vgg_head_fwd: conv1_1's halo produced on
    the matrix cores inside conv1_2's persistent workgroups); with it switched off the two launches it replaces run: one launch
    more, the same step to the last-bit rounding of conv1_1's activation (selfsup/vgg16.py:345-346)."""
    got = probe(**env)
    assert got['n_launches'] == base['n_launches'] + 1, (got['n_launches'], base['n_launches'])
    assert same(base, got, 2e-4), (base, got)


@pytest.mark.timeout(600)
def test_bn_direct_rows_threshold_is_the_same_step():
    """IMM_BN_DIRECT_ROWS (default 512): layers whose convolution leaves more partial rows of batch-norm sums get a parallel
    pre-reduction launch (imm_rows_reduce) in front of the fused finalize + apply pass — the same sums grouped differently
    (batch 8: the 128 x 128 layers' persistent grids leave 512 rows)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    r512, r256, r1024 = probe(PROBE_BATCH=8), probe(PROBE_BATCH=8, IMM_BN_DIRECT_ROWS=256), probe(PROBE_BATCH=8, IMM_BN_DIRECT_ROWS=1024)
    assert r256['n_launches'] > r512['n_launches'] >= r1024['n_launches'], (r256['n_launches'], r512['n_launches'], r1024['n_launches'])
    assert same(r512, r256, 2e-4) and same(r512, r1024, 2e-4), (r512, r256, r1024)


@pytest.mark.timeout(300)
def test_halo2_on_32x32_tiles_is_the_same_step(base):
    """IMM_HALO2X=1: the 64 -> 64 channel launches of conv_halo2 without batch-norm sums (VGG conv1_2's data gradient; its forward
    when the head is two launches) on v_mfma_f32_32x32x16 tiles (conv_halo2x_kernel: built and measured in round 6, slower, off by
    default) — the same step to the accumulation order of the nine taps (selfsup/vgg16.py:346)."""
    for env in ({'IMM_HALO2X': 1}, {'IMM_HALO2X': 1, 'IMM_VGG_HEAD': 0}):
        got = probe(**env)
        assert same(base, got, 2e-4), (env, base, got)


@pytest.mark.timeout(300)
def test_error_sums_in_one_launch_are_the_same_step(base):
    """IMM_SSE_ALL=0: the image pair's error sum and the deep layers' error sums as the two launches imm_masked_sse_all replaces
    (imm_model.py:126-147): one launch more, every partial sum — and so the step — bit for bit the same."""
    got = probe(IMM_SSE_ALL=0)
    assert got['n_launches'] == base['n_launches'] + 1, (got['n_launches'], base['n_launches'])
    assert all(got[k] == base[k] for k in ('loss0', 'loss1', 'mu_abs_sum', 'params_abs_sum')), (base, got)


@pytest.mark.timeout(300)
def test_normalise_on_load_is_the_same_step(base):
    """IMM_NOL=1: the apply passes of encoder conv_1..3 / renderer conv_5, conv_7 folded into their readers' LDS tiles
    (imm_conv2d_nol, imm_wgrad_job.x_scale): fewer launches, the same step to accumulation order."""
    got = probe(IMM_NOL=1)
    assert got['n_launches'] <= base['n_launches'], (got['n_launches'], base['n_launches'])
    assert same(base, got, 2e-4), (base, got)


@pytest.mark.timeout(300)
def test_first_convolution_from_the_image_is_the_same_step(base):
    """IMM_CONV_FIRST=1: conv_1 of both encoders straight from the f32 image (imm_conv_first), the tap-unrolled copies packed at the
    tail of the image-encoder lane for the filter gradients: the same step to accumulation order."""
    got = probe(IMM_CONV_FIRST=1)
    assert got['n_launches'] == base['n_launches']
    assert same(base, got, 2e-4), (base, got)


@pytest.mark.timeout(300)
def test_single_stream_and_eager_equal_the_two_lane_graph(base):
    """IMM_TWO_STREAMS=0 (one stream, no fork/join) and eager execution: the same launches in another schedule — bitwise."""
    one = probe(IMM_TWO_STREAMS=0)
    eager = probe(PROBE_GRAPH=0)
    for got in (one, eager):
        assert all(got[k] == base[k] for k in ('loss0', 'loss1', 'mu_abs_sum', 'params_abs_sum')), (base, got)


@pytest.mark.timeout(400)
def test_confined_lanes_are_the_same_step(base):
    """Round 6, opt-in (measured without gain: DESIGN.md item 59).  IMM_VGG_SPLIT=1: the ground-truth half of the VGG forward on
    lane 2, persistent launches confined to IMM_GT_CUS compute units (imm_set_cu_limit), the prediction half behind the renderer,
    error sums waiting for the lane's events; IMM_WG_CUS: the renderer's filter gradients planned for a share of the chip on lane 2
    beside the encoders' backward; IMM_WG_LANES: the filter-gradient kernel variants on several lanes; IMM_SSE_INPUT_LANE: the image-space
    error sum beside the VGG launches.  Same step up to the tile choice / split counts (accumulation order); and each of them on ONE
    stream (list order alone) is bitwise the three-lane graph."""
    for env in (dict(IMM_VGG_SPLIT=1, IMM_GT_CUS=64), dict(IMM_WG_CUS=96), dict(IMM_VGG_SPLIT=1, IMM_GT_CUS=0, IMM_WG_CUS=64),
                dict(IMM_WG_LANES=3, IMM_SSE_INPUT_LANE=1)):
        got = probe(**env)
        assert same(base, got, 2e-4), (env, base, got)
        one = probe(IMM_TWO_STREAMS=0, **env)
        assert all(one[k] == got[k] for k in ('loss0', 'loss1', 'mu_abs_sum', 'params_abs_sum')), (env, got, one)


@pytest.mark.timeout(300)
def test_debug_stamps_and_skip_tags(base):
    """IMM_DEBUG_STAMPS=all: a probe after every launch, monotone on the main lane, results untouched.
    IMM_DEBUG_SKIP_TAGS (timing experiment; results become wrong by design): the tagged launches are gone from the step."""
    st = probe(IMM_DEBUG_STAMPS='all', PROBE_GRAPH=0)
    assert st['stamps'] > base['n_launches'] and st['stamps_monotone_lane0']
    assert all(st[k] == base[k] for k in ('loss0', 'loss1', 'params_abs_sum'))
    sk = probe(IMM_DEBUG_SKIP_TAGS='clip_adam')
    assert sk['step_count'] == 0 and base['step_count'] == 2                # no optimizer launch: the counters never moved
    assert sk['loss0'] == base['loss0'] and sk['params_abs_sum'] != base['params_abs_sum']
