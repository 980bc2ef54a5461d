"""CPU ORACLE for the IMM conditional-generation training step.  TEST INFRASTRUCTURE ONLY.

This file is a *restatement*, in plain PyTorch-CPU tensor ops (fp32 by default, fp64 on request),
of the TensorFlow-1.10 graph that tomasjakab/imm builds for one training step.  It is the checker
that `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use; nothing in the
product package `imm_amd/` may import it (the product path fails loudly without its HIP library).

PARITY UNPINNED.  The reference cannot be imported or run here (`import tensorflow` fails, TF 1.10
is a Python-2-era pin in /root/reference/requirements.txt:1) and the reference repository ships no
tests, fixtures or golden vectors.  The TF1 op semantics S1..S12 below are
restated from the TF 1.10 op definitions and pinned by hand-derived known-answer tests in
tests/test_oracle_kat.py and by an independent fp64 numpy loop implementation (oracle/np_ref.py).

Reference call sites followed (all paths relative to /root/reference):
  imm/models/imm_model.py:34-78     get_gaussian_maps            -> gaussian_maps()
  imm/models/imm_model.py:182-217   IMMModel.encoder             -> encoder()
  imm/models/imm_model.py:220-230   IMMModel.image_encoder       -> (encoder(); only block 4 is live)
  imm/models/imm_model.py:233-276   IMMModel.pose_encoder        -> pose_encoder()
  imm/models/imm_model.py:279-357   IMMModel.model               -> model_forward()
  imm/models/imm_model.py:154-179   IMMModel.simple_renderer     -> renderer()
  imm/models/imm_model.py:111-151   _colorization_reconstruction_loss -> perceptual_loss()
  imm/models/imm_model.py:360-405   IMMModel.loss                -> total_loss()
  imm/models/imm_model.py:408-410   _loss_mask                   -> loss_mask_at()
  imm/models/base_model.py:33-50    _decay, _exp_running_avg     -> weight_decay_loss(), exp_running_avg()
  imm/tf_utils/nn_utils.py:24-51,87-109,151-209  conv_block      -> conv_block()
  imm/models/selfsup/build_vgg16.py:14-35, vgg16.py:141-240,289-375, ops.py:16-26 -> vgg16_features()
  imm/train/cnn_train_multi.py:66-106,157-166,231-243 -> train_step() (tower mean -> clip -> Adam)
  scripts/train.py:87-98            lr schedule / Adam           -> learning_rate(), adam_apply()
  imm/datasets/tps_dataset.py:47-67 smooth loss mask             -> smooth_mask()

TF1 semantics restated:
  S1  conv2d = cross-correlation NHWC x HWIO, SAME: out=ceil(in/s), pad_total=max((out-1)s+k-in,0),
      pad_before = pad_total//2 (the extra pixel goes bottom/right).
  S2  tf.image.resize_images (bilinear, align_corners=False, legacy): src = dst*(in/out),
      lower=floor(src), upper=min(lower+1,in-1), lerp=src-lower.
  S3  resize_bilinear(align_corners=True): src = dst*(in-1)/(out-1).
  S4  tf.layers.batch_normalization(fused=True): eps 1e-3, momentum 0.99; train: biased batch var for
      normalisation, unbiased var into the moving average; eval: moving stats.
  S5  weights truncated_normal(std 0.01) (resample beyond 2 sigma), biases 0, gamma 1, beta 0.
  S6  l2_regularizer(s)(w) = s*sum(w^2)/2, on conv kernels `w` only.
  S7  linspace(-1,1,n)[i] = -1 + 2i/(n-1); softmax over axis 1.
  S8  max_pool 2x2/2 SAME on even sides: no padding.
  S9  clip_by_norm(g,c) = g*c/max(||g||,c) per tensor; Adam in TF form
      lr_t = lr*sqrt(1-b2^t)/(1-b1^t); var -= lr_t*m/(sqrt(v)+eps).
  S10 the 6 "bug-fix" output channels of the last renderer conv get zero loss gradient but non-zero
      weight-decay gradient.
  S11 multi tower: BN stats per tower, gradients averaged THEN clipped, printed loss = tower mean.
  S12 eval: BN uses moving stats, the loss normaliser is computed but not assigned.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
WEIGHT_DECAY = 1e-5
INIT_STD = 0.01
PERCEPTUAL_WS = [100.0, 1.6, 2.3, 1.8, 2.8, 100.0]   # imm_model.py:131
RUNNING_AVG_RHO = 0.99                                # base_model.py:39
VGG_GRAY_MEAN = 114.451                               # build_vgg16.py:26
VGG_LAYERS = [  # (name, cin, cout) up to conv5_2; pools after conv1_2, conv2_2, conv3_3, conv4_3
    ('conv1_1', 1, 64), ('conv1_2', 64, 64),
    ('conv2_1', 64, 128), ('conv2_2', 128, 128),
    ('conv3_1', 128, 256), ('conv3_2', 256, 256), ('conv3_3', 256, 256),
    ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512),
    ('conv5_1', 512, 512), ('conv5_2', 512, 512),
]
VGG_POOL_AFTER = {'conv1_2', 'conv2_2', 'conv3_3', 'conv4_3'}


# ----------------------------------------------------------------------------------------------
# config
# ----------------------------------------------------------------------------------------------
class Cfg(dict):
    """Tiny attr-dict with the `hasattr` behaviour the reference relies on (imm_model.py:285,349)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v


def default_model_config(n_maps=10):
    """configs/experiments/celeba-10pts.yaml:25-45 (`model:` block)."""
    return Cfg(
        gauss_std=0.10, gauss_mode='rot', n_maps=n_maps, n_filters=32, block_sizes=[1, 1, 1],
        n_filters_render=32, renderer_stride=2, min_res=16, same_n_filt=False,
        reconstruction_loss='perceptual',
        perceptual=dict(l2=True, comp=['input', 'conv1_2', 'conv2_2', 'conv3_2', 'conv4_2', 'conv5_2'],
                        net_file='synthetic'),
        loss_mask=True, confidence=False, channels_bug_fix=True)


# ----------------------------------------------------------------------------------------------
# TF1 primitive restatements
# ----------------------------------------------------------------------------------------------
def same_pad(n_in, k, s):
    """S1: returns (pad_before, pad_after, n_out)."""
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return total // 2, total - total // 2, n_out


def conv2d_same(x, w, b=None, stride=1):
    """tf.nn.conv2d(x NHWC, w HWIO, SAME) + bias_add (nn_utils.py:100,108)."""
    kh, kw = w.shape[0], w.shape[1]
    pt, pb, _ = same_pad(x.shape[1], kh, stride)
    pl, pr, _ = same_pad(x.shape[2], kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def _resize_axis(x, axis, out, align_corners):
    n = x.shape[axis]
    if out == n:
        return x
    if align_corners and out > 1:
        scale = (n - 1) / (out - 1)
    else:
        scale = n / out
    src = torch.arange(out, dtype=torch.float64) * scale
    lo = torch.floor(src).to(torch.long)
    hi = torch.clamp(lo + 1, max=n - 1)
    lerp = (src - lo.to(torch.float64)).to(x.dtype)
    shape = [1] * x.dim()
    shape[axis] = out
    lerp = lerp.reshape(shape)
    a = x.index_select(axis, lo)
    b = x.index_select(axis, hi)
    return a + (b - a) * lerp


def resize_bilinear(x, out_h, out_w, align_corners=False):
    """S2 / S3 on NHWC tensors (imm_model.py:175, :334, :409)."""
    # TF interpolates rows and columns separably: top/bottom rows lerped in x, then in y.
    x = _resize_axis(x, 2, out_w, align_corners)
    x = _resize_axis(x, 1, out_h, align_corners)
    return x


def batch_norm(x, gamma, beta, moving_mean, moving_var, training):
    """S4.  Returns y and the (new_moving_mean, new_moving_var) pair (== inputs when not training)."""
    if training:
        mean = x.mean(dim=(0, 1, 2))
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
        n = x.shape[0] * x.shape[1] * x.shape[2]
        unbiased = var * (n / max(n - 1, 1))
        new_mm = moving_mean * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM)
        new_mv = moving_var * BN_MOMENTUM + unbiased.detach() * (1 - BN_MOMENTUM)
    else:
        mean, var = moving_mean, moving_var
        new_mm, new_mv = moving_mean, moving_var
    y = (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta
    return y, (new_mm, new_mv)


def max_pool2(x):
    """S8 (ops.py:16-26): 2x2/2 on even sides."""
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def linspace_pm1(n, dtype):
    """S7."""
    if n == 1:
        return torch.tensor([-1.0], dtype=dtype)
    return (-1.0 + 2.0 * torch.arange(n, dtype=torch.float64) / (n - 1)).to(dtype)


def gaussian_maps(mu, shape_hw, inv_std, mode='ankush'):
    """imm_model.py:34-78.  mu [B,K,2] (y,x) -> [B,H,W,K]."""
    mu_y, mu_x = mu[:, :, 0:1], mu[:, :, 1:2]
    y = linspace_pm1(shape_hw[0], mu.dtype)
    x = linspace_pm1(shape_hw[1], mu.dtype)
    if mode in ('rot', 'flat'):
        mu_y, mu_x = mu_y.unsqueeze(-1), mu_x.unsqueeze(-1)
        y = y.reshape(1, 1, shape_hw[0], 1)
        x = x.reshape(1, 1, 1, shape_hw[1])
        dist = ((y - mu_y) ** 2 + (x - mu_x) ** 2) * inv_std ** 2
        if mode == 'rot':
            g = torch.exp(-dist)
        else:
            g = torch.exp(-torch.pow(dist + 1e-5, 0.25))
    elif mode == 'ankush':
        y = y.reshape(1, 1, shape_hw[0])
        x = x.reshape(1, 1, shape_hw[1])
        g_y = torch.exp(-torch.sqrt(1e-4 + torch.abs((mu_y - y) * inv_std)))
        g_x = torch.exp(-torch.sqrt(1e-4 + torch.abs((mu_x - x) * inv_std)))
        g = g_y.unsqueeze(3) * g_x.unsqueeze(2)
    else:
        raise ValueError('Unknown mode: ' + str(mode))
    return g.permute(0, 2, 3, 1)


def soft_argmax(heat):
    """imm_model.py:252-264.  heat [B,H,W,K] -> mu [B,K,2], py [B,H,K], px [B,W,K]."""
    def coord(other_axis, n):
        p = torch.softmax(heat.mean(dim=other_axis), dim=1)
        c = linspace_pm1(n, heat.dtype).reshape(1, n, 1)
        return (p * c).sum(dim=1), p
    gy, py = coord(2, heat.shape[1])
    gx, px = coord(1, heat.shape[2])
    return torch.stack([gy, gx], dim=2), py, px


def smooth_mask(h, w, margin=10, step=20, b=0.4, dtype=torch.float32):
    """imm/datasets/tps_dataset.py:47-67."""
    def smooth_step(n, bb):
        x = torch.linspace(-1.0, 1.0, n, dtype=torch.float32)
        return 0.5 + 0.5 * torch.tanh(x / bb)

    def strip(size):
        return torch.cat([torch.zeros(margin), smooth_step(step, b), torch.ones(size - 2 * margin - 2 * step),
                          smooth_step(step, -b), torch.zeros(margin)])
    return (strip(h)[:, None] * strip(w)[None]).to(dtype)


# ----------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------
def truncated_normal(rng, shape, std):
    """S5: resample every draw outside 2 sigma."""
    z = rng.standard_normal(shape)
    bad = np.abs(z) > 2.0
    while bad.any():
        z[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(z) > 2.0
    return (z * std).astype(np.float32)


def encoder_spec(n_filters):
    f = n_filters
    return [(7, 3, f, 1), (3, f, f, 1), (3, f, 2 * f, 2), (3, 2 * f, 2 * f, 1), (3, 2 * f, 4 * f, 2),
            (3, 4 * f, 4 * f, 1), (3, 4 * f, 8 * f, 2), (3, 8 * f, 8 * f, 1)]   # (k, cin, cout, stride)


def renderer_spec(cfg, image_size, n_out):
    """imm_model.py:154-179 -> list of (k, cin, cout, batch_norm, upsample_after)."""
    filters = cfg.n_filters_render * 8
    cin = cfg.n_filters * 8 + cfg.n_maps
    size = 16
    spec = []
    while size <= image_size:
        spec.append((3, cin, filters, True, False))
        if size == image_size:
            spec.append((3, filters, n_out, False, False))
            break
        spec.append((3, filters, filters, True, True))
        cin = filters
        size *= 2
        if filters >= 8:
            filters //= 2
    return spec


def n_renderer_out(cfg):
    extra = len(cfg.perceptual.comp) if getattr(cfg, 'channels_bug_fix', False) else 0
    return 3 + extra


def init_params(cfg, image_size=128, seed=1, vgg_seed=2):
    """Seeded synthetic initialisation.  Returns (trainable OrderedDict, state dict).

    Trainable names follow the TF variable names of SURVEY.md §5 (checkpoint paragraph), in graph
    construction order: image_encoder, pose_encoder (+ its 1x1 conv), renderer.
    """
    rng = np.random.default_rng(seed)
    P, S = OrderedDict(), OrderedDict()

    def add_conv(scope, k, cin, cout, bn):
        P[scope + '/w'] = torch.from_numpy(truncated_normal(rng, (k, k, cin, cout), INIT_STD))
        P[scope + '/b'] = torch.zeros(cout)
        if bn:
            P[scope + '/gamma'] = torch.ones(cout)
            P[scope + '/beta'] = torch.zeros(cout)
            S[scope + '/moving_mean'] = torch.zeros(cout)
            S[scope + '/moving_variance'] = torch.ones(cout)

    for enc in ('image_encoder', 'pose_encoder'):
        for i, (k, cin, cout, _s) in enumerate(encoder_spec(cfg.n_filters)):
            add_conv('model/%s/encoder/conv_%d' % (enc, i + 1), k, cin, cout, True)
        if enc == 'pose_encoder':
            add_conv('model/pose_encoder/conv_1', 1, cfg.n_filters * 8, cfg.n_maps, False)
    for i, (k, cin, cout, bn, _u) in enumerate(renderer_spec(cfg, image_size, n_renderer_out(cfg))):
        add_conv('model/renderer/conv_%d' % (i + 1), k, cin, cout, bn)

    for k, name in enumerate(cfg.perceptual.comp):
        S['loss/%s_agg' % name] = torch.tensor(PERCEPTUAL_WS[k])
    vrng = np.random.default_rng(vgg_seed)
    for name, cin, cout in VGG_LAYERS:
        fan_in = 9 * cin
        S['vgg16/%s/weights' % name] = torch.from_numpy(
            (vrng.standard_normal((3, 3, cin, cout)) * math.sqrt(2.0 / fan_in)).astype(np.float32))
        S['vgg16/%s/biases' % name] = torch.from_numpy((vrng.standard_normal(cout) * 0.1).astype(np.float32))
    return P, S


def synthetic_inputs(batch, image_size=128, seed=0):
    """image, future_image ~ U[0,255) fp32 NHWC; mask = smooth mask."""
    rng = np.random.default_rng(seed)
    im = torch.from_numpy(rng.uniform(0.0, 255.0, (batch, image_size, image_size, 3)).astype(np.float32))
    fut = torch.from_numpy(rng.uniform(0.0, 255.0, (batch, image_size, image_size, 3)).astype(np.float32))
    mask = smooth_mask(image_size, image_size).reshape(1, image_size, image_size, 1).repeat(batch, 1, 1, 1)
    return {'image': im, 'future_image': fut, 'mask': mask.contiguous()}


# ----------------------------------------------------------------------------------------------
# network
# ----------------------------------------------------------------------------------------------
class _Ctx:
    """Carries params / state / new BN statistics through one forward."""

    def __init__(self, P, S, training, act_round=None):
        self.P, self.S, self.training = P, S, training
        self.new_state = {}
        self.acts = OrderedDict()
        # act_round: optional callable emulating the product path's storage rounding of activations
        self.round = act_round if act_round is not None else (lambda t: t)


def conv_block(ctx, x, scope, stride=1, bn=True, relu=True):
    """nn_utils.py:151-209: conv + bias -> fused BN -> ReLU."""
    P = ctx.P
    y = conv2d_same(x, P[scope + '/w'], P[scope + '/b'], stride)
    ctx.acts[scope + ':conv'] = y
    if bn:
        y = ctx.round(y)
        y, (mm, mv) = batch_norm(y, P[scope + '/gamma'], P[scope + '/beta'],
                                 ctx.S[scope + '/moving_mean'], ctx.S[scope + '/moving_variance'], ctx.training)
        ctx.new_state[scope + '/moving_mean'] = mm
        ctx.new_state[scope + '/moving_variance'] = mv
    if relu:
        y = torch.relu(y)
    if bn or relu:
        y = ctx.round(y)
    ctx.acts[scope] = y
    return y


def encoder(ctx, x, scope, cfg):
    """imm_model.py:182-217.  Returns the 4 block outputs."""
    blocks = []
    for i, (_k, _ci, _co, s) in enumerate(encoder_spec(cfg.n_filters)):
        x = conv_block(ctx, x, '%s/encoder/conv_%d' % (scope, i + 1), stride=s)
        if i % 2 == 1:
            blocks.append(x)
    return blocks


def pose_encoder(ctx, x, cfg, map_sizes):
    """imm_model.py:233-276."""
    feat = encoder(ctx, x, 'model/pose_encoder', cfg)[-1]
    heat = conv_block(ctx, feat, 'model/pose_encoder/conv_1', bn=False, relu=False)
    mu, py, px = soft_argmax(heat)
    maps = [gaussian_maps(mu, [s, s], 1.0 / cfg.gauss_std, mode=cfg.gauss_mode) for s in map_sizes]
    return mu, maps, heat, py, px


def render_sizes(cfg, max_size):
    """imm_model.py:295-303."""
    sizes, size = [], max_size
    while True:
        sizes.append(size)
        if size <= cfg.min_res:
            break
        size = size // cfg.renderer_stride
    return sizes


def renderer(ctx, x, cfg, image_size):
    """imm_model.py:154-179."""
    for i, (_k, _ci, _co, bn, up) in enumerate(renderer_spec(cfg, image_size, n_renderer_out(cfg))):
        x = conv_block(ctx, x, 'model/renderer/conv_%d' % (i + 1), bn=bn, relu=bn)
        if up:
            x = ctx.round(resize_bilinear(x, 2 * x.shape[1], 2 * x.shape[2]))
    return x


def model_forward(ctx, im, future_im, cfg, all_maps=False):
    """imm_model.py:279-357.  Only the 16x16 joint embedding feeds the renderer (:161)."""
    size = future_im.shape[1]
    assert future_im.shape[1] == future_im.shape[2]
    sizes = render_sizes(cfg, size)
    emb = encoder(ctx, im, 'model/image_encoder', cfg)[-1]
    mu, maps, heat, py, px = pose_encoder(ctx, future_im, cfg, sizes if all_maps else [sizes[-1]])
    rs = sizes[-1]
    if emb.shape[1] != rs:   # :324-335 (only when the image side is not 128)
        emb = ctx.round(resize_bilinear(emb, rs, rs, align_corners=True))
    joint = torch.cat([emb, ctx.round(maps[-1])], dim=-1)
    ctx.acts['joint'] = joint
    out = renderer(ctx, joint, cfg, size)
    pred = out[..., :3]
    return pred, mu, maps, heat, py, px


def vgg16_features(S, ims, act_round=None):
    """build_vgg16.py:14-35 + vgg16.py:289-375 (frozen; caffe-BN already folded into W,b)."""
    rnd = act_round if act_round is not None else (lambda t: t)
    x = ims.mean(dim=3, keepdim=True) / 255.0 - VGG_GRAY_MEAN / 255.0
    net = {'input': ims}
    for name, _ci, _co in VGG_LAYERS:
        x = rnd(torch.relu(conv2d_same(x, S['vgg16/%s/weights' % name], S['vgg16/%s/biases' % name])))
        net[name] = x
        if name in VGG_POOL_AFTER:
            x = max_pool2(x)
    return net


def loss_mask_at(mask, side):
    """imm_model.py:408-410 (S2: integer down-factor => strided pick)."""
    return resize_bilinear(mask, side, side)


def exp_running_avg(x, x_avg):
    """base_model.py:39-50: differentiable in x."""
    return x_avg + (1.0 - RUNNING_AVG_RHO) * (x - x_avg)


def perceptual_loss(ctx, gt, pred, mask, cfg):
    """imm_model.py:111-151."""
    names = list(cfg.perceptual.comp)
    feats = vgg16_features(ctx.S, torch.cat([gt, pred], dim=0), ctx.round)
    f_e = (lambda t: t * t) if cfg.perceptual.l2 else torch.abs
    terms, ms = [], []
    for k, name in enumerate(names):
        f = feats[name]
        b = f.shape[0] // 2
        l = f_e(f[:b] - f[b:])
        mk = loss_mask_at(mask, l.shape[1]) if mask is not None else None
        m = (l * mk).mean() if mk is not None else l.mean()
        wl = exp_running_avg(m, ctx.S['loss/%s_agg' % name])
        if ctx.training:
            ctx.new_state['loss/%s_agg' % name] = wl.detach()
        l = l / wl
        terms.append((l * mk).mean() if mk is not None else l.mean())
        ms.append(m)
    ctx.acts['vgg'] = feats
    return 1000.0 * sum(terms), terms, ms


def cost_ema_update(state, costs, decay=0.99):
    """BaseModel._add_cost_summary (base_model.py:52-60): tf.train.ExponentialMovingAverage(decay).apply([cost]) on a TENSOR.  The
    reference pins tensorflow-gpu==1.10.0 (requirements.txt:1), whose ExponentialMovingAverage.__init__ has zero_debias=False: the
    shadow of a tensor starts at 0 (create_zeros_slot) and `average(cost)` is the RAW biased value
        shadow <- decay * shadow + (1 - decay) * cost
    (biased towards 0 by the factor 1 - decay^t for the first few hundred steps — that is what the reference's `<cost>_avg` summary
    shows; ADVICE r5: round 5 had assumed the zero-debiased form of moving_averages.assign_moving_average's default).
    state = [shadow_0, .., shadow_{n-1}, local_step] (python floats; the step count is kept for bookkeeping only); returns
    (new state, averages)."""
    n = len(costs)
    new = [decay * state[i] + (1.0 - decay) * float(costs[i]) for i in range(n)] + [state[n] + 1.0]
    return new, [new[i] for i in range(n)]


def weight_decay_loss(P):
    """base_model.py:33-37 + nn_utils.py:44-46 (S6): conv kernels only."""
    return sum(WEIGHT_DECAY * 0.5 * (v ** 2).sum() for k, v in P.items() if k.endswith('/w'))


def forward(P, S, inputs, cfg, training=True, build_loss=True, act_round=None, all_maps=False):
    """IMMModel.build (imm_model.py:413-490) as a function.  Returns dict of outputs."""
    ctx = _Ctx(P, S, training, act_round)
    im, fut = inputs['image'], inputs['future_image']
    mask = inputs.get('mask')
    pred, mu, maps, heat, py, px = model_forward(ctx, im, fut, cfg, all_maps)
    out = {'future_im_pred': pred, 'gauss_yx': mu, 'pose_embeddings': maps, 'heatmaps': heat,
           'gauss_y_prob': py, 'gauss_x_prob': px, 'acts': ctx.acts, 'new_state': ctx.new_state}
    if build_loss:
        if cfg.loss_mask:
            if mask is None:
                raise RuntimeError('No loss mask recieved but is required.')
        else:
            mask = None
        if cfg.reconstruction_loss == 'perceptual':
            rec, terms, ms = perceptual_loss(ctx, fut, pred, mask, cfg)
            w_rec = 1.0
            out['loss_terms'], out['loss_means'] = terms, ms
        elif cfg.reconstruction_loss == 'l2':
            l = (pred - fut) ** 2
            rec = 1000.0 * ((l * mask).mean() if mask is not None else l.mean())
            w_rec = 1.0 / 255.0
        else:
            raise ValueError('Reconsutruction loss-type: ' + cfg.reconstruction_loss + ' not understood')
        wl = weight_decay_loss(P)
        out['reconstruction_loss'], out['weights_loss'] = rec, wl
        out['loss'] = w_rec * rec + wl
    return out


# ----------------------------------------------------------------------------------------------
# training step
# ----------------------------------------------------------------------------------------------
def learning_rate(step, start_val=1e-3, decay_step=100000, decay=0.95, lr_multiple=1.0):
    """scripts/train.py:92-96 (staircase exponential decay)."""
    return lr_multiple * start_val * decay ** (step // decay_step)


def clip_by_norm(g, c):
    """S9 (cnn_train_multi.py:98,238)."""
    n = torch.sqrt((g * g).sum())
    return g * c / torch.maximum(n, torch.tensor(c, dtype=g.dtype))


def new_adam_state(P):
    return {'t': 0, 'm': OrderedDict((k, torch.zeros_like(v)) for k, v in P.items()),
            'v': OrderedDict((k, torch.zeros_like(v)) for k, v in P.items())}


def adam_apply(P, grads, opt, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer.apply_gradients (S9)."""
    opt['t'] += 1
    t = opt['t']
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    newP = OrderedDict()
    for k, p in P.items():
        g = grads[k]
        opt['m'][k] = b1 * opt['m'][k] + (1 - b1) * g
        opt['v'][k] = b2 * opt['v'][k] + (1 - b2) * g * g
        newP[k] = p - lr_t * opt['m'][k] / (torch.sqrt(opt['v'][k]) + eps)
    return newP


def adadelta_apply(P, grads, opt, lr, rho=0.95, eps=1e-6):
    """tf.train.AdadeltaOptimizer(lr, rho=0.95, epsilon=1e-06) (scripts/train.py:99-100; TF 1.10 ApplyAdadelta):
    accum = rho accum + (1-rho) g^2; update = sqrt(accum_update + eps) * rsqrt(accum + eps) * g;
    accum_update = rho accum_update + (1-rho) update^2; var -= lr * update.  opt['v'] = accum, opt['m'] = accum_update (zeros)."""
    newP = OrderedDict()
    for k, p in P.items():
        g = grads[k]
        opt['v'][k] = rho * opt['v'][k] + (1 - rho) * g * g
        u = torch.sqrt(opt['m'][k] + eps) / torch.sqrt(opt['v'][k] + eps) * g
        opt['m'][k] = rho * opt['m'][k] + (1 - rho) * u * u
        newP[k] = p - lr * u
    return newP


def new_adagrad_state(P, initial_accumulator_value=0.1):
    return {'v': OrderedDict((k, torch.full_like(v, initial_accumulator_value)) for k, v in P.items())}


def adagrad_apply(P, grads, opt, lr):
    """tf.train.AdagradOptimizer(lr) (scripts/train.py:101-102; initial_accumulator_value 0.1; TF 1.10 ApplyAdagrad):
    accum += g^2; var -= lr * g * rsqrt(accum)."""
    newP = OrderedDict()
    for k, p in P.items():
        g = grads[k]
        opt['v'][k] = opt['v'][k] + g * g
        newP[k] = p - lr * g / torch.sqrt(opt['v'][k])
    return newP


def loss_and_grads(P, S, inputs, cfg, act_round=None):
    """One tower: loss + d loss / d trainable (tf.gradients through everything, incl. `wl`)."""
    Pg = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in P.items())
    out = forward(Pg, S, inputs, cfg, training=True, act_round=act_round)
    gl = torch.autograd.grad(out['loss'], list(Pg.values()), allow_unused=True)
    grads = OrderedDict((k, (g if g is not None else torch.zeros_like(v)).detach())
                        for (k, v), g in zip(Pg.items(), gl))
    return out, grads


def train_step(P, S, opt, tower_inputs, cfg, clip=1.0, lr=1e-3, act_round=None):
    """cnn_train_multi.py:109-192 / :195-250: towers -> mean grads -> per-tensor clip -> Adam.

    `tower_inputs` is a list of per-tower input dicts (length 1 == train_single).  State updates
    (BN moving stats, loss normalisers) are taken from the LAST tower (cnn_train_multi.py:155).
    Returns (new_P, new_S, info).
    """
    outs, gsum = [], None
    for inp in tower_inputs:
        out, g = loss_and_grads(P, S, inp, cfg, act_round)
        outs.append(out)
        gsum = g if gsum is None else OrderedDict((k, gsum[k] + g[k]) for k in g)
    n = len(tower_inputs)
    gmean = OrderedDict((k, v / n) for k, v in gsum.items())
    gclip = OrderedDict((k, clip_by_norm(v, clip)) for k, v in gmean.items()) if clip is not None else gmean
    newP = adam_apply(P, gclip, opt, lr)
    newS = OrderedDict(S)
    newS.update({k: v.detach() for k, v in outs[-1]['new_state'].items()})
    info = {'loss': sum(float(o['loss'].detach()) for o in outs) / n, 'outs': outs, 'grads': gmean, 'clipped': gclip}
    return newP, newS, info


def to_dtype(d, dtype):
    return type(d)((k, v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items())
