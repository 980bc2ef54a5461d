"""Static launch program of one IMM training step on one MI355X.

The reference builds a TF1 graph once and replays it with session.run
(/root/reference/imm/train/cnn_train_multi.py:459).  The MI355X-native counterpart is this engine:
at construction it allocates every activation / gradient / workspace buffer in HBM once (NHWC,
16-bit activations, f32 statistics and parameters) and records the step as three ordered lists of
kernel launches on libimm_hip.so —

    prog_fwd : inputs -> encoders -> landmark bottleneck -> renderer -> VGG16 features -> loss
    prog_bwd : loss gradient -> VGG dgrad -> renderer / encoders (BN bwd, wgrad, dgrad) -> flat grads
    prog_opt : (grads already all-reduced) per-tensor clip + Adam -> re-pack 16-bit weights

— which run eagerly or are captured into HIP graphs and replayed.  There is no autograd engine, no
allocation and no host synchronisation inside a step.

Network definition follows /root/reference/imm/models/imm_model.py (encoder :182-217, pose_encoder
:233-276, model :279-357, simple_renderer :154-179, _colorization_reconstruction_loss :111-151) and
imm/models/selfsup/vgg16.py:343-370; TF1 op semantics as listed in SURVEY.md §8a (S1-S12).
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L
from . import ops

BN_EPS = 1e-3          # tf.layers.batch_normalization default (nn_utils.py:201)
BN_MOMENTUM = 0.99
# Partial rows of batch-norm sums that the fused finalize + apply passes reduce themselves; layers with more rows get a parallel
# pre-reduction launch (imm_rows_reduce) in front.  IMM_BN_DIRECT_ROWS: A/B only (profiles/r06_bn_direct_rows_ab.txt).
BN_DIRECT_ROWS = int(os.environ.get('IMM_BN_DIRECT_ROWS', '512'))
WEIGHT_DECAY = 1e-5    # base_model.py:62-69
INIT_STD = 0.01
PERCEPTUAL_WS = [100.0, 1.6, 2.3, 1.8, 2.8, 100.0]   # imm_model.py:131
SUPPORTED_COMP = ['input', 'conv1_2', 'conv2_2', 'conv3_2', 'conv4_2', 'conv5_2']
VGG_LAYERS = [('conv1_1', 1, 64), ('conv1_2', 64, 64), ('conv2_1', 64, 128), ('conv2_2', 128, 128),
              ('conv3_1', 128, 256), ('conv3_2', 256, 256), ('conv3_3', 256, 256),
              ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512),
              ('conv5_1', 512, 512), ('conv5_2', 512, 512)]
VGG_POOL_AFTER = ('conv1_2', 'conv2_2', 'conv3_3', 'conv4_3')
WGRAD_MULTI_MAX_JOBS, WGRAD_MULTI_MAX_VARIANTS = 64, 16     # WGM_MAX_JOBS / WGM_MAX_LAUNCH of csrc/conv_wgrad.hip


def plan_wgrad_chunks(groups, max_jobs, max_slots):
    """Split {(variant key, per-CU residency): [jobs]} into chunks that one imm_conv2d_wgrad_multi table can hold: at most
    `max_jobs` jobs and `max_slots` LAUNCH SLOTS, counted the way imm_conv2d_wgrad_multi_plan counts them — a group of the generic
    kernel (key // 100000 == 0) takes one slot PER JOB, every other group one slot (with the specialised kernels
    disabled, or on odd shapes, more than 16 generic jobs in one chunk still failed with "more than 16 kernel variants").
    Deeper configurations are issued as several launches instead of failing at engine build.  Pure host logic (CPU-tested)."""
    chunks, cur, cur_jobs, cur_slots = [], [], 0, 0
    for gk, members in groups.items():
        generic = gk[0] // 100000 == 0
        members = list(members)
        while members:
            room_jobs, room_slots = max_jobs - cur_jobs, max_slots - cur_slots
            take = min(len(members), room_jobs, room_slots) if generic else (min(len(members), room_jobs) if room_slots >= 1 else 0)
            if take <= 0:
                chunks.append(cur); cur, cur_jobs, cur_slots = [], 0, 0
                continue
            part, members = members[:take], members[take:]
            cur.append((gk, part)); cur_jobs += len(part); cur_slots += len(part) if generic else 1
    if cur:
        chunks.append(cur)
    return chunks
COST_EMA_STATE = 'summaries/cost_movavg'      # {reconstruction_loss, weights_loss, loss_total shadows, update count}
VGG_TAPS = {'conv1_2': 1, 'conv2_2': 2, 'conv3_2': 3, 'conv4_2': 4, 'conv5_2': 5}


def encoder_spec(n_filters):
    """(k, cin, cout, stride) of imm_model.py:190-214."""
    f = n_filters
    return [(7, 3, f, 1), (3, f, f, 1), (3, f, 2 * f, 2), (3, 2 * f, 2 * f, 1), (3, 2 * f, 4 * f, 2),
            (3, 4 * f, 4 * f, 1), (3, 4 * f, 8 * f, 2), (3, 8 * f, 8 * f, 1)]


def renderer_spec(cfg, image_size, n_out):
    """(k, cin, cout, batch_norm, upsample_after) of imm_model.py:154-179."""
    filters = cfg.n_filters_render * 8
    cin = cfg.n_filters * 8 + cfg.n_maps
    size, spec = 16, []
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


def render_sizes(cfg, max_size):
    """imm_model.py:295-303."""
    sizes, size = [], max_size
    while True:
        sizes.append(size)
        if size <= cfg.min_res:
            break
        size = size // cfg.renderer_stride
    return sizes


def n_renderer_out(cfg):
    extra = len(cfg.perceptual.comp) if getattr(cfg, 'channels_bug_fix', False) else 0
    return 3 + extra


def trainable_spec(cfg, image_size):
    """Ordered (name, shape, weight_decay) of every trainable tensor, TF variable names."""
    out = []

    def add(scope, k, cin, cout, bn):
        out.append((scope + '/w', (k, k, cin, cout), WEIGHT_DECAY))
        out.append((scope + '/b', (cout,), 0.0))
        if bn:
            out.append((scope + '/gamma', (cout,), 0.0))
            out.append((scope + '/beta', (cout,), 0.0))

    for enc in ('image_encoder', 'pose_encoder'):
        for i, (k, ci, co, _s) in enumerate(encoder_spec(cfg.n_filters)):
            add('model/%s/encoder/conv_%d' % (enc, i + 1), k, ci, co, True)
        if enc == 'pose_encoder':
            add('model/pose_encoder/conv_1', 1, cfg.n_filters * 8, cfg.n_maps, False)
    for i, (k, ci, co, bn, _u) in enumerate(renderer_spec(cfg, image_size, n_renderer_out(cfg))):
        add('model/renderer/conv_%d' % (i + 1), k, ci, co, bn)
    return out


def truncated_normal(rng, shape, std):
    z = rng.standard_normal(shape)
    bad = np.abs(z) > 2.0
    while bad.any():
        z[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(z) > 2.0
    return (z * std).astype(np.float32)


def synthetic_vgg_weights(seed=2):
    """Stand-in for vgg16.caffemodel.h5 (not available offline): He-normal kernels, N(0,0.1) biases."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, cin, cout in VGG_LAYERS:
        w['vgg16/%s/weights' % name] = torch.from_numpy(
            (rng.standard_normal((3, 3, cin, cout)) * math.sqrt(2.0 / (9 * cin))).astype(np.float32))
        w['vgg16/%s/biases' % name] = torch.from_numpy((rng.standard_normal(cout) * 0.1).astype(np.float32))
    return w


class _Launch:
    __slots__ = ('fn', 'tag', 'flops', 'bytes', 'name', 'lane', 'variant', 'cus', 'group')

    def __init__(self, fn, tag, flops=0.0, nbytes=0.0, name='', lane=0, variant='', cus=0, group=''):
        self.fn, self.tag, self.flops, self.bytes, self.name, self.lane, self.variant = fn, tag, flops, nbytes, name, lane, variant
        # cus: the launch is issued under imm_set_cu_limit(cus) (0 = whole device); group: 'gt' = the frozen VGG's ground-truth
        # lane (launches AND its fork / join / event marks): needed by the loss only (forward_model_only drops it)
        self.cus, self.group = cus, group


class _ConvLayer:
    """Geometry + buffers of one trainable convolution block (conv+bias [-> BN -> ReLU])."""
    pass


class IMMEngine:
    def __init__(self, cfg, batch, image_size, device='cuda:0', act_dtype=torch.bfloat16, seed=1, vgg_weights=None,
                 hparams=None, world_size=1, dp_buckets=None):
        if cfg.gauss_mode not in L.GAUSS_MODES:
            raise ValueError('Unknown mode: ' + str(cfg.gauss_mode))       # imm_model.py:75
        if cfg.reconstruction_loss not in ('perceptual', 'l2'):
            raise ValueError('Reconsutruction loss-type: ' + str(cfg.reconstruction_loss) + ' not understood')   # imm_model.py:389
        # loss features (imm_model.py:124-147): any ordered selection of the raw image and the five tapped VGG layers; the
        # initial normalisers ws[k] go by POSITION in the list, as in the reference (:131,144)
        self.loss_kind = cfg.reconstruction_loss
        self.comp = list(cfg.perceptual.comp) if self.loss_kind == 'perceptual' else ['input']
        if self.loss_kind == 'perceptual':
            if not self.comp or len(set(self.comp)) != len(self.comp) or len(self.comp) > len(PERCEPTUAL_WS):
                raise ValueError('perceptual.comp must name 1..%d distinct features, got %r' % (len(PERCEPTUAL_WS), self.comp))
            if any(n not in SUPPORTED_COMP for n in self.comp):
                raise NotImplementedError('perceptual.comp entries must come from %r (the layers the shipped configs tap), got %r'
                                          % (SUPPORTED_COMP, self.comp))
        self.l1 = self.loss_kind == 'perceptual' and not cfg.perceptual.l2       # f_e = tf.abs (imm_model.py:132)
        if image_size % 16 or image_size < 64:
            raise ValueError('image side must be a multiple of 16 and >= 64')
        L.load()   # fail loudly, now, if the HIP library is missing
        self.cfg, self.B, self.S, self.dev, self.dt = cfg, int(batch), int(image_size), torch.device(device), act_dtype
        # act_dtype torch.float32 = the WITNESS engine (round 6): the same launch program with f32 activation storage and plain f32
        # convolution kernels (csrc/conv_f32.hip) — a test instrument (~100x slower), not a product path: the reference computes in
        # fp32 (imm_model.py:97) and only an exact-arithmetic run can tell storage noise from a wiring error.  The fused MFMA-only
        # launches (pose head, one-launch stride-2 data gradient, tap-in-epilogue) fall back to their unfused sequences.
        self.f32 = act_dtype == torch.float32
        self.K = int(cfg.n_maps)
        self.world_size = world_size
        # gradient-exchange buckets the backward program is laid out for (1 = one all-reduce after the backward pass; 2 = the
        # renderer's filter gradients are issued and reduced early so that their bucket can travel while the encoders' backward
        # runs).  Decided HERE, once, and read back by TrainStep (IMM_DP_BUCKETS is only the default of this argument).
        self.dp_buckets = int(dp_buckets if dp_buckets is not None else os.environ.get('IMM_DP_BUCKETS', '1'))
        self.use_mask = bool(cfg.loss_mask)
        self.n_cu, self.arch = ops.device_info()
        self._alloc_bytes = 0

        # ---- flat parameter / gradient / optimizer-state buffers -------------------------------
        self.spec = trainable_spec(cfg, self.S)
        sizes = [int(np.prod(s)) for _n, s, _w in self.spec]
        self.tab = ops.SegmentTable(sizes, [w for _n, _s, w in self.spec], self.dev)
        self.params = self._zeros(self.tab.total)
        self.grads = self._zeros(self.tab.total)
        self.adam_m = self._zeros(self.tab.total)
        self.adam_v = self._zeros(self.tab.total)
        self.pview, self.gview = OrderedDict(), OrderedDict()
        for i, (name, shape, _wd) in enumerate(self.spec):
            o0, o1 = self.tab.offsets[i], self.tab.offsets[i + 1]
            self.pview[name] = self.params[o0:o1].view(shape)
            self.gview[name] = self.grads[o0:o1].view(shape)
        self.opt_blk_partial = self._zeros(self.tab.nblk)
        self.seg_norm2 = self._zeros(self.tab.nseg)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=self.dev)    # TF global_step (lr schedule)
        self.adam_t = torch.zeros(1, dtype=torch.int32, device=self.dev)        # Adam updates applied to m/v (beta powers)
        self.lr_state = self._zeros(2)
        self.wd_loss = self._zeros(1)
        hp = dict(lr_start=1e-3, lr_decay=0.95, lr_step=100000, lr_multiple=1.0, beta1=0.9, beta2=0.999, eps=1e-8,
                  clip=1.0, grad_scale=1.0 / world_size, optim='adam', scale_growth_interval=1000, scale_max=2.0 ** 24)
        hp.update(hparams or {})
        # Loss scaling.  The reference computes in fp32 (imm_model.py:97); with f16 storage the gradient tensors of a trained
        # model (|dy| down to 1e-8) fall below f16's range, so the seeds of the backward pass are multiplied by a power of two
        # S (imm_perceptual_finalize) and the flat gradients divided by it inside imm_clip_adam_step, which also skips the
        # update and halves S when a gradient overflowed, and doubles S after `scale_growth_interval` clean steps.  All of it
        # lives in device memory (graph replay needs no host value).  bf16 has f32's range: no scaling.
        # hparams['loss_scale']: initial S (power of two), 0 / None = off; default 2^12 for f16, off for bf16.
        ls0 = hp.pop('loss_scale', 4096.0 if act_dtype == torch.float16 else 0.0) or 0.0
        if ls0 and (ls0 < 1.0 or math.log2(ls0) != int(math.log2(ls0))):
            raise ValueError('loss_scale must be a power of two >= 1, got %r' % (ls0,))
        self.loss_scale_state = None
        if ls0:
            self.loss_scale_state = self._zeros(4)      # {S, clean steps in a row, skipped steps, last step overflowed}
            self.loss_scale_state[0] = float(ls0)
        self._loss_scale_init = float(ls0)
        # scripts/train.py:97-104: Adam | Adadelta(rho 0.95, eps 1e-6) | Adagrad(initial accumulator 0.1)
        self.optim = str(hp['optim']).lower()
        if self.optim not in L.OPTIMIZERS:
            raise ValueError('Optimizer = %s not suppoerted' % hp['optim'])
        if self.optim == 'adadelta':
            hp.update(beta1=0.95, eps=1e-6)
        hp['optim'] = L.OPTIMIZERS[self.optim]
        self.hp = ops.OptHParams(**hp)

        # ---- non-trainable state ------------------------------------------------------------------
        self.state = OrderedDict()           # BN moving statistics, loss normalisers
        self.nfeat = len(self.comp)
        self.loss_agg = torch.tensor(PERCEPTUAL_WS[:self.nfeat], dtype=torch.float32, device=self.dev)
        self.vgg_w = OrderedDict()
        for k, v in (vgg_weights or synthetic_vgg_weights()).items():
            self.vgg_w[k] = torch.empty(tuple(v.shape), dtype=v.dtype, device=self.dev)
            ops.upload(self.vgg_w[k], v, k)               # (pinned staging + read-back: see ops.upload)

        self.prog_pack, self.prog_fwd, self.prog_bwd, self.prog_opt = [], [], [], []
        # IMM_TWO_STREAMS=0: everything on one stream (A/B of the two-lane schedule; the results are identical)
        self.two_streams = os.environ.get('IMM_TWO_STREAMS', '1') != '0'
        # IMM_NOL=1: normalise on load (round 4) — the batch-norm apply pass of a block folded into the LDS halo tiles of its two
        # readers (next convolution forward, its filter gradient) where both are LDS-halo kernels: encoder conv_1..3, renderer
        # conv_5 / conv_7.  Built, parity-tested, and MEASURED SLOWER on the step (3.137 -> 3.216 ms, same box, DESIGN.md item 47:
        # the affine + ReLU costs ~3.5 VALU instructions per element at one wave per SIMD with nothing to hide them behind —
        # forward convolutions +5..8 us for 11..15 us of apply pass saved, filter gradients +100 us), so it is off by default.
        self.nol = os.environ.get('IMM_NOL', '0') != '0' and not self.f32
        # IMM_DEBUG_SKIP_TAGS=tag,tag: timing experiment only (results become wrong): drop every launch whose tag is listed, to
        # measure how much of the step's critical path a kernel class occupies under graph replay / stream concurrency
        self._skip_tags = set(t for t in os.environ.get('IMM_DEBUG_SKIP_TAGS', '').split(',') if t)
        # IMM_DEBUG_STAMPS=marks|all: device wall-clock probes between the launches (lane boundaries / every launch) -> a
        # profiler-free timeline of a graph replay, read with stamp_report()
        self._stamp_mode = os.environ.get('IMM_DEBUG_STAMPS', '')
        self._stamp_buf = torch.zeros(8192, dtype=torch.int64, device=self.dev) if self._stamp_mode else None
        self._stamp_names = []
        self._side = None
        self._pack_jobs, self._reduce_jobs, self._wgrad_pending, self._colsum_pending = [], [], [], []
        self._training = True
        self._build_network()
        self.init_parameters(seed)

    # ------------------------------------------------------------------------------------------
    # allocation helpers
    # ------------------------------------------------------------------------------------------
    def _zeros(self, *shape, dtype=torch.float32):
        t = torch.zeros(*shape, dtype=dtype, device=self.dev)
        self._alloc_bytes += t.numel() * t.element_size()
        return t

    def _act(self, *shape):
        return self._zeros(*shape, dtype=self.dt)

    # ------------------------------------------------------------------------------------------
    # parameters
    # ------------------------------------------------------------------------------------------
    def init_parameters(self, seed=1):
        """truncated-normal(0.01) kernels, zero biases, gamma 1 / beta 0; Adam state reset."""
        rng = np.random.default_rng(seed)
        # the whole flat vector is built on the host and uploaded ONCE through pinned memory, verified (ops.upload: per-tensor
        # copies from pageable temporaries were the source of the one-rank-in-eight divergence of round 5)
        host = torch.zeros(self.tab.total, dtype=torch.float32)
        for i, (name, shape, _wd) in enumerate(self.spec):
            o0 = self.tab.offsets[i]
            n = int(np.prod(shape))
            if name.endswith('/w'):
                host[o0:o0 + n] = torch.from_numpy(truncated_normal(rng, shape, INIT_STD)).reshape(-1)
            elif name.endswith('/gamma'):
                host[o0:o0 + n] = 1.0
        ops.upload(self.params, host, 'the initial parameters')
        for k, v in self.state.items():
            v.fill_(1.0 if k.endswith('moving_variance') else 0.0)
        self.loss_agg.copy_(torch.tensor(PERCEPTUAL_WS[:self.nfeat]))
        self.reset_optimizer_slots(); self.grads.zero_(); self.step_count.zero_()
        self.run(self.prog_pack)

    def reset_optimizer_slots(self):
        """Fresh optimizer state: Adam m = v = 0, t = 0; Adadelta accumulators 0; Adagrad accumulator 0.1 (TF's
        initial_accumulator_value).  Slot buffers: adam_m (m / accum_update), adam_v (v / accum)."""
        self.adam_m.zero_()
        self.adam_v.fill_(0.1 if self.optim == 'adagrad' else 0.0)
        self.adam_t.zero_()
        if self.loss_scale_state is not None:
            self.loss_scale_state.zero_()
            self.loss_scale_state[0] = self._loss_scale_init

    def load_parameters(self, named, state=None):
        """named: {tf_variable_name: tensor}.  Missing names raise (no silent partial restore)."""
        for name in self.pview:
            ops.upload(self.pview[name], named[name], name)
        if state:
            for k, v in state.items():
                if k in self.state:
                    ops.upload(self.state[k], v, k)
                elif k.startswith('loss/') and k.endswith('_agg'):
                    if self.loss_kind == 'perceptual' and k[5:-4] in self.comp:
                        self.loss_agg[self.comp.index(k[5:-4])] = float(v)
                elif k.startswith('vgg16/'):
                    ops.upload(self.vgg_w[k], v, k)
                elif k == COST_EMA_STATE:
                    ops.upload(self.cost_ema, torch.as_tensor(v, dtype=torch.float32).reshape(4), k)
                else:
                    raise KeyError(k)
            self._pack_vgg()
        self.run(self.prog_pack)

    def named_parameters(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.pview.items())

    def named_state(self):
        out = OrderedDict((k, v.detach().clone()) for k, v in self.state.items())
        if self.loss_kind == 'perceptual':
            for i, name in enumerate(self.comp):
                out['loss/%s_agg' % name] = self.loss_agg[i].detach().clone()
        # the shadow variables of the cost moving averages (base_model.py:52-60: tf.train.Saver stores them, so a resumed run's
        # `_avg` curves continue) + the update count
        out[COST_EMA_STATE] = self.cost_ema.detach().clone()
        return out

    # ------------------------------------------------------------------------------------------
    # network construction
    # ------------------------------------------------------------------------------------------
    def _add(self, prog, fn, tag, flops=0.0, nbytes=0.0, name='', desc=None, variant=''):
        """desc: the imm_conv_desc of a convolution launch — its kernel variant (imm_conv2d_variant) is recorded with the launch
        (tools/layer_table.py: layer -> kernel -> time -> floor)."""
        cus = int(getattr(self, '_cur_cus', 0))
        if desc is not None:
            if cus:
                ops.set_cu_limit(cus)            # the variant the launch will take under its CU limit
            try:
                variant = '%s:%d' % ops.conv2d_variant(desc, self.dt)
            finally:
                if cus:
                    ops.set_cu_limit(0)
        prog.append(_Launch(fn, tag, flops, nbytes, name or getattr(self, '_cur_scope', ''), getattr(self, '_cur_lane', 0), variant,
                            cus, getattr(self, '_cur_group', '')))

    def _signal(self, prog, key, lane=None):
        """Record an event on `lane` (default: the current lane) that other lanes can wait for."""
        prog.append(_Launch(None, 'record:' + key, lane=self._cur_lane if lane is None else lane, group=getattr(self, '_cur_group', '')))

    def _wait(self, prog, key, lane):
        prog.append(_Launch(None, 'wait:' + key, lane=lane))

    def _mark(self, prog, what, lane=1):
        """'fork': side stream `lane` may start once everything issued so far on the main stream has finished; 'join': the main
        stream waits for side stream `lane`.  Launches registered with that lane in between run on the side stream: the two
        encoders (and their backward passes) are independent chains of small kernels that do not fill 256 CUs on their own
        (lane 1); lane 2 carries chip-filling work CONFINED to a share of the CUs (_Launch.cus) beside those chains: the
        ground-truth half of the frozen VGG16 forward, the renderer's filter gradients."""
        prog.append(_Launch(None, what, lane=lane, group=getattr(self, '_cur_group', '')))

    def _nol_consumer_ok(self, H, W, ci, ldx, co, k, stride, bn, out_f32):
        """Can a convolution of this geometry take the RAW output of the conv + BN + ReLU block in front of it (normalise on
        load)?  Both readers of the normalised tensor must rebuild it in LDS: the forward convolution (imm_conv2d_nol) and
        its filter gradient (an LDS-halo variant of imm_conv2d_wgrad_multi).  IMM_CONV_DISABLE=nol turns the path off."""
        if not self.nol or k != 3 or ci != ldx:
            return False
        ldy = ops.round_up(co, 4) if out_f32 else ops.round_up(co, 8)
        flags = L.CONV_BIAS | (L.CONV_OUT_F32 if out_f32 else 0) | (L.CONV_STATS if bn else 0)
        fd = ops.fwd_desc(self.B, H, W, ci, ldx, co, ldy, k, stride, flags)
        if not ops.conv2d_nol_supported(fd):
            return False
        lddy = ldy if not out_f32 else ops.round_up(co, 32)
        key, _wps, _units, _pcu = ops.conv2d_wgrad_variant(fd, lddy, self.dt)
        return key // 100000 == 2

    def _conv_block(self, scope, x, H, W, ci_real, ci_pad, ldx, co, k, stride, bn, relu, needs_dgrad,
                    out=None, ldo=None, out_f32=False, kw=None, up2x=False, fwd_launch=True, nol_src=None, defer_apply=False,
                    first_src=None):
        """Registers forward launches now and returns a layer record whose .backward(d_out, dx) registers the
        backward launches later (in reverse order).
        nol_src: the conv + BN + ReLU block in front of this convolution whose normalised output is NOT stored: x is its raw
        conv output y, and this convolution (and its filter gradient) apply relu(scale * y + shift) on load.
        defer_apply: this block is such a producer — statistics and a finalize launch, no apply pass, no `out`.
        first_src: the f32 image behind the tap-unrolled input x of the first encoder convolution: the forward launch reads IT
        (imm_conv_first builds the unrolled tile in LDS); x is then only the filter gradient's operand."""
        B, dt, dev = self.B, self.dt, self.dev
        self._cur_scope = scope
        lay = _ConvLayer()
        lay.scope, lay.bn, lay.relu, lay.x, lay.ldx = scope, bn, relu, x, ldx
        lay.nol_src = nol_src
        assert nol_src is None or (nol_src.bn and x is nol_src.y and ldx == nol_src.ldy)
        kw = k if kw is None else kw      # kw != k only for the tap-unrolled first encoder conv (7x1 over 21 channels)
        lay.ci_real, lay.ci_pad, lay.co, lay.k, lay.stride, lay.H, lay.W, lay.kw = ci_real, ci_pad, co, k, stride, H, W, kw
        flags = L.CONV_BIAS | (L.CONV_OUT_F32 if out_f32 else 0)
        ldy = ops.round_up(co, 4) if out_f32 else ops.round_up(co, 8)
        lay.ldy = ldy
        fd = ops.fwd_desc(B, H, W, ci_pad, ldx, co, ldy, k, stride, flags | (L.CONV_STATS if bn else 0), kw=kw)
        fd_eval = ops.fwd_desc(B, H, W, ci_pad, ldx, co, ldy, k, stride, flags, kw=kw)
        lay.fd, lay.Ho, lay.Wo = fd, fd.ho, fd.wo
        npix = B * fd.ho * fd.wo
        lay.npix = npix
        w, b = self.pview[scope + '/w'], self.pview[scope + '/b']
        rows = ops.round_up(co, 128)
        lay.wt = self._zeros(rows, fd.kpad, dtype=dt)
        self._pack_jobs.append(((w.data_ptr(), lay.wt.data_ptr(), 0, k, kw, ci_real, co, ci_pad, rows, fd.kpad), rows * fd.kpad))
        lay.y = self._zeros(B, fd.ho, fd.wo, ldy, dtype=torch.float32 if out_f32 else dt)
        flops = 2.0 * npix * k * kw * ci_real * co
        if bn:
            cus_q = int(getattr(self, '_cur_cus', 0))
            if cus_q:
                ops.set_cu_limit(cus_q)         # the partial rows of the tile plan the launch will take under its CU limit
            try:
                nblk = (ops.conv_first_stats_blocks(B, H) if first_src is not None else
                        ops.conv_stats_blocks(fd) if nol_src is None else ops.conv2d_nol_stats_blocks(fd))
            finally:
                if cus_q:
                    ops.set_cu_limit(0)
            lay.stats = self._zeros(nblk, 2, co)
            lay.scale, lay.shift, lay.mean, lay.rstd = (self._zeros(co) for _ in range(4))
            mm, mv = self._zeros(co), self._zeros(co)
            mv.fill_(1.0)
            self.state[scope + '/moving_mean'], self.state[scope + '/moving_variance'] = mm, mv
            gamma, beta = self.pview[scope + '/gamma'], self.pview[scope + '/beta']
            # an up-sampled block taken by the fused finalize + apply + up-sample pass never stores its own normalised tensor (nobody
            # reads it): it is not allocated either (a zero-filled `out` that looks valid).  lay.nol marks "normalise on
            # load" (the consumers read y); lay.out is None alone no longer means that.
            fused_up = bool(up2x) and not defer_apply and nblk <= BN_DIRECT_ROWS and co % 32 == 0
            if out is None and not defer_apply and not fused_up:
                out, ldo = self._act(B, fd.ho, fd.wo, co), co
            lay.out, lay.ldo = out, (ldo if out is not None else None)
            lay.nol = bool(defer_apply)

            def f_conv():
                if first_src is not None:
                    ops.conv_first(first_src, lay.wt, b, lay.y, ldy, lay.stats if self._training else None, B, H, co,
                                   L.CONV_BIAS | (L.CONV_STATS if self._training else 0))
                elif nol_src is not None:
                    ops.conv2d_nol(fd if self._training else fd_eval, x, nol_src.scale, nol_src.shift, nol_src.relu, lay.wt, b, lay.y,
                                   lay.stats if self._training else None)
                else:
                    ops.conv2d(fd if self._training else fd_eval, x, lay.wt, b, lay.y, lay.stats if self._training else None)

            def f_fin():
                ops.bn_finalize(lay.stats, nblk, co, npix, gamma, beta, BN_EPS, BN_MOMENTUM, self._training, mm, mv,
                                lay.scale, lay.shift, lay.mean, lay.rstd)
            cbytes = 2.0 * (B * H * W * ci_pad + npix * co + fd.kpad * co)
            self._add(self.prog_fwd, f_conv, 'conv_fwd', flops, cbytes, desc=(fd if (first_src is None and nol_src is None) else None),
                      variant=('first' if first_src is not None else 'halo:nol' if nol_src is not None else ''))
            lay.up = None
            if defer_apply:
                # normalise on load: the consumers rebuild relu(scale * y + shift) in LDS, so the apply pass (read y, write out:
                # 4 B per element of HBM traffic) is gone; what is left of the batch norm's forward is this finalize of the
                # partial rows (one workgroup per 32 channels; <= 768 rows: the producers are persistent kernels)
                assert up2x is False and out is None
                lay.out, lay.ldo = None, None
                self._add(self.prog_fwd, f_fin, 'bn_finalize')
            elif nblk <= BN_DIRECT_ROWS and co % 32 == 0:
                # few partial rows: the finalize is redone by every workgroup of the apply pass (one launch, one kernel
                # boundary and a 6-9 us latency chain less per layer); the renderer's x2 up-sampling rides along
                if up2x:
                    lay.up = self._act(B, 2 * fd.ho, 2 * fd.wo, co)
                # an up-sampled block's own normalised tensor has no reader (the next convolution and its filter gradient read the
                # up-sampled one, the batch-norm backward reads y): only `up` is written
                x_w = None if lay.up is not None else out
                assert (x_w is None) == fused_up
                self._add(self.prog_fwd, lambda: ops.bn_apply_fused(lay.stats, nblk, co, npix, gamma, beta, BN_EPS, BN_MOMENTUM,
                                                                    self._training, mm, mv, lay.scale, lay.shift, lay.mean, lay.rstd,
                                                                    lay.y, ldy, relu, x_w, (ldo if x_w is not None else co), lay.up, co, fd.ho, fd.wo),
                          'bn_apply', 0.0, npix * co * (4.0 + (8.0 if lay.up is not None else 0.0)))
            elif co % 32 == 0:
                # many partial rows: 32-row groups are summed by rows/32 workgroups in parallel, the fused apply pass finishes
                # the <= 32 group rows (instead of a finalize launch whose few workgroups walk every row)
                g = max(32, -(-nblk // 32))
                nred = -(-nblk // g)
                lay.stats_red = self._zeros(nred, 2, co)

                def f_red():
                    if self._training:
                        ops.rows_reduce(lay.stats, nblk, 2 * co, g, lay.stats_red)
                self._add(self.prog_fwd, f_red, 'bn_finalize')
                self._add(self.prog_fwd, lambda: ops.bn_apply_fused(lay.stats_red, nred, co, npix, gamma, beta, BN_EPS, BN_MOMENTUM,
                                                                    self._training, mm, mv, lay.scale, lay.shift, lay.mean, lay.rstd,
                                                                    lay.y, ldy, relu, out, ldo, None, co, fd.ho, fd.wo),
                          'bn_apply', 0.0, npix * co * 4.0)
            else:
                self._add(self.prog_fwd, f_fin, 'bn_finalize')
                self._add(self.prog_fwd, lambda: ops.bn_apply_relu(lay.y, npix, co, ldy, lay.scale, lay.shift, relu, out, ldo),
                          'bn_apply', 0.0, npix * co * 4.0)
        else:
            lay.out, lay.ldo = lay.y, ldy
            lay.fwd_flops = flops
            if fwd_launch and nol_src is not None:
                self._add(self.prog_fwd, lambda: ops.conv2d_nol(fd_eval, x, nol_src.scale, nol_src.shift, nol_src.relu, lay.wt, b, lay.y),
                          'conv_fwd', flops, 2.0 * (B * H * W * ci_pad + fd.kpad * co) + npix * co * (4.0 if out_f32 else 2.0))
            elif fwd_launch:      # (the pose head's convolution is part of the fused imm_pose_head_fwd launch instead)
                self._add(self.prog_fwd, lambda: ops.conv2d(fd_eval, x, lay.wt, b, lay.y), 'conv_fwd', flops,
                          2.0 * (B * H * W * ci_pad + fd.kpad * co) + npix * co * (4.0 if out_f32 else 2.0), desc=fd_eval)

        # ---- backward resources -------------------------------------------------------------------
        lay.needs_dgrad = needs_dgrad
        if needs_dgrad:
            assert kw == k
            # 16-bit gradient of an f32 head is stored with 32 channels so that its dgrad takes the fast
            # (one tap x 32 channels per K tile) path; the extra channels are zeros
            lddy = ldy if not out_f32 else ops.round_up(co, 32)
            lay.lddy = lddy
            lay.dd = None   # filled in backward()
            rows_d = ops.round_up(ci_real, 128)
            lay.s2 = ops.dgrad_s2_class_descs(B, H, W, ci_real, 0, lddy, lddy, k) if stride == 2 else None
            # stride 2, 3x3: the four parity classes as ONE launch over one dy halo (imm_conv2d_dgrad_s2: LDS-halo deep-K kernel,
            # four accumulator sets) where the shape is served; its filter image is the ordinary flipped one (mode 1)
            lay.s2_fused = (lay.s2 is not None and k == 3 and lddy % 64 == 0 and ci_real % 8 == 0 and not self.f32 and
                            ops.conv2d_dgrad_s2_supported(B, fd.ho, fd.wo, lddy, ci_real, ci_real))
            if lay.s2 is not None and not lay.s2_fused:
                # parity-class decomposition: 4 sub-filters (2x2, 2x1, 1x2, 1x1 taps) instead of a 4x-redundant
                # transposed gather over all 9 taps
                lay.wt_s2 = []
                for dd, mode in lay.s2:
                    wt_c = self._zeros(rows_d, dd.kpad, dtype=dt)
                    lay.wt_s2.append(wt_c)
                    self._pack_jobs.append(((w.data_ptr(), wt_c.data_ptr(), mode, k, k, ci_real, co, lddy, rows_d, dd.kpad),
                                            rows_d * dd.kpad))
                lay.wt_d = None
            else:
                kpad_d = ops.round_up(k * k * lddy, 32)
                lay.wt_d = self._zeros(rows_d, kpad_d, dtype=dt)
                self._pack_jobs.append(((w.data_ptr(), lay.wt_d.data_ptr(), 1, k, k, ci_real, co, lddy, rows_d, kpad_d),
                                        rows_d * kpad_d))
        else:
            lay.lddy = ldy if not out_f32 else ops.round_up(co, 32)
        # the filter gradient's pixel splits and slabs are decided when the layer's job joins a multi-problem launch
        # (_flush_wgrads): the split count depends on what else shares that launch
        if bn:
            lay.coef = self._zeros(3, co)
            lay.dy = self._act(B, fd.ho, fd.wo, ldy)
        else:
            lay.cs_partial = self._zeros(ops.colsum_blocks(npix, lay.lddy), lay.lddy)
        return lay

    def _conv_backward(self, lay, d_out, ldd, dx, lddx, up_src=None):
        """d_out: gradient w.r.t. the block output (post BN/ReLU for BN blocks; w.r.t. the conv output,
        16-bit with stride lay.lddy, otherwise).  dx: buffer receiving the input gradient (or None).
        up_src = (dy_up, lddy): the block's output was up-sampled x2 and dy_up is the gradient of the up-sampled tensor: the
        adjoint into d_out is taken inside the batch-norm backward reduction (imm_bn_bwd_reduce_up) instead of a launch of its own."""
        B, co, k = self.B, lay.co, lay.k
        npix = lay.npix
        scope = lay.scope
        self._cur_scope = scope
        gw, gb = self.gview[scope + '/w'], self.gview[scope + '/b']
        if lay.bn:
            gg, gbeta = self.gview[scope + '/gamma'], self.gview[scope + '/beta']
            gamma, beta = self.pview[scope + '/gamma'], self.pview[scope + '/beta']
            # sums (reduce) -> [parallel pre-reduction of > 256 partial rows] -> finalize + apply in one launch.  (Taking the sums
            # in the epilogue of the producer of d_out — built in round 2 behind IMM_BN_FUSE_BWD — removed the 23 reduce launches but
            # made the data-gradient epilogues slower by about as much and cost gradient precision: DESIGN.md item 18; finishing
            # the sums inside the reduce kernel with a last-workgroup ticket was slower too: item 22.  Both paths are gone.)
            nblk = ops.bn_bwd_blocks(npix, co)
            lay.bwd_partial = self._zeros(nblk, 2, co)
            if up_src is not None:
                self._add(self.prog_bwd, lambda: ops.bn_bwd_reduce_up(up_src[0], up_src[1], d_out, ldd, lay.y, lay.ldy, B, lay.Ho, lay.Wo,
                                                                      co, lay.scale, lay.shift, lay.mean, lay.rstd, lay.relu,
                                                                      lay.bwd_partial), 'bn_bwd_reduce', 0.0, npix * co * 12.0)
            else:
                self._add(self.prog_bwd, lambda: ops.bn_bwd_reduce(d_out, ldd, lay.y, lay.ldy, npix, co, lay.scale, lay.shift,
                                                                   lay.mean, lay.rstd, lay.relu, lay.bwd_partial),
                          'bn_bwd_reduce', 0.0, npix * co * 4.0)
            if co % 32 == 0:
                rows, nrows = lay.bwd_partial, nblk
                if nblk > BN_DIRECT_ROWS:
                    g = max(32, -(-nblk // 32))
                    nrows = -(-nblk // g)
                    lay.bwd_red = rows_red = self._zeros(nrows, 2, co)
                    self._add(self.prog_bwd, lambda: ops.rows_reduce(lay.bwd_partial, nblk, 2 * co, g, rows_red), 'bn_bwd_finalize')
                    rows = rows_red
                self._add(self.prog_bwd, lambda: ops.bn_bwd_apply_fused(rows, nrows, co, npix, gamma, d_out, ldd,
                                                                        lay.y, lay.ldy, lay.scale, lay.shift, lay.mean, lay.rstd,
                                                                        lay.relu, gg, gbeta, lay.dy, lay.ldy),
                          'bn_bwd_apply', 0.0, npix * co * 6.0)
            else:
                self._add(self.prog_bwd, lambda: ops.bn_bwd_finalize(lay.bwd_partial, nblk, co, npix, gamma, beta, lay.rstd,
                                                                     gg, gbeta, lay.coef), 'bn_bwd_finalize')
                self._add(self.prog_bwd, lambda: ops.bn_bwd_apply(d_out, ldd, lay.y, lay.ldy, npix, co, lay.scale, lay.shift,
                                                                  lay.mean, lay.rstd, lay.relu, lay.coef, lay.dy, lay.ldy),
                          'bn_bwd_apply', 0.0, npix * co * 6.0)
            dy, lddy = lay.dy, lay.ldy
            # conv bias feeds a batch norm: its gradient is analytically zero (sum of dy == 0); the
            # reference computes rounding noise there.  The flat gradient slice stays 0.
        else:
            dy, lddy = d_out, ldd
            # bias gradient of a convolution without batch norm (renderer head, pose head): nothing in the chain reads it, so the
            # launch is deferred to the image-encoder lane of the backward fork (90 us shorter than the pose lane): -25 us of
            # critical path (15 in front of the renderer's backward, 10.6 on the pose lane)
            self._colsum_pending.append((scope, lambda: ops.colsum(dy, npix, lddy, co, lddy, lay.cs_partial, gb)))
        fd = lay.fd
        flops = 2.0 * npix * k * lay.kw * lay.ci_real * co
        # Filter gradient: nobody reads it before the slab reduction at the end of the backward pass, so the job is only
        # COLLECTED here and issued with the other layers' in one multi-problem launch per kernel variant (_flush_wgrads):
        # the per-layer launches (15-48 us each) leave the serial BN-backward -> data-gradient chain.
        self._wgrad_pending.append((lay, dy, lddy, flops))
        if lay.needs_dgrad and dx is not None and getattr(lay, 's2_fused', False):
            assert lddx % 8 == 0 and lddx >= lay.ci_real, (lddx, lay.ci_real)
            self._add(self.prog_bwd, lambda: ops.conv2d_dgrad_s2(dy, lddy, lay.wt_d, dx, lddx, lay.ci_real, B, lay.Ho, lay.Wo),
                      'conv_dgrad', flops, 2.0 * (npix * lddy + 4 * npix * lay.ci_real + 9 * lddy * lay.ci_real), variant='s2d')
        elif lay.needs_dgrad and dx is not None and getattr(lay, 's2', None) is not None:
            # stride 2: the four input-pixel parity classes as one grouped launch (falls back to four launches inside the
            # library when the members do not take a grouped tile)
            classes = ops.dgrad_s2_class_descs(B, lay.H, lay.W, lay.ci_real, lddx, lddy, lddy, k)
            grp = ops.ConvGroup([dd0 for dd0, _m in classes], list(lay.wt_s2))
            ntaps = sum(dd0.kh * dd0.kw for dd0, _m in classes)
            self._add(self.prog_bwd, (lambda grp=grp: ops.conv2d_group(grp, dy, dx, None, None)), 'conv_dgrad',
                      2.0 * npix * ntaps * lay.ci_real * co,
                      2.0 * (4 * npix * lddy + 4 * npix * lay.ci_real + sum(dd0.kpad for dd0, _m in classes) * lay.ci_real), variant='group')
        elif lay.needs_dgrad and dx is not None:
            # output channels: the padded count when that makes whole 64-channel blocks (the concat layer: 266 -> 320, its
            # packed filter rows beyond ci_real are zeros) — the deep-K kernels need co % 64 == 0
            co_dx = lay.ci_pad if (lay.ci_pad != lay.ci_real and lay.ci_pad % 64 == 0 and lddx >= lay.ci_pad
                                   and lay.wt_d.shape[0] >= lay.ci_pad) else lay.ci_real
            dd = ops.dgrad_desc(B, lay.H, lay.W, co_dx, lddx, lddy, lddy, k, lay.stride, 0)
            assert dd.kpad == lay.wt_d.shape[1], (dd.kpad, lay.wt_d.shape)
            self._add(self.prog_bwd, lambda: ops.conv2d(dd, dy, lay.wt_d, None, dx), 'conv_dgrad', flops,
                      2.0 * (npix * lddy + B * lay.H * lay.W * lay.ci_real + dd.kpad * lay.ci_real), desc=dd)

    def _flush_wgrads(self, name, cus=0):
        """cus: size the members' splits for that many compute units instead of the whole chip (a confined lane).
        Issue the collected filter-gradient jobs as ONE multi-problem launch per kernel variant (imm_conv2d_wgrad_multi) and
        register their slab reductions.  Jobs that share a launch share the chip, so a layer no longer needs enough pixel
        splits to fill 256 CUs on its own: the splits of a group are sized to a common length per workgroup with about
        one round of resident workgroups in total (the library reports how many of a variant's workgroups fit a CU) — slab
        traffic (nsplit x |dW| x 4 B, written and read back by the reduction) falls with the split count."""
        jobs, self._wgrad_pending = self._wgrad_pending, []
        if not jobs:
            return
        groups = OrderedDict()
        for lay, dy, lddy, flops in jobs:
            key, wps, units, pcu = ops.conv2d_wgrad_variant(lay.fd, lddy, self.dt)
            groups.setdefault((key, pcu), []).append((lay, dy, lddy, wps, units, flops))
        chunks = plan_wgrad_chunks(groups, WGRAD_MULTI_MAX_JOBS, WGRAD_MULTI_MAX_VARIANTS)
        for ci_, chunk in enumerate(chunks):
            self._issue_wgrad_chunk(chunk, name if len(chunks) == 1 else '%s [%d/%d]' % (name, ci_ + 1, len(chunks)), cus)

    def _issue_wgrad_chunk(self, chunk, name, cus=0):
        # IMM_WG_LANES=n (round 6 experiment; default 1): the kernel variants of a chunk as separate launches
        # spread over n lanes (largest group on the main lane) instead of one after the other on one stream; IMM_WG_LANE_SHARE =
        # the share of the chip the groups on the side lanes are planned for
        n_lanes = int(os.environ.get('IMM_WG_LANES', '1'))
        if n_lanes > 1 and len(chunk) > 1 and not cus and self._cur_lane == 0:
            share = float(os.environ.get('IMM_WG_LANE_SHARE', '1.0'))
            groups = sorted(chunk, key=lambda g: -sum(m[5] for m in g[1]))
            lanes = sorted(set(1 + (i - 1) % (n_lanes - 1) for i in range(1, len(groups))))
            for ln in lanes:
                self._mark(self.prog_bwd, 'fork', lane=ln)
            for i, g in enumerate(groups):
                self._cur_lane = 0 if i == 0 else 1 + (i - 1) % (n_lanes - 1)
                self._issue_wgrad_chunk([g], '%s [variant %d]' % (name, g[0][0]),
                                        cus=0 if i == 0 else max(8, int(self.n_cu * share) // 8 * 8) if share < 1.0 else 0)
            self._cur_lane = 0
            for ln in lanes:
                self._mark(self.prog_bwd, 'join', lane=ln)
            return
        multi_jobs, flops_total = [], 0.0
        for (key, pcu), members in chunk:
            kind = key // 100000                    # 0 generic kernel (one launch per job), 1 transpose-read, 2 LDS-halo
            target = pcu * (cus or self.n_cu)
            # shortest useful workgroup: 16 steps of 32 pixels / 2 (sliced) or 4 (whole-filter) patches of 128 pixels
            floor_units = [16 if kind != 2 else (2 if wps > 1 else 4) for _l, _d, _ld, wps, _u, _f in members]

            def splits(u):
                return [max(1, min(-(-units // u), max(1, units // fl))) for (_l, _d, _ld, _w, units, _f), fl in zip(members, floor_units)]
            if kind == 0:
                ns = [max(1, min(-(-target // wps), max(1, units // 16))) for _l, _d, _ld, wps, units, _f in members]
            else:
                # the shortest common workgroup length whose total stays WITHIN the target: all workgroups of a group are equally
                # long, so one full round of resident workgroups is ideal and a single workgroup beyond it doubles the launch
                # (measured: 258 one-per-CU workgroups on 256 CUs, 84 us instead of ~45)
                u = max(units for _l, _d, _ld, _w, units, _f in members)
                ns = splits(u)
                while u > 1:
                    u2 = max(1, int(u / 1.05)) if u > 64 else u - 1
                    ns2 = splits(u2)
                    if sum(n * m[3] for n, m in zip(ns2, members)) > target:
                        break
                    u, ns = u2, ns2
            for (lay, dy, lddy, wps, units, flops), nsplit in zip(members, ns):
                lay.nsplit = nsplit
                flops_total += flops
                lay.slab = self._zeros(nsplit, lay.fd.kpad, lay.co)
                ns_ = getattr(lay, 'nol_src', None)
                multi_jobs.append((lay.fd, lay.x, dy, lddy, lay.slab, nsplit,
                                   None if ns_ is None else (ns_.scale, ns_.shift, ns_.relu)))
                gw = self.gview[lay.scope + '/w']
                job = (lay.slab.data_ptr(), gw.data_ptr(), nsplit, lay.k * lay.kw, lay.ci_pad, lay.ci_real, lay.co, lay.fd.kpad)
                self._reduce_jobs.append((job, lay.k * lay.kw * lay.ci_real * lay.co))
        multi = ops.WgradMulti(multi_jobs, self.dt)
        self._wgrad_multis = getattr(self, '_wgrad_multis', []) + [multi]
        self._cur_scope = name
        self._add(self.prog_bwd, lambda: ops.conv2d_wgrad_multi(multi), 'conv_wgrad', flops_total, name=name)

    def _build_network(self):
        cfg, B, S, K, dt = self.cfg, self.B, self.S, self.K, self.dt
        nf = cfg.n_filters
        self.in_image = self._zeros(B, S, S, 3)
        self.in_future = self._zeros(B, S, S, 3)
        self.in_mask = self._zeros(B, S, S) if self.use_mask else None
        sizes = render_sizes(cfg, S)
        s16 = sizes[-1]
        if s16 != 16:
            raise NotImplementedError('renderer starts at 16x16 (min_res 16, renderer_stride 2): got %d' % s16)
        He = S // 8                        # encoder output side
        Cj = ops.round_up(8 * nf + K, 64)  # joint embedding channels, zero padded to whole 64-channel K slices (deep-K kernels)
        self.Cj, self.He = Cj, He
        self.joint = self._act(B, 16, 16, Cj)
        self.d_joint = self._act(B, 16, 16, Cj)

        # ---- encoders -------------------------------------------------------------------------------
        def build_encoder(scope, src):
            # conv_1 is 7x7 over 3 channels: feed it the image with the 7 horizontal taps unrolled into 21(+11
            # zero) channels, i.e. run it as a 7x1 convolution over 32 channels (same arithmetic, HWIO weights
            # [7,7,3,co] are bit-identical to [7,1,21,co])
            k1 = encoder_spec(nf)[0][0]
            ld1 = ops.round_up(3 * k1, 32)
            xin = self._act(B, S, S, ld1)
            f_pack = (lambda: ops.pack_image_taps(src, xin, B, S, S, k1, (k1 - 1) // 2, ld1), 'pack_image',
                      0.0, B * S * S * (12.0 + 2.0 * ld1), scope + '/pack')
            # imm_conv_first reads the f32 image itself: the tap-unrolled copy is then only the operand of conv_1's filter gradient
            # (the end of the backward pass) and its packing leaves the head of the forward chain for the tail of the shorter lane
            co1 = encoder_spec(nf)[0][2]
            # (IMM_CONV_FIRST=1: built, bit-compatible, measured NEUTRAL on the step — 26.7 us against 12 + 20 us for the two launches it
            # takes off the chain, whose packing pass still runs at the lane's tail; DESIGN.md item 49 — so the default stays the 7x1 form)
            direct = (os.environ.get('IMM_CONV_FIRST', '0') != '0' and not self.f32 and k1 == 7 and ld1 == 32 and
                      ops.conv_first_supported(B, S, co1, ops.round_up(co1, 8)))
            if direct:
                self._deferred_packs.append(f_pack)
            else:
                self._add(self.prog_fwd, f_pack[0], f_pack[1], f_pack[2], f_pack[3], name=f_pack[4])
            layers, x, H, ci_real, ci_pad, ldx = [], xin, S, 3 * k1, ld1, ld1
            spec = encoder_spec(nf)
            # normalise on load: block i keeps only its raw conv output when the next convolution (its ONLY reader, forward
            # and filter gradient) can rebuild the normalised tensor in LDS — the high-resolution layers, where the apply
            # pass is pure HBM traffic (conv_2, conv_3, conv_4 of each encoder at 128x128 / 64x64)
            hs, h_ = [], S
            for (_k, _ci, _co, st) in spec:
                hs.append(h_); h_ = -(-h_ // st)
            defer = [i + 1 < len(spec) and self._nol_consumer_ok(hs[i + 1], hs[i + 1], spec[i][2], ops.round_up(spec[i][2], 8),
                                                               spec[i + 1][2], spec[i + 1][0], spec[i + 1][3], True, False)
                     for i in range(len(spec))]
            prev = None
            for i, (k, ci, co, stride) in enumerate(spec):
                last = i == len(spec) - 1
                out, ldo = (None, None)
                if last and scope == 'model/image_encoder' and He == 16:
                    out, ldo = self.joint, Cj       # conv_8 writes straight into the concat buffer
                nol_src = prev if (prev is not None and getattr(prev, 'nol', False)) else None
                lay = self._conv_block('%s/encoder/conv_%d' % (scope, i + 1), x, H, H, ci_real, ci_pad, ldx, co, k,
                                       stride, True, True, needs_dgrad=(i > 0), out=out, ldo=ldo, kw=(1 if i == 0 else None),
                                       nol_src=nol_src, defer_apply=defer[i], first_src=(src if (i == 0 and direct) else None))
                layers.append(lay)
                prev = lay
                if lay.nol:
                    x, H, ci_real, ci_pad, ldx = lay.y, lay.Ho, co, co, lay.ldy
                else:
                    x, H, ci_real, ci_pad, ldx = lay.out, lay.Ho, co, co, lay.ldo
            return layers

        self._deferred_packs = []
        self._mark(self.prog_fwd, 'fork')
        self._cur_lane = 1                      # image encoder: side stream, concurrent with the pose encoder
        # the weight-decay term of the loss depends on the parameters only: computed here, beside the (longer) pose branch,
        # instead of in front of the loss where it sat on the critical path (23 us)
        self._add(self.prog_fwd, lambda: ops.weight_decay_loss(self.params, self.tab, self.opt_blk_partial, self.wd_loss),
                  'wd_loss', name='loss/weight_decay')
        self.enc_im = build_encoder('model/image_encoder', self.in_image)
        if He != 16:   # imm_model.py:324-335: align_corners resize of the 8f-channel embedding down to 16x16
            e = self.enc_im[-1]
            self._add(self.prog_fwd, lambda: ops.resize_ac_fwd(e.out, self.joint, B, He, He, 16, 16, 8 * nf, e.ldo, Cj),
                      'resize_ac')
        self._cur_lane = 0
        self.enc_pose = build_encoder('model/pose_encoder', self.in_future)
        pe = self.enc_pose[-1]
        # the pose head (1x1 convolution -> soft-argmax -> Gaussian maps, imm_model.py:247-264) as ONE launch each way where the
        # shapes allow (every shipped configuration): -1 launch forward, -2 backward on the pose lane, the longer one
        lddy_h = ops.round_up(K, 32)
        self.fused_head = (not self.f32 and (8 * nf) % 32 == 0 and K <= 64 and (He * He) % 16 == 0 and lddy_h in (32, 64) and
                           4 * (He * He * K + 2 * He * K + 2 * K) <= 158 * 1024 and
                           4 * ((2 + 2 * He) * K + 516) + He * He * lddy_h * 2 <= 158 * 1024)
        self.pose_head = self._conv_block('model/pose_encoder/conv_1', pe.out, He, He, 8 * nf, 8 * nf, pe.ldo, K, 1, 1,
                                          False, False, needs_dgrad=True, out_f32=True, fwd_launch=not self.fused_head)
        ph = self.pose_head
        assert ph.lddy == lddy_h
        self.heat, self.ldh = ph.y, ph.ldy
        self.mu = self._zeros(B, K, 2)
        self.py = self._zeros(B, He, K)
        self.px = self._zeros(B, He, K)
        self.inv_std = 1.0 / float(cfg.gauss_std)
        gview = self.joint[..., 8 * nf:]
        if self.fused_head:
            bias_h = self.pview['model/pose_encoder/conv_1/b']
            self._add(self.prog_fwd, lambda: ops.pose_head_fwd(pe.out, pe.ldo, 8 * nf, ph.wt, bias_h, B, He, He, K, self.inv_std, 16,
                                                               self.heat, self.ldh, self.mu, self.py, self.px, gview, Cj, dt,
                                                               cfg.gauss_mode), 'bottleneck', ph.fwd_flops)
        else:
            self._add(self.prog_fwd, lambda: ops.softargmax_gauss_fwd(self.heat, self.ldh, B, He, He, K, self.inv_std, 16,
                                                                      self.mu, self.py, self.px, gview, Cj, dt, cfg.gauss_mode), 'bottleneck')

        if self._deferred_packs:
            # the tap-unrolled image copies (filter-gradient operands of the two first convolutions): at the tail of the image-encoder
            # lane, which is shorter than the pose lane by the pose head
            self._cur_lane = 1
            for fn, tag, fl, nb, nm in self._deferred_packs:
                self._cur_scope = nm
                self._add(self.prog_fwd, fn, tag, fl, nb, name=nm)
            self._cur_lane = 0
        self._mark(self.prog_fwd, 'join')
        # ---- renderer ---------------------------------------------------------------------------------
        self.ren, self.ren_up = [], []
        x, H, ci_real, ci_pad, ldx = self.joint, 16, 8 * nf + K, Cj, Cj
        rspec = renderer_spec(cfg, S, n_renderer_out(cfg))
        prev = None
        for i, (k, ci, co, bn, up) in enumerate(rspec):
            assert ci == ci_real, (ci, ci_real)
            # normalise on load (see build_encoder): a block that is not up-sampled and whose next convolution takes the raw tensor
            defer = False
            if bn and not up and i + 1 < len(rspec) and co % 8 == 0:
                k2, _ci2, co2, bn2, _up2 = rspec[i + 1]
                defer = self._nol_consumer_ok(H, H, co, co, co2, k2, 1, bn2, not bn2)
            nol_src = prev if (prev is not None and prev.bn and getattr(prev, 'nol', False)) else None
            lay = self._conv_block('model/renderer/conv_%d' % (i + 1), x, H, H, ci_real, ci_pad, ldx, co, k, 1, bn, bn,
                                   needs_dgrad=True, out_f32=not bn, up2x=bool(up and bn), nol_src=nol_src, defer_apply=defer)
            self.ren.append(lay)
            prev = lay
            if lay.bn and lay.nol:
                x, ci_real, ci_pad, ldx = lay.y, co, co, lay.ldy
            else:       # (an up-sampled block without a stored `out`: x becomes the up-sampled tensor below, pixel stride co)
                x, ci_real, ci_pad, ldx = lay.out, co, co, (lay.ldo if lay.out is not None else co)
            if up:
                ub = getattr(lay, 'up', None)
                if ub is None:       # not taken by the fused finalize + apply + up-sample pass
                    ub = self._act(B, 2 * H, 2 * H, co)
                    src = lay.out
                    assert src is not None, 'up-sampling pass without a stored normalised tensor'

                    self._add(self.prog_fwd, (lambda src=src, ub=ub, H=H, co=co: ops.upsample2x_fwd(src, ub, B, H, H, co, co, co)),
                              'upsample', 0.0, B * H * H * co * 10.0)
                self.ren_up.append((len(self.ren) - 1, ub, H, co))
                x, H = ub, 2 * H
        self.pred = self.ren[-1].y              # f32 [B,S,S,ldp]; channels 0..2 = future_im_pred
        self.ldp = self.ren[-1].ldy

        self.n_fwd_model = len(self.prog_fwd)   # launches up to here produce future_im_pred / gauss_yx

        # ---- frozen VGG16 on concat([gt, pred]) --------------------------------------------------------
        # The activations keep the reference's concat layout [gt images; pred images] (imm_model.py:126): every VGG launch
        # covers 2B images.  (Running the gt half — which does not depend on the network — ahead on another lane was measured
        # twice without gain: the VGG kernels fill the chip on their own)
        # Only the layers up to the deepest tapped one are built; no tapped layer (perceptual.comp == ['input'] or
        # reconstruction_loss 'l2') => no VGG at all.
        self.vgg_act, self.vgg_wt, self.vgg_wtd, self.vgg_desc = OrderedDict(), {}, {}, {}
        self.vgg_pool = {}
        self.tap_idx = {n: k for k, n in enumerate(self.comp)}          # feature name -> position in the loss
        vnames = [n for n, _ci, _co in VGG_LAYERS]
        taps = [n for n in self.comp if n != 'input'] if self.loss_kind == 'perceptual' else []
        self.vgg_layers = VGG_LAYERS[:max(vnames.index(n) for n in taps) + 1] if taps else []
        nfeat = self.nfeat
        self.sse_partial = self._zeros(nfeat, L.SSE_BLOCKS)
        fuse_ok = not self.l1                    # the fused SSE+pool / unpool+tap passes exist for the squared error only
        fused_sse = set()
        # the image-space term ('input' feature: prediction against the future image) needs the renderer's output only: with
        # IMM_SSE_INPUT_LANE=1 its error sum runs on lane 1 beside the VGG launches instead of in the one-lane tail in front of the
        # loss.  Round 6, measured NEUTRAL (same box, two alternations: 3.2027 / 3.2077, 3.1943 / 3.1950 ms — the 11 us launch
        # leaves the tail, the extra fork / join nodes cost as much): off by default.
        self._sse_input_early = ('input' in self.tap_idx and bool(self.vgg_layers) and
                                 os.environ.get('IMM_SSE_INPUT_LANE', '0') != '0')
        if self._sse_input_early:
            idx0 = self.tap_idx['input']
            self._cur_scope = 'loss'
            self._mark(self.prog_fwd, 'fork', lane=1)
            self._cur_lane = 1
            self._add(self.prog_fwd, lambda: ops.masked_sse_f32(self.in_future, 3, self.pred, self.ldp, B, S, 3, self.in_mask,
                                                                self.sse_partial[idx0], self.l1), 'sse')
            self._cur_lane = 0
        # Round 6, built and MEASURED WITHOUT GAIN, off by default (IMM_VGG_SPLIT=1 turns it on): the
        # ground-truth half of concat([gt, pred]) (imm_model.py:126) depends on the input batch only.  With the split it runs on a
        # lane of its own (lane 2) that forks at the START of the forward program, every persistent convolution CONFINED to
        # `gt_cus` compute units (imm_set_cu_limit; IMM_GT_CUS, 0 = no limit); the prediction half follows the renderer on the
        # main lane with the whole chip and waits for the ground-truth activation of a tapped layer right in front of that
        # layer's error sum.  Same box, ms/step: one launch per layer 3.193 / 3.206 | split with 128 / 96 / 64 / all CUs 3.199 /
        # 3.238 / 3.256 / 3.261 (profiles/r06_lanes_ab.txt).  The per-lane stamps (profiles/r06_lanes_timeline_split.txt) say why:
        # the lane's 0.29 ms of chip time take 0.80 ms beside the chains, the encoder lanes finish at 0.58 ms instead of 0.40, and
        # the half-batch launches of the deep layers (conv4_x / conv5_x: 128 tiles) under-fill the chip the prediction half then
        # has to itself (0.37 ms for what the 2B-image launches do in 0.25).  The chains' "latency-bound" launches are not idle
        # CUs: every one of them puts a 119 KB-LDS workgroup on each of the 256 CUs and runs it at a third of the CU's matrix
        # rate, and neither LDS nor registers leave room for a second kernel's workgroup on the same CU — so a CU given to the
        # lane is a CU taken from the chains.
        split = bool(self.vgg_layers) and os.environ.get('IMM_VGG_SPLIT', '0') != '0'
        self.vgg_split = split
        self.gt_cus = int(os.environ.get('IMM_GT_CUS', '128')) if split else 0
        halves = ([('gt', slice(0, B), 2, self.gt_cus), ('pred', slice(B, 2 * B), 0, 0)] if split else
                  [('', slice(0, 2 * B), 0, 0)])
        gt_prog = []
        for hname, hs, lane, cus in halves:
            if not self.vgg_layers:
                break
            prog = gt_prog if hname == 'gt' else self.prog_fwd
            self._cur_lane, self._cur_cus, self._cur_group = lane, cus, ('gt' if hname == 'gt' else '')
            sfx = ' [%s]' % hname if hname else ''
            nb = hs.stop - hs.start
            if hname != 'pred':
                self.w11 = self._zeros(9, 64); self.b11 = self._zeros(64)
                a = self._act(2 * B, S, S, 64)
                self.vgg_act['conv1_1'] = (a, S)
            a = self.vgg_act['conv1_1'][0]
            hv = {'': 3, 'gt': 1, 'pred': 2}[hname]
            # The head in ONE launch (round 6, imm_vgg_head_fwd: vgg_head.hip): conv1_2's persistent workgroups produce the conv1_1
            # halo of their patches on the matrix cores, so conv1_1's 2B-image activation (the largest tensor of the step) is neither
            # written nor read back; only its prediction half is stored (the ReLU mask of conv1_2's data gradient).  IMM_VGG_HEAD=0 /
            # IMM_CONV_DISABLE=vgg_head / the split program / the f32 witness keep the two launches.
            head_fused = (not split and len(self.vgg_layers) > 1 and not self.f32 and os.environ.get('IMM_VGG_HEAD', '1') != '0'
                          and ops.vgg_head_supported(B, S, dt))
            self.vgg_head_fused = head_fused
            if head_fused:
                self.vgg_gray = torch.empty(ops.vgg_head_scratch_bytes(B, S), dtype=torch.uint8, device=self.dev)
            else:
                self._add(prog, (lambda a=a, hv=hv: ops.vgg_conv1_1_fwd(self.in_future, self.pred, self.ldp, B, S, self.w11, self.b11, a, hv)),
                          'vgg_conv1_1', 2.0 * nb * S * S * 9 * 64, nb * S * S * 128.0, name='vgg16/conv1_1' + sfx)
            x, H = a, S
            for li, (name, cin, cout) in enumerate(self.vgg_layers[1:], start=1):
                fd = ops.fwd_desc(nb, H, H, cin, cin, cout, cout, 3, 1, L.CONV_BIAS | L.CONV_RELU)
                if hname != 'pred':
                    wt = self._zeros(ops.round_up(cout, 128), fd.kpad, dtype=dt)
                    wtd = self._zeros(ops.round_up(cin, 128), ops.round_up(9 * cout, 32), dtype=dt)
                    y = self._act(2 * B, H, H, cout)
                    self.vgg_wt[name], self.vgg_wtd[name], self.vgg_desc[name] = wt, wtd, fd
                    self.vgg_act[name] = (y, H)
                wt, y = self.vgg_wt[name], self.vgg_act[name][0]
                bias = self.vgg_w['vgg16/%s/biases' % name]
                if li == 1 and head_fused:
                    # algorithmic bytes: the two images in, conv1_1's prediction half and conv1_2's activation out
                    self._add(prog, (lambda a=a, wt=wt, bias=bias, y=y: ops.vgg_head_fwd(
                        self.in_future, self.pred, self.ldp, B, S, self.w11, self.b11, wt, bias, a, B, y, self.vgg_gray)),
                              'vgg_fwd', 2.0 * nb * S * S * 9 * (64 + cin * cout),
                              nb * S * S * 12.0 + 2.0 * (B * S * S * 64 + nb * S * S * cout + 9 * cin * cout),
                              name='vgg16/conv1_1+' + name, variant='vgg_head')
                else:
                    self._add(prog, (lambda fd=fd, x=x, wt=wt, bias=bias, y=y, hs=hs: ops.conv2d(fd, x[hs], wt, bias, y[hs])),
                              'vgg_fwd', 2.0 * nb * H * H * 9 * cin * cout, 2.0 * (nb * H * H * (cin + cout) + 9 * cin * cout),
                              name='vgg16/' + name + sfx, desc=fd)
                if hname == 'gt' and name in taps:
                    self._signal(prog, 'gt:' + name)
                x = y
                if name in VGG_POOL_AFTER and li < len(self.vgg_layers) - 1:      # the deepest layer is not pooled: nobody reads it
                    if hname != 'pred':
                        self.vgg_pool[name] = self._act(2 * B, H // 2, H // 2, cout)
                    p = self.vgg_pool[name]
                    if name in taps and fuse_ok and hname != 'gt':
                        # the loss taps this layer AND it is pooled next: one pass computes the masked SSE of the two halves
                        # and both pooled halves (the feature map is read once instead of twice); with the ground-truth lane the
                        # pooled ground-truth half is that lane's (imm_maxpool2_fwd below) and this pass writes the prediction's
                        idx = self.tap_idx[name]
                        fused_sse.add(name)
                        if hname == 'pred':
                            self._wait(prog, 'gt:' + name, lane=0)
                        pa = None if hname == 'pred' else p[:B]
                        self._add(prog, (lambda x=x, p=p, pa=pa, H=H, cout=cout, idx=idx: ops.masked_sse_pool(
                            x[:B], x[B:], B, H, cout, self.in_mask, S, self.sse_partial[idx], pa, p[B:])), 'sse',
                                  0.0, 2 * B * H * H * cout * 2.5, name='vgg16/%s+pool' % name)
                    else:
                        self._add(prog, (lambda x=x, p=p, H=H, cout=cout, hs=hs, nb=nb: ops.maxpool2_fwd(x[hs], p[hs], nb, H, H, cout)),
                                  'maxpool', 0.0, nb * H * H * cout * 2.5, name='vgg16/pool_' + name + sfx)
                    x, H = p, H // 2
        self._cur_lane, self._cur_cus, self._cur_group = 0, 0, ''
        if gt_prog:
            # the ground-truth lane forks in front of everything else (list order = host issue order; the lane's stream is
            # released by an event at the head of the main stream); it is joined in front of the loss below
            self._cur_group = 'gt'
            head = []
            self._mark(head, 'fork', lane=2)
            self._cur_group = ''
            self.prog_fwd[0:0] = head + gt_prog
            self.n_fwd_model += len(head) + len(gt_prog)
        if self.vgg_layers:
            self._pack_vgg()

        # ---- loss -------------------------------------------------------------------------------------------
        nel = []
        for name in self.comp:
            if name == 'input':
                nel.append(float(B * S * S * 3))
            else:
                y, H = self.vgg_act[name]
                nel.append(float(B * H * H * y.shape[-1]))
        self.nel = torch.tensor(nel, dtype=torch.float32, device=self.dev)
        self.loss_out = self._zeros(3 * nfeat + 3)
        # BaseModel._add_cost_summary (base_model.py:52-60): moving averages of reconstruction_loss / weights_loss / loss_total,
        # {shadow[3], local_step}, advanced by every TRAINING step (imm_cost_ema, at the head of the image-encoder lane's backward
        # pass: off the critical path); cost_summaries() reports the shadow as it is: TF 1.10's ExponentialMovingAverage does not
        # zero-debias (zero_debias=False), the `_avg` curves start at 0 like the reference's
        self.cost_ema = self._zeros(4)
        mask = self.in_mask
        l1 = self.l1
        self._cur_scope = 'loss'
        if self._sse_input_early:
            self._mark(self.prog_fwd, 'join', lane=1)
        tail = [name for name in taps if name not in fused_sse]
        # the image pair's error sum rides in the launch of the deep layers' sums (imm_masked_sse_all, round 6: one launch less on the
        # one-lane path in front of the loss; IMM_SSE_ALL=0 keeps the two launches — same partial sums bit for bit)
        sse_all = ('input' in self.tap_idx and not self._sse_input_early and len(tail) > 1 and
                   os.environ.get('IMM_SSE_ALL', '1') != '0')
        if 'input' in self.tap_idx and not self._sse_input_early and not sse_all:
            idx0 = self.tap_idx['input']
            self._add(self.prog_fwd, lambda: ops.masked_sse_f32(self.in_future, 3, self.pred, self.ldp, B, S, 3, mask,
                                                                self.sse_partial[idx0], l1), 'sse')
        if self.vgg_split:
            # the last launch of the ground-truth lane produced the deepest tapped activation: the lane is joined here (every tap
            # that is not pooled next — conv3_2, conv4_2, conv5_2 — is read by the error sums below)
            self._cur_group = 'gt'
            self._mark(self.prog_fwd, 'join', lane=2)
            self._cur_group = ''
        if len(tail) > 1:
            # the deep tapped layers' error sums in one launch (they sit back to back in front of the loss)
            feats = []
            for name in tail:
                y, H = self.vgg_act[name]
                feats.append((y[:B], y[B:], H, y.shape[-1], self.sse_partial[self.tap_idx[name]]))
            self.sse_multi = ops.SseMulti(feats)
            if sse_all:
                idx0 = self.tap_idx['input']
                self._add(self.prog_fwd, lambda: ops.masked_sse_all(self.sse_multi, B, mask, S, self.in_future, 3, self.pred, self.ldp, 3,
                                                                    self.sse_partial[idx0], l1), 'sse', 0.0,
                          sum(2 * B * f[2] * f[2] * f[3] * 2.0 for f in feats) + B * S * S * (3 + self.ldp + 1) * 4.0)
            else:
                self._add(self.prog_fwd, lambda: ops.masked_sse_multi(self.sse_multi, B, mask, S, l1), 'sse', 0.0,
                          sum(2 * B * f[2] * f[2] * f[3] * 2.0 for f in feats))
            tail = []
        for name in tail:
            idx = self.tap_idx[name]
            y, H = self.vgg_act[name]
            c = y.shape[-1]
            self._add(self.prog_fwd, (lambda y=y, H=H, c=c, idx=idx: ops.masked_sse(y[:B], y[B:], B, H, c, mask, S,
                                                                                  self.sse_partial[idx], l1)), 'sse',
                      0.0, 2 * B * H * H * c * 2.0)
        mode = ops.LOSS_L2 if self.loss_kind == 'l2' else ops.LOSS_PERCEPTUAL
        self._add(self.prog_fwd, lambda: ops.perceptual_finalize(self.sse_partial, nfeat, self.nel, self.loss_agg,
                                                                 self._training, self.wd_loss, self.loss_out, l1, mode,
                                                                 self.loss_scale_state), 'loss_finalize')
        self.coef = self.loss_out[2 * nfeat:3 * nfeat]

        self._build_backward()
        # one table-driven launch re-packs every trainable kernel (forward + dgrad layouts) after an update
        self.pack_tab = ops.pack_table([j for j, _n in self._pack_jobs], self.dev)
        self._add(self.prog_pack, lambda: ops.pack_weights_multi(self.pack_tab, dt), 'pack')

        # ---- optimizer --------------------------------------------------------------------------------------
        self._add(self.prog_opt, lambda: ops.clip_adam_step(self.params, self.grads, self.adam_m, self.adam_v, self.tab,
                                                            self.opt_blk_partial, self.seg_norm2, self.step_count,
                                                            self.adam_t, self.lr_state, self.hp, self.loss_scale_state),
                  'clip_adam', 0.0, self.tab.total * 36.0)
        self.prog_opt.extend(self.prog_pack)

    def _pack_vgg(self):
        if not self.vgg_layers:
            return
        w = self.vgg_w
        self.w11.copy_(w['vgg16/conv1_1/weights'].reshape(9, 64))
        self.b11.copy_(w['vgg16/conv1_1/biases'])
        for name, cin, cout in self.vgg_layers[1:]:
            fd = self.vgg_desc[name]
            wm = w['vgg16/%s/weights' % name]
            ops.pack_weights(wm, self.vgg_wt[name], 0, 3, 3, cin, cout, cin, self.vgg_wt[name].shape[0], fd.kpad)
            wtd = self.vgg_wtd[name]
            ops.pack_weights(wm, wtd, 1, 3, 3, cin, cout, cout, wtd.shape[0], wtd.shape[1])

    def _build_backward(self):
        B, S, K, dt, Cj = self.B, self.S, self.K, self.dt, self.Cj
        mask = self.in_mask
        acts = self.vgg_act
        l1 = self.l1
        names = [n for n, _ci, _co in self.vgg_layers]
        taps = self.tap_idx
        # gradient buffers w.r.t. the (pred half of the) VGG activations, reused in place as dz after masking
        dbuf = {n: self._act(B, acts[n][1], acts[n][1], acts[n][0].shape[-1]) for n in names}
        dpool = {n: self._act(B, acts[n][1] // 2, acts[n][1] // 2, acts[n][0].shape[-1]) for n in self.vgg_pool}

        def pred_half(t):
            return t[B:]

        def tap(name, has_in):
            y, H = acts[name]
            c = y.shape[-1]
            self._add(self.prog_bwd, lambda: ops.tap_grad(dbuf[name], has_in, y[B:], y[:B], B, H, c, mask, S, self.coef,
                                                          taps[name], True, l1), 'tap_grad', 0.0, B * H * H * c * 8.0)

        fused_taps = set()

        def dgrad(name, dst, mask_ref, tap_of=None):
            """conv `name`: dz(name) -> gradient w.r.t. its input written to dst (masked by mask_ref>0 if given).  tap_of: the
            input IS a tapped activation: its feature-loss term and ReLU backward go into this launch's epilogue
            (imm_conv2d_tap) when the kernel that takes the shape has that epilogue, instead of a pass of their own."""
            cin, cout = [(ci, co) for n, ci, co in VGG_LAYERS if n == name][0]
            H = acts[name][1]
            wtd = self.vgg_wtd[name]
            src = dbuf[name]
            flops = 2.0 * B * H * H * 9 * cin * cout
            if tap_of is not None:
                dt0 = ops.dgrad_desc(B, H, H, cin, cin, cout, cout, 3, 1, 0)
                if not self.f32 and ops.conv2d_tap_supported(dt0):
                    ya = acts[tap_of][0]
                    self._add(self.prog_bwd, lambda: ops.conv2d_tap(dt0, src, wtd, dst, ya[B:], ya[:B], cin, mask, S, self.coef,
                                                                    taps[tap_of], l1), 'vgg_dgrad', flops,
                              2.0 * (B * H * H * (3 * cin + cout) + 9 * cin * cout), name='vgg16/%s+tap(%s)' % (name, tap_of), desc=dt0)
                    fused_taps.add(tap_of)
                    return
            flags = L.CONV_MASK if mask_ref is not None else 0
            dd = ops.dgrad_desc(B, H, H, cin, cin, cout, cout, 3, 1, flags, ldmask=cin)
            self._add(self.prog_bwd, lambda: ops.conv2d(dd, src, wtd, None, dst, None, mask_ref), 'vgg_dgrad', flops,
                      2.0 * (B * H * H * (cin + cout + (cin if mask_ref is not None else 0)) + 9 * cin * cout), name='vgg16/' + name,
                      desc=dd)

        def unpool(src_name, dy, relu_mask):
            y, H = acts[src_name]
            c = y.shape[-1]
            self._add(self.prog_bwd, lambda: ops.maxpool2_bwd(y[B:], dy, dbuf[src_name], B, H, H, c, relu_mask), 'maxpool_bwd',
                      0.0, B * H * H * c * 5.0)

        def unpool_tap(name, dy):
            """unpool(name, dy, 0) + tap(name, True) in one pass (the tapped layer is the pooled one)."""
            if l1:
                unpool(name, dy, 0); tap(name, True)
                return
            y, H = acts[name]
            c = y.shape[-1]
            self._add(self.prog_bwd, lambda: ops.unpool_tap_grad(dbuf[name], dy, y[B:], y[:B], B, H, c, mask, S, self.coef,
                                                                 taps[name]), 'tap_grad', 0.0, B * H * H * c * 8.5)

        # Walk the frozen network back from the deepest tapped layer (vgg16.py:343-370 reversed).  At the top of iteration
        # `name` the gradient w.r.t. its post-ReLU output is in dpool[name] (layer pooled next), or in dbuf[name] (already
        # ReLU-masked by the epilogue of the data gradient that wrote it, unless the layer is tapped: then the tap pass adds
        # the feature-loss term and applies the mask).
        for i in range(len(names) - 1, 0, -1):
            name, prev = names[i], names[i - 1]
            deepest = i == len(names) - 1
            if name in self.vgg_pool:
                if name in taps:
                    unpool_tap(name, dpool[name])
                else:
                    unpool(name, dpool[name], 1)
            elif name in taps and name not in fused_taps:
                tap(name, not deepest)
            if prev in self.vgg_pool:
                dgrad(name, dpool[prev], None)
            elif prev in taps:
                dgrad(name, dbuf[prev], None, tap_of=prev)
            else:
                dgrad(name, dbuf[prev], pred_half(acts[prev][0]))
        # first layer + 'input' feature -> gradient of the renderer's last convolution
        last = self.ren[-1]
        self.d_pred = self._act(B, S, S, last.lddy)
        in_idx = taps.get('input', -1)
        if names:
            self._add(self.prog_bwd, lambda: ops.vgg_conv1_1_bwd(dbuf['conv1_1'], B, S, self.w11, self.in_future, self.pred,
                                                                 self.ldp, mask, self.coef, self.d_pred, last.lddy, in_idx, l1),
                      'vgg_conv1_1_bwd', 2.0 * B * S * S * 9 * 64)
        else:       # image-space loss only: reconstruction_loss 'l2' (imm_model.py:385-387) or perceptual.comp == ['input']
            self._add(self.prog_bwd, lambda: ops.image_loss_grad(self.in_future, self.pred, self.ldp, B, S, mask, self.coef, in_idx,
                                                                 self.d_pred, last.lddy, l1), 'image_loss_grad')

        # ---- renderer backward -----------------------------------------------------------------------------
        ups = {idx: (ub, H, co) for idx, ub, H, co in self.ren_up}
        d_out, ldd = self.d_pred, last.lddy
        up_src = None
        for i in range(len(self.ren) - 1, -1, -1):
            lay = self.ren[i]
            if i == 0:
                dx, lddx = self.d_joint, Cj
            else:
                prev = self.ren[i - 1]
                if (i - 1) in ups:     # this conv's input is the upsampled output of conv i-1
                    _ub, Hp, cp = ups[i - 1]
                    dx, lddx = self._act(B, 2 * Hp, 2 * Hp, cp), cp
                else:
                    dx, lddx = self._act(B, prev.Ho, prev.Wo, prev.co), prev.co
            self._conv_backward(lay, d_out, ldd, dx, lddx, up_src=up_src)
            up_src = None
            if i > 0:
                if (i - 1) in ups:
                    _ub, Hp, cp = ups[i - 1]
                    d_prev = self._act(B, Hp, Hp, cp)
                    if prev.bn:
                        # the adjoint of the up-sampling rides in conv i-1's batch-norm backward reduction (next iteration)
                        up_src = (dx, lddx)
                    else:
                        self._add(self.prog_bwd, (lambda dx=dx, d_prev=d_prev, Hp=Hp, cp=cp:
                                                  ops.upsample2x_bwd(dx, d_prev, B, Hp, Hp, cp, cp, cp)), 'upsample_bwd',
                                  0.0, B * Hp * Hp * cp * 10.0)
                    d_out, ldd = d_prev, cp
                else:
                    d_out, ldd = dx, lddx

        # By default every layer's filter gradient waits for ONE set of grouped launches at the end of the backward pass (more
        # members per launch, fewer splits).  IMM_DP_BUCKETS=2 (two overlapped all-reduce buckets, imm_amd/train/cnn_train_multi.py):
        # the renderer's are issued and reduced here, so that this bucket (the tail of the flat gradient buffer) can travel while
        # the encoders' backward is still running.
        self.n_bwd_bucket0 = None
        # (The renderer's grouped launches as a third lane beside the encoders' backward chains: 3.37 -> 3.48 ms — the matrix-heavy
        # workgroups take the CUs the latency-bound BN / data-gradient chains need; round 3, same box.)
        if self.dp_buckets >= 2:
            for scope, fn in self._colsum_pending:       # the renderer head's bias gradient belongs to this bucket
                self._add(self.prog_bwd, fn, 'colsum', name=scope)
            self._colsum_pending = []
            self._flush_wgrads('renderer')
            self.reduce_tab_ren = ops.reduce_table([j for j, _n in self._reduce_jobs], [n for _j, n in self._reduce_jobs], self.dev)
            self._add(self.prog_bwd, lambda: ops.wgrad_reduce_multi(self.reduce_tab_ren), 'wgrad_reduce', name='renderer')
            self._reduce_jobs = []
            self.n_bwd_bucket0 = len(self.prog_bwd)
        # Round 6, built and measured, off by default (IMM_WG_CUS=<CUs> turns it on): the renderer's filter
        # gradients (46 % of the trainable filter-gradient FLOPs, ready here) as a CONFINED launch group on lane 2 — splits sized
        # for `wg_cus` compute units — beside the encoders' backward chains.  Same box: off 3.193 / 3.206 ms | 96 CUs 3.172 | 128
        # CUs 3.257 | 64 CUs 3.373.  At 96 CUs the final filter-gradient phase shrinks by 113 us and the encoders' backward grows
        # by 110 (profiles/r06_lanes_timeline_split.txt): zero-sum for the reason given at IMM_VGG_SPLIT above.
        self.wg_cus = int(os.environ.get('IMM_WG_CUS', '0')) if self.dp_buckets < 2 else 0
        if self.wg_cus:
            self._mark(self.prog_bwd, 'fork', lane=2)
            self._cur_lane = 2
            self._flush_wgrads('renderer', cus=self.wg_cus)
            self.reduce_tab_ren = ops.reduce_table([j for j, _n in self._reduce_jobs], [n for _j, n in self._reduce_jobs], self.dev)
            self._add(self.prog_bwd, lambda: ops.wgrad_reduce_multi(self.reduce_tab_ren), 'wgrad_reduce', name='renderer')
            self._reduce_jobs = []
            self._cur_lane = 0
        self.bucket0_offset = self.tab.offsets[[n for n, _s, _w in self.spec].index('model/renderer/conv_1/w')]
        # ---- bottleneck + pose encoder backward (main stream) || image encoder backward (side stream) --------------
        self._mark(self.prog_bwd, 'fork')       # d_joint is complete here
        nf8 = 8 * self.cfg.n_filters
        He = self.He
        ph = self.pose_head
        self.d_heat = self._act(B, He, He, ph.lddy)
        dg = self.d_joint[..., nf8:]
        d_feat = self._act(B, He, He, nf8)
        self._colsum_needs_dheat = False
        if self.fused_head:
            # bottleneck backward + the head's data gradient + its bias-gradient partial rows in one launch; the head's filter
            # gradient joins the multi-problem launch like every other layer's, the bias rows the final slab reduction
            self.head_bias_partial = self._zeros(B, K)
            flops_h = 2.0 * B * He * He * nf8 * K
            self._add(self.prog_bwd, lambda: ops.pose_head_bwd(dg, Cj, B, He, He, K, self.inv_std, 16, self.mu, self.py, self.px,
                                                               self.d_heat, ph.lddy, ph.wt_d, nf8, d_feat, nf8,
                                                               self.head_bias_partial, self.cfg.gauss_mode), 'bottleneck_bwd', flops_h)
            self._wgrad_pending.append((ph, self.d_heat, ph.lddy, flops_h))
            gb_h = self.gview['model/pose_encoder/conv_1/b']
            self._reduce_jobs.append(((self.head_bias_partial.data_ptr(), gb_h.data_ptr(), B, 1, 1, 1, K, 1), K))
        else:
            self._add(self.prog_bwd, lambda: ops.softargmax_gauss_bwd(dg, Cj, B, He, He, K, self.inv_std, 16, self.mu, self.py,
                                                                      self.px, self.d_heat, ph.lddy, self.cfg.gauss_mode), 'bottleneck_bwd')
            if self.two_streams:
                self._signal(self.prog_bwd, 'd_heat', lane=0)          # for the pose head's deferred bias gradient on lane 1
                self._colsum_needs_dheat = True
            self._conv_backward(ph, self.d_heat, ph.lddy, d_feat, nf8)
        self._encoder_backward(self.enc_pose, d_feat, nf8)

        # ---- image encoder backward ----------------------------------------------------------------------------
        self._cur_lane = 1
        self._cur_scope = 'loss'
        self._add(self.prog_bwd, lambda: ops.cost_ema(self.loss_out[3 * self.nfeat:], self.cost_ema, 0.99), 'cost_ema')
        if He == 16:
            self._encoder_backward(self.enc_im, self.d_joint, Cj)
        else:
            e = self.enc_im[-1]
            d_e = self._act(B, He, He, nf8)
            self._add(self.prog_bwd, lambda: ops.resize_ac_bwd(self.d_joint, d_e, B, He, He, 16, 16, nf8, Cj, nf8), 'resize_ac_bwd')
            self._encoder_backward(self.enc_im, d_e, nf8)
        # deferred bias gradients, at the tail of the (shorter) image-encoder lane
        if self._colsum_needs_dheat:
            self._wait(self.prog_bwd, 'd_heat', lane=1)
        for scope, fn in self._colsum_pending:
            self._add(self.prog_bwd, fn, 'colsum', name=scope)
        self._colsum_pending = []
        self._cur_lane = 0
        self._mark(self.prog_bwd, 'join')
        if self.wg_cus:
            self._mark(self.prog_bwd, 'join', lane=2)
        self._flush_wgrads('encoders' if (self.n_bwd_bucket0 is not None or self.wg_cus) else 'all layers')
        # one table-driven launch sums every layer's split-K slabs into the flat gradient buffer
        if self._reduce_jobs:
            self.reduce_tab = ops.reduce_table([j for j, _n in self._reduce_jobs], [n for _j, n in self._reduce_jobs], self.dev)
            self._add(self.prog_bwd, lambda: ops.wgrad_reduce_multi(self.reduce_tab), 'wgrad_reduce',
                      name='encoders' if (self.n_bwd_bucket0 is not None or self.wg_cus) else 'all layers')

    def _encoder_backward(self, layers, d_out, ldd):
        B = self.B
        for i in range(len(layers) - 1, -1, -1):
            lay = layers[i]
            if i > 0:
                prev = layers[i - 1]
                dx, lddx = self._act(B, prev.Ho, prev.Wo, prev.co), prev.co
            else:
                dx, lddx = None, 0
            self._conv_backward(lay, d_out, ldd, dx, lddx)
            d_out, ldd = dx, lddx

    # ------------------------------------------------------------------------------------------
    # execution
    # ------------------------------------------------------------------------------------------
    def run(self, prog):
        """Issue a launch program.  Lane 0 = the caller's current stream; lanes 1, 2 = side streams (encoder branch,
        filter gradients) tied to it with events, so the same code serves eager execution and HIP-graph capture
        (a side stream joins the capture when it waits on an event recorded in the capturing stream)."""
        if not self.two_streams:
            for l in prog:
                if l.fn is not None:
                    self._issue(l)
            return
        main = torch.cuda.current_stream(self.dev)
        streams = {0: main}
        events = {}
        joined = {0}        # lanes that have (transitively) received work ordered after the main stream

        def lane_stream(i):
            if i not in streams:
                streams[i] = self._side_stream(i)
            return streams[i]
        # (TrainStep runs the backward program in two slices when the gradient exchange has two buckets)
        pname = ('fwd' if prog is self.prog_fwd else 'opt' if (prog is self.prog_opt or not prog) else
                 'bwd' if (prog is self.prog_bwd or prog[0] is self.prog_bwd[0]) else
                 'bwd2' if prog[-1] is self.prog_bwd[-1] else 'opt')
        if self._stamp_mode and prog is self.prog_fwd:
            self._stamp_names = []

        def stamp(lane, label):
            if not self._stamp_mode or len(self._stamp_names) >= self._stamp_buf.numel():
                return
            with torch.cuda.stream(lane_stream(lane)):
                ops.debug_stamp(self._stamp_buf, len(self._stamp_names))
            self._stamp_names.append((lane, '%s:%s' % (pname, label)))
        stamp(0, 'start')
        for l in prog:
            if l.fn is not None and l.tag in self._skip_tags:
                continue
            if l.fn is not None:
                if l.lane == 0:
                    self._issue(l)
                else:
                    with torch.cuda.stream(lane_stream(l.lane)):
                        self._issue(l)
                if self._stamp_mode == 'all':
                    stamp(l.lane, '%s %s' % (l.tag, l.name))
            elif l.tag == 'fork':
                stamp(0, 'fork' if l.lane == 1 else 'fork%d' % l.lane)
                ev = torch.cuda.Event(); ev.record(main); lane_stream(l.lane).wait_event(ev)
                stamp(l.lane, 'fork: lane %d released' % l.lane)
            elif l.tag == 'join':
                stamp(l.lane, 'join: lane %d done' % l.lane)
                stamp(0, 'join: lane 0 arrives' if l.lane == 1 else 'join%d: lane 0 arrives' % l.lane)
                ev = torch.cuda.Event(); ev.record(lane_stream(l.lane)); main.wait_event(ev)
                stamp(0, 'join: lane 0 resumes' if l.lane == 1 else 'join%d: lane 0 resumes' % l.lane)
            elif l.tag.startswith('record:'):
                stamp(l.lane, l.tag)
                ev = torch.cuda.Event(); ev.record(lane_stream(l.lane)); events[l.tag[7:]] = ev
            elif l.tag.startswith('wait:'):
                stamp(l.lane, l.tag + ' arrives')
                lane_stream(l.lane).wait_event(events[l.tag[5:]])
                stamp(l.lane, l.tag + ' resumes')
        stamp(0, 'end')

    @staticmethod
    def _issue(l):
        """One launch, under its CU limit (imm_set_cu_limit is thread-local host state read by the launch code: the confined
        grid is baked into the launch, also when it is captured into a graph)."""
        if l.cus:
            ops.set_cu_limit(l.cus)
            try:
                l.fn()
            finally:
                ops.set_cu_limit(0)
        else:
            l.fn()

    def stamp_report(self):
        """[(microseconds since the first probe of the step, lane, label)] of the last step issued with IMM_DEBUG_STAMPS set."""
        torch.cuda.synchronize(self.dev)
        t = self._stamp_buf[:len(self._stamp_names)].cpu().tolist()
        return [((v - t[0]) / 100.0, lane, label) for v, (lane, label) in zip(t, self._stamp_names)]

    def _side_stream(self, i=1):
        if self._side is None:
            self._side = {}
        if i not in self._side:
            self._side[i] = torch.cuda.Stream(device=self.dev)
        return self._side[i]

    def run_timed(self, prog):
        """Eager run with a HIP event pair around every launch (events on the launch stream).
        Returns [(tag, ms, flops, bytes)]."""
        evs = []
        for l in prog:
            if l.fn is None:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._issue(l)
            e1.record()
            evs.append((l, e0, e1))
        torch.cuda.synchronize()
        self.last_variants = [l.variant for l, _e0, _e1 in evs]
        return [(l.tag, e0.elapsed_time(e1), l.flops, l.bytes, l.name) for l, e0, e1 in evs]

    def set_inputs(self, image, future_image, mask=None):
        # device tensors (the loader's, the bench's): stream-ordered copies; host tensors: persistent double-buffered pinned
        # staging, asynchronous, no read-back and no stream synchronisation (ops.PinnedStager; round 5 verified every step's inputs
        # like a one-time upload and serialised a host-fed loop with the device)
        if getattr(self, '_stager', None) is None:
            self._stager = ops.PinnedStager()
        self._stager.copy(self.in_image, image, 'image')
        self._stager.copy(self.in_future, future_image, 'future_image')
        if self.use_mask:
            if mask is None:
                raise RuntimeError('No loss mask recieved but is required.')
            self._stager.copy(self.in_mask, mask, 'mask')

    def forward(self, training=True):
        self._training = bool(training)
        self.run(self.prog_fwd)

    def forward_model_only(self, training=False):
        """IMMModel.build(build_loss=False): encoders, bottleneck, renderer (no VGG, no loss)."""
        self._training = bool(training)
        self.run([l for l in self.prog_fwd[:self.n_fwd_model] if l.group != 'gt'])   # the VGG ground-truth lane: loss only

    def backward(self):
        self.run(self.prog_bwd)

    def snapshot(self):
        """Everything a step mutates (used to warm kernels up before graph capture without side effects).  The clones run on torch's
        CURRENT stream: take the snapshot on the stream the steps run on (TrainStep._capture does), or synchronise before the next
        step is launched on another one — a snapshot taken on the default stream and restored on the step's stream is a race that
        only shows when the device is shared (tools/det_graph.py had it, round 5)."""
        return {'params': self.params.clone(), 'm': self.adam_m.clone(), 'v': self.adam_v.clone(),
                'step': self.step_count.clone(), 'adam_t': self.adam_t.clone(), 'agg': self.loss_agg.clone(), 'grads': self.grads.clone(), 'cost_ema': self.cost_ema.clone(),
                'state': {k: v.clone() for k, v in self.state.items()},
                'loss_scale': None if self.loss_scale_state is None else self.loss_scale_state.clone()}

    def restore(self, snap):
        self.params.copy_(snap['params']); self.adam_m.copy_(snap['m']); self.adam_v.copy_(snap['v'])
        self.step_count.copy_(snap['step']); self.adam_t.copy_(snap['adam_t']); self.loss_agg.copy_(snap['agg']); self.grads.copy_(snap['grads']); self.cost_ema.copy_(snap['cost_ema'])
        for k, v in snap['state'].items():
            self.state[k].copy_(v)
        if self.loss_scale_state is not None:
            self.loss_scale_state.copy_(snap['loss_scale'])
        self.run(self.prog_pack)

    def optimizer_step(self):
        self.run(self.prog_opt)

    # convenience views -----------------------------------------------------------------------------
    @property
    def loss(self):
        return self.loss_out[3 * self.nfeat + 2]

    @property
    def loss_terms(self):
        return self.loss_out[:self.nfeat]

    @property
    def loss_scale(self):
        """Current loss scale S (1.0 when loss scaling is off): `grads` / `gview` hold S x the gradient between backward()
        and optimizer_step().  Reads the device scalar (synchronises)."""
        return 1.0 if self.loss_scale_state is None else float(self.loss_scale_state[0])

    def cost_summaries(self):
        """The reference's cost summaries (base_model.py:52-60, family 'train'): `<name>_raw` = this step's value, `<name>_avg` =
        the moving-average shadow (decay 0.99, started at 0, NOT zero-debiased: tensorflow 1.10's ExponentialMovingAverage default)
        over the training steps so far.  Reads device scalars (synchronises)."""
        raw = [float(v) for v in self.loss_out[3 * self.nfeat:3 * self.nfeat + 3]]
        ema = [float(v) for v in self.cost_ema]
        out = {}
        for i, name in enumerate(('reconstruction_loss', 'weights_loss', 'loss_total')):
            out[name + '_raw'] = raw[i]
            out[name + '_avg'] = ema[i]
        return out

    def vgg_activations(self):
        """{layer: (activation [2B, H, H, C], H)} of the last forward pass, every entry complete.  With the fused head
        (imm_vgg_head_fwd) the step only stores the prediction half of conv1_1 — its ground-truth half has no reader in the
        step — so it is computed here, on demand (summaries, diagnostics, tests), by the stand-alone first-layer kernel."""
        if getattr(self, 'vgg_head_fused', False) and 'conv1_1' in self.vgg_act:
            a = self.vgg_act['conv1_1'][0]
            ops.vgg_conv1_1_fwd(self.in_future, self.pred, self.ldp, self.B, self.S, self.w11, self.b11, a, 1)
        return self.vgg_act

    def vgg_activation_rms(self):
        """selfsup/vgg16.py:232-234: {'activation/<layer>': sqrt(mean(z^2))} of every VGG16 layer output of the last forward pass
        (both halves of the concat([gt, pred]) batch, like the reference's graph).  Summary steps only: one small reduction per layer."""
        if not getattr(self, 'vgg_act', None):
            return {}
        if getattr(self, '_rms_scratch', None) is None:
            self._rms_scratch = (self._zeros(1024), self._zeros(1))
        part, out = self._rms_scratch
        res = {}
        for name, (y, _h) in self.vgg_activations().items():
            ops.rms16(y, part, out)
            res['activation/' + name] = float(out)
        return res

    def named_gradients(self):
        """{tf variable name: gradient of the loss (weight decay excluded), loss scale divided out} after backward()."""
        inv = 1.0 / self.loss_scale
        return OrderedDict((k, v.detach() * inv) for k, v in self.gview.items())

    @property
    def future_im_pred(self):
        return self.pred[..., :3]

    def memory_bytes(self):
        return self._alloc_bytes

    def step_flops(self):
        """Algorithmic conv FLOPs of one training step (2 FLOP/MAC), BASELINE.md §4 counting rule."""
        return sum(l.flops for p in (self.prog_fwd, self.prog_bwd) for l in p if l.fn is not None)
