"""ctypes binding of libimm_hip.so (the C-ABI declared in include/imm_hip.h).

There is NO fallback: if the library is missing or a call fails this raises.  The product path never
routes through the CPU oracle.
"""
import ctypes as C
import os

# torch ships its own libamdhip64; it MUST be in the process before libimm_hip.so is dlopen'ed so that
# both share one HIP runtime (device context, streams).  Loading ours first binds /opt/rocm's copy and
# the two runtimes do not see each other's devices.
import torch  # noqa: F401  (side effect: loads the ROCm runtime libraries bundled with PyTorch)

_HERE = os.path.dirname(os.path.abspath(__file__))
# IMM_HIP_LIB: another build of the same ABI (A/B timing of kernel changes on one box); default = the in-tree library
LIB_PATH = os.environ.get('IMM_HIP_LIB') or os.path.join(_HERE, 'libimm_hip.so')

IMM_BF16, IMM_F16, IMM_F32 = 0, 1, 2      # IMM_F32: f32 storage, the exact-arithmetic witness engine (test instrument)
CONV_BIAS, CONV_RELU, CONV_STATS, CONV_MASK, CONV_OUT_F32 = 1, 2, 4, 8, 16
SSE_BLOCKS = 512
OPTIMIZERS = {'adam': 0, 'adadelta': 1, 'adagrad': 2}      # IMM_OPT_* (scripts/train.py:97-104)
GAUSS_MODES = {'rot': 0, 'flat': 1, 'ankush': 2}     # IMM_GAUSS_* (config key gauss_mode, imm_model.py:48-72)
ABI_VERSION = 21


class ImmHipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'batch', 'hi', 'wi', 'ci', 'ldx', 'ho', 'wo', 'co', 'ldy', 'kh', 'kw', 'stride', 'pad_t', 'pad_l',
        'updiv', 'kpad', 'flags', 'ldmask', 'out_scale', 'out_off_y', 'out_off_x')]


class WgradJob(C.Structure):
    _fields_ = [('desc', ConvDesc), ('x', C.c_void_p), ('dy', C.c_void_p), ('slab', C.c_void_p), ('lddy', C.c_int32),
                ('nsplit', C.c_int32), ('x_scale', C.c_void_p), ('x_shift', C.c_void_p), ('x_relu', C.c_int32),
                ('reserved', C.c_int32)]


class OptHParams(C.Structure):
    _fields_ = [('lr_start', C.c_float), ('lr_decay', C.c_float), ('lr_step', C.c_int32), ('lr_multiple', C.c_float),
                ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float), ('clip', C.c_float),
                ('grad_scale', C.c_float), ('optim', C.c_int32),
                ('scale_growth_interval', C.c_int32), ('scale_max', C.c_float)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    'imm_abi_version': [],
    'imm_device_info': [_P],
    'imm_set_cu_limit': [_I],
    'imm_get_cu_limit': [],
    'imm_graph_begin': [_P],
    'imm_graph_end': [_P, C.POINTER(C.c_void_p)],
    'imm_graph_launch': [_P, _P],
    'imm_graph_destroy': [_P],
    'imm_pack_weights': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'imm_pack_weights_multi': [_P, _P, _I, _I, _I, _P],
    'imm_pack_weights_multi_blocks': [_I, _I, _I],
    'imm_wgrad_reduce_multi': [_P, _P, _I, _I, _P],
    'imm_conv2d': [C.POINTER(ConvDesc), _I, _P, _P, _P, _P, _P, _P, _P],
    'imm_conv_stats_blocks': [C.POINTER(ConvDesc)],
    'imm_conv2d_variant': [C.POINTER(ConvDesc), _I],
    'imm_conv2d_tap_supported': [C.POINTER(ConvDesc)],
    'imm_conv2d_tap': [C.POINTER(ConvDesc), _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _P],
    'imm_conv2d_wgrad': [C.POINTER(ConvDesc), _I, _P, _P, _I, _P, _I, _P],
    'imm_conv2d_wgrad_splits': [C.POINTER(ConvDesc), _I],
    'imm_conv2d_wgrad_variant': [C.POINTER(ConvDesc), _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    'imm_conv2d_wgrad_multi_plan': [_P, _I, _I, _P],
    'imm_conv2d_wgrad_multi': [_P, _P, _P],
    'imm_conv2d_wgrad_reduce': [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    'imm_colsum': [_P, _I, _L, _I, _I, _I, _P, _P, _P],
    'imm_colsum_blocks': [_L, _I],
    'imm_bn_finalize': [_P, _I, _I, _L, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    'imm_bn_apply_relu': [_P, _I, _L, _I, _I, _P, _P, _I, _P, _I, _P],
    'imm_bn_apply_fused': [_P, _I, _I, _L, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _I, _I, _I, _P],
    'imm_bn_bwd_apply_fused': [_P, _I, _I, _L, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P],
    'imm_rows_reduce': [_P, _I, _I, _I, _P, _P],
    'imm_bn_bwd_reduce': [_P, _I, _P, _I, _I, _L, _I, _P, _P, _P, _P, _I, _P, _P],
    'imm_bn_bwd_blocks': [_L, _I],
    'imm_bn_bwd_reduce_up': [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    'imm_bn_bwd_finalize': [_P, _I, _I, _I, _L, _P, _P, _P, _I, _P, _P, _P, _P],
    'imm_bn_bwd_apply': [_P, _I, _P, _I, _I, _L, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P],
    'imm_upsample2x_fwd': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'imm_upsample2x_bwd': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'imm_resize_ac_fwd': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'imm_resize_ac_bwd': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'imm_maxpool2_fwd': [_P, _P, _I, _I, _I, _I, _I, _P],
    'imm_maxpool2_bwd': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'imm_pack_image': [_P, _P, _I, _L, _P],
    'imm_pack_image_taps': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'imm_softargmax_gauss_fwd': [_P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _I, _I, _I, _P],
    'imm_softargmax_gauss_bwd': [_P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _I, _I, _P],
    'imm_gauss_render_f32': [_P, _I, _I, _F, _I, _P, _I, _P],
    'imm_pose_head_fwd': [_P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    'imm_pose_head_bwd': [_P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _I, _I, _P, _I, _I, _P, _I, _P, _P],
    'imm_vgg_conv1_1_fwd': [_P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P],
    'imm_vgg_head_supported': [_I, _I, _I],
    'imm_vgg_head_fwd': [_P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P],
    'imm_vgg_conv1_1_bwd': [_P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _I, _P, _I, _P],
    'imm_image_loss_grad': [_P, _P, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _P],
    'imm_copy_f32': [_P, _P, _L, _P],
    'imm_debug_stamp': [_P, _I, _P],
    'imm_tps_warp': [_P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P],
    'imm_tps_warp_pad': [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P],
    'imm_conv2d_group': [_P, _I, _I, _P, _P, _P, _P, _P, _P],
    'imm_conv_first_supported': [_I, _I, _I, _I],
    'imm_conv_first_stats_blocks': [_I, _I],
    'imm_conv_first': [_P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    'imm_conv2d_nol_supported': [C.POINTER(ConvDesc)],
    'imm_conv2d_nol_stats_blocks': [C.POINTER(ConvDesc)],
    'imm_conv2d_nol': [C.POINTER(ConvDesc), _I, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    'imm_conv2d_dgrad_s2_supported': [_I, _I, _I, _I, _I, _I],
    'imm_conv2d_dgrad_s2': [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    'imm_conv2d_group_stats_blocks': [_P, _I],
    'imm_upsample2x_bwd_bn': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    'imm_upsample2x_bwd_bn_blocks': [_I, _I, _I, _I],
    'imm_crc32c': [_P, C.c_uint64, C.POINTER(C.c_uint32)],
    'imm_rccl_unique_id': [_P],
    'imm_rccl_init': [_I, _I, _P, C.POINTER(C.c_void_p)],
    'imm_rccl_allreduce': [_P, _P, _L, _P],
    'imm_rccl_destroy': [_P],
    'imm_resize_crop_u8': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'imm_unpool_tap_grad': [_P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    'imm_masked_sse_pool': [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P],
    'imm_masked_sse_multi': [_I, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _P],
    'imm_masked_sse_all': [_I, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P, _P],
    'imm_masked_sse': [_P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    'imm_masked_sse_f32': [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P],
    'imm_perceptual_finalize': [_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P],
    'imm_cost_ema': [_P, _P, _F, _P],
    'imm_rms16': [_P, _L, _I, _P, _I, _P, _P],
    'imm_tap_grad': [_P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P, _I, _I, _I, _P],
    'imm_weight_decay_loss': [_P, _P, _P, _P, _I, _P, _P, _P, _P],
    'imm_clip_adam_step': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, C.POINTER(OptHParams), _P, _P],
}

# byte-size twins of the row-count queries (int64 result; < 0 = unsupported)
_SIGS64 = {
    'imm_conv2d_workspace_bytes': [C.POINTER(ConvDesc)],
    'imm_conv2d_group_workspace_bytes': [_P, _I],
    'imm_conv2d_wgrad_workspace_bytes': [C.POINTER(ConvDesc), _I, _I],
    'imm_conv2d_wgrad_multi_table_bytes': [_I],
    'imm_colsum_workspace_bytes': [_L, _I],
    'imm_bn_bwd_workspace_bytes': [_L, _I],
    'imm_upsample2x_bwd_bn_workspace_bytes': [_I, _I, _I, _I],
    'imm_masked_sse_workspace_bytes': [_I],
    'imm_vgg_head_scratch_bytes': [_I, _I],
}

_lib = None


def declared_symbols():
    return sorted(list(_SIGS) + list(_SIGS64) + ['imm_last_error', 'imm_source_digest'])


def load():
    """Load (once) and return the ctypes handle; raises ImmHipError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImmHipError('libimm_hip.so not found at %s: build it with `python imm_amd/build.py` '
                          '(there is no CPU fallback for the product path)' % LIB_PATH)
    if 'IMM_HIP_LIB' not in os.environ and os.path.isdir(os.path.join(_HERE, 'csrc')):
        # source checkout: the binary must have been built from THESE sources (a stale .so next to edited kernels is the
        # classic silent failure); rebuild when a compiler is at hand, otherwise refuse
        from . import build as _b
        want, have = _b.source_digest(), _b.library_digest(LIB_PATH)
        if want != have:
            if not _b.have_compiler():
                raise ImmHipError('libimm_hip.so is stale (built from sources %s.., checkout is %s..) and hipcc is not '
                                  'available to rebuild it' % (have[:12] or '?', want[:12]))
            _b.build(verbose=False)
    lib = C.CDLL(LIB_PATH)
    for name in ('imm_last_error', 'imm_source_digest'):
        getattr(lib, name).restype = C.c_char_p
        getattr(lib, name).argtypes = []
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, args in _SIGS64.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int64
    v = lib.imm_abi_version()
    if v != ABI_VERSION:
        raise ImmHipError('libimm_hip.so ABI %d != expected %d' % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise ImmHipError('%s failed (%d): %s' % (what, rc, load().imm_last_error().decode()))


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise ImmHipError('%s failed (%d): %s' % (name, rc, load().imm_last_error().decode()))
