"""IMMModel — drop-in construction/config surface of /root/reference/imm/models/imm_model.py:95-490,
backed by the MI355X engine (imm_amd/engine.py -> libimm_hip.so).

Same constructor and `build` keywords and return arity as the reference:
    IMMModel(config, global_step=None, dtype=..., name='IMMModel')
    build(inputs, training_pl, costs_collection='costs', scope=None, var_device='/cpu:0',
          output_tensors=False, build_loss=True) -> (None, loss, avg_ops[, tensors])
Differences forced by leaving TF1: `inputs` holds torch tensors (NHWC float32, image values in
[0,255], mask in [0,1]), `training_pl` is a Python bool (the reference's notebook already passes
False), `dtype` is the 16-bit activation storage type (torch.bfloat16 / torch.float16), and `build`
EXECUTES the forward pass on the GPU instead of adding nodes to a graph.  Errors keep the
reference's types: ValueError for an unknown gauss mode (:75) / loss type (:389), RuntimeError when
the loss mask is required but missing (:369), AssertionError for non-square inputs (:292,432).
"""
import colorsys

import torch

from ..engine import IMMEngine
from .base_model import BaseModel


def get_n_colors(n):
    """Deterministic distinct colours for the landmark summary image (the reference draws random
    distinct colours, imm/utils/utils.py via colorize_landmark_maps imm_model.py:81-92)."""
    return [colorsys.hsv_to_rgb(i / float(max(n, 1)), 0.9, 1.0) for i in range(n)]


def colorize_landmark_maps(maps):
    """imm_model.py:81-92: [B,H,W,N] -> [B,H,W,3], max over landmarks of map*colour."""
    n = maps.shape[-1]
    colors = torch.tensor(get_n_colors(n), dtype=maps.dtype, device=maps.device)      # [N,3]
    return (maps.unsqueeze(-1) * colors.reshape(1, 1, 1, n, 3)).amax(dim=3)


class IMMModel(BaseModel):
    def __init__(self, config, global_step=None, dtype=torch.bfloat16, name='IMMModel', device=None, seed=1,
                 vgg_weights=None, hparams=None, world_size=1, dp_buckets=None):
        super(IMMModel, self).__init__(dtype, name)
        self._config = config
        self._global_step = global_step
        self._device = device
        self._seed = seed
        self.vgg_source = 'caller-supplied tensors'
        if vgg_weights is None:
            # imm_model.py:124: perceptual.net_file names the pretrained colourisation VGG16 (loaded with the caffe BN folded
            # as in selfsup/vgg16.py:17-92).  Like the reference (dd.io.load raises) a missing file is an ERROR: training
            # against a random perceptual network must be asked for explicitly with net_file: 'synthetic' (benchmarks, tests).
            import os
            perc = getattr(config, 'perceptual', None)
            net_file = getattr(perc, 'net_file', None) if perc is not None else None
            if net_file == 'synthetic':
                self.vgg_source = 'synthetic (seeded He-normal stand-in)'
            elif isinstance(net_file, str) and os.path.exists(net_file):
                from ..utils.vgg_weights import load_vgg16
                vgg_weights = load_vgg16(net_file)
                self.vgg_source = os.path.abspath(net_file)
            else:
                # raised by require_vgg() as soon as a loss is built; landmark inference (build_loss=False, scripts/test.py)
                # never touches the perceptual network, in the reference (imm_model.py:446-448) as here
                self._vgg_missing = net_file
                self.vgg_source = 'MISSING: %r' % (net_file,)
        self._vgg_weights = vgg_weights
        self._hparams = hparams
        self._world_size = world_size
        self._dp_buckets = dp_buckets
        self._engines = {}
        self.engine = None

    def require_vgg(self):
        """The loss needs the perceptual network: a missing perceptual.net_file is an error (the reference's dd.io.load
        raises), never a silent fall-back to random weights."""
        cfg = self._config
        if getattr(cfg, 'reconstruction_loss', 'perceptual') != 'perceptual':
            return       # 'l2' (imm_model.py:385-387) never touches the perceptual network
        # The perceptual branch opens net_file UNCONDITIONALLY (imm_model.py:124-127: build_vgg16(ims, pretrained_file=...) before
        # any feature is selected), so a missing file fails in the reference even for perceptual.comp == ['input'], where no VGG
        # layer is evaluated: same here.
        if hasattr(self, '_vgg_missing'):
            missing = self._vgg_missing
            raise FileNotFoundError("perceptual.net_file %r does not exist (set it to the vgg16.caffemodel.h5 / .npz file, or to "
                                    "'synthetic' to train against seeded random VGG16 weights)" % (missing,))

    # -- engine management ---------------------------------------------------------------------------
    @staticmethod
    def _mirror(dst, src):
        dst.load_parameters(src.named_parameters(), src.named_state())
        dst.adam_m.copy_(src.adam_m); dst.adam_v.copy_(src.adam_v)
        dst.step_count.copy_(src.step_count); dst.adam_t.copy_(src.adam_t)
        if dst.loss_scale_state is not None and src.loss_scale_state is not None:
            dst.loss_scale_state.copy_(src.loss_scale_state)

    def _get_engine(self, batch, size):
        """One engine (buffers + launch programs) per (batch, size); the variables are shared between instantiations
        like the reference's reuse_variables: a new engine starts from the current one, and while a TrainStep trains
        `self._master`, every other engine is refreshed from it when selected (periodic test passes with other batch
        sizes, cnn_train_multi.py:471-508)."""
        key = (int(batch), int(size))
        master = getattr(self, '_master', None)
        if key not in self._engines:
            dev = self._device or ('cuda:%d' % torch.cuda.current_device())
            eng = IMMEngine(self._config, batch, size, device=dev, act_dtype=self.dtype, seed=self._seed,
                            vgg_weights=self._vgg_weights, hparams=self._hparams, world_size=self._world_size,
                            dp_buckets=self._dp_buckets)
            src = master if master is not None else self.engine
            if src is not None:
                self._mirror(eng, src)
            self._engines[key] = eng
        elif master is not None and self._engines[key] is not master:
            self._mirror(self._engines[key], master)
        self.engine = self._engines[key]
        return self.engine

    # -- reference surface -----------------------------------------------------------------------------
    def build(self, inputs, training_pl, costs_collection='costs', scope=None, var_device='/cpu:0',
              output_tensors=False, build_loss=True):
        im, future_im = inputs['image'], inputs['future_image']
        future_im_size = list(future_im.shape[1:3])
        assert future_im_size[0] == future_im_size[1]
        assert list(im.shape) == list(future_im.shape)
        mask = inputs.get('mask')
        cfg = self._config
        if build_loss and cfg.loss_mask and mask is None:
            raise RuntimeError('No loss mask recieved but is required.')
        if build_loss:
            self.require_vgg()
        eng = self._get_engine(im.shape[0], future_im_size[0])
        if mask is None and eng.use_mask:
            mask = torch.ones(im.shape[0], future_im_size[0], future_im_size[0], 1)
        # host tensors go to set_inputs as they are (f32): it stages them through its persistent pinned buffers (ops.PinnedStager)
        # — never a raw `.to(device)` of a pageable tensor
        def f32(t):
            return t.to(eng.dev, torch.float32) if t.is_cuda else t.to(torch.float32)
        eng.set_inputs(f32(im), f32(future_im), None if mask is None else f32(mask))
        training = bool(training_pl)
        if build_loss:
            eng.forward(training)
            loss = eng.loss
        else:
            eng.forward_model_only(training)
            loss = None
        if not output_tensors:
            return None, loss, self._avg_ops
        tensors = {}
        tensors.update(inputs)
        size = future_im_size[0]
        full_maps = torch.empty(eng.B, size, size, eng.K, device=eng.dev)
        from .. import ops
        ops.gauss_render_f32(eng.mu, eng.B, eng.K, eng.inv_std, size, full_maps, eng.cfg.gauss_mode)
        tensors.update({'future_im': future_im, 'im': im,
                        'pose_embedding': colorize_landmark_maps(full_maps),
                        'future_im_pred': eng.future_im_pred,
                        'gauss_yx': eng.mu,
                        # collection 'tensors' extras (imm_model.py:250,266-267)
                        'heatmaps': eng.heat[..., :eng.K], 'gauss_y_prob': eng.py, 'gauss_x_prob': eng.px})
        return None, loss, self._avg_ops, tensors
