"""Frozen perceptual-network weights: the reference's colourisation VGG16 (`vgg16.caffemodel.h5`, README.md:68) ->
the `vgg16/<layer>/{weights,biases}` dictionary IMMEngine(vgg_weights=...) / load_parameters() take.

Reference behaviour restated (imm/models/selfsup/vgg16.py:17-47 conv kernels, :74-92 biases; build_vgg16.py:16-31 calls
them with pre_adjust_batch_norm=True, batch_norm=False): a caffe conv blob `data[name]['0']` is [out, in, kh, kw] and is
transposed to HWIO; a 3-channel conv1_1 is BGR->RGB flipped (the shipped file's conv1_1 has ONE input channel:
grayscale); the caffe BatchNorm blobs `data['batch_'+name]` = (mean sum '0', variance sum '1', scale factor '2') are
folded into the convolution: sigma = sqrt(1e-5 + var/scale), mu = mean/scale, W /= sigma (per output channel),
b = (b - mu) / sigma.

File formats: `.npz` (either the final keys `vgg16/conv1_1/weights` ... or caffe-style `conv1_1/0`, `batch_conv1_1/2`
...) and the original `.h5` (Caffe HDF5 snapshot, `/data/<layer>/<blob index>`) through the pure-Python reader
imm_amd/utils/hdf5_lite.py (no h5py / libhdf5 needed)."""
import numpy as np

VGG_CONV_LAYERS = ('conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3',
                   'conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3')
BN_EPS = 1e-5


def fold_batch_norm(w_hwio, bias, bn_mean_sum, bn_var_sum, bn_scale):
    """Fold a caffe BatchNorm (accumulated mean / variance sums and their scale factor) that FOLLOWS the convolution
    into the convolution's kernel and bias.  Returns (w, b) float32."""
    scale = np.asarray(bn_scale, dtype=np.float64).reshape(-1)[0]
    sigma = np.sqrt(BN_EPS + np.asarray(bn_var_sum, np.float64) / scale)
    mu = np.asarray(bn_mean_sum, np.float64) / scale
    w = np.asarray(w_hwio, np.float64) / sigma                       # broadcast over the output-channel axis (last)
    b = (np.asarray(bias, np.float64) - mu) / sigma
    return w.astype(np.float32), b.astype(np.float32)


def from_caffe_blobs(data, layers=VGG_CONV_LAYERS, fold_bn=True):
    """data: mapping layer -> {'0': W [out,in,kh,kw], '1': bias} and optionally 'batch_'+layer -> {'0','1','2'}.
    Returns {'vgg16/<layer>/weights' (HWIO), 'vgg16/<layer>/biases'} for the layers present."""
    out = {}
    for name in layers:
        if name not in data:
            continue
        blob = data[name]
        w = np.array(blob['0'], dtype=np.float32).transpose(2, 3, 1, 0)            # -> [kh, kw, in, out]
        if name == 'conv1_1' and w.shape[2] == 3:
            w = w[:, :, ::-1]                                                        # caffe BGR -> RGB
        b = np.array(blob['1'], dtype=np.float32) if '1' in blob else np.zeros(w.shape[3], np.float32)
        bn = data.get('batch_' + name) if fold_bn else None
        if bn is not None:
            w, b = fold_batch_norm(w, b, bn['0'], bn['1'], bn['2'])
        out['vgg16/%s/weights' % name] = np.ascontiguousarray(w)
        out['vgg16/%s/biases' % name] = b
    return out


def _nest(flat):
    data = {}
    for k in flat:
        parts = k.strip('/').split('/')
        if parts[0] == 'data':
            parts = parts[1:]
        if len(parts) == 2:
            data.setdefault(parts[0], {})[parts[1]] = np.asarray(flat[k])
    return data


def load_vgg16(path, fold_bn=True):
    """Returns the weight dictionary for IMMEngine(vgg_weights=...) as torch tensors."""
    import torch
    if path.endswith('.npz'):
        flat = np.load(path)
        if any(k.startswith('vgg16/') for k in flat.files):
            arrs = {k: flat[k] for k in flat.files if k.startswith('vgg16/')}
        else:
            arrs = from_caffe_blobs(_nest({k: flat[k] for k in flat.files}), fold_bn=fold_bn)
    elif path.endswith('.h5') or path.endswith('.hdf5'):
        # pure-Python reader (imm_amd/utils/hdf5_lite.py): the Caffe snapshot layout /data/<layer>/<blob index>, what the
        # reference gets from deepdish.io.load (vgg16.py:74-92)
        from .hdf5_lite import H5File
        tree = H5File(path).load('/')
        root = tree['data'] if 'data' in tree else tree
        data = {g: {k: np.asarray(v) for k, v in blobs.items()} for g, blobs in root.items() if isinstance(blobs, dict)}
        arrs = from_caffe_blobs(data, fold_bn=fold_bn)
    else:
        raise ValueError('unknown VGG16 weight file type: %s' % path)
    need = [n for n in VGG_CONV_LAYERS[:12] if 'vgg16/%s/weights' % n not in arrs]
    if need:
        raise KeyError('VGG16 weight file %s lacks layers %s' % (path, need))
    if arrs['vgg16/conv1_1/weights'].shape[2] != 1:
        raise ValueError('the IMM perceptual network is the grayscale colourisation VGG16 (conv1_1 with one input channel, '
                         'build_vgg16.py:22-26); got %d input channels' % arrs['vgg16/conv1_1/weights'].shape[2])
    return {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in arrs.items()}
