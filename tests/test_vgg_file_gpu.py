"""SURVEY §8f rank 1 at FULL size: a Caffe-layout `vgg16.caffemodel.h5` with the real shapes of the colourisation VGG16
(conv1_1 [64,1,3,3] ... conv5_3 [512,512,3,3] + every batch_<layer> group, 59 MB) is written by the REAL libhdf5
(tests/golden/make_h5_golden.c, mode `full`, compiled at test time: the file is too large to commit), read by the
pure-Python imm_amd/utils/hdf5_lite.py, folded by load_vgg16 (imm/models/selfsup/vgg16.py:17-47,74-92 restated) and
  * (CPU)  compared bit for bit with a numpy restatement of the generator + the folding formula;
  * (GPU)  fed through IMMModel(perceptual.net_file=<file>) into the engine, whose training step (loss terms, landmarks,
           one Adam update) is compared with the oracle given the SAME folded weights — the step no longer only ever runs
           on the seeded synthetic VGG.
Without gcc/libhdf5 the same blobs go through the `.npz` branch of load_vgg16 (the hdf5 part is then skipped, loudly)."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ['conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3',
         'conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3']
COUTS = [64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512]


def hash_unit(l, b, n):
    """make_h5_golden.c: hash_unit (uint32 arithmetic restated with uint64 + masks)."""
    m = np.uint64(0xffffffff)
    e = np.arange(n, dtype=np.uint64)
    h = (e * np.uint64(2654435761) + np.uint64(l * 97 + b * 13 + 12345)) & m
    h ^= h >> np.uint64(15); h = (h * np.uint64(2246822519)) & m
    h ^= h >> np.uint64(13); h = (h * np.uint64(3266489917)) & m
    h ^= h >> np.uint64(16)
    return h.astype(np.float64) / 4294967296.0


def blob(l, b, shape, lo, hi):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * hash_unit(l, b, n)).astype(np.float32).reshape(shape)


def caffe_blobs():
    """{layer: {'0','1'}, 'batch_'+layer: {'0','1','2'}} exactly as write_full() of make_h5_golden.c stores them."""
    data, cin = {}, 1
    for l, (name, cout) in enumerate(zip(NAMES, COUTS)):
        a = np.sqrt(6.0 / (9.0 * cin))
        scale = np.float32(2.0 + 0.25 * l)
        data[name] = {'0': blob(l, 0, (cout, cin, 3, 3), -a, a), '1': blob(l, 1, (cout,), -0.1, 0.1)}
        data['batch_' + name] = {'0': blob(l, 2, (cout,), -0.05 * float(scale), 0.05 * float(scale)),
                                 '1': blob(l, 3, (cout,), 0.5 * float(scale), 1.5 * float(scale)),
                                 '2': np.array([scale], np.float32)}
        cin = cout
    return data


@pytest.fixture(scope='module')
def vgg_file(tmp_path_factory):
    """Path of the generated weight file: .h5 written by libhdf5 when gcc + the library are present, else .npz."""
    d = tmp_path_factory.mktemp('vggfile')
    inc, lib = '/opt/conda/include', '/opt/conda/lib'
    if shutil.which('gcc') and os.path.exists(os.path.join(inc, 'hdf5.h')) and os.path.exists(os.path.join(lib, 'libhdf5.so')):
        exe, out = str(d / 'make_h5_golden'), str(d / 'vgg16_full.caffemodel.h5')
        subprocess.check_call(['gcc', os.path.join(HERE, 'golden', 'make_h5_golden.c'), '-I' + inc, '-L' + lib,
                               '-Wl,-rpath,' + lib, '-lhdf5_hl', '-lhdf5', '-lm', '-o', exe])
        subprocess.check_call([exe, out, 'full'])
        return out
    out = str(d / 'vgg16_full.npz')
    flat = {}
    for g, blobs in caffe_blobs().items():
        for k, v in blobs.items():
            flat['%s/%s' % (g, k)] = v
    np.savez(out, **flat)
    return out


def test_full_size_file_is_read_and_folded_exactly(vgg_file):
    from imm_amd.utils.vgg_weights import load_vgg16, fold_batch_norm
    if not vgg_file.endswith('.h5'):
        print('\nNOTE: gcc/libhdf5 not found - the hdf5_lite part of this test is SKIPPED, .npz branch only')
    got = load_vgg16(vgg_file)
    data = caffe_blobs()
    if vgg_file.endswith('.h5'):
        from imm_amd.utils.hdf5_lite import H5File
        tree = H5File(vgg_file).load('/')['data']
        assert sorted(tree) == sorted(data)
        for g in ('conv1_1', 'conv3_2', 'conv5_3', 'batch_conv4_1'):
            for k, v in data[g].items():
                assert np.array_equal(np.asarray(tree[g][k]), v), (g, k)      # hdf5_lite vs the generator, bit for bit
    for name in NAMES[:12]:
        w = data[name]['0'].transpose(2, 3, 1, 0)
        bn = data['batch_' + name]
        w_ref, b_ref = fold_batch_norm(w, data[name]['1'], bn['0'], bn['1'], bn['2'])
        assert np.array_equal(got['vgg16/%s/weights' % name].numpy(), w_ref), name
        assert np.array_equal(got['vgg16/%s/biases' % name].numpy(), b_ref), name
    # independent restatement of the folding for one layer (vgg16.py:33-39,80-85): W/sigma, (b - mu)/sigma
    bn = data['batch_conv2_1']
    sigma = np.sqrt(1e-5 + bn['1'].astype(np.float64) / float(bn['2'][0]))
    mu = bn['0'].astype(np.float64) / float(bn['2'][0])
    np.testing.assert_allclose(got['vgg16/conv2_1/biases'].numpy(), (data['conv2_1']['1'] - mu) / sigma, rtol=1e-6)
    np.testing.assert_allclose(got['vgg16/conv2_1/weights'].numpy()[1, 2, 5, :],
                               data['conv2_1']['0'][:, 5, 1, 2] / sigma, rtol=1e-6)


@pytest.mark.gpu
def test_step_with_weights_loaded_from_the_file_matches_oracle(vgg_file):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils.box import Box
    from imm_amd.utils.vgg_weights import load_vgg16
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    cfg = O.default_model_config(10)
    cfg_file = Box(dict(cfg))
    cfg_file.perceptual = Box(dict(cfg.perceptual)); cfg_file.perceptual.net_file = vgg_file
    model = IMMModel(cfg_file, dtype=torch.bfloat16, device='cuda:0')
    assert model.vgg_source == os.path.abspath(vgg_file)
    inputs = O.synthetic_inputs(2, 128, seed=0)
    ts = TrainStep(model, 2, 128, world_size=1, use_graph=False)
    eng = ts.engine
    _, loss, _, t = model.build(inputs, True, output_tensors=True)
    torch.cuda.synchronize()
    P, St = O.init_params(cfg, 128)
    W = load_vgg16(vgg_file)
    for k in list(St):
        if k.startswith('vgg16/'):
            assert not torch.equal(St[k], W[k])              # really different from the synthetic stand-in
            St[k] = W[k]
            assert torch.equal(eng.vgg_w[k].cpu(), W[k]), k  # the engine holds the file's folded weights
    out = O.forward(P, St, inputs, cfg, training=True)
    mu_err = float((t['gauss_yx'].cpu() - out['gauss_yx']).abs().max())
    terms_eng = [float(v) for v in eng.loss_terms.cpu()]
    terms_ref = [float(v) for v in out['loss_terms']]
    terms_rel = max(abs(a - b) / abs(b) for a, b in zip(terms_eng, terms_ref))
    loss_rel = abs(float(loss) - float(out['loss'])) / abs(float(out['loss']))
    print('\nVGG_FILE mu_maxabs %.3g loss_rel %.3g terms_rel %.3g terms %s' % (mu_err, loss_rel, terms_rel, terms_ref))
    assert mu_err < 1e-3 and loss_rel < 1e-3 and terms_rel < 1e-2
    # every tapped VGG feature of the file-loaded network against the oracle's (bf16 storage drift, no jumps)
    for name, (y, _h) in eng.vgg_activations().items():
        ref = out['acts']['vgg'][name]
        e = float((y.float().cpu() - ref).norm() / ref.norm())
        assert e < 0.2, (name, e)
    # backward through the file-loaded network: the well-conditioned tensors of the last renderer convs
    eng.backward()
    torch.cuda.synchronize()
    Pe = type(P)((k, bf(v) if k.endswith('/w') else v) for k, v in P.items())
    Se = type(St)((k, bf(v) if (k.startswith('vgg16/') and k.endswith('/weights') and 'conv1_1' not in k) else v) for k, v in St.items())
    _o, g_e = O.loss_and_grads(Pe, Se, inputs, cfg, act_round=bf)
    for k, lim in (('model/renderer/conv_8/w', 2e-2), ('model/renderer/conv_8/b', 5e-3), ('model/renderer/conv_7/gamma', 5e-2)):
        e = float((eng.gview[k].cpu() - g_e[k]).norm() / g_e[k].norm())
        print('VGG_FILE grad %-32s rel %.3g' % (k, e))
        assert e < lim, (k, e)
