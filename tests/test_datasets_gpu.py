"""Input pipelines on the GPU: imm_resize_crop_u8 through the C-ABI against the numpy restatement
(bit-exact: same float32 operations in the same order), and the CelebA / AFLW loaders end to end — decoded files ->
device batches -> training step / evaluation."""
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dataset_fixtures import make_aflw_tree, make_celeba_tree     # noqa: E402
from oracle import image_oracle as IO                             # noqa: E402
from oracle import tps_oracle as T                                # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def celeba(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('celeba'))
    names, pixels = make_celeba_tree(root, n=40)
    return root, names, pixels


@pytest.fixture(scope='module')
def aflw(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('aflw'))
    return root, make_aflw_tree(root)


def _pack(imgs):
    offs, total = [], 0
    for im in imgs:
        offs.append(total)
        total += (im.size + 15) & ~15
    buf = np.zeros(total, np.uint8)
    for im, o in zip(imgs, offs):
        buf[o:o + im.size] = im.reshape(-1)
    return (torch.from_numpy(buf).to(DEV), torch.tensor(offs, dtype=torch.int64, device=DEV),
            torch.tensor([im.shape[:2] for im in imgs], dtype=torch.int32, device=DEV))


@pytest.mark.parametrize('sizes,c,resize,margin,final', [
    ([(218, 178), (200, 160), (1024, 1024), (64, 300)], 3, 160, 16, 128),       # the CelebA geometry, ragged sources
    ([(90, 100), (128, 128), (129, 127)], 3, 128, 0, 128),                      # AFLW: no crop; identity-size source
    ([(40, 30)], 1, 80, 8, 64),
    ([(33, 47), (2, 2), (1, 5)], 4, 17, 3, 11),                                 # degenerate sources, odd sizes
    ([(10, 10)] * 70, 3, 20, 2, 16),                                            # batch > 64
], ids=['celeba', 'aflw', 'grey', 'odd', 'many'])
def test_resize_crop_bit_exact(sizes, c, resize, margin, final):
    from imm_amd import ops
    rng = np.random.RandomState(len(sizes))
    imgs = [rng.randint(0, 256, size=(h, w, c)).astype(np.uint8) for h, w in sizes]
    src, offs, hw = _pack(imgs)
    out = torch.full((len(imgs), final, final, c), float('nan'), device=DEV)
    ops.resize_crop_u8(src, offs, hw, c, (resize, resize), (margin, margin), (final, final), out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i, im in enumerate(imgs):
        want = IO.resize_bilinear(im, resize, resize)[margin:margin + final, margin:margin + final]
        np.testing.assert_array_equal(got[i], want)


def test_resize_crop_into_stack_channels_and_errors():
    from imm_amd import ops
    from imm_amd._lib import ImmHipError
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 256, size=(50, 60, 3)).astype(np.uint8) for _ in range(3)]
    src, offs, hw = _pack(imgs)
    stack = torch.full((3, 32, 32, 4), -7.0, device=DEV)
    ops.resize_crop_u8(src, offs, hw, 3, (40, 40), (4, 4), (32, 32), stack[..., 1:])
    torch.cuda.synchronize()
    got = stack.cpu().numpy()
    assert (got[..., 0] == -7.0).all()                       # the mask plane is untouched
    for i, im in enumerate(imgs):
        np.testing.assert_array_equal(got[i, ..., 1:], IO.resize_bilinear(im, 40, 40)[4:36, 4:36])
    with pytest.raises(ImmHipError):                          # crop window leaves the resized image
        ops.resize_crop_u8(src, offs, hw, 3, (40, 40), (10, 10), (32, 32), stack[..., 1:])
    with pytest.raises(ImmHipError):
        ops.resize_crop_u8(src, offs, hw, 5, (40, 40), (4, 4), (32, 32), stack, ld_dst=4)


def test_celeba_loader_without_tps(celeba):
    from imm_amd.datasets import CelebADataset
    root, names, pixels = celeba
    ds = CelebADataset(root, 'test', dataset='mafl', order_stream=True, tps=False)
    batches = list(ds.get_dataset(4, device=DEV))
    assert [b['image'].shape[0] for b in batches] == [4, 2]
    want_mask = IO.smooth_mask(128, 128)
    k = 0
    for b in batches:
        assert b['image'].dtype == torch.float32 and tuple(b['image'].shape[1:]) == (128, 128, 3)
        assert b['future_image'] is b['image'] or torch.equal(b['future_image'], b['image'])
        assert tuple(b['mask'].shape) == (b['image'].shape[0], 128, 128, 1)
        np.testing.assert_array_equal(b['mask'][0, ..., 0].cpu().numpy(), want_mask)
        for i in range(b['image'].shape[0]):
            px = pixels[names[30 + k]]
            np.testing.assert_array_equal(b['image'][i].cpu().numpy(), IO.celeba_image(px, 128))
            lm = IO.resize_points(ds._keypoints[k][:, [1, 0]], px.shape[:2], [160, 160]) - 16
            np.testing.assert_array_equal(b['landmarks'][i].cpu().numpy(), lm.astype(np.float32))
            k += 1
        assert torch.equal(b['future_landmarks'], b['landmarks'])
        assert b['left_eye'].tolist() == [0] * b['image'].shape[0] and b['right_eye'].tolist() == [1] * b['image'].shape[0]
    # a second pass over the loader restarts the stream (initializable iterator)
    again = list(ds.get_dataset(4, device=DEV, prefetch=False))
    assert torch.equal(again[1]['image'], batches[1]['image'])


def test_aflw_loader_without_tps(aflw):
    from imm_amd.datasets import AFLWDataset
    root, pixels = aflw
    ds = AFLWDataset(root, 'val', order_stream=True, tps=False, image_size=[64, 64])
    (b,) = list(ds.get_dataset(8, device=DEV))
    assert b['image'].shape == (2, 64, 64, 3) and b['size'].shape == (2, 2)
    for i, nm in enumerate(['train_018.png', 'train_019.png']):
        np.testing.assert_array_equal(b['image'][i].cpu().numpy(), IO.aflw_image(pixels[nm], 64))
        assert b['size'][i].tolist() == list(pixels[nm].shape[:2])
        np.testing.assert_array_equal(b['landmarks'][i].cpu().numpy(),
                                      IO.resize_points(ds._keypoints[i][:, [1, 0]], pixels[nm].shape[:2], [64, 64]))


def test_celeba_loader_with_tps_matches_oracle(celeba):
    """Full pipeline of tps_dataset.py:134-158: the same TPS parameters (replayed from the seeds) through the numpy
    restatements."""
    from imm_amd.data.tps import TPSPairAugmenter
    from imm_amd.datasets import CelebADataset
    root, names, pixels = celeba
    ds = CelebADataset(root, 'test', dataset='mafl', order_stream=True, max_samples=4)
    random.seed(11); np.random.seed(11)
    (b,) = list(ds.get_dataset(4, device=DEV, prefetch=False))
    random.seed(11); np.random.seed(11)
    replay = TPSPairAugmenter((128, 128), device=DEV)
    wt = replay.target.sample_params(4).cpu().numpy()
    ws = replay.source.sample_params(4).cpu().numpy()
    img = np.stack([IO.celeba_image(pixels[n], 128) for n in names[30:34]])
    mask = np.broadcast_to(IO.smooth_mask(128, 128)[None, :, :, None], (4, 128, 128, 1))
    ref = T.apply_pair(img, mask, wt, ws)
    for k in ('image', 'future_image', 'mask'):
        got = b[k].cpu().numpy()
        assert got.shape == ref[k].shape
        np.testing.assert_allclose(got, ref[k], rtol=0, atol=2e-2)
        assert float(np.abs(got - ref[k]).mean()) < 5e-4
    assert float(np.abs(b['image'].cpu().numpy() - img).mean()) > 1.0           # the warps did something
    assert 'landmarks' in b


def test_training_and_eval_from_loaders(celeba):
    """CelebA loader -> TrainStep (2 steps) and MAFL loaders -> Ridge regression error: the pipeline feeds the hot path."""
    from imm_amd.datasets import CelebADataset
    from imm_amd.eval import eval_imm
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils.box import Box
    from oracle import imm_oracle as O
    root, _, _ = celeba
    model = IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.bfloat16, device=DEV)
    ts = TrainStep(model, 4, 128, world_size=1, use_graph=True)
    train = CelebADataset(root, 'train', dataset='celeba').get_dataset(4, repeat=True, device=DEV)
    it = iter(train)
    losses = []
    for _ in range(3):
        losses.append(float(ts.step(next(it))))
    ts.synchronize()
    it.close()
    assert all(np.isfinite(losses)) and int(model.engine.step_count) == 3
    tr = CelebADataset(root, 'train', dataset='mafl', order_stream=True, tps=False).get_dataset(4, device=DEV)
    te = CelebADataset(root, 'test', dataset='mafl', order_stream=True, tps=False).get_dataset(4, device=DEV)
    err = eval_imm.evaluate_regression(model, tr, te, [128, 128], batch_size=4)
    assert np.isfinite(err) and err > 0


def _run_script(path, argv):
    import runpy
    old = sys.argv
    sys.argv = [path] + argv
    try:
        runpy.run_path(path, run_name='__main__')
    finally:
        sys.argv = old


def test_train_and_test_scripts_on_dataset_tree(celeba, tmp_path, capsys):
    """scripts/train.py with the configured dataset (CelebA loader -> train loop services: summaries every 10 steps,
    test pass every n_test, checkpoint every ncheckpoint, resume from the restored global step), then scripts/test.py
    on the checkpoint with the MAFL loaders."""
    import json
    import yaml
    root, _, _ = celeba
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = yaml.safe_load(open(os.path.join(repo, 'configs', 'experiments', 'celeba-10pts.yaml')))
    base['training'].update({'datadir': root, 'logdir': str(tmp_path / 'logs'), 'batch': 4, 'ncheckpoint': 2, 'n_test': 2})
    base['training']['test_dset_params']['max_samples'] = 6
    base['model']['perceptual']['net_file'] = 'synthetic'
    cfg = str(tmp_path / 'exp.yaml')
    with open(cfg, 'w') as f:
        yaml.safe_dump(base, f)
    train_py = os.path.join(repo, 'scripts', 'train.py')
    _run_script(train_py, ['--configs', cfg, '--num-steps', '3'])
    out = capsys.readouterr().out
    logs = tmp_path / 'logs'
    assert (logs / 'model.ckpt-0.pt').exists() and (logs / 'model.ckpt-2.pt').exists()
    assert 'test: step 0' in out and 'test: step 2' in out and 'iteration through test set finished' in out
    recs = [json.loads(l) for l in open(logs / 'summaries.jsonl')]
    assert [r['step'] for r in recs if r['tag'] == 'train'] == [0]
    assert [r['step'] for r in recs if r['tag'] == 'test'] == [0, 2] and all(r['n_samples'] == 6 for r in recs if r['tag'] == 'test')
    assert abs(recs[0]['lr'] - 1e-3) < 1e-9 and len(recs[0]['loss_terms']) == 6
    # the reference's own scalar summaries (round 5): cost moving averages — TF 1.10's biased shadow that starts at 0, so after
    # the first training step the average is 0.01 x the value (base_model.py:52-60) — and the activation scale of every VGG16 layer (selfsup/vgg16.py:232-234)
    r0 = recs[0]
    assert abs(r0['loss_total_raw'] - r0['loss']) <= 1e-5 * abs(r0['loss'])
    for name in ('reconstruction_loss', 'weights_loss', 'loss_total'):
        assert abs(r0[name + '_avg'] - 0.01 * r0[name + '_raw']) <= 1e-4 * abs(0.01 * r0[name + '_raw']), name
    assert abs(r0['reconstruction_loss_raw'] + r0['weights_loss_raw'] - r0['loss_total_raw']) <= 1e-5 * r0['loss_total_raw']
    acts = [k for k in r0 if k.startswith('activation/')]
    assert 'activation/conv1_2' in acts and 'activation/conv5_2' in acts and len(acts) == 12 and all(r0[k] > 0 for k in acts)
    # the same summaries as a TensorBoard event file, with the image summaries of step 0
    import glob
    from imm_amd.utils import tf_events as E
    (evpath,) = glob.glob(str(logs / 'events.out.tfevents.*'))
    evs = E.read_events(evpath)
    first = [e for e in evs if 'train/loss' in e['scalars']][0]
    assert first['step'] == 0 and abs(first['scalars']['train/lr'] - 1e-3) < 1e-9
    assert sorted(first['images']) == ['train/future_im', 'train/future_im_pred', 'train/im', 'train/pose_embedding']
    assert first['images']['train/im'][:3] == (128, 3 * 128, 3)
    assert [e['step'] for e in evs if 'test/loss' in e['scalars']] == [0, 2]
    ck = torch.load(logs / 'model.ckpt-2.pt', map_location='cpu')
    assert ck['step'] == 3
    # resume: the loop continues at the restored global step (3) and stops at num_steps (5)
    _run_script(train_py, ['--configs', cfg, '--num-steps', '5', '--checkpoint', str(logs / 'model.ckpt-2.pt'), '--restore-optim',
                           '--tf-checkpoints'])
    out = capsys.readouterr().out
    assert 'step 3, loss' in out and 'step 0, loss' not in out and (logs / 'model.ckpt-4.pt').exists()
    assert torch.load(logs / 'model.ckpt-4.pt', map_location='cpu')['step'] == 5
    # TensorFlow bundles next to the .pt files; resuming from the bundle prefix continues at its global step
    assert (logs / 'model.ckpt-4.index').exists() and (logs / 'model.ckpt-4.data-00000-of-00001').exists()
    _run_script(train_py, ['--configs', cfg, '--num-steps', '6', '--checkpoint', str(logs / 'model.ckpt-4'), '--restore-optim'])
    out = capsys.readouterr().out
    assert 'TensorFlow bundle' in out and 'step 5, loss' in out and 'step 4, loss' not in out
    _run_script(os.path.join(repo, 'scripts', 'test.py'), ['--configs', cfg, '--train-dataset', 'mafl', '--test-dataset', 'mafl',
                                                           '--iteration', '4', '--batch-size', '4'])
    assert 'error on mafl datset test set:' in capsys.readouterr().out
    _run_script(os.path.join(repo, 'scripts', 'test.py'), ['--configs', cfg, '--train-dataset', 'mafl', '--test-dataset', 'mafl',
                                                           '--checkpoint', str(logs / 'model.ckpt-4.pt'), '--batch-size', '4'])
    out = capsys.readouterr().out
    assert 'RESULTS' in out and 'error on mafl datset test set:' in out


def test_periodic_test_pass_sees_trained_weights(celeba):
    """Engines built for other batch sizes mirror the trained engine every time they are selected (a stale copy would
    report the loss of the initial weights forever)."""
    from imm_amd.datasets import CelebADataset
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train import cnn_train_multi as tru
    from imm_amd.utils.box import Box
    from oracle import imm_oracle as O
    root, _, _ = celeba
    model = IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.bfloat16, device=DEV,
                     hparams=dict(lr_start=1e-2, lr_decay=0.95, lr_step=100000, lr_multiple=1.0, clip=1.0))
    ts = tru.TrainStep(model, 4, 128, world_size=1, use_graph=True)
    test = CelebADataset(root, 'test', dataset='mafl', order_stream=True, max_samples=3, tps=False).get_dataset(3, device=DEV)
    train = iter(CelebADataset(root, 'train', dataset='celeba').get_dataset(4, repeat=True, device=DEV))
    l0 = tru.run_test_pass(model, test, 0, verbose=False)
    for _ in range(4):
        ts.step(next(train))
    ts.synchronize()
    l1 = tru.run_test_pass(model, test, 4, verbose=False)
    train.close()
    other = model._engines[(3, 128)]
    assert other is not ts.engine and int(other.step_count) == 4
    assert torch.equal(other.params, ts.engine.params)
    assert l0 != l1


def test_visualize_script(celeba, tmp_path, capsys):
    """scripts/visualize.py (examples/visualize.ipynb upstream): images from a folder -> landmarks drawn on a sheet."""
    from PIL import Image
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils import tf_checkpoint as T
    from imm_amd.utils.box import Box
    from oracle import imm_oracle as O
    root, names, pixels = celeba
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    imdir = tmp_path / 'faces'
    imdir.mkdir()
    for n in names[:5]:
        Image.fromarray(pixels[n]).save(imdir / (n[:-4] + '.png'))
    eng = IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.bfloat16, device=DEV)._get_engine(5, 128)
    ckpt = str(tmp_path / 'model.ckpt')
    T.save_tf_checkpoint(eng, ckpt, with_optimizer=False)
    out_png, out_npy = str(tmp_path / 'sheet.png'), str(tmp_path / 'lm.npy')
    _run_script(os.path.join(repo, 'scripts', 'visualize.py'), [
        '--configs', os.path.join(repo, 'tests', 'configs', 'paths.yaml'), os.path.join(repo, 'tests', 'configs', 'smoke-10pts.yaml'),
        '--images-dir', str(imdir), '--checkpoint', ckpt, '--out', out_png, '--save-landmarks', out_npy])
    assert 'wrote' in capsys.readouterr().out
    sheet = Image.open(out_png)
    assert sheet.size == (4 * 384, 2 * 384)                      # 5 images: 4 columns x 2 rows of 3x-enlarged tiles
    lm = np.load(out_npy)
    assert lm.shape == (5, 10, 2) and (lm >= 0).all() and (lm <= 128).all()
    # same landmarks as the engine run directly on the PIL-resized images
    imgs = np.stack([np.array(Image.fromarray(pixels[n]).resize((128, 128)), dtype=np.float32) for n in names[:5]])
    x = torch.from_numpy(imgs).to(DEV)
    eng.set_inputs(x, x, torch.ones(5, 128, 128, 1)); eng.forward_model_only(False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(lm, ((eng.mu.float().cpu().numpy() + 1) / 2) * 128, atol=1e-3)
