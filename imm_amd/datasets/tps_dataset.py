"""Single-image datasets turned into (image, future_image, mask) pairs by two random thin-plate-spline warps
(imm/datasets/tps_dataset.py) — host index/sampling logic with the reference's method names, pixel work on the GPU.

Per batch (tps_dataset.py:134-158 + the dataset's `_proc_im_pair`):
  host   sample_image_pair -> [shuffle] -> batch -> JPEG decode (thread pool)         (PairBatchLoader)
  GPU    imm_resize_crop_u8: u8 -> float, bilinear align_corners resize, central crop, written as channels 1..3 of the
         mask||image stack; channel 0 = the smooth border mask (tps_dataset.py:47-67)
  GPU    imm_tps_warp x2: future = target_warp(stack), image = source_warp(future)     (tps_dataset.py:70-96)
Landmarks are rescaled on the host exactly as the reference does (they are not warped: `landmarks and tps` is refused,
tps_dataset.py:28-29)."""
import os.path as osp
from collections import OrderedDict

import numpy as np
import torch

from .. import ops
from ..data.tps import TPSPairAugmenter
from .impair_dataset import ImagePairDataset, PairBatchLoader


def smooth_step(n, b):
    """tps_dataset.py:47-50: 0.5 + 0.5 * tanh(linspace(-1, 1, n) / b), float32."""
    x = np.linspace(-1.0, 1.0, n).astype(np.float32)
    return (np.float32(0.5) + np.float32(0.5) * np.tanh(x / np.float32(b))).astype(np.float32)


def smooth_mask(h, w, margin, step):
    """tps_dataset.py:53-67: separable mask, 0 on a `margin`-wide border, tanh ramps of `step` pixels, 1 inside."""
    b = 0.4

    def strip(size):
        return np.concatenate([np.zeros(margin, np.float32), smooth_step(step, b),
                               np.ones(size - 2 * margin - 2 * step, np.float32), smooth_step(step, -b),
                               np.zeros(margin, np.float32)])
    return strip(h)[:, None] * strip(w)[None]


class TPSDataset(ImagePairDataset):
    LANDMARK_LABELS = {}          # label name -> index into the landmarks (e.g. the two eyes, used by the evaluation)
    N_LANDMARKS = 0
    EXTRA_FIELDS = {}             # further per-sample fields of a dataset: name -> (dtype, shape)

    def __init__(self, data_dir, subset, max_samples=None, image_size=[128, 128], order_stream=False, landmarks=False,
                 tps=True, vertical_points=10, horizontal_points=10, rotsd=[0.0, 5.0], scalesd=[0.0, 0.1],
                 transsd=[0.1, 0.1], warpsd=[0.001, 0.005, 0.001, 0.01], name='TPSDataset'):
        super(TPSDataset, self).__init__(data_dir, subset, image_size=image_size, jittering=False, name=name)
        if landmarks and tps:
            raise ValueError('Outputing landmarks is not supported with TPS transform.')
        self._max_samples = max_samples
        self._order_stream = order_stream
        self._tps = tps
        self._tps_args = dict(vertical_points=vertical_points, horizontal_points=horizontal_points, rotsd=tuple(rotsd),
                              scalesd=tuple(scalesd), transsd=tuple(transsd), warpsd=tuple(warpsd))
        self._aug = {}           # device -> TPSPairAugmenter (the two device-side warps, tps_dataset.py:34-41)
        self._tps_host = None    # host-side parameter caches of the two samplers
        self._mask_dev = {}
        self._staging = None
        self._images, self._keypoints, self._image_dir = [], None, ''

    def num_samples(self):
        raise NotImplementedError()

    def _fields(self):
        """name -> (dtype, shape) of one sample, what the reference spells out per dataset in _get_sample_dtype /
        _get_sample_shape (celeba_dataset.py:120-133, aflw_dataset.py:64-78)."""
        f = OrderedDict([('image', ('string', None)), ('landmarks', ('float32', [self.N_LANDMARKS, 2]))])
        f.update(self.EXTRA_FIELDS)
        for k in self.LANDMARK_LABELS:
            f[k] = ('int32', [])
        return f

    def _get_sample_dtype(self):
        return OrderedDict((k, v[0]) for k, v in self._fields().items())

    def _get_sample_shape(self):
        return OrderedDict((k, v[1]) for k, v in self._fields().items())

    # -- sample stream (host) -------------------------------------------------------------------------------
    def _get_smooth_step(self, n, b):
        return smooth_step(n, b)

    def _get_smooth_mask(self, h, w, margin, step):
        return smooth_mask(h, w, margin, step)

    def _get_image(self, idx):
        """tps_dataset.py:99-105: file name, landmarks as (y, x), the dataset's landmark labels."""
        inputs = {'image': osp.join(self._image_dir, self._images[idx]), 'landmarks': self._keypoints[idx][:, [1, 0]]}
        inputs.update({k: v for k, v in self.LANDMARK_LABELS.items()})
        return inputs

    def _get_random_image(self):
        return self._get_image(np.random.randint(len(self._images)))

    def _get_ordered_stream(self):
        for i in range(len(self._images)):
            yield self._get_image(i)

    def sample_image_pair(self):
        """tps_dataset.py:118-131: random draws forever (or `max_samples` of them); `order_stream` walks the index once
        (the generator ends when the index or max_samples is exhausted)."""
        g = self._get_ordered_stream() if self._order_stream else None
        i_samp = 0
        while self._max_samples is None or i_samp < self._max_samples:
            if g is not None:
                try:
                    yield next(g)
                except StopIteration:
                    return
            else:
                yield self._get_random_image()
            if self._max_samples is not None:
                i_samp += 1

    # -- dataset-specific geometry ----------------------------------------------------------------------------
    def _geometry(self):
        """(resize_size, margin): images are resized to resize_size^2 and the central image_size^2 window starting at
        `margin` is kept."""
        return int(self._image_size[0]), 0

    def _proc_landmarks(self, sample, original_hw):
        return sample.get('landmarks')

    # -- device stage -----------------------------------------------------------------------------------------
    def _apply_tps(self, stack, device):
        """tps_dataset.py:70-96 on the filled mask||image stack [B,H,W,4]."""
        aug = self._aug[str(device)]
        return aug.warp_stack(stack)

    def __getstate__(self):
        """Picklable for the loader process: device-side members are rebuilt lazily where they are needed."""
        st = dict(self.__dict__)
        st['_aug'], st['_mask_dev'], st['_staging'] = {}, {}, None
        return st

    def _host_samplers(self):
        """(target, source) host-side TPS parameter caches (tps_dataset.py:34-41)."""
        if getattr(self, '_tps_host', None) is None:
            from ..data.tps import TPSParamCache
            a = self._tps_args
            kw = dict(vertical_points=a['vertical_points'], horizontal_points=a['horizontal_points'])
            self._tps_host = (TPSParamCache(rotsd=a['rotsd'][0], scalesd=a['scalesd'][0], transsd=a['transsd'][0], warpsd=a['warpsd'][:2], **kw),
                              TPSParamCache(rotsd=a['rotsd'][1], scalesd=a['scalesd'][1], transsd=a['transsd'][1], warpsd=a['warpsd'][2:], **kw))
        return self._tps_host

    def _host_pack(self, samples, decoded, out=None):
        """Everything of a batch that is host arithmetic, as plain numpy: the decoded images packed back to back
        (16-byte aligned starts) into `out` (a u8 buffer; allocated when None), their offsets and sizes, the TPS parameters
        of both warps (target first, like the reference draws them), rescaled landmarks and the label columns."""
        b = len(samples)
        sizes = np.array([d.shape[:2] for d in decoded], dtype=np.int32)
        nbytes = [int(d.size) for d in decoded]
        offs = np.zeros(b, dtype=np.int64)
        total = 0
        for i, n in enumerate(nbytes):
            offs[i] = total
            total += (n + 15) & ~15
        if out is None:
            out = np.empty(total, np.uint8)
        if out.size < total:
            raise ValueError('batch of %d bytes does not fit the %d-byte staging slot' % (total, out.size))
        for d, o, n in zip(decoded, offs, nbytes):
            assert d.shape[2] == 3 and d.dtype == np.uint8
            out[o:o + n] = d.reshape(-1)
        meta = {'n': b, 'total': total, 'offsets': offs, 'sizes': sizes}
        if self._tps:
            tgt, src = self._host_samplers()
            meta['w_target'] = tgt.sample(b)
            meta['w_source'] = src.sample(b)
        lms = [self._proc_landmarks(s, hw) for s, hw in zip(samples, sizes)]
        if lms and lms[0] is not None:
            meta['landmarks'] = np.stack(lms).astype(np.float32)
        for k in self._get_sample_dtype().keys():
            if k not in ('image', 'landmarks'):
                meta[k] = np.stack([np.asarray(s[k]) for s in samples])
        return out, meta

    def _device_stage(self, packed, meta, device):
        """packed: u8 numpy view holding meta['total'] bytes (pageable: copied into pinned staging here; the pinned
        buffers are alternated and guarded by events)."""
        dev = torch.device(device)
        if dev.type != 'cuda':
            from .._lib import ImmHipError
            raise ImmHipError('the input pipeline\'s pixel stages (resize/crop/TPS) run on the GPU only; device=%s' % device)
        key = str(dev)
        b, total = int(meta['n']), int(meta['total'])
        height, width = self._image_size[:2]
        assert height == width
        final = int(height)
        resize_sz, margin = self._geometry()
        # The loader has its own (non-blocking) stream: issued on the legacy default stream its copies and kernels would wait
        # for the blocking streams a replayed training graph runs on — the H2D of batch i+1 could not start before step i
        # had finished, and any pageable copy would stall the host for that long (measured: 6.6 k images/s from files vs
        # 8.6 k resident).  Everything the batch needs on the device travels in ONE pinned buffer (pixels, offsets, sizes,
        # TPS parameters, landmarks, labels): one async H2D per batch, nothing pageable.  Two such buffers are alternated; a
        # buffer is rewritten only after the event recorded behind its copy has completed.
        if self._staging is None:
            self._staging, self._staging_ev, self._staging_i, self._streams = [None, None], [None, None], 0, {}
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=dev)
        st = self._streams[key]
        i = self._staging_i = self._staging_i ^ 1
        if self._staging_ev[i] is not None:
            self._staging_ev[i].synchronize()
        small = [('offsets', np.ascontiguousarray(meta['offsets'], dtype=np.int64)), ('sizes', np.ascontiguousarray(meta['sizes'], dtype=np.int32))]
        if self._tps:
            small += [('w_target', np.ascontiguousarray(meta['w_target'], dtype=np.float32)),
                      ('w_source', np.ascontiguousarray(meta['w_source'], dtype=np.float32))]
        if 'landmarks' in meta:
            small.append(('landmarks', np.ascontiguousarray(meta['landmarks'], dtype=np.float32)))
        for k in self._get_sample_dtype().keys():
            if k in meta and k not in ('image', 'landmarks'):
                a = np.ascontiguousarray(meta[k])
                small.append((k, a.astype(np.int32) if a.dtype.kind in 'iu' and a.dtype.itemsize != 8 else a))
        pix = (total + 15) & ~15
        need = pix + sum((a.nbytes + 15) & ~15 for _k, a in small)
        if self._staging[i] is None or self._staging[i].numel() < need:
            self._staging[i] = torch.empty(max(need, 1 << 20), dtype=torch.uint8).pin_memory()
        stage = self._staging[i].numpy()
        stage[:total] = packed[:total]
        pos, where = pix, {}
        for k, a in small:
            stage[pos:pos + a.nbytes] = a.view(np.uint8).reshape(-1)
            where[k] = (pos, a)
            pos += (a.nbytes + 15) & ~15
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            blob = self._staging[i][:need].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
            self._staging_ev[i] = ev

            def view(k):
                p0, a = where[k]
                return blob[p0:p0 + a.nbytes].view(getattr(torch, str(a.dtype))).reshape(a.shape)

            src, offs_d, hw_d = blob[:total], view('offsets'), view('sizes')
            if key not in self._mask_dev:
                self._mask_dev[key] = ops.to_device_pinned(self._get_smooth_mask(height, width, 10, 20), dev)
                if self._tps:
                    self._aug[key] = TPSPairAugmenter((width, height), device=dev, **self._tps_args)
            mask = self._mask_dev[key]
            if self._tps:
                aug = self._aug[key]
                stack = aug.stack(b)
                stack[..., 0] = mask
                ops.resize_crop_u8(src, offs_d, hw_d, 3, (resize_sz, resize_sz), (margin, margin), (final, final), stack[..., 1:])
                out = aug.warp_stack(stack, w_target=view('w_target'), w_source=view('w_source'))
                out['mask'] = out['mask'].unsqueeze(-1)
            else:
                image = torch.empty(b, final, final, 3, dtype=torch.float32, device=dev)
                ops.resize_crop_u8(src, offs_d, hw_d, 3, (resize_sz, resize_sz), (margin, margin), (final, final), image)
                out = {'image': image, 'future_image': image, 'mask': mask.reshape(1, final, final, 1).expand(b, -1, -1, -1)}
            if 'landmarks' in where:
                out['landmarks'] = view('landmarks')
                out['future_landmarks'] = out['landmarks']
            for k, _a in small[2:]:
                if k not in ('w_target', 'w_source', 'landmarks'):
                    out[k] = view(k)
        # hand-over: the caller's stream is ordered behind the loader's (a GPU-side dependency, the host does not wait)
        cur.wait_stream(st)
        for v in out.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return out

    def _device_batch(self, samples, decoded, device):
        packed, meta = self._host_pack(samples, decoded)
        return self._device_stage(packed, meta, device)

    def get_dataset(self, batch_size, repeat=False, shuffle=False, num_preprocess_threads=12, keep_aspect=True,
                    prefetch=True, device=None, rank=0, world=1, loader_process=None):
        """tps_dataset.py:134-158.  Returns an iterable of device batches (dicts of tensors): 'image', 'future_image'
        [B,S,S,3] float32 in [0,255], 'mask' [B,S,S,1], 'landmarks'/'future_landmarks' [B,N,2] (y, x) pixels, and the
        dataset's label keys."""
        if device is None:
            device = 'cuda:%d' % torch.cuda.current_device()
        return PairBatchLoader(self, batch_size, repeat=repeat, shuffle=shuffle, num_preprocess_threads=num_preprocess_threads,
                               prefetch=prefetch, device=device, rank=rank, world=world, loader_process=loader_process)
