"""Interface of a dataset returning image pairs — host side (imm/datasets/impair_dataset.py).

Same constructor arguments and hook names as the reference's `ImagePairDataset`; the tf.data graph of the reference
(`from_generator -> repeat -> shuffle(2000) -> map(_proc_im_pair, threads) -> batch -> [map(_apply_tps)] -> prefetch(1)`,
impair_dataset.py:236-253 / tps_dataset.py:134-158) becomes `PairBatchLoader`: the sample stream, the shuffle buffer,
the batching and the JPEG decode (thread pool) run on the host in a producer thread; the consumer packs the decoded u8
images into one pinned buffer, copies it to HBM once and runs the geometry on the GPU (imm_resize_crop_u8, then
imm_tps_warp).  There is no CPU implementation of the pixel work: without libimm_hip.so iterating a loader raises.

The box helpers (`_find_common_box`, `_fit_bbox`, `_crop_to_box`) and the flip/swap jitter are host arithmetic on
numpy arrays with the reference's semantics; the shipped datasets (CelebA, AFLW) do not use them (TPSDataset passes
jittering=False and overrides `_proc_im_pair`)."""
import atexit
import collections
import os
import queue
import random
import subprocess
import sys
import tempfile
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def decode_image(image, channels=3):
    """impair_dataset.py:38-51 + data_utils/image_utils.py:9-36: a file name is read and decoded (JPEG/PNG, by content)
    to u8 HWC with `channels` channels; an array passes through."""
    if isinstance(image, (str, bytes)):
        from PIL import Image
        with Image.open(image) as im:
            im = im.convert('RGB' if channels == 3 else 'L')
            a = np.asarray(im, dtype=np.uint8)
        return a if a.ndim == 3 else a[:, :, None]
    a = np.asarray(image)
    if a.dtype != np.uint8:
        raise TypeError('image tensors must be uint8 HWC (got %s)' % a.dtype)
    return a if a.ndim == 3 else a[:, :, None]


class DecodeWorkers(object):
    """`n` decoder PROCESSES (imm_amd/datasets/_decode_worker.py) writing u8 HWC pixels into a ring of fixed-size slots
    in a shared-memory file; `decode(item, slot)` is called from a thread pool (one blocking request per idle worker).
    tf.data's `map(..., num_parallel_calls=12)` runs its decoders on C++ threads; PIL's per-image Python work holds the
    GIL, so the parallelism that scales here is processes."""

    def __init__(self, n, n_slots, slot_bytes=4 << 20):
        self.n, self.n_slots, self.slot_bytes = int(n), int(n_slots), int(slot_bytes)
        shm_dir = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
        fd, self.ring_path = tempfile.mkstemp(prefix='imm_decode_ring_', dir=shm_dir)
        os.ftruncate(fd, self.n_slots * self.slot_bytes)      # sparse: only touched pages are ever backed
        import mmap
        self._mm = mmap.mmap(fd, self.n_slots * self.slot_bytes)
        os.close(fd)
        self.ring = np.frombuffer(self._mm, dtype=np.uint8)      # plain ndarray views (np.memmap's subclass is slow to slice)
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_decode_worker.py')
        self.procs = [subprocess.Popen([sys.executable, worker, self.ring_path, str(self.slot_bytes)], stdin=subprocess.PIPE,
                                       stdout=subprocess.PIPE, universal_newlines=True, bufsize=1) for _ in range(self.n)]
        self.idle = queue.Queue()
        for p in self.procs:
            self.idle.put(p)
        self._closed = False
        atexit.register(self.close)

    def _view(self, reply, path, slot, channels):
        if not reply:
            raise RuntimeError('a decode worker exited')
        parts = reply.split()
        if parts[0] == 'err':
            raise RuntimeError('decode of %s failed in the worker: %s' % (path, reply[4:].strip()))
        if parts[0] == 'big':
            return decode_image(path, channels)
        h, w, c = int(parts[0]), int(parts[1]), int(parts[2])
        o = slot * self.slot_bytes
        return self.ring[o:o + h * w * c].reshape(h, w, c)

    def decode_chunk(self, items, slots, channels=3):
        """Decode `items` (file names or u8 arrays) on ONE idle worker, pipelined: all requests are written, then the
        replies read.  -> list of u8 HWC arrays: views of the ring slots (valid until a slot is handed out again) for
        files, the array itself for in-memory images."""
        out = [None] * len(items)
        todo = []
        for i, item in enumerate(items):
            path = item.decode() if isinstance(item, bytes) else item
            if not isinstance(path, str) or '\n' in path:
                out[i] = decode_image(item, channels)
            else:
                todo.append((i, path))
        if todo:
            p = self.idle.get()
            try:
                p.stdin.write(''.join('%d %d %s\n' % (slots[i], channels, path) for i, path in todo))
                p.stdin.flush()
                replies = [p.stdout.readline() for _ in todo]
            finally:
                self.idle.put(p)
            for (i, path), reply in zip(todo, replies):
                out[i] = self._view(reply, path, slots[i], channels)
        return out

    def submit_batch(self, pool, items, base_slot, channels=3):
        """The batch split into one contiguous chunk per worker, chunks queued on the thread pool; returns a function
        that waits for them and returns the list of arrays."""
        n = len(items)
        k = max(1, min(self.n, n))
        bounds = [n * j // k for j in range(k + 1)]
        futs = [pool.submit(self.decode_chunk, items[a:b], list(range(base_slot + a, base_slot + b)), channels)
                for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        return lambda: [a for f in futs for a in f.result()]

    def decode(self, item, slot, channels=3):
        return self.decode_chunk([item], [slot], channels)[0]

    def close(self):
        if self._closed:
            return
        self._closed = True
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
        try:
            os.unlink(self.ring_path)       # the mapping itself stays valid for arrays still referencing it
        except OSError:
            pass


class ImagePairDataset(object):
    """Abstract class for sampling image pairs (impair_dataset.py:15-37)."""

    def __init__(self, data_dir, subset, image_size=[128, 128], bbox_padding=[10, 10], crop_to_bbox=False, jittering=None,
                 augmentations=['flip', 'swap'], name='PairDataset'):
        self._data_dir = data_dir
        self._subset = subset
        self._image_size = list(image_size)
        self.image_size = list(image_size)
        self._bbox_padding = bbox_padding
        self._crop_to_bbox = crop_to_bbox
        self._jittering = jittering
        self._augmentations = augmentations
        self._name = name

    # -- box / point helpers (host arithmetic) -----------------------------------------------------------
    def _find_common_box(self, box1, box2):
        """Union of two [ymin, xmin, ymax, xmax] boxes (impair_dataset.py:54-62)."""
        box1, box2 = np.asarray(box1), np.asarray(box2)
        return np.concatenate([np.minimum(box1[:2], box2[:2]), np.maximum(box1[2:], box2[2:])])

    def _fit_bbox(self, box, image_sz):
        """Grow one side so that the box has the target aspect ratio, centre kept; int32 (truncation) like
        impair_dataset.py:65-93."""
        box = np.asarray(box, dtype=np.float32)
        im_h, im_w = np.float32(image_sz[0]), np.float32(image_sz[1])
        h, w = box[2] - box[0], box[3] - box[1]
        r_im, r = im_w / im_h, w / h
        cy, cx = box[0] + h / np.float32(2), box[1] + w / np.float32(2)
        if r < r_im:
            w = r_im * h
        else:
            h = (np.float32(1) / r_im) * w
        out = np.array([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], dtype=np.float32)
        return out.astype(np.int32)

    def _crop_to_box(self, image, bbox, pad=True):
        """Crop an HWC array to the box, zero-padding where the box leaves the image (impair_dataset.py:96-113)."""
        bbox = [int(v) for v in bbox]
        if pad:
            h, w = image.shape[:2]
            top, left = -min(0, bbox[0]), -min(0, bbox[1])
            bottom, right = -min(0, h - bbox[2]), -min(0, w - bbox[3])
            image = np.pad(image, [[top, bottom], [left, right], [0, 0]])
            bbox = [bbox[0] + top, bbox[1] + left, bbox[2] + top, bbox[3] + left]
        return image[bbox[0]:bbox[2], bbox[1]:bbox[3]]

    def _resize_points(self, points, size, new_size):
        """points [N,2] * (new_size / size), float32 ratio, cast back to the points' dtype (impair_dataset.py:116-123)."""
        points = np.asarray(points)
        ratio = np.asarray(new_size, dtype=np.float32) / np.asarray(size, dtype=np.float32)
        return (points.astype(np.float32) * ratio[None]).astype(points.dtype)

    def _jitter_im_and_points(self, im0, im1, p0, p1, flip=True, swap=True, rng=random):
        """Random horizontal flip of both images (+ their (y, x) points) and random swap (impair_dataset.py:151-173):
        each with probability 1/2."""
        if flip and not rng.random() < 0.5:
            im0, im1 = im0[:, ::-1, :], im1[:, ::-1, :]
            if p0 is not None:
                max_x = np.float32(im0.shape[1] - 1)
                p0 = np.stack([p0[:, 0], max_x - p0[:, 1]], axis=1)
                p1 = np.stack([p1[:, 0], max_x - p1[:, 1]], axis=1)
        if swap and not rng.random() < 0.5:
            im0, im1, p0, p1 = im1, im0, p1, p0
        return im0, im1, p0, p1

    def _jitter_im(self, im0, im1, flip=True, swap=True, rng=random):
        im0, im1, _, _ = self._jitter_im_and_points(im0, im1, None, None, flip, swap, rng)
        return im0, im1

    # -- to be provided by datasets -------------------------------------------------------------------------
    def _get_sample_dtype(self):
        raise NotImplementedError()

    def _get_sample_shape(self):
        return {k: None for k in self._get_sample_dtype().keys()}

    def sample_image_pair(self):
        """Generator of sample dicts ('image' = file name or u8 array, plus per-sample annotations)."""
        raise NotImplementedError()

    def num_samples(self):
        raise NotImplementedError()


_END = object()
_TEMP_FILES = set()     # shared-memory files of live loaders: removed at interpreter exit if an iterator was never closed


def _cleanup_temp_files():
    for path in list(_TEMP_FILES):
        try:
            os.unlink(path)
        except OSError:
            pass
        _TEMP_FILES.discard(path)


atexit.register(_cleanup_temp_files)


def _loader_server_main(loader, ring_path, cap, n_slots, conn, sem, seed):
    """Body of the loader process: PairBatchLoader.host_batches() -> dataset._host_pack() into shared-memory slots."""
    import mmap
    import signal
    # the parent stops this process with SIGTERM: turn it into a normal exit so that `finally` below runs (it closes
    # the decode workers and removes their shared-memory ring)
    signal.signal(signal.SIGTERM, lambda *_a: sys.exit(0))
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    fd = os.open(ring_path, os.O_RDWR)
    mm = mmap.mmap(fd, cap * n_slots)
    os.close(fd)
    ring = np.frombuffer(mm, dtype=np.uint8)
    try:
        k = 0
        for samples, decoded in loader.host_batches():
            sem.acquire()
            slot = k % n_slots
            _buf, meta = loader.dataset._host_pack(samples, decoded, ring[slot * cap:(slot + 1) * cap])
            conn.send((slot, meta))
            k += 1
        conn.send(None)
    except SystemExit:
        pass
    except BaseException as e:                  # surfaced in the consumer
        try:
            conn.send(RuntimeError('loader process: %s: %s' % (type(e).__name__, e)))
        except Exception:
            pass
    finally:
        loader.close()


class PairBatchLoader(object):
    """Iterable over device batches; every `iter()` restarts the sample stream (make_initializable_iterator +
    initializer in cnn_train_multi.py:384-400,478).  `rank`/`world` deal batches round-robin to data-parallel ranks
    (one process per GPU; the reference's in-graph towers split one batch instead, cnn_train_multi.py:131-139)."""

    DECODE_AHEAD = 2      # batches being decoded beyond the one about to be yielded
    RING_BATCHES = 8      # decoded batches that can be alive at once: decoding (3) + queue (2) + being consumed + slack

    def __init__(self, dataset, batch_size, repeat=False, shuffle=False, num_preprocess_threads=12, prefetch=True,
                 device='cuda:0', shuffle_buffer=2000, rank=0, world=1, rng=None, decode_processes=None,
                 loader_process=None):
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.repeat, self.shuffle, self.prefetch = bool(repeat), bool(shuffle), bool(prefetch)
        self.num_threads = max(1, int(num_preprocess_threads))
        self.device = device
        self.shuffle_buffer = int(shuffle_buffer)
        self.rank, self.world = int(rank), int(world)
        self.rng = rng if rng is not None else random
        # decoder processes: default = num_preprocess_threads (IMM_DECODE_PROCESSES overrides; 0 = decode on threads of
        # this process, which is what small test datasets use)
        env = os.environ.get('IMM_DECODE_PROCESSES')
        self.decode_processes = int(env) if env is not None else (self.num_threads if decode_processes is None else int(decode_processes))
        self._workers = None
        self._ring_pos = 0
        # loader process: the whole host side (sample stream, decode dispatch, packing, TPS parameter draws) runs in ONE
        # child process and hands over packed batches through shared memory, so that no Python thread competes with the
        # training loop for the GIL (measured: 6.7 k -> see DESIGN 8 images/s when training from files).  Default: on for
        # repeating (training) streams, off for one-pass streams (a process start costs seconds).  IMM_LOADER_PROCESS=0/1.
        env = os.environ.get('IMM_LOADER_PROCESS')
        self.loader_process = (env != '0') if env is not None else (self.repeat if loader_process is None else bool(loader_process))
        self._server = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st['_workers'], st['_server'] = None, None
        if st['rng'] is random:
            st['rng'] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if self.rng is None:
            self.rng = random

    # host stages ---------------------------------------------------------------------------------------------
    def _samples(self):
        while True:
            n = 0
            for s in self.dataset.sample_image_pair():
                n += 1
                yield s
            if not self.repeat or n == 0:
                return

    def _shuffled(self, it):
        """tf.data's shuffle(buffer): keep `shuffle_buffer` samples, emit a uniformly random one, refill."""
        buf = []
        for s in it:
            if len(buf) < self.shuffle_buffer:
                buf.append(s)
                continue
            i = self.rng.randrange(len(buf))
            out, buf[i] = buf[i], s
            yield out
        self.rng.shuffle(buf)
        for s in buf:
            yield s

    def host_batches(self):
        """Generator of (samples, decoded u8 arrays): everything that runs before the GPU (usable without one).  With
        decoder processes the arrays are views of a shared-memory ring: a batch stays valid while at most RING_BATCHES - 1
        newer ones have been produced (the loader consumes them immediately)."""
        it = self._samples()
        if self.shuffle:
            it = self._shuffled(it)
        if self.decode_processes > 0 and self._workers is None:
            self._workers = DecodeWorkers(self.decode_processes, self.RING_BATCHES * self.batch_size)
        n_par = self.decode_processes if self.decode_processes > 0 else self.num_threads

        def submit(pool, batch):
            if self._workers is None:
                futs = [pool.submit(decode_image, t['image']) for t in batch]
                return lambda: [f.result() for f in futs]
            base = self._ring_pos * self.batch_size
            self._ring_pos = (self._ring_pos + 1) % self.RING_BATCHES
            return self._workers.submit_batch(pool, [t['image'] for t in batch], base)

        # DECODE_AHEAD batches are in the decoders at once (a single batch split over the workers is latency-bound: every
        # worker idles until the slowest chunk is back); batches are yielded in stream order
        pending = collections.deque()
        with ThreadPoolExecutor(n_par) as pool:
            batch, index = [], 0
            for s in it:
                batch.append(s)
                if len(batch) == self.batch_size:
                    if index % self.world == self.rank:
                        pending.append((batch, submit(pool, batch)))
                        if len(pending) > self.DECODE_AHEAD:
                            b, wait = pending.popleft()
                            yield b, wait()
                    batch, index = [], index + 1
            if batch and index % self.world == self.rank:      # tf.data's batch() keeps the ragged remainder
                pending.append((batch, submit(pool, batch)))
            while pending:
                b, wait = pending.popleft()
                yield b, wait()

    def close(self):
        if self._workers is not None:
            self._workers.close()
            self._workers = None

    # device stage ----------------------------------------------------------------------------------------------
    SERVER_SLOTS = 4

    def _iter_server(self):
        import mmap
        import multiprocessing as mp
        ctx = mp.get_context('spawn')
        cap = self.batch_size * (4 << 20)
        shm_dir = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
        fd, ring_path = tempfile.mkstemp(prefix='imm_loader_ring_', dir=shm_dir)
        _TEMP_FILES.add(ring_path)
        os.ftruncate(fd, cap * self.SERVER_SLOTS)            # sparse: only the bytes of real batches are ever backed
        mm = mmap.mmap(fd, cap * self.SERVER_SLOTS)
        os.close(fd)
        ring = np.frombuffer(mm, dtype=np.uint8)
        recv, send = ctx.Pipe(duplex=False)
        sem = ctx.Semaphore(self.SERVER_SLOTS - 1)           # batches packed but not yet consumed
        seed = int(np.random.randint(0, 2 ** 31 - 1))        # the parent's numpy seed governs the child's draws
        proc = ctx.Process(target=_loader_server_main, args=(self, ring_path, cap, self.SERVER_SLOTS, send, sem, seed), daemon=True)
        proc.start()
        send.close()
        self._server = proc
        try:
            while True:
                try:
                    item = recv.recv()
                except EOFError:
                    raise RuntimeError('the loader process exited unexpectedly (exit code %s)' % proc.exitcode)
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                slot, meta = item
                out = self.dataset._device_stage(ring[slot * cap:(slot + 1) * cap], meta, self.device)
                sem.release()                                 # the slot's bytes are in pinned staging now
                yield out
        finally:
            self._server = None
            if proc.is_alive():
                proc.terminate()
            proc.join(timeout=5)
            recv.close()
            try:
                os.unlink(ring_path)
            except OSError:
                pass
            _TEMP_FILES.discard(ring_path)

    def __iter__(self):
        if self.loader_process:
            for out in self._iter_server():
                yield out
            return
        if not self.prefetch:
            for samples, decoded in self.host_batches():
                yield self.dataset._device_batch(samples, decoded, self.device)
            return
        q = queue.Queue(maxsize=2)          # prefetch(1): one batch decoded ahead of the one being consumed
        stop = threading.Event()

        def produce():
            try:
                for item in self.host_batches():
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(_END)
            except BaseException as e:      # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield self.dataset._device_batch(item[0], item[1], self.device)
        finally:
            stop.set()
