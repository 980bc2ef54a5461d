"""Training runtime — counterpart of /root/reference/imm/train/cnn_train_multi.py.

Reference: single process, in-graph towers (`train_multi` :109-192): split the batch evenly
(:132), one tower per GPU, `average_gradients` (:66-106: per-variable mean over towers THEN
tf.clip_by_norm per tensor), one `apply_gradients` (:164), BN statistics per tower, variables
hosted on the CPU and re-broadcast every step.
MI355X-native: one process per GPU (torchrun), every rank owns full weights + Adam state in HBM;
the per-tower mean becomes ONE sum all-reduce of the flat f32 gradient buffer over RCCL/xGMI
(`torch.distributed`, backend "nccl" == RCCL), the 1/N scale is folded into the fused
clip+Adam kernel (order preserved: average, then clip, then Adam).  Forward+backward and the
optimizer are HIP graphs replayed each step; nothing synchronises with the host inside a step.
BN statistics and the loss normalisers stay rank-local exactly like the reference's per-tower
statistics (:155, S11).
"""
import json
import os
import time

import torch
import torch.distributed as dist

from .. import ops


def split_inputs(inputs, num_splits, index):
    """imm/utils/utils.py:113-121 split_tensors: even split along the batch axis (cnn_train_multi.py:126,132)."""
    out = {}
    for k, v in inputs.items():
        n = v.shape[0]
        assert n % num_splits == 0, 'Batch size must be divisible by number of GPUs'
        per = n // num_splits
        out[k] = v[index * per:(index + 1) * per]
    return out


def average_gradients(flat_grads, world_size, group=None, force=False, async_op=False):
    """cnn_train_multi.py:66-106.  Sum-all-reduce of (a bucket of) the flat gradient buffer; the division by the
    number of towers happens inside imm_clip_adam_step (grad_scale = 1/world_size), before the
    per-tensor clip, as in the reference.  With async_op the RCCL work handle is returned (the collective runs
    on RCCL's stream, ordered after everything already enqueued on the current stream).
    Backend "gloo" (CPU-side tests of the multi-rank path, several ranks sharing one GPU): device tensors are bounced
    through the host — same sum, same flat layout, no claim about speed."""
    if world_size > 1 or (force and dist.is_initialized()):
        if flat_grads.is_cuda and dist.get_backend(group) == 'gloo':
            # pinned staging buffer + explicit stream synchronisation on both legs (a pageable bounce buffer that is freed right
            # after an asynchronous 16 MB host-to-device copy was the one non-deterministic piece of the two-rank test).  Both legs
            # are kernel copies (imm_copy_f32 reads / writes the pinned buffer directly) rather than hipMemcpyAsync: introduced while
            # chasing the one-rank-in-eight divergence of round 5 — which turned out to be the upload of the INITIAL weights
            # (ops.upload), not this exchange; kept because it needs no copy engine and is as fast.
            host = _pinned_like(flat_grads)
            stream = torch.cuda.current_stream(flat_grads.device)
            ops.copy_f32(host, flat_grads.detach())
            stream.synchronize()                      # the producing graph and the copy have finished
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            ops.copy_f32(flat_grads, host)
            stream.synchronize()                      # the staging buffer may be reused
            return None
        return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return None


_PINNED = {}


def _pinned_like(t):
    key = (t.numel(), t.dtype)
    if key not in _PINNED:
        _PINNED[key] = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
    return _PINNED[key].view(t.shape)


def mean_tower_loss(loss, world_size, group=None):
    """The loss the reference prints and summarises is the mean over towers (cnn_train_multi.py:173 avg_tower_loss): a
    scalar all-reduce, issued on logging steps only.  Returns a Python float."""
    if world_size > 1 and dist.is_initialized():
        t = loss.detach().reshape(1).to(torch.float32).clone()
        if t.is_cuda and dist.get_backend(group) == 'gloo':
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return float(t) / world_size
    return float(loss)


class TrainStep:
    """One rank's training step: fwd+bwd graph -> gradient all-reduce -> clip+Adam graph."""

    COLLECTIVES = ('pg', 'native', 'graph')

    def __init__(self, model, batch_per_rank, image_size, world_size=1, use_graph=True, group=None, split_graphs=False,
                 collective=None, debug_poison=None):
        """collective (how the gradient exchange of a multi-rank step is issued; identical sums):
             'pg'      torch.distributed.all_reduce on the process group's stream between the two graphs (default)
             'native'  imm_rccl_allreduce of the C-ABI on a stream of this object, ordered by events (the process group then
                       only carries the 128-byte unique id at start-up)
             'graph'   imm_rccl_allreduce issued on the CAPTURING stream between the backward and the optimizer launches: a
                       step at N > 1 is ONE graph launch, like at N = 1 (needs use_graph)
           None = IMM_RCCL_GRAPH=1 -> 'graph', IMM_RCCL_NATIVE=1 -> 'native', else 'pg' (the environment only supplies the
           default of this argument).  The bucket count is the one the ENGINE was built for (IMMModel(dp_buckets=...) /
           IMM_DP_BUCKETS at engine construction): it fixes the layout of the backward program and is read back from it.
           With 'graph' and two buckets the renderer bucket's all-reduce is captured on a FORKED stream of the same graph
           (concurrent with the encoders' backward nodes) and joined in front of the optimizer nodes.
           debug_poison (None = IMM_DEBUG_POISON_GRADS): after the optimizer has been enqueued the flat gradient buffer is
           filled with NaN on the step's stream.  Every element of it is rewritten by the next backward pass, so a correctly
           ordered step never sees the poison; an exchange that is NOT ordered between "backward finished" and "optimizer
           started" (a missing event / stream wait) reads or leaves NaNs and shows up as a NaN update.  So that this also works
           with a ONE-rank communicator on a single-GPU box (where RCCL's in-place all-reduce touches nothing), every exchange is
           followed — on the stream the collective itself was issued on — by `flat *= 2` (exact in f32): an optimizer that starts
           before the exchange has finished sees unscaled gradients, an exchange that starts before the backward has finished
           doubles NaNs; tests compare with the eager one-stream sequence fwd, bwd, grads *= 2, clip + Adam."""
        self.model = model
        self.world_size = world_size
        # two graphs with the all-reduce in between (always for world_size > 1; selectable at 1 to test that path)
        self.split = split_graphs or world_size > 1
        self.group = group
        model.require_vgg()                       # training against a missing perceptual network is an error, not a fallback
        self.engine = model._get_engine(batch_per_rank, image_size)
        model._master = self.engine               # the engine whose variables are trained; others mirror it (eval batches)
        # ONE bucket by default: [fwd + bwd graph] -> all-reduce of the flat gradient buffer -> [clip + Adam graph].
        # Two buckets (opt-in): the renderer's gradients (the tail of the flat buffer, 43 % of the 16.6 MB) are all-reduced
        # on RCCL's stream while the encoders' backward graph still runs, the encoder bucket follows it.  The asynchronous
        # two-bucket branch has only ever run with gloo on one device (where it degenerates to synchronous host bounces); it
        # stays opt-in until a >= 2-GPU RCCL run has compared its update with the one-bucket update.
        self.buckets = 2 if self.engine.n_bwd_bucket0 is not None else 1
        if abs(self.engine.hp.grad_scale - 1.0 / world_size) > 1e-9:
            raise ValueError('engine was built for world_size %g' % (1.0 / self.engine.hp.grad_scale))
        self.use_graph = use_graph
        if collective is None:
            collective = ('graph' if os.environ.get('IMM_RCCL_GRAPH', '0') != '0' else
                          'native' if os.environ.get('IMM_RCCL_NATIVE', '0') != '0' else 'pg')
        if collective not in self.COLLECTIVES:
            raise ValueError('collective must be one of %r, got %r' % (self.COLLECTIVES, collective))
        if collective == 'graph' and not use_graph:
            raise ValueError("collective 'graph' needs use_graph=True")
        self.collective = collective if self.split else None
        if debug_poison is None:
            debug_poison = os.environ.get('IMM_DEBUG_POISON_GRADS', '0') != '0'
        self.debug_poison = bool(debug_poison)
        self._poison_mask = None
        self.after_backward = None                # test hook: f(engine) on the step's stream after the backward graph(s) of a
                                                  # split step, before the exchange (fault injection: an overflow on ONE rank)
        # 'graph' has been validated with a one-rank communicator on the single-GPU test box only; multi-GPU timing has to come
        # from a node
        self.graph_resident = self.split and use_graph and collective == 'graph'
        self.native_comm = None
        if self.split and collective in ('native', 'graph'):
            rank = dist.get_rank(group) if dist.is_initialized() else 0
            with torch.cuda.device(self.engine.dev):
                self.native_comm = ops.RcclComm(rank, world_size, group)
            self.comm_stream = torch.cuda.Stream(device=self.engine.dev)
        self.stream = torch.cuda.Stream(device=self.engine.dev)
        self._graphs = None
        torch.cuda.synchronize(self.engine.dev)   # engine construction ran on the default stream

    def _probe(self, flat):
        """debug_poison: the marker every exchange leaves behind on the stream it was issued on (see __init__)."""
        if self.debug_poison:
            flat.mul_(2.0)

    def _poison(self):
        """debug_poison: NaN into every element of the flat gradient buffer that a backward pass rewrites (found once by running a
        backward pass over a NaN-filled buffer: the analytically-zero bias gradients in front of a batch norm are written at
        engine construction only and keep their zeros)."""
        eng = self.engine
        if self._poison_mask is None:
            snap = eng.snapshot()
            eng.grads.fill_(float('nan'))
            eng.forward(True); eng.backward()
            torch.cuda.current_stream(eng.dev).synchronize()
            self._poison_mask = ~torch.isnan(eng.grads)
            eng.restore(snap)
            torch.cuda.current_stream(eng.dev).synchronize()
        eng.grads.masked_fill_(self._poison_mask, float('nan'))

    # -- graph capture --------------------------------------------------------------------------------
    def _capture(self):
        eng = self.engine
        with torch.cuda.stream(self.stream):
            if self.debug_poison and self._poison_mask is None:
                snap0 = eng.snapshot(); self._poison(); eng.restore(snap0)      # builds the mask outside capture
            # warm every kernel up once outside capture (code-object loading, LDS attribute calls),
            # then roll the state back so the warm-up leaves no trace
            snap = eng.snapshot()
            eng.forward(True); eng.backward(); eng.optimizer_step()
            self.stream.synchronize()
            eng.restore(snap)
            self.stream.synchronize()
            eng._training = True
            if self.graph_resident:
                # one graph: fwd, bwd, all-reduce of the flat gradient buffer (RCCL kernels captured on this stream), clip + Adam
                self.native_comm.all_reduce_sum(eng.grads)          # once outside capture: RCCL's lazy channel set-up
                self.stream.synchronize()
                g = ops.Graph()
                g.capture_begin()
                if self.buckets < 2:
                    eng.run(eng.prog_fwd); eng.run(eng.prog_bwd)
                    self.native_comm.all_reduce_sum(eng.grads); self._probe(eng.grads)
                else:
                    # fork: the renderer bucket (the tail of the flat buffer) is all-reduced on comm_stream — captured into the
                    # same graph through the event — while the encoders' backward nodes run; the encoder bucket follows on the
                    # same stream (one communicator: its collectives are issued in one order on every rank); join before the
                    # optimizer nodes
                    n0, off = eng.n_bwd_bucket0, eng.bucket0_offset
                    eng.run(eng.prog_fwd); eng.run(eng.prog_bwd[:n0])
                    fork = torch.cuda.Event(); fork.record(self.stream)
                    self.comm_stream.wait_event(fork)
                    with torch.cuda.stream(self.comm_stream):
                        self.native_comm.all_reduce_sum(eng.grads[off:]); self._probe(eng.grads[off:])
                    eng.run(eng.prog_bwd[n0:])
                    tail = torch.cuda.Event(); tail.record(self.stream)
                    self.comm_stream.wait_event(tail)
                    with torch.cuda.stream(self.comm_stream):
                        self.native_comm.all_reduce_sum(eng.grads[:off]); self._probe(eng.grads[:off])
                    join = torch.cuda.Event(); join.record(self.comm_stream)
                    self.stream.wait_event(join)
                eng.run(eng.prog_opt)
                if self.debug_poison:
                    self._poison()
                g.capture_end()
                self._graphs = (g,)
            elif not self.split:
                g = ops.Graph()
                g.capture_begin()
                eng.run(eng.prog_fwd); eng.run(eng.prog_bwd); eng.run(eng.prog_opt)
                g.capture_end()
                self._graphs = (g,)
            elif self.buckets < 2:
                g1 = ops.Graph()
                g1.capture_begin()
                eng.run(eng.prog_fwd); eng.run(eng.prog_bwd)
                g1.capture_end()
                g2 = ops.Graph()
                g2.capture_begin()
                eng.run(eng.prog_opt)
                g2.capture_end()
                self._graphs = (g1, g2)
            else:
                # bucketed: [fwd + VGG/renderer backward] | all-reduce(renderer grads) overlapped with
                # [encoder backward] | all-reduce(encoder grads) | [clip + Adam + re-pack]
                n0 = eng.n_bwd_bucket0
                g1 = ops.Graph()
                g1.capture_begin()
                eng.run(eng.prog_fwd); eng.run(eng.prog_bwd[:n0])
                g1.capture_end()
                g1b = ops.Graph()
                g1b.capture_begin()
                eng.run(eng.prog_bwd[n0:])
                g1b.capture_end()
                g2 = ops.Graph()
                g2.capture_begin()
                eng.run(eng.prog_opt)
                g2.capture_end()
                self._graphs = (g1, g1b, g2)

    def step(self, inputs=None):
        """Runs one training step on this rank's shard; returns the (device) scalar loss."""
        eng = self.engine
        self.stream.wait_stream(torch.cuda.current_stream(eng.dev))   # producers of `inputs`
        with torch.cuda.stream(self.stream):
            if inputs is not None:
                eng.set_inputs(inputs['image'], inputs['future_image'], inputs.get('mask'))
                for v in (inputs['image'], inputs['future_image'], inputs.get('mask')):
                    if torch.is_tensor(v) and v.is_cuda:      # produced on another stream (the loader's): keep the
                        v.record_stream(self.stream)          # allocator from recycling it before this copy ran
            if self.use_graph:
                if self._graphs is None:
                    self._capture()
                self._graphs[0].launch()
                hook = self.after_backward
                if self.graph_resident:
                    pass                              # the collective and the optimizer are nodes of that graph
                elif self.split and self.native_comm is not None:
                    if hook is not None and self.buckets < 2:
                        hook(eng)
                    self._native_exchange(eng)
                elif self.split and self.buckets < 2:
                    if hook is not None:
                        hook(eng)
                    average_gradients(eng.grads, self.world_size, self.group, force=True); self._probe(eng.grads)
                    self._graphs[1].launch()
                elif self.split:
                    off = eng.bucket0_offset
                    w0 = average_gradients(eng.grads[off:], self.world_size, self.group, force=True, async_op=True)
                    self._graphs[1].launch()          # encoder backward overlaps the first bucket's all-reduce
                    if hook is not None:
                        hook(eng)                     # (the encoder bucket: the renderer's is already travelling)
                    w1 = average_gradients(eng.grads[:off], self.world_size, self.group, force=True, async_op=True)
                    for w in (w0, w1):
                        if w is not None:
                            w.wait()                  # stream-level wait (RCCL stream -> this stream), not a host sync
                    self._probe(eng.grads)            # (the process group's own stream is not ours to enqueue on)
                    self._graphs[2].launch()
                if self.debug_poison and not self.graph_resident and self.split:
                    self._poison()                    # after the optimizer graph: see __init__
            else:
                eng.forward(True)
                eng.backward()
                if self.after_backward is not None:
                    self.after_backward(eng)
                average_gradients(eng.grads, self.world_size, self.group)
                eng.optimizer_step()
        # the returned loss (a view of the engine's result buffer) is consumed on the CALLER's stream: order that stream after
        # this step (an event wait on the device, no host synchronisation)
        torch.cuda.current_stream(eng.dev).wait_stream(self.stream)
        return eng.loss

    def _native_exchange(self, eng):
        """graphs[0] has been launched on self.stream: the gradient exchange through imm_rccl_allreduce on self.comm_stream
        (ordered by events, no host wait), then the remaining graphs.  Two buckets: the renderer's gradients travel while
        the encoders' backward graph runs."""
        def on_comm(flat):
            ev = torch.cuda.Event(); ev.record(self.stream)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self.native_comm.all_reduce_sum(flat); self._probe(flat)
        if self.buckets < 2:
            on_comm(eng.grads)
        else:
            off = eng.bucket0_offset
            on_comm(eng.grads[off:])
            self._graphs[1].launch()              # encoder backward, concurrent with the first bucket's all-reduce
            on_comm(eng.grads[:off])
        done = torch.cuda.Event(); done.record(self.comm_stream)
        self.stream.wait_event(done)
        self._graphs[-1].launch()

    def forward_only(self, inputs=None):
        """The reference's `fwd_only` iteration (cnn_train_multi.py:447-449: session.run of the loss alone, "useful for timing"):
        forward + perceptual loss in training mode as ONE graph replay (captured on first use) — no backward pass, no gradient
        exchange, no update; returns the (device) scalar loss.  Forward-side state advances as in a training step (BN moving
        statistics, loss normalisers)."""
        eng = self.engine
        self.stream.wait_stream(torch.cuda.current_stream(eng.dev))
        with torch.cuda.stream(self.stream):
            if inputs is not None:
                eng.set_inputs(inputs['image'], inputs['future_image'], inputs.get('mask'))
                for v in (inputs['image'], inputs['future_image'], inputs.get('mask')):
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(self.stream)
            if self.use_graph:
                if getattr(self, '_fwd_graph', None) is None:
                    snap = eng.snapshot()
                    eng.forward(True)                     # warm-up outside capture, rolled back
                    self.stream.synchronize()
                    eng.restore(snap)
                    self.stream.synchronize()
                    g = ops.Graph()
                    g.capture_begin()
                    eng._training = True
                    eng.run(eng.prog_fwd)
                    g.capture_end()
                    self._fwd_graph = g
                eng._training = True
                self._fwd_graph.launch()
            else:
                eng.forward(True)
        torch.cuda.current_stream(eng.dev).wait_stream(self.stream)
        return eng.loss

    def measure_phases(self, steps=5):
        """Device-clock phases of a replayed step (makes a first multi-GPU run self-explaining): the step is
        re-captured with one-thread wall-clock probes at the program boundaries (IMM_DEBUG_STAMPS='marks' machinery), `steps`
        steps run — EVERY rank must call this (the steps contain the collectives) —, and the medians are returned in ms:
          forward, backward (first backward launch .. last one, incl. the filter gradients),
          exchange_exposed = optimizer start - backward end: the part of the gradient exchange (and of the host's hand-off
                             between the two graphs in 'pg' mode) that nothing hides,
          optimizer (clip + Adam + re-pack).
        The probes cost a few microseconds per step; the timed windows of bench.py run without them (the graphs captured here are
        dropped again)."""
        eng = self.engine
        old_mode, old_buf = eng._stamp_mode, eng._stamp_buf
        eng._stamp_mode = 'marks'
        if eng._stamp_buf is None:
            eng._stamp_buf = torch.zeros(8192, dtype=torch.int64, device=eng.dev)
        self._graphs = None
        rows = []
        try:
            for _ in range(max(1, steps)):
                self.step(None)
                self.synchronize()
                rep = eng.stamp_report()
                t = {}
                for us, _lane, label in rep:
                    t[label] = us                         # (a label repeats only across slices: the last occurrence counts)
                bwd_end = t.get('bwd2:end', t.get('bwd:end'))
                if None in (t.get('fwd:start'), t.get('fwd:end'), t.get('bwd:start'), bwd_end, t.get('opt:start'), t.get('opt:end')):
                    continue
                rows.append(((t['fwd:end'] - t['fwd:start']) / 1e3, (bwd_end - t['bwd:start']) / 1e3,
                             (t['opt:start'] - bwd_end) / 1e3, (t['opt:end'] - t['opt:start']) / 1e3))
        finally:
            eng._stamp_mode, eng._stamp_buf = old_mode, old_buf
            self._graphs = None
            self.synchronize()
        if not rows:
            return None
        med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(4)]
        return {'forward': round(med[0], 4), 'backward': round(med[1], 4), 'exchange_exposed': round(med[2], 4),
                'optimizer': round(med[3], 4), 'steps': len(rows)}

    def synchronize(self):
        self.stream.synchronize()


def setup_training(opts, model_factory, clip_value=None, use_graph=True):
    """cnn_train_multi.py:342-368.  `opts` keeps the reference keys: gpu_ids, batch_size, image_size.
    The rank/world come from the torchrun environment (one process per GPU)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert opts['batch_size'] % world == 0, 'Batch size must be divisible by number of GPUs'
    model = model_factory.create()
    return TrainStep(model, opts['batch_size'] // world, opts['image_size'], world_size=world, use_graph=use_graph)


class SummaryWriter(object):
    """Summaries of the train loop — the counterpart of the tf.summary.FileWriter of cnn_train_multi.py:436,447-452,489,
    504-508: every record goes to <log_dir>/summaries.jsonl (one JSON object per line) AND to a TensorBoard event file
    (imm_amd/utils/tf_events.py) as scalars `<tag>/<key>`; `images` ({name: HxWxC uint8}) become image summaries."""

    def __init__(self, log_dir):
        from ..utils.tf_events import EventFileWriter
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, 'summaries.jsonl')
        self._f = open(self.path, 'a')
        self.events = EventFileWriter(log_dir)

    def add_summary(self, record, step, images=None):
        rec = {'step': int(step), 'time': time.time()}
        rec.update(record)
        self._f.write(json.dumps(rec) + '\n')
        family = str(record.get('tag', 'train'))
        scalars = {}
        for k, v in record.items():
            if k == 'tag':
                continue
            if isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    scalars['%s/%s/%d' % (family, k, i)] = float(x)
            else:
                scalars['%s/%s' % (family, k)] = float(v)
        self.events.add_scalars(scalars, step, images={'%s/%s' % (family, k): v for k, v in (images or {}).items()})

    def flush(self):
        self._f.flush()
        self.events.flush()

    def close(self):
        self._f.close()
        self.events.close()


def image_summaries(eng, max_outputs=3):
    """The reference's image summaries (imm_model.py:456-468): inputs `im`, `future_im`, the clipped prediction
    `future_im_pred` and the colourised landmark maps `pose_embedding`; the first `max_outputs` samples side by side."""
    from ..models.imm_model import colorize_landmark_maps
    n = min(int(max_outputs), eng.B)
    S = eng.S

    def tile(x):           # [n,S,S,3] in 0..255 -> one S x n*S uint8 image
        x = x[:n].float().clamp(0, 255).to(torch.uint8).cpu().numpy()
        return x.transpose(1, 0, 2, 3).reshape(S, n * S, 3)

    full = torch.empty(eng.B, S, S, eng.K, device=eng.dev)
    ops.gauss_render_f32(eng.mu, eng.B, eng.K, eng.inv_std, S, full, eng.cfg.gauss_mode)
    maps = colorize_landmark_maps(full[:n])
    maps = maps / maps.amax().clamp_min(1e-12) * 255.0
    return {'im': tile(eng.in_image), 'future_im': tile(eng.in_future), 'future_im_pred': tile(eng.future_im_pred),
            'pose_embedding': tile(maps)}


def run_test_pass(model, test_dataset, step, writer=None, verbose=True):
    """cnn_train_multi.py:471-508: one pass over the test split in inference mode (BN moving averages), loss per batch;
    returns the sample-weighted mean loss.  `test_dataset` is re-iterated from its start each time (:478)."""
    total, count, it_n = 0.0, 0, 0
    for inputs in test_dataset:
        t0 = time.time()
        _, loss, _ = model.build(inputs, training_pl=False, build_loss=True)
        value = float(loss)
        n = int(inputs['image'].shape[0])
        total, count = total + value * n, count + n
        if verbose:
            dt = time.time() - t0
            print('test: step %d, loss = %.4f (%.1f examples/sec) %.3f sec/batch' % (step, value, n / dt, dt))
        it_n += 1
    if verbose:
        print('iteration through test set finished')
    mean = total / max(count, 1)
    if writer is not None and count:
        writer.add_summary({'tag': 'test', 'loss': mean, 'n_samples': count}, step)
        writer.flush()
    return mean


def train_loop(opts, train_step, data_iter, num_steps, log_every=10, checkpoint_fn=None, test_dataset=None, model=None,
               summary_writer=None, fwd_only=False):
    """cnn_train_multi.py:371-516 (session loop).  fwd_only (:378,447-449, "useful for timing"): every iteration evaluates the loss
    only — TrainStep.forward_only: forward + perceptual loss as one graph replay, no backward, no update, global_step does not
    move — for `num_steps - start` iterations, with the same examples/sec lines; no summaries, test passes or checkpoints.
    Steps run from the restored global step to num_steps (:441-444);
    NaN assert (:463); examples/sec (:466-469); every opts['n_summary'] steps a scalar summary (:447-452); every
    opts['n_test'] steps a pass over `test_dataset` in inference mode (:471-508); every opts['n_checkpoint'] steps
    `checkpoint_fn(step)` (:511-513, `step % n == 0` like the reference, so also at the first step).  Unlike the
    reference the loss is read back (a device sync) only on logging/summary steps, not every step."""
    t_start, n_seen = time.time(), 0
    rank = int(os.environ.get('RANK', '0'))
    eng = train_step.engine
    start_step = int(eng.step_count)
    n_summary = int(opts.get('n_summary') or 0)
    n_image = int(opts.get('n_image_summary', 100) or 0)      # image summaries: every n_image steps (upstream: every summary)
    n_test = int(opts.get('n_test') or 0)
    n_ckpt = int(opts.get('n_checkpoint') or 0)
    scaled = getattr(eng, 'loss_scale_state', None) is not None
    step = start_step
    last_event_step = None          # (loss scaling) the step whose events were evaluated last: a skipped update repeats its index
    while fwd_only and step < num_steps:
        t0 = time.time()
        loss = train_step.forward_only(next(data_iter))
        n_seen += opts['batch_size']
        if (step - start_step) % log_every == 0:
            train_step.synchronize()
            loss_value = mean_tower_loss(loss, train_step.world_size, train_step.group)
            assert loss_value == loss_value, 'Model diverged with loss = NaN'
            dt = time.time() - t0
            if rank == 0:
                print('step %d, loss = %.4f (%.1f examples/sec; %.3f sec/batch) [fwd_only]' % (step, loss_value, opts['batch_size'] / dt, dt))
        step += 1
    while step < num_steps:
        t0 = time.time()
        loss = train_step.step(next(data_iter))
        n_seen += opts['batch_size']
        do_log = (step - start_step) % log_every == 0
        do_sum = bool(n_summary) and step % n_summary == 0      # the same on every rank: the loss mean is a collective
        synced = False
        if scaled and (do_log or do_sum or (n_test and step % n_test == 0) or (n_ckpt and step % n_ckpt == 0)):
            # f16 storage with a dynamic loss scale: an update whose gradients overflowed is SKIPPED on the device, so between two
            # synchronisations the host's count can run AHEAD of global_step (never behind).  Whenever the host's count says an
            # event may fire, look at the device first and follow it (a summary / checkpoint named after a drifted count,
            # or fired twice / not at all).  `global_step - 1` = the index of the update this iteration applied; when this very
            # iteration was skipped it is the previous index, whose events are not repeated (no multiple of n there or already done).
            train_step.synchronize(); synced = True
            true_pre = int(eng.step_count) - 1
            if true_pre != step:
                # the host's count had run ahead: BOTH cadences are re-evaluated at the device's index (the log line
                # printed the rewound index on an off-cadence step, could repeat an index, and the collective ran for it)
                step = true_pre
                do_log = step >= 0 and (step - start_step) % log_every == 0
                do_sum = bool(n_summary) and step >= 0 and step % n_summary == 0
            if step == last_event_step or step < 0:
                do_log = do_sum = False       # this index's events have been evaluated (a skipped update repeats its index)
        fire = not scaled or (step >= 0 and step != last_event_step)
        if do_log or do_sum:
            train_step.synchronize(); synced = True
            loss_value = mean_tower_loss(loss, train_step.world_size, train_step.group)   # every rank takes part
            assert loss_value == loss_value, 'Model diverged with loss = NaN'
            dt = time.time() - t0
            if rank == 0 and do_log:
                print('step %d, loss = %.4f (%.1f examples/sec; %.3f sec/batch)' % (step, loss_value,
                                                                                    opts['batch_size'] / dt, dt))
            if rank == 0 and do_sum and summary_writer is not None:
                images = image_summaries(eng) if n_image and step % n_image == 0 else None
                rec = {'tag': 'train', 'loss': loss_value, 'lr': float(eng.lr_state[1]),
                       'loss_terms': [float(v) for v in eng.loss_terms], 'examples_per_sec': opts['batch_size'] / dt}
                # the reference's own scalar summaries: cost moving averages (base_model.py:52-60) and the activation scale of
                # every VGG16 layer (selfsup/vgg16.py:232-234)
                rec.update(eng.cost_summaries())
                rec.update(eng.vgg_activation_rms())
                summary_writer.add_summary(rec, step, images=images)
                summary_writer.flush()
        if fire and test_dataset is not None and model is not None and n_test and step % n_test == 0:
            train_step.synchronize(); synced = True
            if rank == 0:
                run_test_pass(model, test_dataset, step, summary_writer)
            if train_step.world_size > 1:
                torch.distributed.barrier(group=train_step.group)
        if fire and checkpoint_fn is not None and n_ckpt and step % n_ckpt == 0:
            train_step.synchronize(); synced = True
            if rank == 0:
                checkpoint_fn(step)
        if synced:
            last_event_step = step
        step += 1
        if scaled and (synced or step >= num_steps):
            # f16 storage with a dynamic loss scale: an update whose gradients overflowed is SKIPPED on the device (weights,
            # slots, global_step and Adam's t untouched; imm_clip_adam_step) — the host's count follows the device's
            # global_step, so checkpoints / summaries carry the saved global_step and the run ends with global_step ==
            # num_steps.  (The forward-side running averages — BN moving statistics, loss normalisers — do advance on a
            # skipped step: they belong to the forward pass, which ran.)  Read only where the loop synchronises anyway.
            train_step.synchronize()
            step = int(eng.step_count)
    train_step.synchronize()
    if rank == 0:
        print('Avg. samples per second %.2f' % (n_seen / max(time.time() - t_start, 1e-9)))
