"""world_size-2 data-parallel semantics on CPU (gloo): even batch split -> per-rank (per-"tower") gradients
-> ONE sum all-reduce of the flat gradient buffer -> 1/N scale -> per-tensor clip -> Adam must equal the
reference's in-graph tower averaging (cnn_train_multi.py:66-106), restated by oracle.train_step with two
towers in one process.  The gradient arithmetic here is the oracle's (test infrastructure); what is under
test is the host-side DP logic shared with the GPU path: split_inputs, the flat layout (trainable_spec /
SegmentTable offsets) and average_gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import imm_oracle as O
    from imm_amd import engine as E, ops
    from imm_amd.train.cnn_train_multi import average_gradients, split_inputs
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(10)
    S_IMG = 64
    P, St = O.init_params(cfg, S_IMG)
    full = O.synthetic_inputs(2, S_IMG, seed=3)
    mine = split_inputs(full, world, rank)
    out, grads = O.loss_and_grads(P, St, mine, cfg)
    # flat buffer in the engine's layout
    spec = E.trainable_spec(Box(dict(cfg)), S_IMG)
    tab = ops.SegmentTable([int(torch.tensor(s).prod()) for _n, s, _w in spec], [w for _n, _s, w in spec], 'cpu')
    flat = torch.zeros(tab.total)
    for i, (name, shape, _wd) in enumerate(spec):
        flat[tab.offsets[i]:tab.offsets[i + 1]] = grads[name].reshape(-1)
    average_gradients(flat, world)                       # sum all-reduce (gloo here, RCCL on the GPUs)
    flat *= 1.0 / world                                   # grad_scale of imm_clip_adam_step
    g = {name: flat[tab.offsets[i]:tab.offsets[i + 1]].reshape(shape) for i, (name, shape, _wd) in enumerate(spec)}
    g = {k: O.clip_by_norm(v, 1.0) for k, v in g.items()}
    newP = O.adam_apply(P, g, O.new_adam_state(P), lr=O.learning_rate(0))
    loss = torch.tensor([float(out['loss'])])
    dist.all_reduce(loss)                                 # printed loss = tower mean (cnn_train_multi.py:173)
    if rank == 0:
        ret['params'] = {k: v.detach().clone() for k, v in newP.items()}
        ret['loss'] = float(loss) / world
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_equals_two_tower_reference():
    from oracle import imm_oracle as O
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    cfg = O.default_model_config(10)
    P, St = O.init_params(cfg, 64)
    full = O.synthetic_inputs(2, 64, seed=3)
    towers = [{k: v[i:i + 1] for k, v in full.items()} for i in range(2)]
    refP, _refS, info = O.train_step(P, St, O.new_adam_state(P), towers, cfg, clip=1.0, lr=O.learning_rate(0))
    got = ret['params']
    assert abs(ret['loss'] - info['loss']) / abs(info['loss']) < 1e-5
    worst = 0.0
    for k, v in refP.items():
        if k.endswith('/b'):
            continue      # noise gradients -> sign-of-noise Adam updates (DESIGN.md numerics)
        d_ref, d_got = (v - P[k]).flatten().double(), (got[k] - P[k]).flatten().double()
        cos = float((d_ref * d_got).sum() / (d_ref.norm() * d_got.norm() + 1e-30))
        worst = max(worst, 1 - cos)
    assert worst < 2e-2, worst     # same math, different summation order (threads / all-reduce)
