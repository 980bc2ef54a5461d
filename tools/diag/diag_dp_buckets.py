"""Diagnosis: two ranks sharing one GPU (gloo), one vs two all-reduce buckets, against the single-process two-tower emulation.
Prints, per step, checksums of the flat gradient buffer before / after the exchange and of the parameters."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def rank_main(rank, world, port, buckets, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import test_dp_gpu as T
    from imm_amd.train import cnn_train_multi as M
    full, _t = T._towers()
    mine = M.split_inputs(full, world, rank)
    ts = M.TrainStep(T._model(world, buckets), T.B_GLOBAL // world, T.S_IMG, world_size=world, use_graph=True)
    eng = ts.engine
    log = []
    orig = M.average_gradients

    def spy(flat, *a, **k):
        torch.cuda.synchronize()
        before = float(flat.double().abs().sum())
        r = orig(flat, *a, **k)
        torch.cuda.synchronize()
        log.append((flat.numel(), before, float(flat.double().abs().sum())))
        return r
    M.average_gradients = spy
    for it in range(2):
        loss = ts.step(mine if it == 0 else None)
        ts.synchronize()
        log.append(('step', it, float(loss), float(eng.params.double().abs().sum()), int(eng.step_count)))
    ret[rank] = log
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    import torch.multiprocessing as mp
    import test_dp_gpu as T
    A, Bn, losses, gA0, gB0 = T.emulate_two_towers(2)
    print('EMU  grads0 |A| %.6f |B| %.6f |A+B| %.6f   params after 2 steps %.6f  losses %r' % (
        float(gA0.double().abs().sum()), float(gB0.double().abs().sum()), float((gA0 + gB0).double().abs().sum()),
        float(A.params.double().abs().sum()), losses))
    for buckets in (1, 2):
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(rank_main, args=(2, 29500 + 7 * buckets + os.getpid() % 100, buckets, ret), nprocs=2, join=True)
            for r in (0, 1):
                print('BUCKETS %d rank %d:' % (buckets, r))
                for e in ret[r]:
                    print('    ', e)
