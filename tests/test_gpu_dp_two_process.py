"""Two data-parallel ranks with the REAL HIP engine on one GPU (pytest -m gpu).

RCCL refuses two ranks on one device, so the collective here is gloo over host copies of the gradient buckets; everything else is the
production path in two separate processes: rank-sliced draws (dist.shard_draws), the split backward (decoder + head range final before
the encoder backward is enqueued), dist.bucket_plan's decoder-first order with the status slot riding in the first bucket, Adam with the
1/world mean folded in.  Result: both ranks end with the parameters a single process gets from the same step at the global batch.
Batches are small (4 utterances per rank) so that both ranks' persistent recurrence grids (64 workgroups each) are co-resident."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import weights as W
    from oracle.gen_fixtures import draws_for, synth_batch
    from speechsplit_amd import dist as D
    from speechsplit_amd.engine import Engine
    dist.init_process_group('gloo', rank=rank, world_size=world)
    Bg, T = 8, 128
    hp = W.default_hparams(max_len_pad=T)
    w = W.make_weights('G3', hp, 3)
    mel, f0, emb, lens = synth_batch(33, Bg, T, 64)
    steps = []
    for it in range(2):
        dr = draws_for(43 + it, Bg, 4)
        steps.append((torch.from_numpy(np.stack([d[0] for d in dr])), torch.from_numpy(np.stack([d[1] for d in dr]))))
    eng = Engine('G3', hp, Bg // world, T, device='cuda:0')
    eng.load_weights(w)
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    melr, embr, f0r, lenr = D.shard_batch((mel, emb, f0, lens), rank, world)
    k = eng.grad_split
    plan = D.bucket_plan(eng.grads.numel(), k)
    losses = []
    for sc, ls in steps:
        d = D.shard_draws(sc, ls, Bg, rank, world)
        eng.g3_train_step(melr, f0r, embr, lenr, d, no_adam=True, split_backward=True)      # decoder + head range is final here
        torch.cuda.synchronize()
        assert float(eng.grads[:k].abs().max()) == 0.0                                    # the encoder's has not been produced yet
        lo, hi = plan[0]
        g1 = eng.grads[lo:hi].cpu()
        h1 = dist.all_reduce(g1, async_op=True)                                           # first bucket in flight ...
        eng.train_finish(no_adam=True)                                                    # ... while the encoder backward runs
        torch.cuda.synchronize()
        g2 = eng.grads[plan[1][0]:plan[1][1]].cpu()
        h2 = dist.all_reduce(g2, async_op=True)
        h1.wait()
        h2.wait()
        assert float(g1[-4]) == 0.0                                                       # status slot: nobody aborted
        eng.grads[lo:hi].copy_(g1)
        eng.grads[plan[1][0]:plan[1][1]].copy_(g2)
        eng.adam_step(1.0 / world)
        lt = eng.loss.cpu().clone()
        dist.all_reduce(lt)
        losses.append(float(lt) / world)
    eng.check()
    params = eng.params.cpu()
    both = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(both, params)
    if rank == 0:
        ref = Engine('G3', hp, Bg, T, device='cuda:0')                                    # the single-process run at the global batch
        ref.load_weights(w)
        ref.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
        rl = [float(ref.g3_train_step(mel, f0, emb, lens, d)) for d in steps]
        ref.check()
        rp = ref.params.cpu()
        amax = float(rp.abs().max())
        q.put((float((both[0] - both[1]).abs().max()), float((params - rp).abs().max()) / amax, losses, rl))
    dist.barrier()
    dist.destroy_process_group()


def _lockstep_worker(rank, world, port, q):
    """Rank 1's first step aborts (its persistent recurrence's bounded wait is made to expire).  In lockstep mode neither rank refuses
    the following step: both keep exchanging buckets, both skip the updates on the device, both learn of the failure at the common
    check(), clear it there and go on together."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import weights as W
    from oracle.gen_fixtures import draws_for, synth_batch
    from speechsplit_amd import dist as D, engine as E
    dist.init_process_group('gloo', rank=rank, world_size=world)
    Bg, T = 8, 128
    hp = W.default_hparams(max_len_pad=T)
    w = W.make_weights('G3', hp, 3)
    mel, f0, emb, lens = synth_batch(33, Bg, T, 64)
    eng = E.Engine('G3', hp, Bg // world, T, device='cuda:0')
    eng.load_weights(w)
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    eng.set_lockstep(True)
    melr, embr, f0r, lenr = D.shard_batch((mel, emb, f0, lens), rank, world)
    k = eng.grad_split
    plan = D.bucket_plan(eng.grads.numel(), k)
    p0 = eng.params.clone()

    def step(it):
        dr = draws_for(43 + it, Bg, 4)
        sc, ls = torch.from_numpy(np.stack([d[0] for d in dr])), torch.from_numpy(np.stack([d[1] for d in dr]))
        eng.g3_train_step(melr, f0r, embr, lenr, D.shard_draws(sc, ls, Bg, rank, world), no_adam=True)      # never refuses in lockstep mode
        torch.cuda.synchronize()
        for lo, hi in plan:
            g = eng.grads[lo:hi].cpu()
            dist.all_reduce(g)
            eng.grads[lo:hi].copy_(g)
        eng.adam_step(1.0 / world)
        return (sc, ls)

    if rank == 1:
        E.tune('seq_spin_log2', 0)
    step(0)
    if rank == 1:
        E.tune('seq_spin_log2', 18)
    step(1)                                                    # enqueued by BOTH ranks although a status word is set on both by now
    torch.cuda.synchronize()
    skipped = torch.equal(eng.params, p0)
    status = eng.status()
    raised = False
    try:
        eng.check()
    except RuntimeError:
        raised = True
    eng.clear_abort()
    d2 = step(2)
    eng.check()
    params = eng.params.cpu()
    both = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(both, params)
    flags = torch.tensor([int(skipped), int(raised), status])
    allf = [torch.zeros_like(flags) for _ in range(world)]
    dist.all_gather(allf, flags)
    if rank == 0:
        ref = E.Engine('G3', hp, Bg, T, device='cuda:0')      # a single process that only ever saw the third step
        ref.load_weights(w)
        ref.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
        ref.g3_train_step(mel, f0, emb, lens, d2)
        ref.check()
        rp = ref.params.cpu()
        q.put((float((both[0] - both[1]).abs().max()), float((params - rp).abs().max()) / float(rp.abs().max()), [f.tolist() for f in allf]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_stay_in_lockstep_through_an_abort():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lockstep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    spread, err, flags = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert flags[0][:2] == [1, 1] and flags[1][:2] == [1, 1], flags          # both skipped the two updates, both checks raised
    assert flags[1][2] & 1 and flags[0][2] & 2, flags                       # rank 1: its own abort; rank 0: told through the status slot
    assert spread == 0.0 and err <= 2e-4, (spread, err)


def test_two_rank_engine_step_equals_single_process_step():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    spread, err, losses, ref_losses = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert spread == 0.0                                   # replicas stay bit-identical: same reduced gradients, same update
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-5 * abs(b), (losses, ref_losses)
    assert err <= 2e-4, err                                # two Adam steps of lr 1e-4; summation order differs -> fp32 tolerance, no lr-sized outliers
