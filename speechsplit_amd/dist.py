"""Data-parallel plumbing for the training step: one process per GPU, utterances sharded
across ranks, ONE all-reduce over the flat gradient arena per step (RCCL over xGMI on the GPU box; gloo in the CPU
tests), replicated Adam.  Every op on the path is per-utterance (GroupNorm is per-sample, the LSTMs carry no
cross-utterance state), so the only coupling is the mean in the loss: with equal shards, the global gradient is
the average of the rank-local gradients.

The reference has no distributed code at all (single device, solver.py:38); the equivalence target is its
single-process result at the global batch, given the same resampling draws: rank r consumes the slice of the
global ``rand(B*7)`` / ``randint(B*7)`` streams that belongs to its utterances.
"""
import os

import torch


def world_info():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    import torch.distributed as dist
    rank, local, world = world_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = device if device is not None else torch.device(f'cuda:{local}')
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_range(n, rank, world):
    """Utterances [lo, hi) of a global batch of n owned by `rank` (equal shards; n must divide)."""
    if n % world:
        raise ValueError(f'global batch {n} is not divisible by world size {world}')
    per = n // world
    return rank * per, (rank + 1) * per


def shard_batch(batch, rank, world):
    """Slice every tensor of a collated batch (mel, emb, f0, len) along dim 0."""
    lo, hi = shard_range(batch[0].shape[0], rank, world)
    return tuple(t[lo:hi] for t in batch)


def shard_draws(scales, len_seg, global_batch, rank, world):
    """scales / len_seg: [ncalls, global_batch * S] drawn once for the whole batch -> this rank's [ncalls, local * S]."""
    lo, hi = shard_range(global_batch, rank, world)
    S = scales.shape[-1] // global_batch
    return scales[..., lo * S:hi * S].contiguous(), len_seg[..., lo * S:hi * S].contiguous()


def allreduce_mean_(flat_grads, world, group=None):
    """Sum-all-reduce the flat gradient arena in place and return the scale (1/world) the optimiser must apply.
    (The engine's Adam kernel takes the scale as an argument, so no extra pass over the 78 MB arena is needed.)"""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world
