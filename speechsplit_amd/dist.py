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


# ---------------------------------------------------------------------------------------------------------------------------------
# The gradient arena and its two buckets.  Backward produces the head + decoder gradients first (they are 80 % of the bytes), so
# their all-reduce is launched first and runs beside the encoder backward; the encoder bucket follows.  The arena's last four
# floats are the STATUS SLOT: a rank whose step is invalid (a persistent kernel gave up, include/speechsplit_amd.h) writes 1.0
# there, the sum reaches every rank inside the decoder bucket, and every rank's Adam kernel skips the update.
def arena_layout(kind, hp, max_batch=1):
    """[(name, offset, shape)], arena numel (status slot included) and the encoder | decoder split of the engine's flat arenas.
    Needs the C library but no GPU (ss_create touches no device memory)."""
    import ctypes as C
    from . import _capi
    lib = _capi.lib()
    hps = _capi.hparams_struct(hp)
    h = lib.ss_create({'G3': 3, 'G6': 6}[kind], C.byref(hps), max_batch, hp.max_len_pad)
    if not h:
        raise RuntimeError('speechsplit_amd: ' + lib.ss_last_error().decode())
    try:
        table = []
        name = C.create_string_buffer(256)
        off, nd, shp = C.c_long(), C.c_int(), (C.c_long * 3)()
        for i in range(lib.ss_num_params(h)):
            _capi.check(lib.ss_param_info(h, i, name, 256, C.byref(off), C.byref(nd), C.byref(shp)))
            table.append((name.value.decode(), off.value, tuple(shp[k] for k in range(nd.value))))
        return table, int(lib.ss_arena_numel(h)), int(lib.ss_grad_split(h))
    finally:
        lib.ss_destroy(h)


def bucket_plan(numel, split):
    """[(lo, hi)] in launch order: head + decoder (+ status slot) first, then the encoder."""
    return [(split, numel), (0, split)] if 0 < split < numel else [(0, numel)]


def reduce_bucket(flat, lo, hi, group=None, async_op=True):
    """Sum-all-reduce flat[lo:hi] in place; returns the work handle (None when not distributed)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def reduce_arena(flat, split, group=None):
    """All buckets of `bucket_plan`, launched in order; waits for them and returns the Adam grad_scale (1 / world)."""
    import torch.distributed as dist
    handles = [reduce_bucket(flat, lo, hi, group) for lo, hi in bucket_plan(flat.numel(), split)]
    for h in handles:
        if h is not None:
            h.wait()
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    return 1.0 / world


def balanced_shards(lengths, world):
    """Config 5 (BASELINE.json: variable-length batches): utterance indices per rank with equal COUNT and near-equal total
    frames.  Longest-first greedy into the rank with the fewest frames that still has room; deterministic (ties by index)."""
    n = len(lengths)
    if n % world:
        raise ValueError(f'{n} utterances do not divide over {world} ranks')
    per = n // world
    order = sorted(range(n), key=lambda i: (-int(lengths[i]), i))
    shards, load = [[] for _ in range(world)], [0] * world
    for i in order:
        r = min((r for r in range(world) if len(shards[r]) < per), key=lambda r: (load[r], r))
        shards[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(s) for s in shards]
