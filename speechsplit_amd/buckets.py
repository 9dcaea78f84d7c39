"""Length-bucketed batches (BASELINE.json config 5: "variable-length (96-192 frames)").

The reference has no bucketing: its collator crops 64..128 frames and pads EVERY utterance to hparams.max_len_pad = 192
(reference data_loader.py:106-116), and its BLSTMs and its loss run over the padding (model.py:81,214-215,251; solver.py:166) --
so a kernel that skipped pad frames would change the result.  What the reference does offer is the
hyper-parameter: with max_len_pad = T it pads to T.  A length bucket is exactly that: all utterances of a batch whose crop
length falls in (T - 8, T] are padded to T (T a multiple of the code down-sampling factor 8, model.py:87,223-227), and the step
runs with max_len_pad = T (SS_STEP_BUCKET in include/speechsplit_amd.h) -- the reference's arithmetic for that T, 8 frames of
padding at most instead of up to 96.

Scheduling.  On this engine a step's time is dominated by the decoder recurrences, whose cost is proportional to T and
almost independent of the batch size up to 64 utterances per GPU.  Cutting one rank's 64 utterances into per-length sub-batches
would therefore multiply the step time; the schedule that pays is the classic bucketed sampler: EVERY global batch comes from
one bucket, all ranks run the same T in lock step (balanced by construction: equal utterance count, frame totals within
8 * B / world), and consecutive steps visit different buckets.  `dist.balanced_shards` evens out the residual length
differences inside a bucket when one host collates for all ranks.
"""
import numpy as np
import torch

STEP = 8          # bucket granularity = code down-sampling factor (T % 8 == 0 is a hard requirement of the model)


def bucket_of(n, lo=96, hi=192, step=STEP):
    """Smallest multiple of `step` that holds n frames, clamped to [lo, hi]."""
    return int(min(hi, max(lo, -(-int(n) // step) * step)))


def buckets(lo=96, hi=192, step=STEP):
    return list(range(lo, hi + 1, step))


class BucketedDeviceBatcher(object):
    """Loader-contract iterable -- (melsp [B,T,80], spk_emb [B,82], pitch [B,T,1], len_org int64[B]) -- over a DeviceCorpus whose
    batches each come from ONE length bucket T in {lo, lo+8, .., hi}; crops are min_len..max_len frames (config 5: 96..192),
    assembled on the GPU by ss_collate.  Data parallel: construct it with (rank, world) on every rank with the same seed -- the
    bucket sequence comes from a generator all ranks share, the utterances and crops from a per-rank one, and each rank
    assembles only its own batch_size / world utterances (`per_rank`: Solver does not slice these batches again)."""
    per_rank = True

    def __init__(self, hparams, corpus, min_len=96, max_len=192, rank=0, world=1, seed=0, steps_per_epoch=None):
        self.hp, self.corpus = hparams, corpus
        self.lo, self.hi = bucket_of(min_len, min_len, max_len), bucket_of(max_len, min_len, max_len)
        self.min_len, self.max_len = int(min_len), int(max_len)
        if hparams.batch_size % world:
            raise ValueError('batch_size must divide over the ranks')
        self.B = hparams.batch_size // world
        self.rank, self.world = rank, world
        self.shared = np.random.RandomState(seed)                  # same stream on every rank: which bucket comes next
        self.local = np.random.RandomState(seed * 1000003 + 17 * rank + 1)
        self.steps = steps_per_epoch or max(1, len(corpus) * hparams.samplier // hparams.batch_size)
        ok = np.nonzero(np.asarray(corpus.lens) > self.max_len)[0]
        if len(ok) == 0:
            raise ValueError('no utterance is longer than max_len')
        self.long_items = ok

    def __len__(self):
        return self.steps

    def next_bucket(self):
        """Crop lengths are uniform over min_len..max_len, so a bucket's share is the number of lengths it holds."""
        n = int(self.shared.randint(self.min_len, self.max_len + 1))
        return bucket_of(n, self.lo, self.hi)

    def assemble(self, T):
        from . import _capi
        c = self.corpus
        lo_n = max(self.min_len, T - STEP + 1)
        items = self.local.choice(self.long_items, size=self.B)
        lens = self.local.randint(lo_n, min(T, self.max_len) + 1, size=self.B)
        left = np.array([self.local.randint(0, int(c.lens[i]) - int(n)) for i, n in zip(items, lens)])
        row0 = c.starts[items] + left
        dev = c.device
        meta = torch.from_numpy(np.stack([row0, lens.astype(np.int64), items.astype(np.int64)])).pin_memory().to(dev, non_blocking=True)
        row0_d, len_d, item_d = meta[0].contiguous(), meta[1].to(torch.int32), meta[2].to(torch.int32)
        mel = torch.empty(self.B, T, c.mel.shape[1], device=dev)
        f0 = torch.empty(self.B, T, 1, device=dev)
        emb = torch.empty(self.B, c.emb.shape[1], device=dev)
        P = lambda t: _capi.C.c_void_p(t.data_ptr())
        _capi.check(_capi.lib().ss_collate(P(c.mel), P(c.f0), P(c.emb), P(row0_d), P(len_d), P(item_d), self.B, T, c.mel.shape[1],
                                           c.emb.shape[1], P(mel), P(f0), P(emb),
                                           _capi.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return mel, emb, f0, meta[1]

    def __iter__(self):
        for _ in range(self.steps):
            yield self.assemble(self.next_bucket())
