"""Host-side batch producer with the reference's interface (reference data_loader.py:14-175): ``get_loader(hparams)``
returns an iterable of ``(melsp f32[B,192,80], spk_emb f32[B,82], pitch f32[B,192,1], len_org int64[B])``.

This is the step BEFORE the hot path; it is restated so that the reference's
``main.py`` finds the names it imports.  The collator reproduces the reference's intent -- random crop of
min_len_seq..max_len_seq frames, clip to [0,1], zero-pad mel / pad F0 with -1e10 -- without the stray
``pdb.set_trace()`` at data_loader.py:108.  ``SyntheticUtterances`` stands in when the corpus
(assets/spmel, assets/raptf0: download links only) is absent.
"""
import os
import pickle

import numpy as np
import torch
from torch.utils import data
from torch.utils.data.sampler import Sampler


class Utterances(data.Dataset):
    """train.pkl = list of [speaker, onehot82, relpath...] (make_metadata.py); loads the .npy files into RAM.

    As in the reference, ONE item per train.pkl entry, built from the entry's FIRST listed file (reference data_loader.py:63-64,72
    reads sbmt[2] only; `len(dataset)` is the number of speakers, :88-90).  all_utterances=True loads every listed file instead
    (one item per file) -- more data per epoch than the reference sees, so it is opt-in."""

    def __init__(self, root_dir, feat_dir, mode, all_utterances=False):
        self.root_dir, self.feat_dir, self.mode = root_dir, feat_dir, mode
        meta = pickle.load(open(os.path.join(root_dir, 'train.pkl'), 'rb'))
        self.items = []
        for sbmt in meta:
            spk, emb = sbmt[0], sbmt[1]
            for rel in (sbmt[2:] if all_utterances else sbmt[2:3]):
                sp = np.load(os.path.join(root_dir, rel))
                f0 = np.load(os.path.join(feat_dir, rel))
                self.items.append((sp, np.asarray(emb, np.float32), f0))
        self.num_tokens = len(self.items)

    def __getitem__(self, index):
        return self.items[index]

    def __len__(self):
        return self.num_tokens


class SyntheticUtterances(data.Dataset):
    """Seeded stand-in corpus: `n` utterances of 150-400 frames, mel in [0,1], normalised log-F0 in [0,1] with
    ~40 % unvoiced frames (-1e10), one-hot speaker embedding over 82."""

    def __init__(self, n=64, seed=0):
        rs = np.random.RandomState(seed)
        self.items = []
        for _ in range(n):
            L = int(rs.randint(150, 400))
            mel = rs.rand(L, 80).astype(np.float32)
            f0 = rs.rand(L).astype(np.float32)
            f0[rs.rand(L) < 0.4] = -1e10
            emb = np.zeros(82, np.float32)
            emb[rs.randint(0, 82)] = 1
            self.items.append((mel, emb, f0))

    def __getitem__(self, i):
        return self.items[i]

    def __len__(self):
        return len(self.items)


class MyCollator(object):
    def __init__(self, hparams):
        self.min_len_seq, self.max_len_seq, self.max_len_pad = hparams.min_len_seq, hparams.max_len_seq, hparams.max_len_pad

    def __call__(self, batch):
        mels, embs, f0s, lens = [], [], [], []
        for mel, emb, f0 in batch:
            # the reference draws TWO values each time and uses the first (size=2, data_loader.py:106-107): same generator
            # consumption here, so the same numpy seed gives the same crops.  (An utterance shorter than the crop makes the
            # reference raise in randint(0, <= 0); here it is taken whole.)
            n = int(np.random.randint(self.min_len_seq, self.max_len_seq + 1, size=2)[0])          # data_loader.py:106
            n = min(n, self.max_len_pad)
            left = int(np.random.randint(0, max(len(mel) - n, 1), size=2)[0])                       # :107
            a = np.clip(mel[left:left + n], 0, 1)                                        # :110-113
            c = f0[left:left + n]
            mels.append(np.pad(a, ((0, self.max_len_pad - a.shape[0]), (0, 0)), 'constant'))           # :115
            f0s.append(np.pad(c[:, None], ((0, self.max_len_pad - c.shape[0]), (0, 0)), 'constant',
                              constant_values=-1e10))                                                  # :116
            embs.append(emb)
            lens.append(a.shape[0])
        return (torch.from_numpy(np.stack(mels).astype(np.float32)), torch.from_numpy(np.stack(embs)),
                torch.from_numpy(np.stack(f0s).astype(np.float32)), torch.from_numpy(np.asarray(lens, np.int64)))


class MultiSampler(Sampler):
    """Each index `n_repeats` times per pass, optionally shuffled (data_loader.py:133-151)."""

    def __init__(self, num_samples, n_repeats, shuffle=False):
        self.num_samples, self.n_repeats, self.shuffle = num_samples, n_repeats, shuffle
        self.sample_idx_array = None

    def gen_sample_array(self):
        arr = torch.arange(self.num_samples, dtype=torch.int64).repeat(self.n_repeats)
        if self.shuffle:
            arr = arr[torch.randperm(len(arr))]
        self.sample_idx_array = arr
        return arr

    def __iter__(self):
        return iter(self.gen_sample_array())

    def __len__(self):
        return self.num_samples * self.n_repeats


class DeviceCorpus(object):
    """The whole corpus resident in HBM (the reference's ``Utterances`` keeps it in host RAM): mel [rows, 80] and F0 [rows]
    of all utterances back to back, start rows, and the speaker-embedding table.  A few GB at most against 288 GB."""

    def __init__(self, dataset, device='cuda'):
        items = [dataset[i] for i in range(len(dataset))]
        lens = np.asarray([len(it[0]) for it in items], np.int64)
        self.lens = lens
        self.starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
        self.mel = torch.from_numpy(np.concatenate([np.asarray(it[0], np.float32) for it in items])).to(device)
        self.f0 = torch.from_numpy(np.concatenate([np.asarray(it[2], np.float32).reshape(-1) for it in items])).to(device)
        self.emb = torch.from_numpy(np.stack([np.asarray(it[1], np.float32) for it in items])).to(device)
        self.device = self.mel.device

    def __len__(self):
        return len(self.lens)


class DeviceBatcher(object):
    """Iterable with the loader's contract -- (melsp [B,192,80], spk_emb [B,82], pitch [B,192,1], len_org int64[B]) -- whose
    batches are assembled ON the GPU from a DeviceCorpus by one kernel (ss_collate).  Per batch the host draws the crops
    with exactly the collator's generator calls (two ``np.random.randint(..., size=2)`` per utterance, data_loader.py:106-107) and sends
    3*B integers; no mel / F0 crosses PCIe.  Same seeds + same sampler order => the same batches as MyCollator, bit for bit."""

    def __init__(self, hparams, corpus, sampler=None, drop_last=True, rank=0, world=1):
        """rank / world: data parallel with one batcher per rank (same seeds on every rank).  Every rank walks the same sampler
        order and draws the crops of the whole global batch -- the generator stays aligned with a single-process run -- but
        assembles, i.e. launches ss_collate for, only its own batch_size / world utterances (`per_rank`: Solver does not slice
        these batches again)."""
        self.hp, self.corpus = hparams, corpus
        self.sampler = sampler or MultiSampler(len(corpus), hparams.samplier, shuffle=hparams.shuffle)
        self.B = hparams.batch_size
        self.drop_last = drop_last
        self.rank, self.world = rank, world
        self.per_rank = world > 1
        if self.B % world:
            raise ValueError('batch_size must divide over the ranks')

    def __len__(self):
        return len(self.sampler) // self.B

    def assemble(self, indices):
        from . import _capi
        hp, c = self.hp, self.corpus
        row0, lens = [], []
        for i in indices:
            n = int(np.random.randint(hp.min_len_seq, hp.max_len_seq + 1, size=2)[0])      # the collator's generator calls (size=2)
            n = min(n, hp.max_len_pad)
            left = int(np.random.randint(0, max(int(c.lens[i]) - n, 1), size=2)[0])
            lens.append(min(n, int(c.lens[i]) - left))
            row0.append(int(c.starts[i]) + left)
        if self.world > 1 and len(indices) == self.B:      # this rank's shard of the global batch
            lo, hi = self.rank * (self.B // self.world), (self.rank + 1) * (self.B // self.world)
            row0, lens, indices = row0[lo:hi], lens[lo:hi], indices[lo:hi]
        B, T, dev = len(indices), hp.max_len_pad, c.device
        meta = torch.tensor([row0, lens, list(map(int, indices))], dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        row0_d, len_d, item_d = meta[0].contiguous(), meta[1].to(torch.int32), meta[2].to(torch.int32)
        mel = torch.empty(B, T, c.mel.shape[1], device=dev)
        f0 = torch.empty(B, T, 1, device=dev)
        emb = torch.empty(B, c.emb.shape[1], device=dev)
        P = lambda t: _capi.C.c_void_p(t.data_ptr())
        _capi.check(_capi.lib().ss_collate(P(c.mel), P(c.f0), P(c.emb), P(row0_d), P(len_d), P(item_d), B, T, c.mel.shape[1],
                                           c.emb.shape[1], P(mel), P(f0), P(emb), _capi.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return mel, emb, f0, meta[1]

    def __iter__(self):
        batch = []
        for idx in self.sampler:
            batch.append(int(idx))
            if len(batch) == self.B:
                yield self.assemble(batch)
                batch = []
        if batch and not self.drop_last:
            yield self.assemble(batch)


def get_device_loader(hparams, dataset=None, device='cuda', rank=0, world=1):
    """Like get_loader, but the corpus lives in HBM and batches are assembled there (DeviceBatcher); rank / world: each rank
    assembles its own shard."""
    if dataset is None:
        if os.path.exists(os.path.join(hparams.root_dir, 'train.pkl')):
            dataset = Utterances(hparams.root_dir, hparams.feat_dir, hparams.mode)
        else:
            print(f'[speechsplit_amd] {hparams.root_dir}/train.pkl not found: using the synthetic corpus')
            dataset = SyntheticUtterances(max(4 * hparams.batch_size, 64))
    return DeviceBatcher(hparams, DeviceCorpus(dataset, device), rank=rank, world=world)


def get_loader(hparams, dataset=None):
    """data_loader.py:156-175.  Falls back to the synthetic corpus when hparams.root_dir has no train.pkl."""
    if dataset is None:
        if os.path.exists(os.path.join(hparams.root_dir, 'train.pkl')):
            dataset = Utterances(hparams.root_dir, hparams.feat_dir, hparams.mode)
        else:
            print(f'[speechsplit_amd] {hparams.root_dir}/train.pkl not found: using the synthetic corpus')
            dataset = SyntheticUtterances(max(4 * hparams.batch_size, 64))
    sampler = MultiSampler(len(dataset), hparams.samplier, shuffle=hparams.shuffle)
    # data_loader.py:165: without it every worker process inherits the same numpy state and produces identical crops
    worker_init_fn = lambda x: np.random.seed((torch.initial_seed()) % (2 ** 32))
    return data.DataLoader(dataset=dataset, batch_size=hparams.batch_size, sampler=sampler,
                           num_workers=hparams.num_workers, drop_last=True, pin_memory=torch.cuda.is_available(),
                           worker_init_fn=worker_init_fn, collate_fn=MyCollator(hparams))
