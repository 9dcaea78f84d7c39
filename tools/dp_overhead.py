#!/usr/bin/env python3
"""Cost of the data-parallel schedule itself on one GPU: fused step vs. the split step with the two (one-rank) RCCL all-reduces."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
from oracle import weights as W
from oracle.gen_fixtures import synth_batch
from speechsplit_amd import engine as E
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
B, T = 64, 128
hp = W.default_hparams(max_len_pad=T)
mel, f0, emb, lens = [t.cuda() for t in synth_batch(1, B, T, 64)]
sc, ls = E.draw_interp(B, 4, hp)
sc, ls = sc.cuda(), ls.cuda()
eng = E.Engine('G3', hp, B, T)
eng.load_weights(W.make_weights('G3', hp, 0))
for name, fn in (('fused step', lambda: eng.g3_train_step(mel, f0, emb, lens, (sc, ls))),
                 ('data-parallel schedule, world 1', lambda: eng.dp_train_step(mel, f0, emb, lens, (sc, ls), 1))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    print(f'{name:34s}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step', flush=True)
dist.destroy_process_group()
