#!/usr/bin/env python3
"""Cost of the data-parallel schedules themselves on ONE GPU (world 1: the all-reduces move no bytes), next to the
fused step: what the stream / hardware-queue interplay of each schedule costs before any communication."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
from oracle import weights as W
from oracle.gen_fixtures import synth_batch
from speechsplit_amd import engine as E
B, T = 64, 128
hp = W.default_hparams(max_len_pad=T)
mel, f0, emb, lens = [t.cuda() for t in synth_batch(1, B, T, 64)]
sc, ls = E.draw_interp(B, 4, hp)
sc, ls = sc.cuda(), ls.cuda()


def timed(label, fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    print(f'{label:64s}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step', flush=True)


eng = E.Engine('G3', hp, B, T)
eng.load_weights(W.make_weights('G3', hp, 0))
timed('fused step, before RCCL is initialised', lambda: eng.g3_train_step(mel, f0, emb, lens, (sc, ls)))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29534')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
dist.all_reduce(torch.zeros(1024, device='cuda'))
torch.cuda.synchronize()
timed('fused step, communicator up', lambda: eng.g3_train_step(mel, f0, emb, lens, (sc, ls)))
for schedule in ('after', 'overlap', 'join'):
    timed(f"dp_train_step(schedule='{schedule}'), world 1", lambda: eng.dp_train_step(mel, f0, emb, lens, (sc, ls), 1, schedule=schedule))
timed('split step + finish, no collectives (engine streams joined)',
      lambda: (eng.g3_train_step(mel, f0, emb, lens, (sc, ls), no_adam=True, split_backward=True), eng.train_finish(no_adam=True), eng.adam_step(1.0)))
g, k = eng.grads, eng.grad_split
timed('the two all-reduces alone', lambda: (dist.all_reduce(g[k:]), dist.all_reduce(g[:k])))
dist.destroy_process_group()
