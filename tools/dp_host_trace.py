#!/usr/bin/env python3
"""Host time spent in each call of the data-parallel step (world 1): does any of them block the host?"""
import os, sys, time, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
from oracle import weights as W
from oracle.gen_fixtures import synth_batch
from speechsplit_amd import engine as E, _capi
B, T = 64, 128
hp = W.default_hparams(max_len_pad=T)
mel, f0, emb, lens = [t.cuda() for t in synth_batch(1, B, T, 64)]
sc, ls = E.draw_interp(B, 4, hp)
sc, ls = sc.cuda(), ls.cuda()
eng = E.Engine('G3', hp, B, T)
eng.load_weights(W.make_weights('G3', hp, 0))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29535')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
lens_i = lens.int()
cs = torch.cuda.Stream()
k = eng.grad_split
g = eng.grads
marks = {}


def step(trace):
    t = [time.perf_counter()]
    _capi.check(eng.lib.ss_g3_train_step(eng.h, E._ptr(mel), E._ptr(f0), E._ptr(emb), E._ptr(lens_i), E._ptr(sc), E._ptr(ls), B, T, 1.0, 7, E._ptr(eng.loss), E._stream()))
    t.append(time.perf_counter())
    _capi.check(eng.lib.ss_wait_decoder_grads(eng.h, ctypes.c_void_p(cs.cuda_stream)))
    t.append(time.perf_counter())
    eng.train_finish(no_adam=True)
    t.append(time.perf_counter())
    with torch.cuda.stream(cs):
        h1 = dist.all_reduce(g[k:], async_op=True)
    t.append(time.perf_counter())
    h2 = dist.all_reduce(g[:k], async_op=True)
    t.append(time.perf_counter())
    h1.wait()
    h2.wait()
    t.append(time.perf_counter())
    eng.adam_step(1.0)
    t.append(time.perf_counter())
    if trace is not None:
        trace.append([1e3 * (b - a) for a, b in zip(t, t[1:])])


for _ in range(5):
    step(None)
torch.cuda.synchronize()
tr = []
t0 = time.perf_counter()
for _ in range(20):
    step(tr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
names = ['split step', 'wait_decoder_grads', 'train_finish', 'all_reduce 1 (comm stream)', 'all_reduce 2', 'work.wait x2', 'adam_step']
for i, n in enumerate(names):
    col = sorted(r[i] for r in tr)
    print(f'{n:28s}: median {col[10]:.3f} ms, max {col[-1]:.3f} ms')
print(f'host loop {1e3 * (t1 - t0) / 20:.3f} ms/step, device done {1e3 * (t2 - t0) / 20:.3f} ms/step')
dist.destroy_process_group()
