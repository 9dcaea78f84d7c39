#!/usr/bin/env python3
"""Host-side cost of enqueuing one training step (no device sync in between) vs. the device time of the step."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as W
from oracle.gen_fixtures import synth_batch
from speechsplit_amd import engine as E
B, T = 64, 128
hp = W.default_hparams(max_len_pad=T)
mel, f0, emb, lens = [t.cuda() for t in synth_batch(1, B, T, 64)]
sc, ls = E.draw_interp(B, 4, hp)
sc, ls = sc.cuda(), ls.cuda()
eng = E.Engine('G3', hp, B, T)
eng.load_weights(W.make_weights('G3', hp, 0))
for rep in range(2):
    for _ in range(5):
        eng.g3_train_step(mel, f0, emb, lens, (sc, ls))
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter()
        eng.g3_train_step(mel, f0, emb, lens, (sc, ls))
        host.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'run {rep}: host enqueue per step: first {host[0] * 1e3:.2f} ms, median {sorted(host)[10] * 1e3:.2f} ms, last {host[-1] * 1e3:.2f} ms; '
          f'enqueue of 20 steps {1e3 * (t1 - t0):.1f} ms, device done after {1e3 * (t2 - t0):.1f} ms ({1e3 * (t2 - t0) / 20:.2f} ms/step)', flush=True)
