#!/usr/bin/env python3
"""Gradient arena of one training step (no Adam) under two ss_tune settings: max |a - b| / max |a| per registered tensor.
usage: ab_grads.py <batch> <f32|bf16> key=value [key=value ...]   (the step runs once with the defaults, once with the given knobs)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import hparams as HP, model as M
from speechsplit_amd.engine import Engine, draw_interp, tune
B, prec = int(sys.argv[1]), sys.argv[2]
knobs = [kv.split('=') for kv in sys.argv[3:]]
T = 128
hp = HP.default_hparams(max_len_pad=T, batch_size=B)
dev = torch.device('cuda:0')
eng = Engine('G3', hp, B, T, device=dev)
eng.load_weights(M.init_weights('G3', hp, 0))
eng.set_precision(prec)
g = torch.Generator().manual_seed(5)
mel = torch.rand(B, T, 80, generator=g).to(dev)
f0 = torch.rand(B, T, 1, generator=g).to(dev)
emb = torch.zeros(B, hp.dim_spk_emb)
emb[torch.arange(B), torch.arange(B) % hp.dim_spk_emb] = 1
emb = emb.to(dev)
lens = torch.full((B,), T, dtype=torch.int32).to(dev)
draw = tuple(t.to(dev) for t in draw_interp(B, 4, hp))
res = []
for rep in range(2):
    if rep:
        for k, v in knobs:
            tune(k, int(v))
    loss = eng.g3_train_step(mel, f0, emb, lens, draw, no_adam=True)
    torch.cuda.synchronize()
    eng.check()
    res.append((float(loss), {n: t.clone() for n, t in eng.grad_views().items()}))
worst = 0.0
for n in res[0][1]:
    a, b = res[0][1][n].double(), res[1][1][n].double()
    r = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
    worst = max(worst, r)
    if r > 1e-6:
        print(f'{n:48s} {r:.3e}')
print(f'loss {res[0][0]:.6f} vs {res[1][0]:.6f}; worst tensor difference {worst:.3e}')
