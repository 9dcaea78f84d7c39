#!/usr/bin/env python3
"""How far the fp16 x 2 contractions move a training step: gradients with the fast paths on vs. off (bf16 x 3 everywhere),
per tensor, max-norm relative; and both against each other's loss."""
import json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as W
from oracle.gen_fixtures import draws_for, synth_batch
from speechsplit_amd import engine as E


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


B, T = 64, 128
hp = W.default_hparams(max_len_pad=T)
mel, f0, emb, lens = synth_batch(7, B, T, 64)
draws = draws_for(8, B, 4)
st = (np.stack([d[0] for d in draws]), np.stack([d[1] for d in draws]))
res = {}
for tag, fwd, bwd in (('bf16x3', 0, 0), ('fwd fp16x2', 1, 0), ('fwd+bwd fp16x2', 1, 1)):
    E.tune('fwd_f16x2', fwd)
    E.tune('bwd_f16x2', bwd)
    eng = E.Engine('G3', hp, B, T)
    eng.load_weights(W.make_weights('G3', hp, 0))
    loss = float(eng.g3_train_step(mel, f0, emb, lens, st, no_adam=True))
    res[tag] = (loss, {n: v.clone() for n, v in eng.grad_views().items()}, eng.debug_buffer('out', B, T))
    eng.check()
E.tune('fwd_f16x2', 1)
E.tune('bwd_f16x2', 1)
ref = res['bf16x3']
for tag in ('fwd fp16x2', 'fwd+bwd fp16x2'):
    l, g, o = res[tag]
    errs = sorted(((rel(g[n], ref[1][n]), n) for n in g), reverse=True)
    print(f'{tag:16s}: loss rel diff {abs(l - ref[0]) / ref[0]:.2e}, output rel diff {rel(o, ref[2]):.2e}, gradients: worst {errs[0][0]:.2e} ({errs[0][1]}), '
          f'median {errs[len(errs) // 2][0]:.2e}')
