#!/usr/bin/env python3
"""Soak test of the training step: thousands of steps at several batch sizes, the persistent kernels' bounded-wait
checked regularly (Engine.check), losses finite; reports steps/s.  usage: soak.py [steps]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as W
from oracle.gen_fixtures import synth_batch
from speechsplit_amd import engine as E
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
T = 128
hp = W.default_hparams(max_len_pad=T)
for B in (64, 48, 17, 5):
    eng = E.Engine('G3', hp, max(8, B), T)
    eng.load_weights(W.make_weights('G3', hp, 3))
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    mel, f0, emb, lens = [t.cuda() for t in synth_batch(B, B, T, 64)]
    t0 = time.perf_counter()
    n = steps if B == 64 else steps // 4
    for it in range(n):
        sc, ls = E.draw_interp(B, 4, hp)
        loss = eng.g3_train_step(mel, f0, emb, lens, (sc.cuda(), ls.cuda()))
        if it % 50 == 49:
            eng.check()
            lv = float(loss)
            assert lv == lv and lv < 1e6, (B, it, lv)
            if it % 500 == 499:
                print(f'B={B} step {it + 1}: loss {lv:.5f}, {(it + 1) / (time.perf_counter() - t0):.1f} steps/s', flush=True)
    eng.check()
    print(f'B={B}: {n} steps ok, final loss {float(loss):.5f}', flush=True)
    del eng
# the 16-bit data path and Generator_6 as well (round 4)
for kind, B, Tx, prec in (('G3', 64, 128, 'bf16'), ('G3', 32, 192, 'bf16'), ('G6', 32, 192, 'bf16'), ('G6', 32, 192, 'f32')):
    hpx = W.default_hparams(max_len_pad=Tx)
    eng = E.Engine(kind, hpx, B, Tx)
    eng.load_weights(W.make_weights(kind, hpx, 3))
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    eng.set_precision(prec)
    mel, f0, emb, lens = [t.cuda() for t in synth_batch(B, B, Tx, 64)]
    if kind == 'G6':
        from speechsplit_amd.utils import quantize_f0_torch
        onehot, qidx = quantize_f0_torch(f0[:, :, 0].clone())
        onehot, qidx = onehot.contiguous(), qidx.to(torch.int32).contiguous()
    n = steps // 4
    t0 = time.perf_counter()
    for it in range(n):
        sc, ls = E.draw_interp(B, 4 if kind == 'G3' else 3, hpx)
        d = (sc.cuda(), ls.cuda())
        loss = eng.g3_train_step(mel, f0, emb, lens, d) if kind == 'G3' else eng.g6_train_step(mel, onehot, qidx, d)
        if it % 50 == 49:
            eng.check()
            lv = float(loss)
            assert lv == lv and lv < 1e6, (kind, B, it, lv)
    eng.check()
    assert eng.scratch_fallbacks() == 0
    print(f'{kind} {B}x{Tx} {prec}: {n} steps ok, final loss {float(loss):.5f}, {n / (time.perf_counter() - t0):.1f} steps/s', flush=True)
    del eng
print('soak ok')
