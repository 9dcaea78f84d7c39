#!/usr/bin/env python3
"""Interleaved A/B of ss_tune settings on the headline training step, ONE process, ONE engine: box-to-box and run-to-run spread
(0.1-0.2 ms) is larger than most of the effects that are left, so the settings alternate round by round on the same engine and
the median per setting is reported.

    python tools/ab_inproc.py [--rounds 7] [--steps 20] [--batch 64] [--frames 128] [--model G3] --base "k=v k=v" "k=v ..." "k=v ..."

--base lists the default value of every knob that any setting changes (it is re-applied before each setting); "-" is the base itself.
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--model', default='G3')
    ap.add_argument('--precision', default='f32')
    ap.add_argument('--lr', type=float, default=1e-4, help='what-if runs with wrong gradients: a tiny rate keeps the weights sane')
    ap.add_argument('--base', default='')
    ap.add_argument('settings', nargs='+')
    a = ap.parse_args()
    from bench import synth
    from speechsplit_amd import hparams as HP, model as M
    from speechsplit_amd.engine import Engine, draw_interp, tune
    dev = torch.device('cuda:0')
    B, T, kind = a.batch, a.frames, a.model
    hp = HP.default_hparams(max_len_pad=T, batch_size=B)
    eng = Engine(kind, hp, B, T, device=dev)
    eng.load_weights(M.init_weights(kind, hp, 0))
    eng.set_adam(a.lr, 0.9, 0.999, 1e-8, 0)
    eng.set_precision(a.precision)
    mel, f0, emb, lens = synth(B, T, 1000, dev)
    ncalls = 4 if kind == 'G3' else 3
    if kind == 'G6':
        from speechsplit_amd.utils import quantize_f0_torch
        onehot, qidx = quantize_f0_torch(f0[:, :, 0].clone())
        onehot, qidx = onehot.contiguous(), qidx.to(torch.int32).contiguous()

    def step():
        d = draw_interp(B, ncalls, hp)
        if kind == 'G3':
            eng.g3_train_step(mel, f0, emb, lens, d)
        else:
            eng.g6_train_step(mel, onehot, qidx, d)

    def apply(s):
        for kv in s.split():
            if kv != '-':
                k, v = kv.split('=')
                tune(k, int(v))

    res = {s: [] for s in a.settings}
    for r in range(a.rounds + 1):
        order = a.settings if r % 2 == 0 else a.settings[::-1]
        for s in order:
            apply(a.base)
            apply(s)
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            if r:                                  # round 0 warms everything up
                res[s].append((time.perf_counter() - t0) / a.steps * 1e3)
    eng.check()
    for s in a.settings:
        v = sorted(res[s])
        print(f'{s:44s} median {statistics.median(v):.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}  ({len(v)} rounds x {a.steps} steps)', flush=True)


if __name__ == '__main__':
    main()
