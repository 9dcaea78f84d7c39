#!/usr/bin/env python3
"""Host time of one training-step call (enqueue only) against the GPU time of the step: how far ahead of the GPU the host runs.
    python tools/host_time.py [--batch 64] [--frames 128] [--model G3] [--force-dp]"""
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
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--model', default='G3')
    ap.add_argument('--force-dp', action='store_true', help="the data-parallel step on the engine's own communicator at world 1")
    a = ap.parse_args()
    from bench import synth
    from speechsplit_amd import hparams as HP, model as M
    from speechsplit_amd.engine import Engine, draw_interp
    dev = torch.device('cuda:0')
    B, T, kind = a.batch, a.frames, a.model
    hp = HP.default_hparams(max_len_pad=T, batch_size=B)
    eng = Engine(kind, hp, B, T, device=dev)
    eng.load_weights(M.init_weights(kind, hp, 0))
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    if a.force_dp:
        eng.comm_init(0, 1)
    mel, f0, emb, lens = synth(B, T, 1000, dev)
    ncalls = 4 if kind == 'G3' else 3
    if kind == 'G6':
        from speechsplit_amd.utils import quantize_f0_torch
        onehot, qidx = quantize_f0_torch(f0[:, :, 0].clone())
        onehot, qidx = onehot.contiguous(), qidx.to(torch.int32).contiguous()

    def step():
        t0 = time.perf_counter()
        d = draw_interp(B, ncalls, hp)
        t1 = time.perf_counter()
        if kind == 'G3':
            if a.force_dp:
                eng.dp_train_step_native(mel, f0, emb, lens, d)
            else:
                eng.g3_train_step(mel, f0, emb, lens, d)
        elif a.force_dp:
            eng.g6_dp_train_step_native(mel, onehot, qidx, d)
        else:
            eng.g6_train_step(mel, onehot, qidx, d)
        return t1 - t0, time.perf_counter() - t1
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    draws, calls = [], []
    for _ in range(12):                      # three steps from an empty queue each time: no back-pressure from the GPU on the host
        torch.cuda.synchronize()
        for _ in range(3):
            d, c = step()
            draws.append(d * 1e3)
            calls.append(c * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 40 * 1e3
    eng.check()
    print(f'{kind} {B} x {T}{" (data-parallel step, world 1)" if a.force_dp else ""}: host draws {statistics.median(draws):.3f} ms + step call {statistics.median(calls):.3f} ms '
          f'(max {max(calls):.3f}) per step; the step on the GPU {gpu:.3f} ms', flush=True)


if __name__ == '__main__':
    main()
