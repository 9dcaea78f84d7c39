#!/usr/bin/env python3
"""Soak of the data-parallel step on a one-rank RCCL group: every schedule for many steps, Engine.check() regularly,
and the final loss compared with the same number of fused steps (bit-for-bit equality is not
expected over hundreds of steps: split-K GEMMs accumulate with atomics, and Adam amplifies the order of summation)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
from oracle import weights as W
from oracle.gen_fixtures import synth_batch
from speechsplit_amd import engine as E
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
B, T = 64, 128
hp = W.default_hparams(max_len_pad=T)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29537')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
mel, f0, emb, lens = [t.cuda() for t in synth_batch(3, B, T, 64)]
torch.manual_seed(7)
draws = [tuple(x.cuda() for x in E.draw_interp(B, 4, hp)) for _ in range(steps)]
ref = None
ref_loss = None
for schedule in (None, None, 'overlap', 'after', 'join'):      # the fused step twice: run-to-run spread of the atomics' summation order
    eng = E.Engine('G3', hp, B, T)
    eng.load_weights(W.make_weights('G3', hp, 3))
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    t0 = time.perf_counter()
    for it in range(steps):
        if schedule is None:
            loss = eng.g3_train_step(mel, f0, emb, lens, draws[it])
        else:
            loss = eng.dp_train_step(mel, f0, emb, lens, draws[it], 1, schedule=schedule)
        if it % 100 == 99:
            eng.check()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = eng.params.clone()
    if ref is None:
        ref = p
    err = float((p - ref).abs().max() / ref.abs().max())
    lv = float(loss)
    if ref_loss is None:
        ref_loss = lv
    print(f'{str(schedule):8s}: {steps} steps, {dt / steps * 1e3:.3f} ms/step, final loss {lv:.6f}, max parameter difference to the first fused run {err:.2e}', flush=True)
    assert lv == lv and abs(lv - ref_loss) < 0.05 * ref_loss
    del eng
dist.destroy_process_group()
print('soak_dp ok')
