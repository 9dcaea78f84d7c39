#!/usr/bin/env python3
"""Where the Solver.train loop's per-iteration overhead over the bare fused step comes from: the same 64 x 128 step driven with
(a) a fixed resident batch + fresh host draws (bench.py's loop), (b) + a batch assembled per step by DeviceBatcher on the compute
stream, (c) + the same through DevicePrefetcher (its own stream), (d) Solver.train itself."""
import contextlib, io, os, sys, tempfile, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from types import SimpleNamespace
from speechsplit_amd import data_loader as DL, hparams as HP, model as M, solver, staging
from speechsplit_amd.engine import Engine, draw_interp
B, T, N = 64, 128, 40
hp = HP.default_hparams(batch_size=B, max_len_pad=T)
dev = torch.device('cuda:0')
eng = Engine('G3', hp, B, T, device=dev)
eng.load_weights(M.init_weights('G3', hp, 0))
eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
loader = DL.get_device_loader(hp, dataset=DL.SyntheticUtterances(4 * B, seed=5))
it = iter(loader)
fixed = next(it)


def timed(fn, n=N):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def step(batch):
    mel, emb, f0, ln = batch
    eng.g3_train_step(mel, f0, emb, ln, draw_interp(B, 4, hp))


def host_only(fn, n=N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return t


print(f'(a) fixed batch, fresh draws:                 {timed(lambda: step(fixed)):.3f} ms/it', flush=True)
state = {'it': iter(loader)}
def nxt():
    try:
        return next(state['it'])
    except StopIteration:
        state['it'] = iter(loader)
        return next(state['it'])
print(f'(b) + DeviceBatcher.assemble on the step stream: {timed(lambda: step(nxt())):.3f} ms/it', flush=True)
print(f'    host time of assemble alone:                {host_only(nxt):.3f} ms/it', flush=True)
print(f'    host time of draw_interp alone:             {host_only(lambda: draw_interp(B, 4, hp)):.3f} ms/it', flush=True)
pf = staging.DevicePrefetcher(loader, dev)
print(f'(c) + DevicePrefetcher (own stream):             {timed(lambda: step(next(pf))):.3f} ms/it', flush=True)
with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
    cfg = SimpleNamespace(num_iters=8, g_lr=1e-4, beta1=0.9, beta2=0.999, resume_iters=None, use_tensorboard=False, device_id=0, log_dir=tmp, sample_dir=tmp,
                          model_save_dir=tmp, log_step=10, sample_step=10 ** 9, model_save_step=10 ** 9)
    s = solver.Solver(loader, cfg, hp)
    s.validation_pt = []
    s.train()
    torch.cuda.synchronize()
    s.num_iters = N
    t0 = time.perf_counter()
    s.train()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
print(f'(d) Solver.train():                              {dt:.3f} ms/it', flush=True)
