#!/usr/bin/env python3
"""How the fused step's time depends on how many HIP streams the process created BEFORE the engine (HIP maps streams
round-robin onto 4 hardware queues, so the engine's branch streams end up sharing queues differently)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as W
from oracle.gen_fixtures import synth_batch
from speechsplit_amd import engine as E
B, T = 64, 128
own = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # 0: round 1's behaviour (the step runs on the caller's stream)
E.tune('own_streams', own)
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 1       # 0: round-1 enqueue order (filler work enqueued in front of critical-path launches)
E.tune('prio_order', prio)
if len(sys.argv) > 4:
    E.tune('probe_queues', int(sys.argv[4]))
print(f'own_streams = {own}, prio_order = {prio}', flush=True)
hp = W.default_hparams(max_len_pad=T)
mel, f0, emb, lens = [t.cuda() for t in synth_batch(1, B, T, 64)]
sc, ls = E.draw_interp(B, 4, hp)
sc, ls = sc.cuda(), ls.cuda()
w = W.make_weights('G3', hp, 0)
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
keep = []
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1      # only this many lone streams (for a profiler run)
for n_before in (range(0, 6) if only < 0 else [only]):
    for _ in range(n_before if only >= 0 else min(n_before, 1)):                                   # one more lone stream in the process before this engine's four
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        keep.append(st)
    eng = E.Engine('G3', hp, B, T)
    eng.load_weights(w)
    print('   ', eng.lib.ss_stream_report(eng.h).decode(), flush=True)
    for _ in range(5):
        eng.g3_train_step(mel, f0, emb, lens, (sc, ls))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        eng.g3_train_step(mel, f0, emb, lens, (sc, ls))
    torch.cuda.synchronize()
    print(f'{n_before} lone stream(s) created before the engine: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step', flush=True)
    del eng
