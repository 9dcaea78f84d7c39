#!/usr/bin/env python3
"""Stage-by-stage parity report of the HIP engine against the CPU oracle (runs on the GPU box).

Writes gpurun_out/diag.txt.  Every section is independent and never aborts the run, so one call localises
a wrong kernel.  Usage: python tools/gpu_diag.py [section ...]   (sections: gemm interp g3eval g3train step g6 time)
"""
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import interp_np, ref_model, weights as W          # noqa: E402
from oracle.gen_fixtures import synth_batch, draws_for          # noqa: E402
from speechsplit_amd import engine as E                         # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, 'diag.txt'), 'w')
torch.set_num_threads(16)


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + '\n')
    LOG.flush()


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def sec_gemm():
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in [(128, 128, 64), (256, 512, 400), (100, 80, 164), (2048, 512, 2560), (333, 257, 66), (64, 64, 16),
                      (8192, 80, 1024)]:
        for ta, tb in [(False, False), (False, True), (True, True)]:
            A = torch.randn((K, M) if ta else (M, K), generator=g)
            Bm = torch.randn((K, N) if tb else (N, K), generator=g)
            bias = torch.randn(N, generator=g)
            ref = (A.t() if ta else A).double() @ (Bm if tb else Bm.t()).double() + bias.double()
            for ks in (1, 3):
                c = E.gemm(A.to(dev), Bm.to(dev), bias.to(dev), ta, tb, ks)
                say(f'gemm M{M} N{N} K{K} ta{int(ta)} tb{int(tb)} ks{ks}: rel {rel(c, ref):.2e}')


def sec_interp(eng128, eng192):
    z = np.load(os.path.join(GOLD, 'interp.npz'))
    for i in range(int(z['n'])):
        pad = int(z[f'c{i}_max_len_pad'])
        eng = eng128 if pad == 128 else eng192
        x = torch.from_numpy(z[f'c{i}_x'])
        y, i0, lam, cnt = eng.interp_forward(x, z[f'c{i}_len_seq'], z[f'c{i}_scales'], z[f'c{i}_len_seg'], want_plan=True)
        ri0, rlam, rcnt, rn = interp_np.interp_plan(z[f'c{i}_scales'], z[f'c{i}_len_seg'], z[f'c{i}_len_seq'], max_len_pad=pad)
        ok_y = np.array_equal(y.cpu().numpy(), z[f'c{i}_y'])
        say(f'interp case {i}: y bit-exact {ok_y} (max diff {np.abs(y.cpu().numpy() - z[f"c{i}_y"]).max():.3e}) '
            f'i0 {np.array_equal(i0.cpu().numpy(), ri0)} lam {np.array_equal(lam.cpu().numpy(), rlam)} '
            f'counts {np.array_equal(cnt.cpu().numpy(), rcnt)}')
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(i))
        dx = eng.interp_backward(dy, x.shape[1])
        rdx = interp_np.interp_backward(dy.numpy(), ri0, rlam, rn, x.shape[1])
        say(f'   backward rel {rel(dx, torch.from_numpy(rdx)):.2e}')


def compare_taps(eng, tap, B, T):
    names = set(eng.debug_names())
    for k in sorted(tap):
        cand = [k, k.replace('dec.lstm.out', 'dec.lstm.out'), k.replace('enc1.lstm1.out', 'enc1.lstm1.out')]
        nm = None
        for c in cand:
            if c in names:
                nm = c
        if nm is None:
            say(f'   tap {k}: (no engine buffer)')
            continue
        buf = eng.debug_buffer(nm, B, T)
        say(f'   {k:22s} rel {rel(buf, tap[k]):.2e}')


def sec_g3eval(eng):
    z = np.load(os.path.join(GOLD, 'demo_config1.npz'))
    hp = W.default_hparams()
    w = W.make_weights('G3', hp, int(z['seed_g3']))
    eng.load_weights(w)
    P = ref_model.as_params(w, False)
    for n in range(2):
        mel = torch.from_numpy(z[f'u{n}_mel_pad'])
        onehot = torch.from_numpy(interp_np.onehot(z[f'u{n}_qidx'].astype(np.int64)))[None]
        emb = torch.from_numpy(z[f'u{n}_emb'])
        x_f0 = torch.cat((mel, onehot), -1)
        ref_model.TAP = {}
        with torch.no_grad():
            ref = ref_model.generator_3(P, hp, x_f0, mel, emb)
        tap, ref_model.TAP = ref_model.TAP, None
        out = eng.g3_forward(x_f0, mel, emb)
        say(f'g3 eval demo u{n}: out vs oracle rel {rel(out, ref):.2e}; vs reference fixture {rel(out, torch.from_numpy(z[f"u{n}_out3"])):.2e}')
        compare_taps(eng, tap, 1, 192)
        rh = eng.g3_rhythm(mel)
        say(f'   rhythm codes rel {rel(rh, torch.from_numpy(z[f"u{n}_rhythm"])):.2e}')


def grads_report(eng, P, top=8):
    gv = eng.grad_views()
    rows = []
    for n, p in P.items():
        if p.grad is None:
            continue
        rows.append((rel(gv[n], p.grad), n))
    rows.sort(reverse=True)
    say(f'   grads: worst {rows[0][0]:.2e} ({rows[0][1]}), median {rows[len(rows) // 2][0]:.2e}')
    for r, n in rows[:top]:
        say(f'      {r:.2e}  {n}')
    return rows[0][0]


def sec_g3train(eng, B=2, T=128, eval_mode=False):
    hp = W.default_hparams(max_len_pad=T)
    w = W.make_weights('G3', hp, 3)
    eng.load_weights(w)
    P = ref_model.as_params(w)
    mel, f0, emb, lens = synth_batch(31, B, T, 64 if T == 128 else 96)
    draws = draws_for(41, B, 4)
    # B1-style call: x_f0 built on the host exactly as solver.py:160-163 does, forward in train mode, custom d_out
    xi = ref_model.interp(torch.cat((mel, f0), -1), lens.numpy(), draws[0], hp)
    onehot, _ = ref_model.quantize_f0(xi[:, :, -1])
    x_in = torch.cat((xi[:, :, :-1], onehot), -1)
    ref_model.TAP = {}
    ref = ref_model.generator_3(P, hp, x_in, mel, emb, draws[1:4], training=not eval_mode)
    tap, ref_model.TAP = ref_model.TAP, None
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(7)) * 1e-3
    ref.backward(gy)
    sc = np.stack([d[0] for d in draws[1:4]])
    ls = np.stack([d[1] for d in draws[1:4]])
    out = eng.g3_forward(x_in, mel, emb, (sc, ls), training=not eval_mode)
    say(f'g3 {"eval" if eval_mode else "train"}-mode forward B{B} T{T}: out rel {rel(out, ref):.2e}')
    compare_taps(eng, tap, B, T)
    eng.g3_backward(gy)
    torch.cuda.synchronize()
    grads_report(eng, P)


def sec_step(eng, tag='b2_t128'):
    rec = json.load(open(os.path.join(GOLD, 'train_steps.json')))[tag]
    B, T = rec['B'], rec['T']
    hp = W.default_hparams(max_len_pad=T)
    w = W.make_weights('G3', hp, rec['wseed'])
    eng.load_weights(w)
    eng.adam_m.zero_()
    eng.adam_v.zero_()
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    st = ref_model.TrainState(w)
    mel, f0, emb, lens = synth_batch(rec['bseed'], B, T, 64 if T == 128 else 96)
    nsteps = len(rec['losses'])
    draws = draws_for(rec['dseed'], B, 4 * nsteps)
    for it in range(nsteps):
        d4 = draws[4 * it:4 * it + 4]
        sc = np.stack([d[0] for d in d4])
        ls = np.stack([d[1] for d in d4])
        loss = eng.g3_train_step(mel, f0, emb, lens, (sc, ls))
        lo, out = st.step_g3(hp, mel, f0, emb, lens.numpy(), d4)
        say(f'step {tag} it{it}: loss engine {float(loss):.8f} oracle {float(lo):.8f} reference {rec["losses"][it]:.8f}')
        if it == 0:
            o = eng.debug_buffer('out', B, T)
            say(f'   out vs reference fixture rel {rel(o, torch.from_numpy(np.load(os.path.join(GOLD, f"train_{tag}_out.npy")))):.2e}')
            xm = eng.debug_buffer('in.mel', B, T)
            say(f'   resampled mel bit-exact vs reference: {np.array_equal(xm.cpu().numpy(), np.load(os.path.join(GOLD, f"train_{tag}_xin_mel.npy")))}')
            xo = eng.debug_buffer('in.f0', B, T)[:, :, :257].argmax(-1).cpu().numpy()
            say(f'   quantised f0 classes equal: {np.array_equal(xo, np.load(os.path.join(GOLD, f"train_{tag}_xin_f0idx.npy")).astype(np.int64))}')
            gv = eng.grad_views()
            worst = 0
            for n, s in rec['grads'].items():
                flat = gv[n].reshape(-1).cpu()
                for p, v in zip(s['pos'], s['val']):
                    worst = max(worst, abs(float(flat[p]) - v) / (s['amax'] + 1e-30))
                worst = max(worst, abs(float(gv[n].double().norm()) - s['l2']) / (s['l2'] + 1e-30))
            say(f'   grads vs reference fixture (samples, l2): worst rel {worst:.2e}')
        pv = eng.param_views()
        worst = max(rel(pv[n], p) for n, p in st.P.items())
        say(f'   params after Adam vs oracle: worst rel {worst:.2e}')


def sec_g6(eng):
    rec = json.load(open(os.path.join(GOLD, 'g6_train.json')))
    B, T = rec['B'], rec['T']
    hp = W.default_hparams(max_len_pad=T)
    w = W.make_weights('G6', hp, rec['wseed'])
    eng.load_weights(w)
    P = ref_model.as_params(w)
    mel, f0, emb, lens = synth_batch(rec['bseed'], B, T, 96)
    qidx = torch.from_numpy(interp_np.quantize_f0(f0[:, :, 0].numpy()))
    onehot = torch.nn.functional.one_hot(qidx, 257).float()
    draws = draws_for(rec['dseed'], B, 3)
    ref_model.TAP = {}
    loss, logits = ref_model.g6_loss(P, hp, mel, onehot, qidx, draws)
    tap, ref_model.TAP = ref_model.TAP, None
    loss.backward()
    sc = np.stack([d[0] for d in draws])
    ls = np.stack([d[1] for d in draws])
    out = eng.g6_forward(mel, onehot, (sc, ls), training=True)
    say(f'g6 train fwd: logits rel {rel(out, logits):.2e}; vs reference fixture {rel(out, torch.from_numpy(np.load(os.path.join(GOLD, "g6_train_logits.npy")))):.2e}')
    compare_taps(eng, tap, B, T)
    l2 = eng.g6_train_step(mel, onehot, qidx, (sc, ls), no_adam=True)
    say(f'   CE loss engine {float(l2):.7f} oracle {float(loss):.7f} reference {rec["loss"]:.7f}')
    grads_report(eng, P)


def sec_time(B=64, T=128, steps=10):
    hp = W.default_hparams(max_len_pad=T)
    eng = E.Engine('G3', hp, B, T)
    eng.load_weights(W.make_weights('G3', hp, 0))
    mel, f0, emb, lens = synth_batch(1, B, T, 64)
    mel, f0, emb, lens = mel.cuda(), f0.cuda(), emb.cuda(), lens.cuda().int()
    sc, ls = E.draw_interp(B, 4, hp)
    sc, ls = sc.cuda(), ls.cuda()
    for _ in range(3):
        eng.g3_train_step(mel, f0, emb, lens, (sc, ls))
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        eng.g3_train_step(mel, f0, emb, lens, (sc, ls))
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    say(f'time: B{B} T{T} train step {dt * 1e3:.2f} ms -> {B / dt:.1f} utt/s, loss {float(eng.loss):.6f}')


def main():
    want = sys.argv[1:] or ['gemm', 'interp', 'g3eval', 'g3train', 'step', 'g6', 'time']
    say('device', torch.cuda.get_device_name(0))
    engs = {}

    def eng(kind, T, B=8):
        k = (kind, T)
        if k not in engs:
            engs[k] = E.Engine(kind, W.default_hparams(max_len_pad=T), B, T)
        return engs[k]

    plan = {
        'gemm': lambda: sec_gemm(),
        'interp': lambda: sec_interp(eng('G3', 128, 16), eng('G3', 192)),
        'g3eval': lambda: sec_g3eval(eng('G3', 192)),
        'g3train': lambda: (sec_g3train(eng('G3', 128, 16), 2, 128), sec_g3train(eng('G3', 128, 16), 2, 128, eval_mode=True)),
        'step': lambda: (sec_step(eng('G3', 128, 16), 'b2_t128'), sec_step(eng('G3', 192), 'b2_t192')),
        'g6': lambda: sec_g6(eng('G6', 192)),
        'time': lambda: sec_time(),
    }
    for name in want:
        say(f'==== {name}')
        try:
            plan[name]()
            torch.cuda.synchronize()
        except Exception:
            say('EXCEPTION in section', name)
            say(traceback.format_exc())
    say('done')


if __name__ == '__main__':
    main()
