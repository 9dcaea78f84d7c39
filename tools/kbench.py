#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box: A/B the tuning knobs of the step-LSTM and GEMM kernels in ONE process
(interleaved rounds, HIP events on the launch stream), each variant first checked against a PyTorch fp32 reference.
Writes gpurun_out/kbench.txt.   Usage: python tools/kbench.py [lstm] [gemm] [step]"""
import os
os.environ.setdefault('SS_DIAG_LIB', '1')      # the ablation modes (seq_prio > 1, gemm_diag, lstm_mode) exist only in the -DSS_DIAG build: make -C speechsplit_amd/csrc diag
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import _capi                     # noqa: E402
from speechsplit_amd import engine as E               # noqa: E402

OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, 'kbench.txt'), 'a')
lib = _capi.lib()


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + '\n')
    LOG.flush()


def P(t):
    return C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def tune(k, v):
    _capi.check(lib.ss_tune(k.encode(), int(v)))


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def lstm_ref(xproj, whh, B, T, H):
    """xproj [B,T,2,4H] (bias included), whh [2,4H,H] -> out [B,T,2H], gates [B,T,2,4H] activated, c [B,T,2H] (double)."""
    out = torch.zeros(B, T, 2 * H, dtype=torch.float64, device=xproj.device)
    cs = torch.zeros_like(out)
    ga = torch.zeros(B, T, 2, 4 * H, dtype=torch.float64, device=xproj.device)
    x = xproj.double()
    w = whh.double()
    for d in range(2):
        h = torch.zeros(B, H, dtype=torch.float64, device=xproj.device)
        c = torch.zeros_like(h)
        for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
            a = x[:, t, d] + h @ w[d].t()
            i, f, g, o = a.split(H, 1)
            i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
            c = f * c + i * g
            h = o * torch.tanh(c)
            out[:, t, d * H:(d + 1) * H] = h
            cs[:, t, d * H:(d + 1) * H] = c
            ga[:, t, d] = torch.cat((i, f, g, o), 1)
    return out, ga, cs


def bench_lstm(B=64, T=128, H=512):
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    TP = T + 4
    xproj = (torch.randn(B, T, 2, 4 * H, generator=g) * 0.5).to(dev)
    whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1).to(dev) / H ** 0.5
    d_out = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(dev)
    ref_out, ref_ga, ref_c = lstm_ref(xproj[:4, :16], whh, 4, 16, H)          # small slice for the correctness check

    def slabs(Bx, Tx, xp):
        gates = torch.zeros(Bx, Tx + 4, 8 * H, device=dev)
        gates[:, 2:2 + Tx] = xp.reshape(Bx, Tx, 8 * H)
        return gates, torch.zeros(Bx, Tx + 4, 2 * H, device=dev), torch.zeros(Bx, Tx + 4, 2 * H, device=dev)

    scratch = torch.zeros(8 * H * H + 16 * B * H + 2 * B * H + 1024, device=dev)
    dpad = torch.zeros(B, TP, 2 * H, device=dev)
    dpad[:, 2:2 + T] = d_out
    results = []
    for nw, gs in [(4, (4, 8)), (8, (2, 4)), (16, (1, 2))]:
        for gi in gs:
            tune('lstm_nw', nw)
            tune('lstm_g', gi)
            # correctness (fwd) on the small slice
            gs_, o_, c_ = slabs(4, 16, xproj[:4, :16].contiguous())
            _capi.check(lib.ss_op_lstm_fwd(P(gs_), P(whh[0]), P(whh[1]), P(o_), P(c_), P(scratch), scratch.numel(), 4, 16, H, S()))
            err = float((o_[:, 2:18].double() - ref_out).abs().max())
            gates, out, cs = slabs(B, T, xproj)
            xp_keep = gates.clone()

            def fwd():
                gates.copy_(xp_keep)
                _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S()))

            def copy_only():
                gates.copy_(xp_keep)
            tf = timeit(fwd)[0] - timeit(copy_only)[0]
            ga_keep = gates.clone()

            def bwd():
                gates.copy_(ga_keep)
                _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, S()))
            tb = timeit(bwd)[0] - timeit(copy_only)[0]
            results.append((nw, gi, tf, tb, err))
            say(f'lstm H{H} B{B} T{T} nw{nw} g{gi}: fwd {tf / T:.2f} us/step  bwd {tb / T:.2f} us/step  (fwd err {err:.1e})')
    # bwd grouping variants
    for nw, gi in [(4, 16), (8, 8), (8, 16), (16, 4), (16, 8)]:
        tune('lstm_nw', nw)
        tune('lstm_g', gi)
        gates, out, cs = slabs(B, T, xproj)
        xp_keep = gates.clone()
        gates.copy_(xp_keep)
        _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S()))
        ga_keep = gates.clone()

        def bwd():
            gates.copy_(ga_keep)
            _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, S()))

        def copy_only():
            gates.copy_(ga_keep)
        tb = timeit(bwd)[0] - timeit(copy_only)[0]
        say(f'lstm bwd nw{nw} g{gi}: {tb / T:.2f} us/step')


def bench_gemm():
    dev = 'cuda'
    shapes = [  # (name, M, N, K, ta, tb, ksplit)
        ('proj   NT 8192x4096x1024', 8192, 4096, 1024, False, False, 1),
        ('conv   NT 8192x512x2560 ', 8192, 512, 2560, False, False, 1),
        ('dX     NN 8448x1024x2048', 8448, 1024, 2048, False, True, 1),
        ('dW_ih  TN 2048x1024x8448', 2048, 1024, 8448, True, True, 4),
        ('dW_hh  TN 2048x512x8447 ', 2048, 512, 8447, True, True, 8),
        ('convdW TN 512x2560x8444 ', 512, 2560, 8444, True, True, 6),
        ('head   NT 8192x80x1024  ', 8192, 80, 1024, False, False, 1),
    ]
    for name, M, N, K, ta, tb, ks in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((K, N) if tb else (N, K), device=dev)
        c = torch.zeros(M, N, device=dev)
        ref = ((A.t() if ta else A).double() @ (Bm if tb else Bm.t()).double())
        for mode in (0, 1):
            for want in (256, 1024):
                tune('gemm_mode', mode)
                tune('gemm_want', want)
                c.zero_()
                E.gemm(A, Bm, None, ta, tb, ks, out=c)
                err = float((c.double() - ref).abs().max() / ref.abs().max())
                t, tmin = timeit(lambda: E.gemm(A, Bm, None, ta, tb, ks, out=c), iters=7)
                say(f'gemm {name} ks{ks} {"bf16x3" if mode else "fp32  "} want{want}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF  (min {tmin:.1f} us, err {err:.1e})')
    tune('gemm_mode', 1)
    tune('gemm_want', 1024)


def bench_gemm_diag():
    dev = 'cuda'
    shapes = [('proj   NT 8192x4096x1024', 8192, 4096, 1024, False, False, 1), ('conv   NT 8192x512x2560', 8192, 512, 2560, False, False, 1),
              ('dX     NN 8448x1024x4096', 8448, 1024, 4096, False, True, 1), ('dW_ih  TN 2048x1024x8448', 2048, 1024, 8448, True, True, 4)]
    for name, M, N, K, ta, tb, ks in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((K, N) if tb else (N, K), device=dev) * 0.05
        A[0, :7] = torch.tensor([1e-3, 3e-5, 1e-6, 2e-8, 100.0, -250.0, 0.0])       # small and large magnitudes in one reduction
        c = torch.zeros(M, N, device=dev)
        ref = ((A.t() if ta else A).double() @ (Bm if tb else Bm.t()).double())
        for rnd in range(2):
            for f16 in (False, True):
                c.zero_()
                E.gemm(A, Bm, None, ta, tb, ks, out=c, f16x2=f16)
                err = float((c.double() - ref).abs().max() / ref.abs().max())
                t, tmin = timeit(lambda: E.gemm(A, Bm, None, ta, tb, ks, out=c, f16x2=f16), iters=7)
                say(f'gemm {name} [{"fp16 x 2 (3 MFMA)" if f16 else "bf16 x 3 (6 MFMA)":20s}]: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF  err {err:.1e}')


def bench_gemm_ablate():
    """where a k-tile's time goes in the fp16 x 2 in-loop-split kernel (timing only: most modes give wrong results)"""
    dev = 'cuda'
    shapes = [('proj   NT 8192x4096x1024', 8192, 4096, 1024, False, False, 1), ('dX  NT-tr 8192x1024x4096', 8192, 1024, 4096, False, True, 1),
              ('dW_ih  TN 2048x1024x8192 ks4', 2048, 1024, 8192, True, True, 4), ('conv   NT 8192x512x2560', 8192, 512, 2560, False, False, 1)]
    modes = [(0, 'full'), (4, 'every k-tile re-reads tile 0 (no memory latency)'), (128, 'no global loads'), (64, 'no split / LDS store'), (32, 'no barriers'),
             (256, 'fragment reads all to one address'), (512, 'B operand taken as pre-split (timing only)'), (1536, 'both operands taken as pre-split (timing only)'), (192, 'no split, no loads'), (224, 'no split, no loads, no barriers'), (8, 'plain tile order (no XCD / L2 grouping)')]
    for name, M, N, K, ta, tb, ks in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((K, N) if tb else (N, K), device=dev)
        c = torch.zeros(M, N, device=dev)
        for dg, what in modes:
            tune('gemm_diag', dg)
            t, tmin = timeit(lambda: E.gemm(A, Bm, None, ta, tb, ks, out=c, f16x2=True), iters=7)
            say(f'gemm {name} [{what:50s}]: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF')
    tune('gemm_diag', 0)


def bench_gemm_ws():
    """256-thread kernel vs the wave-specialised 512-thread form (fp16 x 2, 128 x 128 tiles), each checked against fp64"""
    dev = 'cuda'
    shapes = [('proj   NT 8192x4096x1024', 8192, 4096, 1024, False, False, 1), ('dX  NT-tr 8192x1024x4096', 8192, 1024, 4096, False, True, 1),
              ('dW_ih  TN 2048x1024x8192 ks4', 2048, 1024, 8192, True, True, 4), ('dW_hh  TN 2048x512x8192 ks8', 2048, 512, 8192, True, True, 8),
              ('conv   NT 8192x512x2560', 8192, 512, 2560, False, False, 1), ('convdW TN 512x2560x8192 ks6', 512, 2560, 8192, True, True, 6),
              ('odd    NT 1000x520x1000', 1000, 520, 1000, False, False, 1), ('odd    TN 300x260x4100 ks3', 300, 260, 4100, True, True, 3),
              ('one k-tile NT 256x256x32', 256, 256, 32, False, False, 1), ('three k-tiles NT 256x256x96', 256, 256, 96, False, False, 1)]
    for name, M, N, K, ta, tb, ks in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((K, N) if tb else (N, K), device=dev)
        c = torch.zeros(M, N, device=dev)
        ref = ((A.t() if ta else A).double() @ (Bm if tb else Bm.t()).double())
        for rnd in range(2):
            for ws in (0, 2):
                tune('gemm_ws', ws)
                tune('gemm_want', 1)
                c.zero_()
                E.gemm(A, Bm, None, ta, tb, ks, out=c, f16x2=True)
                err = float((c.double() - ref).abs().max() / ref.abs().max())
                t, tmin = timeit(lambda: E.gemm(A, Bm, None, ta, tb, ks, out=c, f16x2=True), iters=7)
                say(f'gemm {name} [{"wave-specialised" if ws else "256 threads     "}]: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF  err {err:.1e}')
    tune('gemm_ws', 1)
    tune('gemm_want', 1024)


def bench_gemm_tiles():
    """tile choice per decoder shape: gemm_want forces 128 x 128 (1), 128 x 64 or 64 x 64 by asking for more workgroups"""
    dev = 'cuda'
    shapes = [('proj   NT 8192x4096x1024', 8192, 4096, 1024, False, False, 1), ('dX  NT-tr 8192x1024x4096', 8192, 1024, 4096, False, True, 1),
              ('dW_ih  TN 2048x1024x8192 ks4', 2048, 1024, 8192, True, True, 4), ('dW_ih  TN 2048x1024x8192 ks2', 2048, 1024, 8192, True, True, 2),
              ('dW_ih  TN 2048x1024x8192 ks8', 2048, 1024, 8192, True, True, 8), ('dW_hh  TN 2048x512x8192 ks8', 2048, 512, 8192, True, True, 8),
              ('dW_hh  TN 2048x512x8192 ks4', 2048, 512, 8192, True, True, 4), ('conv   NT 8192x512x2560', 8192, 512, 2560, False, False, 1)]
    for name, M, N, K, ta, tb, ks in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((K, N) if tb else (N, K), device=dev)
        c = torch.zeros(M, N, device=dev)
        t128 = (M // 128) * (N // 128) * ks
        for want, what in ((1, '128 x 128'), (t128 + 1, '128 x 64'), (2 * t128 + 1, '64 x 64')):
            tune('gemm_want', want)
            t, tmin = timeit(lambda: E.gemm(A, Bm, None, ta, tb, ks, out=c, f16x2=True), iters=7)
            say(f'gemm {name} [{what:9s} tiles, {t128 * (1 if want == 1 else (2 if want == t128 + 1 else 4)):5d} workgroups]: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF')
    tune('gemm_want', 1024)


def bench_gemm_skinny():
    """the decoder's layer-0 input gradient: 8448 x 164 x 4096 (narrow N), split-K and tile variants"""
    dev = 'cuda'
    M, N, K = 8448, 164, 4096
    A = torch.randn(M, K, device=dev)
    Bm = torch.randn(K, N, device=dev)
    c = torch.zeros(M, N, device=dev)
    ref = A.double() @ Bm.double()
    for ks in (1, 2, 4, 8):
        for want in (1, 300, 600, 1200, 5000):
            tune('gemm_want', want)
            c.zero_()
            E.gemm(A, Bm, None, False, True, ks, out=c, f16x2=True)
            err = float((c.double() - ref).abs().max() / ref.abs().max())
            t, tmin = timeit(lambda: E.gemm(A, Bm, None, False, True, ks, out=c, f16x2=True), iters=7)
            say(f'gemm dX0 NT 8448x164x4096 ks{ks} want{want:5d}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF  err {err:.1e}')
    tune('gemm_want', 1024)


def bench_step():
    from oracle import weights as W
    from oracle.gen_fixtures import synth_batch
    B, T = 64, 128
    hp = W.default_hparams(max_len_pad=T)
    mel, f0, emb, lens = [t.cuda() for t in synth_batch(1, B, T, 64)]
    sc, ls = E.draw_interp(B, 4, hp)
    sc, ls = sc.cuda(), ls.cuda()
    eng = E.Engine('G3', hp, B, T)
    eng.load_weights(W.make_weights('G3', hp, 0))
    for rnd in range(2):
        for ps, ov in [(1, 1), (0, 1)]:
            tune('seq_prio', ps)          # s_setprio 3 in the persistent recurrences on / off
            tune('overlap', ov)
            t, tmin = timeit(lambda: eng.g3_train_step(mel, f0, emb, lens, (sc, ls)), iters=10, warm=3)
            eng.check()
            say(f'train step seq_prio{ps} overlap{ov}: {t / 1e3:.3f} ms  ({B / t * 1e6:.0f} utt/s) min {tmin / 1e3:.3f}, loss {float(eng.loss):.6f}')
    tune('persist', 1)
    tune('overlap', 1)
    tune('seq_prio', 1)


def bench_lstm_modes(B=64, T=128, H=512):
    dev = 'cuda'
    gates = torch.zeros(B, T + 4, 8 * H, device=dev)
    out = torch.zeros(B, T + 4, 2 * H, device=dev)
    cs = torch.zeros(B, T + 4, 2 * H, device=dev)
    whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
    scratch = torch.zeros(8 * H * H + 16 * B * H + 2 * B * H + 1024, device=dev)
    names = {0: 'full', 1: 'loads, no MFMA', 2: 'MFMA, no operand loads', 3: 'empty kernel', 4: 'loads+MFMA, no epilogue'}
    for nw, g in [(4, 4), (8, 4), (16, 2)]:
        tune('lstm_nw', nw)
        tune('lstm_g', g)
        for mode in (0, 1, 2, 3, 4):
            tune('lstm_mode', mode)
            t = timeit(lambda: _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S())))[0]
            say(f'lstm fwd nw{nw} g{g} mode {mode} ({names[mode]}): {t / T:.2f} us/step')
    tune('lstm_mode', 0)
    # host launch rate: an empty kernel launched 128 times is also the floor of the eager path
    e = torch.zeros(1, device=dev)
    t = timeit(lambda: [e.add_(1) for _ in range(128)])[0]
    say(f'torch tiny kernel x128: {t / 128:.2f} us each')


def bench_seq(B=64, T=128, H=512):
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    xproj = (torch.randn(B, T, 2, 4 * H, generator=g) * 0.5).to(dev)
    whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1).to(dev) / H ** 0.5
    d_out = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(dev)
    scratch = torch.zeros(max(8 * H * H + 16 * B * H + 2 * B * H + 1024, ((B + 15) // 16) * (H // 16) ** 2 * 1024 + 4096), device=dev)
    gates = torch.zeros(B, T + 4, 8 * H, device=dev)
    gates[:, 2:2 + T] = xproj.reshape(B, T, 8 * H)
    xp_keep = gates.clone()
    out = torch.zeros(B, T + 4, 2 * H, device=dev)
    cs = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad[:, 2:2 + T] = d_out
    res = {}
    for rnd in range(2):
        for ps in (0, 1):
            tune('persist', ps)
            tune('lstm_nw', 16)

            def fwd():
                gates.copy_(xp_keep)
                _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S()))

            def copy_only():
                gates.copy_(xp_keep)
            tf = timeit(fwd)[0] - timeit(copy_only)[0]
            ga_keep = gates.clone()
            o_keep = out.clone()

            def bwd():
                gates.copy_(ga_keep)
                _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, S()))
            tb = timeit(bwd)[0] - timeit(copy_only)[0]
            bwd()
            res[ps] = (o_keep, gates.clone())
            say(f'lstm layer H{H} B{B} T{T} persist{ps}: fwd {tf:.0f} us ({tf / T:.2f}/step)  bwd {tb:.0f} us ({tb / T:.2f}/step)')
    say(f'   persist vs step kernels: out max diff {float((res[0][0] - res[1][0]).abs().max()):.2e}, '
        f'dgates max diff {float((res[0][1] - res[1][1]).abs().max()):.2e}')
    tune('persist', 1)
    # where a backward step's time goes: seq_prio bit 0 = s_setprio, bits 1.. = ablations (wrong results)
    for dg, what in [(0, 'full'), (1, 'no exchange loads (no polling)'), (4, 'no operand fetch'), (8, 'no slab stores'), (16, 'no wait (one poll load)'),
                     (64, 'no warm-up reads'), (12, 'no operand fetch, no slab stores'), (13, 'no polling, no fetch, no slab stores')]:
        tune('seq_prio', 1 | (dg << 1))
        tb = timeit(bwd)[0] - timeit(copy_only)[0]
        say(f'   bwd ablation [{what:40s}]: {tb / T:.2f} us/step')
    tune('seq_prio', 1)
    # forward: 16 = no waiting (one load pass, tags unchecked), 1 = no loads, 2 = no payload stores, 8... unused
    def fwd_only():
        _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S()))
    for dg, what in [(0, 'full'), (16, 'no waiting'), (17, 'no waiting, no exchange loads'), (19, '... and no payload stores'), (23, '... and no input prefetch'), (31, '... and no slab stores'), (4, 'no input prefetch only'), (8, 'no slab stores only')]:
        tune('seq_prio', 1 | (dg << 1))
        tf = timeit(fwd_only)[0]
        say(f'   fwd ablation [{what:40s}]: {tf / T:.2f} us/step')
    tune('seq_prio', 1)


def bench_seqtag(B=64, T=128, H=512):
    """flag hand-off vs tagged payload, interleaved rounds in one process; results compared with each other"""
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    xproj = (torch.randn(B, T, 2, 4 * H, generator=g) * 0.5).to(dev)
    whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1).to(dev) / H ** 0.5
    d_out = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(dev)
    scratch = torch.zeros(max(8 * H * H + 16 * B * H + 2 * B * H + 1024, 2 * ((B + 15) // 16) * (H // 16) ** 2 * 1024 + 4096), device=dev)
    gates = torch.zeros(B, T + 4, 8 * H, device=dev)
    gates[:, 2:2 + T] = xproj.reshape(B, T, 8 * H)
    xp_keep = gates.clone()
    out = torch.zeros(B, T + 4, 2 * H, device=dev)
    cs = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad[:, 2:2 + T] = d_out
    res = {}

    def fwd():
        gates.copy_(xp_keep)
        _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S()))

    def copy_only():
        gates.copy_(xp_keep)
    for rnd in range(3):
        for tg in (0, 1):
            tune('seq_tag', tg)
            tf = timeit(fwd, iters=9)[0] - timeit(copy_only, iters=9)[0]
            ga_keep = gates.clone()
            o_keep = out.clone()

            def bwd():
                gates.copy_(ga_keep)
                _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, S()))
            tb = timeit(bwd, iters=9)[0] - timeit(copy_only, iters=9)[0]
            bwd()
            res[tg] = (o_keep, gates.clone())
            say(f'lstm layer H{H} B{B} T{T} seq_tag{tg}: fwd {tf:.0f} us ({tf / T:.2f}/step)  bwd {tb:.0f} us ({tb / T:.2f}/step; includes a 16 MB memset)')
    tune('seq_tag', 1)
    say(f'   tagged vs flags (forward): out max diff {float((res[0][0] - res[1][0]).abs().max()):.2e}, '
        f'dgates max diff {float((res[0][1] - res[1][1]).abs().max()):.2e} (max |dgates| {float(res[0][1].abs().max()):.2e})')


def bench_seqvar(B=64, T=128, H=512, variants=(0, 2, 4)):
    """backward recurrence, memory-wave variants (ss_tune("seq_var")), interleaved rounds in one process; results compared bit for bit"""
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    xproj = (torch.randn(B, T, 2, 4 * H, generator=g) * 0.5).to(dev)
    whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1).to(dev) / H ** 0.5
    d_out = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(dev)
    scratch = torch.zeros(max(8 * H * H + 16 * B * H + 2 * B * H + 1024, 2 * ((B + 15) // 16) * (H // 16) ** 2 * 1024 + 4096), device=dev)
    gates = torch.zeros(B, T + 4, 8 * H, device=dev)
    gates[:, 2:2 + T] = xproj.reshape(B, T, 8 * H)
    out = torch.zeros(B, T + 4, 2 * H, device=dev)
    cs = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad[:, 2:2 + T] = d_out
    _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S()))
    ga_keep = gates.clone()
    flush = torch.empty(160 << 20, device=dev)          # 640 MB: larger than the memory-side cache

    def copy_only():
        gates.copy_(ga_keep)

    def bwd():
        gates.copy_(ga_keep)
        _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, S()))

    def cold_bwd():
        gates.copy_(ga_keep)
        flush.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, S()))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3
    ref = None
    warm = {v: [] for v in variants}
    cold = {v: [] for v in variants}
    for rnd in range(4):
        for v in (variants if rnd % 2 == 0 else variants[::-1]):
            tune('seq_var', v)
            tb = timeit(bwd, iters=9)[0] - timeit(copy_only, iters=9)[0]
            bwd()
            r = gates.clone()
            if ref is None:
                ref = r
            elif not torch.equal(r, ref):
                say(f'   seq_var {v}: RESULT DIFFERS, max diff {float((r - ref).abs().max()):.3e}')
            if rnd:
                warm[v].append(tb)
                cold[v].append(sorted(cold_bwd() for _ in range(5))[2])
    tune('seq_var', 0)
    for v in variants:
        w, c = sorted(warm[v]), sorted(cold[v])
        say(f'B{B} T{T} H{H} seq_var {v:2d}: bwd warm {w[1]:.0f} us ({w[1] / T:.2f}/step; {w[0]:.0f}..{w[-1]:.0f})   cold {c[1]:.0f} us ({c[1] / T:.2f}/step; {c[0]:.0f}..{c[-1]:.0f})')


if __name__ == '__main__':
    want = sys.argv[1:] or ['lstm', 'gemm', 'step']
    say('====', ' '.join(want), torch.cuda.get_device_name(0))
    if 'lstm' in want:
        bench_lstm()
    if 'seqtag' in want:
        bench_seqtag()
        bench_seqtag(B=16, T=64)
        bench_seqtag(B=48, T=192)
    if 'seq' in want:
        bench_seq()
    if 'seqvar' in want:
        bench_seqvar()
    if 'modes' in want:
        bench_lstm_modes()
    if 'gemm' in want:
        bench_gemm()
    if 'gskinny' in want:
        bench_gemm_skinny()
    if 'gtiles' in want:
        bench_gemm_tiles()
    if 'gws' in want:
        bench_gemm_ws()
    if 'gabl' in want:
        bench_gemm_ablate()
    if 'gdiag' in want:
        bench_gemm_diag()
    if 'step' in want:
        bench_step()
