#!/usr/bin/env python3
"""Image GEMM (csrc/gemm_img.hip) vs the in-loop-split GEMM on the shapes of the training step: correctness against fp64 and isolated rates
per tile configuration, interleaved rounds in one process.  python tools/img_bench.py [diag]
With `diag` (needs the -DSS_DIAG library: SS_DIAG_LIB=1) also the k-loop ablations (1: no DMA in the loop, 2: no MFMAs)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import engine as E, _capi

dev = 'cuda'


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def timeit(fn, n=10, rounds=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(n):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        best.append(ev[0].elapsed_time(ev[1]) / n * 1e3)
    best.sort()
    return best[len(best) // 2]


SHAPES = [  # name, M, N, K, ta, tb, ksplit per cfg (0: 256x256, 1: 128x128, 2: 256x128), old ksplit
    ('dec proj  NT 8192x4096x1024', 8192, 4096, 1024, False, False, (1, 1, 1), 1),
    ('dec dX    NN 8192x1024x4096', 8192, 1024, 4096, False, True, (2, 1, 1), 1),
    ('dec dW_ih TN 2048x1024x8448', 2048, 1024, 8448, True, True, (8, 2, 4), 4),
    ('dec dW_hh TN 2048x512x8448', 2048, 512, 8448, True, True, (16, 4, 8), 8),
    ('conv fwd  NT 8192x512x2560', 8192, 512, 2560, False, False, (4, 1, 2), 1),
    ('conv fwd  NT 8192x256x1280', 8192, 256, 1280, False, False, (8, 2, 4), 1),
    ('conv dW   TN 512x2560x8448', 512, 2560, 8448, True, True, (13, 3, 6), 6),
    ('big       NT 8192x8192x4096', 8192, 8192, 4096, False, False, (1, 1, 1), 1),
]


def main():
    diag = 'diag' in sys.argv
    lib = _capi.lib()
    g = torch.Generator().manual_seed(1)
    print('--- image GEMM: rates in TFLOP/s algorithmic (executed MFMA rate is 3x); us per launch incl. the split-K reduce', flush=True)
    for name, M, N, K, ta, tb, kss, ks_old in SHAPES:
        A = torch.randn(M, K, generator=g).to(dev)
        B = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        As = A.t().contiguous() if ta else A
        Bs = B.t().contiguous() if tb else B
        ai, bi = E.split_image(As), E.split_image(Bs)
        out = torch.zeros(M, N, device=dev)
        fl = 2.0 * M * N * K
        ref = None
        if M * N * K <= 2 ** 36:
            ref = A.double() @ B.double().t()
        line = f'  {name}:'
        t_old = timeit(lambda: E.gemm(As, Bs, None, ta, tb, ks_old, out=out, f16x2=True))
        line += f' in-loop split {t_old:7.1f} us = {fl / t_old / 1e6:6.1f} TF |'
        for cfg in (0, 1, 2):
            ks = kss[cfg]
            part = torch.empty(ks * M * N, device=dev) if ks > 1 else None
            c = E.gemm_img(ai, bi, ta, tb, None, ks, cfg, out=out, part=part)
            err = rel(c, ref) if ref is not None else float('nan')
            t = timeit(lambda: E.gemm_img(ai, bi, ta, tb, None, ks, cfg, out=out, part=part))
            line += f' cfg{cfg} ks{ks}: {t:7.1f} us = {fl / t / 1e6:6.1f} TF (err {err:.1e}) |'
            if diag:
                for dg in (1, 2):
                    _capi.check(lib.ss_tune(b'gemm_diag', dg))
                    td = timeit(lambda: E.gemm_img(ai, bi, ta, tb, None, ks, cfg, out=out, part=part))
                    line += f' d{dg} {td:7.1f} |'
                _capi.check(lib.ss_tune(b'gemm_diag', 0))
        t_sp = timeit(lambda: E.split_image(As))
        line += f' split_image(A) {t_sp:6.1f} us'
        print(line, flush=True)
        # the single-piece (bf16) form on the same shape: plain bf16 operands, one MFMA per product
        a16, b16 = As.to(torch.bfloat16), Bs.to(torch.bfloat16)
        ref16 = (A.to(torch.bfloat16).double() @ B.to(torch.bfloat16).double().t()) if ref is not None else None
        line = f'  {"  ... bf16 single-piece":28s}:' + ' ' * 41
        for cfg in (0, 1, 2):
            ks = kss[cfg]
            part = torch.empty(ks * M * N, device=dev) if ks > 1 else None
            c = E.gemm_img(a16, b16, ta, tb, None, ks, cfg, out=out, part=part)
            err = rel(c, ref16) if ref16 is not None else float('nan')
            t = timeit(lambda: E.gemm_img(a16, b16, ta, tb, None, ks, cfg, out=out, part=part))
            line += f' cfg{cfg} ks{ks}: {t:7.1f} us = {fl / t / 1e6:6.1f} TF (err {err:.1e}) |'
        print(line, flush=True)


if __name__ == "__main__":
    main()
