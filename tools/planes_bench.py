#!/usr/bin/env python3
"""Pre-split ("planes") GEMM vs the in-loop-split GEMM on the shapes of the training step: correctness against fp64 and
isolated rates.  python tools/planes_bench.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import engine as E

dev = 'cuda'


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3      # us


def main():
    g = torch.Generator().manual_seed(1)
    print('--- correctness vs fp64 (max-norm relative)')
    for M, N, K, ks in [(256, 256, 64, 1), (300, 200, 100, 1), (1000, 520, 1024, 1), (512, 512, 4096, 4), (2048, 1024, 2112, 8)]:
        A = torch.randn(M, K, generator=g).to(dev)
        B = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        ref = A.double() @ B.double().t() + (bias.double() if ks == 1 else 0)
        c = E.gemm_planes(E.split_planes(A), E.split_planes(B), M, N, bias if ks == 1 else None, ks)
        # transposing split: the same product from reduction-major sources
        ct = E.gemm_planes(E.split_planes(A.t().contiguous(), transpose=True), E.split_planes(B.t().contiguous(), transpose=True), M, N,
                           bias if ks == 1 else None, ks)
        print(f'  {M}x{N}x{K} ksplit {ks}: planes {rel(c, ref):.2e}  transposed-split planes {rel(ct, ref):.2e}  '
              f'in-loop split {rel(E.gemm(A, B, bias if ks == 1 else None, ksplit=ks, f16x2=True), ref):.2e}', flush=True)

    print('--- isolated rates (TFLOP/s algorithmic; executed MFMA rate is 3x)')
    shapes = [('dec proj NT   8192x4096x1024', 8192, 4096, 1024, 1, False),
              ('dec dX  NN->NT 8192x1024x4096', 8192, 1024, 4096, 1, False),
              ('dec dW_ih TN  2048x1024x8448', 2048, 1024, 8448, 8, True),
              ('dec dW_hh TN  2048x512x8448', 2048, 512, 8448, 16, True),
              ('conv fwd NT   8192x512x2560', 8192, 512, 2560, 1, False),
              ('conv dW TN    512x2560x8448', 512, 2560, 8448, 8, True)]
    for name, M, N, K, ks, tn in shapes:
        A = torch.randn(M, K, generator=g).to(dev)
        B = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        pa, pb = E.split_planes(A), E.split_planes(B)
        out = torch.zeros(M, N, device=dev)
        t_new = timeit(lambda: E.gemm_planes(pa, pb, M, N, None, ks, out=out))
        if tn:        # what the step runs today: both operands reduction-major, split-K as pick_ksplit chooses (512 tiles target)
            At, Bt = A.t().contiguous(), B.t().contiguous()
            ks_old = max(1, min(32, 512 // (-(-M // 128) * -(-N // 128)), K // 256))
            t_old = timeit(lambda: E.gemm(At, Bt, None, True, True, ks_old, out=out, f16x2=True))
            t_split = timeit(lambda: (E.split_planes(At, transpose=True), E.split_planes(Bt, transpose=True)))
        else:
            t_old = timeit(lambda: E.gemm(A, B, None, ksplit=1, out=out, f16x2=True))
            t_split = timeit(lambda: (E.split_planes(A), E.split_planes(B)))
        fl = 2.0 * M * N * K
        print(f'  {name}: planes {t_new:7.1f} us = {fl / t_new / 1e6:6.1f} TF | in-loop split {t_old:7.1f} us = {fl / t_old / 1e6:6.1f} TF '
              f'| split passes of both operands {t_split:6.1f} us (incl. allocation)', flush=True)


if __name__ == "__main__":
    main()
