#!/usr/bin/env python3
"""A/B of the two LDS images for reduction-major GEMM operands (ss_tune gemm_tr): time per shape, HIP events."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import engine as E, _capi
lib = _capi.lib()
_capi.check(lib.ss_tune(b'gemm_want', 256))
shapes = [('dW_ih TN', 2048, 1024, 8448, True, True, 4), ('dW_hh TN', 2048, 512, 8448, True, True, 8), ('conv dW TN', 512, 2560, 8444, True, True, 6),
          ('dX NN', 8192, 1024, 4096, False, True, 1), ('proj NT (unaffected)', 8192, 4096, 1024, False, False, 1)]
for name, M, N, K, ta, tb, ks in shapes:
    A = torch.randn((K, M) if ta else (M, K), device='cuda')
    B = torch.randn((K, N) if tb else (N, K), device='cuda')
    c = torch.zeros(M, N, device='cuda')
    ref = None
    for tr in (0, 1):
        _capi.check(lib.ss_tune(b'gemm_tr', tr))
        for _ in range(3):
            E.gemm(A, B, None, ta, tb, ks, out=c, f16x2=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            E.gemm(A, B, None, ta, tb, ks, out=c, f16x2=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        c.zero_()
        E.gemm(A, B, None, ta, tb, 1, out=c, f16x2=True)
        if ref is None:
            ref = c.clone()
        print(f'{name:22s} {M}x{N}x{K} ks{ks} gemm_tr={tr}: {us:7.1f} us  {2 * M * N * K / us / 1e6:6.1f} TFLOP/s   max diff vs tr=0: {float((c - ref).abs().max()):.2e}', flush=True)
_capi.check(lib.ss_tune(b'gemm_tr', 1))
