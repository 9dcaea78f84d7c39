#!/usr/bin/env python3
"""Ablation of the planes GEMM's k-loop (SS_DIAG_LIB=1, make diag): full / no DMA in the loop / no MFMAs."""
import os, sys, torch
os.environ['SS_DIAG_LIB'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import engine as E
from tools.planes_bench import timeit
g = torch.Generator().manual_seed(1)
for name, M, N, K, ks in [('proj 8192x4096x1024', 8192, 4096, 1024, 1), ('dW_ih 2048x1024x8448 ks8', 2048, 1024, 8448, 8), ('big 8192x8192x4096', 8192, 8192, 4096, 1)]:
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) * 0.05).cuda()
    pa, pb = E.split_planes(A), E.split_planes(B)
    out = torch.zeros(M, N, device='cuda')
    res = []
    for diag in (0, 1, 2):
        E.tune('gemm_diag', diag)
        t = timeit(lambda: E.gemm_planes(pa, pb, M, N, None, ks, out=out))
        res.append(f'{["full", "no DMA", "no MFMA"][diag]} {t:7.1f} us = {2.0 * M * N * K / t / 1e6:6.1f} TF')
    E.tune('gemm_diag', 0)
    print(f'{name}: ' + ' | '.join(res), flush=True)
