#!/usr/bin/env python3
"""Where a wave of the 128x128 NT bf16x3 GEMM spends its cycles per k-tile (ss_debug_gemm_phases)."""
import os
os.environ.setdefault('SS_DIAG_LIB', '1')      # the k-loop phase probe (gemm_diag 16) exist only in the -DSS_DIAG build: make -C speechsplit_amd/csrc diag
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import engine as E, _capi
lib = _capi.lib()
M, N, K = 8192, 4096, 1024
a, b = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
c = torch.empty(M, N, device='cuda')
E.tune('gemm_diag', 16)
for _ in range(3):
    E.gemm(a, b, out=c)
buf = (C.c_ulonglong * 24)()
_capi.check(lib.ss_debug_gemm_phases(buf, 1))
for _ in range(10):
    E.gemm(a, b, out=c)
_capi.check(lib.ss_debug_gemm_phases(buf, 1))
E.tune('gemm_diag', 0)
kt = buf[5]          # k-tiles summed over 64 workgroups x 10 launches (wave 0)
names = ['split + LDS store', 'barrier 1', 'global load issue', 'fragments + MFMA', 'barrier 2']
print(f'k-tiles probed: {kt}')
for w in range(4):
    vals = [buf[w * 6 + i] / kt for i in range(5)]
    print(f'wave {w}: ' + '  '.join(f'{n} {v:7.0f}' for n, v in zip(names, vals)) + f'   total {sum(vals):7.0f} cycles / k-tile')
