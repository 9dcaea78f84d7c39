#!/usr/bin/env python3
"""Runs each GEMM shape of the training step a few times (for rocprofv3 --pmc collection)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import engine as E, _capi
lib = _capi.lib()
want = int(sys.argv[1]) if len(sys.argv) > 1 else 256
_capi.check(lib.ss_tune(b'gemm_want', want))
if len(sys.argv) > 2:      # timing ablations: run with SS_DIAG_LIB=1 (make -C speechsplit_amd/csrc diag)
    _capi.check(lib.ss_tune(b'gemm_diag', int(sys.argv[2])))
shapes = [(8192, 4096, 1024, False, False, 1), (8192, 512, 2560, False, False, 1), (8192, 1024, 4096, False, True, 1),
          (2048, 1024, 8448, True, True, 4), (2048, 512, 8447, True, True, 8)]
for M, N, K, ta, tb, ks in shapes:
    A = torch.randn((K, M) if ta else (M, K), device='cuda')
    B = torch.randn((K, N) if tb else (N, K), device='cuda')
    c = torch.zeros(M, N, device='cuda')
    for _ in range(4):
        E.gemm(A, B, None, ta, tb, ks, out=c, f16x2=True)      # the variant the training step runs (fp16 x 2; fixed scale here)
    torch.cuda.synchronize()
print('ok')
