#!/usr/bin/env python3
"""Runs each GEMM shape of the training step a few times with the kernel the step uses for it (for rocprofv3 --pmc collection)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (name, bench.py kernel class, kernel: 'img' = gemm_img.hip over operand images / 'old' = gemm_bf16x3.hip with the in-loop split, M, N, K, ta, tb, ksplit)
SHAPES = [('proj NT 8192x4096x1024 (image GEMM, 256x256)', 'dec_proj', 'img', 8192, 4096, 1024, False, False, 1),
          ('conv NT 8192x512x2560 (image GEMM, 128x128)', 'conv_fwd', 'img', 8192, 512, 2560, False, False, 1),
          ('dX NN 8192x1024x4096', 'dec_dx', 'old', 8192, 1024, 4096, False, True, 1),
          ('dW_ih TN 2048x1024x8448 ks4', 'dec_dw', 'old', 2048, 1024, 8448, True, True, 4),
          ('dW_hh TN 2048x512x8447 ks8', 'dec_dw', 'old', 2048, 512, 8447, True, True, 8),
          ('dW_ih TN 2048x1024x8448 ks4 (image GEMM, 256x128; not the step\'s default)', 'dec_dw_img', 'img', 2048, 1024, 8448, True, True, 4)]
LAUNCHES = 4

if __name__ == '__main__':
    import torch
    from speechsplit_amd import engine as E, _capi
    lib = _capi.lib()
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    _capi.check(lib.ss_tune(b'gemm_want', want))
    if len(sys.argv) > 2:      # timing ablations: run with SS_DIAG_LIB=1 (make -C speechsplit_amd/csrc diag)
        _capi.check(lib.ss_tune(b'gemm_diag', int(sys.argv[2])))
    for name, _, kern, M, N, K, ta, tb, ks in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device='cuda')
        B = torch.randn((K, N) if tb else (N, K), device='cuda')
        c = torch.zeros(M, N, device='cuda')
        if kern == 'img':
            ai, bi = E.split_image(A), E.split_image(B)
            part = torch.empty(ks * M * N, device='cuda') if ks > 1 else None
        for _ in range(LAUNCHES):
            if kern == 'img':
                E.gemm_img(ai, bi, ta, tb, None, ks, -1 if ks == 1 else 2, out=c, part=part)
            else:
                E.gemm(A, B, None, ta, tb, ks, out=c, f16x2=True)      # the variant the training step runs (fp16 x 2; fixed scale here)
        torch.cuda.synchronize()
    print('ok')
