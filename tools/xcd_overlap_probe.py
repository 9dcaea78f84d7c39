#!/usr/bin/env python3
"""Does a GEMM confined to the XCDs a persistent recurrence does NOT use leave the recurrence alone?  One decoder BLSTM layer at B = 32
(4 groups -> XCDs 0-3 under the round-robin placement, lstm_seq.hip) on stream A; an independent image GEMM (csrc/gemm_img.hip) on
stream B in three forms: plain grid (workgroups on every XCD, co-resident with the recurrence), work queue on all XCDs, work queue on
XCDs 4-7 only (ImgGemmDesc::xcc_allow).  HIP events give each kernel's span alone and together.
    python tools/xcd_overlap_probe.py [B]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import _capi                     # noqa: E402
from speechsplit_amd import engine as E               # noqa: E402

lib = _capi.lib()


def P(t):
    return C.c_void_p(t.data_ptr())


def main():
    dev = 'cuda'
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    T, H = 128, 512
    free = {16: 0xFC, 32: 0xF0, 48: 0xC0}.get(B, 0xFF)          # XCDs without a recurrence group
    g = torch.Generator(device='cpu').manual_seed(0)
    xproj = (torch.randn(B, T, 2, 4 * H, generator=g) * 0.5).to(dev)
    whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1).to(dev) / H ** 0.5
    scratch = torch.zeros(max(8 * H * H + 16 * 64 * H + 2 * 64 * H + 1024, 4 * (H // 16) ** 2 * 1024 + 4096) + (1 << 22), device=dev)
    gates = torch.zeros(B, T + 4, 8 * H, device=dev)
    gates[:, 2:2 + T] = xproj.reshape(B, T, 8 * H)
    out = torch.zeros(B, T + 4, 2 * H, device=dev)
    cs = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad[:, 2:2 + T] = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(dev)
    sa = torch.cuda.Stream()
    sbs = [torch.cuda.Stream() for _ in range(4)]
    R = B * T
    shapes = [('proj NT %dx4096x1024' % R, R, 4096, 1024, False, False, 1, 0),
              ('dW_ih TN 2048x1024x%d ks4' % R, 2048, 1024, R, True, True, 4, 2),
              ('dW_ih TN 2048x1024x%d ks8 256x256' % R, 2048, 1024, R, True, True, 8, 0)]

    def rec(kind):
        if kind == 'fwd':
            _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, C.c_void_p(sa.cuda_stream)))
        else:
            _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, C.c_void_p(sa.cuda_stream)))

    def ev():
        return torch.cuda.Event(enable_timing=True)

    print(f'==== {torch.cuda.get_device_name(0)}  B = {B}, T = {T}, H = {H}: recurrence groups on XCDs 0-{2 * ((B + 15) // 16) - 1}, GEMM allowed mask {free:#x}', flush=True)
    for kind in ('fwd', 'bwd'):
        for name, M, N, K, ta, tb, ks, cfg in shapes:
            A = torch.randn((K, M) if ta else (M, K), device=dev)
            Bm = (torch.randn((K, N) if tb else (N, K), device=dev) * 0.05)
            ai, bi = E.split_image(A), E.split_image(Bm)
            c = torch.zeros(M, N, device=dev)
            part = torch.empty(ks * M * N, device=dev) if ks > 1 else None
            masks = [free] if len(sys.argv) <= 2 else [int(m, 0) for m in sys.argv[2:]]
            for form, xcc in [('grid', 0), ('queue, all XCDs', 255)] + [(f'queue, XCDs {m:#x}', m) for m in masks]:
                reps = 2 if xcc in (0, 255) else 1              # about one recurrence's worth of GEMM work either way

                def gemm(sb):
                    E.tune('img_xcc', xcc)
                    with torch.cuda.stream(sb):
                        for _ in range(reps):
                            E.gemm_img(ai, bi, ta, tb, None, ks, cfg, out=c, part=part)
                    E.tune('img_xcc', 0)
                best = None
                for sb in sbs:
                    rows = []
                    for it in range(6):
                        torch.cuda.synchronize()
                        a0, a1 = ev(), ev()
                        a0.record(sa); rec(kind); a1.record(sa)
                        torch.cuda.synchronize()
                        g0, g1 = ev(), ev()
                        g0.record(sb); gemm(sb); g1.record(sb)
                        torch.cuda.synchronize()
                        w0, r1, q1 = ev(), ev(), ev()
                        w0.record(sa)
                        sb.wait_event(w0)
                        rec(kind); r1.record(sa)
                        gemm(sb); q1.record(sb)
                        torch.cuda.synchronize()
                        rows.append((a0.elapsed_time(a1) * 1e3, g0.elapsed_time(g1) * 1e3, w0.elapsed_time(r1) * 1e3, w0.elapsed_time(q1) * 1e3))
                    rows = rows[2:]
                    med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(4)]
                    if best is None or max(med[2], med[3]) < max(best[2], best[3]):
                        best = med
                med = best
                print(f'{kind} recurrence + {reps} x {name} [{form}]: alone rec {med[0]:.0f} us, gemm {med[1]:.0f} us | together: rec ends {med[2]:.0f} us '
                      f'(stretch {med[2] - med[0]:+.0f}), all done {med[3]:.0f} us (serial {med[0] + med[1]:.0f}, saved {med[0] + med[1] - max(med[2], med[3]):.0f})', flush=True)


if __name__ == '__main__':
    main()
