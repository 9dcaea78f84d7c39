#!/usr/bin/env python3
"""Per-workgroup placement log of ONE work-queue image GEMM launched beside a persistent recurrence (B = 32): which XCD each workgroup
ran on, how many tiles it took, when it started and ended (ss_debug_img_wq).  python tools/xcd_overlap_log.py [mask] [fwd|bwd]"""
import ctypes as C
import os
import sys
from collections import defaultdict

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
    mask = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0xF0
    kind = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
    B, T, H = 32, 128, 512
    g = torch.Generator(device='cpu').manual_seed(0)
    xproj = (torch.randn(B, T, 2, 4 * H, generator=g) * 0.5).to(dev)
    whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1).to(dev) / H ** 0.5
    scratch = torch.zeros(max(8 * H * H + 16 * 64 * H + 2 * 64 * H + 1024, 4 * (H // 16) ** 2 * 1024 + 4096) + (1 << 22), device=dev)
    gates = torch.zeros(B, T + 4, 8 * H, device=dev)
    gates[:, 2:2 + T] = xproj.reshape(B, T, 8 * H)
    out = torch.zeros(B, T + 4, 2 * H, device=dev)
    cs = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad = torch.zeros(B, T + 4, 2 * H, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    M, N, K = B * T, 4096, 1024
    A = torch.randn(M, K, device=dev)
    Bm = torch.randn(N, K, device=dev) * 0.05
    ai, bi = E.split_image(A), E.split_image(Bm)
    c = torch.zeros(M, N, device=dev)
    torch.cuda.synchronize()

    def rec():
        if kind == 'fwd':
            _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, C.c_void_p(sa.cuda_stream)))
        else:
            _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, C.c_void_p(sa.cuda_stream)))

    def gemm():
        E.tune('img_xcc', mask | 0x100)
        with torch.cuda.stream(sb):
            E.gemm_img(ai, bi, False, False, None, 1, 0, out=c)
        E.tune('img_xcc', 0)

    for together in (False, True, True):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(sa)
        sb.wait_event(e0)
        if together:
            rec()
        e1.record(sa)
        gemm()
        e2.record(sb)
        torch.cuda.synchronize()
        buf = (C.c_uint * (4 + 4 * 256))()
        _capi.check(lib.ss_debug_img_wq(buf, 4 + 4 * 256))
        w = list(buf)
        rows = [w[4 + 4 * i: 8 + 4 * i] for i in range(256)]
        t_min = min(r[2] for r in rows if r[2])
        per = defaultdict(list)
        for r in rows:
            per[r[0]].append(r)
        print(f'--- {"beside the " + kind + " recurrence" if together else "alone"}: mask {mask:#x}; recurrence span {e0.elapsed_time(e1) * 1e3:.0f} us, GEMM done at {e0.elapsed_time(e2) * 1e3:.0f} us; tiles handed out {w[0]}')
        for x in sorted(per):
            rs = per[x]
            tiles = sum(r[1] for r in rs)
            st = sorted((r[2] - t_min) & 0xFFFFFFFF for r in rs)
            en = sorted((r[3] - t_min) & 0xFFFFFFFF for r in rs)
            print(f'   XCD {x & 15}{" (left without work)" if x & 0x100 else ""}: {len(rs)} workgroups, {tiles} tiles; start {st[0] / 100:.0f}..{st[-1] / 100:.0f} us, end {en[0] / 100:.0f}..{en[-1] / 100:.0f} us')


if __name__ == '__main__':
    main()
