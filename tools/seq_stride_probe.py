#!/usr/bin/env python3
"""Is the persistent backward recurrence sensitive to how far apart a step's operand rows lie?  The slabs are batch-major
[B, T+4, C]: one step reads 16 utterance rows (T+4) * C * 4 bytes apart, so the distance grows with T while the work per step does
not.  Differential step time between sequence lengths (fixed launch cost cancels).  Writes gpurun_out/seq_stride_probe.txt."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import _capi                     # noqa: E402

lib = _capi.lib()
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, 'seq_stride_probe.txt'), 'a')


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + '\n')
    LOG.flush()


def P(t):
    return C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=9, warm=3):
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
    return ts[len(ts) // 2]


def main(B=64, H=512):
    dev = 'cuda'
    res = {}
    for T in (16, 32, 64, 128, 256, 512):
        g = torch.Generator(device='cpu').manual_seed(0)
        whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1).to(dev) / H ** 0.5
        scratch = torch.zeros(max(8 * H * H + 16 * B * H + 2 * B * H + 1024, 2 * ((B + 15) // 16) * (H // 16) ** 2 * 1024 + 4096), device=dev)
        gates = torch.zeros(B, T + 4, 8 * H, device=dev)
        gates[:, 2:2 + T] = (torch.randn(B, T, 8 * H, generator=g) * 0.5).to(dev)
        out = torch.zeros(B, T + 4, 2 * H, device=dev)
        cs = torch.zeros(B, T + 4, 2 * H, device=dev)
        dpad = torch.zeros(B, T + 4, 2 * H, device=dev)
        dpad[:, 2:2 + T] = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(dev)

        def fwd():
            _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S()))

        def bwd():
            _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, S()))
        tf = timeit(fwd)
        tb = timeit(bwd)
        res[T] = (tf, tb)
        say(f'B{B} H{H} T{T:4d}: utterance rows {(T + 4) * 8 * H * 4 / 1e6:6.2f} MB apart (gates)   fwd {tf:8.1f} us   bwd {tb:8.1f} us')
        del gates, out, cs, dpad
    ks = sorted(res)
    for a, b in zip(ks, ks[1:]):
        say(f'   T {a:4d} -> {b:4d}: fwd {(res[b][0] - res[a][0]) / (b - a):.2f} us/step   bwd {(res[b][1] - res[a][1]) / (b - a):.2f} us/step')


def layouts(B=64, T=128, H=512):
    """batch-major vs time-major slabs on the same values, operands cold (a 1 GB copy runs between launches, as the rest of a
    training step does) and warm (back-to-back launches)"""
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1).to(dev) / H ** 0.5
    scratch = torch.zeros(max(8 * H * H + 16 * B * H + 2 * B * H + 1024, 2 * ((B + 15) // 16) * (H // 16) ** 2 * 1024 + 4096), device=dev)
    xp = torch.zeros(B, T + 4, 8 * H, device=dev)
    xp[:, 2:2 + T] = (torch.randn(B, T, 8 * H, generator=g) * 0.5).to(dev)
    dp = torch.zeros(B, T + 4, 2 * H, device=dev)
    dp[:, 2:2 + T] = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(dev)
    junk_a = torch.zeros(128 << 20, device=dev)
    junk_b = torch.zeros(128 << 20, device=dev)
    res = {}
    for tm in (0, 1, 0, 1):
        _capi.check(lib.ss_tune(b'op_time_major', tm))
        f = (lambda t: t.transpose(0, 1).contiguous()) if tm else (lambda t: t.clone())
        gates, dpad = f(xp), f(dp)
        out, cs = torch.zeros_like(dpad), torch.zeros_like(dpad)
        keep = gates.clone()

        def fwd():
            _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, S()))

        def bwd():
            _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, S()))

        def timed(fn, cold, prewarm=False):
            ts = []
            for _ in range(7):
                if fn is fwd:
                    gates.copy_(keep)
                if cold:
                    junk_b.copy_(junk_a)
                if prewarm:          # one streaming read of the operand slabs right before the launch (memory-side cache warm-up)
                    sink = gates.sum() + cs.sum() + dpad.sum()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            return ts[len(ts) // 2]
        tfw, tfc = timed(fwd, False), timed(fwd, True)
        fwd_gates = gates.clone()
        o = out.clone()
        tbw, tbc = timed(bwd, False), timed(bwd, True)
        tbp = timed(bwd, True, True)
        gates.copy_(fwd_gates)
        bwd()
        res[tm] = ((o.transpose(0, 1) if tm else o).contiguous(), (gates.transpose(0, 1) if tm else gates).contiguous())
        say(f'B{B} T{T} H{H} {"time-major [T+4,B,C] " if tm else "batch-major [B,T+4,C]"}: fwd warm {tfw:6.1f} us ({tfw / T:.2f}/step) cold {tfc:6.1f} us ({tfc / T:.2f}/step)   '
            f'bwd warm {tbw:6.1f} us ({tbw / T:.2f}/step) cold {tbc:6.1f} us ({tbc / T:.2f}/step) cold + streaming pre-read {tbp:6.1f} us ({tbp / T:.2f}/step)')
    _capi.check(lib.ss_tune(b'op_time_major', 0))
    say(f'   layouts agree: out max diff {float((res[0][0] - res[1][0]).abs().max()):.2e}, dgates max diff {float((res[0][1] - res[1][1]).abs().max()):.2e}')


if __name__ == '__main__':
    say('====', torch.cuda.get_device_name(0))
    if 'layouts' in sys.argv[1:]:
        layouts()
        layouts(T=192)
        layouts(B=32, T=192)
    else:
        main()
