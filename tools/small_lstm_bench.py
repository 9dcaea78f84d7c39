#!/usr/bin/env python3
"""Isolated timing of the encoder-size BLSTM recurrences (ss_op_lstm_fwd / ss_op_lstm_bwd, csrc/lstm_small.hip): microseconds per launch and
per time step, single-wave kernels (small_lds = 1, where 4H <= 64) against the LDS kernels with barriers (small_lds = 2)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import _capi
from speechsplit_amd.engine import tune, _ptr, _stream
lib = _capi.lib()
dev = torch.device('cuda:0')
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for H in (1, 8, 32):
    for B in (16, 32, 64):
        R = B * (T + 4)
        g = torch.Generator(device='cpu').manual_seed(1)
        gates0 = (torch.randn(R, 8 * H, generator=g) * 0.5).to(dev)
        whf = (torch.randn(4 * H, H, generator=g) * 0.3).to(dev)
        whb = (torch.randn(4 * H, H, generator=g) * 0.3).to(dev)
        dout = (torch.randn(B, T + 4, 2 * H, generator=g) * 0.1).to(dev)
        out = torch.zeros(B, T + 4, 2 * H, device=dev)
        csave = torch.zeros(B, T + 4, 2 * H, device=dev)
        scratch = torch.zeros(1 << 16, device=dev)
        row = []
        for mode in (1, 2):
            tune('small_lds', mode)
            gates = gates0.clone()
            for which in ('fwd', 'bwd'):
                def call():
                    if which == 'fwd':
                        _capi.check(lib.ss_op_lstm_fwd(_ptr(gates), _ptr(whf), _ptr(whb), _ptr(out), _ptr(csave), _ptr(scratch), scratch.numel(), B, T, H, _stream()))
                    else:
                        _capi.check(lib.ss_op_lstm_bwd(_ptr(gates), _ptr(whf), _ptr(whb), _ptr(dout), _ptr(csave), _ptr(scratch), scratch.numel(), B, T, H, _stream()))
                for _ in range(5):
                    call()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(50):
                    call()
                b.record()
                torch.cuda.synchronize()
                row.append(a.elapsed_time(b) * 1e3 / 50)
        tune('small_lds', 1)
        print(f'H={H:2d} B={B:2d} T={T}: fwd {row[0]:6.1f} us ({row[0] / T:.2f}/step)  bwd {row[1]:6.1f} us ({row[1] / T:.2f}/step)   | LDS kernels: fwd {row[2]:6.1f}  bwd {row[3]:6.1f}', flush=True)
