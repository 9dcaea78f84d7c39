#!/usr/bin/env python3
"""Persistent recurrence kernels vs. the per-step kernels on one layer: seq_debug.py <B> [ablation bits].
B = 64 gives XCD-local groups (ordinary payload stores), B = 16 / 48 groups that span XCDs (sc1 path)."""
import os
os.environ.setdefault('SS_DIAG_LIB', '1')      # the recurrence ablation bits exist only in the -DSS_DIAG build: make -C speechsplit_amd/csrc diag
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import engine as E, _capi
import ctypes as C
lib = _capi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
B, T, H = int(sys.argv[1]), 6, 512
DIAG = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(0)
dev = 'cuda'
xproj = (torch.randn(B, T, 2, 4 * H, generator=g) * 0.5).to(dev)
whh = ((torch.rand(2, 4 * H, H, generator=g) * 2 - 1) / H ** 0.5).to(dev)
d_out = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(dev)
scratch = torch.zeros(max(8 * H * H + 16 * B * H + 2 * B * H + 1024, ((B + 15) // 16) * (H // 16) ** 2 * 1024 + 4096), device=dev)
res = {}
for ps in (0, 1):
    E.tune('persist', ps)
    E.tune('seq_prio', 1 | (DIAG << 1) if ps else 1)
    gates = torch.zeros(B, T + 4, 8 * H, device=dev)
    gates[:, 2:2 + T] = xproj.reshape(B, T, 8 * H)
    out = torch.zeros(B, T + 4, 2 * H, device=dev)
    cs = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad = torch.zeros(B, T + 4, 2 * H, device=dev)
    dpad[:, 2:2 + T] = d_out
    _capi.check(lib.ss_op_lstm_fwd(P(gates), P(whh[0]), P(whh[1]), P(out), P(cs), P(scratch), scratch.numel(), B, T, H, None))
    _capi.check(lib.ss_op_lstm_bwd(P(gates), P(whh[0]), P(whh[1]), P(dpad), P(cs), P(scratch), scratch.numel(), B, T, H, None))
    torch.cuda.synchronize()
    res[ps] = gates[:, 2:2 + T].reshape(B, T, 2, 4, H).clone()
E.tune('persist', 1)
E.tune('seq_prio', 1)
d = (res[0] - res[1]).abs()
print('max diff', float(d.max()), 'ref max', float(res[0].abs().max()))
for dr in range(2):
    for t in range(T):
        print(f'dir {dr} t {t}: max diff {float(d[:, t, dr].max()):.3e}   per gate {[round(float(d[:, t, dr, gg].max()), 6) for gg in range(4)]}')
dd = d[:, :, 0].amax(dim=(1, 2))      # per (b, j) for dir 0
print('per batch row (dir0):', [round(float(x), 5) for x in dd.amax(dim=1)])
print('per j block of 16 (dir0):', [round(float(dd[:, k * 16:(k + 1) * 16].max()), 5) for k in range(H // 16)])
