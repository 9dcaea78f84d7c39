"""Error of one decoder-size BLSTM layer (forward output, input gradient, weight gradients) against the same layer in float64 on the CPU:
same weights, same inputs -- isolates what the recurrence kernels' arithmetic (products, gate non-linearities, sums) costs, without the
trajectory noise of a trained-state comparison.  python tools/lstm_nl_error.py [H] [T] [B] [weight scale]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechsplit_amd import engine as E                                       # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    ws = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0          # trained LSTMs have larger weights than U(+-1/sqrt(H))
    In = 1024 if H == 512 else 512
    g = torch.Generator().manual_seed(7)
    k = ws / H ** 0.5
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * k
    w_ih, w_hh = (u(4 * H, In), u(4 * H, In)), (u(4 * H, H), u(4 * H, H))
    b_ih, b_hh = (u(4 * H), u(4 * H)), (u(4 * H), u(4 * H))
    x = torch.rand(B, T, In, generator=g) * 2 - 1
    d_out = torch.randn(B, T, 2 * H, generator=g) * 1e-3
    lstm = torch.nn.LSTM(In, H, 1, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for d, sfx in ((0, ''), (1, '_reverse')):
            getattr(lstm, 'weight_ih_l0' + sfx).copy_(w_ih[d])
            getattr(lstm, 'weight_hh_l0' + sfx).copy_(w_hh[d])
            getattr(lstm, 'bias_ih_l0' + sfx).copy_(b_ih[d])
            getattr(lstm, 'bias_hh_l0' + sfx).copy_(b_hh[d])
    x64 = x.double().requires_grad_(True)
    y64, _ = lstm(x64)
    (y64 * d_out.double()).sum().backward()
    l32 = torch.nn.LSTM(In, H, 1, batch_first=True, bidirectional=True)
    l32.load_state_dict({k_: v.float() for k_, v in lstm.state_dict().items()})
    x32 = x.clone().requires_grad_(True)
    y32, _ = l32(x32)
    (y32 * d_out).sum().backward()
    dev = 'cuda:0'
    c = lambda t: tuple(a.to(dev) for a in t)
    y, dx, grads = E.blstm_layer(x.to(dev), c(w_ih), c(w_hh), c(b_ih), c(b_hh), d_out.to(dev))
    print(f'BLSTM H={H} T={T} B={B} weight scale {ws}: error against float64 (max-norm relative): engine / PyTorch-CPU fp32')
    print(f'  output        {rel(y, y64):.2e} / {rel(y32, y64):.2e}')
    print(f'  d input       {rel(dx, x64.grad):.2e} / {rel(x32.grad, x64.grad):.2e}')
    for d, sfx in ((0, ''), (1, '_reverse')):
        gw_ih, gw_hh, gb = grads[d]
        for nm, gg in (('weight_ih_l0', gw_ih), ('weight_hh_l0', gw_hh), ('bias_ih_l0', gb)):
            r64 = getattr(lstm, nm + sfx).grad
            print(f'  d {nm + sfx:22s} {rel(gg, r64):.2e} / {rel(getattr(l32, nm + sfx).grad, r64):.2e}')


if __name__ == '__main__':
    main()
