"""How does the matrix pipe round when it adds products to an fp32 accumulator?  A contraction of all-POSITIVE operands has no cancellation, so
round-to-nearest accumulation leaves a zero-mean error of ~sqrt(n) * 2^-25 relative after n accumulations, while an accumulator that
TRUNCATES (rounds towards zero) leaves a systematic deficit of ~n * 2^-24.  Prints mean signed and rms relative error against float64 for the
engine's GEMM kernels (fp32 MFMA, bf16 x 3, fp16 x 2) over reduction lengths 256 .. 32768, K-contiguous and reduction-major layouts.

    python tools/mfma_round_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechsplit_amd import engine as E                                       # noqa: E402


def main():
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(5)
    print('mean signed / rms relative error vs float64; positive operands in [0.5, 1); M = N = 128')
    for K in (256, 1024, 4096, 16384, 32768):
        a = (torch.rand(128, K, generator=g) * 0.5 + 0.5)
        b = (torch.rand(128, K, generator=g) * 0.5 + 0.5)
        ref = a.double() @ b.double().t()
        ad, bd = a.to(dev), b.to(dev)
        at, bt = ad.t().contiguous(), bd.t().contiguous()
        row = [f'K {K:6d}:']
        for tag, mode, kw in (('fp32-mfma', 0, {}), ('bf16x3', 1, {}), ('f16x2', 1, {'f16x2': True}), ('bf16x1', 1, {'bf16': True})):
            E.tune('gemm_mode', mode)
            for lay, c in (('NT', E.gemm(ad, bd, **kw)), ('TN', E.gemm(at, bt, ta=True, tb=True, **kw))):
                err = (c.double().cpu() - ref) / ref
                row.append(f'{tag} {lay} {float(err.mean()):+.2e}/{float(err.pow(2).mean().sqrt()):.1e}')
        E.tune('gemm_mode', 1)
        print('  '.join(row))
    # the same sum in fp32 on the vector ALU, chained (what a CPU GEMM's inner loop does), for scale
    for K in (4096, 32768):
        a = (torch.rand(16, K, generator=g) * 0.5 + 0.5)
        b = (torch.rand(16, K, generator=g) * 0.5 + 0.5)
        ref = (a.double() * b.double()).sum(1)
        acc = torch.zeros(16)
        for k in range(K):
            acc = acc + a[:, k] * b[:, k]
        err = (acc.double() - ref) / ref
        print(f'K {K:6d}: sequential fp32 chain on the CPU {float(err.mean()):+.2e}/{float(err.pow(2).mean().sqrt()):.1e}; torch.mm fp32 on the CPU '
              f'{float(((a @ b.t()).diag().double() - ref).div(ref).mean()):+.2e}')


if __name__ == '__main__':
    main()
