"""Where the engine's distance from exact arithmetic comes from, at a TRAINED state: train the engine N steps, hand the state to the
oracle evaluated in float64 (same inputs, draws, ReLU branches), and compare one further forward + backward under several arithmetic
settings of the engine (ss_tune) -- and the fp32 oracle itself -- against it.  Prints, per setting: loss and output error, and the
per-tensor gradient errors (max-norm relative to the tensor's maximum): worst tensors, median, number beyond 1e-4.

    python tools/trained_error_budget.py [G6|G3] [steps]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import interp_np, ref_model, weights as W                         # noqa: E402
from oracle.gen_fixtures import draws_for, synth_batch                        # noqa: E402
from speechsplit_amd import engine as E                                       # noqa: E402

LR = 1e-4


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def stack(draws):
    return np.stack([d[0] for d in draws]), np.stack([d[1] for d in draws])


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'G6'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    B, T, len_lo = (32, 192, 96) if kind == 'G6' else (64, 128, 64)
    ncalls = 4 if kind == 'G3' else 3
    hp = W.default_hparams(max_len_pad=T)
    w = W.make_weights(kind, hp, 3)
    eng = E.Engine(kind, hp, B, T)
    eng.load_weights(w)
    eng.set_adam(LR, 0.9, 0.999, 1e-8, 0)
    batches = [synth_batch(1400 + i, B, T, len_lo) for i in range(4)]

    def run(mel, f0, emb, lens, d, no_adam):
        if kind == 'G3':
            return eng.g3_train_step(mel, f0, emb, lens, d, no_adam=no_adam)
        q = torch.from_numpy(interp_np.quantize_f0(f0[:, :, 0].numpy()))
        return eng.g6_train_step(mel, torch.nn.functional.one_hot(q, 257).float(), q, d, no_adam=no_adam)

    E.tune('deterministic', 1)                 # a reproducible trained state, the one tests/test_gpu_configs.py compares at
    for i in range(steps):
        loss = run(*batches[i % 4], stack(draws_for(20000 + i, B, ncalls)), False)
    eng.check()
    E.tune('deterministic', 0)
    print(f'{kind} {B}x{T}: trained {steps} steps, loss {float(loss):.4f}')
    pv = {n: v.clone().cpu() for n, v in eng.param_views().items()}
    mel, f0, emb, lens = synth_batch(1300 + B, B, T, len_lo)
    draws = draws_for(1400 + B + steps, B, ncalls)
    qidx = torch.from_numpy(interp_np.quantize_f0(f0[:, :, 0].numpy()))
    onehot = torch.nn.functional.one_hot(qidx, 257).float()

    def oracle(dtype, masks):
        P = {n: v.to(dtype).requires_grad_(True) for n, v in pv.items()}
        ref_model.MASK, ref_model.MASK_STATS = masks, {}
        try:
            if kind == 'G3':
                xi = ref_model.interp(torch.cat((mel, f0), -1), lens.numpy(), draws[0], hp)
                oh, _ = ref_model.quantize_f0(xi[:, :, -1])
                out = ref_model.generator_3(P, hp, torch.cat((xi[:, :, :-1], oh), -1).to(dtype), mel.to(dtype), emb.to(dtype), draws[1:4], training=True)
                lo = torch.nn.functional.mse_loss(mel.to(dtype), out, reduction='mean')
            else:
                out = ref_model.generator_6(P, hp, mel.to(dtype), onehot.to(dtype), draws, training=True)
                lo = torch.nn.functional.cross_entropy(out.reshape(-1, out.shape[-1]), qidx.reshape(-1))
            lo.backward()
        finally:
            ref_model.MASK, ref_model.MASK_STATS = None, None
        return float(lo), out.detach(), {n: p.grad for n, p in P.items()}

    def engine_pass(tunes):
        for k, v in tunes.items():
            E.tune(k, v)
        lo = float(run(mel, f0, emb, lens, stack(draws), True))
        eng.check()
        out = eng.debug_buffer('out', B, T).cpu()
        g = {n: v.clone().cpu() for n, v in eng.grad_views().items()}
        m = {k: v.cpu() for k, v in eng.relu_masks(B, T).items()}
        return lo, out, g, m

    settings = [('default', {}),
                ('bwd exact (bf16x3)', {'bwd_f16x2': 0}),
                ('fwd+bwd exact GEMMs', {'bwd_f16x2': 0, 'fwd_f16x2': 0}),
                ('+ per-step fp32-MFMA recurrences', {'bwd_f16x2': 0, 'fwd_f16x2': 0, 'persist': 0}),
                ('+ fp32-MFMA GEMMs', {'bwd_f16x2': 0, 'fwd_f16x2': 0, 'persist': 0, 'gemm_mode': 0}),
                ('default again', {'bwd_f16x2': 1, 'fwd_f16x2': 1, 'persist': 1, 'gemm_mode': 1})]
    first = engine_pass({})
    lo64, out64, g64 = oracle(torch.float64, first[3])
    lo32, out32, g32 = oracle(torch.float32, first[3])

    def report(tag, lo, out, g):
        errs = sorted(((rel(g[n], g64[n]), n) for n in g64), reverse=True)
        beyond = [f'{n} {e:.1e}' for e, n in errs if e >= 1e-4]
        print(f'{tag:38s} loss {abs(lo - lo64) / abs(lo64):.1e}  out {rel(out, out64):.1e}  grads: worst {errs[0][0]:.1e} ({errs[0][1]}), '
              f'median {errs[len(errs) // 2][0]:.1e}, beyond 1e-4: {beyond}')

    def by_class(tag, g):
        # the same errors grouped by what kind of reduction produced the tensor: (median, worst) per class
        cls = {}
        for n in g64:
            k = ('lstm.bias' if '.bias_' in n else 'lstm.W_ih' if 'weight_ih' in n else 'lstm.W_hh' if 'weight_hh' in n else
                 'head.' + n.rsplit('.', 1)[1] if 'linear_projection' in n else 'conv.' + n.rsplit('.', 1)[1] if '.conv.' in n else
                 'gn.' + n.rsplit('.', 1)[1])
            k = ('dec ' if n.startswith('decoder') else 'enc ') + k
            cls.setdefault(k, []).append(rel(g[n], g64[n]))
        print(f'    {tag}: ' + '; '.join(f'{k} {sorted(v)[len(v) // 2]:.1e}/{max(v):.1e}' for k, v in sorted(cls.items())))

    if kind == 'G6':
        # the loss kernel on its own: the engine's d(loss)/d(logits) against float64 cross-entropy gradients of the ENGINE's logits, and the
        # same for PyTorch's fp32 cross_entropy on those logits
        z = eng.debug_buffer('out', B, T).cpu()
        dz = eng.debug_buffer('d_out', B, T).cpu()
        z64 = z.double().reshape(-1, z.shape[-1]).requires_grad_(True)
        torch.nn.functional.cross_entropy(z64, qidx.reshape(-1)).backward()
        z32 = z.reshape(-1, z.shape[-1]).clone().requires_grad_(True)
        torch.nn.functional.cross_entropy(z32, qidx.reshape(-1)).backward()
        print(f'loss kernel alone (same logits): d_logits engine {rel(dz.reshape(-1, z.shape[-1]), z64.grad):.1e}, PyTorch fp32 {rel(z32.grad, z64.grad):.1e}; '
              f'column sums (the head bias gradient): engine {rel(dz.reshape(-1, z.shape[-1]).double().sum(0), z64.grad.sum(0)):.1e}, '
              f'PyTorch fp32 {rel(z32.grad.double().sum(0), z64.grad.sum(0)):.1e}; max |logit| {float(z.abs().max()):.1f}')
    print('errors against the float64 oracle (max-norm relative per tensor):')
    report('fp32 oracle (PyTorch CPU)', lo32, out32, g32)
    by_class('fp32 oracle, median/worst per class', g32)
    by_class('engine default, median/worst per class', first[2])
    if os.environ.get('SS_BUDGET_SETTINGS'):                 # e.g. SS_BUDGET_SETTINGS=default: only the named settings
        keep = os.environ['SS_BUDGET_SETTINGS'].split(',')
        settings = [s for s in settings if s[0] in keep]
    for tag, tunes in settings:
        try:
            lo, out, g, _ = engine_pass(tunes)
        except Exception as ex:                 # a knob this build does not have
            print(f'{tag:38s} skipped: {ex}')
            continue
        report('engine, ' + tag, lo, out, g)
    # the same state, inputs and arithmetic again and again: what varies is the order of the fp32 atomics (split-K, bias sums)
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    for r in range(reps):
        lo, out, g, _ = engine_pass({})
        errs = sorted(((rel(g[n], g64[n]), n) for n in g64), reverse=True)
        print(f'repeat {r}: ' + ', '.join(f'{n} {e:.1e}' for e, n in errs[:4]))
    if reps:
        E.tune('deterministic', 1)
        for r in range(2):
            lo, out, g, _ = engine_pass({})
            errs = sorted(((rel(g[n], g64[n]), n) for n in g64), reverse=True)
            print(f'deterministic mode {r}: ' + ', '.join(f'{n} {e:.1e}' for e, n in errs[:4]))
        E.tune('deterministic', 0)


if __name__ == '__main__':
    main()
