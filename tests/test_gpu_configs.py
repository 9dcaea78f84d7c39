"""Oracle parity of the HIP engine at the per-GPU shapes BASELINE.json's configs name, checked on the GPU box against the CPU
oracle run there (pytest -m gpu).  The oracle (oracle/ref_model.py) is pinned to the reference by tests/golden (CPU suite).

  config 2   Generator_3 fp32, 16 x 128                     -> test_g3_fp32_config_shapes[b16_t128]
  headline   Generator_3 fp32, 64 x 128 (4 batch tiles, the 256-workgroup persistent grid bench.py times)   [b64_t128]
  config 5   Generator_3, 64 utterances/GPU x 192 frames, lengths 96..192                                   [b64_t192]
  config 4   Generator_6, 32 utterances/GPU x 192 frames     -> test_g6_config4_shape
  configs 3-5 in the bf16 product mode against the FP32 oracle, gradients included  -> test_bf16_mode_against_fp32_oracle

Bars.  fp32: loss 1e-5, output and EVERY element of EVERY gradient tensor 1e-4 (max-norm relative per tensor).  No tensor gets
a looser bound: the oracle is handed the ReLU branch the engine took (ss_debug_relu_mask -> ref_model.MASK), and the test
asserts that this only ever overrode the oracle AT the kink (|GroupNorm output| < 2e-5 where the two disagree).
Adam: the update is p -= lr * g / (|g| + eps) at step 1, i.e. +-lr wherever |g| >> 1e-8 -- a gradient element whose sign is
within rounding flips a full 2*lr = 2e-4, which is 50x the 1e-4 * max|p| bar of a decoder tensor.  So the optimiser is held
to two checks that mean something: (a) the engine's update equals torch.optim.Adam's formulas applied to the engine's own
gradients to 1e-6 * max|p|, and (b) against the oracle's weights no element is off by more than 2.1 * lr and at most 0.1 %
of the elements of a tensor by more than 1e-4 * max|p| (printed).
bf16 product mode (operands of every contraction rounded to bf16, everything else fp32) against the fp32 oracle: stated
bounds BF16_BOUNDS below, measured worst cases are printed.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import interp_np, ref_model, weights as W
from oracle.gen_fixtures import draws_for, synth_batch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4
LR = 1e-4


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope='module')
def E():
    from speechsplit_amd import engine
    return engine


def stack_draws(draws):
    return np.stack([d[0] for d in draws]), np.stack([d[1] for d in draws])


class Case:
    """One engine + one oracle TrainState stepping side by side on the same batch and draws."""

    def __init__(self, E, kind, B, T, len_lo, wseed, bseed, precision='f32'):
        self.kind, self.B, self.T = kind, B, T
        self.hp = W.default_hparams(max_len_pad=T)
        w = W.make_weights(kind, self.hp, wseed)
        self.eng = E.Engine(kind, self.hp, B, T)
        self.eng.set_precision(precision)
        self.eng.load_weights(w)
        self.eng.set_adam(LR, 0.9, 0.999, 1e-8, 0)
        self.st = ref_model.TrainState(w, LR)
        self.mel, self.f0, self.emb, self.lens = synth_batch(bseed, B, T, len_lo)
        self.ncalls = 4 if kind == 'G3' else 3
        if kind == 'G6':
            self.qidx = torch.from_numpy(interp_np.quantize_f0(self.f0[:, :, 0].numpy()))
            self.onehot = torch.nn.functional.one_hot(self.qidx, 257).float()
        self.dseed = bseed + 100

    def step(self, it, kink_bound=2e-5, override=True):
        """Engine: forward, loss, backward (gradients kept), then Adam.  Oracle: the same step taking the engine's ReLU branches.
        Returns dict(loss=(gpu, cpu), out=(gpu, cpu), grads={name: (gpu, cpu)}, before/after params of the engine)."""
        eng, B, T = self.eng, self.B, self.T
        draws = draws_for(self.dseed + it, B, self.ncalls)
        if self.kind == 'G3':
            loss = eng.g3_train_step(self.mel, self.f0, self.emb, self.lens, stack_draws(draws), no_adam=True)
        else:
            loss = eng.g6_train_step(self.mel, self.onehot, self.qidx, stack_draws(draws), no_adam=True)
        eng.check()
        r = dict(loss_gpu=float(loss), out_gpu=eng.debug_buffer('out', B, T).cpu(),
                 grads_gpu={n: v.clone().cpu() for n, v in eng.grad_views().items()},
                 p_before={n: v.clone().cpu() for n, v in eng.param_views().items()})
        masks = {k: v.cpu() for k, v in eng.relu_masks(B, T).items()}
        eng.adam_step()
        r['p_after'] = {n: v.clone().cpu() for n, v in eng.param_views().items()}
        if override:
            ref_model.MASK, ref_model.MASK_STATS = masks, {}
        try:
            if self.kind == 'G3':
                lo, out = self.st.step_g3(self.hp, self.mel, self.f0, self.emb, self.lens.numpy(), draws)
            else:
                lo, out = self.st.step_g6(self.hp, self.mel, self.onehot, self.qidx, draws)
            stats = ref_model.MASK_STATS
        finally:
            ref_model.MASK, ref_model.MASK_STATS = None, None
        if override:
            assert set(stats) == set(masks), (sorted(stats), sorted(masks))
            flips = {k: v for k, v in stats.items() if v[0]}
            print(f'[{self.kind} {B}x{T} step {it}] ReLU branches overridden at the kink: {flips or "none"}')
            for k, (n, zmax) in stats.items():
                assert zmax < kink_bound, (k, n, zmax)          # the override never touched a clearly signed value
        r.update(loss_cpu=float(lo), out_cpu=out, grads_cpu={n: p.grad.clone() for n, p in self.st.P.items()},
                 p_cpu={n: p.detach().clone() for n, p in self.st.P.items()})
        return r


def check_fp32_step(r, tag):
    assert abs(r['loss_gpu'] - r['loss_cpu']) <= 1e-5 * abs(r['loss_cpu']), (tag, r['loss_gpu'], r['loss_cpu'])
    assert rel(r['out_gpu'], r['out_cpu']) < TOL, tag
    worst = ('', 0.0)
    for n, g in r['grads_cpu'].items():
        e = rel(r['grads_gpu'][n], g)
        worst = max(worst, (n, e), key=lambda x: x[1])
        assert e < TOL, (tag, n, e)                          # every element of every tensor, one bar
    print(f'[{tag}] loss {r["loss_gpu"]:.8f} (oracle {r["loss_cpu"]:.8f}); worst gradient tensor {worst[0]}: {worst[1]:.2e}')


def check_adam(r, tag, step):
    """(a) the engine's update == torch.optim.Adam's single-tensor formulas on the engine's own gradients (first step only:
    moments start at zero); (b) against the oracle's weights: hard bound 2.1 * lr, at most 2e-5 of a tensor's elements (or 2) beyond 1e-4 * max|p|."""
    tot, off = 0, 0
    for n, pc in r['p_cpu'].items():
        pa, pb, g = r['p_after'][n].double(), r['p_before'][n].double(), r['grads_gpu'][n].double()
        amax = float(pb.abs().max())
        if step == 0:
            m = 0.1 * g
            v = 0.001 * g * g
            upd = (LR / (1 - 0.9)) * m / (v.sqrt() / np.sqrt(1 - 0.999) + 1e-8)
            assert float((pa - (pb - upd)).abs().max()) <= 1e-6 * amax + 1e-9, (tag, n)
        d = (pa - pc.double()).abs()
        assert float(d.max()) <= 2.1 * LR * (step + 1), (tag, n, float(d.max()))
        bad = int((d > TOL * amax).sum())
        tot += d.numel()
        off += bad
        assert bad <= max(2, 2e-5 * d.numel()), (tag, n, bad, d.numel())      # measured: 0 of 19.4 M (round 2); 2e-5 leaves room for the sign of a ~0 gradient
    print(f'[{tag}] post-Adam weights: {off} of {tot} elements beyond 1e-4 * max|p| of their tensor (sign of a ~0 gradient)')


# --------------------------------------------------------------------------------------------- fp32, Generator_3
@pytest.mark.parametrize('shape', [(16, 128, 64), (64, 128, 64), (64, 192, 96)], ids=['b16_t128', 'b64_t128', 'b64_t192'])
def test_g3_fp32_config_shapes(E, shape):
    B, T, len_lo = shape
    c = Case(E, 'G3', B, T, len_lo, wseed=0, bseed=900 + B + T)
    losses = []
    for it in range(2):                   # the second step runs on the first one's update
        r = c.step(it)
        check_fp32_step(r, f'G3 {B}x{T} step {it}')
        check_adam(r, f'G3 {B}x{T} step {it}', it)
        losses.append((r['loss_gpu'], r['loss_cpu']))
    # resampled inputs: bit-exact against the oracle's index path at this size too
    draws = draws_for(c.dseed + 1, B, 4)
    xi = ref_model.interp(torch.cat((c.mel, c.f0), -1), c.lens.numpy(), draws[0], c.hp)
    assert np.array_equal(c.eng.debug_buffer('in.mel', B, T).cpu().numpy(), xi[:, :, :80].numpy())
    cls = c.eng.debug_buffer('in.f0', B, T)[:, :, :257].argmax(-1).cpu().numpy()
    assert np.array_equal(cls, interp_np.quantize_f0(xi[:, :, -1].numpy()))


def test_g3_all_fp32_mfma_mode_b64_t128(E):
    """The headline shape once more with EVERY product fp32-wide (bench.py's alt_precisions.all_fp32_mfma): GEMMs on v_mfma_f32_32x32x2_f32
    (ss_tune gemm_mode 0) and the decoder recurrences as one fp32-MFMA launch per time step (persist 0) -- the reference's arithmetic width
    (model.py has no autocast anywhere).  Same bars as the default mode."""
    E.tune('gemm_mode', 0)
    E.tune('persist', 0)
    try:
        c = Case(E, 'G3', 64, 128, 64, wseed=0, bseed=900 + 64 + 128)
        r = c.step(0)
        check_fp32_step(r, 'G3 64x128, all products on the fp32 MFMA')
        check_adam(r, 'G3 64x128, all products on the fp32 MFMA', 0)
    finally:
        E.tune('gemm_mode', 1)
        E.tune('persist', 1)


# --------------------------------------------------------------------------------------------- the same comparison WITHOUT the ReLU hand-over
@pytest.mark.parametrize('case', [('G3', 16, 128, 64), ('G3', 64, 128, 64), ('G3', 64, 192, 96), ('G6', 32, 192, 96)],
                         ids=['g3_b16_t128', 'g3_b64_t128', 'g3_b64_t192', 'g6_b32_t192'])
def test_no_override_relu_diagnostic(E, case):
    """Round-3 review: every whole-model gradient comparison runs with the oracle taking the ENGINE's ReLU branches, so a sign bug in
    gn_relu_bwd could hide behind the hook.  Here the oracle decides its own branches.  A GroupNorm output within rounding of zero may then
    land on the other side in the two implementations, and that element's gradient contribution differs -- a handful of elements, each
    worth at most one frame of one channel -- so the tensors beyond 1e-4 are REPORTED, not asserted; what is asserted is that nothing is
    off by more than 1e-2 (a wrong branch rule would put whole tensors off by O(1)), that loss and output hold their usual bars (they do
    not depend on the backward's branches) and that the decoder's and the head's gradients, whose backward passes through no ReLU, hold 1e-4."""
    kind, B, T, len_lo = case
    c = Case(E, kind, B, T, len_lo, wseed=0 if kind == 'G3' else 4, bseed=(900 if kind == 'G3' else 51) + B + T)
    r = c.step(0, override=False)
    assert abs(r['loss_gpu'] - r['loss_cpu']) <= 1e-5 * abs(r['loss_cpu'])
    assert rel(r['out_gpu'], r['out_cpu']) < TOL
    errs = {n: rel(r['grads_gpu'][n], g) for n, g in r['grads_cpu'].items()}
    beyond = {n: e for n, e in errs.items() if not e < TOL}
    print(f'[{kind} {B}x{T}, oracle with its OWN ReLU branches] gradient tensors beyond 1e-4: '
          f'{ {n: float(f"{e:.1e}") for n, e in sorted(beyond.items())} or "none"}; worst {max(errs.values()):.2e}')
    for n, e in errs.items():
        assert e < 1e-2, (n, e)
        if n.startswith('decoder.'):
            assert e < TOL, (n, e)               # the decoder's gradients do not pass through any ReLU's backward


# --------------------------------------------------------------------------------------------- parity away from initialisation
@pytest.mark.parametrize('case', [('G3', 64, 128, 64, 1000), ('G6', 32, 192, 96, 600)], ids=['g3_64x128_1000steps', 'g6_32x192_600steps'])
def test_trained_state_step_matches_oracle(E, case):
    """Round-2 review: every oracle comparison ran on near-initialisation weights, while the fp16 x 2 arithmetic leans on operand-range
    assumptions.  Here the ENGINE trains for N steps on a rotating set of batches (weights, Adam moments and the step counter all far
    from their start), the whole state -- weights, exp_avg, exp_avg_sq, step -- is handed to the oracle's TrainState, and one further step
    is compared at the standing bars: loss 1e-5, output and every gradient element 1e-4, post-Adam weights."""
    kind, B, T, len_lo, steps = case
    c = Case(E, kind, B, T, len_lo, wseed=3, bseed=1300 + B)
    eng = c.eng
    batches = [synth_batch(1400 + i, B, T, len_lo) for i in range(4)]
    loss0 = lossN = None
    E.tune('deterministic', 1)                                 # a reproducible trained state (bit-identical from run to run)
    for i in range(steps):
        mel, f0, emb, lens = batches[i % 4]
        d = stack_draws(draws_for(20000 + i, B, c.ncalls))
        if kind == 'G3':
            loss = eng.g3_train_step(mel, f0, emb, lens, d)
        else:
            q = torch.from_numpy(interp_np.quantize_f0(f0[:, :, 0].numpy()))
            loss = eng.g6_train_step(mel, torch.nn.functional.one_hot(q, 257).float(), q, d)
        if i == 0:
            loss0 = float(loss)
    eng.check()
    E.tune('deterministic', 0)
    lossN = float(loss)
    assert lossN < 0.7 * loss0, (loss0, lossN)                 # it did train
    pv = {n: v.clone().cpu() for n, v in eng.param_views().items()}
    moved = max(float((pv[n] - p.detach()).abs().max()) for n, p in c.st.P.items())
    assert moved > 0.02, moved                                 # far from the initial weights (lr * steps = 0.06 .. 0.1 at most)
    c.st = ref_model.TrainState({n: v.numpy() for n, v in pv.items()}, LR)
    c.st.load_adam({n: v.cpu().numpy() for n, v in eng.views(eng.adam_m).items()}, {n: v.cpu().numpy() for n, v in eng.views(eng.adam_v).items()}, steps)
    p_before = {n: v.clone() for n, v in pv.items()}
    masks_for64 = {}
    orig_masks = eng.relu_masks

    def keep_masks(*a, **k):                                   # the ReLU branches Case.step hands to the fp32 oracle, kept for a float64 evaluation
        m = orig_masks(*a, **k)
        masks_for64.update({kk: v.cpu() for kk, v in m.items()})
        return m

    eng.relu_masks = keep_masks
    for kv in os.environ.get('SS_TRAINED_TUNE', '').split(','):       # diagnosis: e.g. SS_TRAINED_TUNE=bwd_f16x2=0 for the compared step only
        if '=' in kv:
            E.tune(kv.split('=')[0], int(kv.split('=')[1]))
    r = c.step(steps)
    tag = f'{kind} {B}x{T} after {steps} engine steps (loss {loss0:.4f} -> {lossN:.4f}, weights moved by up to {moved:.3f})'
    # Bars as everywhere: loss 1e-5, output and every element of every gradient tensor 1e-4 of its tensor's maximum against the fp32 oracle.
    # (Round 3 sent tensors beyond the bar to a float64 arbiter with a 16x allowance.  Round 4 removed what could be removed: the gate
    # non-linearities were 3e-7 ABSOLUTE -- 1e-6 relative for |x| ~ 0.1 .. 0.5, where cell states live, and the recurrences carry every such
    # rounding forward -- and are now good to a few ulp; the bias-type sums, fp32 chains met through atomics, now run in float64 in a fixed
    # order; the encoder BLSTMs' weight gradients are exact-product fp32 sums.  What remains is the product width of the default mode, see
    # below.)  The training run is deterministic (ss_tune("deterministic")), so the compared state is the same on every run.
    assert abs(r['loss_gpu'] - r['loss_cpu']) <= 1e-5 * abs(r['loss_cpu']), (tag, r['loss_gpu'], r['loss_cpu'])
    assert rel(r['out_gpu'], r['out_cpu']) < TOL, tag
    errs = {n: rel(r['grads_gpu'][n], g) for n, g in r['grads_cpu'].items()}
    worst = max(errs.items(), key=lambda x: x[1])
    med = sorted(errs.values())[len(errs) // 2]
    print(f'[{tag}] loss {r["loss_gpu"]:.8f} (oracle {r["loss_cpu"]:.8f}); gradient tensors vs the fp32 oracle: worst {worst[0]} {worst[1]:.2e}, median {med:.2e}')
    beyond = {n: e for n, e in errs.items() if not e < TOL}
    if beyond:
        # The default mode multiplies fp16 x 2 pieces: every operand of every product carries 22 significand bits where the reference's fp32 has
        # 24.  Measured against float64 (profiles/r04/trained_error_budget.txt) that puts the engine's trained-state gradients 2 - 10x as far from
        # exact as PyTorch-CPU's, depending on the state, and at an ill-conditioned one (a bias gradient that is the difference of two nearly
        # equal totals over B * T frames; Encoder_t's four-number W_hh gradient) a tensor can land beyond 1e-4 of the fp32 oracle.  Such a tensor
        # is accepted only on evidence that it is the stated product width and nothing else:
        #   (a) the SAME step from the SAME state with every product fp32-wide (gemm_mode 0, persist 0: bench.py's all_fp32_mfma) meets the strict
        #       bar for it -- or, where even the fp32 oracle is 5e-5 or more from float64, is no further from float64 than 3x the oracle is;
        #   (b) the default mode stays within 5e-4 of float64 (5x the bar: a hard cap, not a ratio to anything).
        # Any other excess fails.
        P64 = {n: v.double().requires_grad_(True) for n, v in p_before.items()}
        draws = draws_for(c.dseed + steps, B, c.ncalls)
        ref_model.MASK, ref_model.MASK_STATS = masks_for64, {}
        try:
            if kind == 'G3':
                xi = ref_model.interp(torch.cat((c.mel, c.f0), -1), c.lens.numpy(), draws[0], c.hp)      # resampling and re-quantisation in fp32, as the step does
                onehot, _ = ref_model.quantize_f0(xi[:, :, -1])
                out64 = ref_model.generator_3(P64, c.hp, torch.cat((xi[:, :, :-1], onehot), -1).double(), c.mel.double(), c.emb.double(), draws[1:4], training=True)
                loss64 = torch.nn.functional.mse_loss(c.mel.double(), out64, reduction='mean')
            else:
                logits = ref_model.generator_6(P64, c.hp, c.mel.double(), c.onehot.double(), draws, training=True)
                loss64 = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), c.qidx.reshape(-1))
            loss64.backward()
        finally:
            ref_model.MASK, ref_model.MASK_STATS = None, None
        # the same step, same state, all products fp32-wide (the parameters are back at p_before: Case.step applied Adam, undo it)
        for n, v in eng.param_views().items():
            v.copy_(p_before[n].to(v.device))
        E.tune('gemm_mode', 0)
        E.tune('persist', 0)
        try:
            dr = stack_draws(draws)
            if kind == 'G3':
                eng.g3_train_step(c.mel, c.f0, c.emb, c.lens, dr, no_adam=True)
            else:
                eng.g6_train_step(c.mel, c.onehot, c.qidx, dr, no_adam=True)
            eng.check()
            g_wide = {n: v.clone().cpu() for n, v in eng.grad_views().items()}
        finally:
            E.tune('gemm_mode', 1)
            E.tune('persist', 1)
        for n in sorted(beyond):
            g64 = P64[n].grad
            e_gpu, e_cpu, e_wide = rel(r['grads_gpu'][n], g64), rel(r['grads_cpu'][n], g64), rel(g_wide[n], g64)
            w_vs32 = rel(g_wide[n], r['grads_cpu'][n])
            print(f'[{tag}] {n} is {beyond[n]:.2e} from the fp32 oracle; against float64: engine {e_gpu:.2e}, engine with fp32-wide products {e_wide:.2e} '
                  f'({w_vs32:.2e} from the fp32 oracle), fp32 oracle {e_cpu:.2e}')
            assert w_vs32 < TOL or (e_cpu >= 5e-5 and e_wide <= 3.0 * e_cpu), (tag, n, 'not the product width', w_vs32, e_wide, e_cpu)
            assert e_gpu < 5e-4, (tag, n, e_gpu)
    # one Adam step from the SAME state on both sides
    tot = off = 0
    for n, pc in r['p_cpu'].items():
        pa, pb = r['p_after'][n].double(), r['p_before'][n].double()
        d = (pa - pc.double()).abs()
        assert float(d.max()) <= 2.1 * LR, (tag, n, float(d.max()))
        bad = int((d > TOL * float(pb.abs().max())).sum())
        tot += d.numel()
        off += bad
        assert bad <= max(2, 2e-5 * d.numel()), (tag, n, bad, d.numel())
    print(f'[{tag}] post-Adam weights: {off} of {tot} elements beyond 1e-4 * max|p| of their tensor')


# --------------------------------------------------------------------------------------------- fp32, Generator_6 (config 4's shape)
def test_g6_config4_shape(E):
    c = Case(E, 'G6', 32, 192, 96, wseed=4, bseed=951)
    for it in range(2):
        r = c.step(it)
        check_fp32_step(r, f'G6 32x192 step {it}')
        check_adam(r, f'G6 32x192 step {it}', it)


def test_g6_small_fixture_elementwise_and_trajectory(E):
    """The reference-generated Generator_6 fixture (B=2, T=192): logits and loss against the REFERENCE's numbers, every gradient
    element against the oracle (which the CPU suite pins to the same fixture's per-tensor statistics), two Adam steps."""
    rec = json.load(open(os.path.join(GOLD, 'g6_train.json')))
    c = Case(E, 'G6', rec['B'], rec['T'], 96, wseed=rec['wseed'], bseed=rec['bseed'])
    c.dseed = rec['dseed']
    r = c.step(0)
    assert abs(r['loss_gpu'] - rec['loss']) <= 1e-5 * rec['loss']
    assert rel(r['out_gpu'], np.load(os.path.join(GOLD, 'g6_train_logits.npy'))) < TOL
    check_fp32_step(r, 'G6 fixture step 0')
    for n, s in rec['grads'].items():        # reference statistics: sampled positions, not only norms
        flat = r['grads_gpu'][n].reshape(-1).double()
        for p, v in zip(s['pos'], s['val']):
            assert abs(float(flat[p]) - v) <= TOL * s['amax'] + 1e-12, (n, p)
    check_adam(r, 'G6 fixture step 0', 0)
    c.dseed = rec['dseed'] + 7 - 1           # any further draws: both sides consume the same ones
    r = c.step(1)
    check_fp32_step(r, 'G6 fixture step 1')


# --------------------------------------------------------------------------------------------- bf16 product mode vs the fp32 oracle
# Bounds of ss_set_precision(BF16) against the FP32 oracle (max-norm relative per tensor unless noted), DERIVED, not fitted:
#   * one bf16 rounding (nearest) is uniform in +-2^-9 of the value: rms 2^-9 / sqrt(3) = 1.1e-3; a product of two rounded operands
#     1.6e-3; a sum of K such terms with random signs keeps that relative rms error (error and sum both grow like sqrt(K));
#   * max-norm over ~10^6 output elements is ~5 sigma, and a tensor's maximum is ~3x its rms: eps1 = 1.6e-3 * 5 / 3 = 2.7e-3 per
#     contraction, relative to the tensor's maximum;
#   * the forward path chains D = 9 rounded contractions (3 conv layers, 2 encoder BLSTM projections, 3 decoder projections, head;
#     the recurrences' W_hh products stay fp32-grade) with O(1) gains in between: eps_out = eps1 * sqrt(9) = 8e-3.  Bound: x 4 for the
#     longer recurrences' accumulation over 128..192 steps and the unknown gains = 3.2e-2, stated as 4e-2;
#   * a gradient tensor is a sum of products of a forward quantity (error eps_out) and a backward one (another 9 contractions:
#     eps_out again): median tensor sqrt(2) * 8e-3 = 1.1e-2, stated as 2e-2; the WORST tensor is one whose entries cancel (small LSTM
#     biases: |sum| ~ a tenth of the terms' rms * sqrt(K)), a cancellation factor of 10 on the median bound: 2e-1;
#   * the loss is a mean over B*T*80 squared errors: its relative error is eps_out^2-order plus eps_out / sqrt(B*T*80) << 1e-3.
# Measured on MI355X (profiles/r02/parity_config_shapes.txt): loss 5e-7 .. 3e-5, output 7e-3 .. 2.1e-2 (192 frames), worst gradient
# tensor 2e-2 .. 1.0e-1, median gradient tensor 4e-3 .. 7e-3 -- all inside, none within a factor 1.9 of its bound.
BF16_BOUNDS = dict(loss=1e-3, out=4e-2, grad=2e-1, grad_median=2e-2)


@pytest.mark.parametrize('case', [('G3', 32, 128, 64), ('G6', 32, 192, 96), ('G3', 64, 192, 96), ('G3', 21, 136, 64), ('G6', 5, 104, 64)],
                         ids=['config3_g3_32x128', 'config4_g6_32x192', 'config5_g3_64x192', 'ragged_g3_21x136', 'ragged_g6_5x104'])
def test_bf16_mode_against_fp32_oracle(E, case):
    kind, B, T, len_lo = case
    c = Case(E, kind, B, T, len_lo, wseed=0 if kind == 'G3' else 4, bseed=700 + B + T, precision='bf16')
    r = c.step(0, kink_bound=5e-2)            # a bf16-product GroupNorm output may sit 1e-2 from the oracle's
    el = abs(r['loss_gpu'] - r['loss_cpu']) / abs(r['loss_cpu'])
    eo = rel(r['out_gpu'], r['out_cpu'])
    eg = {n: rel(r['grads_gpu'][n], g) for n, g in r['grads_cpu'].items()}
    worst = max(eg.items(), key=lambda x: x[1])
    med = float(np.median(list(eg.values())))
    print(f'[bf16 {kind} {B}x{T}] loss {el:.2e}  output {eo:.2e}  gradients: worst {worst[0]} {worst[1]:.2e}, median {med:.2e}')
    assert el < BF16_BOUNDS['loss'] and eo < BF16_BOUNDS['out']
    assert worst[1] < BF16_BOUNDS['grad'] and med < BF16_BOUNDS['grad_median']
    assert worst[1] > 1e-4                    # really the reduced-precision arithmetic
    # the resampling index path never touches the contractions: bit-exact in this mode as well
    if kind == 'G3':
        draws = draws_for(c.dseed, B, 4)
        xi = ref_model.interp(torch.cat((c.mel, c.f0), -1), c.lens.numpy(), draws[0], c.hp)
        assert np.array_equal(c.eng.debug_buffer('in.mel', B, T).cpu().numpy(), xi[:, :, :80].numpy())


# --------------------------------------------------------------------------------------------- per-block reference vectors (fixture set F2)
def test_conv_block_against_reference_vectors(E):
    """tests/golden/blocks.npz holds the REFERENCE's conv block (Encoder_t's 80 -> 128) forward and backward vectors; they
    go through the engine's conv_block_fwd / conv_block_bwd (ss_op_conv_block), so a whole-model failure localises."""
    z = np.load(os.path.join(GOLD, 'blocks.npz'))
    w = W.make_weights('G3', W.default_hparams(), 3)
    pre = 'encoder_2.convolutions.0'
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x = t(z['conv_x'].transpose(0, 2, 1))
    dy = t(z['conv_gy'].transpose(0, 2, 1))
    y, dx, gw, gb, gg, gbe = E.conv_block(x, t(w[pre + '.0.conv.weight']), t(w[pre + '.0.conv.bias']), t(w[pre + '.1.weight']),
                                          t(w[pre + '.1.bias']), dy)
    assert rel(y, z['conv_y'].transpose(0, 2, 1)) < TOL
    assert rel(dx, z['conv_gx'].transpose(0, 2, 1)) < TOL
    assert rel(gw, z['conv_gw']) < TOL and rel(gb, z['conv_gb']) < TOL
    assert rel(gg, z['conv_ggamma']) < TOL and rel(gbe, z['conv_gbeta']) < TOL


@pytest.mark.parametrize('blk', [('lstm_t', 'encoder_2.lstm', 1), ('lstm_1', 'encoder_1.lstm_1', 2), ('lstm_2', 'encoder_1.lstm_2', 1),
                                 ('lstm_d', 'decoder.lstm', 3)], ids=['h1', 'h8x2', 'h32', 'h512x3'])
@pytest.mark.parametrize('persist', [1, 0])
def test_blstm_blocks_against_reference_vectors(E, blk, persist):
    """Every BLSTM shape on the path, the REFERENCE's module outputs and gradients (blocks.npz) through ss_op_lstm_fwd /
    ss_op_lstm_bwd: small single-launch kernels (H = 1, 8, 32), and for H = 512 the persistent and the per-step kernels."""
    name, prefix, layers = blk
    if persist == 0 and name != 'lstm_d':
        pytest.skip('one schedule for the small recurrences')
    z = np.load(os.path.join(GOLD, 'blocks.npz'))
    w = W.make_weights('G3', W.default_hparams(), 3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    par = lambda l: tuple((t(w[f'{prefix}.{k}_l{l}']), t(w[f'{prefix}.{k}_l{l}_reverse'])) for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'))
    E.tune('persist', persist)
    try:
        xs = [t(z[f'{name}_x'])]
        for l in range(layers):
            xs.append(E.blstm_layer(xs[l], *par(l)))
        assert rel(xs[-1], z[f'{name}_y']) < TOL
        d = t(z[f'{name}_gy'])
        for l in range(layers - 1, -1, -1):
            _, d, grads = E.blstm_layer(xs[l], *par(l), d_out=d)
        assert rel(d, z[f'{name}_gx']) < TOL
        assert rel(grads[0][1][:96, :96], z[f'{name}_gwhh0']) < TOL       # layer 0, forward direction, top-left corner
        assert rel(grads[1][2], z[f'{name}_gbih0r']) < TOL                # layer 0, reverse direction, bias
    finally:
        E.tune('persist', 1)
    torch.cuda.synchronize()
