"""Pins the CPU oracle (oracle/) against vectors produced by the imported reference
(oracle/gen_fixtures.py -> tests/golden).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import interp_np, ref_model, weights as W
from oracle.gen_fixtures import draws_for, synth_batch

torch.set_num_threads(4)


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _stats(t):
    a = t.detach().double().reshape(-1)
    return float(a.sum()), float(a.norm())


@pytest.mark.parametrize('kind', ['G3', 'G6'])
def test_param_table_matches_reference_state_dict(gold_dir, kind):
    ref = json.load(open(os.path.join(gold_dir, f'keys_{kind}.json')))
    spec = W.param_spec(kind, W.default_hparams())
    assert [n for n, _ in spec] == ref['params']
    shapes = dict(zip(ref['keys'], ref['shapes']))
    for n, s in spec:
        assert list(s) == shapes[n], n
    assert sum(int(np.prod(s)) for _, s in spec) == ref['numel']
    assert [k for k in ref['keys'] if k not in dict(spec)] == W.buffer_spec(kind)
    assert ref['numel'] == (19437800 if kind == 'G3' else 3485849)


def test_interp_bit_exact(gold_dir):
    z = np.load(os.path.join(gold_dir, 'interp.npz'))
    for i in range(int(z['n'])):
        pad = int(z[f'c{i}_max_len_pad'])
        y = interp_np.interp_forward(z[f'c{i}_x'], z[f'c{i}_len_seq'], z[f'c{i}_scales'], z[f'c{i}_len_seg'],
                                     max_len_pad=pad)
        ref = z[f'c{i}_y']
        assert y.shape == ref.shape
        assert np.array_equal(y, ref), f'case {i}: max diff {np.abs(y - ref).max()}'
        # torch flavour used inside the model oracle
        hp = W.default_hparams(max_len_pad=pad)
        yt = ref_model.interp(torch.from_numpy(z[f'c{i}_x']), z[f'c{i}_len_seq'],
                              (z[f'c{i}_scales'], z[f'c{i}_len_seg']), hp).numpy()
        assert np.array_equal(yt, ref)


def test_interp_truncation_case_present(gold_dir):
    z = np.load(os.path.join(gold_dir, 'interp.npz'))
    seen = False
    for i in range(int(z['n'])):
        pad = int(z[f'c{i}_max_len_pad'])
        _, _, counts, nrows = interp_np.interp_plan(z[f'c{i}_scales'], z[f'c{i}_len_seg'], z[f'c{i}_len_seq'],
                                                    max_len_pad=pad)
        seen |= bool((counts > pad).any())
        assert (nrows <= pad).all()
    assert seen, 'fixtures should include count > max_len_pad'


def test_interp_backward_is_adjoint():
    rs = np.random.RandomState(0)
    B, T, C = 3, 128, 5
    x = rs.randn(B, T, C).astype(np.float32)
    sc = (rs.rand(B * 7) + 0.5).astype(np.float32)
    sg = rs.randint(19, 32, B * 7)
    i0, lam, _, nrows = interp_np.interp_plan(sc, sg, [128, 100, 64], max_len_pad=128)
    y = interp_np.interp_apply(x, i0, lam, nrows)
    dy = rs.randn(*y.shape).astype(np.float32)
    dx = interp_np.interp_backward(dy, i0, lam, nrows, T)
    assert abs(float((y.astype(np.float64) * dy).sum()) - float((x.astype(np.float64) * dx).sum())) < 1e-3


def test_quantize(gold_dir):
    z = np.load(os.path.join(gold_dir, 'quantize.npz'))
    idx = interp_np.quantize_f0(z['x'])
    assert np.array_equal(idx, z['idx'])
    assert np.array_equal(idx, z['onehot_argmax'])
    assert (z['onehot_sum'] == 1).all()
    oh = interp_np.onehot(idx)
    assert oh.shape[-1] == 257 and (oh.sum(-1) == 1).all()


def test_blocks(gold_dir):
    z = np.load(os.path.join(gold_dir, 'blocks.npz'))
    hp = W.default_hparams()
    P = ref_model.as_params(W.make_weights('G3', hp, 3))
    x = torch.from_numpy(z['conv_x']).requires_grad_(True)
    y = ref_model.conv_gn_relu(x, P, 'encoder_2.convolutions.0')
    assert _rel(y.detach().numpy(), z['conv_y']) < 1e-6
    y.backward(torch.from_numpy(z['conv_gy']))
    assert _rel(x.grad.numpy(), z['conv_gx']) < 1e-5
    assert _rel(P['encoder_2.convolutions.0.0.conv.weight'].grad.numpy(), z['conv_gw']) < 1e-5
    assert _rel(P['encoder_2.convolutions.0.0.conv.bias'].grad.numpy(), z['conv_gb']) < 1e-5
    assert _rel(P['encoder_2.convolutions.0.1.weight'].grad.numpy(), z['conv_ggamma']) < 1e-5
    assert _rel(P['encoder_2.convolutions.0.1.bias'].grad.numpy(), z['conv_gbeta']) < 1e-5
    for name, prefix, layers in (('lstm_t', 'encoder_2.lstm', 1), ('lstm_1', 'encoder_1.lstm_1', 2),
                                 ('lstm_2', 'encoder_1.lstm_2', 1), ('lstm_d', 'decoder.lstm', 3)):
        for fn in (ref_model.blstm, ref_model.lstm_explicit):
            for p in P.values():
                p.grad = None
            x = torch.from_numpy(z[f'{name}_x']).requires_grad_(True)
            y = fn(x, P, prefix, layers)
            assert _rel(y.detach().numpy(), z[f'{name}_y']) < 2e-6, (name, fn.__name__)
            y.backward(torch.from_numpy(z[f'{name}_gy']))
            assert _rel(x.grad.numpy(), z[f'{name}_gx']) < 1e-5, (name, fn.__name__)
            assert _rel(P[prefix + '.weight_hh_l0'].grad.numpy()[:96, :96], z[f'{name}_gwhh0']) < 1e-5
            assert _rel(P[prefix + '.bias_ih_l0_reverse'].grad.numpy(), z[f'{name}_gbih0r']) < 1e-5


def test_demo_config1(gold_dir):
    """BASELINE config 1: demo.pkl utterances, eval forward (solver.py:206-221, demo.ipynb cell 0)."""
    z = np.load(os.path.join(gold_dir, 'demo_config1.npz'))
    hp = W.default_hparams()
    P3 = ref_model.as_params(W.make_weights('G3', hp, int(z['seed_g3'])), False)
    P6 = ref_model.as_params(W.make_weights('G6', hp, int(z['seed_g6'])), False)
    for n in range(2):
        mel = torch.from_numpy(z[f'u{n}_mel_pad'])
        qidx = interp_np.quantize_f0(z[f'u{n}_f0_pad'])
        assert np.array_equal(qidx, z[f'u{n}_qidx'].astype(np.int64))
        onehot = torch.from_numpy(interp_np.onehot(qidx))[None]
        emb = torch.from_numpy(z[f'u{n}_emb'])
        with torch.no_grad():
            out3 = ref_model.generator_3(P3, hp, torch.cat((mel, onehot), -1), mel, emb)
            rhythm = ref_model.encoder_t(mel.transpose(2, 1), P3, hp)
            out6 = ref_model.generator_6(P6, hp, mel, onehot)
        assert out3.shape == (1, 192, 80) and out6.shape == (1, 192, 257)
        assert _rel(out3.numpy(), z[f'u{n}_out3']) < 1e-6
        assert _rel(rhythm.numpy(), z[f'u{n}_rhythm']) < 1e-6
        assert _rel(out6.numpy(), z[f'u{n}_out6']) < 1e-6


def test_demo_conversion_conditions(gold_dir):
    """demo.ipynb cell 0: F0 conversion through Generator_6 and the seven conversion conditions, oracle vs reference."""
    z, c = np.load(os.path.join(gold_dir, 'demo_config1.npz')), np.load(os.path.join(gold_dir, 'demo_conversion.npz'))
    hp = W.default_hparams()
    P3 = ref_model.as_params(W.make_weights('G3', hp, int(z['seed_g3'])), False)
    P6 = ref_model.as_params(W.make_weights('G6', hp, int(z['seed_g6'])), False)
    mel = [torch.from_numpy(z[f'u{n}_mel_pad']) for n in range(2)]
    oh = [torch.from_numpy(interp_np.onehot(z[f'u{n}_qidx'].astype(np.int64)))[None] for n in range(2)]
    emb = [torch.from_numpy(z[f'u{n}_emb']) for n in range(2)]
    lens = [int(z[f'u{n}_len']) for n in range(2)]
    with torch.no_grad():
        logits = ref_model.generator_6(P6, hp, mel[0], oh[1])[0]
        assert _rel(logits.numpy(), c['f0_logits']) < 1e-6
        q = logits.argmax(-1)
        assert np.array_equal(q.numpy(), c['f0_pred_idx'].astype(np.int64))
        oh_con = torch.nn.functional.one_hot(q, 257).float()[None]
        for cond in ['R', 'F', 'U', 'RF', 'RU', 'FU', 'RFU']:
            y = ref_model.generator_3(P3, hp, torch.cat((mel[0], oh_con if 'F' in cond else oh[0]), -1),
                                      mel[1] if 'R' in cond else mel[0], emb[1] if 'U' in cond else emb[0])
            keep = lens[1] if 'R' in cond else lens[0]
            assert _rel(y[0, :keep].numpy(), c[f'out_{cond}']) < 1e-6, cond


def _synth(seed, B, T, lo):
    from oracle.gen_fixtures import synth_batch
    return synth_batch(seed, B, T, lo)


def _draws(seed, B, n):
    from oracle.gen_fixtures import draws_for
    return draws_for(seed, B, n)


@pytest.mark.parametrize('tag', ['b2_t128', 'b2_t192'])
def test_train_step(gold_dir, tag):
    rec = json.load(open(os.path.join(gold_dir, 'train_steps.json')))[tag]
    B, T = rec['B'], rec['T']
    hp = W.default_hparams(max_len_pad=T)
    st = ref_model.TrainState(W.make_weights('G3', hp, rec['wseed']))
    mel, f0, emb, lens = _synth(rec['bseed'], B, T, 64 if T == 128 else 96)
    nsteps = len(rec['losses'])
    draws = _draws(rec['dseed'], B, 4 * nsteps)
    for it in range(nsteps):
        loss, out = ref_model.g3_loss(st.P, hp, mel, f0, emb, lens.numpy(), draws[4 * it:4 * it + 4])
        st.opt.zero_grad()
        loss.backward()
        if it == 0:
            assert _rel(out.detach().numpy(), np.load(os.path.join(gold_dir, f'train_{tag}_out.npy'))) < 1e-5
            for n, s in rec['grads'].items():
                g = st.P[n].grad
                gs, gl = _stats(g)
                assert abs(gl - s['l2']) <= 1e-4 * s['l2'] + 1e-12, n
                flat = g.reshape(-1)
                for p, v in zip(s['pos'], s['val']):
                    assert abs(float(flat[p]) - v) <= 1e-4 * s['amax'] + 1e-12, (n, p)
        st.opt.step()
        if it == 0:
            for n, s in rec['params_after'].items():
                flat = st.P[n].detach().reshape(-1)
                for p, v in zip(s['pos'], s['val']):
                    assert abs(float(flat[p]) - v) <= 1e-6 * s['amax'] + 1e-9, (n, p)
        assert abs(float(loss) - rec['losses'][it]) <= 1e-5 * abs(rec['losses'][it]), it


def test_g6_train_forward_and_ce(gold_dir):
    rec = json.load(open(os.path.join(gold_dir, 'g6_train.json')))
    hp = W.default_hparams(max_len_pad=rec['T'])
    P = ref_model.as_params(W.make_weights('G6', hp, rec['wseed']))
    mel, f0, emb, lens = _synth(rec['bseed'], rec['B'], rec['T'], 96)
    qidx = torch.from_numpy(interp_np.quantize_f0(f0[:, :, 0].numpy()))
    onehot = torch.nn.functional.one_hot(qidx, 257).float()
    draws = _draws(rec['dseed'], rec['B'], 3)
    loss, logits = ref_model.g6_loss(P, hp, mel, onehot, qidx, draws)
    assert _rel(logits.detach().numpy(), np.load(os.path.join(gold_dir, 'g6_train_logits.npy'))) < 1e-5
    assert abs(float(loss) - rec['loss']) < 1e-5 * rec['loss']
    loss.backward()
    for n, s in rec['grads'].items():
        gl = float(P[n].grad.double().norm())
        assert abs(gl - s['l2']) <= 1e-4 * s['l2'] + 1e-12, n


def test_relu_mask_override_is_neutral_with_own_branches():
    """ref_model.MASK (the hook the GPU parity tests use to hand the oracle the engine's ReLU branches): fed the oracle's OWN
    branches it must change nothing -- same loss, same gradients, zero disagreements -- and a flipped element must be reported."""
    hp = W.default_hparams(max_len_pad=64)
    w = W.make_weights('G3', hp, 3)
    mel, f0, emb, lens = synth_batch(5, 2, 64, 64)
    draws = draws_for(6, 2, 4)
    P = ref_model.as_params(w)
    ref_model.TAP = {}
    l0, _ = ref_model.g3_loss(P, hp, mel, f0, emb, lens.numpy(), draws)
    tap, ref_model.TAP = ref_model.TAP, None
    l0.backward()
    g0 = {n: p.grad.clone() for n, p in P.items()}
    # the oracle's own branches: GroupNorm of the tapped conv outputs
    masks = {}
    for k, v in tap.items():
        if k.endswith('.conv') and not k.startswith('zmin:'):
            pre = {'enc2.c': 'encoder_2.convolutions.0'}.get(k[:-5]) or \
                f"encoder_1.convolutions_{k[6]}.{k[8]}"          # 'enc1.c<stream>_<layer>.conv'
            z = torch.nn.functional.group_norm(v.transpose(1, 2), v.shape[-1] // 16, P[pre + '.1.weight'].detach(),
                                               P[pre + '.1.bias'].detach(), eps=1e-5)
            masks[k[:-5]] = (z > 0).transpose(1, 2)
    assert len(masks) == 7
    P2 = ref_model.as_params(w)
    ref_model.MASK, ref_model.MASK_STATS = masks, {}
    try:
        l1, _ = ref_model.g3_loss(P2, hp, mel, f0, emb, lens.numpy(), draws)
        stats = dict(ref_model.MASK_STATS)
        l1.backward()
        assert float(l1) == float(l0) and all(v == (0, 0.0) for v in stats.values()), stats
        for n, p in P2.items():
            assert torch.equal(p.grad, g0[n]), n
        flipped = {k: v.clone() for k, v in masks.items()}
        flipped['enc2.c'][0, 0, 0] = ~flipped['enc2.c'][0, 0, 0]
        ref_model.MASK, ref_model.MASK_STATS = flipped, {}
        with torch.no_grad():
            ref_model.g3_loss(P2, hp, mel, f0, emb, lens.numpy(), draws)
        assert ref_model.MASK_STATS['enc2.c'][0] == 1 and ref_model.MASK_STATS['enc2.c'][1] > 0
    finally:
        ref_model.MASK, ref_model.MASK_STATS = None, None


def test_collator_against_reference_fixture(gold_dir):
    """speechsplit_amd.data_loader.MyCollator against the REFERENCE collator's output
    (tests/golden/collate.npz, generated by oracle/gen_fixtures.py from reference data_loader.py:101-128): same numpy seed,
    same items -> the same crops, clipping, padding (mel with 0, F0 with -1e10) and lengths, bit for bit."""
    from speechsplit_amd import data_loader as DL, hparams as HPM
    z = np.load(os.path.join(gold_dir, 'collate.npz'))
    ds = DL.SyntheticUtterances(int(z['corpus_n']), seed=int(z['corpus_seed']))
    np.random.seed(int(z['np_seed']))
    mel, emb, f0, ln = DL.MyCollator(HPM.default_hparams())([ds[int(i)] for i in z['items']])
    assert mel.dtype == torch.float32 and ln.dtype == torch.int64 and f0.shape == (6, 192, 1)
    assert np.array_equal(mel.numpy(), z['mel']) and np.array_equal(emb.numpy(), z['emb'])
    assert np.array_equal(f0.numpy(), z['f0']) and np.array_equal(ln.numpy(), z['len_org'])


def test_feature_preprocessing_against_reference_fixture(gold_dir):
    """SURVEY.md section 8(f) row N4, host half: the 30 Hz Butterworth filtfilt + 0.96 scaling + per-speaker dither of
    make_spect_f0.py:49-54 (speechsplit_amd.features.preprocess_wav) reproduces the waveform the reference's functions gave
    (tests/golden/features.npz), including the one-sample append for lengths that are multiples of 256."""
    from speechsplit_amd import features as F
    z = np.load(os.path.join(gold_dir, 'features.npz'))
    for u in range(2):
        wav = F.preprocess_wav(z[f'u{u}_x'], np.random.RandomState(int(z[f'u{u}_spk'])))
        assert wav.shape == z[f'u{u}_wav'].shape and np.array_equal(wav, z[f'u{u}_wav'])
    b, a = F.butter_highpass(30, 16000, order=5)
    assert b.shape == (6,) and abs(a[0] - 1.0) < 1e-12
