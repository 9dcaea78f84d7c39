"""CPU-only checks of the C-ABI library and of the host-side mirror of the reference's Python surface.
No kernel is launched here (there is no GPU in the build container)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import interp_np, weights as W
from oracle.gen_fixtures import draws_for
from speechsplit_amd import _capi, data_loader, dist, hparams as HP, utils
from speechsplit_amd.engine import draw_interp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _capi.lib()
    header = open(os.path.join(ROOT, 'include', 'speechsplit_amd.h')).read()
    declared = set(re.findall(r'\b(ss_[a-z0-9_]+)\s*\(', header))
    declared -= {'ss_engine', 'ss_hparams'}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ss_abi_version() == 2


@pytest.mark.parametrize('kind', ['G3', 'G6'])
def test_param_table_matches_reference(gold_dir, kind):
    lib = _capi.lib()
    hp = HP.default_hparams()
    hps = _capi.hparams_struct(hp)
    h = lib.ss_create(3 if kind == 'G3' else 6, C.byref(hps), 4, 192)
    assert h
    ref = json.load(open(os.path.join(gold_dir, f'keys_{kind}.json')))
    shapes = dict(zip(ref['keys'], ref['shapes']))
    name = C.create_string_buffer(256)
    off, nd, shp = C.c_long(), C.c_int(), (C.c_long * 3)()
    names, total, prev_end = [], 0, 0
    for i in range(lib.ss_num_params(h)):
        assert lib.ss_param_info(h, i, name, 256, C.byref(off), C.byref(nd), C.byref(shp)) == 0
        n = name.value.decode()
        s = [shp[k] for k in range(nd.value)]
        assert s == shapes[n], n
        assert off.value % 4 == 0 and off.value >= prev_end          # 16-byte aligned, non-overlapping, ordered
        prev_end = off.value + int(np.prod(s))
        total += int(np.prod(s))
        names.append(n)
    assert names == ref['params']                                     # parameters() order of the reference
    assert total == ref['numel']
    assert lib.ss_arena_numel(h) >= prev_end
    assert lib.ss_workspace_bytes(h) > 0
    lib.ss_destroy(h)


def test_create_rejects_bad_arguments():
    lib = _capi.lib()
    hps = _capi.hparams_struct(HP.default_hparams())
    assert not lib.ss_create(5, C.byref(hps), 4, 192)
    assert b'kind' in lib.ss_last_error()
    bad = _capi.hparams_struct(HP.default_hparams(chs_grp=8))
    assert not lib.ss_create(3, C.byref(bad), 4, 192)
    assert not lib.ss_create(3, C.byref(hps), 0, 192)


def test_unbound_engine_fails_loudly():
    lib = _capi.lib()
    hps = _capi.hparams_struct(HP.default_hparams())
    h = lib.ss_create(3, C.byref(hps), 2, 192)
    assert lib.ss_adam_step(h, 1.0, None) != 0
    assert lib.ss_g3_forward(h, None, None, None, None, None, 2, 192, 0, None, None) != 0
    assert b'bound' in lib.ss_last_error()
    lib.ss_destroy(h)


def test_engine_refuses_to_run_without_gpu():
    from speechsplit_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError):
        Engine('G3', HP.default_hparams(), 2)


def test_hparams_bag():
    hp = HP.default_hparams()
    assert hp.max_len_pad == 192 and hp.dim_f0 == 257 and hp.freq == 8
    hp2 = hp.copy().parse('max_len_pad=128,batch_size=64')
    assert hp2.max_len_pad == 128 and hp2.batch_size == 64 and hp.max_len_pad == 192
    assert 'max_len_pad' in HP.hparams_debug_string()
    with pytest.raises(ValueError):
        hp.copy().parse('nope=1')


def test_quantizers_match_reference_fixture(gold_dir):
    z = np.load(os.path.join(gold_dir, 'quantize.npz'))
    enc, idx = utils.quantize_f0_torch(torch.from_numpy(z['x']))
    assert np.array_equal(idx.numpy(), z['idx'])
    assert np.array_equal(enc.argmax(-1).numpy(), z['onehot_argmax'])
    e2, i2 = utils.quantize_f0_numpy(z['x'][1])
    assert np.array_equal(i2, z['idx'][1]) and e2.shape == (64, 257)
    x, pad = utils.pad_seq_to_2(np.ones((1, 100, 80), np.float32), 192)
    assert x.shape == (1, 192, 80) and pad == 92


def test_draws_replay_reference_rng_order():
    """speechsplit_amd.engine.draw_interp must consume the CPU generator exactly as the reference does
    (rand(B*7) then randint(19,32,(B*7,1)) per InterpLnr call) -- checked against the fixture generator's replay."""
    hp = HP.default_hparams()
    ref = draws_for(77, 6, 4)
    torch.manual_seed(77)
    sc, ls = draw_interp(6, 4, hp)
    for i in range(4):
        assert np.array_equal(sc[i].numpy(), ref[i][0])
        assert np.array_equal(ls[i].numpy(), ref[i][1].astype(np.int32))


@pytest.mark.parametrize('kind', ['G3', 'G6'])
def test_module_state_dict_matches_reference(gold_dir, kind):
    from speechsplit_amd import model
    ref = json.load(open(os.path.join(gold_dir, f'keys_{kind}.json')))
    cls = model.Generator_3 if kind == 'G3' else model.Generator_6
    torch.manual_seed(0)
    m = cls(HP.default_hparams())
    sd = m.state_dict()
    assert list(sd.keys()) == ref['keys']
    assert [list(v.shape) for v in sd.values()] == ref['shapes']
    assert [n for n, _ in m.named_parameters()] == ref['params']
    assert m.train() is m and m.eval() is m
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 192, 337), torch.zeros(1, 192, 80), torch.zeros(1, 82)) if kind == 'G3' else \
            m(torch.zeros(1, 192, 80), torch.zeros(1, 192, 257))
    # an optimizer built before .to(device) (solver.py:62-65) must see the same Parameter objects afterwards
    ids = [id(p) for p in m.parameters()]
    m.float()
    assert ids == [id(p) for p in m.parameters()]


def test_module_init_follows_reference_distributions(gold_dir):
    """Initialiser bounds (xavier-uniform gain sqrt(2) for convs, 1 for the head, U(+-1/sqrt(hidden)) for LSTMs,
    GroupNorm (1, 0)) against statistics of the reference constructed with torch.manual_seed(0)."""
    from speechsplit_amd import model
    ref = json.load(open(os.path.join(gold_dir, 'init_seed0_G3.json')))
    torch.manual_seed(0)
    m = model.Generator_3(HP.default_hparams())
    for n, p in m.named_parameters():
        a = p.detach()
        r = ref[n]
        if n.endswith('.1.weight'):
            assert float(a.min()) == 1.0 == float(a.max())
        elif n.endswith('.1.bias'):
            assert float(a.abs().max()) == 0.0
        else:
            k = a.numel()
            if k >= 256:
                assert abs(float(a.abs().max()) - r['amax']) <= 0.05 * r['amax'] + 1e-3, n  # same uniform bound
            if k >= 16:
                assert abs(float(a.double().norm()) - r['l2']) <= (0.03 + 2.0 / k ** 0.5) * r['l2'], n   # same variance


def test_collator_and_loader_contract():
    hp = HP.default_hparams(batch_size=4)
    np.random.seed(0)
    loader = data_loader.get_loader(hp, dataset=data_loader.SyntheticUtterances(16, seed=1))
    mel, emb, f0, ln = next(iter(loader))
    assert mel.shape == (4, 192, 80) and emb.shape == (4, 82) and f0.shape == (4, 192, 1) and ln.shape == (4,)
    assert mel.dtype == torch.float32 and ln.dtype == torch.int64
    assert float(mel.min()) >= 0 and float(mel.max()) <= 1
    for b in range(4):
        L = int(ln[b])
        assert 64 <= L <= 128
        assert (f0[b, L:] == -1e10).all() and (mel[b, L:] == 0).all()
    assert len(data_loader.MultiSampler(5, 8)) == 40


def test_shard_helpers():
    lo, hi = dist.shard_range(8, 1, 2)
    assert (lo, hi) == (4, 8)
    with pytest.raises(ValueError):
        dist.shard_range(7, 0, 2)
    sc = torch.arange(4 * 8 * 7, dtype=torch.float32).view(4, 56)
    ls = sc.to(torch.int32)
    a, b = dist.shard_draws(sc, ls, 8, 1, 2)
    assert a.shape == (4, 28) and torch.equal(a, sc[:, 28:]) and torch.equal(b, ls[:, 28:])
    batch = (torch.zeros(8, 3), torch.zeros(8, 2), torch.zeros(8, 1), torch.arange(8))
    assert dist.shard_batch(batch, 0, 4)[3].tolist() == [0, 1]


def test_mel_filter_bank_published_properties():
    """speechsplit_amd.features.mel_filter_bank restates `librosa.filters.mel` (make_spect_f0.py:15) from the published algorithm; librosa is
    absent here, so parity is UNPINNED and only the published properties are checked: the Slaney mel scale's fixed points, triangular
    filters between neighbouring band edges, non-negative weights, unit area under every filter ('slaney' norm), peaks in band order."""
    import numpy as np
    from speechsplit_amd import features as F
    assert abs(float(F._hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(F._hz_to_mel(200.0)) - 3.0) < 1e-12
    assert abs(float(F._hz_to_mel(6400.0)) - 42.0) < 1e-9                      # 27 mels per factor of 6.4 above 1 kHz
    for f in (90.0, 999.0, 1000.0, 1001.0, 7600.0):
        assert abs(float(F._mel_to_hz(F._hz_to_mel(f))) - f) < 1e-9 * f
    w = F.mel_filter_bank()
    assert w.shape == (80, 513) and w.dtype == np.float32 and float(w.min()) >= 0.0
    fft_f = np.linspace(0, 8000, 513)
    edges = F._mel_to_hz(np.linspace(F._hz_to_mel(90.0), F._hz_to_mel(7600.0), 82))
    peaks = []
    for i in range(80):
        nz = np.nonzero(w[i])[0]
        assert len(nz) >= 1 and fft_f[nz[0]] > edges[i] - 1e-9 and fft_f[nz[-1]] < edges[i + 2] + 1e-9      # support inside its band
        peaks.append(int(w[i].argmax()))
        # a triangle of height 2 / width sampled every 15.625 Hz: the area under the continuous filter is 1
        width = edges[i + 2] - edges[i]
        assert abs(float(w[i].max()) - 2.0 / width) <= 2.0 / width * (15.625 / min(edges[i + 1] - edges[i], edges[i + 2] - edges[i + 1])) + 1e-6
    assert peaks == sorted(peaks)
    assert float(w[:, fft_f < 90.0].max()) == 0.0 and float(w[:, fft_f > 7600.0].max()) == 0.0


def test_mel_filter_bank_closed_form_slaney_values():
    """Pin of speechsplit_amd.features.mel_filter_bank against hand-computed values of the published definition (librosa.filters.mel with its
    defaults: Slaney mel scale, 'slaney' area norm; make_spect_f0.py:15 calls it with sr 16000, n_fft 1024, fmin 90, fmax 7600, n_mels 80).
    librosa itself is absent here, so this is a pin against the closed form, not against the reference's output -- it says so, and it is
    the strongest pin this environment allows.  Derivation (scalar arithmetic, independent of the function under test):
      mel(90) = 90 / (200/3) = 1.35;  mel(7600) = 15 + 27 ln(7.6) / ln(6.4) = 44.4995755;  82 edges, step 0.53271081 mel.
      filter 0 (linear region): edges 90, 125.5141, 161.0281 Hz -> bins 6..10 (k x 15.625 Hz), height 2 / 71.0281 = 0.0281579
          bin 8 (125.0 Hz): (125 - 90) / 35.5141 x 0.0281579 = 0.02775029
      filter 25 (straddles 1 kHz, where the scale turns logarithmic): edges 977.8513, 1013.8789, 1051.7004 Hz -> bins 63..67
      filter 79 (top of the log region): edges 7063.2029, 7326.6870, 7600 Hz -> bins 453..486, peak at bin 469."""
    import numpy as np
    from speechsplit_amd import features as F
    w = F.mel_filter_bank()
    want = {
        0: {6: 0.002973245, 7: 0.01536177, 8: 0.02775029, 9: 0.01617692, 10: 0.003788397},
        25: {63: 0.004903904, 64: 0.01664939, 65: 0.02583195, 66: 0.01464359, 67: 0.003455222},
        79: {453: 0.0002110063, 460: 0.001757626, 468: 0.003525191, 469: 0.003706199, 470: 0.003493199, 480: 0.0013632, 486: 8.519999e-05},
    }
    for i, vals in want.items():
        nz = np.nonzero(w[i])[0]
        if i != 79:
            assert list(nz) == sorted(vals), (i, list(nz))
        else:
            assert nz[0] == 453 and nz[-1] == 486 and len(nz) == 34 and int(w[i].argmax()) == 469
        for k, v in vals.items():
            assert abs(float(w[i, k]) - v) <= 2e-6 * v + 1e-12, (i, k, float(w[i, k]), v)


def test_roofline_traffic_profile_matches_kernel_sources():
    """bench.py reports `roofline.traffic` from the committed rocprofv3 --pmc passes only while the GEMM kernel sources are the ones those
    counters were read from (tools/kernel_sha.py); this test makes a stale profile a red test here instead of a silent null in the bench line."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    from kernel_sha import kernel_sources_sha
    recs = json.load(open(os.path.join(root, 'profiles', 'r04', 'gemm_pmc.json')))
    assert recs and all(r.get('kernel_sources_sha') == kernel_sources_sha() for r in recs), \
        'profiles/r04/gemm_pmc.json is stale: rerun tools/run_profiles.sh (the PMC passes) after changing gemm_*.hip / common.h'
    dw = [r for r in recs if r['class'] == 'dec_dw']
    assert dw and all(r['hbm_read_bytes'] > 0 and r['hbm_write_bytes'] > 0 for r in dw)


def test_ss_tune_env_hook_applies_knobs_at_load():
    """SS_TUNE="key=value,..." is applied once when the library is loaded (A/B runs of tests and tools); an unknown key is an error, not a silently
    ignored typo.  (ss_tune only sets process-global knobs: no GPU needed.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = 'from speechsplit_amd import _capi; _capi.lib(); print("loaded")'
    ok = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, SS_TUNE='dec_tail_split=2, conv_dw_off=0'), capture_output=True, text=True)
    assert ok.returncode == 0 and 'loaded' in ok.stdout, ok.stderr[-500:]
    bad = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, SS_TUNE='no_such_knob=1'), capture_output=True, text=True)
    assert bad.returncode != 0 and 'SS_TUNE' in bad.stderr and 'unknown key' in bad.stderr
