"""Length buckets (BASELINE config 5) and the data-parallel step variants on the GPU (pytest -m gpu)."""
import numpy as np
import pytest
import torch

from oracle import interp_np, ref_model, weights as W
from oracle.gen_fixtures import draws_for, synth_batch
from conftest import assert_same_trajectory

pytestmark = pytest.mark.gpu
TOL = 1e-4


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


def test_bucketed_steps_equal_reference_run_per_bucket(E):
    """the oracle of a length bucket is the reference step with max_len_pad = T_bucket.  ONE engine (max_frames
    192) walks buckets 104 -> 152 -> 192 -> 104 with Adam in between; every step is compared with the oracle TrainState doing
    the same with hparams.max_len_pad = T: loss, output, every gradient element (the engine's ReLU branches handed over)."""
    B = 8
    hp192 = W.default_hparams(max_len_pad=192)
    w = W.make_weights('G3', hp192, 5)
    eng = E.Engine('G3', hp192, B, 192)
    eng.load_weights(w)
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    st = ref_model.TrainState(w)
    for it, T in enumerate((104, 152, 192, 104)):
        hp = W.default_hparams(max_len_pad=T)
        mel, f0, emb, lens = synth_batch(300 + it, B, T, T - 7)              # crop lengths inside the bucket: (T - 8, T]
        assert int(lens.min()) > T - 8
        draws = draws_for(400 + it, B, 4)
        loss = eng.g3_train_step(mel, f0, emb, lens, stack_draws(draws), no_adam=True, bucket=True)
        eng.check()
        out = eng.debug_buffer('out', B, T).cpu()
        grads = {n: v.clone().cpu() for n, v in eng.grad_views().items()}
        masks = {k: v.cpu() for k, v in eng.relu_masks(B, T).items()}
        eng.adam_step()
        ref_model.MASK, ref_model.MASK_STATS = masks, {}
        try:
            lo, ro = st.step_g3(hp, mel, f0, emb, lens.numpy(), draws)
            stats = ref_model.MASK_STATS
        finally:
            ref_model.MASK, ref_model.MASK_STATS = None, None
        assert all(z < 2e-5 for _, z in stats.values()), stats
        assert abs(float(loss) - float(lo)) <= 1e-5 * float(lo), (T, float(loss), float(lo))
        assert rel(out, ro) < TOL, T
        for n, p in st.P.items():
            assert rel(grads[n], p.grad) < TOL, (T, n)
        xi = ref_model.interp(torch.cat((mel, f0), -1), lens.numpy(), draws[0], hp)
        assert np.array_equal(eng.debug_buffer('in.mel', B, T).cpu().numpy(), xi[:, :, :80].numpy())      # index path: bit-exact
    with pytest.raises(RuntimeError, match='SS_STEP_BUCKET'):              # without the flag the engine insists on its max_len_pad
        eng.g3_train_step(*synth_batch(1, B, 152, 150)[:4], stack_draws(draws_for(2, B, 4)))
    # ... and that is the max_len_pad it was created with, whatever bucket ran last (round-2 advisor finding: a 192-frame batch sent
    # WITHOUT the flag after a 104-frame bucket used to be refused)
    mel, f0, emb, lens = synth_batch(310, B, 192, 185)
    draws = draws_for(410, B, 4)
    loss = eng.g3_train_step(mel, f0, emb, lens, stack_draws(draws), no_adam=True)
    with torch.no_grad():
        lo, _ = ref_model.g3_loss(st.P, hp192, mel, f0, emb, lens.numpy(), draws)           # the oracle at the same (four times updated) weights
    assert abs(float(loss) - float(lo)) <= 1e-5 * float(lo), (float(loss), float(lo))


def test_solver_walks_buckets_and_full_length(E, tmp_path):
    """Solver.train_on_batch recognises buckets by their frame count: T, max_len_pad, T in a row (the full-length batch goes out
    without SS_STEP_BUCKET) -- the sequence a BucketedDeviceBatcher produces in about 8 % of its steps."""
    from types import SimpleNamespace
    from speechsplit_amd import solver as S, hparams as HPM
    hp = HPM.default_hparams(batch_size=4)              # max_len_pad = 192
    cfg = SimpleNamespace(num_iters=3, g_lr=1e-4, beta1=0.9, beta2=0.999, resume_iters=None, use_tensorboard=False, device_id=0,
                          log_dir=str(tmp_path), sample_dir=str(tmp_path), model_save_dir=str(tmp_path), log_step=1, sample_step=10 ** 9,
                          model_save_step=10 ** 9)
    s = S.Solver([], cfg, hp)
    losses = []
    for it, T in enumerate((144, 192, 104, 192, 192, 144)):
        mel, f0, emb, lens = synth_batch(500 + it, 4, T, T - 7)
        losses.append(float(s.train_on_batch((mel, emb, f0, lens))))
        s.eng.check()
    assert all(np.isfinite(losses)), losses


def test_bucketed_device_batcher(E):
    """Batches of one bucket each, crops of 96..192 frames inside (T - 8, T], assembled by ss_collate: equal to the host-side
    crop / clip / pad of the same rows; two ranks with one seed walk the same bucket sequence and draw different utterances."""
    from speechsplit_amd import buckets as K, data_loader as DL, hparams as HPM
    hp = HPM.default_hparams(batch_size=8)
    ds = DL.SyntheticUtterances(24, seed=4)
    corpus = DL.DeviceCorpus(ds, 'cuda')
    a = K.BucketedDeviceBatcher(hp, corpus, 96, 192, rank=0, world=2, seed=3, steps_per_epoch=12)
    b = K.BucketedDeviceBatcher(hp, corpus, 96, 192, rank=1, world=2, seed=3, steps_per_epoch=12)
    seen = set()
    mel_cat, f0_cat = corpus.mel.cpu().numpy(), corpus.f0.cpu().numpy()
    for (ma, ea, fa, la), (mb, eb, fb, lb) in zip(a, b):
        T = ma.shape[1]
        assert mb.shape[1] == T and T in K.buckets() and ma.shape == (4, T, 80) and fa.shape == (4, T, 1) and la.dtype == torch.int64
        assert int(la.min()) > T - 8 and int(la.max()) <= T and int(la.min()) >= 96
        assert not torch.equal(ma, mb)
        seen.add(T)
        for i in range(4):                       # rank 0's batch against a host crop of the same corpus rows
            n = int(la[i])
            row = ma[i, :n].cpu().numpy()
            hits = np.nonzero((np.clip(mel_cat[:, 0], 0, 1) == row[0, 0]))[0]
            ok = any(r + n <= len(mel_cat) and np.array_equal(np.clip(mel_cat[r:r + n], 0, 1), row) for r in hits)
            assert ok and float(ma[i, n:].abs().max() if n < T else 0) == 0.0
            assert float(fa[i, n:].max() if n < T else -1e10) == -1e10
    assert len(seen) >= 4


def test_native_rccl_dp_step_on_one_rank(E):
    """ss_comm_init / ss_g3_dp_train_step (RCCL dlopen'd, collectives launched by the engine itself, decoder bucket on the side
    stream) on a communicator of one rank: two steps equal two plain fused steps; ss_allreduce_grads leaves a one-rank sum alone."""
    B, T = 4, 128
    hp = W.default_hparams(max_len_pad=T)
    w = W.make_weights('G3', hp, 6)
    mel, f0, emb, lens = synth_batch(27, B, T, 64)
    d = stack_draws(draws_for(28, B, 4))
    res = []
    for native in (False, True):
        eng = E.Engine('G3', hp, B, T)
        eng.load_weights(w)
        eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
        if native:
            eng.comm_init(0, 1)
        for _ in range(2):
            loss = eng.dp_train_step_native(mel, f0, emb, lens, d) if native else eng.g3_train_step(mel, f0, emb, lens, d)
        eng.check()
        if native:
            g = eng.grads.clone()
            eng.allreduce_grads()
            torch.cuda.synchronize()
            assert torch.equal(g, eng.grads)
        res.append((float(loss), eng.params.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * res[0][0]
    assert_same_trajectory(res[1][1], res[0][1])


def test_native_g6_dp_step_and_modelled_collectives(E):
    """ss_g6_dp_train_step on a one-rank communicator equals the fused Generator_6 step; ss_tune("dp_model", 8) -- every collective
    replaced by a stand-in kernel of the modelled duration, no communicator -- leaves the gradients alone and folds 1/8 into Adam, for
    both generators and for both bucket plans (per-layer buckets on the communication stream / round 2's two buckets)."""
    B, T = 4, 192
    hp = W.default_hparams(max_len_pad=T)
    w6 = W.make_weights('G6', hp, 4)
    mel, f0, emb, lens = synth_batch(51, B, T, 96)
    qidx = torch.from_numpy(interp_np.quantize_f0(f0[:, :, 0].numpy()))
    onehot = torch.nn.functional.one_hot(qidx, 257).float()
    d3 = stack_draws(draws_for(61, B, 3))

    def run6(mode):
        eng = E.Engine('G6', hp, B, T)
        eng.load_weights(w6)
        eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
        if mode == 'native':
            eng.comm_init(0, 1)
        for _ in range(2):
            if mode == 'plain':
                loss = eng.g6_train_step(mel, onehot, qidx, d3)
            elif mode == 'plain8':
                loss = eng.g6_train_step(mel, onehot, qidx, d3, grad_scale=1.0 / 8)
            else:
                loss = eng.g6_dp_train_step_native(mel, onehot, qidx, d3)
        eng.check()
        return float(loss), eng.params.clone()

    a, b = run6('plain'), run6('native')
    assert abs(a[0] - b[0]) <= 1e-6 * a[0]
    assert_same_trajectory(b[1], a[1])
    E.tune('dp_model', 8)
    try:
        c, m = run6('plain8'), run6('model')
        assert abs(c[0] - m[0]) <= 1e-6 * c[0]
        assert_same_trajectory(m[1], c[1])
        # Generator_3, both bucket plans
        w3 = W.make_weights('G3', hp, 6)
        d4 = stack_draws(draws_for(28, B, 4))
        res = []
        for plan in ('plain8', 1, 0):
            if plan != 'plain8':
                E.tune('dp_buckets', plan)
            eng = E.Engine('G3', hp, B, T)
            eng.load_weights(w3)
            eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
            for _ in range(2):
                loss = eng.g3_train_step(mel, f0, emb, lens, d4, grad_scale=1.0 / 8) if plan == 'plain8' else eng.dp_train_step_native(mel, f0, emb, lens, d4)
            eng.check()
            res.append((float(loss), eng.params.clone()))
        for r in res[1:]:
            assert abs(r[0] - res[0][0]) <= 1e-6 * res[0][0]
            assert_same_trajectory(r[1], res[0][1])
    finally:
        E.tune('dp_model', 0)
        E.tune('dp_buckets', 1)


def test_g6_dp_step_on_a_one_rank_group(E):
    import os
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        B, T = 4, 192
        hp = W.default_hparams(max_len_pad=T)
        w = W.make_weights('G6', hp, 4)
        mel, f0, emb, lens = synth_batch(51, B, T, 96)
        qidx = torch.from_numpy(interp_np.quantize_f0(f0[:, :, 0].numpy()))
        onehot = torch.nn.functional.one_hot(qidx, 257).float()
        d = stack_draws(draws_for(61, B, 3))
        res = []
        for dp in (False, True):
            eng = E.Engine('G6', hp, B, T)
            eng.load_weights(w)
            eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
            for _ in range(2):
                loss = eng.dp_g6_train_step(mel, onehot, qidx, d, 1) if dp else eng.g6_train_step(mel, onehot, qidx, d)
            eng.check()
            res.append((float(loss), eng.params.clone()))
        assert abs(res[0][0] - res[1][0]) <= 1e-6 * res[0][0]
        assert_same_trajectory(res[1][1], res[0][1])
    finally:
        dist.destroy_process_group()


def test_generator6_trainer_and_per_rank_batcher(E, tmp_path):
    """SolverF0: the Solver shell around ss_g6_train_step (BASELINE config 4's model; the reference has no trainer for it) trains,
    logs and writes a checkpoint with the reference's Generator_6 key names; DeviceBatcher(rank, world) hands each rank its shard of
    the batch a single process would assemble."""
    import json
    import os
    from types import SimpleNamespace
    from speechsplit_amd import data_loader as DL, hparams as HP, solver
    hp = HP.default_hparams(batch_size=4)
    ds = DL.SyntheticUtterances(16, seed=2)
    cfg = SimpleNamespace(num_iters=3, g_lr=1e-4, beta1=0.9, beta2=0.999, resume_iters=None, use_tensorboard=False, device_id=0,
                          log_dir=str(tmp_path), sample_dir=str(tmp_path), model_save_dir=str(tmp_path), log_step=1, sample_step=1000,
                          model_save_step=3)
    np.random.seed(0)
    torch.manual_seed(0)
    s = solver.SolverF0(DL.get_device_loader(hp, dataset=ds), cfg, hp)
    p0 = s.eng.params.clone()
    s.train()
    s.eng.check()
    assert not torch.equal(p0, s.eng.params)
    ck = torch.load(os.path.join(str(tmp_path), '3-G.ckpt'), weights_only=False)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'keys_G6.json')
    assert list(ck['model'].keys()) == json.load(open(gold))['keys']
    assert float(ck['optimizer']['state'][0]['step']) == 3.0
    # per-rank assembly: the two ranks' shards, concatenated, are the single-process batch
    corpus = DL.DeviceCorpus(ds, 'cuda')
    idx = [3, 7, 1, 12]
    np.random.seed(5)
    whole = DL.DeviceBatcher(hp, corpus).assemble(idx)
    parts = []
    for r in range(2):
        np.random.seed(5)
        parts.append(DL.DeviceBatcher(hp, corpus, rank=r, world=2).assemble(idx))
    for k in range(4):
        assert torch.equal(torch.cat([parts[0][k], parts[1][k]]), whole[k])


@pytest.mark.parametrize('kind', ['G3', 'G6'])
def test_bucket_schedule_reduces_every_gradient_exactly_once(E, kind):
    """At world 1 an all-reduce is the identity, so nothing above notices a gradient range that is handed to a collective twice or never.
    ss_tune("dp_model", 2) + ("dp_emulate", 1): every stand-in collective MULTIPLIES its range by 2 -- the sum of two identical ranks --
    and the Adam step behind it carries 1/2: each element's first moment must then equal the one-GPU step's (Adam's update itself is
    invariant to the gradient's scale, the moments are not: 0.1 * g where the element was reduced once, 0.2 * g twice, 0.05 * g never)."""
    B, T = 4, 128 if kind == 'G3' else 192
    hp = W.default_hparams(max_len_pad=T)
    w = W.make_weights(kind, hp, 6)
    mel, f0, emb, lens = synth_batch(27, B, T, 64)
    d = stack_draws(draws_for(28, B, 4 if kind == 'G3' else 3))
    qidx = torch.from_numpy(interp_np.quantize_f0(f0[:, :, 0].numpy()))
    onehot = torch.nn.functional.one_hot(qidx, 257).float()
    res = []
    E.tune('deterministic', 1)
    try:
        for emulate in (False, True):
            eng = E.Engine(kind, hp, B, T)
            eng.load_weights(w)
            eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
            if emulate:
                E.tune('dp_model', 2)
                E.tune('dp_emulate', 1)
                if kind == 'G3':
                    eng.dp_train_step_native(mel, f0, emb, lens, d)
                else:
                    eng.g6_dp_train_step_native(mel, onehot, qidx, d)
            elif kind == 'G3':
                eng.g3_train_step(mel, f0, emb, lens, d)
            else:
                eng.g6_train_step(mel, onehot, qidx, d)
            eng.check()
            res.append({n: v.clone() for n, v in eng.views(eng.adam_m).items()})
    finally:
        E.tune('dp_emulate', 0)
        E.tune('dp_model', 0)
        E.tune('deterministic', 0)
    for n, m in res[0].items():
        den = float(m.abs().max()) + 1e-30
        assert float((res[1][n] - m).abs().max()) <= 1e-5 * den, (kind, n)
