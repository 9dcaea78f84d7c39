#!/usr/bin/env python3
"""Headline benchmark: Generator_3 training step (solver.py:157-172) throughput in utterances/s.

  python bench.py --gpus N --steps K --warmup W

N=1: one process.  N>1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
one rank per GPU; each rank processes its own 64-utterance shard (weak scaling), gradients are averaged with one
RCCL all-reduce over the flat gradient arena, then every rank applies the same Adam update.

Workload (BASELINE.json metric / SURVEY.md section 8(d)): batch 64 per GPU, 128 frames, 80-bin mel + F0, fp32,
max_len_pad=128, synthetic inputs resident in HBM before the timed region, weights from a fixed seed.
A step = resample -> quantise -> forward -> MSE -> backward -> [all-reduce] -> Adam, nothing skipped.

Rank 0 prints ONE JSON line (contract in the task statement) including `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work, SURVEY.md section 8(d): 19 405 064 MAC / frame / utterance, fwd+bwd = 3 x 2 x MAC
MAC_PER_FRAME_G3 = 19405064
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_* dense peak


def synth(B, T, seed, device):
    g = torch.Generator().manual_seed(seed)
    mel = torch.rand(B, T, 80, generator=g)
    f0 = torch.rand(B, T, 1, generator=g)
    uv = torch.rand(B, T, 1, generator=g) < 0.4
    lens = torch.randint(64, T + 1, (B,), generator=g)
    pad = torch.arange(T)[None, :, None] >= lens[:, None, None]
    f0 = torch.where(uv | pad, torch.full_like(f0, -1e10), f0)
    mel = torch.where(pad, torch.zeros_like(mel), mel)
    emb = torch.nn.functional.one_hot(torch.randint(0, 82, (B,), generator=g), 82).float()
    return [t.to(device) for t in (mel, f0, emb, lens.to(torch.int32))]


def cpu_baseline(T, seconds_budget=20.0):
    """The oracle (PyTorch-CPU restatement of the reference step, same ATen/oneDNN kernels) timed on the host cores,
    on a bounded sample of the same workload: batch 16 instead of 64 (utt/s is flat in B on CPU)."""
    from oracle import ref_model, weights as W
    from oracle.gen_fixtures import synth_batch, draws_for
    # the GPU box shares its host: use the CPU share of one GPU slot (16), not every core the kernel lists
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get('SS_CPU_THREADS', min(avail, 16)))
    torch.set_num_threads(cores)
    print(f'[bench] cpu_baseline: {cores} threads (of {avail} visible)', file=sys.stderr, flush=True)
    B = 16
    hp = W.default_hparams(max_len_pad=T)
    st = ref_model.TrainState(W.make_weights('G3', hp, 0))
    mel, f0, emb, lens = synth_batch(1, B, T, 64)
    draws = draws_for(2, B, 4)
    st.step_g3(hp, mel, f0, emb, lens.numpy(), draws)          # warm-up
    n, t0 = 0, time.time()
    while n < 3 or (time.time() - t0 < seconds_budget and n < 20):
        st.step_g3(hp, mel, f0, emb, lens.numpy(), draws)
        n += 1
        print(f'[bench] cpu_baseline step {n}: {(time.time() - t0) / n:.2f} s/step', file=sys.stderr, flush=True)
    dt = (time.time() - t0) / n
    return dict(value=round(B / dt, 2), unit='utterances/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{n} full training steps of Generator_3 at batch {B} x {T} frames (PyTorch-CPU oracle, fp32)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='utterances per GPU')
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', choices=['f32', 'bf16'], default='f32',
                    help="f32 (default): the 1e-4 parity mode the headline is quoted on; bf16: BASELINE configs 2-4's arithmetic")
    ap.add_argument('--stream', choices=['default', 'own'], default='default', help='launch the steps on the default stream or on a stream of their own')
    ap.add_argument('--tune', action='append', default=[], metavar='KEY=VALUE', help='experiment knob of the HIP library (ss_tune)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py: for --gpus N > 1 launch with torch.distributed.run --nproc-per-node N')
    torch.cuda.set_device(local)
    dev = torch.device(f'cuda:{local}')

    from oracle import weights as W            # weight generator only (frozen-stream seeds); not on the timed path
    from speechsplit_amd.engine import Engine, draw_interp, tune
    for kv in args.tune:
        k, v = kv.split('=')
        tune(k, int(v))
    B, T = args.batch, args.frames
    hp = W.default_hparams(max_len_pad=T)
    eng = Engine('G3', hp, B, T, device=dev)
    eng.load_weights(W.make_weights('G3', hp, 0))
    # The communicator comes AFTER the engine: HIP spreads a process's streams over 4 hardware queues in creation order, and
    # the engine's four streams should get one each (with RCCL's streams created first the same step measured 0.3 - 0.7 ms
    # slower on one GPU: tools/stream_order_effect.py, tools/soak_dp.py).
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
    eng.set_precision(args.precision)
    mel, f0, emb, lens = synth(B, T, 1000 + rank, dev)
    torch.manual_seed(1234)
    sc_all, ls_all = draw_interp(B * world, 4, hp)          # one global draw stream, each rank takes its utterances' slice
    S = sc_all.shape[1] // (B * world)
    sc = sc_all.view(4, world, B * S)[:, rank].contiguous().to(dev)
    ls = ls_all.view(4, world, B * S)[:, rank].contiguous().to(dev)

    def step():
        if world == 1:
            eng.g3_train_step(mel, f0, emb, lens, (sc, ls))
        else:
            # RCCL sum of the gradient arena over xGMI: decoder + head bucket from the engine's side stream beside the encoder backward,
            # then the encoder bucket (Engine.dp_train_step, schedule 'overlap');
            # the mean is taken inside the Adam kernel
            eng.dp_train_step(mel, f0, emb, lens, (sc, ls), world)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.stream == 'own':
        own = torch.cuda.Stream(dev)
        own.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(own)
    for _ in range(args.warmup):
        step()
    barrier()
    eng.profile(True)            # hipEvent pairs around the dominant kernel's launches, on their launch stream (ss_profile)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    eng.check()                  # outside the timed region: no persistent kernel reported a bounded-wait expiry
    ms = dt / args.steps * 1e3
    value = B * world / (dt / args.steps)

    if rank == 0:
        # roofline of the dominant kernel family: the fp32 MFMA GEMM.  Timed live with HIP events on the launch stream
        # at the shape that carries the most FLOPs in the step (decoder input projection, layer 1/2).
        print(f'[bench] gpu: {ms:.3f} ms/step, {value:.1f} utt/s on {world} GPU(s)', file=sys.stderr, flush=True)
        roof = gemm_roofline(eng, B, T, args.precision)
        print(f'[bench] roofline: {roof}', file=sys.stderr, flush=True)
        out = {
            'metric': 'utterances/sec (fwd+bwd, 128-frame 80-mel, batch 64) at 1/2/4/8 MI355X',
            'value': round(value, 1), 'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': f'Generator_3 full training step (resample+quantise+fwd+MSE+bwd+Adam), '
                                   f'{B} utterances/GPU x {T} frames x 80 mel + F0, max_len_pad={T}',
                       'global_batch': B * world, 'seq_len': T, 'parallelism': f'dp{world}'},
            'step_tflops': round(3 * 2 * MAC_PER_FRAME_G3 * T * B * world / (dt / args.steps) / 1e12, 2),
            'roofline': roof,
            'cpu_baseline': None if args.no_cpu_baseline else cpu_baseline(T),
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def gemm_roofline(eng, B, T, precision='f32'):
    """Roofline of the dominant kernel, measured INSIDE the timed region: the engine brackets every launch of the decoder
    input-projection GEMM of layers 1-2 (gemm_bf16x3_kernel<128,128,NT>; per launch M = B*T rows, N = 4096 = both directions,
    K = 1024, fp32 in / fp32 out / fp32 accumulate) with hipEvents on its launch stream (ss_profile in
    include/speechsplit_amd.h).  achieved = algorithmic FLOPs of one launch (2*M*N*K) / mean launch duration.

    The kernel forms every fp32 product on the 16-bit matrix pipe (dense peak 2.5 PFLOP/s for fp16 and bf16 alike) from a
    split of both operands: forward contractions, whose operands are bounded by construction, use fp16 x 2 = 3
    v_mfma_f32_32x32x16_f16 per multiply-add (22 significand bits relative to the operand's maximum); the general path is
    bf16 x 3 = 6 MFMAs.  The ceiling in algorithmic fp32 FLOP/s is therefore 2500 / 3 = 833 TFLOP/s for this kernel; the
    fp32 MFMA peak (157.3 TFLOP/s) is reported beside it."""
    n, us, flops = eng.profile(False)
    if n == 0:
        return None
    sec = us / n / 1e6
    ach = flops / sec / 1e12
    # HBM bytes per launch of this shape from the PMC passes kept under profiles/ (separate rocprofv3 --pmc runs, gfx950
    # corrections applied by tools/pmc_summary.py); counters cannot be collected from inside the benchmark process
    traffic = None
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'gemm_pmc.json')))[0]
        if rec['shape'].startswith('proj NT 8192x4096x1024') and B * T == 8192 and 'bf16x3' in rec.get('kernel', ''):
            traffic = rec['hbm_read_bytes'] + rec['hbm_write_bytes']
    except Exception:
        pass
    if precision == 'bf16':      # one bf16 MFMA per product: the pipe's own peak
        return {'bound': 'mfma', 'kernel': 'gemm_bf16x3_kernel<128,128,NT,1 plane> (decoder input projection, layers 1-2; bf16 operands, fp32 accumulate)',
                'achieved': round(ach, 2), 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                'launches_timed': n, 'flops_per_launch': flops, 'us_per_launch': round(sec * 1e6, 2), 'traffic': None}
    products = 3
    peak = PEAK_BF16_MFMA_TFLOPS / products
    return {'bound': 'mfma', 'kernel': 'gemm_bf16x3_kernel<128,128,NT,fp16x2> (decoder input projection, layers 1-2; fp32 via 3 fp16 MFMAs per product)',
            'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'peak_basis': 'dense 16-bit MFMA 2500 TFLOP/s / 3 MFMA products per fp32 multiply-add',
            'mfma_tflops_executed': round(products * ach, 1), 'fp32_mfma_peak': PEAK_F32_MFMA_TFLOPS,
            'vs_fp32_mfma_peak': round(ach / PEAK_F32_MFMA_TFLOPS, 4),
            'launches_timed': n, 'flops_per_launch': flops, 'us_per_launch': round(sec * 1e6, 2), 'traffic': traffic}


if __name__ == '__main__':
    main()
