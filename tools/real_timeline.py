#!/usr/bin/env python3
"""Timeline of one training step from the engine's own hipEvent brackets (ss_profile_timeline): where each GEMM / recurrence class really
starts and ends when the host runs ahead of the device, without a tracing tool's host-side lag (rocprofv3 stretches a 3.0 ms step to 3.3 ms and
moves launches that wait for the host).  Two consecutive steps are bracketed; the second step's first record marks the end of the first.
usage: real_timeline.py [batch] [f32|bf16] [frames] [G3|G6] [dp] [key=value ...]   (dp: the data-parallel step with the engine's RCCL communicator at world 1)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speechsplit_amd import hparams as HP, model as M
from speechsplit_amd.engine import Engine, draw_interp, tune
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else 'f32'
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
hp = HP.default_hparams(max_len_pad=T, batch_size=B)
kind = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] in ('G3', 'G6') else 'G3'
for kv in sys.argv[4:]:
    if '=' in kv:
        tune(kv.split('=')[0], int(kv.split('=')[1]))
dev = torch.device('cuda:0')
eng = Engine(kind, hp, B, T, device=dev)
eng.load_weights(M.init_weights(kind, hp, 0))
eng.set_adam(1e-4, 0.9, 0.999, 1e-8, 0)
eng.set_precision(prec)
g = torch.Generator().manual_seed(5)
mel = torch.rand(B, T, 80, generator=g).to(dev)
f0 = torch.rand(B, T, 1, generator=g).to(dev)
emb = torch.zeros(B, hp.dim_spk_emb)
emb[torch.arange(B), torch.arange(B) % hp.dim_spk_emb] = 1
emb = emb.to(dev)
lens = torch.full((B,), T, dtype=torch.int32).to(dev)
draws = [tuple(t.to(dev) for t in draw_interp(B, 4 if kind == 'G3' else 3, hp)) for _ in range(16)]
if kind == 'G6':
    from speechsplit_amd.utils import quantize_f0_torch
    onehot, qidx = quantize_f0_torch(f0[:, :, 0].clone())
    onehot, qidx = onehot.contiguous(), qidx.to(torch.int32).contiguous()


dp = 'dp' in sys.argv[4:]
if dp:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    eng.comm_init(0, 1)


def one(d):
    if kind == 'G3' and dp:
        eng.dp_train_step_native(mel, f0, emb, lens, d)
    elif kind == 'G3':
        eng.g3_train_step(mel, f0, emb, lens, d)
    else:
        eng.g6_train_step(mel, onehot, qidx, d)


for i in range(10):
    one(draws[i])
torch.cuda.synchronize()
for i in range(3):
    one(draws[i])
eng.profile('timeline')
for i in range(3):
    one(draws[3 + i])
torch.cuda.synchronize()
tl = eng.profile_timeline()
eng.profile(False)
# step boundaries: the first record of a step is its first forward GEMM
firsts = [i for i, r in enumerate(tl) if r[0] == tl[0][0] and (i == 0 or tl[i - 1][0] != tl[0][0])]
n = len(tl) // 3
t0 = tl[n][1]
print(f'# {"Generator_3" if kind == "G3" else "Generator_6"} {B} x {T} {prec}{" data-parallel step (world 1)" if dp else ""}: second of three bracketed steps ({n} brackets per step; every bracket holds the next launch of its stream back by 4-8 us); '
      f'step period {tl[2 * n][1] - tl[n][1]:.0f} us')
print('#  start_us    end_us    dur_us  stream class')
names = ('main', 'side', 'branch2', 'branch3')
for k, a, b, st in tl[n:2 * n]:
    print(f'{a - t0:9.1f} {b - t0:9.1f} {b - a:9.1f}  {names[st]:7s} {k}')
print(f'{tl[2 * n][1] - t0:9.1f}  next step starts')
