"""Host-side mirror of the reference's ``Solver`` (boundary B2; reference solver.py:18-269).

Same constructor (``Solver(vcc_loader, config, hparams)``), same ``config`` attributes, same ``train()`` loop
structure, log line and checkpoint files; the six statements of the step body (solver.py:157-172) are ONE C-ABI
call into the HIP engine (``ss_g3_train_step``).  The demo.pkl validation block (solver.py:206-227) and
the ablation forwards (solver.py:245-251) are ``validate``; tensorboard and the matplotlib plots are out of scope.

Data parallelism is new (the reference is single-device): under ``torchrun`` each rank takes its shard of every
batch and of the resampling draws, gradients are summed with one RCCL all-reduce over the flat arena, and every
rank applies the same Adam update (speechsplit_amd/dist.py).
"""
import datetime
import os
import time

import torch

from . import dist as _dist
from .engine import draw_interp
from .model import Generator_3 as Generator
from .model import Generator_6
from .model import InterpLnr
from .staging import DevicePrefetcher
from .utils import pad_seq_to_2, quantize_f0_numpy


class Solver(object):
    """Solver for training"""

    def __init__(self, vcc_loader, config, hparams):
        self.vcc_loader = vcc_loader
        self.hparams = hparams
        self.num_iters = config.num_iters
        self.g_lr = config.g_lr
        self.beta1 = config.beta1
        self.beta2 = config.beta2
        self.resume_iters = config.resume_iters
        self.use_tensorboard = getattr(config, 'use_tensorboard', False)
        self.rank, self.local_rank, self.world = _dist.world_info()
        # data-parallel exchange (world > 1): 'native' = the engine's own RCCL communicator and bucket schedule (ss_g3_dp_train_step /
        # ss_g6_dp_train_step), 'torch' = torch.distributed all-reduces around the engine's no-Adam step (Engine.dp_train_step).  The native
        # schedule has never run on more than one GPU (the development boxes have one); the torch path is the fallback until it has.
        self.dp_backend = getattr(config, 'dp_backend', None) or os.environ.get('SS_DP_BACKEND', 'native')
        if self.dp_backend not in ('native', 'torch'):
            raise ValueError("dp_backend must be 'native' or 'torch', got {!r}".format(self.dp_backend))
        self.dp_schedule = getattr(config, 'dp_schedule', None) or os.environ.get('SS_DP_SCHEDULE', 'overlap')
        self.use_cuda = torch.cuda.is_available()
        if not self.use_cuda:
            raise RuntimeError('speechsplit_amd.Solver needs a ROCm GPU (the engine has no CPU fallback)')
        dev_id = self.local_rank if self.world > 1 else config.device_id
        self.device = torch.device('cuda:{}'.format(dev_id))
        torch.cuda.set_device(self.device)
        self.log_dir = config.log_dir
        self.sample_dir = config.sample_dir
        self.model_save_dir = config.model_save_dir
        self.log_step = config.log_step
        self.sample_step = config.sample_step
        self.model_save_step = config.model_save_step
        self.build_model()

    GENERATOR = Generator          # SolverF0 below trains Generator_6 with the same shell

    def build_model(self):
        per_rank = self.hparams.batch_size // max(self.world, 1)
        self.G = self.GENERATOR(self.hparams, max_batch=per_rank)
        self.Interp = InterpLnr(self.hparams)
        self.print_network(self.G, 'G')
        self.G.to(self.device)                      # creates the engine, parameters now live in its arena
        self.eng = self.G._eng
        self.eng.set_adam(self.g_lr, self.beta1, self.beta2, 1e-8, 0)      # solver.py:62
        self.step_count = 0
        if self.world > 1:
            _dist.init('nccl', self.device)
            import torch.distributed as dist
            dist.broadcast(self.eng.params, src=0)   # identical replicas
            self.eng.set_lockstep(True)              # never refuse a step on the status word: check() reports it on every rank at the same iteration
            # the engine's own RCCL communicator: its data-parallel step launches the collectives on the engine's streams and costs
            # nothing over the one-GPU step, where torch.distributed's path measured +0.5 ms
            if self.dp_backend == 'native':
                self.eng.comm_init(self.rank, self.world)

    def print_network(self, model, name):
        num_params = sum(p.numel() for p in model.parameters())
        if self.rank == 0:
            print(model)
            print(name)
            print("The number of parameters: {}".format(num_params))

    # ---- checkpoints in the reference's format: {'model': state_dict, 'optimizer': Adam.state_dict()} (solver.py:198-202)
    def optimizer_state_dict(self):
        mv, vv = self.eng.views(self.eng.adam_m), self.eng.views(self.eng.adam_v)
        state = {i: {'step': torch.tensor(float(self.step_count)), 'exp_avg': mv[n].detach().cpu().clone(),
                     'exp_avg_sq': vv[n].detach().cpu().clone()} for i, n in enumerate(self.G._names)}
        group = {'lr': self.g_lr, 'betas': (self.beta1, self.beta2), 'eps': 1e-8, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(len(self.G._names)))}
        return {'state': state if self.step_count else {}, 'param_groups': [group]}

    def load_optimizer_state_dict(self, sd):
        group = sd['param_groups'][0]
        self.g_lr = group['lr']
        self.beta1, self.beta2 = group['betas']
        mv, vv = self.eng.views(self.eng.adam_m), self.eng.views(self.eng.adam_v)
        step = 0
        for i, n in enumerate(self.G._names):
            st = sd['state'].get(i)
            if st is None:
                continue
            mv[n].copy_(st['exp_avg'])
            vv[n].copy_(st['exp_avg_sq'])
            step = int(st['step'])
        self.step_count = step
        self.eng.set_adam(self.g_lr, self.beta1, self.beta2, group.get('eps', 1e-8), step)

    def save_model(self, it):
        G_path = os.path.join(self.model_save_dir, '{}-G.ckpt'.format(it))
        torch.save({'model': {k: v.detach().cpu() for k, v in self.G.state_dict().items()},
                    'optimizer': self.optimizer_state_dict()}, G_path)

    def restore_model(self, resume_iters):
        print('Loading the trained models from step {}...'.format(resume_iters))
        G_path = os.path.join(self.model_save_dir, '{}-G.ckpt'.format(resume_iters))
        ckpt = torch.load(G_path, map_location=lambda storage, loc: storage, weights_only=False)
        self.G.load_state_dict(ckpt['model'])
        self.load_optimizer_state_dict(ckpt['optimizer'])

    # ---- one iteration of solver.py:141-172 on a collated batch
    def train_on_batch(self, batch, draws=None):
        x_real_org, emb_org, f0_org, len_org = batch
        per_rank = getattr(self.vcc_loader, 'per_rank', False)       # the loader already produced this rank's shard
        Bg = x_real_org.shape[0] * (self.world if per_rank else 1)
        if draws is None:
            draws = draw_interp(Bg, 4, self.hparams)                 # same generator calls, same order as the reference
        if self.world > 1:
            if not per_rank:
                x_real_org, emb_org, f0_org, len_org = _dist.shard_batch(batch, self.rank, self.world)
            draws = _dist.shard_draws(draws[0], draws[1], Bg, self.rank, self.world)
        to = dict(device=self.device, non_blocking=True)
        # a batch whose frame count is not hparams.max_len_pad is a length bucket (speechsplit_amd/buckets.py): max_len_pad = T
        bucket = x_real_org.shape[1] != self.hparams.max_len_pad
        if self.world > 1 and self.dp_backend == 'torch':
            loss = self.eng.dp_train_step(x_real_org.to(**to), f0_org.to(**to), emb_org.to(**to), len_org.to(**to), draws, self.world,
                                          schedule=self.dp_schedule, bucket=bucket)
        elif self.world > 1:
            loss = self.eng.dp_train_step_native(x_real_org.to(**to), f0_org.to(**to), emb_org.to(**to), len_org.to(**to), draws,
                                                 bucket=bucket)
        else:
            loss = self.eng.g3_train_step(x_real_org.to(**to), f0_org.to(**to), emb_org.to(**to), len_org.to(**to), draws,
                                          bucket=bucket)
        self.step_count += 1
        return loss

    # ---- validation on demo.pkl-style data (solver.py:206-227) and the ablation forwards of solver.py:245-251 (no plots)
    def validate(self, validation_pt, ablations=False):
        """validation_pt: list of [speaker, emb f32[1,82], (mel f32[L,80], f0 f32[L], L, uid)] (assets/demo.pkl layout).
        Returns (mean sum-reduced MSE, per-utterance dict of outputs).  G stays in eval mode afterwards, as in the reference."""
        import numpy as np
        self.G = self.G.eval()
        losses, outs = [], {}
        with torch.no_grad():
            for val_sub in validation_pt:
                emb = torch.from_numpy(np.asarray(val_sub[1], np.float32)).to(self.device)
                mel, f0, L = val_sub[2][0], val_sub[2][1], val_sub[2][2]
                x_real_pad, _ = pad_seq_to_2(mel[np.newaxis, :, :], 192)                                   # solver.py:213
                f0_pad = np.pad(f0, (0, 192 - L), 'constant', constant_values=(0, 0))                      # :215
                onehot = torch.from_numpy(quantize_f0_numpy(f0_pad)[0][np.newaxis]).to(self.device)        # :216-218
                x_real = torch.from_numpy(x_real_pad.astype(np.float32)).to(self.device)
                x_f0 = torch.cat((x_real, onehot), dim=-1)
                out = self.G(x_f0, x_real, emb)                                                            # :221
                losses.append(float(torch.nn.functional.mse_loss(x_real, out, reduction='sum')))           # :222
                rec = {'out': out}
                if ablations:                                                                              # :245-251
                    rec['woF'] = self.G(torch.cat((x_real, torch.zeros_like(onehot)), -1), x_real, emb)
                    rec['woR'] = self.G(x_f0, torch.zeros_like(x_real), emb)
                    rec['woC'] = self.G(torch.cat((torch.zeros_like(x_real), onehot), -1), x_real, emb)
                outs[val_sub[0]] = rec
        return float(np.mean(losses)), outs

    def train(self):
        data_loader = self.vcc_loader
        data_iter = DevicePrefetcher(data_loader, self.device)       # next batch staged H2D while this step runs
        validation_pt = getattr(self, 'validation_pt', None)
        if validation_pt is None and os.path.exists('assets/demo.pkl'):   # solver.py:16
            import pickle
            validation_pt = pickle.load(open('assets/demo.pkl', 'rb'))
        start_iters = 0
        if self.resume_iters:
            print('Resuming ...')
            start_iters = self.resume_iters
            self.num_iters += self.resume_iters
            self.restore_model(self.resume_iters)
        if self.rank == 0:
            print('Current learning rates, g_lr: {}.'.format(self.g_lr))
            print('Start training...')
        keys = ['G/loss_id']
        start_time = time.time()
        # The log line's loss is read WITHOUT draining the GPU: on a log step the loss is copied to pinned host memory behind the step
        # (non-blocking) and the line is printed once that copy has landed -- an iteration or two later, same text, same order.  A blocking
        # .item() every log_step iterations left the GPU waiting for the host to enqueue the next step again: 0.2 ms per iteration at
        # log_step = 10 (bench.py solver_loop: 5.51 vs 5.29 ms for the bare step).
        pending = []                                              # (copy-done event, pinned loss, iteration, elapsed text)

        def flush(block):
            while pending and (block or pending[0][0].query()):
                ev, host_loss, it, et = pending.pop(0)
                ev.synchronize()
                if self.rank == 0:
                    log = "Elapsed [{}], Iteration [{}/{}]".format(et, it, self.num_iters)
                    for tag in keys:
                        log += ", {}: {:.8f}".format(tag, float(host_loss))
                    print(log)

        for i in range(start_iters, self.num_iters):
            batch = next(data_iter)
            self.G = self.G.train()
            loss_dev = self.train_on_batch(batch)
            if (i + 1) % self.log_step == 0:
                t = loss_dev
                if self.world > 1:
                    # data parallel: every rank synchronises and checks at the SAME iteration (a rank that noticed a failure an
                    # iteration earlier than the others would leave them waiting in a collective)
                    import torch.distributed as dist
                    torch.cuda.synchronize(self.device)
                    self.eng.check()
                    t = loss_dev.clone()
                    dist.all_reduce(t)
                    t = t / self.world
                elif self.eng.status():                           # host-visible status word, no sync: a persistent kernel that gave up
                    self.eng.check()                              # waiting in an earlier step has said so by now (raises)
                host_loss = torch.empty((), dtype=torch.float32).pin_memory()
                host_loss.copy_(t.reshape(()), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                pending.append((ev, host_loss, i + 1, str(datetime.timedelta(seconds=time.time() - start_time))[:-7]))
            flush(self.world > 1)
            if (i + 1) % self.model_save_step == 0 and self.rank == 0:
                flush(True)
                self.save_model(i + 1)
                print('Saved model checkpoints into {}...'.format(self.model_save_dir))
            if (i + 1) % self.sample_step == 0 and validation_pt is not None and self.rank == 0:
                flush(True)
                val_loss, _ = self.validate(validation_pt)
                print('Validation loss: {}'.format(val_loss))
        flush(True)
        self.eng.check()                                          # the last steps too (synchronises)


class SolverF0(Solver):
    """Trainer for Generator_6, the F0 converter (reference model.py:324-351; demo.ipynb cell 0 uses a pretrained one, `640000-P.ckpt`).
    The reference ships NO training loop for it: this one is this repo's choice and says so -- the `Solver` shell
    unchanged (constructor, config attributes, log line, `<iter>-G.ckpt` layout with the reference's Generator_6 key names), the step
    body = quantise the batch's F0 (utils.quantize_f0_torch), feed the one-hot as `f0_trg`, cross-entropy of the 257-way logits
    against the same classes (demo.ipynb takes their argmax), backward, Adam: ONE `ss_g6_train_step` call.  BASELINE config 4."""
    GENERATOR = Generator_6

    def train_on_batch(self, batch, draws=None):
        from .utils import quantize_f0_torch
        x_real_org, emb_org, f0_org, len_org = batch
        per_rank = getattr(self.vcc_loader, 'per_rank', False)
        Bg = x_real_org.shape[0] * (self.world if per_rank else 1)
        if draws is None:
            draws = draw_interp(Bg, 3, self.hparams)                 # Encoder_6 resamples three times (model.py:128)
        if self.world > 1:
            if not per_rank:
                x_real_org, emb_org, f0_org, len_org = _dist.shard_batch(batch, self.rank, self.world)
            draws = _dist.shard_draws(draws[0], draws[1], Bg, self.rank, self.world)
        to = dict(device=self.device, non_blocking=True)
        mel, f0 = x_real_org.to(**to), f0_org.to(**to)
        onehot, idx = quantize_f0_torch(f0[:, :, 0])
        bucket = mel.shape[1] != self.hparams.max_len_pad
        if self.world > 1 and self.dp_backend == 'torch':
            loss = self.eng.dp_g6_train_step(mel, onehot, idx.to(torch.int32), draws, self.world, bucket=bucket)
        elif self.world > 1:
            loss = self.eng.g6_dp_train_step_native(mel, onehot, idx.to(torch.int32), draws, bucket=bucket)
        else:
            loss = self.eng.g6_train_step(mel, onehot, idx.to(torch.int32), draws, bucket=bucket)
        self.step_count += 1
        return loss

    def validate(self, validation_pt, ablations=False):
        raise NotImplementedError('the reference validates Generator_3 only (solver.py:206-227)')
