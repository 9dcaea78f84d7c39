"""Thin Python owner of one HIP engine: allocates the four parameter arenas and the workspace as torch CUDA
tensors (PyTorch is only the allocator / stream provider here), hands their device pointers to the C ABI and
exposes zero-copy parameter / gradient views under the reference's state_dict names."""
import ctypes as C

import numpy as np
import torch

from . import _capi

KIND = {'G3': 3, 'G6': 6, 'interp': 0}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def draw_interp(batch, ncalls, hp, generator=None):
    """Draw the InterpLnr randomness exactly as the reference consumes it (model.py:392-393 then 399-402, per
    call, from the default CPU generator unless one is given).  Returns (scales f32[ncalls, B*S], len_seg i32[...])."""
    S = hp.max_len_seq // hp.min_len_seg + 1
    sc, ls = [], []
    for _ in range(ncalls):
        sc.append(torch.rand(batch * S, generator=generator) + 0.5)
        ls.append(torch.randint(low=hp.min_len_seg, high=hp.max_len_seg, size=(batch * S, 1), generator=generator))
    return torch.stack(sc), torch.stack(ls).reshape(ncalls, -1).to(torch.int32)


class Engine:
    def __init__(self, kind, hp, max_batch, max_frames=None, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('speechsplit_amd.Engine needs a ROCm GPU (no CPU fallback)')
        self.lib = _capi.lib()
        self.kind = kind
        self.hp = hp
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.max_batch = int(max_batch)
        self.max_frames = int(max_frames or hp.max_len_pad)
        self._stagers = {}
        self._hps = _capi.hparams_struct(hp)
        self.h = self.lib.ss_create(KIND[kind], C.byref(self._hps), self.max_batch, self.max_frames)
        if not self.h:
            raise RuntimeError('speechsplit_amd: ' + self.lib.ss_last_error().decode())
        self.table = []
        name = C.create_string_buffer(256)
        off, nd, shp = C.c_long(), C.c_int(), (C.c_long * 3)()
        for i in range(self.lib.ss_num_params(self.h)):
            _capi.check(self.lib.ss_param_info(self.h, i, name, 256, C.byref(off), C.byref(nd), C.byref(shp)))
            self.table.append((name.value.decode(), off.value, tuple(shp[k] for k in range(nd.value))))
        n = self.lib.ss_arena_numel(self.h)
        with torch.cuda.device(self.device):
            self.params = torch.zeros(n, device=self.device)
            self.grads = torch.zeros(n, device=self.device)
            self.adam_m = torch.zeros(n, device=self.device)
            self.adam_v = torch.zeros(n, device=self.device)
            self.ws = torch.empty(self.lib.ss_workspace_bytes(self.h), dtype=torch.uint8, device=self.device)
            self.loss = torch.zeros(1, device=self.device)
            _capi.check(self.lib.ss_bind(self.h, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                         _ptr(self.ws), self.ws.numel(), _stream()))

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.lib.ss_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def views(self, arena):
        return {n: arena[o:o + int(np.prod(s))].view(*s) for n, o, s in self.table}

    def param_views(self):
        return self.views(self.params)

    def grad_views(self):
        return self.views(self.grads)

    def load_weights(self, weights):
        """weights: dict name -> numpy / tensor with the reference's shapes."""
        pv = self.param_views()
        for n, _, s in self.table:
            w = weights[n]
            w = torch.from_numpy(np.ascontiguousarray(w)) if isinstance(w, np.ndarray) else w.detach()
            assert tuple(w.shape) == tuple(s), (n, w.shape, s)
            pv[n].copy_(w.to(torch.float32))

    def set_adam(self, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, step=0):
        _capi.check(self.lib.ss_set_adam(self.h, lr, beta1, beta2, eps, int(step), _stream()))

    # ------------------------------------------------------------------ helpers
    def _f(self, t):
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def _i(self, t):
        return t.to(device=self.device, dtype=torch.int32).contiguous()

    def _draws(self, draws):
        """Device tensors pass through untouched; host draws go through a pinned ring with one non-blocking copy on the
        compute stream (staging.DrawStager) -- never a pageable, blocking `.to(device)`."""
        if draws is None:
            return None, None
        sc, ls = draws
        sc, ls = torch.as_tensor(sc), torch.as_tensor(ls)
        if sc.is_cuda and ls.is_cuda:
            return self._f(sc), self._i(ls)
        sc, ls = sc.reshape(sc.shape[0], -1), ls.reshape(ls.shape[0], -1)
        key = tuple(sc.shape)
        st = self._stagers.get(key)
        if st is None:
            from .staging import DrawStager
            st = self._stagers[key] = DrawStager(self.device, key[0], key[1])
        return st.stage(sc, ls)

    # ------------------------------------------------------------------ Generator_3
    def g3_forward(self, x_f0, x_org, c_trg, draws=None, training=False):
        B, T, _ = x_org.shape
        x_f0, x_org, c_trg = self._f(x_f0), self._f(x_org), self._f(c_trg)
        if c_trg.shape[0] != B:
            c_trg = c_trg.expand(B, -1).contiguous()
        sc, ls = self._draws(draws)
        out = torch.empty(B, T, self.hp.dim_freq, device=self.device)
        _capi.check(self.lib.ss_g3_forward(self.h, _ptr(x_f0), _ptr(x_org), _ptr(c_trg), _ptr(sc), _ptr(ls), B, T,
                                           int(training), _ptr(out), _stream()))
        return out

    def g3_backward(self, d_out):
        d_out = self._f(d_out)
        _capi.check(self.lib.ss_g3_backward(self.h, _ptr(d_out), _stream()))

    def g3_rhythm(self, x_org):
        B, T, _ = x_org.shape
        x_org = self._f(x_org)
        codes = torch.empty(B, T // self.hp.freq_2, 2 * self.hp.dim_neck_2, device=self.device)
        _capi.check(self.lib.ss_g3_rhythm(self.h, _ptr(x_org), B, T, _ptr(codes), _stream()))
        return codes

    def g3_train_step(self, mel, f0, emb, len_org, draws, grad_scale=1.0, no_adam=False, split_backward=False, bucket=False):
        """solver.py:160-172 fused.  split_backward: return after the decoder + head gradients (arena offsets >=
        self.grad_split) are complete; train_finish() then runs the encoder backward (data-parallel overlap).
        bucket: the batch's frame count is its length bucket and the step runs with max_len_pad = T (SS_STEP_BUCKET)."""
        B, T, _ = mel.shape
        mel, f0, emb, len_org = self._f(mel), self._f(f0), self._f(emb), self._i(len_org)
        sc, ls = self._draws(draws)
        assert sc.shape[0] == 4 and ls.shape[0] == 4
        flags = (1 if no_adam else 0) | (2 if split_backward else 0) | (16 if bucket else 0)
        _capi.check(self.lib.ss_g3_train_step(self.h, _ptr(mel), _ptr(f0), _ptr(emb), _ptr(len_org), _ptr(sc), _ptr(ls),
                                              B, T, float(grad_scale), flags, _ptr(self.loss), _stream()))
        return self.loss

    def train_finish(self, grad_scale=1.0, no_adam=True):
        _capi.check(self.lib.ss_train_finish(self.h, float(grad_scale), 1 if no_adam else 0, _stream()))

    @property
    def grad_split(self):
        return int(self.lib.ss_grad_split(self.h))

    def dp_train_step(self, mel, f0, emb, len_org, draws, world, group=None, schedule='overlap', bucket=False):
        """One data-parallel Generator_3 step on this rank's shard: backward, sum of the gradient arena over the ranks in the
        two buckets of dist.bucket_plan (decoder + head + status slot first, then the encoder), the same Adam update on every
        rank with the 1/world mean folded in.  Collectives: torch.distributed (backend 'nccl' = RCCL); the native path without
        PyTorch in the data path is dp_train_step_native.

        schedule='overlap' (default): the backward is enqueued exactly as in the one-GPU step (SS_STEP_SPLIT_NO_JOIN: the
            decoder's weight-gradient GEMMs stay on an engine stream beside the encoder backward); the first bucket's
            all-reduce is issued from a stream ordered behind the decoder chain and those GEMMs, BEFORE the encoder backward is
            enqueued, so it can run beside it.  (ProcessGroupNCCL launches the collective on its own internal stream, which
            first waits for the stream the call was issued from.)  Unmeasured at world > 1 on hardware: the one-GPU boxes this
            was developed on elide the collective.
        schedule='after': the one-GPU step unchanged, then the buckets; nothing hidden.
        schedule='join': SS_STEP_SPLIT_BACKWARD joins the engine streams, first bucket reduced right away (+0.6 ms at world 1:
            the decoder's weight-gradient GEMMs then run alone instead of beside the encoder backward)."""
        from . import dist as D
        k = self.grad_split
        plan = D.bucket_plan(self.grads.numel(), k)
        side = self.lib.ss_side_stream(self.h) if schedule == 'overlap' else None
        if schedule == 'overlap' and not side:
            schedule = 'after'                         # engine without branch streams
        if schedule == 'after':
            self.g3_train_step(mel, f0, emb, len_org, draws, no_adam=True, bucket=bucket)
            handles = [D.reduce_bucket(self.grads, lo, hi, group) for lo, hi in plan]
        elif schedule == 'join':
            self.g3_train_step(mel, f0, emb, len_org, draws, no_adam=True, split_backward=True, bucket=bucket)
            handles = [D.reduce_bucket(self.grads, *plan[0], group)]
            self.train_finish(no_adam=True)
            handles.append(D.reduce_bucket(self.grads, *plan[1], group))
        elif schedule == 'overlap':
            sc, ls = self._draws(draws)
            B, T, _ = mel.shape
            mel, f0, emb, len_org = self._f(mel), self._f(f0), self._f(emb), self._i(len_org)
            # SS_STEP_NO_ADAM | SS_STEP_SPLIT_BACKWARD | SS_STEP_SPLIT_NO_JOIN
            _capi.check(self.lib.ss_g3_train_step(self.h, _ptr(mel), _ptr(f0), _ptr(emb), _ptr(len_org), _ptr(sc), _ptr(ls),
                                                  B, T, 1.0, 1 | 2 | 4 | (16 if bucket else 0), _ptr(self.loss), _stream()))
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = torch.cuda.ExternalStream(side, device=self.device)
            cs = self._side_stream
            # order the issuing stream behind the decoder chain and its weight-gradient GEMMs; this must precede the encoder
            # backward (a cross-stream wait on ROCm covers what the other stream holds when it is issued)
            _capi.check(self.lib.ss_wait_decoder_grads(self.h, C.c_void_p(cs.cuda_stream)))
            with torch.cuda.stream(cs):
                handles = [D.reduce_bucket(self.grads, *plan[0], group)]
            self.train_finish(no_adam=True)
            handles.append(D.reduce_bucket(self.grads, *plan[1], group))
        else:
            raise ValueError(f'unknown data-parallel schedule {schedule!r}')
        for h in handles:
            if h is not None:
                h.wait()                               # the current stream waits for the collectives
        self.adam_step(1.0 / world)
        return self.loss

    def dp_g6_train_step(self, mel, f0_onehot, target_idx, draws, world, group=None, bucket=False):
        """Data-parallel Generator_6 step (BASELINE config 4): the arena is 14 MB, one pass of the bucket plan behind the backward."""
        from . import dist as D
        self.g6_train_step(mel, f0_onehot, target_idx, draws, no_adam=True, bucket=bucket)
        D.reduce_arena(self.grads, self.grad_split, group)
        self.adam_step(1.0 / world)
        return self.loss

    # ---- native RCCL (no PyTorch in the data path): ss_comm_* / ss_g3_dp_train_step of the C ABI
    def comm_init(self, rank=0, world=1, group=None):
        """Create this engine's RCCL communicator.  Rank 0 draws the 128-byte id; with world > 1 it is handed round through the
        already initialised torch.distributed group (any backend: it is 128 bytes of bootstrap, not data)."""
        buf = C.create_string_buffer(128)
        if rank == 0:
            _capi.check(self.lib.ss_comm_unique_id(buf))
        if world > 1:
            import torch.distributed as dist
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            buf = C.create_string_buffer(box[0], 128)
        _capi.check(self.lib.ss_comm_init(self.h, buf, int(rank), int(world)))
        self.comm_world = int(world)

    def scratch_fallbacks(self):
        """Launches that found the step's scratch exhausted and took their slower path (ss_scratch_fallbacks); 0 in a healthy run."""
        return int(self.lib.ss_scratch_fallbacks(self.h))

    def dp_profile(self, on=True):
        """hipEvent brackets round every collective of the native data-parallel steps (ss_dp_profile)."""
        _capi.check(self.lib.ss_dp_profile(self.h, 1 if on else 0))

    def dp_profile_read(self):
        """[(arena offset or -1 for the grouped rest, elements, start_us, end_us)] of the last data-parallel step's collectives, times
        relative to the end of the backward on the main stream (negative: hidden beside it).  Synchronises."""
        n = self.lib.ss_dp_profile_read(self.h, None, 0)
        if n <= 0:
            return []
        buf = (C.c_double * (4 * n))()
        n = self.lib.ss_dp_profile_read(self.h, buf, n)
        if n < 0:
            _capi.check(n)
        return [(int(buf[4 * i]), int(buf[4 * i + 1]), float(buf[4 * i + 2]), float(buf[4 * i + 3])) for i in range(n)]

    def allreduce_grads(self, lo=0, hi=None):
        hi = self.grads.numel() if hi is None else hi
        _capi.check(self.lib.ss_allreduce_grads(self.h, int(lo), int(hi - lo), _stream()))

    def dp_train_step_native(self, mel, f0, emb, len_org, draws, bucket=False):
        """ss_g3_dp_train_step: the overlapped two-bucket schedule with the collectives launched by the engine itself, the
        decoder bucket ON the engine stream that carries the decoder's weight-gradient GEMMs."""
        B, T, _ = mel.shape
        mel, f0, emb, len_org = self._f(mel), self._f(f0), self._f(emb), self._i(len_org)
        sc, ls = self._draws(draws)
        _capi.check(self.lib.ss_g3_dp_train_step(self.h, _ptr(mel), _ptr(f0), _ptr(emb), _ptr(len_org), _ptr(sc), _ptr(ls), B, T,
                                                 16 if bucket else 0, _ptr(self.loss), _stream()))
        return self.loss

    # ------------------------------------------------------------------ Generator_6
    def g6_forward(self, x_org, f0_trg, draws=None, training=False):
        B, T, _ = x_org.shape
        x_org, f0_trg = self._f(x_org), self._f(f0_trg)
        sc, ls = self._draws(draws)
        out = torch.empty(B, T, self.hp.dim_f0, device=self.device)
        _capi.check(self.lib.ss_g6_forward(self.h, _ptr(x_org), _ptr(f0_trg), _ptr(sc), _ptr(ls), B, T, int(training),
                                           _ptr(out), _stream()))
        return out

    def g6_backward(self, d_out):
        d_out = self._f(d_out)
        _capi.check(self.lib.ss_g6_backward(self.h, _ptr(d_out), _stream()))

    def g6_train_step(self, mel, f0_onehot, target_idx, draws, grad_scale=1.0, no_adam=False, bucket=False):
        B, T, _ = mel.shape
        mel, f0_onehot, target_idx = self._f(mel), self._f(f0_onehot), self._i(target_idx)
        sc, ls = self._draws(draws)
        _capi.check(self.lib.ss_g6_train_step(self.h, _ptr(mel), _ptr(f0_onehot), _ptr(target_idx), _ptr(sc), _ptr(ls), B, T,
                                              float(grad_scale), (1 if no_adam else 0) | (16 if bucket else 0), _ptr(self.loss), _stream()))
        return self.loss

    def g6_dp_train_step_native(self, mel, f0_onehot, target_idx, draws, bucket=False):
        """ss_g6_dp_train_step: Generator_6's data-parallel step with the engine's own RCCL communicator (per-layer buckets on the
        engine's communication stream), as dp_train_step_native for Generator_3."""
        B, T, _ = mel.shape
        mel, f0_onehot, target_idx = self._f(mel), self._f(f0_onehot), self._i(target_idx)
        sc, ls = self._draws(draws)
        _capi.check(self.lib.ss_g6_dp_train_step(self.h, _ptr(mel), _ptr(f0_onehot), _ptr(target_idx), _ptr(sc), _ptr(ls), B, T,
                                                 16 if bucket else 0, _ptr(self.loss), _stream()))
        return self.loss

    # ------------------------------------------------------------------ optimiser / misc
    def adam_step(self, grad_scale=1.0):
        _capi.check(self.lib.ss_adam_step(self.h, float(grad_scale), _stream()))

    def check(self):
        """Synchronise and raise if a kernel reported an asynchronous failure."""
        _capi.check(self.lib.ss_check(self.h, _stream()))

    def status(self):
        """Engine status word without synchronising (ss_status): 0 ok, bit 0 aborted recurrence, 1 remote abort, 2 parameter range."""
        return int(self.lib.ss_status(self.h))

    def clear_abort(self):
        _capi.check(self.lib.ss_clear_abort(self.h, _stream()))

    def set_lockstep(self, on=True):
        """Data-parallel member whose gradients are exchanged outside the engine (torch.distributed): entry points never refuse on the
        status word, check() reports it at a point all ranks reach together (ss_set_lockstep; the native communicator sets it itself)."""
        _capi.check(self.lib.ss_set_lockstep(self.h, 1 if on else 0))

    def zero_grads(self):
        _capi.check(self.lib.ss_zero_grads(self.h, _stream()))

    def interp_forward(self, x, len_seq, scales, len_seg, want_plan=False):
        B, T, Cc = x.shape
        P = self.hp.max_len_pad
        x, len_seq = self._f(x), self._i(torch.as_tensor(len_seq))
        sc, ls = self._f(torch.as_tensor(scales)), self._i(torch.as_tensor(len_seg))
        y = torch.empty(B, P, Cc, device=self.device)
        i0 = torch.empty(B, P, dtype=torch.int32, device=self.device) if want_plan else None
        lam = torch.empty(B, P, device=self.device) if want_plan else None
        cnt = torch.empty(B, dtype=torch.int32, device=self.device) if want_plan else None
        _capi.check(self.lib.ss_interp_forward(self.h, _ptr(x), _ptr(len_seq), _ptr(sc), _ptr(ls), B, T, Cc, _ptr(y),
                                               _ptr(i0), _ptr(lam), _ptr(cnt), _stream()))
        return (y, i0, lam, cnt) if want_plan else y

    def interp_backward(self, dy, T):
        B, P, Cc = dy.shape
        dy = self._f(dy)
        dx = torch.empty(B, T, Cc, device=self.device)
        _capi.check(self.lib.ss_interp_backward(self.h, _ptr(dy), B, T, Cc, _ptr(dx), _stream()))
        return dx

    def debug_names(self):
        buf = C.create_string_buffer(1 << 14)
        self.lib.ss_debug_names(self.h, buf, len(buf))
        return [s for s in buf.value.decode().split('\n') if s]

    def relu_masks(self, B, T):
        """Test hook (ss_debug_relu_mask): {conv block name: bool [B,T,Co]} -- the ReLU branch taken in the last forward."""
        out = {}
        for name in self.debug_names():
            if not name.endswith('.conv'):
                continue
            blk = name[:-5]
            p, rows, cols = C.c_void_p(), C.c_long(), C.c_long()
            _capi.check(self.lib.ss_debug_buffer(self.h, name.encode(), C.byref(p), C.byref(rows), C.byref(cols)))
            m = torch.empty(B, T, cols.value, device=self.device)
            _capi.check(self.lib.ss_debug_relu_mask(self.h, blk.encode(), _ptr(m), _stream()))
            out[blk] = m > 0
        return out

    def set_precision(self, precision):
        """'f32' (default, the 1e-4 parity mode) or 'bf16' (bf16-rounded GEMM operands, fp32 accumulate / state)."""
        code = {'f32': 0, 'fp32': 0, 'bf16': 1}[precision]
        _capi.check(self.lib.ss_set_precision(self.h, code))
        self.precision = 'bf16' if code else 'f32'

    PROF_CLASSES = ('dec_proj', 'dec_proj0', 'dec_dw', 'dec_dx', 'conv_fwd', 'conv_dw', 'conv_dx', 'rec_fwd', 'rec_bwd',
                    'enc_lstm', 'head')

    PROF_TIMELINE = ('enc_rec', 'gn', 'wgrad', 'adam', 'prep')      # timeline-only classes (SS_PROF_ENC_REC ..): non-GEMM launches, no flops

    def profile(self, classes, every=1):
        """ss_profile: bracket the launches of the named classes (True: all; False / empty: stop) with hipEvents; starting
        clears the record.  every = n: only every n-th training step is bracketed (ss_profile_sample)."""
        _capi.check(self.lib.ss_profile_sample(self.h, int(every)))
        names = self.PROF_CLASSES + self.PROF_TIMELINE
        if classes is True:
            mask = (1 << len(self.PROF_CLASSES)) - 1
        elif classes == 'timeline':                  # every class including the timeline-only ones
            mask = (1 << len(names)) - 1
        elif not classes:
            mask = 0
        else:
            mask = sum(1 << names.index(c) for c in classes)
        _capi.check(self.lib.ss_profile(self.h, mask))

    def profile_read(self):
        """{class: (launches, total_us, total_flops)} of the launches recorded since profile(True) (ss_profile_read)."""
        out = {}
        for k, name in enumerate(self.PROF_CLASSES):
            n, us, fl = C.c_int(), C.c_double(), C.c_double()
            _capi.check(self.lib.ss_profile_read(self.h, k, C.byref(n), C.byref(us), C.byref(fl)))
            if n.value:
                out[name] = (n.value, us.value, fl.value)
        return out

    def profile_timeline(self, cap=8192):
        """[(class name, start_us, end_us, stream)] of the recorded brackets in enqueue order, relative to the first one's start
        (ss_profile_timeline); stream: 0 main, 1 side, 2 / 3 the branch streams."""
        buf = (C.c_double * (3 * cap))()
        n = self.lib.ss_profile_timeline(self.h, buf, cap)
        if n < 0:
            _capi.check(n)
        names = self.PROF_CLASSES + self.PROF_TIMELINE
        return [(names[int(buf[3 * i]) % 100], buf[3 * i + 1], buf[3 * i + 2], int(buf[3 * i]) // 100) for i in range(n)]

    def debug_buffer(self, name, B, T):
        """Real frames of an internal haloed slab as a [B, T, C] tensor (copy)."""
        p, rows, cols = C.c_void_p(), C.c_long(), C.c_long()
        _capi.check(self.lib.ss_debug_buffer(self.h, name.encode(), C.byref(p), C.byref(rows), C.byref(cols)))
        off = p.value - self.ws.data_ptr()
        n = rows.value * cols.value
        flat = self.ws[off:off + 4 * n].view(torch.float32)
        return flat.view(B, T + 4, cols.value)[:, 2:2 + T].clone()


def split_image(x, scale=16.0):
    """Test hook (ss_op_split_image): fp32 [rows, cols] (cols % 8 == 0) -> its operand image, a float32-typed tensor of the same shape whose
    bytes are, per 8 elements, 16 B of fp16 hi pieces and 16 B of fp16 lo pieces of scale * x (csrc/common.h, image format v2)."""
    lib = _capi.lib()
    rows, cols = x.shape
    x = x.contiguous()
    img = torch.empty_like(x)
    _capi.check(lib.ss_op_split_image(_ptr(x), x.stride(0), rows, cols, float(scale), _ptr(img), img.stride(0), _stream()))
    return img


_ZEROS = {}


def gemm_img(a_img, b_img, ta=False, tb=False, bias=None, ksplit=1, cfg=-1, scale_a=16.0, scale_b=16.0, out=None, accumulate=False, a_seg=(0, 0),
             M=None, K=None, part=None):
    """Test hook (ss_op_gemm_img): C[M,N] = A(m,k) B(n,k) over operand images.  a_img [M,K] ([K,M] if ta), b_img [N,K] ([K,N] if tb).
    torch.bfloat16 operands select the single-piece form (plain bf16 matrices, no scales)."""
    lib = _capi.lib()
    bf16 = a_img.dtype == torch.bfloat16
    assert (b_img.dtype == torch.bfloat16) == bf16
    if M is None:
        M = a_img.shape[1] if ta else a_img.shape[0]
    if K is None:
        K = a_img.shape[0] if ta else a_img.shape[1]
    N = b_img.shape[1] if tb else b_img.shape[0]
    dev = a_img.device
    c = torch.zeros(M, N, device=dev) if out is None else out
    if ksplit > 1 and part is None:
        part = torch.empty(ksplit * M * N, device=dev)
    z = _ZEROS.setdefault(str(dev), torch.zeros(1024, device=dev))
    _capi.check(lib.ss_op_gemm_img(_ptr(a_img), a_img.stride(0), _ptr(b_img), b_img.stride(0), _ptr(c), c.stride(0), _ptr(bias), M, N, K,
                                   (1 if ta else 0) | (2 if tb else 0) | (4 if accumulate else 0) | (8 if bf16 else 0), int(ksplit), int(cfg), float(scale_a),
                                   float(scale_b), int(a_seg[0]), int(a_seg[1]), _ptr(part), _ptr(z), _stream()))
    return c


def conv_block(x, w, bias, gamma, beta, dy=None, need_dx=True):
    """Test hook (ss_op_conv_block): relu(GroupNorm(conv5(x))) of one block through the engine's block routines.
    x [B,T,Ci] -> y [B,T,Co]; with dy also (dx, gw, gb, ggamma, gbeta)."""
    lib = _capi.lib()
    B, T, Ci = x.shape
    Co = w.shape[0]
    dev = x.device
    f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    x, w, bias, gamma, beta = f(x), f(w), f(bias), f(gamma), f(beta)
    n = lib.ss_op_conv_block_scratch(B, T, Ci, Co)
    scratch = torch.empty(n, device=dev)
    y = torch.empty(B, T, Co, device=dev)
    if dy is None:
        _capi.check(lib.ss_op_conv_block(_ptr(x), _ptr(w), _ptr(bias), _ptr(gamma), _ptr(beta), None, _ptr(y), None, None, None,
                                         None, None, _ptr(scratch), n, B, T, Ci, Co, _stream()))
        return y
    dy = f(dy)
    dx = torch.empty(B, T, Ci, device=dev) if need_dx else None
    gw, gb, gg, gbe = torch.empty_like(w), torch.empty_like(bias), torch.empty_like(gamma), torch.empty_like(beta)
    _capi.check(lib.ss_op_conv_block(_ptr(x), _ptr(w), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(dy), _ptr(y), _ptr(dx), _ptr(gw),
                                     _ptr(gb), _ptr(gg), _ptr(gbe), _ptr(scratch), n, B, T, Ci, Co, _stream()))
    return y, dx, gw, gb, gg, gbe


def _slab(x):
    """[B,T,C] -> haloed slab [B,T+4,C] (kernels.h: frame t at row t+2, zero halo rows)."""
    B, T, Cc = x.shape
    s = torch.zeros(B, T + 4, Cc, device=x.device)
    s[:, 2:2 + T] = x
    return s


def lstm_wgrad(dg, x, hout):
    """Test hook (ss_op_lstm_wgrad): the fused weight / bias gradient kernel of the encoder BLSTMs.  dg [R, 8H], x [R, In] (may be a
    column view of a wider tensor), hout [R, 2H] -> (gw_ih [2, 4H, In], gw_hh [2, 4H, H], gb [2, 2, 4H])."""
    lib = _capi.lib()
    R, H8 = dg.shape
    H, In = H8 // 8, x.shape[1]
    dev = dg.device
    tiles = ((8 * H + 63) // 64) * ((In + 63) // 64 + (1 if H >= 16 else 2))
    scratch = torch.empty(16 * 4096 * tiles + 256, device=dev)
    gwih, gwhh, gb = torch.zeros(2, 4 * H, In, device=dev), torch.zeros(2, 4 * H, H, device=dev), torch.zeros(2, 2, 4 * H, device=dev)
    _capi.check(lib.ss_op_lstm_wgrad(_ptr(dg), _ptr(x), x.stride(0), _ptr(hout), _ptr(gwih), _ptr(gwhh), _ptr(gb), _ptr(scratch), scratch.numel(), R, H, In,
                                     _stream()))
    return gwih, gwhh, gb


def blstm_layer(x, w_ih, w_hh, b_ih, b_hh, d_out=None):
    """Test hook: one bidirectional LSTM layer through ss_op_lstm_fwd / ss_op_lstm_bwd (the engine's recurrence kernels) with
    the input projection and the weight / input gradients on the engine's GEMM (ss_op_gemm).  w_ih etc. are (forward, reverse)
    pairs with PyTorch's shapes.  Returns out [B,T,2H]; with d_out also (dx, [(gw_ih, gw_hh, gb) per direction])."""
    lib = _capi.lib()
    B, T, In = x.shape
    H = w_hh[0].shape[1]
    dev = x.device
    wcat = torch.cat([w_ih[0], w_ih[1]], 0).contiguous()                    # [8H, In]
    bsum = torch.cat([b_ih[0] + b_hh[0], b_ih[1] + b_hh[1]]).contiguous()
    xs = _slab(x)
    R = B * (T + 4)
    gates = torch.zeros(R, 8 * H, device=dev)
    g_real = gemm(xs.view(R, In), wcat, bsum)                               # all rows, halo rows are then re-zeroed
    gates.copy_(g_real)
    gates.view(B, T + 4, 8 * H)[:, :2] = 0
    gates.view(B, T + 4, 8 * H)[:, T + 2:] = 0
    out = torch.zeros(B, T + 4, 2 * H, device=dev)
    csave = torch.zeros(B, T + 4, 2 * H, device=dev)
    B16 = (B + 15) // 16 * 16
    nscr = 8 * H * H + 16 * B16 * H + 2 * B * H + (4 * ((B + 15) // 16) * (H // 16) ** 2 * 1024 + 8192) // 4 + 4096
    scratch = torch.zeros(max(nscr, 1), device=dev)
    whf, whb = w_hh[0].contiguous(), w_hh[1].contiguous()
    _capi.check(lib.ss_op_lstm_fwd(_ptr(gates), _ptr(whf), _ptr(whb), _ptr(out), _ptr(csave), _ptr(scratch), scratch.numel(),
                                   B, T, H, _stream()))
    y = out[:, 2:2 + T].clone()
    if d_out is None:
        return y
    ds = _slab(d_out.to(dev))
    scratch.zero_()
    _capi.check(lib.ss_op_lstm_bwd(_ptr(gates), _ptr(whf), _ptr(whb), _ptr(ds), _ptr(csave), _ptr(scratch), scratch.numel(),
                                   B, T, H, _stream()))
    dG = gates.view(R, 8 * H)                                               # pre-activation gradients, halo rows zero
    dx = gemm(dG, wcat, tb=True).view(B, T + 4, In)[:, 2:2 + T].clone()     # dX = dG . W_ih (both directions)
    grads = []
    flat_x = xs.view(R, In)
    flat_o = out.view(R, 2 * H)
    for d in range(2):
        dGd = dG[:, d * 4 * H:(d + 1) * 4 * H].contiguous()
        gw_ih = gemm(dGd, flat_x, ta=True, tb=True)                         # [4H, In] = dG^T . X
        hprev = torch.zeros(R, H, device=dev)
        if d == 0:
            hprev[1:] = flat_o[:-1, :H]                                     # forward: h(t-1) is one slab row earlier
        else:
            hprev[:-1] = flat_o[1:, H:]
        gw_hh = gemm(dGd, hprev, ta=True, tb=True)
        grads.append((gw_ih, gw_hh, dGd.sum(0)))
    return y, dx, grads


def tune(key, value):
    """Process-global tuning knob of the HIP library (ss_tune in include/speechsplit_amd.h)."""
    _capi.check(_capi.lib().ss_tune(key.encode(), int(value)))


def gemm(a, b, bias=None, ta=False, tb=False, ksplit=1, out=None, bf16=False, f16x2=False):
    """Test hook for the MFMA GEMM: C[M,N] = A(m,k) B(n,k) (+bias).  a: [M,K] or [K,M] if ta; b: [N,K] or [K,N] if tb."""
    lib = _capi.lib()
    M = a.shape[1] if ta else a.shape[0]
    K = a.shape[0] if ta else a.shape[1]
    N = b.shape[1] if tb else b.shape[0]
    assert (b.shape[0] if tb else b.shape[1]) == K
    c = torch.zeros(M, N, device=a.device) if out is None else out
    _capi.check(lib.ss_op_gemm(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(c), N, _ptr(bias), M, N, K,
                               (1 if ta else 0) | (2 if tb else 0) | (8 if bf16 else 0) | (16 if f16x2 else 0), ksplit, _stream()))
    return c
