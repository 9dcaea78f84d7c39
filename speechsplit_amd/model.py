"""Host-side mirror of the reference's ``model.py`` surface (boundary B1).

``Generator_3`` / ``Generator_6`` / ``InterpLnr`` are ``nn.Module``s with the reference's constructor and
``forward`` signatures (reference model.py:283-351, 355-436) and the reference's exact ``state_dict`` keys, so
``solver.py`` / ``demo.ipynb`` code written against the reference runs against them unchanged:

    G = Generator_3(hparams); opt = torch.optim.Adam(G.parameters(), ...); G.to('cuda:0')
    out = G(x_f0, x_org, c_trg); loss.backward(); opt.step(); G.state_dict(); G.load_state_dict(sd)

There is no PyTorch compute in here.  ``.to(cuda)`` creates the HIP engine and re-points every Parameter's
storage at the engine's flat parameter arena (Parameter objects keep their identity, so an optimizer built
before ``.to`` -- as solver.py:62-65 does -- keeps working); ``forward`` / ``backward`` are one C-ABI call each
through a ``torch.autograd.Function``.  On CPU the modules only hold parameters: calling them raises.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import engine as _engine


def _hp_get(hp, name, default):
    """Optional hyper-parameter: attribute bags differ in what a missing name raises."""
    try:
        return getattr(hp, name)
    except (AttributeError, KeyError):
        return default


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted parameter names."""

    def forward(self, *a, **k):
        raise RuntimeError('speechsplit_amd: sub-modules are name containers; call the Generator itself')


def _spec(kind, hp):
    """[(name, shape, init)] in the reference's parameters() order.  init: ('xavier', gain) | ('conv_bias', fan_in) |
    ('ones',) | ('zeros',) | ('lstm', hidden) | ('linear_bias', fan_in)."""
    relu_gain = math.sqrt(2.0)                       # calculate_gain('relu'), model.py:66

    def conv(pre, ci, co):
        return [(pre + '.0.conv.weight', (co, ci, 5), ('xavier', relu_gain)), (pre + '.0.conv.bias', (co,), ('conv_bias', ci * 5)),
                (pre + '.1.weight', (co,), ('ones',)), (pre + '.1.bias', (co,), ('zeros',))]

    def lstm(pre, cin, hid, layers):
        out = []
        for l in range(layers):
            i = cin if l == 0 else 2 * hid
            for sfx in ('', '_reverse'):
                out += [(f'{pre}.weight_ih_l{l}{sfx}', (4 * hid, i), ('lstm', hid)), (f'{pre}.weight_hh_l{l}{sfx}', (4 * hid, hid), ('lstm', hid)),
                        (f'{pre}.bias_ih_l{l}{sfx}', (4 * hid,), ('lstm', hid)), (f'{pre}.bias_hh_l{l}{sfx}', (4 * hid,), ('lstm', hid))]
        return out

    def enc_t(pre):
        return conv(pre + '.convolutions.0', hp.dim_freq, hp.dim_enc_2) + lstm(pre + '.lstm', hp.dim_enc_2, hp.dim_neck_2, 1)

    s = []
    if kind == 'G3':
        for i in range(3):
            s += conv(f'encoder_1.convolutions_1.{i}', hp.dim_freq if i == 0 else hp.dim_enc, hp.dim_enc)
        s += lstm('encoder_1.lstm_1', hp.dim_enc, hp.dim_neck, 2)
        for i in range(3):
            s += conv(f'encoder_1.convolutions_2.{i}', hp.dim_f0 if i == 0 else hp.dim_enc_3, hp.dim_enc_3)
        s += lstm('encoder_1.lstm_2', hp.dim_enc_3, hp.dim_neck_3, 1)
        s += enc_t('encoder_2')
        din = 2 * hp.dim_neck + 2 * hp.dim_neck_2 + 2 * hp.dim_neck_3 + hp.dim_spk_emb
        s += lstm('decoder.lstm', din, 512, 3)
        s += [('decoder.linear_projection.linear_layer.weight', (hp.dim_freq, 1024), ('xavier', 1.0)),
              ('decoder.linear_projection.linear_layer.bias', (hp.dim_freq,), ('linear_bias', 1024))]
    else:
        s += enc_t('encoder_2')
        for i in range(3):
            s += conv(f'encoder_3.convolutions.{i}', hp.dim_f0 if i == 0 else hp.dim_enc_3, hp.dim_enc_3)
        s += lstm('encoder_3.lstm', hp.dim_enc_3, hp.dim_neck_3, 1)
        s += lstm('decoder.lstm', 2 * hp.dim_neck_2 + 2 * hp.dim_neck_3, 256, 2)
        s += [('decoder.linear_projection.linear_layer.weight', (hp.dim_f0, 512), ('xavier', 1.0)),
              ('decoder.linear_projection.linear_layer.bias', (hp.dim_f0,), ('linear_bias', 512))]
    return s


def _init_tensor(shape, init):
    """Same distributions as the reference's initialisers (model.py:15-17, 37-38 and torch defaults)."""
    t = torch.empty(*shape)
    kind = init[0]
    if kind == 'xavier':
        nn.init.xavier_uniform_(t, gain=init[1])
    elif kind in ('conv_bias', 'linear_bias'):
        b = 1.0 / math.sqrt(init[1])
        nn.init.uniform_(t, -b, b)
    elif kind == 'ones':
        nn.init.ones_(t)
    elif kind == 'zeros':
        nn.init.zeros_(t)
    elif kind == 'lstm':
        b = 1.0 / math.sqrt(init[1])
        nn.init.uniform_(t, -b, b)
    return t


def init_weights(kind, hp, seed=0):
    """{state_dict name: fp32 tensor} drawn with the reference's initialisers (``_spec`` / ``_init_tensor``) from a generator
    seeded with `seed`; the global RNG state is left alone."""
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        return {name: _init_tensor(shape, init) for name, shape, init in _spec(kind, hp)}


class _G3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x_f0, x_org, c_trg, draws, *params):
        ctx.mod = mod
        out = mod._eng.g3_forward(x_f0, x_org, c_trg, draws, training=mod.training)
        ctx.nparams = len(params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        mod = ctx.mod
        mod._eng.g3_backward(d_out.contiguous())
        flat = mod._eng.grads.clone()                 # fresh storage: autograd may keep or accumulate these
        gv = mod._eng.views(flat)
        return (None, None, None, None, None) + tuple(gv[n] for n in mod._names)


class _G6Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x_org, f0_trg, draws, *params):
        ctx.mod = mod
        return mod._eng.g6_forward(x_org, f0_trg, draws, training=mod.training)

    @staticmethod
    def backward(ctx, d_out):
        mod = ctx.mod
        mod._eng.g6_backward(d_out.contiguous())
        flat = mod._eng.grads.clone()
        gv = mod._eng.views(flat)
        return (None, None, None, None) + tuple(gv[n] for n in mod._names)


class _EngineModule(nn.Module):
    KIND = None

    def __init__(self, hparams, max_batch=None):
        super().__init__()
        self.hparams_ = hparams
        self._eng = None
        self._max_batch = max_batch
        self._names = []
        self._plist = []
        for name, shape, init in _spec(self.KIND, hparams):
            node = self
            parts = name.split('.')
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            prm = nn.Parameter(_init_tensor(shape, init))
            node.register_parameter(parts[-1], prm)
            self._names.append(name)
            self._plist.append(prm)
        # the reference registers this int64 buffer on its resampling encoders (model.py:105,157)
        enc = self._modules['encoder_1' if self.KIND == 'G3' else 'encoder_3']
        enc.register_buffer('len_org', torch.tensor(hparams.max_len_pad))

    # ---- device placement: adopt the engine arena as parameter storage
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        dev = self._plist[0].device
        if dev.type == 'cuda':
            self._ensure_engine(dev)
        else:
            self._eng = None
        return self

    def _ensure_engine(self, dev, batch=None):
        need = batch or self._max_batch or _hp_get(self.hparams_, 'batch_size', 16)
        if self._eng is not None and self._eng.device == dev and self._eng.max_batch >= need:
            return
        old = {n: p.data.detach().clone() for n, p in zip(self._names, self._plist)}
        eng = _engine.Engine(self.KIND, self.hparams_, max(need, self._max_batch or 0),
                             max(self.hparams_.max_len_pad, 192), device=dev)
        eng.set_precision(_hp_get(self.hparams_, 'precision', 'f32'))   # optional extra hparam: 'f32' (reference arithmetic) | 'bf16'
        eng.load_weights(old)
        pv = eng.param_views()
        for n, p in zip(self._names, self._plist):
            p.data = pv[n]                            # same Parameter object, storage now inside the arena
        self._eng = eng

    def _draw(self, B):
        hp = self.hparams_
        return _engine.draw_interp(B, 3, hp) if self.training else None

    def extra_repr(self):
        return f'speechsplit_amd HIP engine ({self.KIND}), {sum(p.numel() for p in self._plist)} parameters'


class Generator_3(_EngineModule):
    """SpeechSplit model (reference model.py:283-320)."""
    KIND = 'G3'

    def forward(self, x_f0, x_org, c_trg, draws=None):
        if not x_org.is_cuda:
            raise RuntimeError('speechsplit_amd.Generator_3 runs on a ROCm GPU only: call .to("cuda") first')
        self._ensure_engine(x_org.device, x_org.shape[0])
        if draws is None:
            draws = self._draw(x_org.shape[0])
        return _G3Fn.apply(self, x_f0, x_org, c_trg, draws, *self._plist)

    def rhythm(self, x_org):
        self._ensure_engine(x_org.device, x_org.shape[0])
        return self._eng.g3_rhythm(x_org)


class Generator_6(_EngineModule):
    """F0 converter (reference model.py:324-351)."""
    KIND = 'G6'

    def forward(self, x_org, f0_trg, draws=None):
        if not x_org.is_cuda:
            raise RuntimeError('speechsplit_amd.Generator_6 runs on a ROCm GPU only: call .to("cuda") first')
        self._ensure_engine(x_org.device, x_org.shape[0])
        if draws is None:
            draws = self._draw(x_org.shape[0])
        return _G6Fn.apply(self, x_org, f0_trg, draws, *self._plist)


class _InterpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, len_seq, scales, len_seg):
        ctx.mod, ctx.T = mod, x.shape[1]
        return mod._eng.interp_forward(x, len_seq, scales, len_seg)

    @staticmethod
    def backward(ctx, dy):
        return None, ctx.mod._eng.interp_backward(dy.contiguous(), ctx.T), None, None, None


class InterpLnr(nn.Module):
    """Random resampling (reference model.py:355-436).  ``forward(x, len_seq)`` in training mode draws the segment
    scales / lengths from the default CPU generator in the reference's order; pass ``draws=(scales, len_seg)`` to
    replay recorded draws (bit-exact index path)."""

    def __init__(self, hparams, max_batch=None, max_frames=None):
        super().__init__()
        self.hparams_ = hparams
        self.max_len_seq, self.max_len_pad = hparams.max_len_seq, hparams.max_len_pad
        self.min_len_seg, self.max_len_seg = hparams.min_len_seg, hparams.max_len_seg
        self.max_num_seg = self.max_len_seq // self.min_len_seg + 1
        self._eng = None
        self._max_batch, self._max_frames = max_batch, max_frames

    def forward(self, x, len_seq, draws=None):
        if not self.training:
            return x                                   # model.py:382-383
        if not x.is_cuda:
            raise RuntimeError('speechsplit_amd.InterpLnr runs on a ROCm GPU only')
        B, T, _ = x.shape
        if self._eng is None or self._eng.device != x.device or self._eng.max_batch < B or self._eng.max_frames < T:
            self._eng = _engine.Engine('interp', self.hparams_, max(B, self._max_batch or 0),
                                       max(T, self._max_frames or 0, self.max_len_pad), device=x.device)
        if draws is None:
            sc, ls = _engine.draw_interp(B, 1, self.hparams_)
            draws = (sc[0], ls[0])
        return _InterpFn.apply(self, x, len_seq, draws[0], draws[1])
