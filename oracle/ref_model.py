"""PyTorch-CPU functional restatement of the SpeechSplit hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pure functions over a dict ``P``
of fp32 tensors keyed by the reference's ``state_dict`` names.  The random
resampling draws are explicit inputs (reference model.py:392-393, 399-402 draws
them from the global generator), everything else follows the cited lines.

The arithmetic that lives in PyTorch itself (conv1d, group_norm, lstm, linear,
Adam -- SURVEY.md section 8(a) row X1; torch is unpinned ">= 1.2.0" in the
reference README.md:27, 2.10.0+rocm7.0 here) is called, not restated;
``lstm_explicit`` additionally restates the LSTM cell so the gate order the HIP
kernels assume is itself pinned against torch.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import interp_np

TAP = None      # set to a dict to record intermediates (name -> [B,T,C] tensor) for kernel-level debugging
# ReLU branch override for gradient parity tests.  A GroupNorm output within fp32 rounding of 0 lands on either side of the
# kink in two correct fp32 implementations, and the side decides a whole channel's gradient.  MASK = {block: bool [B,T,C]}
# (the implementation under test reports the branch it took) makes this restatement take the same branch; MASK_STATS then
# records, per block, how many elements disagreed with this restatement's own sign and the largest |z| among them, so a
# test can assert that the override only ever acted AT the kink (|z| ~ 1e-6), never on a clearly signed value.
MASK = None
MASK_STATS = None


def _tap(name, t, nct=False):
    if TAP is not None:
        TAP[name] = (t.transpose(1, 2) if nct else t).detach().clone()


# --------------------------------------------------------------------------- blocks
def conv_gn_relu(x_nct, P, prefix, chs_grp=16, tap=None):
    """relu(GroupNorm(Conv1d(k=5,p=2)))  -- model.py:61-67,76-77 / 109-115,125-126 / 164-185,200-201."""
    w = P[prefix + '.0.conv.weight']
    y = F.conv1d(x_nct, w, P[prefix + '.0.conv.bias'], stride=1, padding=2, dilation=1)
    if tap:
        _tap(tap, y, nct=True)
    y = F.group_norm(y, w.shape[0] // chs_grp, P[prefix + '.1.weight'], P[prefix + '.1.bias'], eps=1e-5)
    if tap and TAP is not None:
        TAP['zmin:' + tap] = float(y.detach().abs().min())     # distance of the closest pre-activation to the ReLU kink
    blk = tap[:-5] if tap and tap.endswith('.conv') else tap
    if MASK is not None and blk in MASK:
        m = MASK[blk].transpose(1, 2)                          # [B,T,C] -> NCT
        if MASK_STATS is not None:
            dis = m != (y.detach() > 0)
            MASK_STATS[blk] = (int(dis.sum()), float(y.detach().abs()[dis].max()) if bool(dis.any()) else 0.0)
        return y * m.to(y.dtype)
    return F.relu(y)


def _lstm_flat(P, prefix, layers):
    flat = []
    for l in range(layers):
        for sfx in ('', '_reverse'):
            flat += [P[f'{prefix}.weight_ih_l{l}{sfx}'], P[f'{prefix}.weight_hh_l{l}{sfx}'],
                     P[f'{prefix}.bias_ih_l{l}{sfx}'], P[f'{prefix}.bias_hh_l{l}{sfx}']]
    return flat


def blstm(x_btc, P, prefix, layers, tap=None):
    """nn.LSTM(batch_first=True, bidirectional=True), zero initial state -- model.py:71,81 etc.
    Evaluated one layer per aten::lstm call (a stacked LSTM is exactly that) so layer outputs can be tapped."""
    hid = P[f'{prefix}.weight_hh_l0'].shape[1]
    B = x_btc.shape[0]
    h0 = x_btc.new_zeros(2, B, hid)
    flat = _lstm_flat(P, prefix, layers)
    out = x_btc
    for l in range(layers):
        out, _, _ = torch._VF.lstm(out, (h0, h0), flat[8 * l:8 * l + 8], True, 1, 0.0, False, True, True)
        if tap:
            _tap(f'{tap}.out{l}', out)
    return out


def lstm_explicit(x_btc, P, prefix, layers):
    """Same as ``blstm`` written out: gates i,f,g,o; c' = f*c + i*g; h' = o*tanh(c')."""
    hid = P[f'{prefix}.weight_hh_l0'].shape[1]
    B, T, _ = x_btc.shape
    inp = x_btc
    for l in range(layers):
        outs = []
        for d, sfx in enumerate(('', '_reverse')):
            w_ih, w_hh = P[f'{prefix}.weight_ih_l{l}{sfx}'], P[f'{prefix}.weight_hh_l{l}{sfx}']
            b = P[f'{prefix}.bias_ih_l{l}{sfx}'] + P[f'{prefix}.bias_hh_l{l}{sfx}']
            h = inp.new_zeros(B, hid)
            c = inp.new_zeros(B, hid)
            hs = [None] * T
            order = range(T) if d == 0 else range(T - 1, -1, -1)
            for t in order:
                a = inp[:, t] @ w_ih.t() + h @ w_hh.t() + b
                i, f, g, o = a.split(hid, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = torch.sigmoid(o) * torch.tanh(c)
                hs[t] = h
            outs.append(torch.stack(hs, 1))
        inp = torch.cat(outs, -1)
    return inp


def interp(x_btc, len_seq, draw, hp):
    """InterpLnr.forward in train mode with explicit draws -- model.py:380-436."""
    scales, len_seg = draw
    i0, lam, _, nrows = interp_np.interp_plan(np.asarray(scales), np.asarray(len_seg), np.asarray(len_seq),
                                              hp.max_len_seg, hp.max_len_pad)
    B, T, C = x_btc.shape
    i0t = torch.from_numpy(i0.astype(np.int64))
    lamt = torch.from_numpy(lam)[..., None]
    a = torch.gather(x_btc, 1, i0t[..., None].expand(-1, -1, C))
    c = torch.gather(x_btc, 1, (i0t + 1).clamp(max=T - 1)[..., None].expand(-1, -1, C))
    y = (1 - lamt) * a + lamt * c                                            # :430
    keep = (torch.arange(hp.max_len_pad)[None, :] < torch.from_numpy(nrows.astype(np.int64))[:, None])
    return torch.where(keep[..., None], y, torch.zeros((), dtype=y.dtype))   # :368-377


def quantize_f0(x_bt):
    """utils.py:62-74 -> (onehot f32[B,T,257], index int64[B,T])."""
    idx = torch.from_numpy(interp_np.quantize_f0(x_bt.detach().numpy()))
    return F.one_hot(idx, 257).to(torch.float32), idx


def codes_from(out_btc, hid, freq):
    """cat(fwd[:, freq-1::freq], bwd[:, ::freq]) -- model.py:84-87, 134-138, 217-227."""
    return torch.cat((out_btc[:, freq - 1::freq, :hid], out_btc[:, ::freq, hid:]), -1)


# --------------------------------------------------------------------------- modules
def encoder_t(x_nct, P, hp, prefix='encoder_2'):
    """Encoder_t.forward (mask is always None) -- model.py:74-89."""
    x = conv_gn_relu(x_nct, P, prefix + '.convolutions.0', hp.chs_grp, tap='enc2.c.conv')
    _tap('enc2.act', x, nct=True)
    out = blstm(x.transpose(1, 2), P, prefix + '.lstm', 1, tap='enc2.lstm')
    return codes_from(out, hp.dim_neck_2, hp.freq_2)


def encoder_7(x_f0_nct, P, hp, draws, training, prefix='encoder_1'):
    """Encoder_7.forward -- model.py:194-229.  ``draws``: three (scales, len_seg) pairs."""
    x = x_f0_nct[:, :hp.dim_freq]
    f0 = x_f0_nct[:, hp.dim_freq:]
    B = x.shape[0]
    for i in range(3):
        x = conv_gn_relu(x, P, f'{prefix}.convolutions_1.{i}', hp.chs_grp, tap=f'enc1.c1_{i}.conv')
        f0 = conv_gn_relu(f0, P, f'{prefix}.convolutions_2.{i}', hp.chs_grp, tap=f'enc1.c2_{i}.conv')
        xf = torch.cat((x, f0), 1).transpose(1, 2)
        if training:
            xf = interp(xf, np.full(B, hp.max_len_pad), draws[i], hp)       # :203, len_org = max_len_pad
        _tap(f'enc.xf{i}', xf)
        x = xf[:, :, :hp.dim_enc].transpose(1, 2)
        f0 = xf[:, :, hp.dim_enc:].transpose(1, 2)
    ox = blstm(x.transpose(1, 2), P, prefix + '.lstm_1', 2, tap='enc1.lstm1')
    of = blstm(f0.transpose(1, 2), P, prefix + '.lstm_2', 1, tap='enc1.lstm2')
    return codes_from(ox, hp.dim_neck, hp.freq), codes_from(of, hp.dim_neck_3, hp.freq_3)


def encoder_6(f0_nct, P, hp, draws, training, prefix='encoder_3'):
    """Encoder_6.forward -- model.py:123-140."""
    x = f0_nct
    B = x.shape[0]
    for i in range(3):
        x = conv_gn_relu(x, P, f'{prefix}.convolutions.{i}', hp.chs_grp, tap=f'enc3.c_{i}.conv')
        if training:
            x = interp(x.transpose(1, 2), np.full(B, hp.max_len_pad), draws[i], hp).transpose(1, 2)
        _tap(f'enc.xf{i}', x, nct=True)
    out = blstm(x.transpose(1, 2), P, prefix + '.lstm', 1, tap='enc3.lstm')
    return codes_from(out, hp.dim_neck_3, hp.freq_3)


def decoder(x_btc, P, layers, prefix='decoder'):
    """Decoder_3 / Decoder_4 forward -- model.py:249-255, 273-279."""
    _tap('dec.in', x_btc)
    h = blstm(x_btc, P, prefix + '.lstm', layers, tap='dec.lstm')
    return F.linear(h, P[prefix + '.linear_projection.linear_layer.weight'],
                    P[prefix + '.linear_projection.linear_layer.bias'])


def generator_3(P, hp, x_f0, x_org, c_trg, draws=None, training=False):
    """Generator_3.forward -- model.py:297-313.  x_f0 [B,T,337], x_org [B,T,80], c_trg [B,82]."""
    T = x_f0.shape[1]
    cx, cf = encoder_7(x_f0.transpose(2, 1), P, hp, draws, training)
    c2 = encoder_t(x_org.transpose(2, 1), P, hp)
    enc = torch.cat((cx.repeat_interleave(hp.freq, 1), c2.repeat_interleave(hp.freq_2, 1),
                     cf.repeat_interleave(hp.freq_3, 1), c_trg[:, None, :].expand(-1, T, -1)), -1)
    return decoder(enc, P, 3)


def generator_6(P, hp, x_org, f0_trg, draws=None, training=False):
    """Generator_6.forward -- model.py:337-351."""
    c2 = encoder_t(x_org.transpose(2, 1), P, hp)
    c3 = encoder_6(f0_trg.transpose(2, 1), P, hp, draws, training)
    enc = torch.cat((c2.repeat_interleave(hp.freq_2, 1), c3.repeat_interleave(hp.freq_3, 1)), -1)
    return decoder(enc, P, 2)


# --------------------------------------------------------------------------- training step
def g3_loss(P, hp, mel, f0, emb, len_org, draws):
    """solver.py:160-166.  mel [B,T,80], f0 [B,T,1], emb [B,82], len_org int[B]; draws: 4 pairs
    (outer InterpLnr first, then the three encoder layers, i.e. RNG consumption order)."""
    x_f0 = torch.cat((mel, f0), -1)
    xi = interp(x_f0, np.asarray(len_org), draws[0], hp)                     # solver.py:161
    onehot, _ = quantize_f0(xi[:, :, -1])                                    # :162
    x_in = torch.cat((xi[:, :, :-1], onehot), -1)                            # :163
    out = generator_3(P, hp, x_in, mel, emb, draws[1:4], training=True)      # :165
    return F.mse_loss(mel, out, reduction='mean'), out                       # :166


def g6_loss(P, hp, mel, f0_onehot, target_idx, draws):
    """Generator_6 step.  The reference has no training loop for Generator_6;
    cross-entropy over the 257 classes is this repo's choice (demo.ipynb takes argmax of the logits)."""
    logits = generator_6(P, hp, mel, f0_onehot, draws, training=True)
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), target_idx.reshape(-1)), logits


def as_params(weights, requires_grad=True):
    return {k: torch.from_numpy(np.array(v, dtype=np.float32)).requires_grad_(requires_grad)
            for k, v in weights.items()}


class TrainState:
    """Parameters + torch.optim.Adam exactly as solver.py:62 builds it (lr 1e-4, betas (0.9, 0.999))."""

    def __init__(self, weights, lr=1e-4, betas=(0.9, 0.999)):
        self.P = as_params(weights)
        self.opt = torch.optim.Adam(list(self.P.values()), lr, list(betas))

    def load_adam(self, exp_avg, exp_avg_sq, step):
        """Continue from an optimiser state (dicts name -> array, and the step count): what solver.py:84-90 restores from a checkpoint."""
        for n, p in self.P.items():
            self.opt.state[p] = {'step': torch.tensor(float(step)),
                                 'exp_avg': torch.as_tensor(np.array(exp_avg[n], dtype=np.float32)).clone().reshape(p.shape),
                                 'exp_avg_sq': torch.as_tensor(np.array(exp_avg_sq[n], dtype=np.float32)).clone().reshape(p.shape)}

    def step_g3(self, hp, mel, f0, emb, len_org, draws):
        loss, out = g3_loss(self.P, hp, mel, f0, emb, len_org, draws)
        self.opt.zero_grad()                                                 # solver.py:170
        loss.backward()                                                      # :171
        self.opt.step()                                                      # :172
        return loss.detach(), out.detach()

    def step_g6(self, hp, mel, f0_onehot, target_idx, draws):
        """The same three statements around g6_loss (this repo's choice of loss, see g6_loss)."""
        loss, out = g6_loss(self.P, hp, mel, f0_onehot, target_idx, draws)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach(), out.detach()
