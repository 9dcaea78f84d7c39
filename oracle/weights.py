"""Parameter table of Generator_3 / Generator_6 and a frozen-stream weight generator.

TEST INFRASTRUCTURE (see oracle/__init__.py).

``param_spec`` restates the reference's ``state_dict()`` key order and shapes
(module registration order of model.py:49-71, 96-121, 147-191, 236-247, 262-271,
285-290, 327-332); ``tests/golden/keys_*.json`` (written by gen_fixtures.py from
the imported reference) pins it.

``make_weights`` draws every tensor from ``numpy.random.RandomState`` (the legacy
MT19937 stream, frozen by numpy's compatibility policy) so a fixture only has to
store a seed, not 77 MB of weights.  Magnitudes follow the reference's
initialisers (model.py:15-17, 37-38; torch defaults for the rest).
"""
import math
import numpy as np


class HP(dict):
    """Attribute bag with the reference's hyper-parameter names (hparams.py:7-43)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def default_hparams(**over):
    hp = HP(freq=8, dim_neck=8, freq_2=8, dim_neck_2=1, freq_3=8, dim_neck_3=32,
            dim_enc=512, dim_enc_2=128, dim_enc_3=256, dim_freq=80, dim_spk_emb=82,
            dim_f0=257, dim_dec=512, len_raw=128, chs_grp=16,
            min_len_seg=19, max_len_seg=32, min_len_seq=64, max_len_seq=128, max_len_pad=192)
    hp.update(over)
    return hp


def _conv_block(prefix, cin, cout):
    return [(prefix + '.0.conv.weight', (cout, cin, 5)), (prefix + '.0.conv.bias', (cout,)),
            (prefix + '.1.weight', (cout,)), (prefix + '.1.bias', (cout,))]


def _lstm(prefix, cin, hid, layers):
    out = []
    for l in range(layers):
        i = cin if l == 0 else 2 * hid
        for sfx in ('', '_reverse'):
            out += [(f'{prefix}.weight_ih_l{l}{sfx}', (4 * hid, i)),
                    (f'{prefix}.weight_hh_l{l}{sfx}', (4 * hid, hid)),
                    (f'{prefix}.bias_ih_l{l}{sfx}', (4 * hid,)),
                    (f'{prefix}.bias_hh_l{l}{sfx}', (4 * hid,))]
    return out


def _encoder_t(prefix, hp):
    return (_conv_block(prefix + '.convolutions.0', hp.dim_freq, hp.dim_enc_2)
            + _lstm(prefix + '.lstm', hp.dim_enc_2, hp.dim_neck_2, 1))


def param_spec(kind, hp):
    """[(name, shape)] in state_dict order, parameters only (the int64 buffer
    ``<enc>.len_org`` is listed by ``buffer_spec``)."""
    s = []
    if kind == 'G3':
        e = 'encoder_1'
        for i in range(3):
            s += _conv_block(f'{e}.convolutions_1.{i}', hp.dim_freq if i == 0 else hp.dim_enc, hp.dim_enc)
        s += _lstm(f'{e}.lstm_1', hp.dim_enc, hp.dim_neck, 2)
        for i in range(3):
            s += _conv_block(f'{e}.convolutions_2.{i}', hp.dim_f0 if i == 0 else hp.dim_enc_3, hp.dim_enc_3)
        s += _lstm(f'{e}.lstm_2', hp.dim_enc_3, hp.dim_neck_3, 1)
        s += _encoder_t('encoder_2', hp)
        din = 2 * hp.dim_neck + 2 * hp.dim_neck_2 + 2 * hp.dim_neck_3 + hp.dim_spk_emb
        s += _lstm('decoder.lstm', din, 512, 3)
        s += [('decoder.linear_projection.linear_layer.weight', (hp.dim_freq, 1024)),
              ('decoder.linear_projection.linear_layer.bias', (hp.dim_freq,))]
    elif kind == 'G6':
        s += _encoder_t('encoder_2', hp)
        e = 'encoder_3'
        for i in range(3):
            s += _conv_block(f'{e}.convolutions.{i}', hp.dim_f0 if i == 0 else hp.dim_enc_3, hp.dim_enc_3)
        s += _lstm(f'{e}.lstm', hp.dim_enc_3, hp.dim_neck_3, 1)
        din = 2 * hp.dim_neck_2 + 2 * hp.dim_neck_3
        s += _lstm('decoder.lstm', din, 256, 2)
        s += [('decoder.linear_projection.linear_layer.weight', (hp.dim_f0, 512)),
              ('decoder.linear_projection.linear_layer.bias', (hp.dim_f0,))]
    else:
        raise ValueError(kind)
    return s


def buffer_spec(kind):
    return ['encoder_1.len_org'] if kind == 'G3' else ['encoder_3.len_org']


def make_weights(kind, hp, seed):
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in param_spec(kind, hp):
        n = int(np.prod(shape))
        u = rs.uniform(-1.0, 1.0, size=n).astype(np.float32).reshape(shape)
        if name.endswith('conv.weight'):
            co, ci, k = shape
            bound = math.sqrt(2.0) * math.sqrt(6.0 / ((ci + co) * k))
            w = u * np.float32(bound)
        elif name.endswith('conv.bias'):
            # fan_in is not recoverable from the bias shape; a fixed small scale is enough
            w = u * np.float32(0.05)
        elif '.lstm' in name:
            hid = shape[0] // 4
            w = u * np.float32(1.0 / math.sqrt(hid))
        elif name.endswith('linear_layer.weight'):
            w = u * np.float32(math.sqrt(6.0 / (shape[0] + shape[1])))
        elif name.endswith('linear_layer.bias'):
            w = u * np.float32(1.0 / 32.0)
        elif name.endswith('.1.weight'):           # GroupNorm gamma: non-trivial on purpose
            w = np.float32(1.0) + u * np.float32(0.1)
        elif name.endswith('.1.bias'):             # GroupNorm beta
            w = u * np.float32(0.1)
        else:
            raise KeyError(name)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out
