"""numpy restatement of the random-resampling bottleneck and the F0 quantiser.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, with the random draws
as explicit inputs:

* ``InterpLnr.forward``/``pad_sequences``  -- reference model.py:368-436
* ``quantize_f0_torch``                    -- reference utils.py:62-74

All float arithmetic is IEEE fp32 with separate mul/mul/add (numpy never fuses),
which is what the reference's three ATen ops do (model.py:430).
"""
import numpy as np

MAX_NUM_SEG_DEFAULT = 7     # max_len_seq // min_len_seg + 1 = 128 // 19 + 1   (model.py:365)


def interp_plan(scales, len_seg, len_seq, max_len_seg=32, max_len_pad=192):
    """Index path of InterpLnr (model.py:389-418).

    scales  f32[B*S]  = rand(B*S) + 0.5            (model.py:392-393)
    len_seg int[B*S]  = randint(min_len_seg, max_len_seg)   (model.py:399-402)
    len_seq int[B]
    Returns (i0 int32[B, max_len_pad], lam f32[B, max_len_pad], counts int32[B],
    nrows int32[B]) where ``counts`` is the un-truncated number of selected
    positions (model.py:418) and ``nrows = min(counts, max_len_pad)`` the rows
    that survive pad_sequences (model.py:375); entries beyond nrows are 0.
    """
    scales = np.asarray(scales, dtype=np.float32).reshape(-1)
    len_seg = np.asarray(len_seg).reshape(-1).astype(np.int64)
    len_seq = np.asarray(len_seq).reshape(-1).astype(np.int64)
    B = len_seq.shape[0]
    S = scales.shape[0] // B
    assert scales.shape[0] == B * S and len_seg.shape[0] == B * S
    j = np.arange(2 * max_len_seg, dtype=np.int64).astype(np.float32)       # :389
    q = (j[None, :] / scales[:, None]).astype(np.float32)                   # :395 fp32 divide
    fl = np.floor(q)                                                        # :396
    lam = (q - fl).astype(np.float32)                                       # :397
    m_seg = fl < (len_seg[:, None] - 1).astype(np.float32)                  # :405
    csum = np.cumsum(len_seg.reshape(B, S), axis=1)                         # :407
    off = np.concatenate([np.zeros((B, 1), np.int64), csum[:, :-1]], 1)     # :409
    org = (fl + off.reshape(-1, 1).astype(np.float32)).astype(np.float32)   # :411
    lim = (np.repeat(len_seq, S) - 1).astype(np.float32)                    # :413-414
    mask = m_seg & (org < lim[:, None])                                     # :416
    counts = mask.reshape(B, -1).sum(1).astype(np.int32)                    # :418
    i0 = np.zeros((B, max_len_pad), np.int32)
    lm = np.zeros((B, max_len_pad), np.float32)
    nrows = np.minimum(counts, max_len_pad).astype(np.int32)
    org_b = org.reshape(B, -1)
    lam_b = lam.reshape(B, -1)
    mask_b = mask.reshape(B, -1)
    for b in range(B):                                                      # :420-423, 432-434
        sel = np.nonzero(mask_b[b])[0][: max_len_pad]
        i0[b, : sel.size] = org_b[b, sel].astype(np.int64)
        lm[b, : sel.size] = lam_b[b, sel]
    return i0, lm, counts, nrows


def interp_apply(x, i0, lam, nrows):
    """Value path (model.py:426-430, 368-377): y = (1-lam)*x[i0] + lam*x[i0+1], zero padded."""
    x = np.asarray(x, dtype=np.float32)
    B, T, C = x.shape
    P = i0.shape[1]
    y = np.zeros((B, P, C), np.float32)
    one = np.float32(1.0)
    for b in range(B):
        n = int(nrows[b])
        if n == 0:
            continue
        a = x[b, i0[b, :n], :]
        c = x[b, i0[b, :n] + 1, :]
        l = lam[b, :n, None]
        y[b, :n] = ((one - l) * a).astype(np.float32) + (l * c).astype(np.float32)
    return y


def interp_forward(x, len_seq, scales, len_seg, max_len_seg=32, max_len_pad=192):
    i0, lam, counts, nrows = interp_plan(scales, len_seg, len_seq, max_len_seg, max_len_pad)
    return interp_apply(x, i0, lam, nrows)


def interp_backward(dy, i0, lam, nrows, T):
    """Adjoint of interp_apply w.r.t. x (what autograd's index_put_(accumulate) does)."""
    dy = np.asarray(dy, dtype=np.float32)
    B, P, C = dy.shape
    dx = np.zeros((B, T, C), np.float32)
    one = np.float32(1.0)
    for b in range(B):
        for r in range(int(nrows[b])):
            l = lam[b, r]
            dx[b, i0[b, r]] += (one - l) * dy[b, r]
            dx[b, i0[b, r] + 1] += l * dy[b, r]
    return dx


def quantize_f0(x, num_bins=256):
    """utils.py:62-74.  x f32[...] in [0,1] (<=0 means unvoiced) -> class index int64[...].

    index 0 = unvoiced, else round_half_even(x * 255) + 1.  The one-hot the
    reference returns is ``np.eye(257)[index]``.
    """
    x = np.asarray(x, dtype=np.float32)
    uv = x <= 0
    xc = np.where(uv, np.float32(0), x)
    assert (xc >= 0).all() and (xc <= 1).all()
    idx = np.rint(xc * np.float32(num_bins - 1)).astype(np.int64) + 1      # np.rint = half-to-even
    idx[uv] = 0
    return idx


def onehot(idx, n=257):
    out = np.zeros(idx.shape + (n,), np.float32)
    np.put_along_axis(out, idx[..., None], 1.0, axis=-1)
    return out
