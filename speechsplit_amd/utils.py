"""Host-side helpers with the reference's names (reference utils.py:46-87).  The DSP half of the reference's
utils.py (butter_highpass / pySTFT / speaker_normalization, used only by make_spect_f0.py) is out of scope."""
import numpy as np
import torch


def quantize_f0_numpy(x, num_bins=256):
    """utils.py:46-58: log-F0 in [0,1] (<= 0 unvoiced) -> (one-hot f32[L, 257], class index int64[L])."""
    assert x.ndim == 1
    x = x.astype(float).copy()
    uv = x <= 0
    x[uv] = 0.0
    assert (x >= 0).all() and (x <= 1).all()
    idx = np.round(x * (num_bins - 1)) + 1
    idx[uv] = 0.0
    enc = np.zeros((len(x), num_bins + 1), dtype=np.float32)
    enc[np.arange(len(x)), idx.astype(np.int32)] = 1.0
    return enc, idx.astype(np.int64)


def quantize_f0_torch(x, num_bins=256):
    """utils.py:62-74 (any device).  Inside the fused training step the engine does this in the resampling kernel."""
    B = x.size(0)
    x = x.reshape(-1).clone()
    uv = x <= 0
    x[uv] = 0
    assert (x >= 0).all() and (x <= 1).all()
    idx = torch.round(x * (num_bins - 1)) + 1
    idx[uv] = 0
    enc = torch.zeros((x.size(0), num_bins + 1), device=x.device)
    enc[torch.arange(x.size(0), device=x.device), idx.long()] = 1
    return enc.view(B, -1, num_bins + 1), idx.view(B, -1).long()


def get_mask_from_lengths(lengths, max_len):
    """utils.py:78-81 (imported by the reference's model.py but never called)."""
    ids = torch.arange(0, max_len, device=lengths.device)
    return (ids >= lengths.unsqueeze(1)).bool()


def pad_seq_to_2(x, len_out=128):
    """utils.py:85-87."""
    len_pad = len_out - x.shape[1]
    assert len_pad >= 0
    return np.pad(x, ((0, 0), (0, len_pad), (0, 0)), 'constant'), len_pad
