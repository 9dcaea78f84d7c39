"""Offline feature extraction (reference make_spect_f0.py:48-73 with utils.py:10-42) -- SURVEY.md section 8(f) row N4.

What runs where.  Host (scipy / numpy, float64, as the reference): the 5th-order 30 Hz Butterworth high-pass applied forwards
and backwards (`signal.filtfilt`, make_spect_f0.py:53) and the 1e-6 dither from the per-speaker generator (:54) -- a sequential
recurrence over the waveform.  GPU (csrc/features.hip through the C ABI): STFT magnitude -> mel projection -> dB -> [0, 1]
scaling, and the F0 normalisation.  NOT built: the mel filter bank itself (`librosa.filters.mel`, make_spect_f0.py:15) and RAPT
(`pysptk.sptk.rapt`, :63) -- both libraries are absent here, so the basis and the raw F0 track are inputs and that part of N4
stays unpinned.  The spectrogram half is pinned by tests/golden/features.npz, generated from the reference's own
`butter_highpass` / `pySTFT` / `speaker_normalization`."""
import ctypes as C

import numpy as np
import torch
from scipy import signal

from . import _capi


def butter_highpass(cutoff, fs, order=5):
    """utils.py:10-14"""
    nyq = 0.5 * fs
    b, a = signal.butter(order, cutoff / nyq, btype='high', analog=False)
    return b, a


def preprocess_wav(x, prng, fs=16000):
    """make_spect_f0.py:49-54: the odd-length fix, high-pass filtfilt, 0.96 scaling and dither.  x float64 [n]."""
    assert fs == 16000
    if x.shape[0] % 256 == 0:
        x = np.concatenate((x, np.array([1e-06])), axis=0)
    b, a = butter_highpass(30, 16000, order=5)
    y = signal.filtfilt(b, a, x)
    return y * 0.96 + (prng.rand(y.shape[0]) - 0.5) * 1e-06


def melspectrogram(wav, mel_basis, device='cuda'):
    """make_spect_f0.py:57-60 on the GPU.  wav float64 [n] (after preprocess_wav), mel_basis float64 [513, n_mels] ->
    float32 [frames, n_mels] tensor on `device`."""
    lib = _capi.lib()
    w = torch.as_tensor(np.ascontiguousarray(wav, dtype=np.float64)).to(device)
    mb = torch.as_tensor(np.ascontiguousarray(mel_basis, dtype=np.float64)).to(device)
    if mb.shape[0] != 513:
        raise ValueError('mel_basis must be [513, n_mels] (1024-point transform)')
    frames = lib.ss_melspec_frames(w.numel())
    out = torch.empty(frames, mb.shape[1], device=device)
    _capi.check(lib.ss_melspec(C.c_void_p(w.data_ptr()), w.numel(), C.c_void_p(mb.data_ptr()), mb.shape[1], C.c_void_p(out.data_ptr()),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def normalize_f0(f0_rapt, device='cuda'):
    """make_spect_f0.py:64-66 + utils.py:35-42.  f0_rapt float [n] with -1e10 for unvoiced frames -> float32 [n] tensor."""
    lib = _capi.lib()
    f = torch.as_tensor(np.ascontiguousarray(f0_rapt, dtype=np.float64)).to(device)
    out = torch.empty(f.numel(), device=device)
    _capi.check(lib.ss_f0_normalize(C.c_void_p(f.data_ptr()), f.numel(), C.c_void_p(out.data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out
