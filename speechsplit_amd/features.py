"""Offline feature extraction (reference make_spect_f0.py:48-73 with utils.py:10-42) -- SURVEY.md section 8(f) row N4.

What runs where.  Host (scipy / numpy, float64, as the reference): the 5th-order 30 Hz Butterworth high-pass applied forwards
and backwards (`signal.filtfilt`, make_spect_f0.py:53) and the 1e-6 dither from the per-speaker generator (:54) -- a sequential
recurrence over the waveform.  GPU (csrc/features.hip through the C ABI): STFT magnitude -> mel projection -> dB -> [0, 1]
scaling, and the F0 normalisation.  The mel filter bank (`librosa.filters.mel`, make_spect_f0.py:15) is restated here from the published
algorithm (`mel_filter_bank`: Slaney's Auditory-Toolbox mel scale and area normalisation, librosa's defaults) -- parity against librosa itself
is UNPINNED (it is absent in this environment, so nothing of the reference's could be run to produce a vector); the function is pinned against
hand-computed closed-form values of the published definition for three filters (linear region, across 1 kHz, top band:
tests/test_capi_host.py::test_mel_filter_bank_closed_form_slaney_values) and against its published properties.  NOT built: RAPT (`pysptk.sptk.rapt`, :63 -- a third-party C algorithm with no source here): the raw F0 track is an input.
The spectrogram half is pinned by tests/golden/features.npz, generated from the reference's own `butter_highpass` / `pySTFT` /
`speaker_normalization`."""
import ctypes as C

import numpy as np
import torch
from scipy import signal

from . import _capi


def _hz_to_mel(f):
    """Slaney's mel scale (Auditory Toolbox; librosa's default, htk=False): linear below 1 kHz (200/3 Hz per mel), logarithmic above
    (27 mels per factor of 6.4)."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mel)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filter_bank(sr=16000, n_fft=1024, n_mels=80, fmin=90.0, fmax=7600.0):
    """The basis make_spect_f0.py:15 takes from `librosa.filters.mel(16000, 1024, fmin=90, fmax=7600, n_mels=80)` (and transposes),
    restated from the published algorithm with librosa's defaults: n_mels + 2 band edges equally spaced on Slaney's mel scale, one
    triangle per band over the 1 + n_fft/2 FFT bin frequencies, each scaled by 2 / (its band's width in Hz) ('slaney' norm: unit area).
    Returns float32 [n_mels, 1 + n_fft // 2]; `melspectrogram` wants its transpose.  Pinned against the closed form, not against librosa (see the module docstring)."""
    fft_f = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


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
