// Offline feature extraction, the part of reference make_spect_f0.py / utils.py that can be pinned in this environment
// (SURVEY.md section 8(f) row N4): STFT magnitude (utils.py:18-31 pySTFT: reflect padding by fft/2, periodic Hann window,
// 1024-point rfft, hop 256) -> mel projection with a CALLER-SUPPLIED basis (make_spect_f0.py:15,58: librosa's filter bank is
// not available here, so it is an input) -> dB with the reference's floor and offsets (make_spect_f0.py:16,59-60) -> float32;
// and the per-utterance F0 normalisation (utils.py:35-42 + make_spect_f0.py:64-66).  The Butterworth filtfilt, the dither and
// RAPT stay on the host (speechsplit_amd/features.py): a 5th-order IIR run forwards and backwards is a sequential f64
// recurrence of ~1e5 samples, and RAPT is a third-party C algorithm (pysptk) that is absent.
//
// Arithmetic: float64 throughout, as numpy computes it (the reference casts to float32 only when saving).  One utterance is
// ~200 frames x 513 bins x 1024 taps = 0.2 G multiply-adds: a direct DFT from an exact twiddle table in LDS, no FFT needed.
#include "common.h"
#include "kernels.h"

namespace ss {

namespace {

constexpr int NFFT = 1024, HOP = 256, NBIN = NFFT / 2 + 1;

// reflect-padded sample (numpy.pad mode='reflect': the edge sample is not repeated)
__device__ __forceinline__ double padded(const double* __restrict__ x, int n, int i) {
    int j = i - NFFT / 2;
    if (j < 0) j = -j;
    if (j >= n) j = 2 * (n - 1) - j;
    return x[j];
}

// grid = (frames), block = 256: frame -> windowed samples in LDS -> |rfft| -> mag in LDS -> mel -> dB -> S
__global__ __launch_bounds__(256) void melspec_kernel(const double* __restrict__ x, int n, const double* __restrict__ mel, int n_mels,
                                                      float* __restrict__ out) {
    __shared__ double fr[NFFT];
    __shared__ double cs[NFFT], sn[NFFT];
    __shared__ double mag[NBIN];
    const int f = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < NFFT; i += 256) {
        const double a = 2.0 * M_PI * (double)i / (double)NFFT;
        const double w = 0.5 - 0.5 * cos(a);                       // scipy.signal.get_window('hann', N, fftbins=True)
        fr[i] = w * padded(x, n, f * HOP + i);
        cs[i] = cos(a);
        sn[i] = sin(a);
    }
    __syncthreads();
    for (int k = tid; k < NBIN; k += 256) {
        double re = 0.0, im = 0.0;
        int idx = 0;                                                // (k * t) mod NFFT, updated incrementally
        for (int t = 0; t < NFFT; ++t) {
            re += fr[t] * cs[idx];
            im -= fr[t] * sn[idx];
            idx = (idx + k) & (NFFT - 1);
        }
        mag[k] = sqrt(re * re + im * im);
    }
    __syncthreads();
    const double min_level = exp(-100.0 / 20.0 * log(10.0));       // make_spect_f0.py:16
    for (int m = tid; m < n_mels; m += 256) {
        double s = 0.0;
        for (int k = 0; k < NBIN; ++k) s += mag[k] * mel[(long)k * n_mels + m];
        const double db = 20.0 * log10(fmax(min_level, s)) - 16.0;   // :59
        out[(long)f * n_mels + m] = (float)((db + 100.0) / 100.0);    // :60, :71 (astype(float32))
    }
}

// one block: mean / population std over the voiced frames (f0 != -1e10), then utils.py:35-42
__global__ __launch_bounds__(256) void f0_norm_kernel(const double* __restrict__ f0, int n, float* __restrict__ out) {
    __shared__ double red[256];
    __shared__ double stat[2];
    __shared__ int cnt[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    int c = 0;
    for (int i = tid; i < n; i += 256)
        if (f0[i] != -1e10) {
            s += f0[i];
            ++c;
        }
    red[tid] = s;
    cnt[tid] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) {
            red[tid] += red[tid + w];
            cnt[tid] += cnt[tid + w];
        }
        __syncthreads();
    }
    const int nv = cnt[0];
    if (tid == 0) stat[0] = nv ? red[0] / nv : 0.0;
    __syncthreads();
    const double mean = stat[0];
    double q = 0.0;
    for (int i = tid; i < n; i += 256)
        if (f0[i] != -1e10) q += (f0[i] - mean) * (f0[i] - mean);
    __syncthreads();
    red[tid] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) stat[1] = nv ? sqrt(red[0] / nv) : 1.0;          // np.std: population
    __syncthreads();
    const double sd = stat[1];
    for (int i = tid; i < n; i += 256) {
        double v = f0[i];
        if (v != -1e10) {
            v = (v - mean) / sd / 4.0;
            v = fmin(fmax(v, -1.0), 1.0);
            v = (v + 1.0) / 2.0;
        }
        out[i] = (float)v;
    }
}

}  // namespace

hipError_t melspec(const double* x, int n, const double* mel, int n_mels, float* out, int frames, hipStream_t s) {
    if (n < NFFT / 2 + 1 || n_mels < 1 || frames < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(melspec_kernel, dim3(frames), dim3(256), 0, s, x, n, mel, n_mels, out);
    return hipGetLastError();
}

hipError_t f0_normalize(const double* f0, int n, float* out, hipStream_t s) {
    if (n < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(f0_norm_kernel, dim3(1), dim3(256), 0, s, f0, n, out);
    return hipGetLastError();
}

}  // namespace ss
