// Bidirectional LSTM recurrences with a tiny hidden size (the encoder bottlenecks: hidden 1, 8 and 32;
// reference model.py:71, 119, 174, 189).  W_hh is at most 128 x 32 floats, so one workgroup per (utterance,
// direction) keeps its gate row (forward) / gate column (backward) of W_hh in registers and walks all
// T steps in a single launch: no per-step launch, no inter-workgroup traffic.  These recurrences are pure latency;
// the next step's global operands are fetched while the current step computes.
//
// Semantics (torch.nn.LSTM, which the reference calls): gates i,f,g,o; c' = f*c + i*g; h' = o*tanh(c'); zero
// initial state; the reverse direction walks t = T-1..0.  The input projection x.W_ih^T + b_ih + b_hh arrives
// precomputed in `gates` (one GEMM for all T), which this kernel overwrites with the activated gates, and the
// backward kernel overwrites again with the pre-activation gradients (consumed by the weight-gradient GEMMs).
#include "common.h"
#include "kernels.h"

namespace ss {

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return ss_sigmoid(x); }

template <int H>
__global__ __launch_bounds__((4 * H > 64 ? 4 * H : 64)) void lstm_small_fwd_kernel(float* __restrict__ gates,
                                                                                   const float* __restrict__ whh_f,
                                                                                   const float* __restrict__ whh_b,
                                                                                   float* __restrict__ out,
                                                                                   float* __restrict__ csave, int T) {
    __shared__ float hs[H];
    __shared__ float gs[4 * H];
    const int b = blockIdx.x, dir = blockIdx.y, n = threadIdx.x;
    const int TP = T + 2 * HALO;
    const float* whh = dir ? whh_b : whh_f;
    const bool gate_thread = n < 4 * H;
    float w[H];
#pragma unroll
    for (int k = 0; k < H; ++k) w[k] = gate_thread ? whh[n * H + k] : 0.f;
    if (n < H) hs[n] = 0.f;
    float c = 0.f;
    float* grow = gates + (long)b * TP * (8 * H) + dir * 4 * H + n;
    auto tau_of = [&](int s) { return HALO + (dir == 0 ? s : T - 1 - s); };
    float x_next = gate_thread ? grow[(long)tau_of(0) * (8 * H)] : 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int tau = tau_of(s);
        const float xn = x_next;
        if (s + 1 < T && gate_thread) x_next = grow[(long)tau_of(s + 1) * (8 * H)];
        if (gate_thread) {
            float acc = xn;
#pragma unroll
            for (int k = 0; k < H; ++k) acc += w[k] * hs[k];
            const float act = (n / H == 2) ? ss_tanh(acc) : sigmoidf_(acc);
            gs[n] = act;
            grow[(long)tau * (8 * H)] = act;
        }
        __syncthreads();
        if (n < H) {
            c = gs[H + n] * c + gs[n] * gs[2 * H + n];
            const float h = gs[3 * H + n] * ss_tanh(c);
            hs[n] = h;
            const long o = ((long)b * TP + tau) * (2 * H) + dir * H + n;
            out[o] = h;
            csave[o] = c;
        }
        __syncthreads();
    }
}

template <int H>
__global__ __launch_bounds__((4 * H > 64 ? 4 * H : 64)) void lstm_small_bwd_kernel(float* __restrict__ gates,
                                                                                   const float* __restrict__ whh_f,
                                                                                   const float* __restrict__ whh_b,
                                                                                   const float* __restrict__ d_out,
                                                                                   const float* __restrict__ csave, int T) {
    __shared__ float dg[4 * H];
    __shared__ float part[4 * H];
    const int b = blockIdx.x, dir = blockIdx.y, n = threadIdx.x;
    const int TP = T + 2 * HALO;
    const float* whh = dir ? whh_b : whh_f;
    // thread n = (gate p, hidden k) owns column k of gate p's H x H block of W_hh: H registers, no LDS traffic for the weights
    float wcol[H];
#pragma unroll
    for (int q = 0; q < H; ++q) wcol[q] = n < 4 * H ? whh[((n / H) * H + q) * H + n % H] : 0.f;
    const bool cell_thread = n < H;
    float* grow = gates + (long)b * TP * (8 * H) + dir * 4 * H + n;
    const long obase = (long)b * TP * (2 * H) + dir * H + n;
    auto tau_of = [&](int s) { return HALO + (dir == 0 ? T - 1 - s : s); };
    float dh_rec = 0.f, dc_rec = 0.f;
    // operands of the step being processed, fetched one step ahead
    float p_do = 0.f, p_i = 0.f, p_f = 0.f, p_g = 0.f, p_o = 0.f, p_c = 0.f, p_cp = 0.f;
    auto fetch = [&](int s, float c_known, bool have_c) {
        const int tau = tau_of(s);
        const int tau_prev = dir == 0 ? tau - 1 : tau + 1;       // previous step in FORWARD order (halo row = 0)
        p_do = d_out[obase + (long)tau * (2 * H)];
        p_i = grow[(long)tau * (8 * H)];
        p_f = grow[(long)tau * (8 * H) + H];
        p_g = grow[(long)tau * (8 * H) + 2 * H];
        p_o = grow[(long)tau * (8 * H) + 3 * H];
        p_c = have_c ? c_known : csave[obase + (long)tau * (2 * H)];
        p_cp = csave[obase + (long)tau_prev * (2 * H)];
    };
    if (cell_thread) fetch(0, 0.f, false);
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int tau = tau_of(s);
        if (cell_thread) {
            const float dh = p_do + dh_rec;
            const float gi = p_i, gf = p_f, gg = p_g, go = p_o, cc = p_c, cp = p_cp;
            if (s + 1 < T) fetch(s + 1, cp, true);               // c of the next processed step == this step's c_prev
            const float tc = ss_tanh(cc);
            const float d_o = dh * tc;
            const float dc = dc_rec + dh * go * (1.0f - tc * tc);
            dc_rec = dc * gf;
            const float dai = dc * gg * gi * (1.0f - gi);
            const float daf = dc * cp * gf * (1.0f - gf);
            const float dag = dc * gi * (1.0f - gg * gg);
            const float dao = d_o * go * (1.0f - go);
            dg[n] = dai;
            dg[H + n] = daf;
            dg[2 * H + n] = dag;
            dg[3 * H + n] = dao;
            float* gr = grow + (long)tau * (8 * H);
            gr[0] = dai;
            gr[H] = daf;
            gr[2 * H] = dag;
            gr[3 * H] = dao;
        }
        __syncthreads();
        if (n < 4 * H) {
            const int p = n / H;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < H; ++q) acc += dg[p * H + q] * wcol[q];
            part[n] = acc;
        }
        __syncthreads();
        if (cell_thread) dh_rec = (part[n] + part[H + n]) + (part[2 * H + n] + part[3 * H + n]);
    }
}

template <int H>
hipError_t fwd_t(float* gates, const float* wf, const float* wb, float* out, float* csave, int B, int T, hipStream_t s) {
    constexpr int NT = 4 * H > 64 ? 4 * H : 64;
    hipLaunchKernelGGL((lstm_small_fwd_kernel<H>), dim3(B, 2), dim3(NT), 0, s, gates, wf, wb, out, csave, T);
    return hipGetLastError();
}
template <int H>
hipError_t bwd_t(float* gates, const float* wf, const float* wb, const float* d_out, const float* csave, int B, int T,
                 hipStream_t s) {
    constexpr int NT = 4 * H > 64 ? 4 * H : 64;
    hipLaunchKernelGGL((lstm_small_bwd_kernel<H>), dim3(B, 2), dim3(NT), 0, s, gates, wf, wb, d_out, csave, T);
    return hipGetLastError();
}

}  // namespace

hipError_t lstm_small_fwd(float* gates, const float* whh_f, const float* whh_b, float* out, float* csave, int B, int T,
                          int H, hipStream_t s) {
    switch (H) {
        case 1: return fwd_t<1>(gates, whh_f, whh_b, out, csave, B, T, s);
        case 2: return fwd_t<2>(gates, whh_f, whh_b, out, csave, B, T, s);
        case 4: return fwd_t<4>(gates, whh_f, whh_b, out, csave, B, T, s);
        case 8: return fwd_t<8>(gates, whh_f, whh_b, out, csave, B, T, s);
        case 16: return fwd_t<16>(gates, whh_f, whh_b, out, csave, B, T, s);
        case 32: return fwd_t<32>(gates, whh_f, whh_b, out, csave, B, T, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t lstm_small_bwd(float* gates, const float* whh_f, const float* whh_b, const float* d_out, const float* csave,
                          int B, int T, int H, hipStream_t s) {
    switch (H) {
        case 1: return bwd_t<1>(gates, whh_f, whh_b, d_out, csave, B, T, s);
        case 2: return bwd_t<2>(gates, whh_f, whh_b, d_out, csave, B, T, s);
        case 4: return bwd_t<4>(gates, whh_f, whh_b, d_out, csave, B, T, s);
        case 8: return bwd_t<8>(gates, whh_f, whh_b, d_out, csave, B, T, s);
        case 16: return bwd_t<16>(gates, whh_f, whh_b, d_out, csave, B, T, s);
        case 32: return bwd_t<32>(gates, whh_f, whh_b, d_out, csave, B, T, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ss
