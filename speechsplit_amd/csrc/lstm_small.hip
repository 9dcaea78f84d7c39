// Bidirectional LSTM recurrences with a tiny hidden size (the encoder bottlenecks: hidden 1, 8 and 32;
// reference model.py:71, 119, 174, 189).  W_hh is at most 128 x 32 floats, so one workgroup per (utterance,
// direction) keeps its gate row (forward) / gate column (backward) of W_hh in registers and walks all
// T steps in a single launch: no per-step launch, no inter-workgroup traffic.  These recurrences are pure latency:
// a step is a few hundred cycles of arithmetic, far less than one trip to memory, so the *_lds kernels first pull the
// utterance's whole operand sequence into LDS (bulk, coalesced, many loads in flight) and then walk the steps out of
// LDS, with results leaving as fire-and-forget stores (0.5-1.1 us per step when every step waited for its own loads,
// ~0.2 us from LDS).  Sequences too long for 160 KB of LDS take the streaming kernels, which fetch one step ahead.
//
// Semantics (torch.nn.LSTM, which the reference calls): gates i,f,g,o; c' = f*c + i*g; h' = o*tanh(c'); zero
// initial state; the reverse direction walks t = T-1..0.  The input projection x.W_ih^T + b_ih + b_hh arrives
// precomputed in `gates` (one GEMM for all T), which this kernel overwrites with the activated gates, and the
// backward kernel overwrites again with the pre-activation gradients (consumed by the weight-gradient GEMMs).
#include "common.h"
#include "kernels.h"

namespace ss {

int g_small_prio = 1;    // the small recurrences run at s_setprio 3
int g_small_lds = 1;     // 1: LDS-staged kernels (single-wave variant where it applies), 2: LDS-staged without the single-wave variant,
                        // 0: always the streaming kernels (A/B experiments)

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return ss_sigmoid(x); }

template <int H>
__global__ __launch_bounds__((4 * H > 64 ? 4 * H : 64)) void lstm_small_fwd_kernel(float* __restrict__ gates,
                                                                                   const float* __restrict__ whh_f,
                                                                                   const float* __restrict__ whh_b,
                                                                                   float* __restrict__ out,
                                                                                   float* __restrict__ csave, int T, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);       // latency chains: issue ahead of co-resident GEMM waves (ss_tune("small_prio"))
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
            const float act = ss_gate(acc, (n / H == 2) ? 2.0f : 1.0f);
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
                                                                                   const float* __restrict__ csave, int T, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);       // latency chains: issue ahead of co-resident GEMM waves (ss_tune("small_prio"))
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

// Bulk copy of `rows` rows of Q float4 each (row r of the destination comes from source row tau(r), rows 2Q float4 apart)
// into LDS: U loads per thread in flight before the first LDS write (written as a plain loop, hipcc waits for every load
// before issuing the next one).
template <int NT, int Q, int U, typename TauOf>
__device__ __forceinline__ void stage_rows(const float4* __restrict__ src, float4* dst, int rows, int n, TauOf tau_of) {
    const int total = rows * Q;
    for (int base = n; base < total; base += NT * U) {
        float4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * NT;
            if (i < total) r[u] = src[(long)tau_of(i / Q) * (2 * Q) + i % Q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * NT;
            if (i < total) dst[i] = r[u];
        }
    }
}

// ---- LDS-staged variants ------------------------------------------------------------------------------------------
// dynamic LDS: the utterance's pre-activations in STEP order, xs[s][4H]
template <int H>
__global__ __launch_bounds__((4 * H > 64 ? 4 * H : 64)) void lstm_small_fwd_lds_kernel(float* __restrict__ gates,
                                                                                       const float* __restrict__ whh_f,
                                                                                       const float* __restrict__ whh_b,
                                                                                       float* __restrict__ out,
                                                                                       float* __restrict__ csave, int T, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);       // latency chains: issue ahead of co-resident GEMM waves (ss_tune("small_prio"))
    constexpr int NT = 4 * H > 64 ? 4 * H : 64;
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* xs = dyn;
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
    float* g0 = gates + (long)b * TP * (8 * H) + dir * 4 * H;
    auto tau_of = [&](int s) { return HALO + (dir == 0 ? s : T - 1 - s); };
    // 4H floats per row = H float4; row stride 2H float4
    stage_rows<NT, H, 16>(reinterpret_cast<const float4*>(g0), reinterpret_cast<float4*>(xs), T, n, tau_of);
    float* grow = g0 + n;
    lds_barrier();
    for (int s = 0; s < T; ++s) {
        const int tau = tau_of(s);
        if (gate_thread) {
            float acc = xs[s * 4 * H + n];
#pragma unroll
            for (int k = 0; k < H; ++k) acc += w[k] * hs[k];
            const float act = ss_gate(acc, (n / H == 2) ? 2.0f : 1.0f);
            gs[n] = act;
            grow[(long)tau * (8 * H)] = act;
        }
        lds_barrier();
        if (n < H) {
            c = gs[H + n] * c + gs[n] * gs[2 * H + n];
            const float h = gs[3 * H + n] * ss_tanh(c);
            hs[n] = h;
            const long o = ((long)b * TP + tau) * (2 * H) + dir * H + n;
            out[o] = h;
            csave[o] = c;
        }
        lds_barrier();
    }
}

// Single-wave variant for 4H <= 64 (H = 1 .. 16): all gate threads sit in one wave, so a step needs no LDS exchange and no
// barrier at all -- h(t-1) is read lane by lane with v_readlane (a scalar operand for the H multiply-adds), the four gates of
// a unit meet in its cell lane through three ds_bpermute, and the next step's pre-activation is already in a register.
// One dependent LDS-crossbar round trip per step instead of four.  Same arithmetic order as the kernels above.
template <int H>
__global__ __launch_bounds__(64) void lstm_small_fwd_wave_kernel(float* __restrict__ gates, const float* __restrict__ whh_f,
                                                                 const float* __restrict__ whh_b, float* __restrict__ out,
                                                                 float* __restrict__ csave, int T, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);       // latency chains: issue ahead of co-resident GEMM waves (ss_tune("small_prio"))
    static_assert(4 * H <= 64, "one wave");
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* xs = dyn;
    const int b = blockIdx.x, dir = blockIdx.y, n = threadIdx.x;
    const int TP = T + 2 * HALO;
    const float* whh = dir ? whh_b : whh_f;
    const bool gate_thread = n < 4 * H;
    float w[H];
#pragma unroll
    for (int k = 0; k < H; ++k) w[k] = gate_thread ? whh[n * H + k] : 0.f;
    float* g0 = gates + (long)b * TP * (8 * H) + dir * 4 * H;
    auto tau_of = [&](int s) { return HALO + (dir == 0 ? s : T - 1 - s); };
    stage_rows<64, H, 16>(reinterpret_cast<const float4*>(g0), reinterpret_cast<float4*>(xs), T, n, tau_of);
    lds_barrier();
    float* grow = g0 + n;
    const int nn = gate_thread ? n : 0;                      // idle lanes mirror lane 0 (never stored)
    float c = 0.f, h = 0.f;
    float xn = xs[nn];
    for (int s = 0; s < T; ++s) {
        const int tau = tau_of(s);
        float acc = xn;
        if (s + 1 < T) xn = xs[(s + 1) * 4 * H + nn];         // next step's pre-activation: no dependence on this step
#pragma unroll
        for (int k = 0; k < H; ++k) acc += w[k] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k));
        const float act = ss_gate(acc, (nn / H == 2) ? 2.0f : 1.0f);
        if (gate_thread) grow[(long)tau * (8 * H)] = act;
        // unit u's gates sit in lanes u, u+H, u+2H, u+3H; lanes >= H compute along (their c / h are never used)
        const int u = n & (H - 1);
        const float gi = __shfl(act, u), gf = __shfl(act, u + H), gg = __shfl(act, u + 2 * H), go = __shfl(act, u + 3 * H);
        c = gf * c + gi * gg;
        h = go * ss_tanh(c);
        if (n < H) {
            const long o = ((long)b * TP + tau) * (2 * H) + dir * H + n;
            out[o] = h;
            csave[o] = c;
        }
    }
}

// dynamic LDS, all in STEP order of the backward walk: ga[T][4H] activated gates, dd[T][H] d_out, cc[T + 1][H] cell states
// (cc[s + 1] is step s's previous cell state in forward time; the last one is a halo row = 0)
template <int H>
__global__ __launch_bounds__((4 * H > 64 ? 4 * H : 64)) void lstm_small_bwd_lds_kernel(float* __restrict__ gates,
                                                                                       const float* __restrict__ whh_f,
                                                                                       const float* __restrict__ whh_b,
                                                                                       const float* __restrict__ d_out,
                                                                                       const float* __restrict__ csave, int T, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);       // latency chains: issue ahead of co-resident GEMM waves (ss_tune("small_prio"))
    constexpr int NT = 4 * H > 64 ? 4 * H : 64;
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* ga = dyn;
    float* dd = ga + (long)T * 4 * H;
    float* cc = dd + (long)T * H;
    __shared__ float dg[4 * H];
    __shared__ float part[4 * H];
    const int b = blockIdx.x, dir = blockIdx.y, n = threadIdx.x;
    const int TP = T + 2 * HALO;
    const float* whh = dir ? whh_b : whh_f;
    float wcol[H];
#pragma unroll
    for (int q = 0; q < H; ++q) wcol[q] = n < 4 * H ? whh[((n / H) * H + q) * H + n % H] : 0.f;
    const bool cell_thread = n < H;
    float* g0 = gates + (long)b * TP * (8 * H) + dir * 4 * H;
    const long obase = (long)b * TP * (2 * H) + dir * H;
    auto tau_of = [&](int s) { return HALO + (dir == 0 ? T - 1 - s : s); };      // s == T: the halo row next to the walk's end
    stage_rows<NT, H, 16>(reinterpret_cast<const float4*>(g0), reinterpret_cast<float4*>(ga), T, n, tau_of);
    if constexpr (H % 4 == 0) {                        // rows of H floats as H/4 float4 (row stride 2H floats = H/2 float4)
        stage_rows<NT, H / 4, 8>(reinterpret_cast<const float4*>(d_out + obase), reinterpret_cast<float4*>(dd), T, n, tau_of);
        stage_rows<NT, H / 4, 8>(reinterpret_cast<const float4*>(csave + obase), reinterpret_cast<float4*>(cc), T + 1, n, tau_of);
    } else {
        for (int base = n; base < (T + 1) * H; base += NT * 8) {
            float rd[8], rc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * NT;
                if (i < (T + 1) * H) rc[u] = csave[obase + (long)tau_of(i / H) * (2 * H) + i % H];
                if (i < T * H) rd[u] = d_out[obase + (long)tau_of(i / H) * (2 * H) + i % H];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * NT;
                if (i < (T + 1) * H) cc[i] = rc[u];
                if (i < T * H) dd[i] = rd[u];
            }
        }
    }
    float* grow = g0 + n;
    float dh_rec = 0.f, dc_rec = 0.f;
    lds_barrier();
    for (int s = 0; s < T; ++s) {
        const int tau = tau_of(s);
        if (cell_thread) {
            const float dh = dd[s * H + n] + dh_rec;
            const float* gr_s = ga + s * 4 * H + n;
            const float gi = gr_s[0], gf = gr_s[H], gg = gr_s[2 * H], go = gr_s[3 * H], cv = cc[s * H + n], cp = cc[(s + 1) * H + n];
            const float tc = ss_tanh(cv);
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
        lds_barrier();
        if (n < 4 * H) {
            const int p = n / H;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < H; ++q) acc += dg[p * H + q] * wcol[q];
            part[n] = acc;
        }
        lds_barrier();
        if (cell_thread) dh_rec = (part[n] + part[H + n]) + (part[2 * H + n] + part[3 * H + n]);
    }
}

constexpr long LDS_BUDGET = 160 * 1024 - 2048;      // dynamic part; the static arrays above are < 2 KB

template <typename K>
hipError_t allow_lds(K kernel, long bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int H>
hipError_t fwd_t(float* gates, const float* wf, const float* wb, float* out, float* csave, int B, int T, hipStream_t s) {
    constexpr int NT = 4 * H > 64 ? 4 * H : 64;
    const long bytes = (long)T * 4 * H * 4;
    if constexpr (4 * H <= 64) {
        if (g_small_lds == 1 && bytes <= LDS_BUDGET) {
            hipError_t e = allow_lds(lstm_small_fwd_wave_kernel<H>, bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((lstm_small_fwd_wave_kernel<H>), dim3(B, 2), dim3(64), bytes, s, gates, wf, wb, out, csave, T, g_small_prio);
            return hipGetLastError();
        }
    }
    if (g_small_lds && bytes <= LDS_BUDGET) {
        hipError_t e = allow_lds(lstm_small_fwd_lds_kernel<H>, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((lstm_small_fwd_lds_kernel<H>), dim3(B, 2), dim3(NT), bytes, s, gates, wf, wb, out, csave, T, g_small_prio);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((lstm_small_fwd_kernel<H>), dim3(B, 2), dim3(NT), 0, s, gates, wf, wb, out, csave, T, g_small_prio);
    return hipGetLastError();
}
template <int H>
hipError_t bwd_t(float* gates, const float* wf, const float* wb, const float* d_out, const float* csave, int B, int T,
                 hipStream_t s) {
    constexpr int NT = 4 * H > 64 ? 4 * H : 64;
    const long bytes = ((long)T * 6 * H + H) * 4;
    if (g_small_lds && bytes <= LDS_BUDGET) {
        hipError_t e = allow_lds(lstm_small_bwd_lds_kernel<H>, bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((lstm_small_bwd_lds_kernel<H>), dim3(B, 2), dim3(NT), bytes, s, gates, wf, wb, d_out, csave, T, g_small_prio);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((lstm_small_bwd_kernel<H>), dim3(B, 2), dim3(NT), 0, s, gates, wf, wb, d_out, csave, T, g_small_prio);
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
