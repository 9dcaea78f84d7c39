// Decoder BLSTM recurrence (hidden 512 for Decoder_3, 256 for Decoder_4; reference model.py:244-245, 268-269):
// 79 % of the model's MACs and all of its serial depth.  One launch per time step; both directions run in the
// same launch.  W_hh (4 MiB fp32 per layer-direction) does not fit one CU's LDS, so each workgroup owns a
// 16-utterance x 16-hidden-unit tile, streams its slice of W_hh (L2-resident across steps) and of h(t-1) straight
// into MFMA operand registers, and applies the cell update to the 256 (utterance, unit) pairs it owns.
//
//   forward  step: a[b, g*H+j] = xproj[b,t,g*H+j] + sum_k h(t-1)[b,k] * W_hh[g*H+j, k]          (K = H)
//   backward step: dh(t)[b,j]  = d_out[b,t,j]     + sum_n da(t+1)[b,n] * W_hh[n, j]              (K = 4H)
//
// Both contractions are "row . row" products of K-contiguous operands, computed with v_mfma_f32_16x16x4_f32.
// Operand trick: lane l of the MFMA supplies A[i = l&15][k = l>>4]; instead of loading one float per MFMA each
// lane loads a float4 at k-offset 4*(l>>4) of a 16-wide chunk and feeds element q to MFMA q.  Across the four
// MFMAs of a chunk every k is used exactly once (in a permuted order, identical for both operands), so loads are
// 16-byte, row-contiguous, and need no LDS.  The NW waves of a workgroup split K and reduce through LDS.
//
// A step is bound by how many bytes a CU keeps in flight from L2 (160 KB fwd / 256 KB bwd per workgroup per step),
// not by the 1.7 us of MFMA work: the loads are issued in groups of G chunks, one group ahead of the MFMAs that
// consume them (NW and G are tuning knobs, ss_tune("lstm_nw" / "lstm_g")).
//
// h(t-1) and c(t-1) are read from the haloed output / cell slabs themselves (row t-1, or the all-zero halo row at
// the first step), so there is no separate state buffer and no branch for the initial state.
#include "common.h"
#include "kernels.h"

namespace ss {

int g_lstm_nw = 8;    // waves per workgroup in the step kernels (4, 8 or 16)
int g_lstm_g = 0;     // chunks per load group (0 = default for the chosen NW)
int g_lstm_mode = 0;  // diagnostics only: 1 = skip MFMAs, 2 = skip operand loads, 3 = empty kernel, 4 = skip epilogue

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// grid = (H/16, ceil(B/16), 2), block = 64*NW
template <int H, int NW, int G>
__global__ __launch_bounds__(64 * NW) void lstm_step_fwd_kernel(float* __restrict__ gates, const float* __restrict__ whh_f,
                                                                const float* __restrict__ whh_b, float* __restrict__ out,
                                                                float* __restrict__ csave, int B, int T, int step, int mode) {
    __shared__ float red[NW][4][16][16];
    if (mode == 3) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dir = blockIdx.z, j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int TP = T + 2 * HALO;
    const int tau = HALO + (dir == 0 ? step : T - 1 - step);
    const int tau_prev = dir == 0 ? tau - 1 : tau + 1;
    const float* whh = dir ? whh_b : whh_f;
    const int li = lane & 15, lk = (lane >> 4) * 4;
    constexpr int kw = H / NW;                              // K range of this wave
    constexpr int nchunk = kw / 16;
    static_assert(kw % 16 == 0 && nchunk % G == 0, "bad NW / G for this H");
    int bA = b0 + li;
    if (bA > B - 1) bA = B - 1;
    const float* Ap = out + ((long)bA * TP + tau_prev) * (2 * H) + dir * H + w * kw + lk;
    const float* Bp[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) Bp[g] = whh + (long)(g * H + j0 + li) * H + w * kw + lk;

    // operands of the cell update (first 256 threads), requested before the contraction
    const int bi = (tid >> 4) & 15, jj = tid & 15;
    const int b = b0 + bi, j = j0 + jj;
    const int bc = b < B ? b : B - 1;
    float* grow = gates + ((long)bc * TP + tau) * (8 * H) + dir * 4 * H + j;
    float xg[4] = {0.f, 0.f, 0.f, 0.f};
    float cp = 0.f;
    if (tid < 256) {
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = grow[g * H];
        cp = csave[((long)bc * TP + tau_prev) * (2 * H) + dir * H + j];
    }

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][G], bv[2][G][4];
    auto load_group = [&](int buf, int c0) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (mode == 2) {
                a[buf][i] = f32x4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[buf][i][g] = f32x4{1.f, 2.f, 3.f, 4.f};
            } else {
                a[buf][i] = ld4(Ap + (c0 + i) * 16);
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[buf][i][g] = ld4(Bp[g] + (c0 + i) * 16);
            }
        }
    };
    load_group(0, 0);
#pragma unroll
    for (int c0 = 0; c0 < nchunk; c0 += G) {
        const int cur = (c0 / G) & 1;
        if (c0 + G < nchunk) load_group(cur ^ 1, c0 + G);
        if (mode == 1) {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] += a[cur][i] * bv[cur][i][g];
        } else {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i][q], bv[cur][i][g][q], acc[g], 0, 0, 0);
        }
    }
    if (mode == 4) {
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 123.456f) out[0] = 1.f;
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][g][(lane >> 4) * 4 + r][li] = acc[g][r];
    __syncthreads();

    if (tid < 256 && b < B) {
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += red[ww][g][bi][jj];
            pre[g] = xg[g] + s;
        }
        const long o = ((long)b * TP + tau) * (2 * H) + dir * H + j;
        const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf(pre[2]), go = sigmoidf_(pre[3]);
        const float c = gf * cp + gi * gg;
        const float h = go * tanhf(c);
        grow[0] = gi;
        grow[H] = gf;
        grow[2 * H] = gg;
        grow[3 * H] = go;
        csave[o] = c;
        out[o] = h;
    }
}

// grid = (H/16, ceil(B/16), 2), block = 64*NW
template <int H, int NW, int G>
__global__ __launch_bounds__(64 * NW) void lstm_step_bwd_kernel(float* __restrict__ gates, const float* __restrict__ whhT,
                                                                const float* __restrict__ d_out,
                                                                const float* __restrict__ csave, float* __restrict__ dcs,
                                                                int B, int T, int step) {
    __shared__ float red[NW][16][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dir = blockIdx.z, j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int TP = T + 2 * HALO;
    const int tau = HALO + (dir == 0 ? T - 1 - step : step);
    const int tau_next = dir == 0 ? tau + 1 : tau - 1;       // processed by the previous backward step (halo at step 0)
    const int tau_prev = dir == 0 ? tau - 1 : tau + 1;       // previous in forward order (c(t-1))
    const int li = lane & 15, lk = (lane >> 4) * 4;
    constexpr int kw = 4 * H / NW;                           // K = 4H split over the waves
    constexpr int nchunk = kw / 16;
    static_assert(kw % 16 == 0 && nchunk % G == 0, "bad NW / G for this H");
    int bA = b0 + li;
    if (bA > B - 1) bA = B - 1;
    const float* Ap = gates + ((long)bA * TP + tau_next) * (8 * H) + dir * 4 * H + w * kw + lk;
    const float* Bp = whhT + ((long)dir * H + j0 + li) * (4 * H) + w * kw + lk;

    const int bi = (tid >> 4) & 15, jj = tid & 15;
    const int b = b0 + bi, j = j0 + jj;
    const int bc = b < B ? b : B - 1;
    const long o = ((long)bc * TP + tau) * (2 * H) + dir * H + j;
    float* grow = gates + ((long)bc * TP + tau) * (8 * H) + dir * 4 * H + j;
    float* dcp = dcs + ((long)dir * B + bc) * H + j;
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, p_do = 0.f, cc = 0.f, cp = 0.f, dc_rec = 0.f;
    if (tid < 256) {
        gi = grow[0];
        gf = grow[H];
        gg = grow[2 * H];
        go = grow[3 * H];
        p_do = d_out[o];
        cc = csave[o];
        cp = csave[((long)bc * TP + tau_prev) * (2 * H) + dir * H + j];
        dc_rec = step == 0 ? 0.f : *dcp;
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][G], v[2][G];
    auto load_group = [&](int buf, int c0) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            a[buf][i] = ld4(Ap + (c0 + i) * 16);
            v[buf][i] = ld4(Bp + (c0 + i) * 16);
        }
    };
    load_group(0, 0);
#pragma unroll
    for (int c0 = 0; c0 < nchunk; c0 += G) {
        const int cur = (c0 / G) & 1;
        if (c0 + G < nchunk) load_group(cur ^ 1, c0 + G);
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i][q], v[cur][i][q], acc1, 0, 0, 0);
                else       acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i][q], v[cur][i][q], acc0, 0, 0, 0);
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][(lane >> 4) * 4 + r][li] = acc0[r] + acc1[r];
    __syncthreads();

    if (tid < 256 && b < B) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += red[ww][bi][jj];
        const float dh = p_do + s;
        const float tc = tanhf(cc);
        const float d_o = dh * tc;
        const float dc = dc_rec + dh * go * (1.0f - tc * tc);
        *dcp = dc * gf;
        grow[0] = dc * gg * gi * (1.0f - gi);
        grow[H] = dc * cp * gf * (1.0f - gf);
        grow[2 * H] = dc * gi * (1.0f - gg * gg);
        grow[3 * H] = d_o * go * (1.0f - go);
    }
}

template <int H, int NW, int G>
hipError_t fwd_l(float* gates, const float* whh_f, const float* whh_b, float* out, float* csave, int B, int T, int step,
                 hipStream_t s) {
    hipLaunchKernelGGL((lstm_step_fwd_kernel<H, NW, G>), dim3(H / 16, cdiv(B, 16), 2), dim3(64 * NW), 0, s, gates, whh_f,
                       whh_b, out, csave, B, T, step, g_lstm_mode);
    return hipGetLastError();
}
template <int H, int NW, int G>
hipError_t bwd_l(float* gates, const float* whhT, const float* d_out, const float* csave, float* dc, int B, int T, int step,
                 hipStream_t s) {
    hipLaunchKernelGGL((lstm_step_bwd_kernel<H, NW, G>), dim3(H / 16, cdiv(B, 16), 2), dim3(64 * NW), 0, s, gates, whhT, d_out,
                       csave, dc, B, T, step);
    return hipGetLastError();
}

}  // namespace

#define FWD_ARGS gates, whh_f, whh_b, out, csave, B, T, step, s
#define BWD_ARGS gates, whhT, d_out, csave, dc, B, T, step, s

hipError_t lstm_step_fwd(float* gates, const float* whh_f, const float* whh_b, float* out, float* csave, int B, int T,
                         int H, int step, hipStream_t s) {
    const int nw = g_lstm_nw, g = g_lstm_g;
    switch (H) {
        case 64: return fwd_l<64, 4, 1>(FWD_ARGS);
        case 128: return fwd_l<128, 4, 2>(FWD_ARGS);
        case 256:
            if (nw >= 16) return fwd_l<256, 16, 1>(FWD_ARGS);
            if (nw >= 8) return fwd_l<256, 8, 2>(FWD_ARGS);
            return g == 1 ? fwd_l<256, 4, 1>(FWD_ARGS) : (g == 2 ? fwd_l<256, 4, 2>(FWD_ARGS) : fwd_l<256, 4, 4>(FWD_ARGS));
        case 512:
            if (nw >= 16) return g == 1 ? fwd_l<512, 16, 1>(FWD_ARGS) : fwd_l<512, 16, 2>(FWD_ARGS);
            if (nw >= 8) return g == 1 ? fwd_l<512, 8, 1>(FWD_ARGS) : (g == 2 ? fwd_l<512, 8, 2>(FWD_ARGS) : fwd_l<512, 8, 4>(FWD_ARGS));
            return g == 1 ? fwd_l<512, 4, 1>(FWD_ARGS)
                          : (g == 2 ? fwd_l<512, 4, 2>(FWD_ARGS) : (g == 8 ? fwd_l<512, 4, 8>(FWD_ARGS) : fwd_l<512, 4, 4>(FWD_ARGS)));
        default: return hipErrorInvalidValue;
    }
}

hipError_t lstm_step_bwd(float* gates, const float* whhT, const float* d_out, const float* csave, float* dc, int B, int T,
                         int H, int step, hipStream_t s) {
    const int nw = g_lstm_nw, g = g_lstm_g;
    switch (H) {
        case 64: return bwd_l<64, 4, 2>(BWD_ARGS);
        case 128: return bwd_l<128, 4, 4>(BWD_ARGS);
        case 256:
            if (nw >= 16) return bwd_l<256, 16, 4>(BWD_ARGS);
            if (nw >= 8) return bwd_l<256, 8, 4>(BWD_ARGS);
            return bwd_l<256, 4, 8>(BWD_ARGS);
        case 512:
            if (nw >= 16) return g == 4 ? bwd_l<512, 16, 4>(BWD_ARGS) : bwd_l<512, 16, 8>(BWD_ARGS);
            if (nw >= 8) return g == 4 ? bwd_l<512, 8, 4>(BWD_ARGS) : (g == 16 ? bwd_l<512, 8, 16>(BWD_ARGS) : bwd_l<512, 8, 8>(BWD_ARGS));
            return g == 4 ? bwd_l<512, 4, 4>(BWD_ARGS) : (g == 16 ? bwd_l<512, 4, 16>(BWD_ARGS) : bwd_l<512, 4, 8>(BWD_ARGS));
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ss
