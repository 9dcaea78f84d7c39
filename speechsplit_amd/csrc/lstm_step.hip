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
// 16-byte, row-contiguous, and need no LDS.  The four waves of a workgroup split K and reduce through LDS.
//
// h(t-1) and c(t-1) are read from the haloed output / cell slabs themselves (row t-1, or the all-zero halo row at
// the first step), so there is no separate state buffer and no branch for the initial state.
#include "common.h"
#include "kernels.h"

namespace ss {

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// grid = (H/16, ceil(B/16), 2), block = 256
template <int H>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(float* __restrict__ gates, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_b, float* __restrict__ out,
                                                            float* __restrict__ csave, int B, int T, int step) {
    __shared__ float red[4][4][16][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dir = blockIdx.z, j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int TP = T + 2 * HALO;
    const int tau = HALO + (dir == 0 ? step : T - 1 - step);
    const int tau_prev = dir == 0 ? tau - 1 : tau + 1;
    const float* whh = dir ? whh_b : whh_f;
    const int li = lane & 15, lk = (lane >> 4) * 4;
    constexpr int kw = H / 4;                               // K range of this wave
    int bA = b0 + li;
    if (bA > B - 1) bA = B - 1;
    const float* Ap = out + ((long)bA * TP + tau_prev) * (2 * H) + dir * H + w * kw + lk;
    const float* Bp[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) Bp[g] = whh + (long)(g * H + j0 + li) * H + w * kw + lk;

    // operands of the cell update, requested before the contraction so their latency hides under it
    const int bi = tid >> 4, jj = tid & 15;
    const int b = b0 + bi, j = j0 + jj;
    const int bc = b < B ? b : B - 1;
    float* grow = gates + ((long)bc * TP + tau) * (8 * H) + dir * 4 * H + j;
    float xg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xg[g] = grow[g * H];
    const float cp = csave[((long)bc * TP + tau_prev) * (2 * H) + dir * H + j];

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int nchunk = kw / 16;
#pragma unroll
    for (int c = 0; c < nchunk; ++c) {
        const f32x4 a = ld4(Ap + c * 16);
        f32x4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = ld4(Bp[g] + c * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], bv[g][q], acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][g][(lane >> 4) * 4 + r][li] = acc[g][r];
    __syncthreads();

    if (b < B) {
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pre[g] = xg[g] + ((red[0][g][bi][jj] + red[1][g][bi][jj]) + (red[2][g][bi][jj] + red[3][g][bi][jj]));
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

// grid = (H/16, ceil(B/16), 2), block = 256
template <int H>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(float* __restrict__ gates, const float* __restrict__ whhT,
                                                            const float* __restrict__ d_out,
                                                            const float* __restrict__ csave, float* __restrict__ dcs,
                                                            int B, int T, int step) {
    __shared__ float red[4][16][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dir = blockIdx.z, j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int TP = T + 2 * HALO;
    const int tau = HALO + (dir == 0 ? T - 1 - step : step);
    const int tau_next = dir == 0 ? tau + 1 : tau - 1;       // processed by the previous backward step (halo at step 0)
    const int tau_prev = dir == 0 ? tau - 1 : tau + 1;       // previous in forward order (c(t-1))
    const int li = lane & 15, lk = (lane >> 4) * 4;
    constexpr int kw = H;                                    // K = 4H split over 4 waves
    int bA = b0 + li;
    if (bA > B - 1) bA = B - 1;
    const float* Ap = gates + ((long)bA * TP + tau_next) * (8 * H) + dir * 4 * H + w * kw + lk;
    const float* Bp = whhT + ((long)dir * H + j0 + li) * (4 * H) + w * kw + lk;
    const int bi = tid >> 4, jj = tid & 15;
    const int b = b0 + bi, j = j0 + jj;
    const int bc = b < B ? b : B - 1;
    const long o = ((long)bc * TP + tau) * (2 * H) + dir * H + j;
    float* grow = gates + ((long)bc * TP + tau) * (8 * H) + dir * 4 * H + j;
    const float gi = grow[0], gf = grow[H], gg = grow[2 * H], go = grow[3 * H];
    const float p_do = d_out[o];
    const float cc = csave[o];
    const float cp = csave[((long)bc * TP + tau_prev) * (2 * H) + dir * H + j];
    float* dcp = dcs + ((long)dir * B + bc) * H + j;
    const float dc_rec = step == 0 ? 0.f : *dcp;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int nchunk = kw / 16;                          // even (H % 64 == 0)
#pragma unroll 8
    for (int c = 0; c < nchunk; c += 2) {
        const f32x4 a0 = ld4(Ap + c * 16), a1 = ld4(Ap + c * 16 + 16);
        const f32x4 v0 = ld4(Bp + c * 16), v1 = ld4(Bp + c * 16 + 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], v0[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], v1[q], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][(lane >> 4) * 4 + r][li] = acc0[r] + acc1[r];
    __syncthreads();

    if (b < B) {
        const float dh = p_do + ((red[0][bi][jj] + red[1][bi][jj]) + (red[2][bi][jj] + red[3][bi][jj]));
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

}  // namespace

template <int H>
static hipError_t fwd_h(float* gates, const float* whh_f, const float* whh_b, float* out, float* csave, int B, int T, int step,
                        hipStream_t s) {
    hipLaunchKernelGGL((lstm_step_fwd_kernel<H>), dim3(H / 16, cdiv(B, 16), 2), dim3(256), 0, s, gates, whh_f, whh_b, out,
                       csave, B, T, step);
    return hipGetLastError();
}
template <int H>
static hipError_t bwd_h(float* gates, const float* whhT, const float* d_out, const float* csave, float* dc, int B, int T,
                        int step, hipStream_t s) {
    hipLaunchKernelGGL((lstm_step_bwd_kernel<H>), dim3(H / 16, cdiv(B, 16), 2), dim3(256), 0, s, gates, whhT, d_out, csave,
                       dc, B, T, step);
    return hipGetLastError();
}

hipError_t lstm_step_fwd(float* gates, const float* whh_f, const float* whh_b, float* out, float* csave, int B, int T,
                         int H, int step, hipStream_t s) {
    switch (H) {
        case 64: return fwd_h<64>(gates, whh_f, whh_b, out, csave, B, T, step, s);
        case 128: return fwd_h<128>(gates, whh_f, whh_b, out, csave, B, T, step, s);
        case 256: return fwd_h<256>(gates, whh_f, whh_b, out, csave, B, T, step, s);
        case 512: return fwd_h<512>(gates, whh_f, whh_b, out, csave, B, T, step, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t lstm_step_bwd(float* gates, const float* whhT, const float* d_out, const float* csave, float* dc, int B, int T,
                         int H, int step, hipStream_t s) {
    switch (H) {
        case 64: return bwd_h<64>(gates, whhT, d_out, csave, dc, B, T, step, s);
        case 128: return bwd_h<128>(gates, whhT, d_out, csave, dc, B, T, step, s);
        case 256: return bwd_h<256>(gates, whhT, d_out, csave, dc, B, T, step, s);
        case 512: return bwd_h<512>(gates, whhT, d_out, csave, dc, B, T, step, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ss
