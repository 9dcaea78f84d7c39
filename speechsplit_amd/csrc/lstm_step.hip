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
// Both MFMA operands are streamed from "fragment-major" copies: the exact 1 KiB (64 lanes x 16 B) a wave-instruction
// consumes is contiguous in memory, so every load instruction touches 8 full 128-byte lines instead of 16 half-used
// ones (the step is bound by L1/L2 line throughput per CU, measured: loads 4.5 us of an 8 us step).  W_hh is
// re-laid once per layer and training step (lstm_pack_w); h(t) / da(t) are written in that form by the epilogue of
// the step that produces them, into a ping-pong pair (zero-filled before step 0 = zero initial state).
// c(t-1) is read from the haloed cell slab itself (row t-1, or the all-zero halo row at the first step).
#include "common.h"
#include "kernels.h"

namespace ss {

int g_lstm_nw = 16;   // waves per workgroup in the step kernels (4, 8 or 16)
int g_lstm_g = 0;     // chunks per load group (0 = default for the chosen NW)
int g_lstm_mode = 0;  // diagnostics only: 1 = skip MFMAs, 2 = skip operand loads, 3 = empty kernel, 4 = skip epilogue

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return ss_sigmoid(x); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// grid = (H/16, ceil(B/16), 2), block = 64*NW
template <int H, int NW, int G>
__global__ __launch_bounds__(64 * NW) void lstm_step_fwd_kernel(float* __restrict__ gates, const float* __restrict__ wfrag,
                                                                const float* __restrict__ hf_cur, float* __restrict__ hf_next,
                                                                float* __restrict__ out, float* __restrict__ csave, int B,
                                                                int T, int step, int mode) {
    __shared__ float red[NW][4][16][16];
    if (mode == 3) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dir = blockIdx.z, j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int TP = T + 2 * HALO;
    const int tau = HALO + (dir == 0 ? step : T - 1 - step);
    const int tau_prev = dir == 0 ? tau - 1 : tau + 1;
    const int li = lane & 15;
    constexpr int kw = H / NW;                              // K range of this wave
    constexpr int nchunk = kw / 16;
    constexpr int NC = H / 16;                              // chunks along K
    static_assert(kw % 16 == 0 && nchunk % G == 0, "bad NW / G for this H");
    const int nbt = gridDim.y;
    // fragment-major operands: [dir][btile][chunk][lane][4] and [dir][jtile][gate][chunk][lane][4]
    const float* Ap = hf_cur + (((long)dir * nbt + blockIdx.y) * NC + w * nchunk) * 256 + lane * 4;
    const float* Bp[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) Bp[g] = wfrag + ((((long)dir * (H / 16) + blockIdx.x) * 4 + g) * NC + w * nchunk) * 256 + lane * 4;

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
                a[buf][i] = ld4(Ap + (c0 + i) * 256);
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[buf][i][g] = ld4(Bp[g] + (c0 + i) * 256);
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
        const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = ss_gate(pre[2], 2.0f), go = sigmoidf_(pre[3]);
        const float c = gf * cp + gi * gg;
        const float h = go * ss_tanh(c);
        grow[0] = gi;
        grow[H] = gf;
        grow[2 * H] = gg;
        grow[3 * H] = go;
        csave[o] = c;
        out[o] = h;
        // h(t) in MFMA-operand order for the next step: chunk = this workgroup's hidden tile
        hf_next[(((long)dir * nbt + blockIdx.y) * NC + blockIdx.x) * 256 + ((jj >> 2) * 16 + bi) * 4 + (jj & 3)] = h;
    }
}

// grid = (H/16, ceil(B/16), 2), block = 64*NW
template <int H, int NW, int G>
__global__ __launch_bounds__(64 * NW) void lstm_step_bwd_kernel(float* __restrict__ gates, const float* __restrict__ wfragT,
                                                                const float* __restrict__ gf_cur, float* __restrict__ gf_next,
                                                                const float* __restrict__ d_out,
                                                                const float* __restrict__ csave, float* __restrict__ dcs,
                                                                int B, int T, int step) {
    __shared__ float red[NW][16][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dir = blockIdx.z, j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int TP = T + 2 * HALO;
    const int tau = HALO + (dir == 0 ? T - 1 - step : step);
    const int tau_prev = dir == 0 ? tau - 1 : tau + 1;       // previous in forward order (c(t-1))
    const int li = lane & 15;
    constexpr int kw = 4 * H / NW;                           // K = 4H split over the waves
    constexpr int nchunk = kw / 16;
    constexpr int NC = 4 * H / 16;
    static_assert(kw % 16 == 0 && nchunk % G == 0, "bad NW / G for this H");
    const int nbt = gridDim.y;
    // da(t+1) of the previous backward step and W_hh^T, both fragment-major
    const float* Ap = gf_cur + (((long)dir * nbt + blockIdx.y) * NC + w * nchunk) * 256 + lane * 4;
    const float* Bp = wfragT + (((long)dir * (H / 16) + blockIdx.x) * NC + w * nchunk) * 256 + lane * 4;

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
            a[buf][i] = ld4(Ap + (c0 + i) * 256);
            v[buf][i] = ld4(Bp + (c0 + i) * 256);
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
        const float tc = ss_tanh(cc);
        const float d_o = dh * tc;
        const float dc = dc_rec + dh * go * (1.0f - tc * tc);
        *dcp = dc * gf;
        float da[4];
        da[0] = dc * gg * gi * (1.0f - gi);
        da[1] = dc * cp * gf * (1.0f - gf);
        da[2] = dc * gi * (1.0f - gg * gg);
        da[3] = d_o * go * (1.0f - go);
        float* gfp = gf_next + ((long)dir * nbt + blockIdx.y) * NC * 256 + ((jj >> 2) * 16 + bi) * 4 + (jj & 3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            grow[g * H] = da[g];                                     // slab copy: operand of the weight-gradient GEMMs
            gfp[(long)(g * (H / 16) + blockIdx.x) * 256] = da[g];    // fragment-major copy: operand of the next step
        }
    }
}

// W_hh [4H][H] of both directions -> forward fragments [dir][jtile][gate][chunk][lane][4] (transposed == 0)
//                                 or backward fragments [dir][jtile][chunk over 4H][lane][4] of W_hh^T (transposed == 1)
__global__ __launch_bounds__(256) void lstm_pack_w_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_b,
                                                          float* __restrict__ frag, int H, int transposed) {
    const long n = 2L * 4 * H * H;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int q = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const int li = lane & 15, kg = lane >> 4;
        long rest = i >> 8;
        float v;
        if (!transposed) {
            const int NC = H / 16;
            const int c = (int)(rest % NC);
            rest /= NC;
            const int g = (int)(rest & 3);
            rest >>= 2;
            const int jt = (int)(rest % (H / 16));
            const int dir = (int)(rest / (H / 16));
            const float* w = dir ? whh_b : whh_f;
            v = w[(long)(g * H + jt * 16 + li) * H + c * 16 + kg * 4 + q];
        } else {
            const int NC = 4 * H / 16;
            const int c = (int)(rest % NC);
            rest /= NC;
            const int jt = (int)(rest % (H / 16));
            const int dir = (int)(rest / (H / 16));
            const float* w = dir ? whh_b : whh_f;
            v = w[(long)(c * 16 + kg * 4 + q) * H + jt * 16 + li];
        }
        frag[i] = v;
    }
}

template <int H, int NW, int G>
hipError_t fwd_l(float* gates, const float* wfrag, const float* hf_cur, float* hf_next, float* out, float* csave, int B, int T,
                 int step, hipStream_t s) {
    hipLaunchKernelGGL((lstm_step_fwd_kernel<H, NW, G>), dim3(H / 16, cdiv(B, 16), 2), dim3(64 * NW), 0, s, gates, wfrag, hf_cur,
                       hf_next, out, csave, B, T, step, g_lstm_mode);
    return hipGetLastError();
}
template <int H, int NW, int G>
hipError_t bwd_l(float* gates, const float* wfragT, const float* gf_cur, float* gf_next, const float* d_out, const float* csave,
                 float* dc, int B, int T, int step, hipStream_t s) {
    hipLaunchKernelGGL((lstm_step_bwd_kernel<H, NW, G>), dim3(H / 16, cdiv(B, 16), 2), dim3(64 * NW), 0, s, gates, wfragT, gf_cur,
                       gf_next, d_out, csave, dc, B, T, step);
    return hipGetLastError();
}

}  // namespace

#define FWD_ARGS gates, wfrag, hf_cur, hf_next, out, csave, B, T, step, s
#define BWD_ARGS gates, wfragT, gf_cur, gf_next, d_out, csave, dc, B, T, step, s

hipError_t lstm_pack_w(const float* whh_f, const float* whh_b, float* frag, int H, int transposed, hipStream_t s) {
    hipLaunchKernelGGL(lstm_pack_w_kernel, dim3(2048), dim3(256), 0, s, whh_f, whh_b, frag, H, transposed);
    return hipGetLastError();
}

hipError_t lstm_step_fwd(float* gates, const float* wfrag, const float* hf_cur, float* hf_next, float* out, float* csave,
                         int B, int T, int H, int step, hipStream_t s) {
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

hipError_t lstm_step_bwd(float* gates, const float* wfragT, const float* gf_cur, float* gf_next, const float* d_out,
                         const float* csave, float* dc, int B, int T, int H, int step, hipStream_t s) {
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
