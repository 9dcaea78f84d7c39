// HBM/L2-bound kernels of the SpeechSplit path: GroupNorm+ReLU (fwd/bwd), bias/affine gradient sums, weight
// re-layouts, decoder-input assembly, losses and the flat Adam update.
#include "common.h"
#include "kernels.h"

namespace ss {

extern int g_deterministic;
int g_gn_part = 0;       // GroupNorm backward: per-utterance affine / bias gradient sums to scratch + ordered reduce instead of f32 atomics (ss_tune("gn_part"))

namespace {

constexpr int GN_MAXIT = 16;     // T <= 256
constexpr float GN_EPS = 1e-5f;

// ------------------------------------------------------------------------------------------------ GroupNorm + ReLU
// One workgroup per (utterance, 64 channels = 4 groups).  Thread (rg = tid>>4, l16 = tid&15) owns float4 channel
// slot l16 of rows rg, rg+16, ...; the whole (T x 64) slab stays in registers between the statistics and the output.
__device__ __forceinline__ float block_group_sum(float v, float* red, float* out4, int tid) {
    red[tid] = v;
    __syncthreads();
    if (tid < 4) {
        double s = 0.0;             // the 64 partial sums of a group in float64 (round 4): the statistics scale every activation of the block
        for (int rg = 0; rg < 16; ++rg)
#pragma unroll
            for (int l = 0; l < 4; ++l) s += (double)red[rg * 16 + tid * 4 + l];
        out4[tid] = (float)s;
    }
    __syncthreads();
    return out4[(tid & 15) >> 2];
}

// The normalised value and the affine output of one element, with the roundings spelled out: the forward kernel, its fused form and the
// backward's recomputation (which decides the ReLU branch) must produce the same bits, and hipcc contracts / vectorises the plain
// expression differently from kernel to kernel (measured: last-bit differences between gn_relu_fwd_kernel and the fused kernel)
__device__ __forceinline__ float gn_h(float x, float mean, float rstd) { return __fmul_rn(__fsub_rn(x, mean), rstd); }
__device__ __forceinline__ float gn_z(float h, float ga, float be) { return __fmaf_rn(h, ga, be); }

__global__ __launch_bounds__(256) void gn_relu_fwd_kernel(const float* __restrict__ x, long x_ld, long x_bs,
                                                          float* __restrict__ y, long y_ld, long y_bs,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ stats, int T, int C) {
    __shared__ float red[256];
    __shared__ float g4[4];
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + l16 * 4;
    const int nit = (T + 15) >> 4;
    const float* xb = x + b * x_bs + (long)HALO * x_ld + c;
    f32x4 v[GN_MAXIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < GN_MAXIT; ++it) {
        const int t = rg + it * 16;
        if (it < nit && t < T) {
            v[it] = *reinterpret_cast<const f32x4*>(xb + (long)t * x_ld);
            s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
        } else {
            v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float inv_n = 1.0f / (16.0f * (float)T);
    const float mean = block_group_sum(s, red, g4, tid) * inv_n;
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < GN_MAXIT; ++it) {
        const int t = rg + it * 16;
        if (it < nit && t < T) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dlt = __fsub_rn(v[it][j], mean);
                ss = __fmaf_rn(dlt, dlt, ss);
            }
        }
    }
    const float var = block_group_sum(ss, red, g4, tid) * inv_n;
    const float rstd = 1.0f / sqrtf(var + GN_EPS);
    if (rg == 0 && (l16 & 3) == 0) {
        const int g = (c >> 4);
        stats[((long)b * (C >> 4) + g) * 2 + 0] = mean;
        stats[((long)b * (C >> 4) + g) * 2 + 1] = rstd;
    }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
    float* yb = y + b * y_bs + (long)HALO * y_ld + c;
#pragma unroll
    for (int it = 0; it < GN_MAXIT; ++it) {
        const int t = rg + it * 16;
        if (it < nit && t < T) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = gn_z(gn_h(v[it][j], mean, rstd), ga[j], be[j]);
                o[j] = z > 0.f ? z : 0.f;
            }
            *reinterpret_cast<f32x4*>(yb + (long)t * y_ld) = o;
        }
    }
}

// GroupNorm + ReLU + the random resampling that follows it in a training forward (model.py:164-170 then 199-206), one pass: the
// (T x 64) tile is normalised in registers exactly as gn_relu_fwd_kernel does, left in LDS (row T stays zero: the halo row the
// gather's i0 + 1 may touch), and the output rows r < nrows[b] are (1 - lam) * tile[i0] + lam * tile[i0 + 1] with the gather's three
// roundings (interp.hip interp_gather_kernel: bit-identical results), rows past nrows zero.  The normalised slab itself is never
// written: nobody but the gather reads it (the backward recomputes from the conv output and the statistics).
// y / y_img: the resampled slab and its pre-split image AT the first real row and the block's first column; P output rows.
template <int MAXIT>
__global__ __launch_bounds__(256, MAXIT <= 8 ? 5 : 4) void gn_relu_gather_kernel(const float* __restrict__ x, long x_ld, long x_bs, float* __restrict__ y, long y_ld,
                                                             long y_bs, float* __restrict__ y_img, const float* __restrict__ img_scale,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ stats, int T, int C, int P, const int* __restrict__ i0,
                                                             const float* __restrict__ lam, const int* __restrict__ nrows, int img_bf16) {
    __shared__ float red[256];
    __shared__ float g4[4];
    extern __shared__ __attribute__((aligned(16))) float gn_tile[];       // [T + 1][64]
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + l16 * 4;
    const int nit = (T + 15) >> 4;
    const float* xb = x + b * x_bs + (long)HALO * x_ld + c;
    f32x4 v[MAXIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int t = rg + it * 16;
        if (it < nit && t < T) {
            v[it] = *reinterpret_cast<const f32x4*>(xb + (long)t * x_ld);
            s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
        } else {
            v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float inv_n = 1.0f / (16.0f * (float)T);
    const float mean = block_group_sum(s, red, g4, tid) * inv_n;
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int t = rg + it * 16;
        if (it < nit && t < T) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dlt = __fsub_rn(v[it][j], mean);
                ss = __fmaf_rn(dlt, dlt, ss);
            }
        }
    }
    const float var = block_group_sum(ss, red, g4, tid) * inv_n;
    const float rstd = 1.0f / sqrtf(var + GN_EPS);
    if (rg == 0 && (l16 & 3) == 0) {
        const int g = (c >> 4);
        stats[((long)b * (C >> 4) + g) * 2 + 0] = mean;
        stats[((long)b * (C >> 4) + g) * 2 + 1] = rstd;
    }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int t = rg + it * 16;
        if (it < nit && t < T) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = gn_z(gn_h(v[it][j], mean, rstd), ga[j], be[j]);
                o[j] = z > 0.f ? z : 0.f;
            }
            *reinterpret_cast<f32x4*>(gn_tile + t * 64 + l16 * 4) = o;
        }
    }
    if (rg == 0) *reinterpret_cast<f32x4*>(gn_tile + T * 64 + l16 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int live = nrows[b];
    const float isc = (y_img && img_scale) ? *img_scale : 16.0f;
    float* yb = y + b * y_bs + blockIdx.x * 64 + l16 * 4;
    const long yie = b * y_bs + blockIdx.x * 64 + l16 * 4;        // element index of this thread's column group in the image
    for (int r = rg; r < P; r += 16) {
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r < live) {
            const int i = i0[(long)b * P + r];
            const float l = lam[(long)b * P + r];
            const float ol = __fsub_rn(1.0f, l);
            const f32x4 a = *reinterpret_cast<const f32x4*>(gn_tile + i * 64 + l16 * 4);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(gn_tile + (i + 1) * 64 + l16 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = ss_lerp_rn(ol, a[j], l, bb[j]);      // model.py:430, three roundings
        }
        *reinterpret_cast<f32x4*>(yb + (long)r * y_ld) = o;
        if (y_img) {
            if (r < live || img_bf16) ss_store_img4(y_img, yie + (long)r * y_ld, o[0], o[1], o[2], o[3], isc, img_bf16);      // (o is zero past the live rows)
            else *reinterpret_cast<f32x4*>(y_img + yie + (long)r * y_ld) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// MAXIT: row iterations per thread the registers are sized for (16 rows per iteration): 12 for T <= 192 -- the (T x 64) tile of normalised
// values and gradients then takes 96 registers instead of 128 and three waves share a SIMD instead of two
template <int MAXIT>
__global__ __launch_bounds__(256, MAXIT <= 12 ? 3 : 2) void gn_relu_bwd_kernel(const float* __restrict__ x, long x_ld, long x_bs,
                                                          float* __restrict__ dy, long dy_ld, long dy_bs,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ stats, float* __restrict__ g_gamma,
                                                          float* __restrict__ g_beta, float* __restrict__ g_bias,
                                                          unsigned* __restrict__ amax, float* __restrict__ part, int B, int T, int C,
                                                          const float* __restrict__ src, long src_ld, long src_bs, int P,
                                                          const float* __restrict__ lam, const int* __restrict__ start, float* __restrict__ dy_img) {
    // dy_img (nullable; 16-bit data path): the conv-output gradient once more as a plain bf16 tensor of dy's geometry (the operand image of the
    // block's weight- and input-gradient contractions), written here instead of by a pass of its own over the slab
    // src (nullable): the gradient of the RESAMPLED block output [P rows at src, first real row / first column of the block] -- the adjoint of
    // the training forward's gather (interp.hip interp_scatter_kernel, same terms in the same order: bit-identical) is then taken on the fly
    // instead of being read from dy, which is only written
    __shared__ float red[256];
    __shared__ float g4[4];
    __shared__ float colred[3][16][64];
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + l16 * 4;
    const int nit = (T + 15) >> 4;
    const int g = c >> 4;
    const float mean = stats[((long)b * (C >> 4) + g) * 2 + 0];
    const float rstd = stats[((long)b * (C >> 4) + g) * 2 + 1];
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
    const float* xb = x + b * x_bs + (long)HALO * x_ld + c;
    float* db = dy + b * dy_bs + (long)HALO * dy_ld + c;
    f32x4 xh[MAXIT], dh[MAXIT];
    f32x4 dgam = {0.f, 0.f, 0.f, 0.f}, dbet = {0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    // the utterance's inverse map and weights first, into LDS: each row's source rows then cost one global load each instead of a chain
    // of three dependent ones (start -> lam -> data), eight rows per thread in a row
    __shared__ int s_start[16 * GN_MAXIT + 2];
    __shared__ float s_lam[16 * GN_MAXIT + 2];
    // ... and (round 4) the utterance's P source rows of this block's 64 channels themselves: a thread's rows each need 2-3 source rows, fetched
    // one after the other behind data-dependent loop bounds -- twelve rows x three dependent HBM trips made the kernel latency-bound (60 us for
    // 60 MB of traffic); one coalesced bulk copy with every load in flight, then the adjoint runs out of LDS (same terms, same order: bit-identical)
    extern __shared__ __attribute__((aligned(16))) float gn_src[];      // [P][64] when src is given
    if (src) {
        for (int i = tid; i <= T; i += 256) s_start[i] = start[(long)b * (T + 1) + i];
        for (int i = tid; i < P && i < 16 * GN_MAXIT + 2; i += 256) s_lam[i] = lam[(long)b * P + i];
        const float* sb0 = src + b * src_bs + blockIdx.x * 64;
        for (int i = tid; i < P * 16; i += 256) {
            const int r = i >> 4, q = i & 15;
            *reinterpret_cast<f32x4*>(gn_src + r * 64 + q * 4) = *reinterpret_cast<const f32x4*>(sb0 + (long)r * src_ld + q * 4);
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int t = rg + it * 16;
        if (it < nit && t < T) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + (long)t * x_ld);
            f32x4 dv;
            if (src) {
                const int* st = s_start;
                const float* lm = s_lam;
                const float* sb = gn_src + l16 * 4;
                const int a0 = st[t], a1 = st[t + 1];
                const int b0 = t > 0 ? st[t - 1] : 0, b1 = t > 0 ? a0 : 0;
                dv = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int r = a0; r < a1; ++r) {
                    const float w = 1.0f - lm[r];
                    const f32x4 g = *reinterpret_cast<const f32x4*>(sb + r * 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dv[j] = __builtin_fmaf(w, g[j], dv[j]);
                }
                for (int r = b0; r < b1; ++r) {
                    const float w = lm[r];
                    const f32x4 g = *reinterpret_cast<const f32x4*>(sb + r * 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dv[j] = __builtin_fmaf(w, g[j], dv[j]);
                }
            } else {
                dv = *reinterpret_cast<const f32x4*>(db + (long)t * dy_ld);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h = gn_h(xv[j], mean, rstd);
                const float z = gn_z(h, ga[j], be[j]);
                const float dz = z > 0.f ? dv[j] : 0.f;
                dgam[j] += dz * h;
                dbet[j] += dz;
                const float dxh = dz * ga[j];
                xh[it][j] = h;
                dh[it][j] = dxh;
                s1 += dxh;
                s2 += dxh * h;
            }
        } else {
            xh[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            dh[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float inv_n = 1.0f / (16.0f * (float)T);
    const float m1 = block_group_sum(s1, red, g4, tid) * inv_n;
    const float m2 = block_group_sum(s2, red, g4, tid) * inv_n;
    f32x4 dbias = {0.f, 0.f, 0.f, 0.f};
    float amx = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int t = rg + it * 16;
        if (it < nit && t < T) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = rstd * (dh[it][j] - m1 - xh[it][j] * m2);
                dbias[j] += o[j];
                amx = fmaxf(amx, fabsf(o[j]));
            }
            *reinterpret_cast<f32x4*>(db + (long)t * dy_ld) = o;
            if (dy_img) ss_store_img4(dy_img, b * dy_bs + (long)(HALO + t) * dy_ld + c, o[0], o[1], o[2], o[3], 1.0f, 1);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        colred[0][rg][l16 * 4 + j] = dgam[j];
        colred[1][rg][l16 * 4 + j] = dbet[j];
        colred[2][rg][l16 * 4 + j] = dbias[j];
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
        if ((tid & 63) == 0) atomicMax(amax, __float_as_uint(amx));
    }
    __syncthreads();
    if (tid < 192) {
        const int which = tid >> 6, cc = tid & 63;
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += colred[which][r][cc];
        float* dst = which == 0 ? g_gamma : which == 1 ? g_beta : g_bias;     // sums over the utterances meet in the arena
        if (part) part[((long)b * 3 + which) * C + blockIdx.x * 64 + cc] = s;  // deterministic mode: summed in utterance order below
        else atomicAdd(dst + blockIdx.x * 64 + cc, s);
    }
}

// deterministic mode: dst[which][c] += sum over b (in order) of part[b][which][c]
__global__ __launch_bounds__(256) void gn_part_reduce_kernel(const float* __restrict__ part, int B, int C, float* __restrict__ g_gamma,
                                                             float* __restrict__ g_beta, float* __restrict__ g_bias) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * C) return;
    const int which = i / C, c = i - which * C;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[((long)b * 3 + which) * C + c];
    float* dst = which == 0 ? g_gamma : which == 1 ? g_beta : g_bias;
    dst[c] += s;
}

// Test hook: the ReLU branch the engine took for every element, recomputed from the saved conv output and statistics with the
// expression gn_relu_bwd_kernel uses (which is also the forward kernel's).  mask [B, T, C] dense, 1.0f where z > 0.
__global__ __launch_bounds__(256) void gn_relu_mask_kernel(const float* __restrict__ x, long x_ld, long x_bs,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stats, float* __restrict__ mask, int T, int C) {
    const int b = blockIdx.y;
    const long n = (long)T * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i / C), c = (int)(i - (long)t * C);
        const int g = c >> 4;
        const float mean = stats[((long)b * (C >> 4) + g) * 2 + 0];
        const float rstd = stats[((long)b * (C >> 4) + g) * 2 + 1];
        const float h = gn_h(x[b * x_bs + (long)(t + HALO) * x_ld + c], mean, rstd);
        const float z = gn_z(h, gamma[c], beta[c]);
        mask[(long)b * n + i] = z > 0.f ? 1.f : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ column sums
// Bias-type gradients (LSTM biases, the head's bias): out[c] += sum over R rows of in[r][c].  At a trained state these are heavily
// cancelling sums (the head's is sum_r (softmax - onehot)); fp32 chains over thousands of rows met through arrival-order atomics put them
// 8x further from exact than PyTorch's cascade sum (round-3 review).  Now: float64 accumulation throughout and a FIXED order --
// grid = (ceil(cols / 64), chunks), block 256 = 64 columns x 4 row lanes; a block adds its rows per lane, the four lanes in order, and
// writes the chunk's float64 partial to scratch; the LAST block of a column block to arrive (self-resetting counter) adds the chunks'
// partials in chunk order and accumulates into the outputs.  Deterministic whatever the arrival order.  chunks == 1: no scratch, no counter.
// Column c < C goes to o0 (and o1 if given), C <= c < 2C (two-direction form) to o2 / o3.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, long ld, int R, int C, int ncols, int rows_per_chunk,
                                                     double* __restrict__ part, unsigned* __restrict__ ctr, float* __restrict__ o0,
                                                     float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ o3) {
    __shared__ double red[4][64];
    __shared__ int s_last;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int chunks = gridDim.y;
    const int r0 = blockIdx.y * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    double s = 0.0;
    if (c < ncols) {
        const float* p = in + c;
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {            // four independent loads in flight per thread
            const float v0 = p[(long)r * ld], v1 = p[(long)(r + 4) * ld], v2 = p[(long)(r + 8) * ld], v3 = p[(long)(r + 12) * ld];
            s += (double)v0;
            s += (double)v1;
            s += (double)v2;
            s += (double)v3;
        }
        for (; r < r1; r += 4) s += (double)p[(long)r * ld];
    }
    red[rl][cl] = s;
    __syncthreads();
    double tot = 0.0;
    if (rl == 0) tot = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
    if (chunks > 1) {
        // Cross-workgroup hand-over WITHOUT fences: a release / acquire fence at agent scope writes back / invalidates the XCD's whole L2 --
        // measured: 500 workgroups doing that took 0.6 ms and slowed the GEMMs running beside them by 15 %.  Instead the partials are stored
        // write-through (sc1), acknowledged (vmcnt) before the arrival counter is bumped, and read back with sc1 loads (as the persistent
        // recurrences exchange their payload, lstm_seq.hip).
        if (rl == 0) __hip_atomic_store(part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 64 + cl, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(ctr + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(chunks - 1);
        __syncthreads();
        if (!s_last) return;
        double t4 = 0.0;                            // row lane rl adds chunks rl, rl + 4, ... in order (eight loads in flight); the four lanes meet in order
        const double* pp = part + (long)blockIdx.x * 64 + cl;
        const long cs = (long)gridDim.x * 64;       // doubles between consecutive chunks
        int k = rl;
        for (; k + 28 < chunks; k += 32) {
            double v[8];
            asm volatile(
                "global_load_dwordx2 %0, %8, off sc1\n\t"
                "global_load_dwordx2 %1, %9, off sc1\n\t"
                "global_load_dwordx2 %2, %10, off sc1\n\t"
                "global_load_dwordx2 %3, %11, off sc1\n\t"
                "global_load_dwordx2 %4, %12, off sc1\n\t"
                "global_load_dwordx2 %5, %13, off sc1\n\t"
                "global_load_dwordx2 %6, %14, off sc1\n\t"
                "global_load_dwordx2 %7, %15, off sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                : "v"(pp + k * cs), "v"(pp + (k + 4) * cs), "v"(pp + (k + 8) * cs), "v"(pp + (k + 12) * cs), "v"(pp + (k + 16) * cs), "v"(pp + (k + 20) * cs),
                  "v"(pp + (k + 24) * cs), "v"(pp + (k + 28) * cs)
                : "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) t4 += v[i];
        }
        for (; k < chunks; k += 4) t4 += __hip_atomic_load(pp + k * cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        red[rl][cl] = t4;
        __syncthreads();
        if (rl == 0) tot = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
        if (threadIdx.x == 0) __hip_atomic_store(ctr + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
    }
    if (rl == 0 && c < ncols) {
        const float t = (float)tot;
        float* a = c < C ? o0 : o2;
        float* b = c < C ? o1 : o3;
        const int cc = c < C ? c : c - C;
        a[cc] += t;
        if (b) b[cc] += t;
    }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, long s_ld, long s_bs,
                                                        float* __restrict__ dst, long d_ld, long d_bs, int T, int C) {
    const int b = blockIdx.y;
    const long n = (long)T * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i / C), c = (int)(i - (long)t * C);
        dst[b * d_bs + t * d_ld + c] = src[b * s_bs + t * s_ld + c];
    }
}

// Batch assembly from an HBM-resident corpus (reference data_loader.py:101-128): utterance b is rows [row0[b], row0[b] + len[b])
// of the concatenated corpus; mel is clipped to [0,1] and zero-padded to T rows, F0 is padded with -1e10, the speaker row is copied.
__global__ __launch_bounds__(256) void collate_kernel(const float* __restrict__ mel_cat, const float* __restrict__ f0_cat,
                                                      const float* __restrict__ emb_tab, const long* __restrict__ row0,
                                                      const int* __restrict__ len, const int* __restrict__ item, int T, int C,
                                                      int E, float* __restrict__ mel, float* __restrict__ f0,
                                                      float* __restrict__ emb) {
    const int b = blockIdx.y;
    const long r0 = row0[b];
    const int n = len[b];
    const long tot = (long)T * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
        const int t = (int)(i / C);
        float v = 0.f;
        if (t < n) {
            v = mel_cat[r0 * C + i];
            v = fminf(fmaxf(v, 0.f), 1.f);
        }
        mel[(long)b * tot + i] = v;
    }
    for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) f0[(long)b * T + t] = t < n ? f0_cat[r0 + t] : -1e10f;
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < E; k += 256) emb[(long)b * E + k] = emb_tab[(long)item[b] * E + k];
}

__device__ __forceinline__ void conv_pack_body(const float* __restrict__ w, int Co, int Ci, int Cp,
                                               float* __restrict__ wf, float* __restrict__ wb, float* __restrict__ wf_img,
                                               float* __restrict__ wb_img, int img_bf16) {
    // wf[co][k][cp] ; wb[ci][k][co] = w[co][ci][4-k].  Cp and Co are multiples of 4: a thread writes one group of four (and its image)
    const long nf = (long)Co * 5 * Cp / 4;
    const long nb = (long)Ci * 5 * Co / 4;
    for (long gi = blockIdx.x * 256L + threadIdx.x; gi < nf + nb; gi += (long)gridDim.x * 256) {
        float v[4];
        if (gi < nf) {
            const long i = gi * 4;
            const int cp = (int)(i % Cp);
            const int k = (int)((i / Cp) % 5);
            const int co = (int)(i / (5L * Cp));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = cp + j < Ci ? w[((long)co * Ci + cp + j) * 5 + k] : 0.f;
            reinterpret_cast<float4*>(wf)[gi] = make_float4(v[0], v[1], v[2], v[3]);
            if (wf_img) ss_store_img4(wf_img, 4 * gi, v[0], v[1], v[2], v[3], 16.0f, img_bf16);
        } else if (wb) {
            const long q = (gi - nf) * 4;
            const int co = (int)(q % Co);
            const int k = (int)((q / Co) % 5);
            const int ci = (int)(q / (5L * Co));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = w[((long)(co + j) * Ci + ci) * 5 + (4 - k)];
            reinterpret_cast<float4*>(wb)[gi - nf] = make_float4(v[0], v[1], v[2], v[3]);
            if (wb_img) ss_store_img4(wb_img, 4 * (gi - nf), v[0], v[1], v[2], v[3], 16.0f, img_bf16);
        }
    }
}
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, int Co, int Ci, int Cp,
                                                        float* __restrict__ wf, float* __restrict__ wb, float* __restrict__ wf_img,
                                                        float* __restrict__ wb_img, int img_bf16) {
    conv_pack_body(w, Co, Ci, Cp, wf, wb, wf_img, wb_img, img_bf16);
}
// grid = (blocks, tasks)
__global__ __launch_bounds__(256) void conv_pack_many_kernel(ConvPackTable tb) {
    const ConvPackTask t = tb.t[blockIdx.y];
    conv_pack_body(t.w, t.Co, t.Ci, t.Cp, t.wf, t.wb, t.wf_img, t.wb_img, tb.img_bf16);
}

__global__ __launch_bounds__(256) void conv_unpack_grad_kernel(const float* __restrict__ gp, int Co, int Ci, int Cp,
                                                               float* __restrict__ g) {
    const long n = (long)Co * Ci * 5;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int k = (int)(i % 5);
        const int ci = (int)((i / 5) % Ci);
        const int co = (int)(i / (5L * Ci));
        g[i] = gp[((long)co * 5 + k) * Cp + ci];
    }
}

// the same for several blocks in one launch (blockIdx.y = block): the one-GPU step unpacks every conv weight gradient at the end of the backward
__global__ __launch_bounds__(256) void conv_unpack_grads_kernel(ConvUnpackTable tb) {
    const ConvUnpackTask t = tb.t[blockIdx.y];
    const long n = (long)t.Co * t.Ci * 5;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int k = (int)(i % 5);
        const int ci = (int)((i / 5) % t.Ci);
        const int co = (int)(i / (5L * t.Ci));
        t.g[i] = t.gp[((long)co * 5 + k) * t.Cp + ci];
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int R, int C, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = in[(long)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) out[(long)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

__global__ __launch_bounds__(256) void add_vec_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// Per-step parameter re-layouts of every LSTM block in ONE launch (they used to be ~30 launches of a few microseconds
// each, a chain the streams beside it ended up waiting for): task t is  dst = a + b  (b == nullptr: a copy).
// grid = (PREP_BLOCKS, number of tasks), block 256
constexpr int PREP_BLOCKS = 64;
__global__ __launch_bounds__(256) void prep_kernel(PrepTable tb) {
    const PrepTask t = tb.t[blockIdx.y];
    const long stride = (long)PREP_BLOCKS * 256;
    const bool vec = (t.n & 3) == 0 && (((size_t)t.a | (size_t)t.b | (size_t)t.dst) & 15) == 0;
    if (vec) {
        const float4* a = reinterpret_cast<const float4*>(t.a);
        const float4* b = reinterpret_cast<const float4*>(t.b);
        float4* d = reinterpret_cast<float4*>(t.dst);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < t.n / 4; i += stride) {
            float4 v = a[i];
            if (b) {
                const float4 w = b[i];
                v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
            }
            d[i] = v;
            if (t.img) ss_store_img4(t.img, 4 * i, v.x, v.y, v.z, v.w, 16.0f, tb.img_bf16);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < t.n; i += stride) t.dst[i] = t.a[i] + (t.b ? t.b[i] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ decoder input
struct CodeSrcPack {
    CodeSrc s[4];
    int n;
};

// grid = (T, B), block 256
__global__ __launch_bounds__(256) void build_dec_in_kernel(CodeSrcPack p, const float* __restrict__ emb, int emb_dim,
                                                           int emb_col, float* __restrict__ dec_in, int ld, int T) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int TP = T + 2 * HALO;
    float* row = dec_in + ((long)b * TP + t + HALO) * ld;
    for (int c = threadIdx.x; c < ld; c += 256) {
        float v = 0.f;
        if (c >= emb_col && c < emb_col + emb_dim) {
            v = emb[(long)b * emb_dim + c - emb_col];      // c_trg broadcast over time (model.py:309)
        } else {
            for (int i = 0; i < p.n; ++i) {
                const int H = p.s[i].H, col = p.s[i].col;
                if (c >= col && c < col + 2 * H) {
                    const int cc = c - col, f = p.s[i].freq;
                    const int blk = t / f;
                    // forward half sampled at the END of each block, backward half at its START (model.py:223-227)
                    const int ts = cc < H ? blk * f + f - 1 : blk * f;
                    v = p.s[i].o[((long)b * TP + ts + HALO) * (2 * H) + cc];
                }
            }
        }
        row[c] = v;
    }
}

// The decoder input repeats in blocks of f frames (every code is up-sampled by the same f, the speaker row is constant): its COMPACT
// form has one row per block, xc [B][T / f][ld].  grid = (T / f, B), block 256
__global__ __launch_bounds__(256) void build_dec_in_compact_kernel(CodeSrcPack p, const float* __restrict__ emb, int emb_dim,
                                                                   int emb_col, float* __restrict__ xc, int ld, int T, int f) {
    const int blk = blockIdx.x, b = blockIdx.y;
    const int TP = T + 2 * HALO;
    float* row = xc + ((long)b * (T / f) + blk) * ld;
    for (int c = threadIdx.x; c < ld; c += 256) {
        float v = 0.f;
        if (c >= emb_col && c < emb_col + emb_dim) {
            v = emb[(long)b * emb_dim + c - emb_col];
        } else {
            for (int i = 0; i < p.n; ++i) {
                const int H = p.s[i].H, col = p.s[i].col;
                if (c >= col && c < col + 2 * H) {
                    const int cc = c - col;
                    const int ts = cc < H ? blk * f + f - 1 : blk * f;      // as build_dec_in_kernel
                    v = p.s[i].o[((long)b * TP + ts + HALO) * (2 * H) + cc];
                }
            }
        }
        row[c] = v;
    }
}

// gradient of the same: d_xc [B][T / f][ld] already holds the sum over each block's frames.  grid = (T, B)
__global__ __launch_bounds__(128) void dec_in_grad_compact_kernel(CodeSrcPack p, const float* __restrict__ d_xc, int ld, int T, int f) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int TP = T + 2 * HALO;
    for (int i = 0; i < p.n; ++i) {
        const int H = p.s[i].H, col = p.s[i].col;
        float* drow = p.s[i].d_o + ((long)b * TP + t + HALO) * (2 * H);
        for (int cc = threadIdx.x; cc < 2 * H; cc += 128) {
            const bool sampled = cc < H ? (t % f == f - 1) : (t % f == 0);
            drow[cc] = sampled ? d_xc[((long)b * (T / f) + t / f) * ld + col + cc] : 0.f;
        }
    }
}

// grid = (T, B): writes the full gradient slab of every encoder BLSTM output (zeros where the code is not sampled)
__global__ __launch_bounds__(128) void dec_in_grad_kernel(CodeSrcPack p, const float* __restrict__ d_dec_in, int ld, int T) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int TP = T + 2 * HALO;
    for (int i = 0; i < p.n; ++i) {
        const int H = p.s[i].H, f = p.s[i].freq, col = p.s[i].col;
        float* drow = p.s[i].d_o + ((long)b * TP + t + HALO) * (2 * H);
        for (int cc = threadIdx.x; cc < 2 * H; cc += 128) {
            const bool sampled = cc < H ? (t % f == f - 1) : (t % f == 0);
            float v = 0.f;
            if (sampled) {
                const int t0 = (t / f) * f;
                for (int u = 0; u < f; ++u) v += d_dec_in[((long)b * TP + t0 + u + HALO) * ld + col + cc];
            }
            drow[cc] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ losses
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ out, long o_ld, long o_bs,
                                                  const float* __restrict__ tgt, long t_ld, long t_bs,
                                                  float* __restrict__ d_out, long d_ld, long d_bs, int T, int C,
                                                  float gscale, float* __restrict__ partials) {
    __shared__ float red[256];
    const int b = blockIdx.y;
    const long n = (long)T * C;
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i / C), c = (int)(i - (long)t * C);
        const float o = out[b * o_bs + t * o_ld + c];
        const float y = tgt[b * t_bs + t * t_ld + c];
        const float e = o - y;
        s += e * e;
        d_out[b * d_bs + t * d_ld + c] = e * gscale;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void finish_loss_kernel(const float* __restrict__ partials, int n, float scale,
                                                          float* __restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * (double)scale);
}

// one wavefront per (b, t) row: log-softmax over C classes.  grid = (T, B), block 64
__global__ __launch_bounds__(64) void ce_kernel(const float* __restrict__ logits, long o_ld, long o_bs,
                                                const int* __restrict__ tgt, float* __restrict__ d_out, long d_ld,
                                                long d_bs, int T, int C, float gscale, float* __restrict__ partials) {
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const float* row = logits + b * o_bs + t * o_ld;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(row[c] - mx);
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    const int y = tgt[(long)b * T + t];
    const float lse = mx + logf(se);
    float* drow = d_out + b * d_bs + t * d_ld;
    for (int c = lane; c < C; c += 64) drow[c] = (expf(row[c] - lse) - (c == y ? 1.f : 0.f)) * gscale;
    if (lane == 0) partials[(long)b * T + t] = lse - row[y];
}

// ------------------------------------------------------------------------------------------------ Adam
// torch.optim.Adam defaults as solver.py:62 constructs it (no amsgrad, no weight decay), single-tensor formulas:
//   m = lerp(m, g, 1-b1); v = v*b2 + (1-b2) g*g; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_prepare_kernel(AdamState* st, unsigned* sticky, const float* status) {
    const unsigned mine = sticky ? __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
    const bool remote = status && *status != 0.f;
    if (mine || remote) {          // this step's gradients are garbage somewhere: leave parameters, moments and step alone
        st->skip = 1u;
        if (remote && sticky && !(mine & SS_STICKY_ABORT)) __hip_atomic_fetch_or(sticky, SS_STICKY_REMOTE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    st->skip = 0u;
    st->step += 1;
    const double t = (double)st->step;
    const double bc1 = 1.0 - pow(st->beta1, t);
    const double bc2 = 1.0 - pow(st->beta2, t);
    st->step_size = (float)(st->lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
    st->f_beta1 = (float)st->beta1;
    st->f_beta2 = (float)st->beta2;
    st->f_eps = (float)st->eps;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n4,
                                                   const AdamState* __restrict__ st, float gscale) {
    if (st->skip) return;
    const float step_size = st->step_size, bc2s = st->bc2_sqrt, b1 = st->f_beta1, b2 = st->f_beta2, eps = st->f_eps;
    const float w1 = (float)(1.0 - st->beta1), w2 = (float)(1.0 - st->beta2);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = gv[j] * gscale;
            mv[j] = mv[j] + w1 * (gg - mv[j]);
            vv[j] = vv[j] * b2 + w2 * gg * gg;
            const float denom = sqrtf(vv[j]) / bc2s + eps;
            pv[j] = pv[j] - step_size * (mv[j] / denom);
        }
        (void)b1;
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
}

}  // namespace

hipError_t gn_relu_fwd(const float* x, long x_ld, long x_bs, float* y, long y_ld, long y_bs, const float* gamma,
                       const float* beta, float* stats, int B, int T, int C, hipStream_t s) {
    if (C % 64 != 0 || T > 16 * GN_MAXIT) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(C / 64, B), dim3(256), 0, s, x, x_ld, x_bs, y, y_ld, y_bs, gamma, beta, stats,
                       T, C);
    return hipGetLastError();
}

hipError_t gn_relu_gather(const float* x, long x_ld, long x_bs, float* y, long y_ld, long y_bs, float* y_img, const float* img_scale,
                          const float* gamma, const float* beta, float* stats, const InterpPlan& p, int B, int T, int C, hipStream_t s, int img_bf16) {
    if (C % 64 != 0 || T > 16 * GN_MAXIT || p.T != T || y_ld % 4 || y_bs % 4 || (((size_t)y) & 15)) return hipErrorInvalidValue;
    if (y_img && (y_ld % 8 || y_bs % 8 || (((size_t)y_img) & (img_bf16 ? 15 : 31)))) y_img = nullptr;             // image format v2: groups of eight (bf16: 16-byte fragments)
    const int lds = (T + 1) * 64 * 4;
    auto kern = T <= 128 ? gn_relu_gather_kernel<8> : (T <= 192 ? gn_relu_gather_kernel<12> : gn_relu_gather_kernel<GN_MAXIT>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(C / 64, B), dim3(256), lds, s, x, x_ld, x_bs, y, y_ld, y_bs, y_img, img_scale, gamma, beta,
                       stats, T, C, p.P, p.i0, p.lam, p.nrows, img_bf16);
    return hipGetLastError();
}

hipError_t gn_relu_bwd(const float* x, long x_ld, long x_bs, float* dy, long dy_ld, long dy_bs, const float* gamma,
                       const float* beta, const float* stats, float* g_gamma, float* g_beta, float* g_bias, float* amax, float* part,
                       int B, int T, int C, hipStream_t s, const InterpPlan* scatter, const float* src, long src_ld, long src_bs, float* dy_img) {
    if (C % 64 != 0 || T > 16 * GN_MAXIT) return hipErrorInvalidValue;
    if (dy_img && (dy_ld % 8 || dy_bs % 8 || (((size_t)dy_img) & 7))) return hipErrorInvalidValue;
    if (scatter && (!src || scatter->T != T || scatter->P > 16 * GN_MAXIT + 2 || src_ld % 4 || src_bs % 4 || (((size_t)src) & 15))) return hipErrorInvalidValue;
    if (!g_deterministic && !g_gn_part) part = nullptr;
    // (an 8-iteration instantiation for T <= 128 makes hipcc hoist every source-row load: 418 registers unbounded, spills when bounded)
    auto kern = T <= 192 ? gn_relu_bwd_kernel<12> : gn_relu_bwd_kernel<GN_MAXIT>;
    const int lds = scatter ? scatter->P * 64 * 4 : 0;                    // the utterance's source rows of the block's 64 channels
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(C / 64, B), dim3(256), lds, s, x, x_ld, x_bs, dy, dy_ld, dy_bs, gamma, beta,
                       stats, g_gamma, g_beta, g_bias, reinterpret_cast<unsigned*>(amax), part, B, T, C, scatter ? src : nullptr, src_ld, src_bs,
                       scatter ? scatter->P : 0, scatter ? scatter->lam : nullptr, scatter ? scatter->start : nullptr, dy_img);
    if (part) hipLaunchKernelGGL(gn_part_reduce_kernel, dim3(cdiv(3 * C, 256)), dim3(256), 0, s, part, B, C, g_gamma, g_beta, g_bias);
    return hipGetLastError();
}

hipError_t gn_relu_mask(const float* x, long x_ld, long x_bs, const float* gamma, const float* beta, const float* stats,
                        float* mask, int B, int T, int C, hipStream_t s) {
    int gx = cdiv((long)T * C, 256 * 4);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gn_relu_mask_kernel, dim3(gx, B), dim3(256), 0, s, x, x_ld, x_bs, gamma, beta, stats, mask, T, C);
    return hipGetLastError();
}

// scratch (nullable): >= colsum_scratch_doubles(cols) float64 words and, in ctr, cdiv(cols, 64) zeroed counters nobody else uses while the
// launch is in flight (they are zero again when it retires); without scratch every column block is ONE workgroup (slower, same sums)
long colsum_scratch_doubles(int cols) {
    const int cblocks = cdiv(cols, 64);
    return (long)(cdiv(1024, cblocks) + 1) * cblocks * 64;
}
static hipError_t colsum_launch(const float* in, long ld, int R, int C, int ncols, double* part, unsigned* ctr, float* o0, float* o1, float* o2, float* o3,
                                hipStream_t s) {
    const int cblocks = cdiv(ncols, 64);
    int chunks = (part && ctr) ? cdiv(1024, cblocks) : 1;
    if (chunks > cdiv(R, 16)) chunks = cdiv(R, 16);
    if (chunks < 1) chunks = 1;
    const int rpc = cdiv(R, chunks);
    chunks = cdiv(R, rpc);
    hipLaunchKernelGGL(colsum_kernel, dim3(cblocks, chunks), dim3(256), 0, s, in, ld, R, C, ncols, rpc, part, ctr, o0, o1, o2, o3);
    return hipGetLastError();
}
hipError_t colsum_acc(const float* in, long ld, int R, int C, float* out, double* part, unsigned* ctr, hipStream_t s) {
    return colsum_launch(in, ld, R, C, C, part, ctr, out, nullptr, nullptr, nullptr, s);
}
hipError_t colsum_bias(const float* in, long ld, int R, int C, float* bih0, float* bhh0, float* bih1, float* bhh1, double* part, unsigned* ctr,
                       hipStream_t s) {
    return colsum_launch(in, ld, R, C, 2 * C, part, ctr, bih0, bhh0, bih1, bhh1, s);
}

hipError_t copy_rows(const float* src, long s_ld, long s_bs, float* dst, long d_ld, long d_bs, int B, int T, int C,
                     hipStream_t s) {
    int gx = cdiv((long)T * C, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(gx, B), dim3(256), 0, s, src, s_ld, s_bs, dst, d_ld, d_bs, T, C);
    return hipGetLastError();
}

hipError_t collate(const float* mel_cat, const float* f0_cat, const float* emb_tab, const long* row0, const int* len,
                   const int* item, int B, int T, int C, int E, float* mel, float* f0, float* emb, hipStream_t s) {
    int gx = cdiv((long)T * C, 256 * 4);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(collate_kernel, dim3(gx, B), dim3(256), 0, s, mel_cat, f0_cat, emb_tab, row0, len, item, T, C, E, mel, f0, emb);
    return hipGetLastError();
}

hipError_t conv_pack(const float* w, int Co, int Ci, int Cp, float* wf, float* wb, float* wf_img, float* wb_img, hipStream_t s, int img_bf16) {
    if (Cp % 4 || Co % 4) return hipErrorInvalidValue;
    const long n = ((long)Co * 5 * Cp + (long)Ci * 5 * Co) / 4;
    int g = cdiv(n, 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(g), dim3(256), 0, s, w, Co, Ci, Cp, wf, wb, wf_img, wb_img, img_bf16);
    return hipGetLastError();
}

hipError_t conv_unpack_grad(const float* gp, int Co, int Ci, int Cp, float* g, hipStream_t s) {
    int gr = cdiv((long)Co * Ci * 5, 256);
    if (gr > 2048) gr = 2048;
    hipLaunchKernelGGL(conv_unpack_grad_kernel, dim3(gr), dim3(256), 0, s, gp, Co, Ci, Cp, g);
    return hipGetLastError();
}

hipError_t conv_pack_many(const ConvPackTable& tb, hipStream_t s) {
    if (tb.n <= 0) return hipSuccess;
    if (tb.n > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_pack_many_kernel, dim3(512, tb.n), dim3(256), 0, s, tb);
    return hipGetLastError();
}

hipError_t conv_unpack_grads(const ConvUnpackTable& tb, hipStream_t s) {
    if (tb.n <= 0) return hipSuccess;
    if (tb.n > CONV_UNPACK_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_unpack_grads_kernel, dim3(512, tb.n), dim3(256), 0, s, tb);
    return hipGetLastError();
}

hipError_t transpose2d(const float* in, int R, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, s, in, R, C, out);
    return hipGetLastError();
}

hipError_t add_vec(const float* a, const float* b, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(add_vec_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, a, b, out, n);
    return hipGetLastError();
}

hipError_t prep_run(const PrepTable& tb, hipStream_t s) {
    if (tb.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(prep_kernel, dim3(PREP_BLOCKS, tb.n), dim3(256), 0, s, tb);
    return hipGetLastError();
}

hipError_t build_dec_in(const CodeSrc* src, int nsrc, const float* emb, int emb_dim, int emb_col, float* dec_in, int ld,
                        int B, int T, hipStream_t s) {
    if (nsrc > 4) return hipErrorInvalidValue;
    CodeSrcPack p;
    p.n = nsrc;
    for (int i = 0; i < nsrc; ++i) p.s[i] = src[i];
    hipLaunchKernelGGL(build_dec_in_kernel, dim3(T, B), dim3(256), 0, s, p, emb, emb_dim, emb_col, dec_in, ld, T);
    return hipGetLastError();
}

hipError_t build_dec_in_compact(const CodeSrc* src, int nsrc, const float* emb, int emb_dim, int emb_col, float* xc, int ld, int B, int T,
                                int f, hipStream_t s) {
    if (nsrc > 4 || f < 1 || T % f) return hipErrorInvalidValue;
    CodeSrcPack p{};
    p.n = nsrc;
    for (int i = 0; i < nsrc; ++i) p.s[i] = src[i];
    hipLaunchKernelGGL(build_dec_in_compact_kernel, dim3(T / f, B), dim3(256), 0, s, p, emb, emb_dim, emb_col, xc, ld, T, f);
    return hipGetLastError();
}

hipError_t dec_in_grad_compact(const CodeSrc* src, int nsrc, const float* d_xc, int ld, int B, int T, int f, hipStream_t s) {
    if (nsrc > 4 || f < 1 || T % f) return hipErrorInvalidValue;
    CodeSrcPack p{};
    p.n = nsrc;
    for (int i = 0; i < nsrc; ++i) p.s[i] = src[i];
    hipLaunchKernelGGL(dec_in_grad_compact_kernel, dim3(T, B), dim3(128), 0, s, p, d_xc, ld, T, f);
    return hipGetLastError();
}

hipError_t dec_in_grad(const CodeSrc* src, int nsrc, const float* d_dec_in, int ld, int B, int T, hipStream_t s) {
    if (nsrc > 4) return hipErrorInvalidValue;
    CodeSrcPack p;
    p.n = nsrc;
    for (int i = 0; i < nsrc; ++i) p.s[i] = src[i];
    hipLaunchKernelGGL(dec_in_grad_kernel, dim3(T, B), dim3(128), 0, s, p, d_dec_in, ld, T);
    return hipGetLastError();
}

hipError_t mse_loss(const float* out, long o_ld, long o_bs, const float* tgt, long t_ld, long t_bs, float* d_out, long d_ld,
                    long d_bs, int B, int T, int C, float grad_scale, float* partials, float* loss, hipStream_t s) {
    const int gx = 8;
    const double n = (double)B * T * C;
    hipLaunchKernelGGL(mse_kernel, dim3(gx, B), dim3(256), 0, s, out, o_ld, o_bs, tgt, t_ld, t_bs, d_out, d_ld, d_bs, T, C,
                       (float)(2.0 / n) * grad_scale, partials);
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(256), 0, s, partials, gx * B, (float)(1.0 / n), loss);
    return hipGetLastError();
}

hipError_t ce_loss(const float* logits, long o_ld, long o_bs, const int* tgt, float* d_out, long d_ld, long d_bs, int B,
                   int T, int C, float grad_scale, float* partials, float* loss, hipStream_t s) {
    const double n = (double)B * T;
    hipLaunchKernelGGL(ce_kernel, dim3(T, B), dim3(64), 0, s, logits, o_ld, o_bs, tgt, d_out, d_ld, d_bs, T, C,
                       (float)(1.0 / n) * grad_scale, partials);
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(256), 0, s, partials, B * T, (float)(1.0 / n), loss);
    return hipGetLastError();
}

namespace {

__global__ void status_publish_kernel(const unsigned* sticky, float* status) {
    *status = __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void param_guard_kernel(const float* __restrict__ p, long n4, float limit, unsigned* sticky) {
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= !(fabsf(v[j]) < limit);       // also true for NaN
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(sticky, SS_STICKY_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one workgroup per conv block
__global__ __launch_bounds__(256) void act_scale_kernel(ActScaleTable tb, float rt, float* __restrict__ out) {
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    float mg = 0.f, mb = 0.f;
    for (int c = threadIdx.x; c < tb.C[b]; c += 256) {
        mg = fmaxf(mg, fabsf(tb.gamma[b][c]));
        mb = fmaxf(mb, fabsf(tb.beta[b][c]));
    }
    for (int o = 32; o > 0; o >>= 1) {
        mg = fmaxf(mg, __shfl_xor(mg, o));
        mb = fmaxf(mb, __shfl_xor(mb, o));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = mg;
        red[1][threadIdx.x >> 6] = mb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mg = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        mb = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        const float bound = rt * mg + mb;
        float sc = 16.0f;
        // (a non-finite bound keeps 16: the parameter guard reports that case)
        while (sc > 1.0f / 1048576.0f && bound * sc > 32768.0f) sc *= 0.5f;
        out[b] = sc;
    }
}

}  // namespace

hipError_t act_scales(const ActScaleTable& tb, int T, float* out, hipStream_t s) {
    if (tb.n < 1) return hipSuccess;
    hipLaunchKernelGGL(act_scale_kernel, dim3(tb.n), dim3(256), 0, s, tb, sqrtf(16.0f * (float)T), out);
    return hipGetLastError();
}

hipError_t status_publish(const unsigned* sticky, float* status, hipStream_t s) {
    hipLaunchKernelGGL(status_publish_kernel, dim3(1), dim3(1), 0, s, sticky, status);
    return hipGetLastError();
}

hipError_t param_guard(const float* p, long n, float limit, unsigned* sticky, hipStream_t s) {
    if (n % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(param_guard_kernel, dim3(512), dim3(256), 0, s, p, n / 4, limit, sticky);
    return hipGetLastError();
}

hipError_t adam_step(float* p, const float* g, float* m, float* v, long n, AdamState* st, float grad_scale, unsigned* sticky,
                     const float* status, hipStream_t s) {
    if (n % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, s, st, sticky, status);
    return adam_range(p, g, m, v, n, st, grad_scale, s);
}

// the update of one range of the arenas under the step state an earlier adam_prepare set (a step may update its ranges as their
// gradients become final)
hipError_t adam_prepare(AdamState* st, unsigned* sticky, const float* status, hipStream_t s) {
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, s, st, sticky, status);
    return hipGetLastError();
}
hipError_t adam_range(float* p, const float* g, float* m, float* v, long n, AdamState* st, float grad_scale, hipStream_t s) {
    if (n % 4 != 0 || (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15)) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, s, p, g, m, v, n / 4, st, grad_scale);
    return hipGetLastError();
}

}  // namespace ss
