// Random-resampling bottleneck (reference InterpLnr, model.py:355-436) and F0 quantiser (utils.py:62-74).
//
// The reference builds the index set with ~25 small ATen ops, a device->host sync (counts.tolist(), model.py:432)
// and a Python loop over the batch (model.py:373-375).  Here one wavefront per utterance evaluates the 7 x 64
// candidate positions, compacts them with wave ballots and emits (i0, lambda, row count) plus the inverse map the
// collision-free backward needs; gather and scatter are row-per-workgroup streaming kernels.  The random draws are
// inputs (the host draws them with the same generator calls as the reference), so the index path is bit-exact.
//
// fp32 rules that make the value path bit-exact against the reference's three separate ATen ops:
// IEEE division (no fast-math), and (1-l)*a + l*b evaluated as mul, mul, add with no FMA contraction.
#include "common.h"
#include "kernels.h"

namespace ss {

namespace {

constexpr int MAXP = 512;   // max_len_pad supported by the per-utterance LDS staging

// grid = B, block = 64 (one wavefront)
__global__ __launch_bounds__(64) void interp_plan_kernel(const float* __restrict__ scales, const int* __restrict__ len_seg,
                                                         const int* __restrict__ len_seq, int len_seq_const, int S,
                                                         int ncand, int P, int T, int* __restrict__ i0,
                                                         float* __restrict__ lam, int* __restrict__ nrows,
                                                         int* __restrict__ counts, int* __restrict__ start) {
    __shared__ int s_i0[MAXP];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int len = len_seq ? len_seq[b] : len_seq_const;
    int base = 0, offset = 0;
    for (int s = 0; s < S; ++s) {
        const float sc = scales[b * S + s];
        const int ls = len_seg[b * S + s];
        const float q = __fdiv_rn((float)lane, sc);          // model.py:395 (int64 / fp32 -> fp32 true divide)
        const float fl = floorf(q);                           // :396
        const float lm = __fsub_rn(q, fl);                    // :397
        const int ifl = (int)fl;
        const bool valid = lane < ncand && ifl < ls - 1 && ifl + offset < len - 1;   // :405, :414, :416
        const unsigned long long m = __ballot(valid);
        const int r = base + __popcll(m & ((1ull << lane) - 1ull));
        if (valid && r < P) {
            i0[(long)b * P + r] = ifl + offset;               // :411, :423
            lam[(long)b * P + r] = lm;
            s_i0[r] = ifl + offset;
        }
        base += __popcll(m);
        offset += ls;                                         // :407-409 exclusive cumsum
    }
    const int n = base < P ? base : P;                        // pad_sequences truncation, :375
    for (int r = n + lane; r < P; r += 64) {
        i0[(long)b * P + r] = 0;
        lam[(long)b * P + r] = 0.f;
    }
    if (lane == 0) {
        nrows[b] = n;
        counts[b] = base;                                     // :418
    }
    __syncthreads();
    // inverse map: start[i] = number of output rows r < n with i0[r] < i  (i0 is non-decreasing in r)
    for (int i = lane; i <= T; i += 64) {
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_i0[mid] < i) lo = mid + 1; else hi = mid;
        }
        start[(long)b * (T + 1) + i] = lo;
    }
}

// y[b,r,:] = (1-lam)*x[b,i0,:] + lam*x[b,i0+1,:]  for r < nrows[b], else 0.      grid = (P, B)
__global__ __launch_bounds__(256) void interp_gather_kernel(const float* __restrict__ x, long x_ld, long x_bs,
                                                            float* __restrict__ y, long y_ld, long y_bs, int C, int P,
                                                            const int* __restrict__ i0, const float* __restrict__ lam,
                                                            const int* __restrict__ nrows, float* __restrict__ y_img, const float* __restrict__ img_scale) {
    // y_img (nullable, launcher: only with C % 8 == 0 and 32-byte aligned rows): the pre-split image of y for the GEMMs that read it
    const int r = blockIdx.x, b = blockIdx.y;
    float* yr = y + b * y_bs + r * y_ld;
    float* yi = y_img ? y_img + b * y_bs + r * y_ld : nullptr;
    if (r >= nrows[b]) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) yr[c] = 0.f;
        if (yi)
            for (int c = threadIdx.x; c < C; c += blockDim.x) yi[c] = 0.f;
        return;
    }
    const int i = i0[(long)b * P + r];
    const float l = lam[(long)b * P + r];
    const float ol = __fsub_rn(1.0f, l);
    const float* xa = x + b * x_bs + (long)i * x_ld;
    const float* xb = xa + x_ld;
    if (yi) {
        const float isc = img_scale ? *img_scale : 16.0f;
        for (int c = 4 * threadIdx.x; c < C; c += 4 * blockDim.x) {
            const float4 a = *reinterpret_cast<const float4*>(xa + c), bb = *reinterpret_cast<const float4*>(xb + c);
            float4 v;
            v.x = ss_lerp_rn(ol, a.x, l, bb.x);      // model.py:430, the same three roundings as below
            v.y = ss_lerp_rn(ol, a.y, l, bb.y);
            v.z = ss_lerp_rn(ol, a.z, l, bb.z);
            v.w = ss_lerp_rn(ol, a.w, l, bb.w);
            *reinterpret_cast<float4*>(yr + c) = v;
            ss_store_group(yi + c, ss_split_group_s(v.x, v.y, v.z, v.w, isc));
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        yr[c] = ss_lerp_rn(ol, xa[c], l, xb[c]);      // model.py:430
}

// Outer call of the training step (solver.py:160-163): x = [mel(80) | f0(1)], resample, quantise the f0 channel,
// emit mel rows into the encoder's haloed input slab and the one-hot rows into the (channel-padded) f0 slab.
// grid = (P, B), block = 128
__global__ __launch_bounds__(128) void interp_quant_kernel(const float* __restrict__ mel, const float* __restrict__ f0,
                                                           int T, int CM, float* __restrict__ ymel, long ym_ld, long ym_bs,
                                                           float* __restrict__ yoh, long yo_ld, long yo_bs, int NOH,
                                                           int* __restrict__ qidx, int P, const int* __restrict__ i0,
                                                           const float* __restrict__ lam, const int* __restrict__ nrows) {
    const int r = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    __shared__ int s_q;
    float* ym = ymel + b * ym_bs + r * ym_ld;
    float* yo = yoh + b * yo_bs + r * yo_ld;
    const bool live = r < nrows[b];
    float f = 0.f;
    if (live) {
        const int i = i0[(long)b * P + r];
        const float l = lam[(long)b * P + r];
        const float ol = __fsub_rn(1.0f, l);
        const float* ma = mel + ((long)b * T + i) * CM;
        for (int c = tid; c < CM; c += 128) ym[c] = ss_lerp_rn(ol, ma[c], l, ma[CM + c]);
        if (tid == 0) f = ss_lerp_rn(ol, f0[(long)b * T + i], l, f0[(long)b * T + i + 1]);
    } else {
        for (int c = tid; c < CM; c += 128) ym[c] = 0.f;
    }
    if (tid == 0) {
        // utils.py:66-71: uv = x <= 0 -> class 0; else round-half-even(x * 255) + 1
        int q = 0;
        if (f > 0.f) q = (int)rintf(__fmul_rn(f, 255.0f)) + 1;
        if (q > NOH - 1) q = NOH - 1;      // the reference asserts x <= 1 (utils.py:68); clamp instead of faulting
        s_q = q;
        qidx[(long)b * P + r] = q;
    }
    __syncthreads();
    const int q = s_q;
    for (int c = tid; c < yo_ld; c += 128) yo[c] = (c == q) ? 1.f : 0.f;
}

// Adjoint of the gather with respect to x, collision-free through the inverse map:
//   dx[b,i,:] = sum_{r: i0[r]==i} (1-lam_r) dy[b,r,:] + sum_{r: i0[r]==i-1} lam_r dy[b,r,:]          grid = (T, B)
__global__ __launch_bounds__(256) void interp_scatter_kernel(const float* __restrict__ dy, long dy_ld, long dy_bs,
                                                             float* __restrict__ dx, long dx_ld, long dx_bs, int C, int P,
                                                             int T, const float* __restrict__ lam,
                                                             const int* __restrict__ start) {
    const int i = blockIdx.x, b = blockIdx.y;
    const int* st = start + (long)b * (T + 1);
    const int a0 = st[i], a1 = st[i + 1];
    const int b0 = i > 0 ? st[i - 1] : 0, b1 = i > 0 ? a0 : 0;
    const float* lm = lam + (long)b * P;
    const float* dyb = dy + b * dy_bs;
    float* dxr = dx + b * dx_bs + (long)i * dx_ld;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int r = a0; r < a1; ++r) acc = __builtin_fmaf(1.0f - lm[r], dyb[r * dy_ld + c], acc);      // (the fused form in gn_relu_bwd_kernel: same terms, same order, same fma)
        for (int r = b0; r < b1; ++r) acc = __builtin_fmaf(lm[r], dyb[r * dy_ld + c], acc);
        dxr[c] = acc;
    }
}

}  // namespace

hipError_t interp_plan(const InterpPlan& p, const float* scales, const int* len_seg, const int* len_seq,
                       int len_seq_const, int B, hipStream_t s) {
    if (p.P > MAXP || p.ncand > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(interp_plan_kernel, dim3(B), dim3(64), 0, s, scales, len_seg, len_seq, len_seq_const, p.S, p.ncand,
                       p.P, p.T, p.i0, p.lam, p.nrows, p.counts, p.start);
    return hipGetLastError();
}

hipError_t interp_gather(const InterpPlan& p, const float* x, long x_ld, long x_bs, float* y, long y_ld, long y_bs, int C,
                         int B, hipStream_t s, float* y_img, const float* img_scale) {
    if (y_img && (C % 8 || x_ld % 4 || y_ld % 8 || x_bs % 4 || y_bs % 8 || (((size_t)x | (size_t)y) & 15) || (((size_t)y_img) & 31))) y_img = nullptr;     // image format v2: groups of eight
    const int threads = y_img ? (C >= 512 ? 128 : 64) : (C >= 256 ? 256 : (C >= 128 ? 128 : 64));
    hipLaunchKernelGGL(interp_gather_kernel, dim3(p.P, B), dim3(threads), 0, s, x, x_ld, x_bs, y, y_ld, y_bs, C, p.P, p.i0,
                       p.lam, p.nrows, y_img, img_scale);
    return hipGetLastError();
}

hipError_t interp_quant(const InterpPlan& p, const float* mel, const float* f0, int CM, float* ymel, long ym_ld, long ym_bs,
                        float* yoh, long yo_ld, long yo_bs, int NOH, int* qidx, int B, hipStream_t s) {
    hipLaunchKernelGGL(interp_quant_kernel, dim3(p.P, B), dim3(128), 0, s, mel, f0, p.T, CM, ymel, ym_ld, ym_bs, yoh, yo_ld,
                       yo_bs, NOH, qidx, p.P, p.i0, p.lam, p.nrows);
    return hipGetLastError();
}

hipError_t interp_scatter(const InterpPlan& p, const float* dy, long dy_ld, long dy_bs, float* dx, long dx_ld, long dx_bs,
                          int C, int B, hipStream_t s) {
    const int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    hipLaunchKernelGGL(interp_scatter_kernel, dim3(p.T, B), dim3(threads), 0, s, dy, dy_ld, dy_bs, dx, dx_ld, dx_bs, C, p.P,
                       p.T, p.lam, p.start);
    return hipGetLastError();
}

}  // namespace ss
