// fp32-grade GEMM over PRE-SPLIT operands ("planes"): C[M,N] (+)= A(m,k) . B(n,k), both operands K-contiguous and already
// stored as the two fp16 pieces of the fp16 x 2 split (gemm_bf16x3.hip): plane h = fp16(s*x), plane l = fp16(s*x - h), s a
// power of two.  Three v_mfma_f32_32x32x16_f16 per k-step (h.l, l.h, h.h), fp32 accumulation, result scaled back by 1/(sa*sb).
//
// Why a second GEMM.  gemm_bf16x3_kernel splits fp32 tiles inside its k-loop: per MFMA it pays VALU split work, register
// staging and (at its 64 x 64 wave tiles) 0.67 KB of LDS fragment reads -- it sits at the ceiling of that structure (~830 TF
// of executed MFMA rate, profiles/r01/gemm_ablation.txt).  Here the split is done ONCE per operand by an HBM-bound pass (or by
// the producer's epilogue), so the k-loop is nothing but LDS-DMA + ds_read + MFMA:
//   * 256 x 256 x 32 block tile, 512 threads = 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = 4 x 2 MFMA tiles: 12 fragment reads
//     per 24 MFMAs (0.5 KB per MFMA), 128 accumulator registers, two waves per SIMD;
//   * tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no VALU): 8 per thread and k-tile; the LDS image is
//     [row][4 x 16 B] per plane with the chunk index XOR-swizzled by (row >> 2) & 3 -- applied on the per-lane SOURCE address,
//     since an LDS-DMA writes lane-linear (cdna_hip_programming.md section 5 caveat); fragment reads are conflict-free b128;
//   * two LDS buffers (2 x 64 KB): the DMA of tile k+1 is in flight while tile k is multiplied; ONE barrier per k-tile.
// Split-K with fp32 atomics as in the other kernels.  Requirements: K % 32 == 0 (the split pass zero-pads), row strides % 8 == 0.
#include "common.h"
#include "kernels.h"

namespace ss {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PBM = 256, PBN = 256, PBK = 32;
constexpr int PLANE_A = PBM * 64, PLANE_B = PBN * 64;                 // bytes per plane and stage
constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;                      // 64 KB

__device__ __forceinline__ int sw_off(int row, int kc) { return row * 64 + ((kc ^ ((row >> 2) & 3)) << 4); }

__global__ __launch_bounds__(512) void gemm_planes_nt_kernel(const PlanesDesc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [2 stages][Ah | Al | Bh | Bl]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                                    // 2 x 4 waves
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {   // XCD-aware order: consecutive tiles of one XCD share panels in its L2 (see gemm_bf16x3.hip)
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        if ((total & 7) == 0) {
            const int lin = bx + gx * (by + gy * bz), chunk = total >> 3;
            const int rem = (lin & 7) * chunk + (lin >> 3);
            bx = rem % gx;
            by = (rem / gx) % gy;
            bz = rem / (gx * gy);
        }
    }
    const int batch = bz / d.ksplit, ks = bz - batch * d.ksplit;
    const int m0 = by * PBM, n0 = bx * PBN;
    const int ktiles = d.K / PBK;
    const int per = (ktiles + d.ksplit - 1) / d.ksplit;
    const int kt0 = ks * per;
    int kt1 = kt0 + per;
    if (kt1 > ktiles) kt1 = ktiles;
    const int nk = kt1 - kt0;
    if (nk <= 0) return;

    // ---- LDS-DMA addressing: thread t moves chunks c = j*512 + t (j = 0, 1) of every plane; chunk c = (row c/4, slot c%4)
    // holds logical k-chunk slot ^ ((row >> 2) & 3).  Rows past the matrix edge are clamped (their products are never stored).
    const _Float16* src[8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * 512 + tid, row = c >> 2, kc = (c & 3) ^ ((row >> 2) & 3);
        int ra = m0 + row, rb = n0 + row;
        ra = ra < d.M ? ra : d.M - 1;
        rb = rb < d.N ? rb : d.N - 1;
        const long oa = (long)batch * d.a_bs + (long)ra * d.lda + (long)kt0 * PBK + kc * 8;
        const long ob = (long)batch * d.b_bs + (long)rb * d.ldb + (long)kt0 * PBK + kc * 8;
        src[j * 4 + 0] = d.ah + oa;
        src[j * 4 + 1] = d.al + oa;
        src[j * 4 + 2] = d.bh + ob;
        src[j * 4 + 3] = d.bl + ob;
    }
    auto issue = [&](int stage) {
        unsigned char* base = smem + stage * STAGE + (wave * 64) * 16;          // wave-uniform; the DMA adds lane * 16
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            typedef __attribute__((address_space(3))) void* lds_t;
            unsigned char* b = base + j * 512 * 16;
            __builtin_amdgcn_global_load_lds((const void*)src[j * 4 + 0], (lds_t)(b), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)src[j * 4 + 1], (lds_t)(b + PLANE_A), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)src[j * 4 + 2], (lds_t)(b + 2 * PLANE_A), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)src[j * 4 + 3], (lds_t)(b + 2 * PLANE_A + PLANE_B), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) src[i] += PBK;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int l31 = lane & 31, kg = lane >> 5;
    int oa[4], ob[2];                     // fragment byte offsets of k16-step 0 (step 1 flips chunk bit 1: XOR 32 B)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) oa[mi] = sw_off(wm * 128 + mi * 32 + l31, kg);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) ob[ni] = sw_off(wn * 64 + ni * 32 + l31, kg);

    // fragments of one k16-step: 8 A (4 tiles x 2 planes) + 4 B reads of 16 B per lane
    struct Frags {
        f16x8 ah[4], al[4], bh[2], bl[2];
    };
    auto load_frags = [&](const unsigned char* st, int k16, Frags& f) {
        const int x = k16 << 5;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            f.bh[ni] = *reinterpret_cast<const f16x8*>(st + 2 * PLANE_A + (ob[ni] ^ x));
            f.bl[ni] = *reinterpret_cast<const f16x8*>(st + 2 * PLANE_A + PLANE_B + (ob[ni] ^ x));
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            f.ah[mi] = *reinterpret_cast<const f16x8*>(st + (oa[mi] ^ x));
            f.al[mi] = *reinterpret_cast<const f16x8*>(st + PLANE_A + (oa[mi] ^ x));
        }
    };
    // the three products of a k16-step, product-major: consecutive MFMAs never touch the same accumulator (a dependent MFMA
    // cannot issue before the previous one has drained); per accumulator the order is h.l, l.h, h.h as everywhere else
    auto multiply = [&](const Frags& f) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                  // every wave's DMA of tile kt has landed (vmcnt(0) + barrier); nobody reads the other stage any more
#ifdef SS_DIAG
        if (kt + 1 < nk && !(d.diag & 1)) issue((kt + 1) & 1);       // diag 1 (wrong results): no DMA in the loop -> compute-only rate
#else
        if (kt + 1 < nk) issue((kt + 1) & 1);
#endif
        const unsigned char* st = smem + (kt & 1) * STAGE;
        Frags f0, f1;
        load_frags(st, 0, f0);
        load_frags(st, 1, f1);            // in flight under the first step's MFMAs
#ifdef SS_DIAG
        if (d.diag & 2) {                 // diag 2 (wrong results): no MFMAs -> DMA + fragment-read rate
            acc[0][0][0] += (float)f0.ah[0][0] + (float)f1.al[3][1] + (float)f0.bh[1][2] + (float)f1.bl[0][3];
            continue;
        }
#endif
        multiply(f0);
        multiply(f1);
    }

    float* Cb = d.c + (long)batch * d.c_bs;
    float unscale = d.unscale;
    if (d.amax_a) unscale /= pow2_scale_of(*d.amax_a);
    if (d.amax_b) unscale /= pow2_scale_of(*d.amax_b);
    const bool add_bias = d.bias != nullptr && ks == 0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wn * 64 + ni * 32 + l31;
            if (n >= d.N) continue;
            const float bv = add_bias ? d.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (m >= d.M) continue;
                if (d.row_period && (m % d.row_period < d.row_lo || m % d.row_period >= d.row_hi)) continue;     // halo rows of a slab stay zero
                float* c = Cb + (long)m * d.ldc + n;
                const float v = acc[mi][ni][r] * unscale + bv;
                if (d.ksplit > 1) atomicAdd(c, v);
                else if (d.accumulate) *c += v;
                else *c = v;
            }
        }
}

// ---- split passes --------------------------------------------------------------------------------------------------------
// rows x cols fp32 (row stride ld) -> two fp16 planes [rows][ldp] (ldp >= round-up of cols to 32, zero-filled beyond cols)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, long ld, int rows, int cols, const float* amax, float fixed_scale,
                                                           _Float16* __restrict__ ph, _Float16* __restrict__ pl, long ldp, int colsp) {
    const float s = amax ? pow2_scale_of(*amax) : fixed_scale;
    const long n8 = (long)rows * (colsp >> 3);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const int r = (int)(i / (colsp >> 3)), c = (int)(i - (long)r * (colsp >> 3)) * 8;
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = (c + j < cols) ? src[(long)r * ld + c + j] * s : 0.f;
            const _Float16 hh = (_Float16)y;
            h[j] = hh;
            l[j] = (_Float16)(y - (float)hh);
        }
        *reinterpret_cast<f16x8*>(ph + (long)r * ldp + c) = h;
        *reinterpret_cast<f16x8*>(pl + (long)r * ldp + c) = l;
    }
}

// transposing: src [rows][cols] fp32 -> planes [cols][ldp] (ldp >= round-up of rows to 32), 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ src, long ld, int rows, int cols, const float* amax, float fixed_scale,
                                                             _Float16* __restrict__ ph, _Float16* __restrict__ pl, long ldp, int rowsp) {
    __shared__ float tile[64][65];
    const float s = amax ? pow2_scale_of(*amax) : fixed_scale;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;                     // 64 x 4
    for (int j = ty; j < 64; j += 4) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? src[(long)r * ld + c] * s : 0.f;
    }
    __syncthreads();
    // thread -> (output row = source column c0 + oc, 8 consecutive source rows)
    const int oc = threadIdx.x >> 3, seg = (threadIdx.x & 7) * 8;
    for (int pass = 0; pass < 2; ++pass) {
        const int cc = oc + pass * 32;
        if (c0 + cc >= cols) continue;
        if (r0 + seg >= rowsp) continue;
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = tile[seg + j][cc];
            const _Float16 hh = (_Float16)y;
            h[j] = hh;
            l[j] = (_Float16)(y - (float)hh);
        }
        *reinterpret_cast<f16x8*>(ph + (long)(c0 + cc) * ldp + r0 + seg) = h;
        *reinterpret_cast<f16x8*>(pl + (long)(c0 + cc) * ldp + r0 + seg) = l;
    }
}

}  // namespace

hipError_t launch_gemm_planes(const PlanesDesc& din, hipStream_t s) {
    PlanesDesc d = din;
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return hipSuccess;
    if (d.K % PBK || d.lda % 8 || d.ldb % 8 || d.a_bs % 8 || d.b_bs % 8) return hipErrorInvalidValue;
    if (d.ksplit < 1) d.ksplit = 1;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_planes_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    dim3 grid(cdiv(d.N, PBN), cdiv(d.M, PBM), d.batch * d.ksplit);
    hipLaunchKernelGGL(gemm_planes_nt_kernel, grid, dim3(512), 2 * STAGE, s, d);
    return hipGetLastError();
}

hipError_t split_planes(const float* src, long ld, int rows, int cols, const float* amax, float fixed_scale, void* ph, void* pl, long ldp,
                        hipStream_t s) {
    const int colsp = (cols + 31) & ~31;
    if (ldp < colsp || ldp % 8) return hipErrorInvalidValue;
    const long n8 = (long)rows * (colsp >> 3);
    int g = cdiv(n8, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(split_planes_kernel, dim3(g), dim3(256), 0, s, src, ld, rows, cols, amax, fixed_scale, (_Float16*)ph, (_Float16*)pl, ldp, colsp);
    return hipGetLastError();
}

hipError_t split_planes_t(const float* src, long ld, int rows, int cols, const float* amax, float fixed_scale, void* ph, void* pl, long ldp,
                          hipStream_t s) {
    const int rowsp = (rows + 31) & ~31;
    if (ldp < rowsp || ldp % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(split_planes_t_kernel, dim3(cdiv(cols, 64), cdiv(rowsp, 64)), dim3(256), 0, s, src, ld, rows, cols, amax, fixed_scale,
                       (_Float16*)ph, (_Float16*)pl, ldp, rowsp);
    return hipGetLastError();
}

}  // namespace ss
