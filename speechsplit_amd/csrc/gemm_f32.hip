// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD).
//
//   C[b][m][n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n])
//
// Every contraction on the SpeechSplit path goes through this kernel: the k=5 convolutions (as a GEMM over
// overlapping rows of a zero-haloed [T+4, C] slab, see Operand in common.h), the LSTM input projections, the
// linear head, and all weight / input gradients.  Operands may be reduction-major ("T") so the same kernel
// serves C = A.B^T, C = A.B and C = A^T.B without materialising transposes.
//
// Tile: BM x BN x 16, 256 threads = 4 waves in a 2x2 grid, each wave (BM/2)x(BN/2) as 32x32 MFMA tiles.
// LDS image is reduction-major ([k][m] and [k][n]) for both operands so a fragment read is one ds_read_b32 per
// lane with the 32 lanes of a half-wave on consecutive banks.  Global -> register -> LDS staging with the next
// tile's loads issued before the MFMAs of the current one (two LDS buffers, one barrier per k-tile).
#include "common.h"

namespace ss {

namespace {

constexpr int BK = 16;
constexpr int PADL = 4;

template <bool VEC>
__device__ __forceinline__ f32x4 load4(const Operand& op, const float* base, int row, int col, int R, int C) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= R || col >= C) return v;
    const long off = (long)row * op.ld;
    if (VEC) {
        if (col + 3 < C) {
            const long o = off + (op.seglen ? (long)(col / op.seglen) * op.segstride + (col % op.seglen) : (long)col);
            return *reinterpret_cast<const f32x4*>(base + o);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = col + j;
        if (c < C) {
            const long o = off + (op.seglen ? (long)(c / op.seglen) * op.segstride + (c % op.seglen) : (long)c);
            v[j] = base[o];
        }
    }
    return v;
}

template <int BM, int BN, bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmDesc d) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + PADL];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + PADL];
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int batch = blockIdx.z / d.ksplit;
    const int ks = blockIdx.z - batch * d.ksplit;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    const int ktiles = (d.K + BK - 1) / BK;
    const int tiles_per_split = (ktiles + d.ksplit - 1) / d.ksplit;
    const int kbeg = ks * tiles_per_split * BK;
    int kend = kbeg + tiles_per_split * BK;
    if (kend > d.K) kend = d.K;
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    const float* Ab = d.A.p + (long)batch * d.A.bstride;
    const float* Bb = d.B.p + (long)batch * d.B.bstride;

    f32x4 ra[NA], rb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256;
            if (!TA) ra[i] = load4<VEC>(d.A, Ab, m0 + (f >> 2), k0 + (f & 3) * 4, d.M, kend);
            else     ra[i] = load4<VEC>(d.A, Ab, k0 + f / (BM / 4), m0 + (f % (BM / 4)) * 4, kend, d.M);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256;
            if (!TB) rb[i] = load4<VEC>(d.B, Bb, n0 + (f >> 2), k0 + (f & 3) * 4, d.N, kend);
            else     rb[i] = load4<VEC>(d.B, Bb, k0 + f / (BN / 4), n0 + (f % (BN / 4)) * 4, kend, d.N);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256;
            if (!TA) {
                const int m = f >> 2, kq = (f & 3) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) As[buf][kq + j][m] = ra[i][j];
            } else {
                *reinterpret_cast<f32x4*>(&As[buf][f / (BM / 4)][(f % (BM / 4)) * 4]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256;
            if (!TB) {
                const int n = f >> 2, kq = (f & 3) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[buf][kq + j][n] = rb[i][j];
            } else {
                *reinterpret_cast<f32x4*>(&Bs[buf][f / (BN / 4)][(f % (BN / 4)) * 4]) = rb[i];
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (nk > 0) {
        gload(kbeg);
        sstore(0);
    }
    __syncthreads();
    const int kh = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = As[buf][kk + kh][wm * (BM / 2) + mi * 32 + l31];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = Bs[buf][kk + kh][wn * (BN / 2) + ni * 32 + l31];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    float* Cb = d.C + (long)batch * d.cstride;
    const bool add_bias = d.bias != nullptr && ks == 0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (BN / 2) + ni * 32 + l31;
            if (n >= d.N) continue;
            const float bv = add_bias ? d.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m >= d.M) continue;
                float* c = Cb + (long)m * d.ldc + n;
                const float v = acc[mi][ni][r] + bv;
                if (d.ksplit > 1) atomicAdd(c, v);
                else if (d.flags & GEMM_ACCUM) *c += v;
                else *c = v;
            }
        }
}

bool vec_ok(const Operand& o) {
    return (((uintptr_t)o.p) & 15) == 0 && (o.ld & 3) == 0 && (o.bstride & 3) == 0 && (o.seglen & 3) == 0 &&
           (o.segstride & 3) == 0;
}

template <int BM, int BN, bool TA, bool TB>
hipError_t launch_cfg(const GemmDesc& d, bool vec, hipStream_t s) {
    dim3 grid(cdiv(d.N, BN), cdiv(d.M, BM), d.batch * d.ksplit);
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, true>), grid, dim3(256), 0, s, d);
    else     hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, false>), grid, dim3(256), 0, s, d);
    return hipGetLastError();
}

template <bool TA, bool TB>
hipError_t launch_layout(const GemmDesc& d, bool vec, hipStream_t s) {
    // Largest tile that still gives every CU work; the 64x64 tile otherwise.
    auto tiles = [&](int bm, int bn) { return (long)cdiv(d.M, bm) * cdiv(d.N, bn) * d.batch * d.ksplit; };
    const long want = 256;
    if (d.N > 64 && d.M > 64 && tiles(128, 128) >= want) return launch_cfg<128, 128, TA, TB>(d, vec, s);
    if (d.M > 64 && tiles(128, 64) >= want) return launch_cfg<128, 64, TA, TB>(d, vec, s);
    return launch_cfg<64, 64, TA, TB>(d, vec, s);
}

}  // namespace

hipError_t launch_gemm(const GemmDesc& din, hipStream_t s) {
    GemmDesc d = din;
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return hipSuccess;
    if (d.ksplit < 1) d.ksplit = 1;
    if (d.ksplit > 1 && !(d.flags & GEMM_ACCUM)) return hipErrorInvalidValue;   // split-K needs a zeroed / live C
    const bool vec = vec_ok(d.A) && vec_ok(d.B);
    const bool ta = d.flags & GEMM_TA, tb = d.flags & GEMM_TB;
    if (!ta && !tb) return launch_layout<false, false>(d, vec, s);
    if (!ta && tb) return launch_layout<false, true>(d, vec, s);
    if (ta && tb) return launch_layout<true, true>(d, vec, s);
    return hipErrorInvalidValue;   // (TA, !TB) is not used on this path
}

}  // namespace ss
