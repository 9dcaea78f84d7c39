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

int g_gemm_bk = 16;        // (unused: BK = 32 measured slower; kept so ss_tune("gemm_bk") stays valid)
int g_gemm_want = 1024;
int g_deterministic = 0;   // ss_tune("deterministic", 1): run-to-run bit-identical results -- split-K only through ordered partial slabs (fp32 atomics commit in arrival
                           // order), ordered bias / affine gradient sums (elementwise.hip), bias gradients of the persistent recurrence
                           // through the ordered column sum.  Costs the weight-gradient GEMMs their split-K parallelism.
int g_gemm_diag = 0;       // A/B experiments: bit 0 = XCD-aware tile order off, bit 1 = two-tile register prefetch
                          // (measured slower: 146 VGPRs cost a resident workgroup per CU)     // minimum number of tiles before the largest tile is chosen, ss_tune("gemm_want")

namespace {

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

template <int BM, int BN, int BK, bool TA, bool TB, bool VEC, int PF>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmDesc d) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + PADL];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + PADL];
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the linear block id
    // is remapped to give every XCD a contiguous run of tiles (consecutive n-tiles of the same m-tiles share operand panels)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int total = gx * gy * gridDim.z;
#ifdef SS_DIAG
        if (!(d.diag & 1) && (total & 7) == 0) {
#else
        if ((total & 7) == 0) {
#endif
            const int lin = bx + gx * (by + gy * bz);
            const int rem = (lin & 7) * (total >> 3) + (lin >> 3);
            bx = rem % gx;
            by = (rem / gx) % gy;
            bz = rem / (gx * gy);
        }
    }
    const int batch = bz / d.ksplit;
    const int ks = bz - batch * d.ksplit;
    const int m0 = by * BM, n0 = bx * BN;

    const int ktiles = (d.K + BK - 1) / BK;   // split-K boundaries are multiples of BK
    const int tiles_per_split = (ktiles + d.ksplit - 1) / d.ksplit;
    const int kbeg = ks * tiles_per_split * BK;
    int kend = kbeg + tiles_per_split * BK;
    if (kend > d.K) kend = d.K;
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    const float* Ab = d.A.p + (long)batch * d.A.bstride;
    const float* Bb = d.B.p + (long)batch * d.B.bstride;

    // Vectorised instances walk K with per-slot pointers set up once (no divisions in the loop): a K-contiguous operand
    // advances `within` inside its K segment (conv taps) and hops by segstride at a segment end; a reduction-major
    // operand advances by BK rows.  The scalar instances (unaligned / odd strides) keep the generic addressing.
    const float* pa[NA];
    const float* pb[NB];
    int wa[NA], wb[NB];          // position inside the current K segment (K-contiguous operands)
    bool oka[NA], okb[NB];       // the fixed coordinate (row for K-contiguous, column for reduction-major) is in range
    if (VEC) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256;
            if (!TA) {
                const int r = m0 + f / (BK / 4), c = kbeg + (f % (BK / 4)) * 4;
                oka[i] = r < d.M;
                const int sg = d.A.seglen ? c / d.A.seglen : 0;
                wa[i] = d.A.seglen ? c - sg * d.A.seglen : c;
                pa[i] = Ab + (long)(oka[i] ? r : 0) * d.A.ld + (long)sg * d.A.segstride + wa[i];
            } else {
                const int c = m0 + (f % (BM / 4)) * 4;
                oka[i] = c < d.M;
                const int cc = oka[i] ? c : 0;
                const int sg = d.A.seglen ? cc / d.A.seglen : 0;
                wa[i] = c;
                pa[i] = Ab + (long)(kbeg + f / (BM / 4)) * d.A.ld + (long)sg * d.A.segstride + (d.A.seglen ? cc - sg * d.A.seglen : cc);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256;
            if (!TB) {
                const int r = n0 + f / (BK / 4), c = kbeg + (f % (BK / 4)) * 4;
                okb[i] = r < d.N;
                const int sg = d.B.seglen ? c / d.B.seglen : 0;
                wb[i] = d.B.seglen ? c - sg * d.B.seglen : c;
                pb[i] = Bb + (long)(okb[i] ? r : 0) * d.B.ld + (long)sg * d.B.segstride + wb[i];
            } else {
                const int c = n0 + (f % (BN / 4)) * 4;
                okb[i] = c < d.N;
                const int cc = okb[i] ? c : 0;
                const int sg = d.B.seglen ? cc / d.B.seglen : 0;
                wb[i] = c;
                pb[i] = Bb + (long)(kbeg + f / (BN / 4)) * d.B.ld + (long)sg * d.B.segstride + (d.B.seglen ? cc - sg * d.B.seglen : cc);
            }
        }
    }
    // one operand slot: load 4 consecutive elements (zero beyond the edges), then advance the slot to the next k-tile
    auto fetch = [&](const Operand& op, const float*& p, int& w, bool ok, bool T, int kpos /*k of this slot in this tile*/,
                     int cmax /*column limit of a reduction-major operand*/) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!T) {
            if (ok && kpos < kend) {
                if (kpos + 3 < kend) v = *reinterpret_cast<const f32x4*>(p);
                else
                    for (int j = 0; j < 4; ++j)
                        if (kpos + j < kend) v[j] = p[j];
            }
            p += BK;
            if (op.seglen) {
                w += BK;
                while (w >= op.seglen) {
                    w -= op.seglen;
                    p += op.segstride - op.seglen;
                }
            }
        } else {
            if (ok && kpos < kend) {
                if (w + 3 < cmax) v = *reinterpret_cast<const f32x4*>(p);
                else
                    for (int j = 0; j < 4; ++j)
                        if (w + j < cmax) v[j] = p[j];
            }
            p += (long)BK * op.ld;
        }
        return v;
    };
    auto gload = [&](f32x4(&ra)[NA], f32x4(&rb)[NB], int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256;
            if (VEC) ra[i] = fetch(d.A, pa[i], wa[i], oka[i], TA, TA ? k0 + f / (BM / 4) : k0 + (f % (BK / 4)) * 4, d.M);
            else if (!TA) ra[i] = load4<false>(d.A, Ab, m0 + f / (BK / 4), k0 + (f % (BK / 4)) * 4, d.M, kend);
            else ra[i] = load4<false>(d.A, Ab, k0 + f / (BM / 4), m0 + (f % (BM / 4)) * 4, kend, d.M);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256;
            if (VEC) rb[i] = fetch(d.B, pb[i], wb[i], okb[i], TB, TB ? k0 + f / (BN / 4) : k0 + (f % (BK / 4)) * 4, d.N);
            else if (!TB) rb[i] = load4<false>(d.B, Bb, n0 + f / (BK / 4), k0 + (f % (BK / 4)) * 4, d.N, kend);
            else rb[i] = load4<false>(d.B, Bb, k0 + f / (BN / 4), n0 + (f % (BN / 4)) * 4, kend, d.N);
        }
    };
    auto sstore = [&](f32x4(&ra)[NA], f32x4(&rb)[NB], int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256;
            if (!TA) {
                const int m = f / (BK / 4), kq = (f % (BK / 4)) * 4;
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
                const int n = f / (BK / 4), kq = (f % (BK / 4)) * 4;
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

    const int kh = lane >> 5, l31 = lane & 31;
    auto compute = [&](int buf) {
        // fragments of k-step kk+2 are requested before the MFMAs of k-step kk are issued
        float a[2][MI], b[2][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[0][mi] = As[buf][kh][wm * (BM / 2) + mi * 32 + l31];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[0][ni] = Bs[buf][kh][wn * (BN / 2) + ni * 32 + l31];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[cur ^ 1][mi] = As[buf][kk + 2 + kh][wm * (BM / 2) + mi * 32 + l31];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[cur ^ 1][ni] = Bs[buf][kk + 2 + kh][wn * (BN / 2) + ni * 32 + l31];
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi], b[cur][ni], acc[mi][ni], 0, 0, 0);
        }
    };

    f32x4 ra0[NA], rb0[NB];
    if (PF == 1) {
        // tile kt+1 is fetched while tile kt is multiplied
        if (nk > 0) {
            gload(ra0, rb0, kbeg);
            sstore(ra0, rb0, 0);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) gload(ra0, rb0, kbeg + (kt + 1) * BK);
            compute(buf);
            if (kt + 1 < nk) sstore(ra0, rb0, buf ^ 1);
            __syncthreads();
        }
    } else {
        // two register sets: tile kt+2 is requested while tile kt is multiplied and tile kt+1 (requested one tile
        // earlier) is written to LDS, so a global load has two full tiles of MFMA time to land
        f32x4 ra1[NA], rb1[NB];
        if (nk > 0) gload(ra0, rb0, kbeg);
        if (nk > 1) gload(ra1, rb1, kbeg + BK);
        if (nk > 0) sstore(ra0, rb0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 2 < nk) gload(ra0, rb0, kbeg + (kt + 2) * BK);
            compute(0);
            if (kt + 1 < nk) sstore(ra1, rb1, 1);
            __syncthreads();
            if (kt + 1 >= nk) break;
            if (kt + 3 < nk) gload(ra1, rb1, kbeg + (kt + 3) * BK);
            compute(1);
            if (kt + 2 < nk) sstore(ra0, rb0, 0);
            __syncthreads();
        }
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
                if (d.row_period) {
                    const int q = (m + d.row_off) % d.row_period;
                    if (q < d.row_lo || q >= d.row_hi) continue;
                }
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
    if (!vec) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 16, TA, TB, false, 1>), grid, dim3(256), 0, s, d);
    else if (g_gemm_diag & 2) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 16, TA, TB, true, 2>), grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 16, TA, TB, true, 1>), grid, dim3(256), 0, s, d);
    return hipGetLastError();
}

template <bool TA, bool TB>
hipError_t launch_layout(const GemmDesc& d, bool vec, hipStream_t s) {
    // Largest tile that still gives every CU work; the 64x64 tile otherwise.
    auto tiles = [&](int bm, int bn) { return (long)cdiv(d.M, bm) * cdiv(d.N, bn) * d.batch * d.ksplit; };
    const long want = g_gemm_want;
    if (d.N > 64 && d.M > 64 && tiles(128, 128) >= want) return launch_cfg<128, 128, TA, TB>(d, vec, s);
    if (d.M > 64 && tiles(128, 64) >= want) return launch_cfg<128, 64, TA, TB>(d, vec, s);
    return launch_cfg<64, 64, TA, TB>(d, vec, s);
}

}  // namespace

hipError_t launch_gemm_bf16x3(const GemmDesc& d, hipStream_t s);
extern int g_gemm_mode;

hipError_t launch_gemm(const GemmDesc& din, hipStream_t s) {
    GemmDesc d = din;
    d.diag = g_gemm_diag;
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return hipSuccess;
    // deterministic mode: a split reduction only through partial slabs added in a fixed order (GemmDesc::part, round 3); without scratch no split
    // (its fp32 atomics would commit in arrival order).  Round 2 dropped split-K altogether there: 2.2x the default step.
    const bool vec = vec_ok(d.A) && vec_ok(d.B);
    const bool seg_ok = (d.A.seglen == 0 || d.A.seglen >= 32) && (d.B.seglen == 0 || d.B.seglen >= 32);   // one wrap per k-tile
    const bool split_kernel = vec && seg_ok && g_gemm_mode == 1;            // the bf16 x 3 / fp16 x 2 kernel: the one with the partial-slab split-K
    if (d.ksplit < 1 || (g_deterministic && !(d.part && split_kernel))) d.ksplit = 1;
    if (d.ksplit > 1 && !(d.flags & GEMM_ACCUM)) return hipErrorInvalidValue;   // split-K needs a zeroed / live C
    if (split_kernel) return launch_gemm_bf16x3(d, s);
    d.part = nullptr;              // the fp32-MFMA kernel's split-K meets in C through atomics
    const bool ta = d.flags & GEMM_TA, tb = d.flags & GEMM_TB;
    if (!ta && !tb) return launch_layout<false, false>(d, vec, s);
    if (!ta && tb) return launch_layout<false, true>(d, vec, s);
    if (ta && tb) return launch_layout<true, true>(d, vec, s);
    return hipErrorInvalidValue;   // (TA, !TB) is not used on this path
}

}  // namespace ss
