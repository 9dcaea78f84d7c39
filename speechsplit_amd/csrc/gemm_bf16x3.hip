// fp32-accurate GEMM on the bf16 matrix cores ("bf16x3"): same contract as gemm_f32.hip (fp32 operands in HBM, fp32
// result), but each fp32 operand value is split on the fly into three bf16 pieces that represent it EXACTLY
//      x = h + m + l,   h = top 8 significand bits, m = next 8, l = last 8   (truncation splits, residuals exact in fp32)
// and the product is evaluated as  Ah.Bh + Ah.Bm + Am.Bh + Ah.Bl + Al.Bh + Am.Bm  with v_mfma_f32_32x32x16_bf16 and
// fp32 accumulation.  The dropped terms (m.l, l.m, l.l) are <= 2^-24 relative to the product, i.e. below one fp32 ulp,
// so the result is an fp32 GEMM up to summation order.  Six bf16 MFMAs cost 6/16 of the two fp32 MFMAs they replace:
// the usable matrix-core rate for fp32 data rises from 157 TF to 2.5 PF / 6 = 417 TF.
//
// Structure: 128x128x32 tile (also 128x64, 64x64), 256 threads = 2x2 waves, 2x2 (or fewer) 32x32 MFMA tiles per wave,
// one LDS buffer of three bf16 planes per operand (48 KB) with the 16-byte chunk index XOR-swizzled by (row>>2)&3 so
// the ds_read_b128 fragment reads are conflict-free, global->register prefetch of the next tile while the current one
// is multiplied, split + ds_write after the barrier.  Operand addressing (segmented K for convolutions, reduction-major
// operands, split-K with atomics, XCD-aware tile order) is shared with the fp32 kernel's conventions.
#include "common.h"

namespace ss {

extern int g_gemm_want;
int g_gemm_tr = 1;       // transposing LDS reads for reduction-major operands of 128-wide tiles: 1 wherever possible, 2 not for TN, 0 never
                         // (round 2: 1 -- since the kernels are held to three waves per SIMD the TN weight gradients gain too, step -0.08 ms)
int g_gemm_ws = 0;        // 128 x 128 fp16 x 2 tiles: 1 = the wave-specialised (512-thread) form where it measured faster in isolation, 2 = always,
                          // 0 = never (default: in the training step, where two convolutions share the chip anyway, 1 measured 0.02 ms slower)
int g_gemm_mode = 1;      // 0: exact-fp32 MFMA kernel (gemm_f32.hip), 1: bf16x3 kernel whenever operands are 16-byte aligned

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;

// timing experiments that produce wrong results (tools/kbench.py gdiag): present only in the -DSS_DIAG build, so the product
// library's k-loop carries none of their branches
#ifdef SS_DIAG
#define GDIAG(d, bits) ((d).diag & (bits))
#else
#define GDIAG(d, bits) 0
#endif

// split two fp32 values into packed bf16 pairs: (h0,h1), (m0,m1), (l0,l1); element 0 in the low half
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u);
    const float r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u);
    const float s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
    h = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    m = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// fp16 x 2 split of two fp32 values pre-scaled by a power of two: y = s*x = h + l with h = fp16(y) (nearest), l = fp16(y - h).
// 22 significand bits for values whose residual stays in fp16's normal range, an absolute floor of 2^-25 / s below it.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F16_SCALE = 16.0f;          // operands bounded by construction: |x| up to 4094 survives
// operand with a measured maximum m: the power of two that brings m into [128, 256) (256x headroom below fp16's 65504),
// capped at 2^60 so that an all-zero / denormal tensor cannot produce an infinite scale
__device__ __forceinline__ float pow2_scale(float m) {
    const int e = (int)((__float_as_uint(m) >> 23) & 0xFFu);
    int se = 261 - e;
    se = se < 1 ? 1 : (se > 187 ? 187 : se);
    return __uint_as_float((unsigned)se << 23);
}
// Four instructions per PAIR of values: the mixed-precision fma scales, subtracts the fp16 piece it reads straight out of the
// packed register, rounds once and writes the chosen half of the destination (hipcc's own code for the plain C++ form: seven).
// The split sits on the same issue port as the MFMAs -- 4 cycles per vector instruction, 8 of its 32 per MFMA -- so with 128 x 128
// tiles every instruction saved per value is ~3 % of the k-loop.
__device__ __forceinline__ void split2_f16(float x0, float x1, float scale, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(scale));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(scale), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(scale), "v"(h));
}

// two fp32 values -> packed bf16 pair, round to nearest even (element 0 in the low half)
__device__ __forceinline__ unsigned round2(float x0, float x1) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
    return r;
}

// byte offset of bf16 element (row, k) inside one plane: rows of 32 bf16 (64 B), 16-B chunks XOR-swizzled
__device__ __forceinline__ int lds_off(int row, int k) { return row * 64 + (((k >> 3) ^ ((row >> 2) & 3)) << 4) + ((k & 7) << 1); }

// NPL = 3: the exact split above (fp32-grade result).  NPL = 1 (GEMM_BF16): operands rounded once to bf16, one MFMA.
// Phase probe (ss_tune("gemm_diag", 16), 128x128 NT only): s_memtime ticks each wave spends in the phases of the k-loop,
// summed over the first 64 workgroups: [wave][0 split+store, 1 barrier, 2 load issue, 3 fragments+MFMA, 4 barrier], [0][5] = k-tiles
__device__ unsigned long long g_gemm_phase[4][6];

// Image of a reduction-major (T) operand when TR is set (128-wide tiles only): the tile is kept in its SOURCE orientation,
// [32 k-rows][128 tile rows] of 16-bit pieces, 256-byte rows with the 16-byte chunks XOR-swizzled; it is filled with one
// float4 load per slot along the contiguous axis (as for a K-contiguous operand; the other image needs four dword loads
// down the source rows per slot) and conflict-free 8-byte LDS writes, and the MFMA fragments come out of it through
// gfx950's transposing LDS read (ds_read_b64_tr_b16: per 16 lanes a block of 4 k-rows x 16 tile rows, delivered
// k-contiguous per lane), two per 8-element fragment.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int lds_off_t(int k, int x) { return k * 256 + (((x >> 3) ^ (((k & 3) << 2) | ((k >> 2) & 3))) << 4) + ((x & 7) << 1); }
// o_lo / o_hi: this lane's byte offsets of its two 4-row blocks (k .. k+3, k+4 .. k+7) inside a plane
__device__ __forceinline__ u32x4 tr_frag(const unsigned char* plane, int o_lo, int o_hi) {
    typedef __attribute__((address_space(3))) s16x4* lptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(plane + o_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(plane + o_hi));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

// WS (wave-specialised, 512 threads): waves 0-3 only multiply, waves 4-7 only stage -- global loads two k-tiles ahead, the split
// and the LDS stores of tile kt+1 while the multipliers work on tile kt out of the other of two LDS buffers; ONE LDS-only barrier
// per k-tile.  A multiplier wave and a staging wave share each SIMD: the split's VALU work issues in the shadow of the MFMAs
// instead of in front of them (in the 256-thread form every wave does both, one after the other, and a k-tile's loads have one
// MFMA phase -- ~0.4 us -- to arrive).
template <int BM, int BN, bool TA, bool TB, int NPL, bool PROBE = false, bool TR = false, bool WS = false>
__global__ __launch_bounds__(WS ? 512 : 256, (NPL == 2 && !WS && !PROBE) ? 3 : 1) void gemm_bf16x3_kernel(const GemmDesc d) {
    constexpr bool TRA = TR && TA && BM == 128, TRB = TR && TB && BN == 128;
    constexpr int MI = BM / 64, NI = BN / 64;
    // plane stride in bytes.  fp16 x 2: 64 bytes of padding between the hi and the lo plane -- a staged IMAGE slot is 16 bytes of ONE
    // piece, neighbouring lanes store hi / lo chunks of the same position, and without the shift each such pair hits the same banks
    constexpr int PA = BM * 64 + (NPL == 2 ? 64 : 0), PB = BN * 64 + (NPL == 2 ? 64 : 0);
    constexpr int NBUF = WS ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char As[NBUF * NPL * PA];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[NBUF * NPL * PB];
    // a slot = 4 consecutive k of one row of the tile.  K-contiguous operand: one float4 (8 slots per row, lanes along k);
    // reduction-major operand: four dword loads down the source rows, lanes along the tile rows (coalesced)
    constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256;

    const bool stager = WS && threadIdx.x >= 256;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int total = gx * gy * gridDim.z;
        if ((total & 7) == 0) {                               // XCD-aware tile order (see gemm_f32.hip)
            const int lin = bx + gx * (by + gy * bz);
            const int chunk = total >> 3;                     // consecutive tiles of one XCD
            int rem = (lin & 7) * chunk + (lin >> 3);
            // Inside an XCD's chunk walk 8 columns x all of the chunk's rows before the next 8 columns: the ~64 workgroups
            // resident on the XCD then share 8 column panels AND 8 row panels in its 4 MB L2, instead of streaming every
            // column panel once per pair of rows (3.4x -> 1.6x the operand bytes from HBM on the projection shape).
            constexpr int GW = 8;
            if (!GDIAG(d, 8) && chunk % gx == 0 && gx % GW == 0 && chunk / gx >= 2) {
                const int rows = chunk / gx, local = rem % chunk;
                const int c = (local / (GW * rows)) * GW + local % GW;
                const int r = (local / GW) % rows;
                rem = rem - local + r * gx + c;
            }
            bx = rem % gx;
            by = (rem / gx) % gy;
            bz = rem / (gx * gy);
        }
    }
    const int batch = bz / d.ksplit;
    const int ks = bz - batch * d.ksplit;
    const int m0 = by * BM, n0 = bx * BN;
    const int ktiles = (d.K + BK - 1) / BK;
    const int tiles_per_split = (ktiles + d.ksplit - 1) / d.ksplit;
    const int kbeg = ks * tiles_per_split * BK;
    int kend = kbeg + tiles_per_split * BK;
    if (kend > d.K) kend = d.K;
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    const float* Ab = d.A.p + (long)batch * d.A.bstride;
    const float* Bb = d.B.p + (long)batch * d.B.bstride;

    // ---- per-slot source pointers, advanced by one k-tile per fetch (no divisions in the loop)
    const float* pa[NA];
    const float* pb[NB];
    int wa[NA], wb[NB];
    bool oka[NA], okb[NB];
    auto setup = [&](const Operand& op, const float* base, bool T, bool TRX, int x0, int X, int BX, int f, const float*& p, int& w, bool& ok) {
        if (TRX) {           // source row k = f / 32, 4 consecutive tile rows (source columns) from 4 * (f % 32)
            const int c = x0 + 4 * (f % 32);
            ok = c < X;                                        // X % 4 == 0 (launcher): a quad is inside or outside as a whole
            const int cc = ok ? c : 0;
            const int sg = op.seglen ? cc / op.seglen : 0;
            w = 0;
            p = base + (long)(kbeg + f / 32) * op.ld + (long)sg * op.segstride + (op.seglen ? cc - sg * op.seglen : cc);
        } else if (!T) {     // row = x0 + f/8, 4 consecutive k starting at (f%8)*4
            const int r = x0 + f / 8, c = kbeg + (f % 8) * 4;
            ok = r < X;
            const int sg = op.seglen ? c / op.seglen : 0;
            w = op.seglen ? c - sg * op.seglen : c;
            p = base + (long)(ok ? r : 0) * op.ld + (long)sg * op.segstride + w;
        } else {             // tile row (source column) = f % BX, 4 consecutive source rows k = (f / BX) * 4 ...
            const int c = x0 + f % BX;
            ok = c < X;
            const int cc = ok ? c : 0;
            const int sg = op.seglen ? cc / op.seglen : 0;
            w = 0;
            p = base + (long)(kbeg + 4 * (f / BX)) * op.ld + (long)sg * op.segstride + (op.seglen ? cc - sg * op.seglen : cc);
        }
    };
#pragma unroll
    for (int i = 0; i < NA; ++i) setup(d.A, Ab, TA, TRA, m0, d.M, BM, tid + i * 256, pa[i], wa[i], oka[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i) setup(d.B, Bb, TB, TRB, n0, d.N, BN, tid + i * 256, pb[i], wb[i], okb[i]);

    typedef f32x4 Slot;
    // fp16 x 2 only: per-operand power-of-two scales (measured maximum or the fixed one), undone in the epilogue
    const float sc_a = NPL == 2 ? (((d.flags & GEMM_A_PRE) || !d.amax_a) ? (d.a_pre_scale ? *d.a_pre_scale : F16_SCALE) : pow2_scale(*d.amax_a)) : 1.0f;
    const float sc_b = NPL == 2 ? (((d.flags & GEMM_B_PRE) || !d.amax_b) ? (d.b_pre_scale ? *d.b_pre_scale : F16_SCALE) : pow2_scale(*d.amax_b)) : 1.0f;
    // full: the whole k-tile lies inside [kbeg, kend), so the load needs no predicate at all (rows / columns past the
    // matrix edge read row / column 0: their products land in accumulator entries the epilogue never stores).
    auto fetch = [&](const Operand& op, bool T, bool TRX, const float*& p, int& w, bool ok, int kpos, bool full) -> Slot {
        Slot s = f32x4{0.f, 0.f, 0.f, 0.f};
        if (TRX) {
            if (full || (ok && kpos < kend)) s = *reinterpret_cast<const f32x4*>(p);
            if (!GDIAG(d, 4)) p += (long)BK * op.ld;
        } else if (!T) {
            if (full) s = *reinterpret_cast<const f32x4*>(p);
            else if (ok && kpos < kend) {
                if (kpos + 3 < kend) s = *reinterpret_cast<const f32x4*>(p);
                else
                    for (int j = 0; j < 4; ++j)
                        if (kpos + j < kend) s[j] = p[j];
            }
            if (!GDIAG(d, 4)) p += BK;             // diag 4 (timing only): every k-tile re-reads tile 0 -> no memory latency
            if (op.seglen) {                          // seglen >= BK (checked by the launcher): at most one wrap per tile
                w += BK;
                const bool wrap = w >= op.seglen;
                w -= wrap ? op.seglen : 0;
                p += wrap ? op.segstride - op.seglen : 0;
            }
        } else {
            if (full) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] = p[(long)j * op.ld];
            } else if (ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (kpos + j < kend) s[j] = p[(long)j * op.ld];
            }
            if (!GDIAG(d, 4)) p += (long)BK * op.ld;
        }
        return s;
    };
    Slot ra[NA], rb[NB];
    auto gload_to = [&](Slot (&qa)[NA], Slot (&qb)[NB], int k0, bool full) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256;
            qa[i] = fetch(d.A, TA, TRA, pa[i], wa[i], oka[i], TRA ? k0 + f / 32 : (TA ? k0 + 4 * (f / BM) : k0 + (f % 8) * 4), full);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256;
            qb[i] = fetch(d.B, TB, TRB, pb[i], wb[i], okb[i], TRB ? k0 + f / 32 : (TB ? k0 + 4 * (f / BN) : k0 + (f % 8) * 4), full);
        }
    };
    auto gload = [&](int k0, bool full) { gload_to(ra, rb, k0, full); };
    // split the prefetched fp32 values and write the three bf16 planes
    const bool pre_a = NPL == 2 && ((d.flags & GEMM_A_PRE) || GDIAG(d, 1024)), pre_b = NPL == 2 && ((d.flags & GEMM_B_PRE) || GDIAG(d, 512));
    auto sstore_one = [&](unsigned char* S, int P, bool T, bool TRX, int BX, int f, const Slot& s, float scale, bool pre = false) {
        const int row = T ? f % BX : f / 8;
        const int k = T ? (f / BX) * 4 : (f % 8) * 4;
        const int o = TRX ? lds_off_t(f / 32, 4 * (f % 32)) : lds_off(row, k);
        if (NPL == 1) {
            *reinterpret_cast<u32x2*>(S + o) = u32x2{round2(s[0], s[1]), round2(s[2], s[3])};
            return;
        }
        if (NPL == 2) {
            if (pre) {      // the operand is an image (format v2): this slot is the 16 bytes of hi pieces (even slot) or of lo pieces (odd slot)
                            // of a group of EIGHT elements -> one 16-byte store into that piece's plane, at the group's position
                const int kk = T ? 0 : (f % 8) * 4, xx = 4 * (f % 32);
                const int plane = TRX ? (f & 1) : ((kk >> 2) & 1);
                const int o8 = TRX ? lds_off_t(f / 32, xx & ~7) : lds_off(row, kk & ~7);
                *reinterpret_cast<u32x4*>(S + plane * P + o8) = __builtin_bit_cast(u32x4, s);
                return;
            }
            unsigned ha, la, hb, lb;
            split2_f16(s[0], s[1], scale, ha, la);
            split2_f16(s[2], s[3], scale, hb, lb);
            *reinterpret_cast<u32x2*>(S + o) = u32x2{ha, hb};
            *reinterpret_cast<u32x2*>(S + P + o) = u32x2{la, lb};
            return;
        }
        unsigned h0, m0_, l0, h1, m1, l1;
        split2(s[0], s[1], h0, m0_, l0);
        split2(s[2], s[3], h1, m1, l1);
        *reinterpret_cast<u32x2*>(S + o) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(S + P + o) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(S + 2 * P + o) = u32x2{l0, l1};
    };
    auto sstore_from = [&](const Slot (&qa)[NA], const Slot (&qb)[NB], int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) sstore_one(As + buf * NPL * PA, PA, TA, TRA, BM, tid + i * 256, qa[i], sc_a, pre_a);
#pragma unroll
        for (int i = 0; i < NB; ++i) sstore_one(Bs + buf * NPL * PB, PB, TB, TRB, BN, tid + i * 256, qb[i], sc_b, pre_b);
    };
    auto sstore = [&]() { sstore_from(ra, rb, 0); };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int l31 = lane & 31, kg = lane >> 5;
    // transposing reads: lane 4q + pp of a 16-lane group addresses row k + q, tile rows x16 + 4pp .. of its block.  The k-step
    // (16 rows = 4096 B) and the plane are constant byte offsets; the second 32-row sub-tile flips chunk bit 2 (XOR 64 B).
    const int tq = (lane & 15) >> 2, tpp = lane & 3;
    const int ta_lo = lds_off_t(kg * 8 + tq, wm * (BM / 2) + (lane & 16) + 4 * tpp), ta_hi = lds_off_t(kg * 8 + 4 + tq, wm * (BM / 2) + (lane & 16) + 4 * tpp);
    const int tb_lo = lds_off_t(kg * 8 + tq, wn * (BN / 2) + (lane & 16) + 4 * tpp), tb_hi = lds_off_t(kg * 8 + 4 + tq, wn * (BN / 2) + (lane & 16) + 4 * tpp);
    const int nfull = (kend - kbeg) / BK;           // k-tiles that need no bounds checks
    // one k-tile's products out of LDS buffer `buf`
    auto multiply = [&](int buf) {
        const unsigned char* Ab_ = As + buf * NPL * PA;
        const unsigned char* Bb_ = Bs + buf * NPL * PB;
#pragma unroll
        for (int ks16 = 0; ks16 < BK / 16; ++ks16) {
            bf16x8 a[NPL][MI], b[NPL][NI];
            const int fskip = GDIAG(d, 256) ? 0 : 1;    // 256: every fragment read hits the same LDS word (no bandwidth, same instruction count)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * (BM / 2) + mi * 32 + l31;
                const int o = lds_off(row, ks16 * 16 + kg * 8) * fskip;
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    if constexpr (TRA) a[p][mi] = __builtin_bit_cast(bf16x8, tr_frag(Ab_ + p * PA + ks16 * 4096, ta_lo ^ (mi * 64), ta_hi ^ (mi * 64)));
                    else a[p][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ab_ + p * PA + o));
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wn * (BN / 2) + ni * 32 + l31;
                const int o = lds_off(row, ks16 * 16 + kg * 8) * fskip;
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    if constexpr (TRB) b[p][ni] = __builtin_bit_cast(bf16x8, tr_frag(Bb_ + p * PB + ks16 * 4096, tb_lo ^ (ni * 64), tb_hi ^ (ni * 64)));
                    else b[p][ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bb_ + p * PB + o));
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    f32x16 c = acc[mi][ni];
                    if constexpr (NPL == 2) {   // fp16 x 2: h.l, l.h, h.h (l.l is below 2^-22)
                        const f16x8 ah = __builtin_bit_cast(f16x8, a[0][mi]), al = __builtin_bit_cast(f16x8, a[NPL - 1][mi]);
                        const f16x8 bh = __builtin_bit_cast(f16x8, b[0][ni]), bl = __builtin_bit_cast(f16x8, b[NPL - 1][ni]);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
                        acc[mi][ni] = c;
                        continue;
                    }
                    if (NPL == 3) {   // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NPL - 2][mi], b[NPL - 2][ni], c, 0, 0, 0);   // m.m
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[NPL - 1][ni], c, 0, 0, 0);         // h.l
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NPL - 1][mi], b[0][ni], c, 0, 0, 0);         // l.h
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[NPL - 2][ni], c, 0, 0, 0);         // h.m
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NPL - 2][mi], b[0][ni], c, 0, 0, 0);         // m.h
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], c, 0, 0, 0);                   // h.h
                    acc[mi][ni] = c;
                }
        }
    };
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    if constexpr (WS) {
        // Tile kt lives in buffer kt & 1.  Stagers keep two register sets in flight: at k-tile kt they store tile kt+1 (requested
        // two k-tiles ago) and request tile kt+3 into the registers that frees.  The barrier that ends k-tile kt tells the
        // multipliers that tile kt+1 is in LDS and the stagers that buffer kt & 1 may be overwritten.
        auto full_at = [&](int t) { return t < nfull; };
        if (stager) {
            Slot sa[NA], sb[NB];                       // second register set (ra / rb is the first)
            if (nk > 0) gload_to(ra, rb, kbeg, full_at(0));
            if (nk > 1) gload_to(sa, sb, kbeg + BK, full_at(1));
            if (nk > 0) sstore_from(ra, rb, 0);
            if (nk > 2) gload_to(ra, rb, kbeg + 2 * BK, full_at(2));
            lds_barrier();
            for (int kt = 0; kt < nk; kt += 2) {
                if (kt + 1 < nk) sstore_from(sa, sb, 1);                                   // tile kt+1
                if (kt + 3 < nk) gload_to(sa, sb, kbeg + (kt + 3) * BK, full_at(kt + 3));
                lds_barrier();
                if (kt + 1 >= nk) break;
                if (kt + 2 < nk) sstore_from(ra, rb, 0);                                   // tile kt+2
                if (kt + 4 < nk) gload_to(ra, rb, kbeg + (kt + 4) * BK, full_at(kt + 4));
                lds_barrier();
            }
            return;
        }
        lds_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            multiply(kt & 1);
            lds_barrier();
        }
    } else {
    if (nk > 0) gload(kbeg, false);
    // (A separate loop over the whole k-tiles only -- no bounds logic in it -- was measured too, A/B in one process against this
    // form: the training step was 0.18 ms SLOWER with it, with either split.  One loop it stays.)
    for (int kt = 0; kt < nk; ++kt) {
        if (PROBE) t0 = __builtin_readcyclecounter();
        if (!GDIAG(d, 64)) sstore();               // tile kt: registers -> bf16 planes   (diag 64 / 128 / 256: timing ablations)
        if (PROBE) { t1 = __builtin_readcyclecounter(); ph[0] += t1 - t0; t0 = t1; }
        if (!GDIAG(d, 32)) __syncthreads();        // diag 32 (timing only, wrong results): no barriers in the k-loop
        if (PROBE) { t1 = __builtin_readcyclecounter(); ph[1] += t1 - t0; t0 = t1; }
        // tile kt+1 in flight during the MFMAs
        if (GDIAG(d, 128)) {
        } else if (kt + 1 < nfull) gload(kbeg + (kt + 1) * BK, true);
        else if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK, false);
        if (PROBE) { t1 = __builtin_readcyclecounter(); ph[2] += t1 - t0; t0 = t1; }
        multiply(0);
        if (PROBE) { t1 = __builtin_readcyclecounter(); ph[3] += t1 - t0; t0 = t1; }
        if (!GDIAG(d, 32)) __syncthreads();        // all fragment reads done before the planes are overwritten
        if (PROBE) { t1 = __builtin_readcyclecounter(); ph[4] += t1 - t0; }
    }
    }
    if (PROBE && lane == 0 && blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z) < 64) {
        for (int i = 0; i < 5; ++i) atomicAdd(&g_gemm_phase[wave][i], ph[i]);
        if (wave == 0) atomicAdd(&g_gemm_phase[0][5], (unsigned long long)nk);
    }

    float* Cb = d.C + (long)batch * d.cstride;
    const bool add_bias = d.bias != nullptr && ks == 0;
    const float unscale = (1.0f / sc_a) * (1.0f / sc_b);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (BN / 2) + ni * 32 + l31;
            if (n >= d.N) continue;
            const float bv = add_bias ? d.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (m >= d.M) continue;
                if (d.row_period) {
                    const int q = (m + d.row_off) % d.row_period;
                    if (q < d.row_lo || q >= d.row_hi) continue;
                }
                float* c = Cb + (long)m * d.ldc + n;
                const float v = acc[mi][ni][r] * unscale + bv;
                if (d.ksplit > 1 && d.part) d.part[((long)bz * d.M + m) * d.N + n] = v;      // partial slab of (batch, k-slice) bz: plain stores, added in order by splitk_reduce
                else if (d.ksplit > 1) atomicAdd(c, v);
                else if (d.flags & GEMM_ACCUM) *c += v;
                else *c = v;
            }
        }
}

template <int BM, int BN, bool TA, bool TB>
hipError_t launch_cfg(const GemmDesc& din, hipStream_t s) {
    GemmDesc d = din;
    dim3 grid(cdiv(d.N, BN), cdiv(d.M, BM), d.batch * d.ksplit);
    // transposing LDS reads for the reduction-major operands of 128-wide tiles: every 4-column quad must be loadable as one
    // aligned float4 and lie inside one segment
    auto quad_ok = [&](const Operand& op, bool T, int X) {
        return !T || (X % 4 == 0 && op.ld % 4 == 0 && op.bstride % 4 == 0 && ((size_t)op.p & 15) == 0 &&
                      (op.seglen == 0 || (op.seglen % 4 == 0 && op.segstride % 4 == 0)));
    };
    constexpr bool CAN_TR = (TA || TB) && (!TA || BM == 128) && (!TB || BN == 128);
    // Pre-split operand images: usable when this launch multiplies fp16 x 2 with the fixed scale and loads the operand in whole
    // groups of four along its contiguous axis -- K-contiguous operands with K % 4 == 0 and no K segments shorter than a group, or
    // reduction-major operands through the transposing-read image (which the code below then has to choose).
    const bool tr_all = CAN_TR && g_gemm_tr && (g_gemm_tr == 1 || !(TA && TB)) && !(d.flags & GEMM_BF16) && quad_ok(d.A, TA, d.M) && quad_ok(d.B, TB, d.N);
    const bool f16 = (d.flags & GEMM_F16X2) && !(d.flags & GEMM_BF16) && !d.diag;
    // (whole groups of EIGHT: image format v2)
    auto pre_ok = [&](const Operand& op, bool T, const float* img, int X) {
        if (!img || !f16 || (((size_t)img) & 31) || op.ld % 8 || op.bstride % 8) return false;
        if (op.seglen && (op.seglen % 8 || op.segstride % 8)) return false;
        if (T) return tr_all && X % 8 == 0;
        return d.K % 8 == 0;
    };
    if (pre_ok(d.A, TA, d.a_pre, d.M)) {
        d.A.p = d.a_pre;
        d.flags |= GEMM_A_PRE;
    }
    if (pre_ok(d.B, TB, d.b_pre, d.N)) {
        d.B.p = d.b_pre;
        d.flags |= GEMM_B_PRE;
    }
    // Measured (tools/kbench.py gws, us 256-thread -> wave-specialised): a grid that leaves one workgroup per CU gains -- encoder
    // convolution 8192 x 512 x 2560 (256 tiles) 116 -> 104 -- because its staging waves are a second set of waves to hide latency
    // behind; grids of two and more workgroups per CU do not (projection 280 -> 276, input gradient 245 -> 254), and a
    // reduction-major pair of operands loses (dW 158 -> 230: four staging waves issue 32 dword loads per k-tile each).
    constexpr bool CAN_WS = BM == 128 && BN == 128;
    const bool ws = g_gemm_ws == 2 || (g_gemm_ws == 1 && !(TA && TB) && (long)grid.x * grid.y * grid.z <= 320);
    if constexpr (CAN_TR) {
        if (g_gemm_tr && (g_gemm_tr == 1 || !(TA && TB)) && !(d.flags & GEMM_BF16) && quad_ok(d.A, TA, d.M) && quad_ok(d.B, TB, d.N)) {
            if constexpr (CAN_WS) {
                if ((d.flags & GEMM_F16X2) && ws && !d.diag) {
                    hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, 2, false, true, true>), grid, dim3(512), 0, s, d);
                    return hipGetLastError();
                }
            }
            if (d.flags & GEMM_F16X2) hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, 2, false, true>), grid, dim3(256), 0, s, d);
            else hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, 3, false, true>), grid, dim3(256), 0, s, d);
            return hipGetLastError();
        }
    }
    if constexpr (CAN_WS) {
        if ((d.flags & GEMM_F16X2) && !(d.flags & GEMM_BF16) && ws && !d.diag) {
            hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, 2, false, false, true>), grid, dim3(512), 0, s, d);
            return hipGetLastError();
        }
    }
    if (d.flags & GEMM_BF16) hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, 1>), grid, dim3(256), 0, s, d);
    else if (d.flags & GEMM_F16X2) hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, 2>), grid, dim3(256), 0, s, d);
    else if (GDIAG(d, 16) && BM == 128 && BN == 128 && !TA && !TB)
        hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, 3, (BM == 128 && BN == 128 && !TA && !TB)>), grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, 3>), grid, dim3(256), 0, s, d);
    return hipGetLastError();
}

template <bool TA, bool TB>
hipError_t launch_layout(const GemmDesc& d, hipStream_t s) {
    auto tiles = [&](int bm, int bn) { return (long)cdiv(d.M, bm) * cdiv(d.N, bn) * d.batch * d.ksplit; };
    // reduction-major operands pay more per staged element, so they prefer the largest tile sooner
    const long want = d.want > 0 ? d.want : ((TA && TB && g_gemm_want > 256) ? 256 : g_gemm_want);
    if (d.N > 64 && d.M > 64 && tiles(128, 128) >= want) return launch_cfg<128, 128, TA, TB>(d, s);
    if (d.M > 64 && tiles(128, 64) >= want) return launch_cfg<128, 64, TA, TB>(d, s);
    return launch_cfg<64, 64, TA, TB>(d, s);
}

}  // namespace

hipError_t gemm_phase_probe(unsigned long long out[24], bool reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_phase), sizeof(unsigned long long) * 24);
    if (e == hipSuccess && reset) {
        unsigned long long z[24] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_phase), z, sizeof(z));
    }
    return e;
}

// called by launch_gemm (gemm_f32.hip) for 16-byte-aligned operands
hipError_t launch_gemm_bf16x3(const GemmDesc& din, hipStream_t s) {
    GemmDesc d = din;
    if (d.ksplit < 2 || d.row_period || d.N % 4 || d.bias) d.part = nullptr;      // partial slabs: plain split-K contractions only (the weight gradients)
    const bool ta = d.flags & GEMM_TA, tb = d.flags & GEMM_TB;
    hipError_t e;
    if (!ta && !tb) e = launch_layout<false, false>(d, s);
    else if (!ta && tb) e = launch_layout<false, true>(d, s);
    else if (ta && tb) e = launch_layout<true, true>(d, s);
    else return hipErrorInvalidValue;
    if (e == hipSuccess && d.part) e = splitk_reduce(d.part, d.ksplit, d.M, d.N, d.batch, d.C, d.ldc, d.cstride, nullptr, true, 0, 0, s);
    return e;
}

}  // namespace ss
