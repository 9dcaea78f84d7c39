// fp32-grade GEMM over operand IMAGES: C[M,N] (+)= sum_k A(m,k) . B(n,k), both operands already split into the two fp16 pieces of
// the fp16 x 2 scheme (common.h: image format v2) by whoever produced them.  Three v_mfma_f32_32x32x16_f16 per k-step (h.l, l.h,
// h.h), fp32 accumulation, the result scaled back by 1 / (sa * sb).
//
// The k-loop contains nothing but LDS-DMA, LDS fragment reads and MFMAs:
//   * k-tiles of 32 go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no vector ALU work); one wave-instruction moves
//     1 KB = whole 128-byte lines: 8 rows x (4 groups x {hi 16 B, lo 16 B}) of a K-contiguous operand, or one k-row x 256 columns
//     (hi and lo) of a reduction-major one.  An LDS-DMA writes lane-linear, so every swizzle is applied to the per-lane SOURCE
//     address; the per-lane part of that address is a 32-bit offset computed once, the k-tile advance is a scalar;
//   * K-contiguous operand in LDS: [row][8 chunks of 16 B], chunk index (2 * k-group + plane) XOR ((row >> 1) & 7): ds_read_b128
//     fragment reads are conflict-free for the hardware's 16-lane groups;
//   * reduction-major operand in LDS: [k-row][hi: BX x 2 B | lo: BX x 2 B] with the 16-byte column chunk XOR ((k & 3) << 2);
//     fragments come out k-contiguous through ds_read_b64_tr_b16 (two per 8-element fragment), conflict-free as well;
//   * a ring of NSLOT k-tile slots; tile t + NSLOT - 1 is requested right behind the ONE barrier of tile t, the wait for a tile's
//     data is a counted s_waitcnt vmcnt(N) that leaves the younger tiles' DMAs in flight, barriers are raw s_barrier (a
//     __syncthreads() would drain the DMAs: cdna_hip_programming.md section 5, "Pipelining across barriers");
//   * split-K writes fp32 partial slabs (plain stores) that splitk_reduce adds up in a fixed order: deterministic, and 1/ksplit-th
//     of the fp32-atomic traffic round 2's kernels paid (MI355X_MICROARCH.md, Global float atomics: 1.3 TB/s chip-wide);
//   * tile order: every XCD gets a contiguous range of the launch whose co-resident workgroups form 2-D blocks of tiles, so the
//     32 CUs of an XCD share a handful of row and column panels in its 4 MB L2.
// Single-piece form (template flag P1, ImgGemmDesc::bf16; round 4, the 16-bit data path of BASELINE configs 3-5): the operands are plain
// bf16 tensors -- a slab that is stored in bf16 IS its own image -- with the geometry in elements as before and 2 bytes per element.  The
// same 128 bytes per tile row and k-tile now hold 64 k-values (eight 16-byte fragments), a reduction-major k-row BX x 2 bytes; one
// v_mfma_f32_32x32x16_bf16 per k16-step and accumulator tile instead of three f16 ones, no scales.  Same DMA geometry, same swizzles, same
// ring; per MFMA 1.5x the LDS bytes of the fp16 x 2 form.
// Tile configurations (launch_gemm_img picks): 256 x 256 (8 waves of 128 x 64, 2 slots, 128 KB, one workgroup per CU),
// 256 x 128 (8 waves of 64 x 64, 3 slots, 144 KB), 128 x 128 (4 waves of 64 x 64, 2 slots, 64 KB: two workgroups per CU).
#include "common.h"
#include "kernels.h"

namespace ss {

int g_img_cfg = -1;       // experiment: force a tile configuration (-1: launch_gemm_img chooses)

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;

#ifdef SS_DIAG
#define IDIAG(d, bits) ((d).diag & (bits))
#else
#define IDIAG(d, bits) 0
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// element offset of column x of a (possibly segmented) row
__device__ __forceinline__ long seg_off(const ImgOperand& op, int x) {
    if (op.seglen == 0) return x;
    const int sg = x / op.seglen;
    return (long)sg * op.segstride + (x - sg * op.seglen);
}

template <int WTM, int WTN, int WGM, int WGN, bool TA, bool TB, int NSLOT, bool P1>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_img_kernel(const ImgGemmDesc d) {
    constexpr int BM = WTM * WGM, BN = WTN * WGN, NW = WGM * WGN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int EB = P1 ? 2 : 4;              // bytes per element of the image geometry
    constexpr int KT = P1 ? 64 : 32;            // k-values per k-tile (128 bytes of a K-contiguous row either way)
    constexpr int KS = KT / 16;                 // k16-steps per k-tile
    constexpr int PART_A = BM * 128, PART_B = BN * 128, SLOT = PART_A + PART_B;
    constexpr int NA = BM / 8 / NW, NB = BN / 8 / NW;            // DMA wave-instructions per wave and k-tile
    static_assert(NA >= 1 && NB >= 1, "tile too small for the wave count");
    static_assert((!TA || BM >= 128) && (!TB || BN >= 128), "reduction-major operands need 128-wide tiles");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    // ---- work-queue form: leave if this XCD is not ours (see ImgGemmDesc::wq)
    __shared__ int s_tile;
    const int total_tiles = d.gm * d.gn * d.batch * d.ksplit;
    // placement log of the work-queue form (test hook: xcc_allow bit 8): per workgroup [XCD, tiles taken, first tick, last tick] (100 MHz) behind the two queue words
    const bool wq_log = d.wq && (d.xcc_allow & 0x100u);
    unsigned log_tiles = 0, log_xcc = 0, log_t0 = 0;
    if (wq_log) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(log_xcc));
        log_t0 = (unsigned)wall_clock64();
    }
    if (d.wq && (d.xcc_allow & 0xFFu) != 0xFFu) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((d.xcc_allow >> (xcc & 15u)) & 1u) {
            if (tid == 0) __hip_atomic_store(d.wq + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            bool seen = false;
            for (int spins = 0; spins < 4096 && !seen; ++spins) {            // bounded (~0.1 ms): placement is round-robin, an allowed workgroup starts within a microsecond
                seen = __hip_atomic_load(d.wq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (!seen) __builtin_amdgcn_s_sleep(2);
            }
            if (seen) {
                if (wq_log && tid == 0) {
                    unsigned* lg = d.wq + 4 + 4 * blockIdx.x;
                    lg[0] = (log_xcc & 15u) | 0x100u;       // left without work
                    lg[1] = 0;
                    lg[2] = log_t0;
                    lg[3] = (unsigned)wall_clock64();
                }
                return;
            }
        }
    }
  for (;;) {
    // ---- which tile: XCD-contiguous ranges, 2-D blocks of tiles inside a range (work-queue form: the next tile of the counter)
    int rem = blockIdx.x;
    if (d.wq) {
        __syncthreads();                    // everyone is through with the previous tile: its LDS slots and s_tile are free
        if (tid == 0) s_tile = (int)__hip_atomic_fetch_add(d.wq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        rem = __builtin_amdgcn_readfirstlane(s_tile);
        if (rem >= total_tiles) {
            if (wq_log && tid == 0) {
                unsigned* lg = d.wq + 4 + 4 * blockIdx.x;
                lg[0] = log_xcc & 15u;
                lg[1] = log_tiles;
                lg[2] = log_t0;
                lg[3] = (unsigned)wall_clock64();
            }
            return;
        }
        ++log_tiles;
    } else {
        const int total = gridDim.x;
        if ((total & 7) == 0) rem = (rem & 7) * (total >> 3) + (rem >> 3);
    }
    const int per_slice = d.gm * d.gn;
    const int z = rem / per_slice;
    int by, bx;
    {
        const int r = rem - z * per_slice;
        const int bsz = d.bh * d.bw, nbx = d.gn / d.bw;
        const int blk = r / bsz, in = r - blk * bsz;
        by = (blk / nbx) * d.bh + in / d.bw;
        bx = (blk % nbx) * d.bw + in % d.bw;
    }
    const int batch = z / d.ksplit, ks = z - batch * d.ksplit;
    const int m0 = by * BM, n0 = bx * BN;
    const int ktiles = (d.K + KT - 1) / KT;
    const int per = (ktiles + d.ksplit - 1) / d.ksplit;
    const int kt0 = ks * per;
    int kt1 = kt0 + per;
    if (kt1 > ktiles) kt1 = ktiles;
    const int nk = kt1 > kt0 ? kt1 - kt0 : 0;

    // ---- LDS-DMA addressing: per lane a 32-bit byte offset from a wave-uniform base; the base advances by one k-tile per issue
    const unsigned char* a_base = (const unsigned char*)d.A.p + ((long)batch * d.A.bstride) * EB;
    const unsigned char* b_base = (const unsigned char*)d.B.p + ((long)batch * d.B.bstride) * EB;
    unsigned voa[NA], vob[NB];
    auto setup = [&](const ImgOperand& op, bool T, int BX, int x0, int X, int i, bool remap) -> unsigned {
        if (!T) {            // rows 8i .. 8i+7 of the tile, lane -> (row, LDS chunk slot); the slot holds logical chunk slot ^ swizzle
            const int row = 8 * i + (lane >> 3), s = lane & 7;
            const int c = s ^ ((row >> 1) & 7);
            int g = x0 + row;
            g = g < X ? g : X - 1;
            if (remap) g = (g / d.rm_T) * d.rm_TP + g % d.rm_T;        // logical row -> slab row (halo rows skipped)
            return (unsigned)((long)g * op.ld * EB + c * 16);
        }
        const int RB = BX * EB;                                    // bytes per k-row: hi half, lo half (single-piece: the bf16 row)
        const int w0 = (i * 1024) % RB + lane * 16;
        const int kr = (i * 1024) / RB + w0 / RB, w = w0 % RB;
        const int plane = P1 ? 0 : w / (BX * 2), ch = (P1 ? w : w % (BX * 2)) >> 4;
        int x = x0 + ((ch ^ ((kr & 3) << 2)) << 3);
        x = x < X ? x : X - 8;                                      // X % 8 == 0 (launcher)
        return (unsigned)(((long)kr * op.ld + seg_off(op, x)) * EB + plane * 16);
    };
#pragma unroll
    for (int j = 0; j < NA; ++j) voa[j] = setup(d.A, TA, BM, m0, d.M, j * NW + wave, !TA && d.rm_T > 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) vob[j] = setup(d.B, TB, BN, n0, d.N, j * NW + wave, false);
    // uniform k-tile offsets (bytes) and, for a segmented K axis, the position inside the segment
    long a_ko, b_ko;
    int a_w = 0, b_w = 0;
    {
        const int k0 = kt0 * KT;
        if (TA) a_ko = (long)k0 * d.A.ld * EB;
        else {
            a_ko = seg_off(d.A, k0) * EB;
            a_w = d.A.seglen ? k0 % d.A.seglen : 0;
        }
        if (TB) b_ko = (long)k0 * d.B.ld * EB;
        else {
            b_ko = seg_off(d.B, k0) * EB;
            b_w = d.B.seglen ? k0 % d.B.seglen : 0;
        }
    }
    int issued = 0;         // k-tiles requested so far (relative to kt0)
    auto issue = [&](int slot) {
        unsigned char* sb = smem + slot * SLOT;
        // reduction-major operands whose K is not a multiple of 32: rows past K come from a block of zeros (A) / are clamped (B)
        const int kabs = (kt0 + issued) * KT;
        const bool tail = TA && TB && kabs + KT > d.K;          // (a K-contiguous operand needs K % KT == 0: gemm_img_supported)
        if (!tail) {
#pragma unroll
            for (int j = 0; j < NA; ++j)
                __builtin_amdgcn_global_load_lds((const void*)(a_base + a_ko + voa[j]), (lds_vp)(sb + (j * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                __builtin_amdgcn_global_load_lds((const void*)(b_base + b_ko + vob[j]), (lds_vp)(sb + PART_A + (j * NW + wave) * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const unsigned char* p = a_base + a_ko + voa[j];
                if (TA) {
                    const int kr = ((j * NW + wave) * 1024) / (BM * EB) + ((BM * EB < 1024) ? (lane * 16) / (BM * EB) : 0);
                    if (kabs + kr >= d.K) p = (const unsigned char*)d.zeros + lane * 16;
                }
                __builtin_amdgcn_global_load_lds((const void*)p, (lds_vp)(sb + (j * NW + wave) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned char* p = b_base + b_ko + vob[j];
                if (TB) {
                    const int kr = ((j * NW + wave) * 1024) / (BN * EB) + ((BN * EB < 1024) ? (lane * 16) / (BN * EB) : 0);
                    if (kabs + kr >= d.K) p -= (long)(kabs + kr - (d.K - 1)) * d.B.ld * EB;       // clamp to the last valid k-row (finite x 0 = 0)
                }
                __builtin_amdgcn_global_load_lds((const void*)p, (lds_vp)(sb + PART_A + (j * NW + wave) * 1024), 16, 0, 0);
            }
        }
        ++issued;
        if (TA) a_ko += (long)KT * d.A.ld * EB;
        else {
            a_ko += 128;
            if (d.A.seglen) {
                a_w += KT;
                if (a_w >= d.A.seglen) {
                    a_w -= d.A.seglen;
                    a_ko += (d.A.segstride - d.A.seglen) * EB;
                }
            }
        }
        if (TB) b_ko += (long)KT * d.B.ld * EB;
        else {
            b_ko += 128;
            if (d.B.seglen) {
                b_w += KT;
                if (b_w >= d.B.seglen) {
                    b_w -= d.B.seglen;
                    b_ko += (d.B.segstride - d.B.seglen) * EB;
                }
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

    // ---- fragment addressing
    const int l31 = lane & 31, kg = lane >> 5;
    const int tq = (lane & 15) >> 2, tpp = lane & 3;
    int fa[MI], fb[NI];         // byte offsets inside the operand's slot part of this lane's fragment of k16-step 0, plane hi
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if (!TA) {
            const int row = wm * WTM + mi * 32 + l31;
            fa[mi] = row * 128 + ((((P1 ? kg : kg << 1)) ^ ((row >> 1) & 7)) << 4);     // chunk = 2 * k-group + plane (single-piece: the k-group)
        } else {
            const int x = wm * WTM + mi * 32 + (lane & 16) + 4 * tpp;
            fa[mi] = (kg * 8 + tq) * (BM * EB) + (((x >> 3) ^ (tq << 2)) << 4) + ((x & 7) << 1);
        }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        if (!TB) {
            const int row = wn * WTN + ni * 32 + l31;
            fb[ni] = row * 128 + ((((P1 ? kg : kg << 1)) ^ ((row >> 1) & 7)) << 4);
        } else {
            const int x = wn * WTN + ni * 32 + (lane & 16) + 4 * tpp;
            fb[ni] = (kg * 8 + tq) * (BN * EB) + (((x >> 3) ^ (tq << 2)) << 4) + ((x & 7) << 1);
        }
    }
    // one fragment (8 consecutive k of one tile row, one plane) of k16-step s
    auto frag = [&](const unsigned char* part, bool T, int BX, int o, int s, int plane) -> f16x8 {
        if (!T) return *reinterpret_cast<const f16x8*>(part + (P1 ? (o ^ (s << 5)) : (o ^ (plane << 4) ^ (s << 6))));
        typedef __attribute__((address_space(3))) s16x4* lptr;
        const unsigned char* p = part + o + s * 16 * (BX * EB) + plane * (BX * 2);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p + 4 * (BX * EB)));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return __builtin_bit_cast(f16x8, u32x4{l2[0], l2[1], h2[0], h2[1]});
    };
    // ---- the k-loop as a software pipeline over fragment STAGES.  A k-tile of 32 is NQ stages: k16-step s = q / MH, and (for wave tiles
    // of four 32-row blocks) the upper / lower pair of blocks h = q % MH.  The fragments of stage q + 1 are requested from LDS BEFORE the
    // MFMAs of stage q are issued -- into the other half of a register double buffer -- so an LDS read has a whole stage of matrix work
    // (12 MFMAs = 384 cycles) to land in, also across the tile boundary: the boundary (wait for the DMA of tile t + 1, barrier, request
    // tile t + NSLOT) sits inside the LAST stage of tile t, whose fragments are in registers by then.  (hipcc's own order was
    // read-everything / s_waitcnt lgkmcnt(0) / multiply, three times per k16-step: 238 us on the 8192 x 4096 x 1024 projection of which
    // 104 were MFMA time, profiles/r03/img_gemm_ablation.txt.)
    constexpr int MH = MI >= 4 ? 2 : 1, MIH = MI / MH, NQ = KS * MH;
    f16x8 ah[2][MIH], al[2][MIH], bh[2][NI], bl[2][NI];
    auto load_stage = [&](int slot, int q) {              // q: compile-time after unrolling
        const unsigned char* sa = smem + slot * SLOT;
        const unsigned char* sb = sa + PART_A;
        const int s = q / MH, h = q % MH, par = q & 1;
        if (h == 0) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                bh[s & 1][ni] = frag(sb, TB, BN, fb[ni], s, 0);
                if (!P1) bl[s & 1][ni] = frag(sb, TB, BN, fb[ni], s, 1);
            }
        }
#pragma unroll
        for (int i = 0; i < MIH; ++i) {
            ah[par][i] = frag(sa, TA, BM, fa[h * MIH + i], s, 0);
            if (!P1) al[par][i] = frag(sa, TA, BM, fa[h * MIH + i], s, 1);
        }
    };
    // part 0: the first product (h . l) of every accumulator tile of the stage; part 1: the other two
    auto mfma_stage = [&](int q, int part) {
        const int s = q / MH, h = q % MH, par = q & 1;
        if (IDIAG(d, 2)) {      // diag 2 (wrong results): no MFMAs -> DMA + fragment-read rate
            if (part == 0) acc[0][0][0] += (float)ah[par][0][0] + (P1 ? 0.f : (float)al[par][MIH - 1][1]) + (float)bh[s & 1][NI - 1][2] + (P1 ? 0.f : (float)bl[s & 1][0][3]);
            return;
        }
        if (P1) {               // one bf16 product per accumulator tile: part 0 the first column block, part 1 the others
#pragma unroll
            for (int i = 0; i < MIH; ++i)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if ((part == 0) != (ni == 0)) continue;
                    acc[h * MIH + i][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[par][i]), __builtin_bit_cast(bf16x8, bh[s & 1][ni]),
                                                                                   acc[h * MIH + i][ni], 0, 0, 0);
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < MIH; ++i)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                f32x16 c = acc[h * MIH + i][ni];
                if (part == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[par][i], bl[s & 1][ni], c, 0, 0, 0);
                else {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[par][i], bh[s & 1][ni], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[par][i], bh[s & 1][ni], c, 0, 0, 0);
                }
                acc[h * MIH + i][ni] = c;
            }
    };
    // boundary in front of tile t (tile t lives in slot t % NSLOT): my LDS reads are through and my DMAs of tile t have landed (with
    // NSLOT - 2 younger tiles still in flight in the steady state); behind the barrier everyone's have, and nobody reads slot
    // (t - 1) % NSLOT any more, which is where tile t + NSLOT - 1 goes
    auto boundary = [&](int t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (NSLOT > 2 && t + NSLOT - 2 < nk) wait_vm<(NSLOT - 2) * (NA + NB)>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NSLOT - 1 < nk && !IDIAG(d, 1)) issue((t + NSLOT - 1) % NSLOT);
    };
#pragma unroll
    for (int t = 0; t < NSLOT - 1; ++t)
        if (t < nk && !IDIAG(d, 1)) issue(t);
    if (nk > 0) {
        boundary(0);
        load_stage(0, 0);
    }
    int slot = 0;
    for (int t = 0; t < nk; ++t) {
        const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // [first products of stage q] [reads of stage q + 1] [rest of stage q]: the wait in front of the first MFMA then covers exactly
            // the stage's own reads (hipcc waits for lgkmcnt(0) whatever is outstanding), and the next stage's reads have the remaining
            // eight MFMAs and the next stage's first four to land in.  The order is pinned: hipcc's scheduler sinks LDS reads to their use
            mfma_stage(q, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < NQ) load_stage(slot, q + 1);
            else if (t + 1 < nk) {
                boundary(t + 1);
                load_stage(nslot, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_stage(q, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        slot = nslot;
    }

    // ---- epilogue
    const float sa_ = d.scale_a ? *d.scale_a : (P1 ? 1.0f : 16.0f), sb_ = d.scale_b ? *d.scale_b : (P1 ? 1.0f : 16.0f);
    const float unscale = (1.0f / sa_) * (1.0f / sb_);
    const bool to_part = d.ksplit > 1;
    float* Cb = to_part ? d.part + (long)z * d.M * d.N : d.C + (long)batch * d.cstride;
    const long ldc = to_part ? d.N : d.ldc;
    const bool add_bias = d.bias != nullptr && !to_part;
    const bool remap_c = d.rm_T > 0 && !to_part;             // (partial slabs are dense in the logical rows; splitk_reduce maps them)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int mb = m0 + wm * WTM + mi * 32 + 4 * kg;
        const int mq = remap_c ? mb / d.rm_T : 0, mr = remap_c ? mb - mq * d.rm_T : 0;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * WTN + ni * 32 + l31;
            if (n >= d.N) continue;
            const float bv = add_bias ? d.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                const int m = mb + dm;
                if (m >= d.M) continue;
                if (d.row_period && !to_part) {
                    const int q = (m + d.row_off) % d.row_period;
                    if (q < d.row_lo || q >= d.row_hi) continue;
                }
                long mrow = m;
                if (remap_c) {                               // rm_T >= 32: at most one utterance boundary inside the 32 rows of an MFMA tile
                    const int off = mr + dm;
                    mrow = (long)mq * d.rm_TP + off + (off >= d.rm_T ? d.rm_TP - d.rm_T : 0);
                }
                float* c = Cb + mrow * ldc + n;
                const float v = acc[mi][ni][r] * unscale + bv;
                if (!to_part && (d.flags & GEMM_ACCUM)) *c += v;
                else *c = v;
            }
        }
    }
    if (!d.wq) return;
  }
}

// C[b][m][n] (+)= sum_ks part[b * ksplit + ks][m][n] (+ bias[n]) in a fixed order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, int M, int N, float* __restrict__ C, long ldc,
                                                            long cstride, const float* __restrict__ bias, int accumulate, int rm_T, int rm_TP) {
    const long slab = (long)M * N;
    const int b = blockIdx.y;
    const float* p0 = part + (long)b * ksplit * slab;
    float* Cb = C + (long)b * cstride;
    const long n4 = slab >> 2;               // N % 4 == 0 (launcher)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 s = *reinterpret_cast<const f32x4*>(p0 + 4 * i);
        for (int k = 1; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(p0 + k * slab + 4 * i);
        const long e = 4 * i;
        const int m = (int)(e / N), n = (int)(e - (long)m * N);
        if (bias) {
            s[0] += bias[n];
            s[1] += bias[n + 1];
            s[2] += bias[n + 2];
            s[3] += bias[n + 3];
        }
        const long mrow = rm_T > 0 ? (long)(m / rm_T) * rm_TP + m % rm_T : m;
        float* c = Cb + mrow * ldc + n;
        if (((ldc & 3) == 0) && ((((size_t)Cb) & 15) == 0)) {
            if (accumulate) s += *reinterpret_cast<const f32x4*>(c);
            *reinterpret_cast<f32x4*>(c) = s;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = accumulate ? c[j] + s[j] : s[j];
        }
    }
}

// fp32 [rows][cols] (row stride ld) -> image of the same geometry (row stride ldi, cols % 8 == 0): per 8 values 16 B of hi pieces, 16 B of
// lo pieces, of scale * value; the scale used is written to *scale_out (the GEMM's epilogue reads it back)
__global__ __launch_bounds__(256) void split_image_kernel(const float* __restrict__ src, long ld, long rows, int cols, const float* __restrict__ amax,
                                                          float fixed_scale, float* __restrict__ img, long ldi, float* __restrict__ scale_out, int bf16) {
    if (bf16) {             // the plain bf16 tensor (round to nearest even), no scale
        const int c8 = cols >> 3;
        const long n8 = rows * c8;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
            const long r = i / c8;
            const int c = (int)(i - r * c8) * 8;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + r * ld + c), v1 = *reinterpret_cast<const f32x4*>(src + r * ld + c + 4);
            const uint2 a = ss_pack_bf16x4(v0[0], v0[1], v0[2], v0[3]), b = ss_pack_bf16x4(v1[0], v1[1], v1[2], v1[3]);
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(img) + 2 * (r * ldi + c)) = uint4{a.x, a.y, b.x, b.y};
        }
        return;
    }
    const float s = amax ? pow2_scale_of(*amax) : fixed_scale;
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
    const int c8 = cols >> 3;
    const long n8 = rows * c8;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + r * ld + c), v1 = *reinterpret_cast<const f32x4*>(src + r * ld + c + 4);
        const uint4 g0 = ss_split_group_s(v0[0], v0[1], v0[2], v0[3], s), g1 = ss_split_group_s(v1[0], v1[1], v1[2], v1[3], s);
        uint4* o = reinterpret_cast<uint4*>(img + r * ldi + c);
        o[0] = uint4{g0.x, g0.y, g1.x, g1.y};
        o[1] = uint4{g0.z, g0.w, g1.z, g1.w};
    }
}

template <int WTM, int WTN, int WGM, int WGN, bool TA, bool TB, int NSLOT, bool P1>
hipError_t launch_one(const ImgGemmDesc& d, int gm, int gn, hipStream_t s) {
    constexpr int BM = WTM * WGM, BN = WTN * WGN;
    constexpr int LDS = NSLOT * (BM + BN) * 128;
    static bool attr_done = false;
    auto kern = gemm_img_kernel<WTM, WTN, WGM, WGN, TA, TB, NSLOT, P1>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    long grid = (long)gm * gn * d.batch * d.ksplit;
    if (d.wq) {                              // work-queue form: what the chip holds at once (every XCD gets an eighth of it), never more than there are tiles
        const long resident = 256L * (LDS <= 80 * 1024 ? 2 : 1);
        if (grid > resident) grid = resident;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WGM * WGN * 64), LDS, s, d);
    return hipGetLastError();
}

template <bool TA, bool TB, bool P1>
hipError_t launch_layout(ImgGemmDesc& d, int cfg, hipStream_t s) {
    const int BM = cfg == 1 ? 128 : 256, BN = cfg == 0 ? 256 : 128;
    d.gm = cdiv(d.M, BM);
    d.gn = cdiv(d.N, BN);
    // co-resident tiles of an XCD (32 at one workgroup per CU) as a 2-D block; shapes that do not divide the grid fall back to rows
    static const int cand[][2] = {{4, 8}, {8, 4}, {2, 16}, {16, 2}, {4, 4}, {2, 8}, {8, 2}, {2, 4}, {4, 2}, {2, 2}};
    d.bh = 1;
    d.bw = d.gn;
    for (auto& c : cand)
        if (d.gm % c[0] == 0 && d.gn % c[1] == 0) {
            d.bh = c[0];
            d.bw = c[1];
            break;
        }
    if (cfg == 0) return launch_one<128, 64, 2, 4, TA, TB, 2, P1>(d, d.gm, d.gn, s);
    if (cfg == 1) return launch_one<64, 64, 2, 2, TA, TB, 2, P1>(d, d.gm, d.gn, s);
    if (cfg == 3) return launch_one<64, 64, 4, 2, TA, TB, 2, P1>(d, d.gm, d.gn, s);       // 256 x 128 on two slots: 96 KB, leaves a CU room for a 48 KB neighbour
    return launch_one<64, 64, 4, 2, TA, TB, 3, P1>(d, d.gm, d.gn, s);
}

}  // namespace

bool gemm_img_supported(const ImgGemmDesc& d) {
    const bool ta = d.flags & GEMM_TA, tb = d.flags & GEMM_TB;
    const int KT = d.bf16 ? 64 : 32;
    auto ok = [&](const ImgOperand& op, bool T, int X) {
        // whole image groups (32 bytes; single-piece: 16); full speed wants whole 128-byte lines (128-byte aligned base and row stride)
        if (((size_t)op.p & (d.bf16 ? 15 : 31)) || op.ld % 8 || op.bstride % 8) return false;
        if (op.seglen && (op.seglen % (T ? 8 : KT) || op.segstride % 8)) return false;      // a k-tile lies inside one segment
        if (!T) return d.K % KT == 0;               // K-contiguous: whole k-tiles
        return X % 8 == 0;                          // reduction-major: whole column groups
    };
    if (d.M < 1 || d.N < 1 || d.K < 1) return false;
    if (d.N % 4) return false;
    if ((ta || tb) && d.K % KT && !d.zeros) return false;
    if (d.rm_T && (d.rm_T < 32 || ta || d.row_period)) return false;
    return ok(d.A, ta, d.M) && ok(d.B, tb, d.N);
}

hipError_t launch_gemm_img(const ImgGemmDesc& din, hipStream_t s) {
    ImgGemmDesc d = din;
    if (d.batch < 1) d.batch = 1;
    if (d.ksplit < 1) d.ksplit = 1;
    if (!gemm_img_supported(d)) return hipErrorInvalidValue;
    if (d.ksplit > 1 && !d.part) return hipErrorInvalidValue;
    const bool ta = d.flags & GEMM_TA, tb = d.flags & GEMM_TB;
    int cfg = d.cfg;
    if (g_img_cfg >= 0) cfg = g_img_cfg;
    if (cfg < 0 || cfg > 3) {
        // the largest tile that still gives every CU a workgroup
        auto wgs = [&](int bm, int bn) { return (long)cdiv(d.M, bm) * cdiv(d.N, bn) * d.batch * d.ksplit; };
        cfg = wgs(256, 256) >= 256 ? 0 : (wgs(256, 128) >= 256 ? 2 : 1);
    }
    hipError_t e;
    if (d.bf16) {
        if (!ta && !tb) e = launch_layout<false, false, true>(d, cfg, s);
        else if (!ta && tb) e = launch_layout<false, true, true>(d, cfg, s);
        else if (ta && tb) e = launch_layout<true, true, true>(d, cfg, s);
        else e = launch_layout<true, false, true>(d, cfg, s);
    } else if (!ta && !tb) e = launch_layout<false, false, false>(d, cfg, s);
    else if (!ta && tb) e = launch_layout<false, true, false>(d, cfg, s);
    else if (ta && tb) e = launch_layout<true, true, false>(d, cfg, s);
    else e = launch_layout<true, false, false>(d, cfg, s);
    if (e != hipSuccess) return e;
    if (d.ksplit > 1) e = splitk_reduce(d.part, d.ksplit, d.M, d.N, d.batch, d.C, d.ldc, d.cstride, d.bias, (d.flags & GEMM_ACCUM) != 0, d.rm_T, d.rm_TP, s);
    return e;
}

hipError_t splitk_reduce(const float* part, int ksplit, int M, int N, int batch, float* C, long ldc, long cstride, const float* bias, bool accumulate,
                         int rm_T, int rm_TP, hipStream_t s) {
    if (N % 4 || ksplit < 1 || batch < 1) return hipErrorInvalidValue;
    const long n4 = (long)M * N / 4;
    int g = cdiv(n4, 256);
    if (g > 2048) g = 2048;
    if (g < 1) return hipSuccess;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g, batch), dim3(256), 0, s, part, ksplit, M, N, C, ldc, cstride, bias, accumulate ? 1 : 0, rm_T, rm_TP);
    return hipGetLastError();
}

hipError_t split_image(const float* src, long ld, long rows, int cols, const float* amax, float fixed_scale, float* img, long ldi, float* scale_out,
                       hipStream_t s, int bf16) {
    if (cols % 8 || ld % 4 || ldi % 8 || (((size_t)src) & 15) || (((size_t)img) & (bf16 ? 15 : 31))) return hipErrorInvalidValue;
    const long n8 = rows * (cols >> 3);
    int g = cdiv(n8, 256);
    if (g > 8192) g = 8192;
    if (g < 1) return hipSuccess;
    hipLaunchKernelGGL(split_image_kernel, dim3(g), dim3(256), 0, s, src, ld, rows, cols, amax, fixed_scale, img, ldi, scale_out, bf16);
    return hipGetLastError();
}

}  // namespace ss
