// Weight and bias gradients of the ENCODER BLSTMs (hidden size 1, 8, 32; model.py:71, 174-175, 116) in ONE launch for every layer of every block.
//
// They used to be a confetti of tiny MFMA GEMMs -- per layer dW_ih and dW_hh (both directions batched) plus a column sum: 12-14 launches per
// step of 64 x 64-tile kernels for a few MFLOP each, 32-way split-K met through fp32 atomics (round-3 review: 8.4 TFLOP/s, "these are
// bandwidth problems, not MFMA problems").  Here one kernel reads every layer's pre-activation gradients dG [R][8H], its input X [R][In] and
// its output Hout [R][2H] once and produces, for both directions d,
//     dW_ih[d][n][k] = sum_r dG[r][d*4H + n] * X[r][k]
//     dW_hh[0][n][k] = sum_r dG[r + 1][n] * Hout[r][k]              h(t-1) of the forward direction is one slab row earlier
//     dW_hh[1][n][k] = sum_r dG[r][4H + n] * Hout[r + 1][H + k]     ... of the reverse direction one row later
//     db_ih[d][n] = db_hh[d][n] = sum_r dG[r][d*4H + n]
// (flat sums over all R = B * (T + 4) slab rows: halo rows of dG and Hout are zero, kernels.h).  Plain fp32 FMAs on the vector ALU -- exact
// products, the reference's arithmetic -- in chains of at most R / row_groups rows, the row groups' partial tiles met in float64 in a
// FIXED order by the last workgroup of a tile to arrive (the fence-free hand-over of elementwise.hip's column sums): deterministic, and far more
// accurate than the 32 arrival-order atomics of 6-MFMA split products it replaces.
//
// Work item = (task, 64-row block of the 8H gate units, 64-column tile of [X | h-part + bias], row group); 256 threads own 4 x 4 patches.
#include "common.h"
#include "kernels.h"

namespace ss {
namespace {

constexpr int RC = 64;          // slab rows per LDS chunk

// eight 16-byte loads `stride` floats apart, write-through-coherent (sc1: the partial tiles were stored by workgroups on other XCDs during this
// launch), issued and waited for inside one asm statement (hipcc does not track asm loads)
__device__ __forceinline__ void load8x4_sc1(const float* p, long stride, f32x4 (&v)[8]) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %9, off sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc1\n\t"
        "global_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\t"
        "global_load_dwordx4 %7, %15, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p), "v"(p + stride), "v"(p + 2 * stride), "v"(p + 3 * stride), "v"(p + 4 * stride), "v"(p + 5 * stride), "v"(p + 6 * stride), "v"(p + 7 * stride)
        : "memory");
}

__global__ __launch_bounds__(256) void lstm_small_wgrad_kernel(const WgradTable tb) {
    __shared__ __attribute__((aligned(16))) float a_s[RC + 1][64];       // dG rows r0 .. r0 + RC of this block's 64 gate units
    __shared__ __attribute__((aligned(16))) float b_s[RC + 1][64];       // X tile (rows r0 ..) or, h tile: [Hout fwd half (32) | Hout reverse half (32)]
    __shared__ int s_last;
    // which task / tile
    int tile = blockIdx.y, ti = 0;
    while (ti + 1 < tb.n && tile >= tb.t[ti + 1].tile0) ++ti;
    const WgradTask t = tb.t[ti];
    tile -= t.tile0;
    const int H = t.H, NROW = 8 * H, In = t.In;
    const int xtiles = (In + 63) / 64;
    const int hpb = 4 * H >= 64 ? 1 : 2;                      // h tiles per 64-row block: one when the block lies in one direction, else one per direction
    const int nb = tile / (xtiles + hpb), ct = tile % (xtiles + hpb);       // ct < xtiles: columns 64 ct .. of X; else an h tile (hidden units + bias column)
    const bool htile = ct >= xtiles;
    const int n0 = nb * 64;                                   // first gate unit (of 8H) of this block
    const int hdir = !htile ? -1 : (hpb == 1 ? (n0 >= 4 * H ? 1 : 0) : ct - xtiles);      // the direction an h tile serves (rows of the other one are ignored)
    const int tid = threadIdx.x, tk = tid & 15, tn = tid >> 4;
    const long R = t.R;
    const int RG = gridDim.x;
    long rpg = (R + RG - 1) / RG;
    rpg = (rpg + RC - 1) / RC * RC;
    const long rbeg = (long)blockIdx.x * rpg;
    long rend = rbeg + rpg;
    if (rend > R) rend = R;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // per-row direction of this thread's four gate units (H < 16: both directions share the block)
    int nd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + 4 * tn + i;
        nd[i] = n < NROW ? (n >= 4 * H ? 1 : 0) : -1;
    }
    const int arow = hdir == 0 ? 1 : 0;                       // forward W_hh: dG of row r + 1 pairs with h of row r
    for (long r0 = rbeg; r0 < rend; r0 += RC) {
        __syncthreads();
        // dG rows r0 .. r0 + RC (one more than the chunk: the forward direction's W_hh pairs row r + 1 with h of row r)
        for (int i = tid; i < (RC + 1) * 16; i += 256) {
            const int rr = i >> 4, c4 = (i & 15) * 4;
            const long r = r0 + rr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < R) {
                const float* p = t.dG + r * NROW + n0 + c4;
                if (n0 + c4 + 3 < NROW && (NROW & 3) == 0) v = *reinterpret_cast<const f32x4*>(p);
                else
                    for (int j = 0; j < 4; ++j)
                        if (n0 + c4 + j < NROW) v[j] = p[j];
            }
            *reinterpret_cast<f32x4*>(&a_s[rr][c4]) = v;
        }
        if (!htile) {
            const int c0 = ct * 64;
            const bool vec = (t.x_ld & 3) == 0 && ((((size_t)t.X) & 15) == 0);
            for (int i = tid; i < RC * 16; i += 256) {
                const int rr = i >> 4, c4 = (i & 15) * 4;
                const long r = r0 + rr;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < R) {
                    const float* p = t.X + r * t.x_ld + c0 + c4;
                    if (vec && c0 + c4 + 3 < In) v = *reinterpret_cast<const f32x4*>(p);
                    else
                        for (int j = 0; j < 4; ++j)
                            if (c0 + c4 + j < In) v[j] = p[j];
                }
                *reinterpret_cast<f32x4*>(&b_s[rr][c4]) = v;
            }
        } else {
            // columns 0 .. H-1: this direction's half of Hout -- of row r (forward) / row r + 1 (reverse) --, column 32: ones (the bias), the rest zero
            for (int i = tid; i < RC * 64; i += 256) {
                const int rr = i >> 6, c = i & 63;
                const long r = r0 + rr + (hdir == 1 ? 1 : 0);
                b_s[rr][c] = c < H ? (r < R ? t.Hout[r * 2 * H + hdir * H + c] : 0.f) : (c == 32 ? 1.0f : 0.f);
            }
        }
        __syncthreads();
        const int nrows = (int)((rend - r0) < RC ? (rend - r0) : RC);
        if (!htile || tk < 9) {      // (h tile: only the first 36 columns are not identically zero)
#pragma unroll 4
            for (int rr = 0; rr < nrows; ++rr) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&a_s[rr + arow][4 * tn]);
                const f32x4 b = *reinterpret_cast<const f32x4*>(&b_s[rr][4 * tk]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(a[i], b[j], acc[i][j]);
            }
        }
    }
    // ---- partial tile of this row group -> scratch (write-through), the last group to arrive adds them up in order (float64) and accumulates
    float* part = tb.part + ((long)blockIdx.y * RG + blockIdx.x) * 4096;      // [RG][64 n][64 k] per tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {             // one 16-byte write-through store per patch row (sixteen 4-byte ones per thread were most of the kernel's time)
        const f32x4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(part + (4 * tn + i) * 64 + 4 * tk), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(tb.ctr + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(RG - 1);
    __syncthreads();
    if (!s_last) return;
    const float* p0 = tb.part + (long)blockIdx.y * RG * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + 4 * tn + i;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        const float* pr = p0 + (4 * tn + i) * 64 + 4 * tk;
        int g = 0;
        for (; g + 8 <= RG; g += 8) {                  // row groups in order, eight loads in flight
            f32x4 v[8];
            load8x4_sc1(pr + (long)g * 4096, 4096, v);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) s4[j] += (double)v[q][j];
        }
        for (; g < RG; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) s4[j] += (double)__hip_atomic_load(pr + (long)g * 4096 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n >= NROW) continue;
        const int d = n >= 4 * H ? 1 : 0, nn = n - d * 4 * H;
        if (htile && d != hdir) continue;            // an h tile serves one direction
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = 4 * tk + j;
            const float v = (float)s4[j];
            if (!htile) {
                const int c = ct * 64 + kk;
                if (c < In) (d ? t.gwih1 : t.gwih0)[(long)nn * In + c] += v;
            } else if (kk < H) {
                (d ? t.gwhh1 : t.gwhh0)[(long)nn * H + kk] += v;
            } else if (kk == 32) {
                (d ? t.gbih1 : t.gbih0)[nn] += v;
                (d ? t.gbhh1 : t.gbhh0)[nn] += v;
            }
        }
    }
    if (tid == 0) __hip_atomic_store(tb.ctr + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
}

}  // namespace

int lstm_small_wgrad_tiles(int H, int In) { return ((8 * H + 63) / 64) * ((In + 63) / 64 + (4 * H >= 64 ? 1 : 2)); }

hipError_t lstm_small_wgrad(const WgradTable& tb, hipStream_t s) {
    if (tb.n <= 0) return hipSuccess;
    if (tb.n > WGRAD_MAX || !tb.part || !tb.ctr || tb.row_groups < 1 || tb.tiles_total < 1) return hipErrorInvalidValue;
    for (int i = 0; i < tb.n; ++i)
        if (tb.t[i].H < 1 || tb.t[i].H > 32 || tb.t[i].In < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lstm_small_wgrad_kernel, dim3(tb.row_groups, tb.tiles_total), dim3(256), 0, s, tb);
    return hipGetLastError();
}

}  // namespace ss
