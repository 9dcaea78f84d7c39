// Persistent form of the decoder BLSTM recurrence: ONE launch walks all T time steps of a layer (both directions).
//
// Why: with one launch per step (lstm_step.hip) a step costs ~7-8 us of which only ~1.7 us is MFMA work; the rest is
// the launch/drain floor (~3.5 us) and re-streaming the workgroup's 128 KB slice of W_hh from L2 every step.  Here
// each workgroup keeps its W_hh slice in REGISTERS for the whole sequence (64 VGPRs per lane at 8 waves) and only
// h(t-1) / da(t+1) of its 16 utterances crosses workgroups per step.
//
// Dependency structure: the workgroup (dir, btile, jtile) needs, at step t, the h(t-1) tiles of the JT workgroups
// with the same (dir, btile) -- nothing else.  So there is no grid-wide barrier, only 2*ceil(B/16) independent
// groups of JT = H/16 workgroups, each with one monotonic arrival counter.  The 1-D block id is laid out so that a
// group is blockIdx % ngroups: with B = 64 that is 8 groups = the 8 XCDs under the observed round-robin placement,
// so a group's traffic stays inside one XCD's L2.  That placement is a speed assumption only: the hand-off follows
// the placement-independent protocol of the CDNA guide (Guideline 16 / MI355X_MICROARCH "Valid forms", counter row):
//   producer: payload stored write-through (sc1) -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier ->
//             one lane adds 1 to the group counter (relaxed, agent scope)
//   consumer: one lane polls the counter with sc1 loads -> workgroup barrier -> every payload load is an sc1 load
//
// Residency: the grid (ngroups * JT <= 256 workgroups of 512 threads) fits the 256 CUs at one workgroup per CU; other
// kernels sharing the chip can only delay it (they never wait on it).  Every spin is bounded: on expiry the
// workgroup raises a global abort word that all pollers watch, and the kernel drains (ss_check() reports it).
#include "common.h"
#include "kernels.h"

namespace ss {

int g_seq_prio = 1;    // 1: persistent recurrence waves run at s_setprio 3

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

constexpr unsigned SPIN_LIMIT = 1u << 18;     // ~ tens of ms of polling before giving up

// wait until *cnt >= want (one lane); returns false on abort / timeout
__device__ __forceinline__ bool wait_count(unsigned* cnt, unsigned want, unsigned* abortp) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if ((spins & 63) == 63 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (spins > SPIN_LIMIT) {
            __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ void store_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// four 16-byte sc1 loads 1 KiB apart, waited for inside the statement (hipcc does not track asm loads)
__device__ __forceinline__ void load4x4_sc1(const float* p, f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(p)
        : "memory");
}

// eight 16-byte sc1 loads (two bases, 1 KiB apart), one wait
__device__ __forceinline__ void load8x4_sc1(const float* p, f32x4 (&a)[8]) {
    const float* p2 = p + 1024;
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %4, %9, off sc1\n\t"
        "global_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %7, %9, off offset:3072 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
        : "v"(p), "v"(p2)
        : "memory");
}

// sixteen 16-byte sc1 loads (four bases), one wait
__device__ __forceinline__ void load16x4_sc1(const float* p, f32x4 (&a)[16]) {
    const float* p1 = p + 1024;
    const float* p2 = p + 2048;
    const float* p3 = p + 3072;
    asm volatile(
        "global_load_dwordx4 %0, %16, off sc1\n\t"
        "global_load_dwordx4 %1, %16, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %2, %16, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %3, %16, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %4, %17, off sc1\n\t"
        "global_load_dwordx4 %5, %17, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %6, %17, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %7, %17, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %8, %18, off sc1\n\t"
        "global_load_dwordx4 %9, %18, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %10, %18, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %11, %18, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %12, %19, off sc1\n\t"
        "global_load_dwordx4 %13, %19, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %14, %19, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %15, %19, off offset:3072 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]),
          "=&v"(a[8]), "=&v"(a[9]), "=&v"(a[10]), "=&v"(a[11]), "=&v"(a[12]), "=&v"(a[13]), "=&v"(a[14]), "=&v"(a[15])
        : "v"(p), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}

// grid = ngroups * (H/16), block = 64*NW.   sync: [0..ngroups) arrival counters, [64] abort word (all zero on entry)
template <int H, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_seq_fwd_kernel(float* __restrict__ gates, const float* __restrict__ wfrag,
                                                               float* __restrict__ hf, float* __restrict__ out,
                                                               float* __restrict__ csave, unsigned* __restrict__ sync,
                                                               int B, int T, int nbt, int prio) {
    constexpr int JT = H / 16, NC = H / 16, kw = H / NW, nchunk = kw / 16;
    static_assert(nchunk == 4, "the persistent forward kernel is written for 4 chunks per wave");
    __shared__ float red[NW][4][16][16];
    __shared__ int s_ok;
    if (prio) __builtin_amdgcn_s_setprio(3);      // the recurrence is the critical path; co-resident GEMM waves are filler
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ngroups = 2 * nbt;
    const int grp = blockIdx.x % ngroups, jt = blockIdx.x / ngroups;
    const int dir = grp / nbt, bt = grp % nbt;
    const int TP = T + 2 * HALO;
    const int li = lane & 15;
    unsigned* cnt = sync + grp;
    unsigned* abortp = sync + 64;

    // this wave's slice of W_hh, resident in registers for the whole sequence
    f32x4 bw[4][nchunk];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < nchunk; ++c)
            bw[g][c] = ld4(wfrag + ((((long)dir * JT + jt) * 4 + g) * NC + w * nchunk + c) * 256 + lane * 4);

    const long half = 2L * nbt * 16 * H;                                   // floats per ping-pong half
    const float* hrd = hf + (((long)dir * nbt + bt) * NC + w * nchunk) * 256 + lane * 4;
    const int bi = (tid >> 4) & 15, jj = tid & 15;
    const int b = bt * 16 + bi, j = jt * 16 + jj;
    const int bc = b < B ? b : B - 1;
    float* hwr = hf + (((long)dir * nbt + bt) * NC + jt) * 256 + ((jj >> 2) * 16 + bi) * 4 + (jj & 3);
    const bool cell = tid < 256;
    float c_state = 0.f, h_val = 0.f;
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    float xg[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
    auto tau_of = [&](int st) { return HALO + (dir == 0 ? st : T - 1 - st); };
    auto grow_of = [&](int tau) { return gates + ((long)bc * TP + tau) * (8 * H) + dir * 4 * H + j; };
    if (cell) {
        const float* g0 = grow_of(tau_of(0));
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = g0[g * H];
    }

    for (int st = 0; st < T; ++st) {
        const int tau = tau_of(st);
        if (cell && st + 1 < T) {                       // next step's input projection, requested before the wait
            const float* gn = grow_of(tau_of(st + 1));
#pragma unroll
            for (int g = 0; g < 4; ++g) xn[g] = gn[g * H];
        }
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (st > 0) {                                   // h(-1) = 0: nothing to multiply at the first step
            if (tid == 0) s_ok = wait_count(cnt, (unsigned)(JT * st), abortp) ? 1 : 0;
            __syncthreads();
            if (!s_ok) return;                          // uniform: every thread reads the same LDS word
            f32x4 a[nchunk];
            load4x4_sc1(hrd + (st & 1) * half, a[0], a[1], a[2], a[3]);
#pragma unroll
            for (int c = 0; c < nchunk; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][q], bw[g][c][q], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][g][(lane >> 4) * 4 + r][li] = acc[g][r];
        __syncthreads();
        if (cell) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) s += red[ww][g][bi][jj];
                pre[g] = xg[g] + s;
            }
            const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf(pre[2]), go = sigmoidf_(pre[3]);
            c_state = gf * c_state + gi * gg;
            h_val = go * tanhf(c_state);
            sv[0] = gi;
            sv[1] = gf;
            sv[2] = gg;
            sv[3] = go;
            if (b < B) store_sc1(hwr + ((st + 1) & 1) * half, h_val);      // the hand-off payload: write-through, first
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[g] = xn[g];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains before the barrier
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // slab copies (consumed only by later kernels) go out after the group has been signalled
        if (cell && b < B) {
            float* gr = grow_of(tau);
#pragma unroll
            for (int g = 0; g < 4; ++g) gr[g * H] = sv[g];
            const long o = ((long)b * TP + tau) * (2 * H) + dir * H + j;
            csave[o] = c_state;
            out[o] = h_val;
        }
    }
}

// Backward: dh(t) = d_out(t) + da(t+1) . W_hh ; da(t) handed to the group in fragment-major form.
template <int H, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_seq_bwd_kernel(float* __restrict__ gates, const float* __restrict__ wfragT,
                                                               float* __restrict__ gf, const float* __restrict__ d_out,
                                                               const float* __restrict__ csave,
                                                               unsigned* __restrict__ sync, int B, int T, int nbt, int prio) {
    constexpr int JT = H / 16, NC = 4 * H / 16, kw = 4 * H / NW, nchunk = kw / 16;
    static_assert(nchunk == 16 || nchunk == 8, "the persistent backward kernel is written for 8 or 16 chunks per wave");
    __shared__ float red[NW][16][16];
    __shared__ int s_ok;
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ngroups = 2 * nbt;
    const int grp = blockIdx.x % ngroups, jt = blockIdx.x / ngroups;
    const int dir = grp / nbt, bt = grp % nbt;
    const int TP = T + 2 * HALO;
    const int li = lane & 15;
    unsigned* cnt = sync + grp;
    unsigned* abortp = sync + 64;

    f32x4 bw[nchunk];                                                   // W_hh^T slice of this wave
#pragma unroll
    for (int c = 0; c < nchunk; ++c) bw[c] = ld4(wfragT + (((long)dir * JT + jt) * NC + w * nchunk + c) * 256 + lane * 4);

    const long half = 2L * nbt * 16 * 4 * H;
    const float* grd = gf + (((long)dir * nbt + bt) * NC + w * nchunk) * 256 + lane * 4;
    const int bi = (tid >> 4) & 15, jj = tid & 15;
    const int b = bt * 16 + bi, j = jt * 16 + jj;
    const int bc = b < B ? b : B - 1;
    float* gwr = gf + ((long)dir * nbt + bt) * NC * 256 + ((jj >> 2) * 16 + bi) * 4 + (jj & 3);
    const bool cell = tid < 256;
    float dc_rec = 0.f;
    float da[4] = {0.f, 0.f, 0.f, 0.f};
    auto tau_of = [&](int st) { return HALO + (dir == 0 ? T - 1 - st : st); };
    struct Ops {
        float gi, gf, gg, go, d_o, cc, cp;
    };
    auto fetch = [&](int st) {
        Ops o{};
        const int tau = tau_of(st), tau_prev = dir == 0 ? tau - 1 : tau + 1;
        const float* gr = gates + ((long)bc * TP + tau) * (8 * H) + dir * 4 * H + j;
        const long oo = ((long)bc * TP + tau) * (2 * H) + dir * H + j;
        o.gi = gr[0];
        o.gf = gr[H];
        o.gg = gr[2 * H];
        o.go = gr[3 * H];
        o.d_o = d_out[oo];
        o.cc = csave[oo];
        o.cp = csave[((long)bc * TP + tau_prev) * (2 * H) + dir * H + j];
        return o;
    };
    Ops cur{}, nxt{};
    if (cell) cur = fetch(0);

    for (int st = 0; st < T; ++st) {
        const int tau = tau_of(st);
        if (cell && st + 1 < T) nxt = fetch(st + 1);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (st > 0) {
            if (tid == 0) s_ok = wait_count(cnt, (unsigned)(JT * st), abortp) ? 1 : 0;
            __syncthreads();
            if (!s_ok) return;
            const float* ap = grd + (st & 1) * half;
            f32x4 a[nchunk];
            if constexpr (nchunk == 16) load16x4_sc1(ap, a);
            else load8x4_sc1(ap, a);
#pragma unroll
            for (int c = 0; c < nchunk; c += 2)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][q], bw[c][q], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c + 1][q], bw[c + 1][q], acc1, 0, 0, 0);
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][(lane >> 4) * 4 + r][li] = acc0[r] + acc1[r];
        __syncthreads();
        if (cell) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += red[ww][bi][jj];
            const float dh = cur.d_o + s;
            const float tc = tanhf(cur.cc);
            const float d_o = dh * tc;
            const float dc = dc_rec + dh * cur.go * (1.0f - tc * tc);
            dc_rec = dc * cur.gf;
            da[0] = dc * cur.gg * cur.gi * (1.0f - cur.gi);
            da[1] = dc * cur.cp * cur.gf * (1.0f - cur.gf);
            da[2] = dc * cur.gi * (1.0f - cur.gg * cur.gg);
            da[3] = d_o * cur.go * (1.0f - cur.go);
            if (b < B) {
                float* gw = gwr + ((st + 1) & 1) * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) store_sc1(gw + (long)(g * JT + jt) * 256, da[g]);       // hand-off payload first
            }
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cell && b < B) {                                                   // slab copy for the weight-gradient GEMMs
            float* gr = gates + ((long)b * TP + tau) * (8 * H) + dir * 4 * H + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) gr[g * H] = da[g];
        }
    }
}

}  // namespace

bool lstm_seq_supported(int B, int H) {
    const int nbt = (B + 15) / 16;
    return (H == 512 || H == 256) && 2 * nbt * (H / 16) <= 256 && 2 * nbt <= 64;
}

hipError_t lstm_seq_fwd(float* gates, const float* wfrag, float* hf, float* out, float* csave, unsigned* sync, int B, int T,
                        int H, bool zero_sync, hipStream_t s) {
    const int nbt = (B + 15) / 16;
    if (!lstm_seq_supported(B, H)) return hipErrorInvalidValue;
    hipError_t e = zero_sync ? hipMemsetAsync(sync, 0, 128 * sizeof(unsigned), s) : hipSuccess;
    if (e != hipSuccess) return e;
    if (H == 512) hipLaunchKernelGGL((lstm_seq_fwd_kernel<512, 8>), dim3(2 * nbt * 32), dim3(512), 0, s, gates, wfrag, hf, out, csave, sync, B, T, nbt, g_seq_prio);
    else          hipLaunchKernelGGL((lstm_seq_fwd_kernel<256, 4>), dim3(2 * nbt * 16), dim3(256), 0, s, gates, wfrag, hf, out, csave, sync, B, T, nbt, g_seq_prio);
    return hipGetLastError();
}

hipError_t lstm_seq_bwd(float* gates, const float* wfragT, float* gf, const float* d_out, const float* csave, unsigned* sync,
                        int B, int T, int H, bool zero_sync, hipStream_t s) {
    const int nbt = (B + 15) / 16;
    if (!lstm_seq_supported(B, H)) return hipErrorInvalidValue;
    hipError_t e = zero_sync ? hipMemsetAsync(sync, 0, 128 * sizeof(unsigned), s) : hipSuccess;
    if (e != hipSuccess) return e;
    if (H == 512) hipLaunchKernelGGL((lstm_seq_bwd_kernel<512, 8>), dim3(2 * nbt * 32), dim3(512), 0, s, gates, wfragT, gf, d_out, csave, sync, B, T, nbt, g_seq_prio);
    else          hipLaunchKernelGGL((lstm_seq_bwd_kernel<256, 8>), dim3(2 * nbt * 16), dim3(512), 0, s, gates, wfragT, gf, d_out, csave, sync, B, T, nbt, g_seq_prio);
    return hipGetLastError();
}

}  // namespace ss
