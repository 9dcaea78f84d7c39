// Persistent form of the decoder BLSTM recurrence: ONE launch walks all T time steps of a layer (both directions).
//
// Why: with one launch per step (lstm_step.hip) a step costs ~7-8 us of which ~1.7 us is fp32 MFMA work; the rest is
// the launch/drain floor (~3.5 us) and re-streaming the workgroup's 128 KB slice of W_hh from L2 every step.  Here
// each workgroup keeps its W_hh slice in REGISTERS for the whole sequence -- as the two fp16 pieces of the fp16 x 2 split
// (gemm_bf16x3.hip), 64 VGPRs per lane -- and multiplies with three v_mfma_f32_16x16x32_f16 per product.
//
// Forward: the workgroup (dir, btile, jtile) needs, at step t, h(t-1) of its 16 utterances from the JT workgroups with
// the same (dir, btile) -- nothing else.  Producers store h as fp16 pieces in A-fragment order, so consumers load whole
// fragments and nobody re-splits.  Backward: every workgroup multiplies its OWN 64 gate units of da(t+1) (straight from
// LDS) into a partial dh for all H hidden units and hands 16x16 fp32 tiles to their owners, who add up JT tiles
// (see lstm_seq_bwd_kernel).  Either way there is no grid-wide barrier, only 2*ceil(B/16) independent groups of
// JT = H/16 workgroups.  The 1-D block id is laid out so that a group is one of eight slots, blockIdx % 8 (for up to 8 groups; unused
// slots' workgroups exit at once): slot = XCD under the observed round-robin placement, so every group has an XCD and its L2 to itself.
//
// Hand-off, two forms (both placement-independent, both bounded):
//   tagged payload (backward always; forward when every group sits on one XCD): every payload dword carries its step's tag in
//       bit 0 and the consumer polls the payload itself -- one L2 trip per step, no ordering between dwords needed ("Tagged
//       payload" below);
//   flag line (forward otherwise; CDNA guide Guideline 16 / MI355X_MICROARCH "Valid forms", counter row):
//       producer: payload stores -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier -> one lane stores the step number
//                 into the workgroup's own flag word (relaxed, agent scope; 32 flags of a group share one 128-byte line)
//       consumer: one wave polls the flag line (one load, lane = member) -> workgroup barrier -> every payload load is an sc1 load
//       Flags instead of a shared arrival counter because same-address atomics serialise in the L2 (measured: 4 -> 8 adds per
//       workgroup and step tripled the step time), a store to an own word and a one-line poll do not.
// Payload stores are write-through (sc1) in general.  Round 0 of every launch has each member publish the XCD it runs on
// (s_getreg XCC_ID) in a group mask: if the whole group shares one XCD -- hence one L2 -- ordinary stores suffice, and
// that is worth 1.8 us per backward step (8 MB of sc1 traffic per step otherwise).  Measured, never assumed: a group that
// spans XCDs (B = 16, 48, ...) takes the sc1 path and stays correct (tools/seq_debug.py).
//
// Memory waves: two extra waves per workgroup own all slab traffic.  One brings the cell threads' operands of the coming steps
// into a ring in LDS by LDS-DMA (no registers, so the request depth is a plain s_waitcnt constant), the other writes the cell
// threads' results, left in LDS, to the slabs.  The waves on the hand-off path have nothing in flight but payload and polls,
// and the step barriers are LDS-only (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() would also wait for every outstanding
// global access of the wave, ~1 us a trip).
//
// Residency: the grid (ngroups * JT <= 256 workgroups of 640 threads) fits the 256 CUs at one workgroup per CU; other
// kernels sharing the chip can only delay it (they never wait on it).  Every spin is bounded: on expiry the
// workgroup raises a global abort word that all pollers watch, and the kernel drains (ss_check() reports it).
//
// Inline-asm rules learnt the hard way (both were silent corruptions): (1) an asm load must be waited for inside the same
// statement, or hipcc re-uses / copies its destination register while the load is in flight; (2) an asm instruction that
// reads an MFMA result needs its own wait states (store16_*), hipcc inserts them only for consumers it can see.
// Also learnt: hipcc's wait-count bookkeeping gives up at loop edges when loads and stores (or loads under control flow) are in
// flight together -- it then waits for everything -- so a wave that is meant to keep requests in flight across iterations issues
// only LDS-DMA loads and counts them itself.
#include "common.h"
#include "kernels.h"

namespace ss {

int g_seq_prio = 1;    // 1: persistent recurrence waves run at s_setprio 3
int g_seq_tag = 1;     // forward recurrence: the hand-off payload carries its own step tag (no flag round trip, see "Tagged payload" below) where
                       // every group sits on one XCD; 0: always the flag line per group.  (The backward is always tagged.)
int g_seq_var = 2;     // backward kernel, warm-up reads: 2 one dword per 128-byte line (ONE load instruction per step: a line comes into the L2 whole
                       // whichever of its bytes is asked for), 4 one dword per 64 bytes (two instructions), 0 round 2's six 1 KB reads.  What the
                       // warm-up costs is its return traffic through the CU's one vector-memory path, which the polls share: 6 KB -> 192 bytes per
                       // step takes the isolated step from 3.05 to 2.78 us (64 x 128), the training step 5.23 -> 5.21 ms (all shapes -0.02..-0.04)
int g_seq_wlead = 0;   // backward kernel: steps between a warm-up read and the operand request it serves (0: the kernel's default)
int g_seq_spin_log2 = 18;   // bounded wait of the group hand-off: 2^18 polls ~ tens of ms.  ss_tune("seq_spin_log2", 4) makes the
                            // first wait of a launch expire, which is how the tests exercise the abort path on hardware

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return ss_sigmoid(x); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// `prio` kernel argument: bit 0 s_setprio, bits 1..15 timing experiments (SS_DIAG builds only), bits 16..20 log2 of the spin limit,
// bit 21 time-major slabs, bits 22..26 warm-up lead of the backward kernel (0: default)
__device__ __forceinline__ unsigned spin_limit_of(int prio) { return 1u << ((prio >> 16) & 31); }

// Group hand-off without read-modify-write traffic: every member owns one word of its group's flag line (32 words = one
// 128-byte line) and publishes "my step s is complete" by storing s there; a consumer wave reads the whole line with ONE load
// (lane i = member i) and is done when every member's value has reached `want`.  (A shared arrival counter costs one
// same-address atomic per member and step, and those serialise in the L2: measured, it was most of the step's latency.)
// Called by all 64 lanes of one wave; returns false on abort / timeout.  On expiry the launch's abort word makes every other
// workgroup drain, and the engine's STICKY word (host-visible, never cleared by a step) records that this step's results are
// garbage: the Adam kernel refuses to apply them and the next C-ABI call reports it.
struct SeqAbort {
    unsigned* launch;      // abort word of this launch (sync[0], zeroed per step)
    unsigned* sticky;      // engine-wide, nullable
    unsigned limit;
};
__device__ __forceinline__ bool wait_flags(const unsigned* flags, int members, unsigned want, const SeqAbort& ab) {
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; ++spins) {
        const unsigned v = lane < members ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
        if (__all((int)(v - want) >= 0)) return true;
        if ((spins & 63) == 63 && __hip_atomic_load(ab.launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (spins > ab.limit) {
            if (lane == 0) {
                __hip_atomic_store(ab.launch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ab.sticky) __hip_atomic_fetch_or(ab.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void publish(unsigned* flag, unsigned v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Tagged payload.  The flag protocol costs three dependent L2 trips per step: payload stores acknowledged ->
// flag store -> a poll that sees it -> payload loads.  Here every payload DWORD carries the tag of its step in bit 0, so the
// consumer polls the payload itself and needs no ordering between dwords at all: one trip.  A ping-pong half is reused every
// second step; its tag alternates per use and starts at 1, so the buffers must be ZERO when a launch starts (the engine keeps
// one exchange buffer per layer and clears them all with one memset per pass).  The tag displaces the last significand bit
// of an fp32 partial sum (backward) or of one fp16 piece per dword (forward, where the low piece absorbs the change of the
// high one): below the fp16 x 2 product's own 2^-22.
__device__ __forceinline__ unsigned tag_of(int st) { return (((unsigned)(st - 1) >> 1) + 1u) & 1u; }      // payload consumed at step st >= 1
// bounded-poll bookkeeping shared by the tagged loops: false = give up (abort raised here or elsewhere)
__device__ __forceinline__ bool poll_continue(unsigned spins, const SeqAbort& ab) {
    if ((spins & 63) == 63 && __hip_atomic_load(ab.launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    if (spins > ab.limit) {
        if ((threadIdx.x & 63) == 0) {
            __hip_atomic_store(ab.launch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ab.sticky) __hip_atomic_fetch_or(ab.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return false;
    }
    __builtin_amdgcn_s_sleep(1);
    return true;
}

// LDS read the compiler cannot see: a wave with LDS-DMA in flight would otherwise be made to wait for all of it first
__device__ __forceinline__ int lds_peek(const int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(unsigned long)(__attribute__((address_space(3))) const int*)p) : "memory");
    return v;
}

__device__ __forceinline__ void store_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// exact 3-way truncation split of fp32 into bf16 pieces x = h + m + l (see gemm_bf16x3.hip); returns the three 16-bit patterns
__device__ __forceinline__ void split1(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned b = __float_as_uint(x);
    const float r = x - __uint_as_float(b & 0xFFFF0000u);
    const unsigned c = __float_as_uint(r);
    const float q = r - __uint_as_float(c & 0xFFFF0000u);
    h = b >> 16;
    m = c >> 16;
    l = __float_as_uint(q) >> 16;
}

// eight fp32 values -> three bf16x8 MFMA fragments
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
    u32x4 H4, M4, L4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h0, m0, l0, h1, m1, l1;
        split1(x[2 * i], h0, m0, l0);
        split1(x[2 * i + 1], h1, m1, l1);
        H4[i] = h0 | (h1 << 16);
        M4[i] = m0 | (m1 << 16);
        L4[i] = l0 | (l1 << 16);
    }
    fh = __builtin_bit_cast(bf16x8, H4);
    fm = __builtin_bit_cast(bf16x8, M4);
    fl = __builtin_bit_cast(bf16x8, L4);
}

// fp16 x 2 split (forward recurrence: |h| < 1 and the weights are bounded, so fixed power-of-two scales are safe):
// y = scale * x = h + l, h = fp16(y) to nearest, l = fp16(y - h): 22 significand bits where the residual is a normal fp16
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float HSCALE = 16384.0f;      // hidden state: |h| < 1  ->  below 2^14
constexpr float WSCALE = 16.0f;         // recurrent weights: up to 4094 in magnitude survive
__device__ __forceinline__ void split1_f16(float y, unsigned& h, unsigned& l) {
    const _Float16 hh = (_Float16)y;
    const _Float16 ll = (_Float16)(y - (float)hh);
    h = (unsigned)__builtin_bit_cast(unsigned short, hh);
    l = (unsigned)__builtin_bit_cast(unsigned short, ll);
}
__device__ __forceinline__ void split8_f16(const float (&x)[8], float scale, f16x8& fh, f16x8& fl) {
    u32x4 H4, L4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h0, l0, h1, l1;
        split1_f16(x[2 * i] * scale, h0, l0);
        split1_f16(x[2 * i + 1] * scale, h1, l1);
        H4[i] = h0 | (h1 << 16);
        L4[i] = l0 | (l1 << 16);
    }
    fh = __builtin_bit_cast(f16x8, H4);
    fl = __builtin_bit_cast(f16x8, L4);
}
__device__ __forceinline__ f32x4 mfma3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc, 0, 0, 0);     // h.l
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc, 0, 0, 0);     // l.h
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);     // h.h   (l.l is below 2^-22)
    return acc;
}
// max over the 16 lanes of a row (all lanes get it); v >= 0, so the bit patterns order like the values
__device__ __forceinline__ float row16_max(float v) {
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));       // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x124, 0xF, 0xF, true));      // row_ror:4
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true));      // row_ror:8
    return __int_as_float(x);
}
// The same three products for N independent accumulators, interleaved so that consecutive MFMAs never depend on each other;
// per accumulator the order of the terms is that of mfma3 (bit-identical results).
template <int N, typename BF>
__device__ __forceinline__ void mfma3_each(const f16x8 (&a)[2], BF&& b, f32x4 (&acc)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b(i, 1), acc[i], 0, 0, 0);     // h.l
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b(i, 0), acc[i], 0, 0, 0);     // l.h
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b(i, 0), acc[i], 0, 0, 0);     // h.h
}
// 16-bit data path (SS_PRECISION_BF16, round 4): the recurrent product from the HIGH fp16 pieces alone -- one MFMA instead of three, half
// the weight registers, half the forward's payload (11 significand bits per operand, fp32 accumulation and state as before)
template <int N, typename BF>
__device__ __forceinline__ void mfma1_each(const f16x8& a, BF&& b, f32x4 (&acc)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b(i), acc[i], 0, 0, 0);
}
__device__ __forceinline__ void load2x2_sc1(const unsigned char* p0, const unsigned char* p1, u32x4 (&r)[2][2]) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %4, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %3, %5, off offset:1024 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[1][0]), "=&v"(r[1][1])
        : "v"(p0), "v"(p1)
        : "memory");
}

// N 16-byte write-through-coherent loads 1 KiB apart from each base, issued and waited for inside ONE asm statement:
// hipcc does not track asm loads, so the destination registers must not be visible to it before the data has landed.
__device__ __forceinline__ void load2x3_sc1(const unsigned char* p0, const unsigned char* p1, const unsigned char* p2, u32x4 (&r)[2][3]) {
    asm volatile(
        "global_load_dwordx4 %0, %6, off sc1\n\t"
        "global_load_dwordx4 %1, %7, off sc1\n\t"
        "global_load_dwordx4 %2, %8, off sc1\n\t"
        "global_load_dwordx4 %3, %6, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %4, %7, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %5, %8, off offset:1024 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[0][2]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[1][2])
        : "v"(p0), "v"(p1), "v"(p2)
        : "memory");
}
__device__ __forceinline__ void load4_sc1(const unsigned char* p, u32x4 (&r)[4]) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
        : "v"(p)
        : "memory");
}
__device__ __forceinline__ void load2_sc1(const unsigned char* p, u32x4 (&r)[2]) {
    asm volatile(
        "global_load_dwordx4 %0, %2, off sc1\n\t"
        "global_load_dwordx4 %1, %2, off offset:1024 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1])
        : "v"(p)
        : "memory");
}

// fp32-grade product on the bf16 pipe: acc += a . b with a = (ah, am, al), b = (bh, bm, bl); smallest terms first
__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// XCD the workgroup runs on.  The L2 is shared (and coherent) inside one XCD only: a group whose members all sit on the
// same XCD can hand its payload over with ordinary stores (they reach the L2 through the write-through L1) instead of
// pushing every tile through to memory with sc1.  Which case holds is DETECTED at run time, never assumed.
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// Round 0 of the group protocol (wave 0 of the workgroup): put my XCD into the group's mask word, publish flag 1, wait for
// all members, read the mask back.  Returns 1 if the group is XCD-local, 0 if it spans XCDs, -1 on abort.  Completion of
// step s is then published as s + 2, and step st waits for st + 1.
__device__ __forceinline__ int group_locality(unsigned* flags, int me, int members, unsigned* mask, const SeqAbort& abortp) {
    const int lane = threadIdx.x & 63;
    if (lane == 0) {
        const unsigned old = __hip_atomic_fetch_or(mask, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old) : "memory");      // the returned value forces the OR to have been performed before the flag
        publish(flags + me, 1u);
    }
    if (!wait_flags(flags, members, 1u, abortp)) return -1;
    const unsigned m = __hip_atomic_load(mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (m & (m - 1)) == 0 ? 1 : 0;
}

// Forward exchange buffer ("xbuf"): h(t) of one group, stored as the two fp16 pieces of every value in the A-fragment
// order of v_mfma_f32_16x16x32_f16, so a consumer wave's loads are whole 1 KiB fragments and nobody re-splits:
//     [2 ping-pong][2 dir][nbt][2 pieces][KC = H/32 k-chunks][64 lanes][8 fp16]        lane = 16 * (k % 32 / 8) + (b % 16)
__device__ __forceinline__ long xb_half(int nbt, int KC) { return 2L * nbt * 2 * KC * 1024; }
__device__ __forceinline__ long xb_group(int dir, int nbt, int bt, int KC) { return ((long)dir * nbt + bt) * 2 * KC * 1024; }

// one lane's bf16 pieces of value (b % 16 = bi, k) go to chunk k/32, lane 16*(k%32/8)+bi, element k%8.  Two lanes with
// adjacent k (even, odd) combine their 16-bit pieces so that the even one stores whole dwords (write-through).
// One 16-byte write-through store per lane (whole 1 KiB tiles per wave: no partial sectors).  NOTE: hipcc inserts the
// MFMA -> VMEM-read wait states only for consumers it can see, so the value handed in must come from a VALU instruction
// (here: the accumulator times the row's unscale factor), never straight out of an MFMA.  The trailing s_nop covers the other
// hazard hipcc cannot see through the asm: the data registers of a store wider than 64 bits must not be rewritten in the two
// wait states after it.
__device__ __forceinline__ void store16_sc1(unsigned char* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store16_plain(unsigned char* p, f32x4 v) {      // XCD-local groups only
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// tag < 0: untagged (flag protocol).  Otherwise bit 0 of each stored dword -- the last significand bit of the EVEN element's
// piece -- is the step tag: the high piece is forced first and the low piece is taken from what is then left, so the pair still
// represents the value to within the low piece's own last bit.
// The eight lanes that hold k % 8 = 0 .. 7 of one utterance row (consecutive lanes of one wave) gather their pieces in three
// exchange rounds, and the first of them stores whole 16-byte fragment slots: a workgroup's h lands as 512 contiguous bytes per
// plane in 2 x 32 lane-stores instead of 2 x 128 four-byte ones (measured neutral on the step time; four times fewer store transactions).
template <bool HI = false>
__device__ __forceinline__ void xb_store(unsigned char* xb_plane0, long plane_stride, int k, int bi, float v, bool local, int tag = -1) {
    unsigned h, l;
    if constexpr (HI) {                                // high pieces only: one plane, the tag in the even element's last bit
        h = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(v * HSCALE));
        if (tag >= 0 && (k & 1) == 0) h = (h & ~1u) | (unsigned)tag;
        const unsigned h0 = h | ((unsigned)__shfl_xor((int)h, 1) << 16);
        const unsigned h1 = (unsigned)__shfl_xor((int)h0, 2);
        const u32x4 hq = {h0, h1, (unsigned)__shfl_xor((int)h0, 4), (unsigned)__shfl_xor((int)h1, 4)};
        if ((k & 7) == 0) {
            unsigned char* q = xb_plane0 + (long)(k >> 5) * 1024 + ((((k & 31) >> 3) * 16 + bi) << 4);
            if (local) store16_plain(q, __builtin_bit_cast(f32x4, hq));
            else store16_sc1(q, __builtin_bit_cast(f32x4, hq));
        }
        return;
    }
    if (tag >= 0 && (k & 1) == 0) {
        const float y = v * HSCALE;
        h = ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)y) & ~1u) | (unsigned)tag;
        const float hv = (float)__builtin_bit_cast(_Float16, (unsigned short)h);
        l = ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)(y - hv)) & ~1u) | (unsigned)tag;
    } else split1_f16(v * HSCALE, h, l);
    const unsigned h0 = h | ((unsigned)__shfl_xor((int)h, 1) << 16), l0 = l | ((unsigned)__shfl_xor((int)l, 1) << 16);      // lanes k % 2 == 0: (k, k+1)
    const unsigned h1 = (unsigned)__shfl_xor((int)h0, 2), l1 = (unsigned)__shfl_xor((int)l0, 2);                             // lanes k % 4 == 0: (k+2, k+3)
    const u32x4 hq = {h0, h1, (unsigned)__shfl_xor((int)h0, 4), (unsigned)__shfl_xor((int)h1, 4)};                           // lanes k % 8 == 0: k .. k+7
    const u32x4 lq = {l0, l1, (unsigned)__shfl_xor((int)l0, 4), (unsigned)__shfl_xor((int)l1, 4)};
    if ((k & 7) == 0) {
        unsigned char* q = xb_plane0 + (long)(k >> 5) * 1024 + ((((k & 31) >> 3) * 16 + bi) << 4);
        if (local) {                                   // group on one XCD: ordinary stores reach the shared L2
            store16_plain(q, __builtin_bit_cast(f32x4, hq));
            store16_plain(q + plane_stride, __builtin_bit_cast(f32x4, lq));
        } else {
            store16_sc1(q, __builtin_bit_cast(f32x4, hq));
            store16_sc1(q + plane_stride, __builtin_bit_cast(f32x4, lq));
        }
    }
}

// grid = ngroups * (H/16), block = 64*NW.   sync (LSTM_SEQ_SYNC_WORDS, all zero on entry, like xb): [0] abort word,
// [1 + group] XCD masks, [64 + 32 * group + member] completion flags
template <int H, int NW, bool TAG, bool HI>
__global__ __launch_bounds__(64 * (NW + 2)) void lstm_seq_fwd_kernel(float* __restrict__ gates, const float* __restrict__ whh_f,
                                                               const float* __restrict__ whh_b, unsigned char* __restrict__ xb,
                                                               float* __restrict__ out, float* __restrict__ csave,
                                                               unsigned* __restrict__ sync, unsigned* __restrict__ sticky,
                                                               const float* __restrict__ xc, int xf, float* __restrict__ out_img, int B, int T,
                                                               int nbt, int prio) {
    // out_img (nullable): the pre-split image of `out` for the GEMMs that consume it (common.h GemmDesc::a_pre), written beside it
    // xc (nullable): the input projections in COMPACT form [B][T / xf][8H] -- the layer's input repeats in blocks of xf frames (the
    // decoder's up-sampled codes, model.py:301-309), so they were computed once per block; `gates` is then only written
    constexpr int JT = H / 16, KC = H / 32, KS = H / NW / 32;       // KS k-steps of 32 per wave
    static_assert(KS == 2, "the persistent forward kernel is written for 64 reduction elements per wave");
    // per-wave partial sums, [unit column][utterance row], rows padded to 20 floats: a lane writes its four accumulator rows
    // with one ds_write_b128 and the cell threads' reads spread over all 32 banks
    // Tagged hand-off: two copies, alternating per step, and ONE barrier per step (a fast wave's partial sums of step t+1 must not
    // land on the copy the cell threads of step t are still reading; by step t+2 they are behind step t+1's barrier).
    __shared__ __attribute__((aligned(16))) float red[TAG ? 2 : 1][NW][4][16][20];
    // Waves NW and NW + 1 are the memory waves (see the backward kernel): the first brings the input projections of the coming
    // steps into a ring in LDS by LDS-DMA, the second writes the activated gates, c and h the cell threads leave in LDS to the
    // slabs one step later.  The waves on the hand-off path then have nothing in flight but the payload and their polls, and the
    // step barriers order LDS traffic only.
    constexpr int OPD = 4;                                                          // steps between a request and its use
    __shared__ __attribute__((aligned(16))) float ops[OPD + 2][16 * 64];            // slot = step % (OPD + 2), float utterance * 64 + gate * 16 + unit
    __shared__ __attribute__((aligned(16))) float st_buf[2][6][16][16];             // [step parity][i, f, g, o, c, h][utterance][unit]
    __shared__ int s_ok;
    if (prio & 1) __builtin_amdgcn_s_setprio(3);  // the recurrence is the critical path; co-resident GEMM waves are filler
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool loader = w == NW, helper = w >= NW;         // wave NW + 1 stores
    const int ngroups = 2 * nbt;
    // Up to 8 groups take the first `ngroups` of EIGHT slots per member index: under the round-robin placement slot = XCD, so every group
    // sits on an XCD of its own (one shared L2: ordinary stores, payload-carried tags) also when there are fewer groups than XCDs; the
    // workgroups of the unused slots leave at once.  (Locality is still measured in round 0, never assumed.)
    const int slots = ngroups <= 8 ? 8 : ngroups;
    const int grp = blockIdx.x % slots, jt = blockIdx.x / slots;
    if (grp >= ngroups) return;
    const int dir = grp / nbt, bt = grp % nbt;
    const int TP = T + 2 * HALO;
    const int li = lane & 15, lq = lane >> 4;
    unsigned* flags = sync + 64 + 32 * grp;
    const SeqAbort abortp{sync, sticky, spin_limit_of(prio)};
    // timing experiments (wrong results): 1 no exchange loads, 2 no payload stores, 4 no input prefetch, 8 no slab stores,
    // 16 no waiting, 32 ordinary stores regardless
#ifdef SS_DIAG
    const int diag = (prio >> 1) & 0x7FFF;
#else
    constexpr int diag = 0;
#endif
    if (w == 0) {
        const int r = group_locality(flags, jt, JT, sync + 1 + grp, abortp);
        if (lane == 0) s_ok = r;
    }
    __syncthreads();
    if (s_ok < 0) return;
    const bool local = s_ok == 1 || (diag & 32);
    __syncthreads();                                       // s_ok is reused by the step loop
    if (TAG && tid == 0) s_ok = 1;                         // tagged: cleared by a wave whose poll gave up; first read behind step 0's barrier

    auto tau_of = [&](int st) { return HALO + (dir == 0 ? st : T - 1 - st); };
    // slab row of (utterance, haloed time): batch-major [B, T+4, C] or time-major [T+4, B, C] (prio bit 21)
    const bool tm = (prio >> 21) & 1;
    auto row_of = [&](int bb, int tau) { return tm ? (long)tau * B + bb : (long)bb * TP + tau; };
    if (helper) {
        // the working waves' barriers of one step, in the same order; false: the launch is being abandoned
        auto step_barriers = [&](int st) -> bool {
            if constexpr (!TAG) {
                if (st > 0) {
                    lds_barrier();
                    if (!lds_peek(&s_ok)) return false;
                }
            }
            lds_barrier();
            if (TAG && !lds_peek(&s_ok)) return false;
            if constexpr (!TAG) lds_barrier();
            return true;
        };
        if (loader) {
            // request i of a step: idx = lane + 64 i -> utterance idx / 16, gate (idx % 16) / 4, units 4 * (idx % 4) .. + 3; steps past
            // the end re-request the last one (into slots nobody reads any more) so that the count below stays a constant
            auto request = [&](int st) {
                const int tau = tau_of(st < T ? st : T - 1);
                typedef __attribute__((address_space(3))) void* lds_t;
                float* slot = ops[st % (OPD + 2)];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = lane + 64 * i, u = idx >> 4, g = (idx & 15) >> 2, q = idx & 3;
                    const int bu = bt * 16 + u < B ? bt * 16 + u : B - 1;
                    const float* rowp = xc ? xc + ((long)bu * (T / xf) + (tau - HALO) / xf) * (8 * H) : gates + row_of(bu, tau) * (8 * H);
                    __builtin_amdgcn_global_load_lds((const void*)(rowp + dir * 4 * H + g * H + jt * 16 + 4 * q), (lds_t)(slot + 256 * i), 16, 0, 0);
                }
            };
            if (!(diag & 4)) {
                for (int st = 0; st < OPD; ++st) request(st);
            }
            bool ok = true;
            for (int st = 0; st < T && ok; ++st) {
                if (!(diag & 4)) {
                    request(st + OPD);                 // its slot held step st - 2: read before step st - 1's barrier
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * OPD) : "memory");      // in order: everything up to step st has landed
                }
                ok = step_barriers(st);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing may land in this LDS after the workgroup has gone
            if (!ok) return;
            lds_barrier();                                         // the storer's last step
        } else {
            // step st - 1's results, behind step st's barrier (the cell threads are writing the other copy by then)
            auto store_step = [&](int st) {
                if (diag & 8) return;
                const int tau = tau_of(st);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int idx = lane + 64 * i, u = idx / 24, o = (idx % 24) >> 2, q = idx & 3;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&st_buf[st & 1][o][u][4 * q]);
                    if (bt * 16 + u >= B) continue;
                    const long r = row_of(bt * 16 + u, tau);
                    float* dst = o < 4 ? gates + r * (8 * H) + dir * 4 * H + o * H + jt * 16 + 4 * q
                                       : (o == 4 ? csave : out) + r * (2 * H) + dir * H + jt * 16 + 4 * q;
                    *reinterpret_cast<f32x4*>(dst) = v;
                    if (o == 5 && out_img) ss_store_img4(out_img, r * (2 * H) + dir * H + jt * 16 + 4 * q, v[0], v[1], v[2], v[3], 16.0f, (int)((unsigned)prio >> 31));      // (prio bit 31: plain bf16 image)
                }
            };
            for (int st = 0; st < T; ++st) {
                if (!step_barriers(st)) return;
                if (st > 0) store_step(st - 1);
            }
            lds_barrier();
            store_step(T - 1);
        }
        return;
    }

    // this wave's slice of W_hh as fp16 pieces, resident in registers for the whole sequence:
    // B fragment of gate g, k-step ks: lane holds W_hh[g*H + jt*16 + li][(w*KS + ks)*32 + 8*lq .. +7]
    f16x8 bw[4][KS][HI ? 1 : 2];
    {
        const float* W = dir == 0 ? whh_f : whh_b;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float* src = W + (long)(g * H + jt * 16 + li) * H + (w * KS + ks) * 32 + 8 * lq;
                const f32x4 v0 = ld4(src), v1 = ld4(src + 4);
                const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if constexpr (HI) {
                    f16x8 lo_unused;
                    split8_f16(x, WSCALE, bw[g][ks][0], lo_unused);
                } else split8_f16(x, WSCALE, bw[g][ks][0], bw[g][ks][1]);
            }
    }

    const long half = xb_half(nbt, KC), plane = (long)KC * 1024;
    const unsigned char* xrd = xb + xb_group(dir, nbt, bt, KC) + (long)(w * KS) * 1024 + lane * 16;
    unsigned char* xwr = xb + xb_group(dir, nbt, bt, KC);
    const int bi = (tid >> 4) & 15, jj = tid & 15;
    const int j = jt * 16 + jj;
    const bool cell = tid < 256;
    float c_state = 0.f, h_val = 0.f;

    for (int st = 0; st < T; ++st) {
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (st > 0) {                                   // h(-1) = 0: nothing to multiply at the first step
            const unsigned char* p0 = xrd + (st & 1) * half;
            const unsigned char* p1 = p0 + plane;
            u32x4 r[KS][2];
            if constexpr (!TAG) {
                // two waves watch the flag line half a round trip apart: the arrival is seen a quarter of a round trip earlier on
                // average (more watchers only delay the members' flag stores: eight per workgroup tripled the step)
                if (w < 2) {
                    if (w == 1) __builtin_amdgcn_s_sleep(8);
                    const bool ok = (diag & 16) ? true : wait_flags(flags, JT, (unsigned)(st + 1), abortp);
                    if (lane == 0 && (w == 0 || !ok)) s_ok = ok ? 1 : 0;
                }
                lds_barrier();
                if (!s_ok) return;                      // uniform: every thread reads the same LDS word
            }
            if (diag & 1) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) r[ks][0] = r[ks][1] = u32x4{0u, 0u, 0u, 0u};
            } else if constexpr (HI) {
                // high pieces only: this wave's two fragments of plane 0 (1 KiB apart)
                u32x4 r1[2];
                if constexpr (TAG) {
                    const unsigned tg = tag_of(st);
                    for (unsigned spins = 0;; ++spins) {
                        load2_sc1(p0, r1);
                        unsigned bad = 0;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) bad |= (r1[ks][0] ^ tg) | (r1[ks][1] ^ tg) | (r1[ks][2] ^ tg) | (r1[ks][3] ^ tg);
                        if (__all(!(bad & 1u)) || (diag & 16) != 0) break;
                        if (!poll_continue(spins, abortp)) {
                            if (lane == 0) s_ok = 0;
                            break;
                        }
                    }
                } else load2_sc1(p0, r1);
                r[0][0] = r1[0];
                r[1][0] = r1[1];
            } else if constexpr (TAG) {
                // every wave polls its own four fragments (64 hidden units = four producers) until all dwords carry this step's tag
                const unsigned tg = tag_of(st);
                for (unsigned spins = 0;; ++spins) {
                    load2x2_sc1(p0, p1, r);
                    unsigned bad = 0;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc) bad |= (r[ks][pc][0] ^ tg) | (r[ks][pc][1] ^ tg) | (r[ks][pc][2] ^ tg) | (r[ks][pc][3] ^ tg);
                    if (__all(!(bad & 1u)) || (diag & 16) != 0) break;
                    if (!poll_continue(spins, abortp)) {
                        if (lane == 0) s_ok = 0;
                        break;
                    }
                }
            } else load2x2_sc1(p0, p1, r);
            if constexpr (HI) {
                mfma1_each<4>(__builtin_bit_cast(f16x8, r[0][0]), [&](int g) -> const f16x8& { return bw[g][0][0]; }, acc);
                mfma1_each<4>(__builtin_bit_cast(f16x8, r[1][0]), [&](int g) -> const f16x8& { return bw[g][1][0]; }, acc);
            } else {
            {
                const f16x8 a[2] = {__builtin_bit_cast(f16x8, r[0][0]), __builtin_bit_cast(f16x8, r[0][1])};
                mfma3_each<4>(a, [&](int g, int pc) -> const f16x8& { return bw[g][0][pc]; }, acc);
            }
            {
                const f16x8 a[2] = {__builtin_bit_cast(f16x8, r[1][0]), __builtin_bit_cast(f16x8, r[1][1])};
                mfma3_each<4>(a, [&](int g, int pc) -> const f16x8& { return bw[g][1][pc]; }, acc);
            }
            }
        }
        auto& rd = red[TAG ? (st & 1) : 0];
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(&rd[w][g][li][lq * 4]) = acc[g];
        lds_barrier();
        if (TAG && !s_ok) return;                       // uniform
        if (cell) {
            const float* xg = ops[st % (OPD + 2)] + bi * 64 + jj;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) s += rd[ww][g][jj][bi];
                pre[g] = xg[g * 16] + s * (1.0f / (HSCALE * WSCALE));
            }
            const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = ss_gate(pre[2], 2.0f), go = sigmoidf_(pre[3]);
            c_state = gf * c_state + gi * gg;
            h_val = go * ss_tanh(c_state);
            // the hand-off payload: write-through, first (rows past B carry garbage nobody stores downstream)
            if (!(diag & 2)) xb_store<HI>(xwr + ((st + 1) & 1) * half, plane, j, bi, h_val, local, TAG ? (int)tag_of(st + 1) : -1);
            float(*sb)[16][16] = st_buf[st & 1];         // slab copies: the storing wave picks them up behind the next barrier
            sb[0][bi][jj] = gi;
            sb[1][bi][jj] = gf;
            sb[2][bi][jj] = gg;
            sb[3][bi][jj] = go;
            sb[4][bi][jj] = c_state;
            sb[5][bi][jj] = h_val;
        }
        if constexpr (!TAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains before the barrier
            lds_barrier();
            if (tid == 0) publish(flags + jt, (unsigned)(st + 2));
        }
    }
    lds_barrier();                                      // hands the last step's slab copies to the storing wave
}

// Backward: dh(t) = d_out(t) + da(t+1) . W_hh.  The reduction runs over all 4H gate units, which live 64 per workgroup,
// so instead of every workgroup fetching all of da(t+1) (128 KB) each one multiplies ITS OWN 64 gate units -- straight
// from LDS, as fp16 x 2 pieces scaled per utterance row by a power of two taken from the row's own maximum, so gradients
// of any magnitude keep 22 significand bits -- into a partial dh for all H hidden units and hands 16x16 fp32 tiles to their owners:
//     producer jt:  P_jt[b, :] = da(t+1)[b, units of jt] . W_hh[units of jt, :]        [16 x 64] . [64 x H]
//     consumer jt': dh(t)[b, j in jt'] = d_out + sum over the JT producers of P_jt[b, j]
// Exchange buffer: [2 ping-pong][2 dir][nbt][JT consumers][JT producers][64 lanes][4 f32] (a tile in MFMA accumulator order);
// a workgroup writes H/16 KB and reads H/16 KB per step.  Every tile read was written one step earlier: no zeroing needed.
// Tiles are stored write-through (sc1) unless round 0 found the whole group on one XCD (group_locality): then ordinary
// stores are used, which is what makes this formulation pay (8 MB of sc1 traffic per step costs ~1.8 us of every step).
// sync: as in the forward kernel.
template <int H, int NW, bool HI>
__global__ __launch_bounds__(64 * (NW + 2)) void lstm_seq_bwd_kernel(float* __restrict__ gates, const float* __restrict__ whh_f,
                                                               const float* __restrict__ whh_b, unsigned char* __restrict__ xb,
                                                               const float* __restrict__ d_out, const float* __restrict__ csave,
                                                               unsigned* __restrict__ sync, unsigned* __restrict__ sticky,
                                                               unsigned* __restrict__ amax, float* __restrict__ gbias_f,
                                                               float* __restrict__ gbias_b, float* __restrict__ dgs, int xf, int B, int T,
                                                               int nbt, int prio, float* __restrict__ dimg) {
    // dimg (nullable; the 16-bit data path): the pre-activation gradients once more as a plain bf16 tensor of the slab's geometry -- the
    // operand image of the layer's weight- and input-gradient contractions, written here instead of by a separate pass over the slab
    // dgs (nullable): the layer's input repeats in blocks of xf frames (see the forward kernel), so its input / weight gradients only
    // need da SUMMED over each block: the storing wave adds the steps of a block up and writes [B][T / xf][8H] (time order, no atomics)
    constexpr int JT = H / 16, CT = JT / NW, PW = JT / NW;      // column tiles (= consumers) / producers handled per wave
    static_assert(CT == 4 || CT == 2, "the persistent backward kernel is written for 2 or 4 column tiles per wave");
    __shared__ __attribute__((aligned(16))) float red[NW][16][20];      // [unit column][utterance row, padded], see the forward kernel
    __shared__ __attribute__((aligned(16))) unsigned short a_lds[2][2][64][8];      // own da(t) as A fragments: [k-step][piece]
    __shared__ __attribute__((aligned(16))) float row_unscale[16];                  // 1 / (row scale * WSCALE) per utterance
    // the memory wave's two mailboxes: the cell threads' operands of a step (gi, gf, gg, go, d_out, c, c_prev) and their da
    // ring of OPD + 1 steps, filled by LDS-DMA: slot = step % (OPD + 1), then float (utterance * 28 + operand * 4) * 4 + unit
    constexpr int OPD = 4;                                                          // steps between an operand request and its use
    const int WLEAD = ((prio >> 22) & 31) == 31 ? 0 : (((prio >> 22) & 31) ? (prio >> 22) & 31 : 3);                  // ... and between a warm-up read and that request (round 3, in the step: 3 -> 5.28 ms, 8 (round 2's) -> 5.34, 12 -> 5.35; 64 x 192: 7.82 vs 7.91)
    const int svar = (prio >> 27) & 15;
    const int wmode = (svar & 2) ? 1 : ((svar & 4) ? 2 : 0);
    __shared__ __attribute__((aligned(16))) float ops[OPD + 1][7 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float warm_sink[6 * 64 * 4];            // where the warm-up reads land (never read)
    __shared__ __attribute__((aligned(16))) float da_st[4][16][16];                 // [gate][utterance][unit]
    __shared__ int s_ok;
    if (prio & 1) __builtin_amdgcn_s_setprio(3);
    // timing experiments (wrong results unless noted): 1 no exchange loads, 2 no products, 4 no operand fetch, 8 no slab stores,
    // 16 no waits, 32 ordinary tile stores even if the group spans XCDs, 64 operands from two hot rows, 128 idle helper wave
    // (results stay right)
#ifdef SS_DIAG
    const int diag = (prio >> 1) & 0x7FFF;
#else
    constexpr int diag = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // Waves NW and NW + 1 are the MEMORY waves: they own no tile and no cell.  The first fetches the cell threads' operands
    // (activated gates, d_out, c(t), c(t-1): seven 64-byte segments per utterance) two steps ahead of their use and hands them over
    // through LDS; the second writes the cell threads' da (left in LDS) to the gradient slab.  The memory counter is per wave and
    // counts loads and stores in order: whatever a wave on the hand-off path has outstanding sits in front of its next poll's
    // s_waitcnt vmcnt(0), and the slab rows of one step are 2 MB apart (an HBM + TLB miss costs ~2 us).  These waves can afford
    // that, those cannot.  Two waves because hipcc's wait-count bookkeeping gives up at the loop edge: with loads and stores in
    // one wave it waited for the stores it had just issued before it touched a two-step-old load.
    const bool loader = w == NW, helper = w >= NW;       // wave NW + 1 stores
    const int ngroups = 2 * nbt;
    // Up to 8 groups take the first `ngroups` of EIGHT slots per member index: under the round-robin placement slot = XCD, so every group
    // sits on an XCD of its own (one shared L2: ordinary stores, payload-carried tags) also when there are fewer groups than XCDs; the
    // workgroups of the unused slots leave at once.  (Locality is still measured in round 0, never assumed.)
    const int slots = ngroups <= 8 ? 8 : ngroups;
    const int grp = blockIdx.x % slots, jt = blockIdx.x / slots;
    if (grp >= ngroups) return;
    const int dir = grp / nbt, bt = grp % nbt;
    const int TP = T + 2 * HALO;
    const int li = lane & 15, lq = lane >> 4;
    unsigned* flags = sync + 64 + 32 * grp;
    const SeqAbort abortp{sync, sticky, spin_limit_of(prio)};
    if (w == 0) {
        const int r = (diag & 16) ? 1 : group_locality(flags, jt, JT, sync + 1 + grp, abortp);
        if (lane == 0) s_ok = r;
    }
    __syncthreads();
    if (s_ok < 0) return;
    const bool local = s_ok == 1 || (diag & 32);
    __syncthreads();                                   // s_ok is reused by the step loop
    if (tid == 0) s_ok = 1;                            // cleared by a wave whose poll gave up; first read behind step 0's barriers

    auto tau_of = [&](int st) { return HALO + (dir == 0 ? T - 1 - st : st); };
    const bool tm = (prio >> 21) & 1;                  // slabs time-major [T+4, B, C] instead of batch-major [B, T+4, C]
    auto row_of = [&](int bb, int tau) { return tm ? (long)tau * B + bb : (long)bb * TP + tau; };
    if (helper) {
        // the working waves' barriers of one step, in the same order; false: the launch is being abandoned
        auto barriers_to_products = [&](int st) -> bool {
            lds_barrier();
            if (!lds_peek(&s_ok)) return false;
            lds_barrier();
            return true;
        };
        if (loader) {
            // request i of a step: idx = lane + 64 i -> utterance idx / 28, operand (idx % 28) / 4, units 4 * (idx % 4) .. + 3.  The DMA
            // puts lane's 16 bytes at slot + 1024 i + 16 lane, which is the layout the cell threads index.  Steps past the end
            // re-request the last step (into slots nobody reads any more) so that the count below stays a constant.
            auto request = [&](int st) {
                const int sc = st < T ? st : T - 1;
                const int tau = tau_of(sc), tau_prev = dir == 0 ? tau - 1 : tau + 1;
                typedef __attribute__((address_space(3))) void* lds_t;
                float* slot = ops[st % (OPD + 1)];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const int idx = lane + 64 * i, u = idx / 28, o = (idx % 28) >> 2, q = idx & 3;
                    const int bu = bt * 16 + u < B ? bt * 16 + u : B - 1;
                    const float* src = o < 4 ? gates + row_of(bu, tau) * (8 * H) + dir * 4 * H + o * H + jt * 16 + 4 * q
                                             : (o == 4 ? d_out : csave) + row_of(bu, o == 6 ? tau_prev : tau) * (2 * H) + dir * H + jt * 16 + 4 * q;
                    __builtin_amdgcn_global_load_lds((const void*)src, (lds_t)(slot + 256 * i), 16, 0, 0);
                }
                // Warm-up.  The requests above take 64-byte pieces out of rows that 32 workgroups pick apart at different moments:
                // served from HBM that is one DRAM row activation per piece, and a step whose operands are cold takes 3.3 us
                // instead of 2.2 (tools/seq_stride_probe.py; the depth of the ring does not matter, the layout of the slabs does
                // not either).  So WLEAD steps earlier the group's operands are read ONCE in pieces DRAM likes -- 24 H bytes
                // per utterance and step, 6 KB of them per workgroup as whole 1 KB lines -- into this XCD's L2.
                if (!(diag & 64)) {
                    const int sw = st + WLEAD < T ? st + WLEAD : T - 1;
                    const int tw = tau_of(sw);
                    constexpr int PER_U = 24 * H / 1024, NG = 16 * H / 1024, ND = 20 * H / 1024;      // 1 KB chunks per utterance: gates, then d_out, then c
                    auto chunk = [&](int c) -> const float* {
                        const int u = c / PER_U, k = c % PER_U;
                        const int bu = bt * 16 + u < B ? bt * 16 + u : B - 1;
                        return k < NG ? gates + row_of(bu, tw) * (8 * H) + dir * 4 * H + k * 256
                                      : (k < ND ? d_out + row_of(bu, tw) * (2 * H) + dir * H + (k - NG) * 256 : csave + row_of(bu, tw) * (2 * H) + dir * H + (k - ND) * 256);
                    };
                    if (wmode == 1) {                  // a line comes into the L2 whole whichever of its bytes is asked for: lane -> (chunk lane / 8, line lane % 8)
                        if (lane < 48) __builtin_amdgcn_global_load_lds((const void*)(chunk(jt * 6 + (lane >> 3)) + 32 * (lane & 7)), (lds_t)warm_sink, 4, 0, 0);
                    } else if (wmode == 2) {           // ... one dword per 64 bytes
                        if (lane < 48) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int pc = 48 * i + lane;
                                __builtin_amdgcn_global_load_lds((const void*)(chunk(jt * 6 + (pc >> 4)) + 16 * (pc & 15)), (lds_t)(warm_sink + 64 * i), 4, 0, 0);
                            }
                        }
                    } else
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const int c = jt * 6 + i, u = c / PER_U, k = c % PER_U;
                        const int bu = bt * 16 + u < B ? bt * 16 + u : B - 1;
                        const float* src = k < NG ? gates + row_of(bu, tw) * (8 * H) + dir * 4 * H + k * 256
                                                  : (k < ND ? d_out + row_of(bu, tw) * (2 * H) + dir * H + (k - NG) * 256
                                                            : csave + row_of(bu, tw) * (2 * H) + dir * H + (k - ND) * 256);
                        __builtin_amdgcn_global_load_lds((const void*)(src + 4 * lane), (lds_t)(warm_sink + 256 * i), 16, 0, 0);
                    }
                }
            };
            if (!(diag & 4)) {
                for (int st = 0; st < OPD; ++st) request(st);
            }
            bool ok = true;
            for (int st = 0; st < T; ++st) {
                if (!(diag & 4)) {
                    request(st + OPD);                 // its slot held step st - 1: read before that step's second barrier
                    if (diag & 64) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * OPD) : "memory");      // in order: everything up to step st has landed
                    else if (wmode == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * OPD) : "memory");
                    else if (wmode == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 * OPD) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(13 * OPD) : "memory");
                }
                if (!barriers_to_products(st)) {
                    ok = false;
                    break;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing may land in this LDS after the workgroup has gone
            if (!ok) return;
        } else {
            f32x4 bsm[4] = {};
            // bias gradients d b_ih = d b_hh = sum over utterances and time of da: this wave sees every da of the workgroup on its way to the
            // slab, so it adds them up -- in float64 (a T-step fp32 chain of a cancelling sum was the least accurate reduction of the
            // backward, profiles/r03/trained_error_budget.txt), off the cell threads' dependent chain
            double bsd[4][4] = {};
            float* gbias = dir == 0 ? gbias_f : gbias_b;
            for (int st = 0; st < T; ++st) {
                if (!barriers_to_products(st)) return;
                if (!(diag & 8)) {
                    const int tau = tau_of(st), t = tau - HALO;
                    const bool close = dgs && (dir == 0 ? t % xf == 0 : t % xf == xf - 1);      // this step completes its block (dir 0 walks down)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = lane + 64 * i, g = idx >> 6, u = (idx & 63) >> 2, q = idx & 3;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&da_st[g][u][4 * q]);
                        if (bt * 16 + u >= B) continue;
                        // (prio bit 31, with dimg only: every consumer of this layer's gradients reads the bf16 tensor -- the fp32 slab keeps the forward's gates)
                        if (!((unsigned)prio >> 31)) *reinterpret_cast<f32x4*>(gates + row_of(bt * 16 + u, tau) * (8 * H) + dir * 4 * H + g * H + jt * 16 + 4 * q) = v;
                        if (dimg) ss_store_img4(dimg, row_of(bt * 16 + u, tau) * (8 * H) + dir * 4 * H + g * H + jt * 16 + 4 * q, v[0], v[1], v[2], v[3], 1.0f, 1);
                        if (gbias) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) bsd[i][j] += (double)v[j];
                        }
                        if (dgs) {
                            bsm[i] += v;
                            if (close) {
                                *reinterpret_cast<f32x4*>(dgs + ((long)(bt * 16 + u) * (T / xf) + t / xf) * (8 * H) + dir * 4 * H + g * H + jt * 16 + 4 * q) = bsm[i];
                                bsm[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                        }
                    }
                }
            }
            // lane = 4 * utterance + unit quad: the 16 utterances of the tile meet through a fixed butterfly, then one atomic per (gate, unit)
            // and workgroup (gbias_*: [2][4H] = b_ih then b_hh gradient of one direction; the batch tiles meet in the arena)
            if (gbias) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        double t = bsd[i][j];
#pragma unroll
                        for (int o = 4; o < 64; o <<= 1) t += __shfl_xor(t, o);
                        if (lane < 4) {
                            atomicAdd(gbias + i * H + jt * 16 + 4 * lane + j, (float)t);
                            atomicAdd(gbias + 4 * H + i * H + jt * 16 + 4 * lane + j, (float)t);
                        }
                    }
            }
        }
    }

    // B fragments, resident for the whole sequence.  Local reduction index k = gate*16 + unit (64 per workgroup): k-step ks,
    // lane (li = column, lq) holds k = 32*ks + 8*lq + e  ->  W_hh[(2*ks + lq/2)*H + jt*16 + 8*(lq%2) + e][(w*CT + ct)*16 + li]
    f16x8 bw[2][CT][HI ? 1 : 2] = {};
    if (!helper) {
        const float* W = dir == 0 ? whh_f : whh_b;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float* src = W + (long)((2 * ks + (lq >> 1)) * H + jt * 16 + 8 * (lq & 1)) * H + (w * CT + ct) * 16 + li;
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = src[(long)i * H];
                if constexpr (HI) {
                    f16x8 lo_unused;
                    split8_f16(x, WSCALE, bw[ks][ct][0], lo_unused);
                } else split8_f16(x, WSCALE, bw[ks][ct][0], bw[ks][ct][1]);
            }
    }

    const long half = 2L * nbt * JT * JT * 1024;
    unsigned char* gb = xb + ((long)dir * nbt + bt) * JT * JT * 1024;
    const unsigned char* xrd = gb + ((long)jt * JT + w * PW) * 1024 + lane * 16;        // tiles for me, from producers w*PW ..
    unsigned char* xwr = gb + ((long)(w * CT) * JT + jt) * 1024 + lane * 16;            // my tiles for consumers w*CT ..
    const int bi = (tid >> 4) & 15, jj = tid & 15;
    const int b = bt * 16 + bi;
    const bool cell = tid < 256;
    float dc_rec = 0.f;
    float da[4] = {0.f, 0.f, 0.f, 0.f};
    struct Ops {
        float gi, gf, gg, go, d_o, cc, cp;
    };
    Ops cur{};
    float amx = 0.f;                                  // max |da| this thread has produced (for the consumers' fp16 scaling)

    for (int st = 0; st < T && !helper; ++st) {
        f32x4 part = {0.f, 0.f, 0.f, 0.f};
        if (st > 0) {
            const unsigned char* p = xrd + (st & 1) * half;
            u32x4 r[PW];
            if (diag & 1) {
#pragma unroll
                for (int i = 0; i < PW; ++i) r[i] = u32x4{0u, 0u, 0u, 0u};
            } else {
                // every wave polls ITS OWN tiles (one producer wave each) until all of their dwords carry this step's tag
                const unsigned tg = tag_of(st);
                for (unsigned spins = 0;; ++spins) {
                    if constexpr (PW == 4) load4_sc1(p, r);
                    else load2_sc1(p, r);
                    unsigned bad = 0;
#pragma unroll
                    for (int i = 0; i < PW; ++i) bad |= (r[i][0] ^ tg) | (r[i][1] ^ tg) | (r[i][2] ^ tg) | (r[i][3] ^ tg);
                    if (__all(!(bad & 1u)) || (diag & 16) != 0) break;
                    if (!poll_continue(spins, abortp)) {
                        if (lane == 0) s_ok = 0;
                        break;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const f32x4 v = __builtin_bit_cast(f32x4, r[i]);
                part += v;
            }
        }
        *reinterpret_cast<f32x4*>(&red[w][li][lq * 4]) = part;
        lds_barrier();
        if (!s_ok) return;                             // uniform: every thread reads the same LDS word
        if (cell) {
            {
                const float* o = ops[st % (OPD + 1)] + bi * 112 + jj;
                cur = Ops{o[0], o[16], o[32], o[48], o[64], o[80], o[96]};
            }
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += red[ww][jj][bi];
            const float dh = cur.d_o + s;
            const float tc = ss_tanh(cur.cc);
            const float d_o = dh * tc;
            const float dc = dc_rec + dh * cur.go * (1.0f - tc * tc);
            dc_rec = dc * cur.gf;
            da[0] = dc * cur.gg * cur.gi * (1.0f - cur.gi);
            da[1] = dc * cur.cp * cur.gf * (1.0f - cur.gf);
            da[2] = dc * cur.gi * (1.0f - cur.gg * cur.gg);
            da[3] = d_o * cur.go * (1.0f - cur.go);
            if (b < B) {
                amx = fmaxf(fmaxf(amx, fmaxf(fabsf(da[0]), fabsf(da[1]))), fmaxf(fabsf(da[2]), fabsf(da[3])));
            }
            // power-of-two scale of this utterance's 64 gate units (16 lanes x 4 gates): its maximum lands in [2^7, 2^8)
            const float rmax = row16_max(fmaxf(fmaxf(fabsf(da[0]), fabsf(da[1])), fmaxf(fabsf(da[2]), fabsf(da[3]))));
            int se = 261 - (int)(__float_as_uint(rmax) >> 23);
            se = se < 1 ? 1 : (se > 187 ? 187 : se);
            const float rsc = __uint_as_float((unsigned)se << 23);
            if (jj == 0) row_unscale[bi] = __uint_as_float((unsigned)(250 - se) << 23);      // 2^(127 - se) / WSCALE
            // own gate units as A fragments in LDS: k = g*16 + jj  ->  k-step g/2, lane 16*((g%2)*2 + jj/8) + bi, element jj%8
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned h, l;
                split1_f16(da[g] * rsc, h, l);
                const int la = ((g & 1) * 2 + (jj >> 3)) * 16 + bi;
                a_lds[g >> 1][0][la][jj & 7] = (unsigned short)h;
                if constexpr (!HI) a_lds[g >> 1][1][la][jj & 7] = (unsigned short)l;
                da_st[g][bi][jj] = da[g];              // for the memory wave: slab copy for the weight-gradient GEMMs
            }
        }
        lds_barrier();
        if (st + 1 < T && !(diag & 2)) {                  // partial dh(t-1) of my gate units for every hidden unit
            f16x8 a[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pc = 0; pc < (HI ? 1 : 2); ++pc) a[ks][pc] = *reinterpret_cast<const f16x8*>(&a_lds[ks][pc][lane][0]);
            const f32x4 us = *reinterpret_cast<const f32x4*>(&row_unscale[lq * 4]);      // accumulator rows 4*lq + r
            unsigned char* q = xwr + ((st + 1) & 1) * half;
            const unsigned tg = tag_of(st + 1);
            // two column tiles at a time: their two product chains interleave (a dependent 16x16x32 MFMA waits for its predecessor),
            // and the first pair's stores drain under the second pair's products
#pragma unroll
            for (int c0 = 0; c0 < CT; c0 += 2) {
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if constexpr (HI) {
                    mfma1_each<2>(a[0][0], [&](int i) -> const f16x8& { return bw[0][c0 + i][0]; }, acc);
                    mfma1_each<2>(a[1][0], [&](int i) -> const f16x8& { return bw[1][c0 + i][0]; }, acc);
                } else {
                    mfma3_each<2>(a[0], [&](int i, int pc) -> const f16x8& { return bw[0][c0 + i][pc]; }, acc);
                    mfma3_each<2>(a[1], [&](int i, int pc) -> const f16x8& { return bw[1][c0 + i][pc]; }, acc);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 v = acc[i] * us;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = __uint_as_float((__float_as_uint(v[c]) & ~1u) | tg);
                    if (local) store16_plain(q + (long)(c0 + i) * JT * 1024, v);     // hand-off payload, group on one XCD: the shared L2 has it
                    else store16_sc1(q + (long)(c0 + i) * JT * 1024, v);              // group spans XCDs: write-through
                }
            }
        }
        // nothing to publish, and no closing barrier either -- red[] is next written behind this step's second barrier, a_lds /
        // row_unscale behind the next step's first one, which no wave reaches before it has finished its products here
    }
    if (amax && cell) {                               // one atomic per wave: positive floats order like their bit patterns
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
        if (lane == 0) atomicMax(amax, __float_as_uint(amx));
    }
}

// Streaming pre-read of a layer's operand slabs, run on a side stream BESIDE the recurrence that is about to consume them.  A
// recurrence step takes 64-byte pieces out of rows that 32 workgroups pick apart at different moments; served from HBM that is a DRAM
// row activation per piece and costs the backward 0.6 us per step (tools/seq_stride_probe.py: 2.85 us cold, 2.27 after one streaming
// read of the slabs).  This kernel reads the same bytes once, in whole lines, in the order the recurrence will want them -- both
// ends of the sequence first, as the two directions walk inwards -- at memory speed (~50 us for 210 MB), so that they wait in the
// memory-side cache.  One workgroup per (chunk of R time rows from either end, utterance); results are never used.
constexpr int PREWARM_R = 4;
__global__ __launch_bounds__(256) void slab_prewarm_kernel(const float* __restrict__ wide, int cw, const float* __restrict__ n0, const float* __restrict__ n1,
                                                           int cn, float* __restrict__ sink, int B, int T, int time_major) {
    const int b = blockIdx.x % B, c = blockIdx.x / B;
    const int TP = T + 2 * HALO;
    float acc = 0.f;
    for (int side = 0; side < 2; ++side) {
        for (int r = 0; r < PREWARM_R; ++r) {
            const int t = side == 0 ? c * PREWARM_R + r : T - 1 - (c * PREWARM_R + r);
            if (t < 0 || t >= T || (side == 1 && t < (T + 1) / 2) || (side == 0 && t >= (T + 1) / 2)) continue;
            const long row = time_major ? (long)(t + HALO) * B + b : (long)b * TP + t + HALO;
            const f32x4* pw = reinterpret_cast<const f32x4*>(wide + row * cw);
            for (int i = threadIdx.x; i < cw / 4; i += 256) {
                const f32x4 v = pw[i];
                acc += v[0] + v[3];
            }
            if (!n0) continue;
            const f32x4* p0 = reinterpret_cast<const f32x4*>(n0 + row * cn);
            const f32x4* p1 = n1 ? reinterpret_cast<const f32x4*>(n1 + row * cn) : nullptr;
            for (int i = threadIdx.x; i < cn / 4; i += 256) {
                const f32x4 v = p0[i];
                acc += v[0] + v[3];
                if (p1) {
                    const f32x4 u = p1[i];
                    acc += u[0] + u[3];
                }
            }
        }
    }
    if (acc == 1.2345e-30f && T < 0) *sink = acc;      // never true: keeps the loads alive
}

}  // namespace

// The persistent kernels spin on each other: every workgroup of a launch must be resident at once.  How many the current
// device holds is asked of the runtime (CU count x resident workgroups per CU for the larger of the two kernels), once per
// device; a device that cannot hold the grid (a partitioned GPU, a smaller part) takes the one-launch-per-step kernels.
static long resident_limit(int H) {
    static long cache[16][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    long& c = cache[dev][H == 512 ? 0 : 1];
    if (c == 0) {
        hipDeviceProp_t prop;
        int per_cu = 0;
        hipError_t e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess)
            e = H == 512 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_seq_bwd_kernel<512, 8, false>, 640, 0)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_seq_bwd_kernel<256, 8, false>, 640, 0);
        c = e == hipSuccess ? (long)prop.multiProcessorCount * per_cu : -1;
        if (c == 0) c = -1;
    }
    return c > 0 ? c : 0;
}

bool lstm_seq_supported(int B, int H) {
    const int nbt = (B + 15) / 16;
    if (!((H == 512 || H == 256) && 2 * nbt * (H / 16) <= 256 && 2 * nbt <= 60)) return false;     // abort word, masks and 32 flags per group fit LSTM_SEQ_SYNC_WORDS
    return 2L * nbt * (H / 16) <= resident_limit(H);
}

long lstm_seq_xbytes(int B, int H, bool backward) {
    const long nbt = (B + 15) / 16, JT = H / 16;
    return backward ? 2 * (2 * nbt * JT * JT * 1024) : 2 * (2 * nbt * 2 * (H / 32) * 1024);
}

static int seq_slots(int nbt) { return 2 * nbt <= 8 ? 8 : 2 * nbt; }       // group slots per member index (see the kernels)
static int seq_prio_arg(bool time_major, bool img_bf16 = false) {
    return (int)((unsigned)(g_seq_prio & 0xFFFF) | ((unsigned)(g_seq_spin_log2 & 31) << 16) | (time_major ? 1u << 21 : 0u) | ((unsigned)(g_seq_wlead & 31) << 22) |
                 ((unsigned)(g_seq_var & 15) << 27) | (img_bf16 ? 1u << 31 : 0u));
}

hipError_t lstm_seq_fwd(float* gates, const float* whh_f, const float* whh_b, void* xbuf, float* out, float* csave,
                        unsigned* sync, unsigned* sticky, const float* xc, int xf, float* out_img, int B, int T, int H, bool zero_state,
                        bool time_major, hipStream_t s, int img_bf16) {
    if (xc && (xf < 1 || T % xf)) return hipErrorInvalidValue;
    const int nbt = (B + 15) / 16;
    if (!lstm_seq_supported(B, H)) return hipErrorInvalidValue;
    if (zero_state) {
        hipError_t e = hipMemsetAsync(sync, 0, LSTM_SEQ_SYNC_WORDS * sizeof(unsigned), s);
        if (e == hipSuccess) e = hipMemsetAsync(xbuf, 0, lstm_seq_xbytes(B, H, false), s);
        if (e != hipSuccess) return e;
    }
    unsigned char* xb = static_cast<unsigned char*>(xbuf);
    const int pa = seq_prio_arg(time_major, (img_bf16 & 1) != 0);
    const bool hi = (img_bf16 & 2) != 0;              // 16-bit data path: the recurrent product from the high fp16 pieces alone
    // Measured (tools/kbench.py seqtag, us per step flags -> tagged): groups that sit on one XCD each (B = 64: 8 groups under the
    // round-robin placement) 2.47 -> 2.03; groups that span XCDs, whose polls and write-through payload cross the fabric,
    // 3.10 -> 3.23 (B = 16) and 3.00 -> 3.47 (B = 48).  The backward gains either way (3.13 -> 2.98, 3.40 -> 2.49, 4.13 -> 3.25).
    const bool tag = g_seq_tag && (2 * nbt <= 8 || (2 * nbt) % 8 == 0);      // groups expected on one XCD each (seq_slots)
#define SS_SEQ_FWD(HH, NWW, TG, HI_, GRID, BLK) hipLaunchKernelGGL((lstm_seq_fwd_kernel<HH, NWW, TG, HI_>), dim3(GRID), dim3(BLK), 0, s, gates, whh_f, whh_b, xb, out, csave, sync, sticky, xc, xf, out_img, B, T, nbt, pa)
    if (H == 512 && tag) { if (hi) SS_SEQ_FWD(512, 8, true, true, seq_slots(nbt) * 32, 640); else SS_SEQ_FWD(512, 8, true, false, seq_slots(nbt) * 32, 640); }
    else if (H == 512)   { if (hi) SS_SEQ_FWD(512, 8, false, true, seq_slots(nbt) * 32, 640); else SS_SEQ_FWD(512, 8, false, false, seq_slots(nbt) * 32, 640); }
    else if (tag)        { if (hi) SS_SEQ_FWD(256, 4, true, true, seq_slots(nbt) * 16, 384); else SS_SEQ_FWD(256, 4, true, false, seq_slots(nbt) * 16, 384); }
    else                 { if (hi) SS_SEQ_FWD(256, 4, false, true, seq_slots(nbt) * 16, 384); else SS_SEQ_FWD(256, 4, false, false, seq_slots(nbt) * 16, 384); }
#undef SS_SEQ_FWD
    return hipGetLastError();
}

hipError_t lstm_seq_bwd(float* gates, const float* whh_f, const float* whh_b, void* xbuf, const float* d_out,
                        const float* csave, unsigned* sync, unsigned* sticky, float* amax, float* gbias_f, float* gbias_b, float* dgs, int xf,
                        int B, int T, int H, bool zero_state, bool time_major, hipStream_t s, float* dimg, int hi) {
    if (dgs && (xf < 1 || T % xf)) return hipErrorInvalidValue;
    const int nbt = (B + 15) / 16;
    if (!lstm_seq_supported(B, H)) return hipErrorInvalidValue;
    if (zero_state) {
        hipError_t e = hipMemsetAsync(sync, 0, LSTM_SEQ_SYNC_WORDS * sizeof(unsigned), s);
        if (e == hipSuccess) e = hipMemsetAsync(xbuf, 0, lstm_seq_xbytes(B, H, true), s);       // every tag starts at 0
        if (e != hipSuccess) return e;
    }
    unsigned char* xb = static_cast<unsigned char*>(xbuf);
    unsigned* am = reinterpret_cast<unsigned*>(amax);
    const dim3 grid(seq_slots(nbt) * (H / 16)), block(640);
    const int pa = seq_prio_arg(time_major, dimg != nullptr && (hi & 4) != 0);       // hi bit 2: skip the fp32 copy of the gradients (bit 31 of the kernel's prio word)
    hi &= 1;
    if (hi) {
        if (H == 512) hipLaunchKernelGGL((lstm_seq_bwd_kernel<512, 8, true>), grid, block, 0, s, gates, whh_f, whh_b, xb, d_out, csave, sync, sticky, am, gbias_f, gbias_b, dgs, xf, B, T, nbt, pa, dimg);
        else          hipLaunchKernelGGL((lstm_seq_bwd_kernel<256, 8, true>), grid, block, 0, s, gates, whh_f, whh_b, xb, d_out, csave, sync, sticky, am, gbias_f, gbias_b, dgs, xf, B, T, nbt, pa, dimg);
    } else if (H == 512) hipLaunchKernelGGL((lstm_seq_bwd_kernel<512, 8, false>), grid, block, 0, s, gates, whh_f, whh_b, xb, d_out, csave, sync, sticky, am, gbias_f, gbias_b, dgs, xf, B, T, nbt, pa, dimg);
    else                 hipLaunchKernelGGL((lstm_seq_bwd_kernel<256, 8, false>), grid, block, 0, s, gates, whh_f, whh_b, xb, d_out, csave, sync, sticky, am, gbias_f, gbias_b, dgs, xf, B, T, nbt, pa, dimg);
    return hipGetLastError();
}


// Gate for work that is meant to run BESIDE a persistent recurrence on the XCDs it leaves free (engine.hip, xcd_dw): one wave waits until every
// member of every group of the launch that owns `sync` has published round 0 of the group protocol -- i.e. the whole recurrence grid is
// resident -- and only then lets its stream go on.  A work-queue GEMM dispatched behind it finds the recurrence's CUs taken: its
// workgroups for those XCDs stay undispatched until the recurrence ends, the others take every tile.  (Dispatched BEFORE the recurrence,
// its persistent workgroups would hold CUs the recurrence needs.)  Bounded: after ~2 ms the gate opens regardless.
__global__ __launch_bounds__(64) void seq_gate_kernel(const unsigned* __restrict__ sync, int ngroups, int JT) {
    const int lane = threadIdx.x;
    const long long t0 = wall_clock64();
    for (;;) {
        bool ok = true;
        for (int i = lane; i < ngroups * JT; i += 64)
            ok = ok && __hip_atomic_load(sync + 64 + 32 * (i / JT) + i % JT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        if (__all(ok)) return;
        if (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;       // the launch aborted
        if (wall_clock64() - t0 > 200000) return;                                                       // 2 ms at 100 MHz
        __builtin_amdgcn_s_sleep(16);
    }
}
hipError_t seq_gate(const unsigned* sync, int B, int H, hipStream_t s) {
    const int nbt = (B + 15) / 16;
    if (!lstm_seq_supported(B, H) || 2 * nbt > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(seq_gate_kernel, dim3(1), dim3(64), 0, s, sync, 2 * nbt, H / 16);
    return hipGetLastError();
}
// XCDs (of 8) that a persistent recurrence over B utterances leaves free: its 2 * ceil(B / 16) groups take one XCD each
int lstm_seq_free_xcds(int B, int H) {
    const int nbt = (B + 15) / 16;
    if (!lstm_seq_supported(B, H) || 2 * nbt > 8) return 0;
    return 8 - 2 * nbt;
}

hipError_t slab_prewarm(const float* wide, int cw, const float* n0, const float* n1, int cn, float* sink, int B, int T, bool time_major, hipStream_t s) {
    if (cw % 4 || cn % 4) return hipErrorInvalidValue;
    const int chunks = ((T + 1) / 2 + PREWARM_R - 1) / PREWARM_R;
    hipLaunchKernelGGL(slab_prewarm_kernel, dim3(chunks * B), dim3(256), 0, s, wide, cw, n0, n1, cn, sink, B, T, time_major ? 1 : 0);
    return hipGetLastError();
}

}  // namespace ss
