// Shared declarations for the SpeechSplit gfx950 kernels (internal; the public C ABI is include/speechsplit_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// One GEMM operand.  Element (row, col) lives at
//   p + batch*bstride + row*ld + (col / seglen)*segstride + (col % seglen)        (seglen == 0: + col)
// The segmented form lets a k=5 "same" convolution over a zero-haloed [T+4, C] slab be read as a plain
// matrix with overlapping rows: col = tap*Cin + ci  ->  row + tap rows further down, channel ci.
struct Operand {
    const float* p;
    long ld;
    long bstride;
    int seglen;
    long segstride;
};

enum GemmFlags {
    GEMM_TA = 1,        // A is reduction-major:  A(m,k) stored at row k, col m   (else row m, col k)
    GEMM_TB = 2,        // B is reduction-major:  B(n,k) stored at row k, col n   (else row n, col k)
    GEMM_ACCUM = 4,     // C += result (plain read-modify-write; atomics when ksplit > 1)
    GEMM_F16X2 = 16,    // fp32-grade for O(1)-ranged operands: fp16 x 2 split with a fixed power-of-two scale, 3 MFMAs per k-step
    GEMM_BF16 = 8,      // reduced precision: operands rounded to bf16 (nearest-even), ONE bf16 MFMA per k-step, fp32 accumulate
    GEMM_A_PRE = 32,    // set by the launcher only: the A / B pointer is the operand's PRE-SPLIT image (GemmDesc::a_pre / b_pre)
    GEMM_B_PRE = 64,
};

struct GemmDesc {
    Operand A, B;
    float* C;
    long ldc, cstride;
    const float* bias;      // per output column n, may be null
    int M, N, K;
    int batch;
    int flags;
    int diag;               // timing experiments only (gemm_f32.hip g_gemm_diag); 0 in production
    int want;               // >0: least number of workgroups the tile choice should produce (0: the library default)
    // row_period > 0: a result row m is stored only if (m + row_off) % row_period lies in [row_lo, row_hi).  Lets a contraction run
    // over ALL rows of a haloed slab as one matrix (no 128-row tile cut at every utterance) while its halo rows stay untouched.
    int row_period, row_off, row_lo, row_hi;
    int ksplit;             // >1: split the reduction over blockIdx.z, atomically accumulate into C (C pre-zeroed or ACCUM)
    // GEMM_F16X2 only: device words holding max|A| / max|B| (as produced by the kernels that wrote the operand); the kernel
    // scales the operand by the power of two that brings this maximum into [128, 256).  Null: the fixed scale for O(1) data.
    const float* amax_a;
    const float* amax_b;
    // Optional IMAGES of an operand (format v2, see ImgGemmDesc below; GEMM_F16X2 only): the same matrix geometry (ld, strides, 4 bytes per
    // element), every aligned group of eight elements along the contiguous axis replaced by its hi / lo fp16 pieces -- written once by whoever
    // produces the operand (weight re-layouts, the forward recurrence's storing wave, the gathers, split_image for gradient slabs).  With both
    // images present the engine runs the contraction on the image GEMM (gemm_img.hip); this kernel takes an image when the form it picks loads
    // that operand in whole groups (K-contiguous, or the transposing-read image) and falls back to the fp32 operand otherwise.
    const float* a_pre;
    const float* b_pre;
    // device words holding the power-of-two scale of a FIXED-scale operand (null: 16): what its image was split with, and what the in-loop
    // split uses when the operand is read as fp32 (amax_* null).  The conv trunk's activations carry one (kernels.h act_scales).
    const float* a_pre_scale;
    const float* b_pre_scale;
    // ksplit > 1 (bf16x3 / fp16x2 kernel): batch * ksplit * M * N floats of scratch -- the partial results go there with plain stores and
    // splitk_reduce adds them into C in a fixed order, instead of meeting in C through fp32 atomics (1.3 TB/s chip-wide, arrival order).  Null: atomics.
    float* part;
    int queue;              // 1 (engine, image GEMM only): work-queue form -- the launch runs on whatever XCDs have CUs to give (ImgGemmDesc::wq)
};

// C[b][m][n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n]);  fp32 in, fp32 MFMA accumulate (exact fp32 fma chain)
hipError_t launch_gemm(const GemmDesc& d, hipStream_t stream);

// ---- gemm_img.hip: the contraction over operand IMAGES (format v2, below) -------------------------------------------------------
// Image format v2: the geometry of the fp32 tensor (4 bytes per element, same row stride), but every aligned group of EIGHT
// elements along the contiguous axis holds 16 bytes of hi pieces (fp16 of scale * x, elements 0..7) followed by 16 bytes of lo pieces
// (fp16 of scale * x - hi).  16 bytes = one MFMA fragment of one piece; a 128-byte line = 32 elements with both pieces, so an LDS-DMA
// of whole lines serves the K-contiguous use (k along the row) and the reduction-major use (k = row index) of the same image alike.
struct ImgOperand {
    const void* p;          // image; element (row, col) of the underlying matrix: see Operand (ld / bstride / segments in ELEMENTS)
    long ld, bstride;
    int seglen;             // K-contiguous use: k = seg * seglen + w -> + seg * segstride + w; reduction-major use: the same along the columns
    long segstride;
};
struct ImgGemmDesc {
    ImgOperand A, B;
    float* C;
    long ldc, cstride;
    const float* bias;
    int M, N, K, batch, ksplit;
    int flags;                          // GEMM_TA / GEMM_TB / GEMM_ACCUM
    int row_period, row_off, row_lo, row_hi;      // as GemmDesc (ksplit == 1 only)
    // rm_T > 0 (K-contiguous A only): logical row m of A and of C is row (m / rm_T) * rm_TP + m % rm_T of the memory -- a contraction over
    // the B * T real rows of haloed slabs [B][rm_TP = T + 4][.] with the halo rows skipped outright (no masked rows, no padding tiles)
    int rm_T, rm_TP;
    float* part;                        // ksplit > 1: batch * ksplit * M * N floats of scratch for the partial slabs
    const float *scale_a, *scale_b;     // device words holding the scale an image was split with; null: the fixed 16
    const void* zeros;                  // >= 1 KB of zero bytes (TA && TB with K % 32 != 0: the A rows past K)
    int cfg;                            // -1: choose; 0: 256 x 256, 1: 128 x 128, 2: 256 x 128
    int bf16;                           // 1: single-piece form -- both operands are plain bf16 tensors (2 bytes per element), no scales (gemm_img.hip)
    int diag;                           // SS_DIAG builds only
    int gm, gn, bh, bw;                 // filled by the launcher: tile grid and the 2-D block shape of the tile order
    // Work-queue form (wq != null): the launch is a fixed number of workgroups that take tiles from the counter wq[0] until it runs
    // out, and a workgroup that finds itself on an XCD outside `xcc_allow` (bit i = XCD i) leaves at once -- so the contraction runs on
    // the allowed XCDs only, beside a persistent recurrence that occupies the others (workgroups go to XCDs round-robin by index and a
    // CU mask cannot empty an XCD, so the partition is made by the kernel itself).  wq: two zeroed words per launch ([1]: an allowed
    // workgroup has started; a workgroup elsewhere only leaves once it is set, or does the work itself after a bounded wait).
    unsigned* wq;
    unsigned xcc_allow;
};
bool gemm_img_supported(const ImgGemmDesc& d);
hipError_t launch_gemm_img(const ImgGemmDesc& d, hipStream_t s);
// C[b][m][n] (+)= sum over ks (in order) of part[b * ksplit + ks][m][n] (+ bias[n]); rm_T > 0: logical row m lives at (m / rm_T) * rm_TP + m % rm_T
hipError_t splitk_reduce(const float* part, int ksplit, int M, int N, int batch, float* C, long ldc, long cstride, const float* bias, bool accumulate,
                         int rm_T, int rm_TP, hipStream_t s);
// fp32 [rows][cols] -> image (cols % 8 == 0).  Scale: pow2_scale_of(*amax) when amax is given (device word), else fixed_scale; written to
// *scale_out (nullable) for the GEMM's epilogue
hipError_t split_image(const float* src, long ld, long rows, int cols, const float* amax, float fixed_scale, float* img, long ldi, float* scale_out,
                       hipStream_t s, int bf16 = 0);       // bf16: the image is the plain bf16 tensor (no scale; amax / fixed_scale / scale_out unused)

// phase probe of the bf16x3 kernel (timing experiments): 4 waves x {5 phases, k-tile count} tick sums; see gemm_bf16x3.hip
hipError_t gemm_phase_probe(unsigned long long out[24], bool reset);

// Gate non-linearities of the recurrence kernels.  They sit on the per-time-step critical path, where libdevice's expf /
// tanhf (range reduction, branches) cost a few hundred cycles per step; these use the hardware exp2 / rcp, with a
// series for tanh where the quotient form would cancel: a few ulp.
#ifdef __HIPCC__
// operand with a measured maximum m: the power of two that brings m into [128, 256) (256x headroom below fp16's 65504), capped
// so that an all-zero / denormal tensor cannot produce an infinite scale (same rule as gemm_bf16x3.hip's pow2_scale)
__device__ __forceinline__ float pow2_scale_of(float m) {
    const int e = (int)((__float_as_uint(m) >> 23) & 0xFFu);
    int se = 261 - e;
    se = se < 1 ? 1 : (se > 187 ? 187 : se);
    return __uint_as_float((unsigned)se << 23);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence over ALL memory, for which
// hipcc emits s_waitcnt vmcnt(0): every barrier then also waits for the outstanding global stores / loads of the wave
// (~1 us each trip).  In the recurrence kernels the step barriers only hand LDS data between waves; what leaves for
// global memory is fire-and-forget (or is waited for explicitly where a hand-off needs it).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// (1 - lam) * a + lam * b with the reference's three roundings (model.py:430: two products, one sum, no fused multiply-add).  HIP's
// __fmul_rn / __fadd_rn are the plain operators, which hipcc is free to contract into an fma after inlining (-ffp-contract=fast is its
// default; seen in gn_relu_gather_kernel); the pragma takes the contract flag off these three operations wherever they end up.
__device__ __forceinline__ float ss_lerp_rn(float ol, float a, float l, float b) {
#pragma clang fp contract(off)
    const float p = ol * a;
    const float q = l * b;
    return p + q;
}
// Four consecutive values -> packed fp16 pieces of 16 x the values, (h0 h1, h2 h3, l0 l1, l2 l3), h = fp16(16 v) to nearest,
// l = fp16(16 v - h) -- exactly what the GEMM's in-loop split produces with its fixed scale.  ss_store_group puts them into an image.
__device__ __forceinline__ uint4 ss_split_group(float v0, float v1, float v2, float v3) {
    const float s = 16.0f;
    uint4 r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r.x) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r.x) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r.y) : "v"(v2), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r.y) : "v"(v3), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r.z) : "v"(v0), "v"(s), "v"(r.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r.z) : "v"(v1), "v"(s), "v"(r.x));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r.w) : "v"(v2), "v"(s), "v"(r.y));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r.w) : "v"(v3), "v"(s), "v"(r.y));
    return r;
}
// Store such a group into a v2 image: `at` is the address the four fp32 values have in a tensor of the image's geometry (image base 32-byte
// aligned, row strides multiples of 8 elements): the hi pieces go to the first 16 bytes of the enclosing group of eight, the lo pieces to
// the second, each at the half this group of four belongs to.
__device__ __forceinline__ void ss_store_group(float* at, uint4 g) {
    const size_t a = (size_t)at;
    char* base = (char*)(a & ~(size_t)31);
    const int half = (int)((a >> 4) & 1) * 8;
    *reinterpret_cast<uint2*>(base + half) = make_uint2(g.x, g.y);
    *reinterpret_cast<uint2*>(base + 16 + half) = make_uint2(g.z, g.w);
}
// Four consecutive values into an image at ELEMENT index `elem` from the image's base.  bf16 == 0: format v2 (fp16 x 2 pieces of scale * v);
// bf16 != 0 (the 16-bit data path, SS_PRECISION_BF16): the image is the plain bf16 tensor -- 8 bytes at 2 * elem, round to nearest even
// (v_cvt_pk_bf16_f32; a NaN stays a NaN), no scale.
__device__ __forceinline__ uint4 ss_split_group_s(float v0, float v1, float v2, float v3, float s);
__device__ __forceinline__ void ss_store_group(float* at, uint4 g);
__device__ __forceinline__ uint2 ss_pack_bf16x4(float v0, float v1, float v2, float v3) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const bf2_t a = __builtin_convertvector(f2_t{v0, v1}, bf2_t), b = __builtin_convertvector(f2_t{v2, v3}, bf2_t);
    return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}
__device__ __forceinline__ void ss_store_img4(float* img, long elem, float v0, float v1, float v2, float v3, float scale, int bf16) {
    if (bf16) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(img) + 2 * elem) = ss_pack_bf16x4(v0, v1, v2, v3);
    else ss_store_group(img + elem, ss_split_group_s(v0, v1, v2, v3, scale));
}
// the same with a caller-supplied power-of-two scale
__device__ __forceinline__ uint4 ss_split_group_s(float v0, float v1, float v2, float v3, float s) {
    uint4 r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r.x) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r.x) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r.y) : "v"(v2), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r.y) : "v"(v3), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r.z) : "v"(v0), "v"(s), "v"(r.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r.z) : "v"(v1), "v"(s), "v"(r.x));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r.w) : "v"(v2), "v"(s), "v"(r.y));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r.w) : "v"(v3), "v"(s), "v"(r.y));
    return r;
}
// sigmoid: 1 / (1 + 2^(-x log2 e)) on the hardware exp2 / rcp (1 ulp each).  The product x * log2(e) rounded to fp32 puts a RELATIVE error of
// |x| * 9e-8 into e^-x; relative to the sigmoid that is weighted by e^-x / (1 + e^-x), so the ABSOLUTE error stays below 2 ulp of the result's
// scale everywhere (the tails where the relative error of e^-x is largest are the tails where the gate is ~0 or ~1) -- a compensated exponent
// and a Newton step on the reciprocal were measured (round 4): +0.1 us per forward recurrence step for nothing a gradient could see.
__device__ __forceinline__ float ss_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504f));
}
// The cell-candidate gate g = tanh(pre-activation) as 2 sigmoid(2x) - 1: one exp2 and one rcp like the other three gates.  Its ABSOLUTE error
// (~1.5e-7) is what matters for g -- it enters the cell state as i * g beside terms of magnitude ~1 -- unlike tanh(c) below, whose RELATIVE error
// at small |c| becomes the relative error of h.  `two` selects the form per lane: 2.0f for the g gate, 1.0f for a sigmoid gate (one code path
// for the single-wave kernels whose lanes hold different gates).
__device__ __forceinline__ float ss_gate(float x, float two) {
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * two * -1.44269504f));
    return __builtin_fmaf(two, r, 1.0f - two);            // two = 1: r;  two = 2: 2 r - 1
}
// tanh to ~3 ulp RELATIVE (round 4; before: 3e-7 absolute, i.e. 1e-6 relative where |x| ~ 0.1 .. 0.5 -- cell states live there -- which is
// what put the engine's trained-state gradients 10x further from float64 than PyTorch's: the recurrences carry every rounding of their
// non-linearities forward).  |x| < 0.7: p = tanh(x / 2) from the odd series through x^11 (|x / 2| < 0.35: next term 1.2e-8 relative; Estrin
// form), then tanh(x) = 2 p / (1 + p^2) -- no cancellation anywhere; above: 1 - 2 / (e^2x + 1), whose subtraction amplifies the quotient's
// rounding by (1 - tanh) / tanh <= 0.65 there (e^2x = inf gives 1 - 0, e^2x = 0 gives 1 - 2).
__device__ __forceinline__ float ss_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.88539008f);
    const float q = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    const float a = 0.5f * x, y = a * a, y2 = y * y;
    const float p01 = __builtin_fmaf(y, -0.333333343f, 1.0f), p23 = __builtin_fmaf(y, -0.0539682545f, 0.13333334f),
                p45 = __builtin_fmaf(y, -0.00886323582f, 0.0218694881f);
    const float p = a * __builtin_fmaf(y2 * y2, p45, __builtin_fmaf(y2, p23, p01));
    const float t = (p + p) * __builtin_amdgcn_rcpf(__builtin_fmaf(p, p, 1.0f));
    return fabsf(x) < 0.7f ? t : q;
}
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace ss
