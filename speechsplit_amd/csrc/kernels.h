// Internal launcher prototypes for the SpeechSplit gfx950 kernels.
//
// Activation layout used by every kernel on the path ("haloed time-major"): [B, TP = T + 4, C] fp32, channels
// contiguous, real frame t at row t + 2, two all-zero rows on either side of every utterance.  The halo rows give
// the k=5 "same" convolution its zero padding, give the LSTM recurrences h(-1) = c(-1) = 0 without a branch, and
// let every weight-gradient contraction run as ONE GEMM over the flat row index b*TP + row.
#pragma once
#include "common.h"

namespace ss {

constexpr int HALO = 2;

// ---------------------------------------------------------------- interp.hip
struct InterpPlan {
    int S;          // segments per utterance (max_len_seq / min_len_seg + 1 = 7)
    int ncand;      // candidate positions per segment (2 * max_len_seg = 64)
    int P;          // output rows (max_len_pad)
    int T;          // input rows
    int* i0;        // [B, P]
    float* lam;     // [B, P]
    int* nrows;     // [B]   rows kept = min(count, P)
    int* counts;    // [B]   un-truncated count (model.py:418)
    int* start;     // [B, T + 1] inverse map for the backward
};
hipError_t interp_plan(const InterpPlan& p, const float* scales, const int* len_seg, const int* len_seq,
                       int len_seq_const, int B, hipStream_t s);
// y_img (nullable): also write the pre-split image of y (GemmDesc::a_pre), same geometry as y
// img_scale (nullable: 16): device word with the power-of-two scale the image is split with (act_scales)
hipError_t interp_gather(const InterpPlan& p, const float* x, long x_ld, long x_bs, float* y, long y_ld, long y_bs, int C,
                         int B, hipStream_t s, float* y_img = nullptr, const float* img_scale = nullptr);
hipError_t interp_quant(const InterpPlan& p, const float* mel, const float* f0, int CM, float* ymel, long ym_ld, long ym_bs,
                        float* yoh, long yo_ld, long yo_bs, int NOH, int* qidx, int B, hipStream_t s);
hipError_t interp_scatter(const InterpPlan& p, const float* dy, long dy_ld, long dy_bs, float* dx, long dx_ld, long dx_bs,
                          int C, int B, hipStream_t s);

// ---------------------------------------------------------------- elementwise.hip
// GroupNorm(16 channels per group, eps 1e-5, biased variance over 16 x T) + ReLU on rows [HALO, HALO+T) of haloed slabs.
hipError_t gn_relu_fwd(const float* x, long x_ld, long x_bs, float* y, long y_ld, long y_bs, const float* gamma,
                       const float* beta, float* stats /*[B, C/16, 2] mean, rstd*/, int B, int T, int C, hipStream_t s);
// the same followed by the training forward's random resampling of the block output (interp_gather), in one pass: y / y_img are the
// resampled slab and its image AT the first real row and the block's first column (p.P output rows); bit-identical to the two kernels
// img_bf16: y_img is the plain bf16 tensor (element offsets, 2 bytes each) instead of a format-v2 image (common.h ss_store_img4)
hipError_t gn_relu_gather(const float* x, long x_ld, long x_bs, float* y, long y_ld, long y_bs, float* y_img, const float* img_scale,
                          const float* gamma, const float* beta, float* stats, const InterpPlan& p, int B, int T, int C, hipStream_t s, int img_bf16 = 0);
// dy (grad of the ReLU output) is replaced in place by the grad of the GroupNorm input (= conv output).
// g_gamma / g_beta / g_bias [C]: every utterance's d_gamma, d_beta, d_convbias are ACCUMULATED here (f32 atomics).
// amax (nullable): receives max |conv-output gradient| written, as for lstm_seq_bwd.
// part (nullable): [B][3][C] scratch; in deterministic mode the per-utterance sums go there and are added in utterance order.
// scatter / src (nullable): take the adjoint of the training forward's gather on the fly from src (the gradient of the resampled output, at its
// first real row and the block's first column) instead of reading dy (interp_scatter fused in; dy is then only written)
hipError_t gn_relu_bwd(const float* x, long x_ld, long x_bs, float* dy, long dy_ld, long dy_bs, const float* gamma,
                       const float* beta, const float* stats, float* g_gamma, float* g_beta, float* g_bias, float* amax, float* part,
                       int B, int T, int C, hipStream_t s,
                       const InterpPlan* scatter = nullptr, const float* src = nullptr, long src_ld = 0, long src_bs = 0,
                       float* dy_img = nullptr);      // dy_img: dy also as a plain bf16 tensor (same geometry; halo rows are the caller's: zero)
// test hook: mask [B, T, C] dense = 1.0f where the block's GroupNorm output is > 0 (the ReLU branch the kernels above take)
hipError_t gn_relu_mask(const float* x, long x_ld, long x_bs, const float* gamma, const float* beta, const float* stats,
                        float* mask, int B, int T, int C, hipStream_t s);
// out[c] += sum_r in[r*ld + c], float64 accumulation in a fixed order (elementwise.hip).  part / ctr (nullable): scratch of
// colsum_scratch_doubles(columns) float64 words and cdiv(columns, 64) zeroed counters (left zero again); without them one workgroup per 64 columns
long colsum_scratch_doubles(int cols);
hipError_t colsum_acc(const float* in, long ld, int R, int C, float* out, double* part, unsigned* ctr, hipStream_t s);
// column sums of a BLSTM layer's [R][2 x C] gradient slab added to (b_ih, b_hh) of both directions
hipError_t colsum_bias(const float* in, long ld, int R, int C, float* bih0, float* bhh0, float* bih1, float* bhh1, double* part, unsigned* ctr,
                       hipStream_t s);
hipError_t copy_rows(const float* src, long s_ld, long s_bs, float* dst, long d_ld, long d_bs, int B, int T, int C,
                     hipStream_t s);
// batch assembly from a device-resident corpus: see collate_kernel (crop rows, clip mel to [0,1], pad mel with 0 / F0 with -1e10)
hipError_t collate(const float* mel_cat, const float* f0_cat, const float* emb_tab, const long* row0, const int* len,
                   const int* item, int B, int T, int C, int E, float* mel, float* f0, float* emb, hipStream_t s);
// conv weight [Co][Ci][5] -> forward pack [Co][5][Cp] (zero-filled for ci >= Ci) and input-grad pack [Ci][5][Co] (taps flipped)
// wf_img / wb_img (nullable): pre-split images of wf / wb (Cp % 4 == 0, Co % 4 == 0)
hipError_t conv_pack(const float* w, int Co, int Ci, int Cp, float* wf, float* wb, float* wf_img, float* wb_img, hipStream_t s, int img_bf16 = 0);
// packed weight grad [Co][5][Cp] -> grad arena [Co][Ci][5] (overwrite)
// every conv block's per-step weight re-layout in ONE launch (seven launches of 6 - 17 us were a chain the trunk's second layer waited for)
struct ConvPackTask {
    const float* w;
    float *wf, *wb, *wf_img, *wb_img;
    int Co, Ci, Cp;
};
struct ConvPackTable {
    ConvPackTask t[8];
    int n;
    int img_bf16;
};
hipError_t conv_pack_many(const ConvPackTable& tb, hipStream_t s);
hipError_t conv_unpack_grad(const float* gp, int Co, int Ci, int Cp, float* g, hipStream_t s);
constexpr int CONV_UNPACK_MAX = 8;
struct ConvUnpackTask {
    const float* gp;      // packed gradient [Co][5][Cp]
    float* g;             // parameter gradient [Co][Ci][5]
    int Co, Ci, Cp;
};
struct ConvUnpackTable {
    ConvUnpackTask t[CONV_UNPACK_MAX];
    int n;
};
hipError_t conv_unpack_grads(const ConvUnpackTable& tb, hipStream_t s);      // several blocks, one launch
hipError_t transpose2d(const float* in, int R, int C, float* out, hipStream_t s);   // out[c][r] = in[r][c]
// out[i] = a[i] + b[i]
hipError_t add_vec(const float* a, const float* b, float* out, int n, hipStream_t s);
// dst = a + b (b null: copy) for a table of up to PREP_MAX vectors, one launch
constexpr int PREP_MAX = 48;
struct PrepTask {
    const float* a;
    const float* b;
    float* dst;
    long n;
    float* img;           // nullable: the pre-split image of dst (GemmDesc::b_pre), written beside it; needs n % 4 == 0 and 16-byte alignment
};
struct PrepTable {
    PrepTask t[PREP_MAX];
    int n;
    int img_bf16;         // the tasks' images are plain bf16 tensors instead of format v2
};
hipError_t prep_run(const PrepTable& tb, hipStream_t s);

struct CodeSrc {          // one encoder BLSTM output feeding the decoder input (model.py:87, 223-227, 301-309)
    const float* o;       // [B, TP, 2*H]
    float* d_o;           // gradient slab of the same shape (backward only)
    int H, freq, col;     // col: first column inside the decoder input
};
hipError_t build_dec_in(const CodeSrc* src, int nsrc, const float* emb, int emb_dim, int emb_col, float* dec_in, int ld,
                        int B, int T, hipStream_t s);
hipError_t dec_in_grad(const CodeSrc* src, int nsrc, const float* d_dec_in, int ld, int B, int T, hipStream_t s);
// compact forms for a decoder input that repeats in blocks of f frames (all codes up-sampled by the same f): one row per block,
// xc / d_xc [B][T/f][ld]; d_xc holds the gradient already summed over each block's frames
hipError_t build_dec_in_compact(const CodeSrc* src, int nsrc, const float* emb, int emb_dim, int emb_col, float* xc, int ld, int B, int T,
                                int f, hipStream_t s);
hipError_t dec_in_grad_compact(const CodeSrc* src, int nsrc, const float* d_xc, int ld, int B, int T, int f, hipStream_t s);

// loss = mean((tgt - out)^2) over B*T*C real elements (solver.py:166); d_out = 2 (out - tgt) / N * scale
hipError_t mse_loss(const float* out, long o_ld, long o_bs, const float* tgt, long t_ld, long t_bs, float* d_out,
                    long d_ld, long d_bs, int B, int T, int C, float grad_scale, float* partials, float* loss,
                    hipStream_t s);
// softmax cross-entropy over C classes against integer targets; mean over B*T rows
hipError_t ce_loss(const float* logits, long o_ld, long o_bs, const int* tgt, float* d_out, long d_ld, long d_bs, int B,
                   int T, int C, float grad_scale, float* partials, float* loss, hipStream_t s);

struct AdamState {        // device-resident so a captured graph can replay the step
    double lr, beta1, beta2, eps;
    long step;
    float step_size, bc2_sqrt, f_beta1, f_beta2, f_eps;
    unsigned skip;        // set by adam_prepare when this step's gradients must not be applied (see adam_step)
};
// Engine status word ("sticky": host-visible, survives steps, cleared only by ss_clear_abort): bit 0 a persistent recurrence kernel's
// bounded wait expired on this rank, bit 1 another rank reported it (data parallel), bit 2 a parameter left the range the
// fixed-scale fp16 x 2 forward products are valid for (or is not finite).
constexpr unsigned SS_STICKY_ABORT = 1u, SS_STICKY_REMOTE = 2u, SS_STICKY_RANGE = 4u;
// sticky (nullable) / status (nullable: the gradient arena's status slot, summed over the ranks by the all-reduce): when
// either is non-zero the update is SKIPPED -- parameters, moments and the step counter stay as they are.
hipError_t adam_prepare(AdamState* st, unsigned* sticky, const float* status, hipStream_t s);
hipError_t adam_range(float* p, const float* g, float* m, float* v, long n, AdamState* st, float grad_scale, hipStream_t s);
hipError_t adam_step(float* p, const float* g, float* m, float* v, long n, AdamState* st, float grad_scale, unsigned* sticky,
                     const float* status, hipStream_t s);
// *status = (*sticky != 0)   (one thread; enqueued behind the decoder's recurrences, in front of the all-reduce that sums it)
hipError_t status_publish(const unsigned* sticky, float* status, hipStream_t s);
// Scale of the fp16 x 2 split for the OUTPUT of each conv block (GroupNorm + ReLU, then resampled: a convex combination), from its affine
// parameters: |y| <= sqrt(16 T) max|gamma| + max|beta| = bound; out[i] = min(16, largest power of two with bound * scale <= 32768).
// 16 -- the scale every other forward operand uses -- whenever bound <= 2048, i.e. for any sane GroupNorm affine.
constexpr int ACT_SCALE_MAX = 8;
struct ActScaleTable {
    const float* gamma[ACT_SCALE_MAX];
    const float* beta[ACT_SCALE_MAX];
    int C[ACT_SCALE_MAX];
    int n;
};
hipError_t act_scales(const ActScaleTable& tb, int T, float* out, hipStream_t s);
// sticky |= SS_STICKY_RANGE if any of the n parameters is not finite or |p| >= limit
hipError_t param_guard(const float* p, long n, float limit, unsigned* sticky, hipStream_t s);

// ---------------------------------------------------------------- features.hip  (offline feature extraction, float64)
// x [n] -> S [frames][n_mels] float32, frames = (n + 256) / 256 (utils.py:18-31, make_spect_f0.py:57-60); mel [513][n_mels]
hipError_t melspec(const double* x, int n, const double* mel, int n_mels, float* out, int frames, hipStream_t s);
// utils.py:35-42 with mean / std over the voiced frames (make_spect_f0.py:64-66); -1e10 marks unvoiced frames
hipError_t f0_normalize(const double* f0, int n, float* out, hipStream_t s);

// ---------------------------------------------------------------- lstm_small.hip  (hidden <= 32: whole recurrence in one launch)
// gates: [B, TP, 8H] holds x.W_ih^T + b_ih + b_hh on entry (column = dir*4H + gate*H + j, gate order i,f,g,o) and the
// activated gates on exit.  out: [B, TP, 2H].  csave: [B, TP, 2H] cell states.  whh: [2][4H][H].
hipError_t lstm_small_fwd(float* gates, const float* whh_f, const float* whh_b, float* out, float* csave, int B, int T,
                          int H, hipStream_t s);
// d_out: [B, TP, 2H] gradient of out.  gates is replaced in place by the pre-activation gradients.
hipError_t lstm_small_bwd(float* gates, const float* whh_f, const float* whh_b, const float* d_out, const float* csave,
                          int B, int T, int H, hipStream_t s);

// ---------------------------------------------------------------- lstm_step.hip  (hidden % 64 == 0: one launch per time step)
// MFMA operands are streamed from fragment-major copies (see lstm_step.hip):
//   wfrag  [2][H/16][4][H/16][64][4]   W_hh for the forward step        (lstm_pack_w, transposed = 0)
//   wfragT [2][H/16][4H/16][64][4]     W_hh^T for the backward step     (lstm_pack_w, transposed = 1)
//   hf     [2 ping-pong][2][ceil(B/16)][H/16][64][4]    h(t)   written by the forward epilogue, zero before step 0
//   gf     [2 ping-pong][2][ceil(B/16)][4H/16][64][4]   da(t)  written by the backward epilogue, zero before step 0
hipError_t lstm_pack_w(const float* whh_f, const float* whh_b, float* frag, int H, int transposed, hipStream_t s);
hipError_t lstm_step_fwd(float* gates, const float* wfrag, const float* hf_cur, float* hf_next, float* out, float* csave,
                         int B, int T, int H, int step, hipStream_t s);
// dc: [2][B][H] running cell-state gradient (no initialisation needed).
hipError_t lstm_step_bwd(float* gates, const float* wfragT, const float* gf_cur, float* gf_next, const float* d_out,
                         const float* csave, float* dc, int B, int T, int H, int step, hipStream_t s);

// ---------------------------------------------------------------- lstm_seq.hip  (persistent: one launch per layer)
// gates / out / csave / d_out as above; whh_* are the parameter tensors themselves ([4H][H] row-major): each workgroup
// splits its slice into fp16 x 2 pieces once and keeps it in registers.  xbuf = lstm_seq_xbytes() bytes of exchange buffer
// (forward: h(t) as fp16 pieces in MFMA fragment order; backward: partial-dh tiles, whose dwords carry their step's tag in bit 0),
// sync = LSTM_SEQ_SYNC_WORDS unsigned words (completion flags, XCD masks, abort word at [0]); both must be all zero at launch:
// zero_state = false means the caller has zeroed them itself.  time_major: the slabs are [T+4][B][C] instead of [B][T+4][C].
constexpr int LSTM_SEQ_SYNC_WORDS = 2048;   // [0] abort, [1..) XCD masks per group, [64 + 32*group + member] completion flags
bool lstm_seq_supported(int B, int H);
long lstm_seq_xbytes(int B, int H, bool backward);
// out_img (fwd, nullable): pre-split image of `out` (GemmDesc::a_pre / b_pre), same shape
// xc / xf (fwd), dgs / xf (bwd), nullable: a layer whose input repeats in blocks of xf frames -- input projections given once per block
// [B][T/xf][8H]; pre-activation gradients additionally written summed per block [B][T/xf][8H]
// sticky (nullable): engine-wide word, host-visible, that a launch ORs 1 into when its bounded wait expires (never cleared by a step)
hipError_t lstm_seq_fwd(float* gates, const float* whh_f, const float* whh_b, void* xbuf, float* out, float* csave,
                        unsigned* sync, unsigned* sticky, const float* xc, int xf, float* out_img, int B, int T, int H, bool zero_state,
                        bool time_major, hipStream_t s, int img_bf16 = 0);       // img_bf16 bit 0: out_img is the plain bf16 tensor, not a format-v2 image; bit 1: products from the high fp16 pieces alone
// amax (nullable): device word that receives max |pre-activation gradient| written (atomic max of the float's bit pattern;
// zero it first) -- the scale the fp16 x 2 GEMMs that consume the gradient slab need
// gbias_f / gbias_b (nullable): [2][4H] gradient accumulators of (b_ih, b_hh) of the forward / reverse direction; the kernel
// adds the sum over utterances and time of the pre-activation gradients to both halves (f32 atomics)
hipError_t lstm_seq_bwd(float* gates, const float* whh_f, const float* whh_b, void* xbuf, const float* d_out,
                        const float* csave, unsigned* sync, unsigned* sticky, float* amax, float* gbias_f, float* gbias_b, float* dgs, int xf,
                        int B, int T, int H, bool zero_state, bool time_major, hipStream_t s, float* dimg = nullptr, int hi = 0);      // dimg: the gradients also as a plain bf16 tensor (slab geometry); hi bit 0: products from the high fp16 pieces alone, bit 2 (with dimg): the gradients ONLY as the bf16 tensor

// ---- lstm_wgrad.hip: weight and bias gradients of the encoder BLSTMs (H <= 32), every layer of every block in one launch
constexpr int WGRAD_MAX = 8;
struct WgradTask {
    const float* dG;          // pre-activation gradients [R][8H] (halo rows zero)
    const float* X;           // the layer's input rows [R][In], row stride x_ld
    long x_ld;
    const float* Hout;        // the layer's output [R][2H] (halo rows zero)
    float *gwih0, *gwih1;     // += dW_ih of the forward / reverse direction [4H][In]
    float *gwhh0, *gwhh1;     // += dW_hh [4H][H]
    float *gbih0, *gbhh0, *gbih1, *gbhh1;      // += bias gradients [4H] (b_ih and b_hh have the same gradient)
    int H, In;
    long R;
    int tile0;                // first blockIdx.y of this task: lstm_small_wgrad_tiles(H, In) tiles each
};
struct WgradTable {
    WgradTask t[WGRAD_MAX];
    int n, tiles_total, row_groups;
    float* part;              // scratch: tiles_total * row_groups * 4096 floats
    unsigned* ctr;            // tiles_total arrival counters, zero at rest
};
int lstm_small_wgrad_tiles(int H, int In);
hipError_t lstm_small_wgrad(const WgradTable& tb, hipStream_t s);

// seq_gate: the stream goes on once the persistent recurrence that owns `sync` is resident (all groups through round 0), see lstm_seq.hip;
// lstm_seq_free_xcds: how many of the 8 XCDs such a launch leaves free (0: none, or not a persistent shape)
hipError_t seq_gate(const unsigned* sync, int B, int H, hipStream_t s);
int lstm_seq_free_xcds(int B, int H);

// streaming pre-read (results unused) of the slabs a persistent recurrence is about to consume: wide [rows][cw] (gates) and one or two
// narrow ones [rows][cn] (cell states; output gradient), both ends of the sequence first.  Meant for a side stream, beside the recurrence.
hipError_t slab_prewarm(const float* wide, int cw, const float* n0, const float* n1, int cn, float* sink, int B, int T, bool time_major, hipStream_t s);

}  // namespace ss
