// Host-side engine: parameter table, workspace plan and the forward / backward / training-step schedules of
// Generator_3 and Generator_6 over the kernels in this directory, behind the C ABI of include/speechsplit_amd.h.
//
// Mirrors (does not translate) reference model.py:46-351 and solver.py:157-172: same operators in the same
// order, but activations live in haloed time-major slabs (kernels.h), the three InterpLnr calls of Encoder_7
// operate on one fused 768-channel slab, channel concats / splits / transposes are pointer arithmetic, and
// every contraction is the one fp32 MFMA GEMM.
#include <dlfcn.h>
#include <map>
#include <string>
#include <vector>
#include <algorithm>
#include <utility>
#include <functional>
#include <cstring>
#include <cstdio>

#include "common.h"
#include "kernels.h"
#include "../../include/speechsplit_amd.h"

using namespace ss;

namespace ss {
extern int g_small_lds, g_small_prio, g_gemm_tr, g_deterministic, g_gn_part;
extern int g_img_cfg;
extern int g_lstm_nw, g_lstm_g, g_lstm_mode, g_gemm_bk, g_gemm_want, g_gemm_diag, g_seq_prio, g_gemm_mode, g_seq_spin_log2, g_seq_tag, g_seq_wlead, g_seq_var, g_gemm_ws;
int g_fwd_f16x2 = 1;   // 1: forward contractions (operands bounded by construction: mel, one-hot, GroupNorm/ReLU outputs, |h| < 1, weights)
                       //    use the fp16 x 2 split (3 MFMAs) instead of bf16 x 3 (6 MFMAs); gradients keep bf16 x 3 (their range is not bounded)
int g_bwd_f16x2 = 1;   // 1: the decoder's and the conv trunk's gradient GEMMs also use fp16 x 2: the gradient operand is scaled by the power
                       //    of two its producer kernel measured (max |value| of the slab), the activation / weight operand by the fixed one
int g_overlap = 1;     // 1: weight-gradient GEMMs on the side stream
int g_dx_batched = 2;  // input-gradient GEMMs per utterance without halo rows; 2: with 128 x 128 tiles from 512 workgroups on
int g_conv_small_old = 1;  // see try_img_gemm
int g_conv_want = 256;    // experiment: workgroup target of the conv GEMMs' tile choice (0: library default)
int g_defer_dw = 1;    // 1: the decoder's weight-gradient GEMMs start after its last input gradient (see lstm_bwd)
int g_side_prio = 0;   // 1: create the side stream with the lowest priority (read at ss_bind).  Measured: 2.3x SLOWER
                       //    (35 ms vs 14.8 ms per step): the low-priority queue starves behind 768 tiny step launches.
int g_branch_low = 0;      // experiment: the probed branch streams are created with the lowest priority (read at ss_bind); measured 5.78 vs 5.80 ms, off
int g_trunk_indep = 1;     // Encoder_7 forward: content and pitch conv stacks run as two INDEPENDENT chains (they share only the resampling plans)
int g_img = 1;             // 1: contractions whose two operands exist as images run on the image GEMM (gemm_img.hip); 0: round 2's kernels only
int g_img_mask = (1 << SS_PROF_DEC_PROJ) | (1 << SS_PROF_CONV_FWD) | (1 << SS_PROF_CONV_DX);     // ... per profile class (bit SS_PROF_*): which classes may take the image GEMM (A/B runs)
int g_img_batch = 1;       // image GEMM: the decoder's weight gradients of both directions in one launch per matrix
int g_img_dw_cfg = -1;     // experiment: tile configuration of the split-K (weight-gradient) image GEMMs (-1: the rule in try_img_gemm)
int g_img_dw_wgs = 256;    // ... and the number of workgroups their split aims at
int g_dp_emulate = 0;      // with dp_model = N: the stand-in collectives multiply their range by N (the sum of N identical ranks), see dp_scale_kernel
int g_dp_model = 0;        // > 1: MODEL a data-parallel run of that many ranks on one GPU: every collective is replaced by a stand-in kernel of
                           // the modelled duration (tools/dp_timeline.sh); no communicator needed
int g_dp_buckets = 1;      // 1: per-layer gradient buckets on the communication stream; 0: round 2's two buckets
int g_cur_klass = -1;      // profile class of the contraction being launched (set by the PGEMM macros)
int g_bf16_img = 1;        // SS_PRECISION_BF16: the 16-bit data path (round 4) -- operand images are plain bf16 tensors written by their producers (weights,
                           // hidden states, resampled activations, pre-activation / conv-output gradients) and the contractions over them run on the
                           // single-piece form of the image GEMM; 0: round 3's bf16 mode (fp32 slabs, operands rounded inside the GEMM)
int g_seq_skip32 = 1;      // ... and a decoder layer whose gradient consumers all read the bf16 tensor does not get the fp32 copy of its pre-activation gradients
int g_seq_hi = 1;          // ... and the persistent recurrences multiply the high fp16 pieces only (one MFMA per product, half the forward's payload)
int g_bf16_img_mask = ~0;  // ... per profile class (bit SS_PROF_*), for A/B runs
int g_wgrad_fused = 1;     // the encoder BLSTMs' weight and bias gradients (H <= 32) in one fused fp32 launch (lstm_wgrad.hip) instead of 12-14 tiny GEMMs + column sums
int g_trunk_bwd_par = 1;   // Encoder_7 backward: the pitch conv stack's blocks stay on the stream of lstm_2's backward (second branch stream), beside the content
                           // stack on the main stream, through all three layers (the two stacks are independent chains; round 3 did this for layer 0 only).
                           // 64 x 128 unchanged (that phase is throughput-bound), 32 x 128 bf16 3.16 -> 3.02 ms, 16 x 128 fp32 3.44 -> 3.29.
                           // (Measured and rejected beside it: each block's weight-gradient GEMM on the side stream beside its input-gradient GEMM --
                           // +8 % at B <= 32: the two cross-stream event hops per block cost more than the overlap gains.)
int g_pack_one = 1;        // every conv block's per-step weight re-layout in one launch at the start of the forward (conv_pack_many)
int g_presplit = 7;        // weights (and, bit 1, the decoder's hidden states) reach the fp16 x 2 GEMMs as pre-split images: bit 0 weights, bit 1 the decoder's hidden states, bit 2 the trunk's resampled activations
int g_compact0 = 1;        // decoder layer 0: input projections, input gradient and W_ih gradient once per block of repeated input frames
int g_batch_dirs = 1;      // BLSTM weight gradients: both directions of a layer in one launch per matrix (batch = 2) + one bias kernel:
                           // 1 = the encoder BLSTMs (36 -> 14 launches), 2 = the decoder too (measured: step +0.18 ms), 0 = never
int g_prewarm = 2;         // streaming pre-read of a decoder layer's operand slabs on a side stream beside its persistent recurrence: bit 1 forward
                           // (step -0.08 ms), bit 0 backward (no gain in the step, off; neither one recurrence ahead: +0.08 ms)
int g_op_time_major = 0;   // experiment: ss_op_lstm_fwd / _bwd take time-major slabs [T+4, B, C] (persistent kernels only)
int g_persist = 1;     // 1: decoder recurrences run as ONE persistent launch per layer (lstm_seq.hip) when the batch fits
int g_split = 0;       // 1: decoder recurrences run as two batch-half chains on two streams (GEMMs of one half fill the
                       //    machine while the other half sits in its latency-bound time loop)
// (hipGraph capture + replay of the fused step: built and measured in rounds 1-3 -- no gain while the step is GPU-bound, and a crash inside
// the ROCm 7.2 runtime's hipGraphLaunch with more parallel branches -- and deleted in round 4; the step is enqueued eagerly.)
int g_own_streams = 0; // 1: every C-ABI call runs on the ENGINE's own main stream (created back to back with its three branch streams at
                       //    ss_bind), ordered behind the caller's stream on entry and in front of it on exit.  Built to make the step time
                       //    independent of how many streams the process created earlier -- it does not: HIP hands a new stream the
                       //    least-loaded of its 4 hardware queues, and which engine stream ends up sharing a queue with the caller's still
                       //    moves the step by 5 % either way (profiles/r02/stream_order_effect.txt: 6.70 - 7.13 ms owned, 6.69 - 7.38 not).
                       //    Off by default until the engine can measure and pick its queue placement.
int g_flat_rows = 1;   // 1: batched-per-utterance GEMMs run flat over the slab rows when T % 128 != 0 (flatten_rows)
int g_prio_order = 1;  // 1: within every phase the critical-path launches are ENQUEUED first and the work that only has to be done by the end of
                       //    the step (decoder / encoder-BLSTM weight gradients, Encoder_t) last.  HIP multiplexes streams onto 4 in-order hardware
                       //    queues; when two engine streams share one (other streams in the process, e.g. RCCL's, shift the assignment), enqueue
                       //    order is execution order, and filler work enqueued first would run in front of the critical path.
int g_probe_queues = 1; // 1: ss_bind measures which candidate streams share a hardware queue and picks branch streams that do not (pick_streams)
static unsigned* g_img_wq = nullptr;      // queue words (+ placement log) of the test hook's work-queue launches
constexpr long IMG_WQ_BYTES = 16 + 16 * 1024;
int g_part_splitk = 1;  // split-K weight gradients of the in-loop-split kernel through partial slabs + an ordered reduce instead of fp32 atomics
int g_unpack_later = 1; // one-GPU step: the conv weight gradients' re-layouts in one launch at the end of the backward
int g_gn_gather = 1;    // training forward of the independent trunk chains: GroupNorm + ReLU + resampling gather in one kernel (gn_relu_gather)
int g_enc_t_first = 1;      // prio schedule, third branch stream: Encoder_t's backward chain (waits for dec_in_grad only) in FRONT of the encoder BLSTMs' weight
                        // gradients, which then go out in ONE fused launch with Encoder_t's.  Generator_6 32 x 192 bf16 2.26 -> 2.23 ms, 16 x 128 3.20 -> 3.18,
                        // headline unchanged
int g_conv_dw_off = 1;      // Generator_6 (one trunk chain, the second branch stream idle behind lstm's backward): the conv blocks' weight-gradient GEMMs leave
                        // the dependent chain gn backward -> input gradient -> next block for that stream; every block then keeps its own conv-output
                        // gradient slab (d_act_l).  With enc_t_first: 2.26 -> 2.17 ms (bf16), 2.50 -> 2.47 (fp32).  Generator_3 has no idle stream there
                        // (tools/real_timeline.py: its four streams end within ~100 us of each other)
int g_dec_tail_split = 9;   // one-GPU step: on the side stream alone the decoder's twelve weight-gradient GEMMs end ~450 us after every other stream
                        // (tools/real_timeline.py).  Layer 0's and the head's leave it for 1: the pitch chain's stream (behind the chain), 2: the third
                        // branch stream (in FRONT of its own work: they start with layer 2's), 3: the main stream (behind the trunk); 4-6 move layer 1
                        // as well (4: -> third, 5: layer 1 -> third + layer 0 -> pitch stream, 6: layer 1 -> pitch stream + layer 0 -> third); 7-10: as
                        // 2 plus single GEMMs of layer 1's reverse direction (7: dW_ih -> third, dW_hh -> pitch; 8: both -> third; 9: dW_hh -> third;
                        // 10: 9 + layer 2's reverse dW_hh -> third).  64 x 128 fp32: 5.16 / 5.05 / 5.04 / 5.16 / 5.03 / 5.12 / 5.14 ms for 0..6;
                        // 2 / 7 / 8 / 9 / 10 on another box: 5.07 / 5.06 / 5.07 / 5.01 / 5.01.  16 x 128: 3.26 -> 3.20 with 2 or 9; the 16-bit mode does
                        // not care (its weight gradients are batched per layer and its end of step is a dependent chain).  With 9 all four streams end
                        // within 200 us of each other under full contention.  Data parallel: layer 0 + head behind the pitch chain on the third stream
                        // +2.5-4 %; layer 0's forward direction + head at the very end of the third stream's and its reverse direction at the end of
                        // the main stream's enqueue order: 5.245 -> 5.226 ms at 64 x 128, +3-5 % at B <= 32: not done.
int g_early_dw = 0;     // 16-bit data path, B <= 48: decoder layer l + 1's weight gradients as work-queue image GEMMs on the XCDs the backward recurrence of
                        // layer l leaves free (lstm_bwd), instead of beside the encoder backward at the end of the step.  Off: 2.99 -> 2.97 ms at
                        // 32 x 128, 2.68 -> 2.67 at 16, 3.38 -> 3.35 at 48 (tools/real_timeline.py): a work-queue launch ends only when the workgroups
                        // parked on the recurrence's XCDs have run, i.e. with the recurrence, so ONE launch fits a recurrence and the layer's second
                        // one starts with the next recurrence (which it delays: 289 -> 357 us); the recurrence beside a launch stretches 283 -> 297 us;
                        // and the end of the step barely moves, because the encoder backward there is a dependent chain (~700 us at B = 32), not
                        // throughput the decoder's GEMMs were taking.
int g_xcd_dw = 0;       // decoder W_ih gradients beside the backward recurrences on the XCDs they leave free (B <= 48), see ss_engine::wq_pool.
                        // Off: measured 4.16 vs 4.09 ms at 32 x 128, 3.57 vs 3.53 at 16 x 128 -- the GEMM does run on the free XCDs beside the
                        // recurrence, but each recurrence stretches by ~40 us (its operand fetches share HBM with the GEMM's streams), the image
                        // pass and the split-K reduce land beside the input-gradient GEMM on the critical path, and at B <= 32 the encoder
                        // backward that loses the work is chain-bound, not throughput-bound (profiles/r03/xcd_overlap.txt)
int g_img_xcc = 0;      // ss_op_gemm_img (test hook): run the image GEMM in its work-queue form on the XCDs of this mask
int g_adam_early = 1;   // one-GPU fused steps: the decoder + head range of Adam beside the encoder backward (ss_tune("adam_early"))
int g_exp = 0;         // bits that switch individual schedule choices back for same-box A/B runs (bench.py --tune exp=N); 0 in production
int g_conv_par = 1;    // 1: the two conv streams of an Encoder_7 layer (and the layer's resampling plan) run on two engine streams in the forward
int g_early_join = 1;  // 1: join events of branch streams are recorded right behind the last kernel the consumer needs (lstm_bwd's dx_ready)
}

namespace {

thread_local std::string g_err;
int fail(const std::string& m) {
    g_err = m;
    return -1;
}
#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t _e = (x);                                                           \
        if (_e != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define CHK(x)                 \
    do {                       \
        int _r = (x);          \
        if (_r != 0) return _r; \
    } while (0)

struct ParamInfo {
    std::string name;
    long offset;
    int ndim;
    long shape[3];
    long numel() const { return shape[0] * (ndim > 1 ? shape[1] : 1) * (ndim > 2 ? shape[2] : 1); }
};

struct ConvBlk {
    int Ci = 0, Co = 0, Cp = 0;
    long w = 0, b = 0, ga = 0, be = 0;     // arena offsets
    float *wf_img = nullptr, *wb_img = nullptr;      // pre-split images of wf / wb (GemmDesc::b_pre)
    float *wf = nullptr, *wb = nullptr, *gp = nullptr, *cout = nullptr, *stats = nullptr, *part = nullptr;
    int amax_i = -1;                       // slot in ss_engine::amax
    int scale_i = -1;                      // slot in ss_engine::act_scale (scale of the fp16 x 2 split of this block's OUTPUT)
    bool need_dx = false;
    bool img_ok() const { return Cp % 8 == 0 && Co % 8 == 0; }       // wf_img / wb_img are written (rows of whole image groups)
};

struct LstmDir {
    long wih, whh, bih, bhh;
};

struct LstmBlk {
    int In = 0, H = 0, L = 0;
    std::vector<LstmDir> pd;               // [L*2]
    std::vector<float*> gates, out, csave; // per layer
    float* bsum = nullptr;                 // [L][2][4H]
    bool out_img_valid = false;            // the last forward wrote out_img
    std::vector<float*> out_img;           // per layer: pre-split image of out (decoder-size blocks on the persistent kernels)
    std::vector<float*> wcat_img;          // per layer: pre-split image of wcat (decoder-size blocks only)
    std::vector<float*> wcat;              // per layer: [W_ih forward ; W_ih reverse] stacked, [8H][In] (input-gradient GEMM over both directions)
    std::vector<float*> wfrag;             // per layer: fragment-major W_hh (forward) / W_hh^T (backward), 2*4H*H floats
    float* hf[2] = {nullptr, nullptr};     // per batch-half chain: ping-pong fragment-major h(t),  2 x [2][ceil16(B)][H]
    float* gf[2] = {nullptr, nullptr};     // per chain: ping-pong fragment-major da(t), 2 x [2][ceil16(B)][4H]
    float* dc[2] = {nullptr, nullptr};     // per chain: [2][B][H]
    unsigned* sync = nullptr;              // persistent kernels: group counters + abort word (one-off ops path)
    // persistent schedule: per-layer start state, contiguous so ONE memset per pass readies every layer's launch.
    //   zf = [L][LSTM_SEQ_SYNC_WORDS] ++ [L][hf]      (forward)        zb = [L][LSTM_SEQ_SYNC_WORDS] ++ [L][exchange tiles]   (backward)
    char *zf = nullptr, *zb = nullptr;
    long zf_bytes = 0, zb_bytes = 0, hf_bytes = 0, gf_bytes = 0;
    unsigned* sync_f(int l) const { return (unsigned*)zf + LSTM_SEQ_SYNC_WORDS * l; }
    unsigned* sync_b(int l) const { return (unsigned*)zb + LSTM_SEQ_SYNC_WORDS * l; }
    void* hf_l(int l) const { return zf + 4L * LSTM_SEQ_SYNC_WORDS * L + hf_bytes * l; }
    int amax0 = -1;                        // first slot in ss_engine::amax (one per layer) when the block's gradient GEMMs may use fp16 x 2
    // backward exchange tiles, one set per layer: the tagged hand-off (lstm_seq.hip) needs them zero at launch
    void* px_l(int l) const { return zb + 4L * LSTM_SEQ_SYNC_WORDS * L + gf_bytes * l; }
    float* dmid[2] = {nullptr, nullptr};   // gradient slabs of inner layer outputs [B,TP,2H]
    // Layer 0's input repeats in blocks of xf frames (decoder: every code is up-sampled by the same factor, the speaker row is constant;
    // xf = 0: not so / not used): compact input xc [B*T/xf][In], its projections xp0 [.][8H], block sums of the pre-activation gradients
    // dgs [.][8H] and the input gradient d_xc [.][In], one row per block
    int xf = 0, xcols = 0;                 // xcols: leading input columns whose gradient is needed (0: all)
    float *xc = nullptr, *xp0 = nullptr, *dgs = nullptr, *d_xc = nullptr;
    bool big() const { return H > 32; }
    // image of the stacked W_ih of layer l (written by lstm_prep when its rows are whole image groups)
    const float* wimg(int l) const { return (!wcat_img.empty() && wcat_img[l] && in_of(l) % 8 == 0) ? wcat_img[l] : nullptr; }
    int in_of(int l) const { return l == 0 ? In : 2 * H; }
};

struct Slab {   // view of a haloed slab: p points at slab row 0, first channel of interest
    float* p = nullptr;
    long ld = 0;
    const float* img = nullptr;      // the slab's pre-split image at the same position, if its producer wrote one (GemmDesc::a_pre / b_pre)
    const float* scale = nullptr;    // device word: power-of-two scale of the fp16 x 2 split of this slab's values (null: 16); conv-block outputs carry one
};

}  // namespace

struct ss_engine {
    int kind;
    ss_hparams hp;
    int maxB, maxT;
    int bound_max_len_pad = 0;             // hp.max_len_pad as given to ss_create: what a step WITHOUT SS_STEP_BUCKET runs with
    int precision = SS_PRECISION_F32;
    std::vector<ParamInfo> params;
    long arena = 0;                        // floats per arena, INCLUDING the 4-float status slot at the end
    long status_off = 0;                   // gradient arena: G[status_off] = 1 when this rank's step is invalid; the data-parallel
                                           // all-reduce sums it, so every rank's Adam kernel sees a non-zero value and skips
    void* comm = nullptr;                  // ncclComm_t of ss_comm_init (RCCL, dlopen'd)
    int comm_rank = 0, comm_world = 1;
    // data-parallel step in flight: gradient ranges are all-reduced on comm_s as soon as their producers are through (dp_bucket)
    bool dp_on = false;
    hipStream_t comm_s = nullptr;
    hipEvent_t ev_comm = nullptr;
    std::vector<std::pair<long, long>> dp_done;      // [offset, end) ranges already handed to a collective in this step
    // ss_dp_profile: hipEvent brackets round every collective of the LAST data-parallel step on the communication stream, plus one event at
    // the backward's end on the main stream -- where each bucket's all-reduce starts and ends relative to it (the overlap a scaling run achieved)
    bool dp_prof = false;
    struct DpRec {
        long off, count;
        hipEvent_t a, b;
    };
    std::vector<DpRec> dp_rec;             // this step's
    std::vector<hipEvent_t> dp_ev_pool;    // events are created once and reused
    int dp_ev_used = 0;
    hipEvent_t dp_bwd_end = nullptr;
    bool lockstep = false;                 // data-parallel member: entry points never refuse on the status word (entry_check)
    unsigned* sticky = nullptr;            // engine status word in host-coherent pinned memory (kernels.h SS_STICKY_*): written by
                                           // kernels, read by the host without synchronising; cleared only by ss_clear_abort

    float *P = nullptr, *G = nullptr, *Mm = nullptr, *Vv = nullptr;
    char* ws = nullptr;
    long ws_bytes = 0;
    int curB = 0, curT = 0;
    bool fwd_training = false;
    const float *late_org = nullptr, *late_emb = nullptr;   // fused training step: x_org / emb still to be copied in (done on the Encoder_t branch)
    bool dec_w_pending = false;            // backward_decoder(late): the decoder's + head's weight gradients are still to be enqueued
    bool dp_dir_buckets = false;           // lstm_weight_grads handed each direction of the layer to a collective itself
    ConvUnpackTable unpack{};              // conv weight gradients waiting for their re-layout (conv_block_bwd)
    bool unpack_later = false;
    // XCD-aware weight gradients (lstm_bwd, ss_tune("xcd_dw")): where the decoder's persistent backward recurrences leave XCDs free
    // (B <= 48: 2 * ceil(B / 16) groups, one XCD each), the W_ih gradient of layer l + 1 runs as a work-queue image GEMM BESIDE the
    // recurrence of layer l -- on the free XCDs, because its 128-144 KB workgroups cannot be dispatched to a CU a recurrence workgroup holds
    unsigned* wq_pool = nullptr;           // zeroed per backward pass: 4 words per work-queue launch
    // column sums (bias gradients): float64 chunk partials from the step's scratch (part) + a ring of arrival counters, zero at rest
    unsigned* colsum_ctr = nullptr;
    static constexpr int COLSUM_CTRS = 2048;
    int colsum_next = 0;
    // encoder-BLSTM weight gradients waiting for their fused launch (lstm_wgrad.hip): lstm_weight_grads appends, wgrad_flush launches
    WgradTable wg{};
    bool wg_defer = false;                 // backward_encoder collects every small block's layers and flushes once at the end
    int wq_next = 0;
    static constexpr int WQ_SLOTS = 16;
    int dec_w_done = 0;                    // bit l: ALL of decoder layer l's weight gradients went out beside a recurrence (early_dw)
    hipStream_t dw_over[2] = {nullptr, nullptr};   // lstm_weight_grads, unbatched decoder path: the reverse direction's dW_ih / dW_hh go to these streams (tail split)
    bool wq_mode = false;                  // lstm_weight_grads: image GEMMs in the work-queue form
    int dec_ih_done = 0;                   // bit l: decoder layer l's W_ih gradient went out beside a recurrence (lstm_late_weights skips it)
    // Early Adam (one GPU, Adam inside the step): the decoder + head range of the arenas (80 % of the bytes) is updated on the side stream
    // right behind its last weight-gradient GEMM, beside the encoder backward (GEMM-bound, HBM mostly idle), instead of at the step's end
    bool adam_early = false;               // this step wants it (set by the fused train steps)
    float adam_early_gs = 1.0f;
    long adam_early_from = -1;             // >= 0: the range [adam_early_from, arena) has been enqueued, the step state prepared
    bool prezero = false;                  // fused training step: zero the gradient arena on a branch stream during the forward
    bool grads_zeroed = false;             // ... done: backward_decoder must not zero it again
    bool bwd_sync_zeroed = false;          // the same for the backward recurrences' sync words / exchange tiles and the work-queue words
    bool have_fwd = false;
    int enc_plan0 = 0;                     // plan[] index of the first encoder InterpLnr call of the last forward

    // components
    ConvBlk c1[3], c2[3], ct;              // Encoder_7 stream 1 / stream 2 (or Encoder_6 in c2), Encoder_t
    LstmBlk l1, l2, lt, ld;                // lstm_1, lstm_2 (or Encoder_6.lstm), Encoder_t.lstm, decoder.lstm
    long head_w = 0, head_b = 0;
    int head_out = 0, dec_in_dim = 0, CE = 0;   // CE = channels of the fused encoder slab (768 for G3, 256 for G6)

    // workspace slabs
    float *in_mel = nullptr, *in_f0 = nullptr, *org = nullptr, *emb = nullptr;
    int f0p = 0;                           // padded one-hot width (264)
    float *act = nullptr, *d_act = nullptr, *d_xf = nullptr, *xf[3] = {nullptr, nullptr, nullptr};
    float* xf_img[3] = {nullptr, nullptr, nullptr};     // pre-split images of xf[0], xf[1] (training forward only: written by the gathers)
    bool xf_img_valid = false;                          // the last forward wrote them
    float *act_t = nullptr, *d_act_t = nullptr;
    float *dec_in = nullptr, *d_dec_in = nullptr, *d_top = nullptr;
    float *d_o1 = nullptr, *d_o2 = nullptr, *d_ot = nullptr;
    float *out_slab = nullptr, *d_out_slab = nullptr;
    float* gp_all = nullptr;               // packed conv weight-gradient images (all blocks)
    // gradient slabs as images for the image GEMM: written by split_image with the measured scale (gscale[i] beside amax[i])
    float* dg_img[3] = {nullptr, nullptr, nullptr};       // decoder layers' pre-activation gradients [B, TP, 8H]
    int dg32_skipped = 0;                                 // ... bit l = and NOT the fp32 slab: gates[l] still holds the forward's activated gates (gemm_on refuses to read it)
    int dg16_written = 0;                                 // 16-bit data path: bit l = decoder layer l's backward recurrence wrote dg_img[l] (plain bf16) in this backward
    float *d_act_l[2] = {nullptr, nullptr}, *d_img_l[2] = {nullptr, nullptr};      // conv-output gradients of trunk layers 1 and 2 when their weight gradients run off the chain (conv_dw_off)
    float *d_img = nullptr, *d_img_t = nullptr;           // conv-output gradients of the trunk [B, TP, CE] / Encoder_t [B, TP, dim_enc_2]
    float* gscale = nullptr;               // [16]
    float* act_scale = nullptr;            // [8] per conv block: scale of its output's fp16 x 2 split (act_scales, from the GroupNorm affine)
    void* zeros = nullptr;                 // 1 KB of zero bytes (image GEMM: reduction rows past K)
    float* part = nullptr;                 // split-K partial slabs of the image GEMM: a bump allocator over part_cap floats, reset per step
    long part_cap = 0, part_off = 0;
    long scratch_fallbacks = 0;            // launches that found the step's scratch exhausted and took the slower path (ss_scratch_fallbacks): 0 in a healthy run
    float* amax = nullptr;                 // [16] max |gradient| of the slabs the fp16 x 2 gradient GEMMs read: decoder layers 0..2, then the 7 convs
    long gp_bytes = 0;
    float *loss_part = nullptr;
    int* qidx = nullptr;
    InterpPlan plan[4];
    AdamState* adam = nullptr;
    std::map<std::string, std::pair<float*, long>> dbg;   // name -> (ptr, cols)
    // weight-gradient GEMMs of a BLSTM layer run on this side stream while the next layer's recurrence (latency-bound,
    // one launch per time step) proceeds on the caller's stream
    std::string stream_report;            // what pick_streams found (ss_stream_report)
    hipStream_t main_s = nullptr;         // the stream the step's dependency chain runs on (see g_own_streams)
    hipStream_t side = nullptr;
    hipStream_t side2 = nullptr;          // independent branches (per-step weight re-layouts, Encoder_t, second encoder BLSTM); second batch-half chain
    hipStream_t side3 = nullptr;          // third independent branch of the encoder backward (Encoder_t)
    hipEvent_t ev_io[2] = {};             // caller stream <-> engine main stream ordering (own_streams)
    hipEvent_t ev_join[4] = {};           // events recorded early: [0] lstm_2 branch's input gradient, [1] decoder chain done, [2] dec_in_grad done, [3] lstm_1 chain done
    hipEvent_t ev_dec[2] = {};            // split step without join: decoder chain done (caller stream) / its weight gradients done (side)
    int dec_pending = 0;                  // 0 none, 1 ev_dec[0] only, 2 both
    hipEvent_t ev[16] = {};
    int ev_next = 0;
    bool side_used = false;
    // ss_profile: hipEvent pairs around the launches of the dominant kernel (decoder input-projection GEMM, layers >= 1)
    // on the stream they are launched on, so a benchmark can report that kernel's duration inside its own timed region
    static constexpr int PROF_CAP = 8192;
    struct ProfRec {
        int klass;
        double flops;
        int stream;                       // 0 caller / main, 1 side, 2 second branch, 3 third branch, -1 other
    };
    std::vector<hipEvent_t> prof_ev;      // 2 per record, created on demand
    std::vector<ProfRec> prof_rec;
    int prof_n = 0;
    unsigned prof_mask = 0;               // bit k set: launches of class k are bracketed
    int prof_every = 1, prof_ctr = 0;     // ss_profile_sample: brackets only in every prof_every-th training step
    bool prof_live = true;                // this step is one of them

    long carve(int B, int T, bool assign);
    // 16-bit data path: the operand images are plain bf16 tensors (2 bytes per element) instead of format-v2 images (4)
    bool img16() const { return precision == SS_PRECISION_BF16 && g_bf16_img; }
    // image pointer `elems` ELEMENTS behind `base` (an image has its tensor's geometry; the bytes per element depend on the format)
    float* ioff(float* base, long elems) const { return img16() ? (float*)((char*)base + 2 * elems) : base + elems; }
    const float* ioff(const float* base, long elems) const { return img16() ? (const float*)((const char*)base + 2 * elems) : base + elems; }
};

namespace {

long align4(long x) { return (x + 3) & ~3L; }

// ------------------------------------------------------------------------------------------------ parameter table
struct TableBuilder {
    std::vector<ParamInfo>& v;
    long off = 0;
    long add(const std::string& name, long a, long b = -1, long c = -1) {
        ParamInfo p;
        p.name = name;
        p.offset = off;
        p.ndim = b < 0 ? 1 : (c < 0 ? 2 : 3);
        p.shape[0] = a;
        p.shape[1] = b < 0 ? 1 : b;
        p.shape[2] = c < 0 ? 1 : c;
        v.push_back(p);
        off = align4(off + p.numel());
        return p.offset;
    }
    void conv(const std::string& pre, int ci, int co, ConvBlk& cb) {
        cb.Ci = ci;
        cb.Co = co;
        cb.Cp = (int)align4(ci);
        cb.w = add(pre + ".0.conv.weight", co, ci, 5);
        cb.b = add(pre + ".0.conv.bias", co);
        cb.ga = add(pre + ".1.weight", co);
        cb.be = add(pre + ".1.bias", co);
    }
    void lstm(const std::string& pre, int in, int hid, int layers, LstmBlk& lb) {
        lb.In = in;
        lb.H = hid;
        lb.L = layers;
        lb.pd.clear();
        for (int l = 0; l < layers; ++l) {
            const int i = l == 0 ? in : 2 * hid;
            for (int d = 0; d < 2; ++d) {
                const std::string sfx = "_l" + std::to_string(l) + (d ? "_reverse" : "");
                LstmDir pd;
                pd.wih = add(pre + ".weight_ih" + sfx, 4 * hid, i);
                pd.whh = add(pre + ".weight_hh" + sfx, 4 * hid, hid);
                pd.bih = add(pre + ".bias_ih" + sfx, 4 * hid);
                pd.bhh = add(pre + ".bias_hh" + sfx, 4 * hid);
                lb.pd.push_back(pd);
            }
        }
    }
};

void build_table(ss_engine* e) {
    const ss_hparams& h = e->hp;
    TableBuilder tb{e->params};
    if (e->kind == SS_INTERP_ONLY) {
        e->arena = 0;
        return;
    }
    if (e->kind == SS_GENERATOR_3) {      // registration order of model.py:161-191, 59-71, 244-247, 288-290
        for (int i = 0; i < 3; ++i)
            tb.conv("encoder_1.convolutions_1." + std::to_string(i), i == 0 ? h.dim_freq : h.dim_enc, h.dim_enc, e->c1[i]);
        tb.lstm("encoder_1.lstm_1", h.dim_enc, h.dim_neck, 2, e->l1);
        for (int i = 0; i < 3; ++i)
            tb.conv("encoder_1.convolutions_2." + std::to_string(i), i == 0 ? h.dim_f0 : h.dim_enc_3, h.dim_enc_3, e->c2[i]);
        tb.lstm("encoder_1.lstm_2", h.dim_enc_3, h.dim_neck_3, 1, e->l2);
        tb.conv("encoder_2.convolutions.0", h.dim_freq, h.dim_enc_2, e->ct);
        tb.lstm("encoder_2.lstm", h.dim_enc_2, h.dim_neck_2, 1, e->lt);
        e->dec_in_dim = 2 * h.dim_neck + 2 * h.dim_neck_2 + 2 * h.dim_neck_3 + h.dim_spk_emb;
        tb.lstm("decoder.lstm", e->dec_in_dim, 512, 3, e->ld);
        e->head_out = h.dim_freq;
        e->head_w = tb.add("decoder.linear_projection.linear_layer.weight", h.dim_freq, 1024);
        e->head_b = tb.add("decoder.linear_projection.linear_layer.bias", h.dim_freq);
        e->CE = h.dim_enc + h.dim_enc_3;
    } else {                              // model.py:330-332, 107-119, 268-271
        tb.conv("encoder_2.convolutions.0", h.dim_freq, h.dim_enc_2, e->ct);
        tb.lstm("encoder_2.lstm", h.dim_enc_2, h.dim_neck_2, 1, e->lt);
        for (int i = 0; i < 3; ++i)
            tb.conv("encoder_3.convolutions." + std::to_string(i), i == 0 ? h.dim_f0 : h.dim_enc_3, h.dim_enc_3, e->c2[i]);
        tb.lstm("encoder_3.lstm", h.dim_enc_3, h.dim_neck_3, 1, e->l2);
        e->dec_in_dim = 2 * h.dim_neck_2 + 2 * h.dim_neck_3;
        tb.lstm("decoder.lstm", e->dec_in_dim, 256, 2, e->ld);
        e->head_out = h.dim_f0;
        e->head_w = tb.add("decoder.linear_projection.linear_layer.weight", h.dim_f0, 512);
        e->head_b = tb.add("decoder.linear_projection.linear_layer.bias", h.dim_f0);
        e->CE = h.dim_enc_3;
    }
    e->ld.amax0 = 0;                      // decoder layers: slots 0..2 of ss_engine::amax (the convs follow from 3)
    e->status_off = align4(tb.off);
    e->arena = e->status_off + 4;
    e->f0p = (int)((h.dim_f0 + 7) & ~7);      // a multiple of 8: the packed conv weights of convolutions_2[0] then have rows of whole image groups
    for (int i = 0; i < 3; ++i) {
        e->c1[i].need_dx = i > 0;
        e->c2[i].need_dx = i > 0;
    }
    e->ct.need_dx = false;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ workspace plan
long ss_engine::carve(int B, int T, bool assign) {
    const long TP = T + 2 * HALO;
    const long R = (long)B * TP;
    long off = 0;
    dbg.clear();
    auto take = [&](long bytes) -> char* {
        char* p = assign ? ws + off : nullptr;
        off += (bytes + 255) & ~255L;
        return p;
    };
    auto slab = [&](const char* name, long cols) -> float* {
        float* p = (float*)take(R * cols * 4);
        if (assign && name) dbg[name] = {p, cols};
        return p;
    };
    auto conv_ws = [&](ConvBlk& cb, const std::string& name) {
        if (cb.Co == 0) return;
        cb.wf = (float*)take((long)cb.Co * 5 * cb.Cp * 4);
        cb.wb = cb.need_dx ? (float*)take((long)cb.Ci * 5 * cb.Co * 4) : nullptr;
        cb.wf_img = (float*)take((long)cb.Co * 5 * cb.Cp * 4);
        cb.wb_img = cb.need_dx ? (float*)take((long)cb.Ci * 5 * cb.Co * 4) : nullptr;
        cb.cout = slab((name + ".conv").c_str(), cb.Co);
        cb.stats = (float*)take((long)B * (cb.Co / 16) * 2 * 4);
        cb.part = (float*)take((long)B * 3 * cb.Co * 4);          // deterministic mode: per-utterance affine / bias gradient sums
    };
    auto lstm_ws = [&](LstmBlk& lb, const std::string& name) {
        if (lb.L == 0) return;
        lb.gates.assign(lb.L, nullptr);
        lb.out.assign(lb.L, nullptr);
        lb.csave.assign(lb.L, nullptr);
        for (int l = 0; l < lb.L; ++l) {
            lb.gates[l] = slab((name + ".gates" + std::to_string(l)).c_str(), 8L * lb.H);
            lb.out[l] = slab((name + ".out" + std::to_string(l)).c_str(), 2L * lb.H);
            lb.csave[l] = slab((name + ".c" + std::to_string(l)).c_str(), 2L * lb.H);
        }
        lb.out_img.assign(lb.L, nullptr);
        if (lb.big())
            for (int l = 0; l < lb.L; ++l) lb.out_img[l] = slab(nullptr, 2L * lb.H);       // halo rows stay zero like those of out
        lb.bsum = (float*)take((long)lb.L * 2 * 4 * lb.H * 4);
        lb.wcat.assign(lb.L, nullptr);
        for (int l = 0; l < lb.L; ++l) lb.wcat[l] = (float*)take(8L * lb.H * lb.in_of(l) * 4);
        lb.wcat_img.assign(lb.L, nullptr);
        if (lb.big())
            for (int l = 0; l < lb.L; ++l) lb.wcat_img[l] = (float*)take(8L * lb.H * lb.in_of(l) * 4);
        if (lb.big()) {
            const long B16 = ((B + 15) / 16) * 16;
            lb.wfrag.assign(lb.L, nullptr);
            for (int l = 0; l < lb.L; ++l) lb.wfrag[l] = (float*)take(2L * lb.H * 4 * lb.H * 4);
            for (int c = 0; c < 2; ++c) {
                lb.hf[c] = (float*)take(2L * 2 * B16 * lb.H * 4);
                lb.gf[c] = (float*)take(2L * 2 * B16 * 4 * lb.H * 4);
                lb.dc[c] = (float*)take(2L * B * lb.H * 4);
            }
            lb.sync = (unsigned*)take(LSTM_SEQ_SYNC_WORDS * 4);
            lb.hf_bytes = lstm_seq_xbytes(B, lb.H, false);       // exchange buffers of the persistent kernels
            lb.gf_bytes = lstm_seq_xbytes(B, lb.H, true);
            lb.zf_bytes = lb.L * (4L * LSTM_SEQ_SYNC_WORDS + lb.hf_bytes);
            lb.zb_bytes = lb.L * (4L * LSTM_SEQ_SYNC_WORDS + lb.gf_bytes);
            lb.zf = (char*)take(lb.zf_bytes);
            lb.zb = (char*)take(lb.zb_bytes);
        }
        if (lb.L > 1) {
            lb.dmid[0] = slab((name + ".dmid0").c_str(), 2L * lb.H);
            lb.dmid[1] = slab((name + ".dmid1").c_str(), 2L * lb.H);
        }
    };
    adam = (AdamState*)take(sizeof(AdamState));        // first: survives geometry changes (offset 0)
    if (kind == SS_INTERP_ONLY) {                      // a bare InterpLnr module only needs one plan
        for (int i = 0; i < 4; ++i) {
            plan[i].S = hp.max_len_seq / hp.min_len_seg + 1;
            plan[i].ncand = 2 * hp.max_len_seg;
            plan[i].P = hp.max_len_pad;
            plan[i].T = T;
        }
        plan[3].i0 = (int*)take((long)B * hp.max_len_pad * 4);
        plan[3].lam = (float*)take((long)B * hp.max_len_pad * 4);
        plan[3].nrows = (int*)take((long)B * 4);
        plan[3].counts = (int*)take((long)B * 4);
        plan[3].start = (int*)take((long)B * (T + 1) * 4);
        return off;
    }
    in_mel = slab("in.mel", hp.dim_freq);
    in_f0 = slab("in.f0", f0p);
    org = slab("in.org", hp.dim_freq);
    emb = (float*)take((long)B * hp.dim_spk_emb * 4);
    for (int i = 0; i < 3; ++i) {
        conv_ws(c1[i], "enc1.c1_" + std::to_string(i));
        conv_ws(c2[i], (kind == SS_GENERATOR_3 ? "enc1.c2_" : "enc3.c_") + std::to_string(i));
        xf[i] = slab(("enc.xf" + std::to_string(i)).c_str(), CE);
        xf_img[i] = i < 2 ? slab(nullptr, CE) : nullptr;       // pre-split images of the resampled activations the next layer's convs read
    }
    act = slab("enc.act", CE);
    d_act = slab("enc.d_act", CE);
    d_img = slab(nullptr, CE);
    for (int i = 0; i < 2; ++i) {
        d_act_l[i] = slab(nullptr, CE);
        d_img_l[i] = slab(nullptr, CE);
    }
    d_img_t = slab(nullptr, hp.dim_enc_2);
    gscale = (float*)take(16 * 4);
    act_scale = (float*)take(8 * 4);
    zeros = take(1024);
    d_xf = slab("enc.d_xf", CE);
    conv_ws(ct, "enc2.c");
    {   // packed weight-gradient images of every conv, contiguous so one memset per backward zeroes them all
        long tot = 0;
        ConvBlk* all[7] = {&c1[0], &c1[1], &c1[2], &c2[0], &c2[1], &c2[2], &ct};
        for (ConvBlk* cb : all) tot += align4((long)cb->Co * 5 * cb->Cp);
        float* p = (float*)take(tot * 4);
        gp_all = p;
        gp_bytes = tot * 4;
        amax = (float*)take(16 * 4);
        int slot = 3;
        for (ConvBlk* cb : all) {
            cb->gp = cb->Co ? p : nullptr;
            cb->scale_i = slot - 3;
            cb->amax_i = slot++;
            p += align4((long)cb->Co * 5 * cb->Cp);
        }
    }
    act_t = slab("enc2.act", hp.dim_enc_2);
    d_act_t = slab("enc2.d_act", hp.dim_enc_2);
    lstm_ws(l1, "enc1.lstm1");
    lstm_ws(l2, kind == SS_GENERATOR_3 ? "enc1.lstm2" : "enc3.lstm");
    lstm_ws(lt, "enc2.lstm");
    lstm_ws(ld, "dec.lstm");
    for (int l = 0; l < ld.L && l < 3; ++l) dg_img[l] = ld.big() ? slab(nullptr, 8L * ld.H) : nullptr;
    if (l1.L) d_o1 = slab("enc1.d_o1", 2L * l1.H);
    d_o2 = slab("enc.d_o2", 2L * l2.H);
    d_ot = slab("enc2.d_ot", 2L * lt.H);
    dec_in = slab("dec.in", dec_in_dim);
    d_dec_in = slab("dec.d_in", dec_in_dim);
    {
        const int F = hp.freq_2;
        const bool same = hp.freq_3 == F && (kind != SS_GENERATOR_3 || hp.freq == F);
        ld.xf = (same && F > 1 && T % F == 0 && ld.big()) ? F : 0;       // structurally possible; ss_tune("compact0") decides per step
        ld.xcols = kind == SS_GENERATOR_3 ? dec_in_dim - hp.dim_spk_emb : 0;
        if (ld.xf) {
            const long R8 = (long)B * (T / F);
            ld.xc = (float*)take(R8 * dec_in_dim * 4);
            ld.xp0 = (float*)take(R8 * 8 * ld.H * 4);
            ld.dgs = (float*)take(R8 * 8 * ld.H * 4);
            ld.d_xc = (float*)take(R8 * dec_in_dim * 4);
        }
    }
    d_top = slab("dec.d_top", 2L * ld.H);
    out_slab = slab("out", head_out);
    d_out_slab = slab("d_out", head_out);
    loss_part = (float*)take(((long)B * T + 8L * B) * 4);
    qidx = (int*)take((long)B * TP * 4);
    wq_pool = (unsigned*)take(WQ_SLOTS * 16);
    colsum_ctr = (unsigned*)take(COLSUM_CTRS * 4);
    for (int i = 0; i < 4; ++i) {
        plan[i].S = hp.max_len_seq / hp.min_len_seg + 1;     // model.py:365
        plan[i].ncand = 2 * hp.max_len_seg;                  // model.py:389
        plan[i].P = hp.max_len_pad;
        plan[i].T = T;
        plan[i].i0 = (int*)take((long)B * hp.max_len_pad * 4);
        plan[i].lam = (float*)take((long)B * hp.max_len_pad * 4);
        plan[i].nrows = (int*)take((long)B * 4);
        plan[i].counts = (int*)take((long)B * 4);
        plan[i].start = (int*)take((long)B * (T + 1) * 4);
    }
    return off;
}

namespace {

hipStream_t S(void* s) { return (hipStream_t)s; }


int geometry(ss_engine* e, int B, int T, hipStream_t s) {
    if (!e->ws) return fail("engine is not bound (call ss_bind first)");
    if (B < 1 || B > e->maxB || T < 1 || T > e->maxT) return fail("batch / frames outside the limits given to ss_create");
    if (e->kind != SS_INTERP_ONLY && (T % e->hp.freq || T % e->hp.freq_2 || T % e->hp.freq_3))
        return fail("T must be a multiple of the code down-sampling factors (model.py:87,223-227)");
    if (B == e->curB && T == e->curT) return 0;
    long need;
    {
        ss_engine tmp = *e;
        need = tmp.carve(B, T, false);
    }
    if (need > e->ws_bytes) return fail("workspace too small");
    // new geometry: halo rows move, so everything the new plan uses, except the Adam state (first 256 bytes), is re-zeroed
    // (0.1 - 0.2 ms per switch at batch 64: the price of a length-bucket change, SS_STEP_BUCKET)
    HIPCHK(hipMemsetAsync(e->ws + 256, 0, need - 256, s));
    e->carve(B, T, true);
    e->curB = B;
    e->curT = T;
    e->have_fwd = false;
    return 0;
}

int g_dw_wgs = 384;        // workgroups a split-K weight-gradient launch aims at (round 2: 384 = 1.5 per CU measured best in the step, 5.38 vs 5.48 ms at 512; 256: 5.5, 768+: 5.7)
int pick_ksplit(int M, int N, long K) {
    const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
    long ks = g_dw_wgs / (tiles > 0 ? tiles : 1);
    if (ks > K / 256) ks = K / 256;
    if (ks > 32) ks = 32;
    if (ks < 1) ks = 1;
    return (int)ks;
}

// Image GEMM (gemm_img.hip) for a contraction whose two operands exist as images.  Returns 1 when it was launched, 0 when the
// contraction has to take round 2's kernels (no images, shape outside what the image kernel supports), < 0 on error.
int try_img_gemm(ss_engine* e, const GemmDesc& d, hipStream_t st) {
    if (!g_img || !d.a_pre || !d.b_pre) return 0;
    const bool b16 = e->img16();                     // the images are plain bf16 tensors: single-piece form, every class that has both
    if (b16) {
        if (g_cur_klass >= 0 && !((g_bf16_img_mask >> g_cur_klass) & 1)) return 0;
    } else {
        if (!(d.flags & GEMM_F16X2) || (d.flags & GEMM_BF16)) return 0;
        if (g_cur_klass >= 0 && !((g_img_mask >> g_cur_klass) & 1) && !d.queue) return 0;
        // conv trunk at B x T <= 2048 rows: the image kernel's smallest tile (128 x 128) gives a 512-channel layer 64 workgroups; round 2's kernel
        // on 64 x 64 tiles fills the chip (16 x 128: 3.22 -> 3.17 ms, 12 x 128: 3.08 -> 3.04; equal at 8 x 128 and from 32 x 128 on)
        if ((g_cur_klass == SS_PROF_CONV_FWD || g_cur_klass == SS_PROF_CONV_DX) && g_conv_small_old && (long)e->curB * e->curT <= 2048) return 0;
    }
    ImgGemmDesc g{};
    g.bf16 = b16 ? 1 : 0;
    g.A = {d.a_pre, d.A.ld, d.A.bstride, d.A.seglen, d.A.segstride};
    g.B = {d.b_pre, d.B.ld, d.B.bstride, d.B.seglen, d.B.segstride};
    g.C = d.C;
    g.ldc = d.ldc;
    g.cstride = d.cstride;
    g.bias = d.bias;
    g.M = d.M;
    g.N = d.N;
    g.K = d.K;
    g.batch = d.batch;
    g.flags = d.flags & (GEMM_TA | GEMM_TB | GEMM_ACCUM);
    g.row_period = d.row_period;
    g.row_off = d.row_off;
    g.row_lo = d.row_lo;
    g.row_hi = d.row_hi;
    g.scale_a = b16 ? nullptr : d.a_pre_scale;
    g.scale_b = b16 ? nullptr : d.b_pre_scale;
    g.zeros = e->zeros;
    g.cfg = -1;
    g.ksplit = 1;
    g.diag = g_gemm_diag;
    // Per-utterance batches of T rows over haloed slabs (and their flatten_rows form) become ONE matrix over the B * T real rows: the
    // kernel maps logical rows to slab rows, so the 256-row tiles neither end at every utterance nor compute halo rows
    const int T = e->curT;
    const long TP = T + 2 * HALO;
    if (!(g.flags & GEMM_TA) && T >= 32 && g.B.bstride == 0) {
        if (g.batch > 1 && g.M == T && !g.row_period && g.A.bstride == TP * g.A.ld && g.cstride == TP * g.ldc) {
            g.M = g.batch * T;
            g.batch = 1;
            g.A.bstride = g.cstride = 0;
            g.rm_T = T;
            g.rm_TP = (int)TP;
        } else if (g.batch == 1 && g.row_period == TP && g.row_off == HALO && g.row_lo == HALO && g.row_hi == HALO + T && (g.M + 2 * HALO) % TP == 0) {
            g.M = (int)((g.M + 2 * HALO) / TP) * T;
            g.row_period = 0;
            g.rm_T = T;
            g.rm_TP = (int)TP;
        }
    }
    // tile and split: measured on the step's shapes (tools/img_bench.py) -- without split-K the largest tile that still gives every CU a
    // workgroup; reductions that are much longer than the result is large (weight gradients, d.ksplit > 1 on entry: C is zeroed or
    // live) are cut along K into partial slabs, on 256 x 128 tiles (128 x 128 for narrow results)
    auto wgs = [&](int bm, int bn) { return (long)cdiv(g.M, bm) * cdiv(g.N, bn) * g.batch; };
    if (d.ksplit > 1 && !g.row_period) {
        g.cfg = (g.M >= 256 && g.N >= 128) ? 2 : 1;
        // 16-bit mode: 128 x 128 tiles (64 KB of LDS, two workgroups per CU).  The 256 x 128 form is 147 KB: one workgroup per CU, so the trunk's
        // weight-gradient GEMMs on the dependent chain queue for CUs behind the decoder's on the side streams (real_timeline: 147 us for a 32-us
        // launch).  32 x 128: 2.96 -> 2.91 ms, 64 x 128: 3.75 -> 3.69; the work-queue form keeps one workgroup per CU by design
        if (b16 && !d.queue) g.cfg = 1;
        if (g_img_dw_cfg >= 0) g.cfg = g_img_dw_cfg;
        const long t = (g.cfg == 2 || g.cfg == 3) ? wgs(256, 128) : (g.cfg == 0 ? wgs(256, 256) : wgs(128, 128));
        long ks = ((d.queue ? 2 * g_img_dw_wgs : g_img_dw_wgs) + t / 2) / (t > 0 ? t : 1);       // work-queue form: twice the tiles (finer hand-over when the recurrence beside it ends)
        if (ks > g.K / 512) ks = g.K / 512;
        if (ks > 16) ks = 16;
        if (ks < 1) ks = 1;
        const long need = ks > 1 ? ks * (long)g.M * g.N * g.batch : 0;
        if (need && g.N % 4 == 0 && e->part && e->part_off + need <= e->part_cap) {
            g.ksplit = (int)ks;
            g.part = e->part + e->part_off;
            e->part_off += (need + 63) & ~63L;
        } else if (need) ++e->scratch_fallbacks;      // the contraction runs unsplit
    } else {
        g.cfg = wgs(256, 256) >= 256 ? 0 : (wgs(256, 128) >= 224 ? 2 : 1);
    }
    if (d.queue && e->wq_pool && e->wq_next < ss_engine::WQ_SLOTS) {
        // work-queue form on every XCD that will give it a CU; one workgroup per CU (128 / 144 KB of LDS), so that none fits beside a
        // recurrence workgroup
        if (g.cfg == 1) g.cfg = 2;
        g.wq = e->wq_pool + 4 * e->wq_next++;
        g.xcc_allow = 0xFFu;
    }
    if (!gemm_img_supported(g)) return 0;
    HIPCHK(launch_gemm_img(g, st));
    return 1;
}

// scratch of one column-sum launch (kernels.h colsum_acc): float64 partials from the step's bump allocator, counters from the ring
// (self-resetting; a launch's counters are not handed out again before COLSUM_CTRS / 64 later launches); false: none left, the launch
// runs with one workgroup per column block
bool colsum_scratch(ss_engine* e, int cols, double** part, unsigned** ctr) {
    *part = nullptr;
    *ctr = nullptr;
    const long need = 2 * colsum_scratch_doubles(cols);          // in floats
    const int nb = cdiv(cols, 64);
    if (!e->part || !e->colsum_ctr || nb > ss_engine::COLSUM_CTRS || e->part_off + need > e->part_cap) {
        ++e->scratch_fallbacks;
        return false;
    }
    *part = (double*)(e->part + e->part_off);                    // part_off is kept at multiples of 64 floats
    e->part_off += (need + 63) & ~63L;
    if (e->colsum_next + nb > ss_engine::COLSUM_CTRS) e->colsum_next = 0;
    *ctr = e->colsum_ctr + e->colsum_next;
    e->colsum_next += nb;
    return true;
}

int prof_begin(ss_engine* e, int klass, hipStream_t st, double flops);
void prof_end(ss_engine* e, int i, hipStream_t st);

// launch the pending encoder-BLSTM weight-gradient tasks (one kernel for all of them) on `st`: everything they read must be complete in
// st's order.  Scratch: partial tiles from the step's bump allocator, arrival counters from the column sums' ring.
int wgrad_flush(ss_engine* e, hipStream_t st) {
    if (e->wg.n == 0) return 0;
    WgradTable& w = e->wg;
    w.row_groups = 16;
    const long need = (long)w.tiles_total * w.row_groups * 4096;
    if (!e->part || !e->colsum_ctr || w.tiles_total > ss_engine::COLSUM_CTRS || e->part_off + need > e->part_cap) {
        w.n = 0;
        w.tiles_total = 0;
        return fail("wgrad_flush: no scratch left for the fused encoder-BLSTM weight gradients (ss_tune(\"wgrad_fused\", 0) selects the GEMM path)");
    }
    w.part = e->part + e->part_off;
    e->part_off += (need + 63) & ~63L;
    if (e->colsum_next + w.tiles_total > ss_engine::COLSUM_CTRS) e->colsum_next = 0;
    w.ctr = e->colsum_ctr + e->colsum_next;
    e->colsum_next += w.tiles_total;
    const int pa_ = prof_begin(e, SS_PROF_WGRAD, st, 0.0);
    const hipError_t rc = lstm_small_wgrad(w, st);
    prof_end(e, pa_, st);
    w.n = 0;
    w.tiles_total = 0;
    HIPCHK(rc);
    return 0;
}

// every contraction of the engine honours its precision mode (ss_set_precision)
int gemm_on(ss_engine* e, GemmDesc& d, hipStream_t st) {
    if (e->precision == SS_PRECISION_BF16) d.flags |= GEMM_BF16;
    const int r = try_img_gemm(e, d, st);
    if (r < 0) return r;
    if (r == 0) {
        if (e->img16()) d.a_pre = d.b_pre = nullptr;        // plain bf16 tensors: not the format-v2 images round 2's kernel can take
        if (e->dg32_skipped) {                              // a decoder layer's fp32 gradient slab was not written: nothing may read it as an operand
            const long R8 = (long)e->curB * (e->curT + 2 * HALO) * 8L * e->ld.H;
            for (int l = 0; l < e->ld.L && l < 3; ++l)
                if (((e->dg32_skipped >> l) & 1) && ((d.A.p >= e->ld.gates[l] && d.A.p < e->ld.gates[l] + R8) || (d.B.p >= e->ld.gates[l] && d.B.p < e->ld.gates[l] + R8)))
                    return fail("internal: a contraction fell back to the fp32 gradient slab of a decoder layer that was only written as bf16 (ss_tune(\"seq_skip32\", 0))");
        }
        // split-K weight gradients: partial slabs + ordered reduce instead of fp32 atomics (ss_tune("part_splitk")), scratch from the step's bump allocator
        // (not for the encoder BLSTMs' tiny matrices: a one-block reduce over 32 slices is 17 us of latency, their atomics are nothing)
        if (g_part_splitk && d.ksplit > 1 && (d.flags & GEMM_TA) && (d.flags & GEMM_TB) && (d.flags & GEMM_ACCUM) && !d.row_period && !d.bias && d.N % 4 == 0 &&
            ((long)d.M * d.N >= 65536 || g_deterministic) && e->part) {        // deterministic mode: every split reduction through ordered slabs (small ones too)
            int ks = d.ksplit;
            const long need = (long)ks * d.M * d.N * (d.batch < 1 ? 1 : d.batch);
            if (e->part_off + need <= e->part_cap) {
                d.part = e->part + e->part_off;
                e->part_off += (need + 63) & ~63L;
            } else ++e->scratch_fallbacks;            // atomics (or, deterministic mode, no split) instead of ordered slabs
        }
        HIPCHK(launch_gemm(d, st));
    }
    return 0;
}
#define GEMM(d) GEMM_ON(d, s)
#define GEMM_ON(d, st) CHK(gemm_on(e, d, st))
// forward contraction: both operands are O(1) by construction
#define GEMM_FWD_ON(d, st)                              \
    do {                                                \
        if (g_fwd_f16x2) (d).flags |= GEMM_F16X2;       \
        GEMM_ON(d, st);                                 \
    } while (0)

// Data parallel: the gradient range [off, off + count) is final once everything enqueued on `producer` so far has run -- hand it to a
// collective on the communication stream right away.  No-op outside a data-parallel step.
int dp_bucket(ss_engine* e, long off, long count, hipStream_t producer);

// make `to` wait for everything enqueued on `from` so far
int fork_join(ss_engine* e, hipStream_t from, hipStream_t to) {
    hipEvent_t ev = e->ev[e->ev_next];
    e->ev_next = (e->ev_next + 1) & 15;
    HIPCHK(hipEventRecord(ev, from));
    HIPCHK(hipStreamWaitEvent(to, ev, 0));
    return 0;
}

// Host-side view of the engine status word: no synchronisation, so it reports what earlier steps left behind.
int sticky_check(ss_engine* e) {
    if (!e->sticky) return 0;
    const unsigned v = *(volatile unsigned*)e->sticky;
    if (!v) return 0;
    if (v & SS_STICKY_RANGE)
        return fail("a parameter is not finite or left the range (|p| < 2048) the fixed-scale fp16 x 2 products of the WEIGHTS are valid for: the step "
                    "was not applied.  Use ss_tune(\"fwd_f16x2\", 0) and ss_tune(\"bwd_f16x2\", 0) (bf16 x 3 products, no range limit), then ss_clear_abort()");
    return fail(v & SS_STICKY_ABORT ? "a persistent LSTM kernel gave up waiting for its group (bounded spin expired) in an earlier step: that step's "
                                      "results were discarded and the parameters left untouched; ss_clear_abort() to continue"
                                    : "another data-parallel rank reported an aborted step: the update was skipped on every rank; ss_clear_abort() to continue");
}

// What a step / forward / Adam entry point does about the status word.  One engine on its own refuses to enqueue (its caller learns
// of the failure at once).  In LOCKSTEP mode (data parallel: ss_comm_init with world > 1, or ss_set_lockstep) it enqueues regardless:
// the word is set asynchronously, so ranks would see it at DIFFERENT host iterations, one would return while the others have already
// enqueued collectives that never get a partner.  The weights are safe either way (the Adam kernel skips on the device); the status is
// reported by ss_check(), which every rank calls at the same iteration, and cleared there by every rank (ss_clear_abort).
int entry_check(ss_engine* e) { return e->lockstep ? 0 : sticky_check(e); }

// Scope of one C-ABI call: work goes to the engine's main stream, which first waits for everything the caller's stream holds;
// on exit the caller's stream waits for the call's work, so the stream-ordered contract of the ABI is unchanged.
struct Own {
    ss_engine* e;
    hipStream_t caller, s;
    bool on = false;
    Own(ss_engine* e_, void* stream) : e(e_), caller((hipStream_t)stream), s((hipStream_t)stream) {
        if (e && e->main_s && g_own_streams && caller != e->main_s &&
            hipEventRecord(e->ev_io[0], caller) == hipSuccess && hipStreamWaitEvent(e->main_s, e->ev_io[0], 0) == hipSuccess) {
            s = e->main_s;
            on = true;
        }
    }
    ~Own() {
        if (on && hipEventRecord(e->ev_io[1], s) == hipSuccess) (void)hipStreamWaitEvent(caller, e->ev_io[1], 0);
    }
    Own(const Own&) = delete;
    Own& operator=(const Own&) = delete;
};


// ---- hardware-queue probe ---------------------------------------------------------------------------------------------------------
// HIP multiplexes a process's streams onto a few in-order hardware queues (4 by default) and does not say which stream got which.
// Two streams on one queue execute strictly one after the other, so the step's branches lose the concurrency they exist for:
// measured 6.34 ms with the four engine streams on four queues against 7.22 ms with the side stream on the main stream's queue
// (profiles/r02/stream_order_effect.txt; which case one gets depends on how many streams the process -- PyTorch, RCCL -- created
// before).  The serialisation is also what makes the assignment MEASURABLE: a kernel that spins for ~400 us on stream A delays a
// trivial kernel on stream B only if A and B share a queue.
__global__ void queue_probe_spin_kernel(long long ticks) {      // wall_clock64: constant 100 MHz
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void queue_probe_nop_kernel() {}

// true if a and b are served by the same hardware queue
bool same_queue(hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb0, hipEvent_t eb1) {
    constexpr long long SPIN_TICKS = 40000;                       // 400 us
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    hipLaunchKernelGGL(queue_probe_spin_kernel, dim3(1), dim3(1), 0, a, SPIN_TICKS);
    (void)hipEventRecord(ea, a);
    (void)hipEventRecord(eb0, b);
    hipLaunchKernelGGL(queue_probe_nop_kernel, dim3(1), dim3(1), 0, b);
    (void)hipEventRecord(eb1, b);
    // poll (a blocking wait may wake up later than the spin lasts): b's kernel finished -- had a's spin finished by then?  On separate
    // queues b is done after a few microseconds and the spin is not.
    bool spin_done = false;
    for (long polls = 0;; ++polls) {
        const bool a_done = hipEventQuery(ea) == hipSuccess;
        if (hipEventQuery(eb1) == hipSuccess) {
            spin_done = a_done;
            break;
        }
        if (a_done || polls > 20000000L) {              // the spin ended first: b was held up behind it (or the device does not answer: count it as shared)
            spin_done = true;
            break;
        }
    }
    (void)hipStreamSynchronize(a);
    return spin_done;
}
// the same measurement, repeated when it says "shared": a co-tenant of the GPU that delays stream b for the length of the spin makes separate
// queues look shared once; it does not do so three times in a row
bool same_queue_voted(hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb0, hipEvent_t eb1) {
    for (int i = 0; i < 3; ++i)
        if (!same_queue(a, b, ea, eb0, eb1)) return false;
    return true;
}

// Choose the engine's three branch streams from a pool of fresh streams so that they and `main` sit on four different hardware queues
// (as far as the device's queue count allows); the rest of the pool is destroyed.
int pick_streams(ss_engine* e, hipStream_t main) {
    constexpr int POOL = 10;
    struct Guard {               // whatever is still in here when the function returns -- early on an error, or at its end -- is destroyed
        hipStream_t pool[POOL] = {};
        hipEvent_t ev[3] = {};
        ~Guard() {
            for (auto& st : pool)
                if (st) (void)hipStreamDestroy(st);
            for (auto& x : ev)
                if (x) (void)hipEventDestroy(x);
        }
    } gd;
    hipStream_t(&pool)[POOL] = gd.pool;
    hipEvent_t(&ev)[3] = gd.ev;
    for (auto& x : ev) HIPCHK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (auto& st : pool) {
        HIPCHK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, g_branch_low ? least : 0));
        hipLaunchKernelGGL(queue_probe_nop_kernel, dim3(1), dim3(1), 0, st);      // first use of a stream sets its queue up: not inside a measurement
    }
    hipLaunchKernelGGL(queue_probe_nop_kernel, dim3(1), dim3(1), 0, main);
    HIPCHK(hipDeviceSynchronize());
    std::vector<hipStream_t> chosen;
    char buf[64];
    e->stream_report.clear();
    for (int pass = 0; pass < 2 && chosen.size() < 3; ++pass)       // pass 1: accept streams that only differ from the main stream's queue
        for (int i = 0; i < POOL && chosen.size() < 3; ++i) {
            if (!pool[i]) continue;
            bool clash = same_queue_voted(main, pool[i], ev[0], ev[1], ev[2]);
            for (size_t k = 0; k < chosen.size() && !clash && pass == 0; ++k) clash = same_queue_voted(chosen[k], pool[i], ev[0], ev[1], ev[2]);
            if (!clash) {
                std::snprintf(buf, sizeof buf, "%scandidate %d%s", chosen.empty() ? "" : ", ", i, pass ? " (shares a queue with another branch)" : "");
                e->stream_report += buf;
                chosen.push_back(pool[i]);
                pool[i] = nullptr;
            }
        }
    for (int i = 0; i < POOL && chosen.size() < 3; ++i)             // a device with fewer queues than streams: take what is there
        if (pool[i]) {
            chosen.push_back(pool[i]);
            pool[i] = nullptr;
            e->stream_report += " +fallback";
        }
    e->side = chosen[0];
    e->side2 = chosen[1];
    e->side3 = chosen[2];
    e->stream_report = "branch streams on hardware queues of their own: " + e->stream_report;
    return 0;
}

// a training step begins: is it one of the bracketed ones (ss_profile_sample)?
void prof_tick(ss_engine* e) {
    e->prof_live = e->prof_every <= 1 || (e->prof_ctr % e->prof_every) == 0;
    ++e->prof_ctr;
}

// ss_profile: bracket one launch with hipEvents on the stream it is launched on
int prof_begin(ss_engine* e, int klass, hipStream_t st, double flops) {
    if (!((e->prof_mask >> klass) & 1u) || !e->prof_live || e->prof_n >= ss_engine::PROF_CAP) return -1;
    const int i = e->prof_n;
    while ((int)e->prof_ev.size() < 2 * (i + 1)) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return -1;
        e->prof_ev.push_back(ev);
    }
    if ((int)e->prof_rec.size() <= i) e->prof_rec.resize(i + 1);
    e->prof_rec[i] = {klass, flops, st == e->side ? 1 : (st == e->side2 ? 2 : (st == e->side3 ? 3 : 0))};
    if (hipEventRecord(e->prof_ev[2 * i], st) != hipSuccess) return -1;
    e->prof_n = i + 1;
    return i;
}
void prof_end(ss_engine* e, int i, hipStream_t st) {
    if (i >= 0) (void)hipEventRecord(e->prof_ev[2 * i + 1], st);
}
// ALGORITHMIC work of a contraction (ss_profile): halo rows are not work -- a reduction over all B * (T + 4) slab rows counts B * T of
// them, a flattened row-masked matrix its B * T stored rows (round-2 review: the weight gradients' count was 3 % high)
double gemm_flops_of(const ss_engine* e, const GemmDesc& d) {
    double M = (double)d.M * (d.batch > 0 ? d.batch : 1), K = (double)d.K;
    const long TP = e->curT + 2 * HALO, R = (long)e->curB * TP;
    if ((d.flags & GEMM_TA) && (d.flags & GEMM_TB) && d.K >= R - 2 * HALO && d.K <= R) K = (double)e->curB * e->curT;
    if (d.row_period == TP && d.batch <= 1) M = (double)((d.M + 2 * HALO) / TP) * e->curT;
    return 2.0 * M * d.N * K;
}
#define gemm_flops(d) gemm_flops_of(e, d)
// GEMM launch bracketed as profile class `k`
#define PGEMM_ON(k, d, st)                                   \
    do {                                                     \
        const int _pi = prof_begin(e, k, st, gemm_flops(d)); \
        g_cur_klass = k;                                     \
        GEMM_ON(d, st);                                      \
        g_cur_klass = -1;                                    \
        prof_end(e, _pi, st);                                \
    } while (0)
#define PGEMM_FWD_ON(k, d, st)                               \
    do {                                                     \
        const int _pi = prof_begin(e, k, st, gemm_flops(d)); \
        g_cur_klass = k;                                     \
        GEMM_FWD_ON(d, st);                                  \
        g_cur_klass = -1;                                    \
        prof_end(e, _pi, st);                                \
    } while (0)

// A per-utterance batched contraction over haloed slabs (M = T rows per batch entry, A / C pointing at slab row HALO) rewritten as
// ONE matrix over all slab rows between the first and the last halo, halo rows masked in the epilogue.  Batched, every utterance
// starts a new row tile: T = 144 needs two 128-row tiles per utterance (78 % of the rows computed are padding), T = 192 wastes 25 %.
// Worth it whenever T is not a multiple of the 128-row tile.  The A operand's rows are addressed exactly as before (row m of the
// flat matrix is slab row m + HALO, taps reach into the neighbouring rows; rows computed for halo positions are never stored).
void flatten_rows(GemmDesc& d, int B, int T) {
    if (T % 128 == 0 || d.batch != B || B < 2 || g_flat_rows == 0) return;
    const int TP = T + 2 * HALO;
    d.M = B * TP - 2 * HALO;
    d.batch = 1;
    d.A.bstride = 0;
    d.cstride = 0;
    d.row_period = TP;
    d.row_off = HALO;
    d.row_lo = HALO;
    d.row_hi = HALO + T;
}

// ---- convolution block -------------------------------------------------------------------------------------
int conv_pack_all(ss_engine* e, ConvBlk& cb, hipStream_t s) {
    const bool img = cb.img_ok();
    HIPCHK(conv_pack(e->P + cb.w, cb.Co, cb.Ci, cb.Cp, cb.wf, cb.wb, img ? cb.wf_img : nullptr, img ? cb.wb_img : nullptr, s, e->img16()));
    return 0;
}

int zero_conv_grads(ss_engine* e, hipStream_t s) {
    HIPCHK(hipMemsetAsync(e->gp_all, 0, e->gp_bytes, s));
    return 0;
}

// where the image of a conv-output gradient slab view goes (the trunk's views share d_img, Encoder_t's has d_img_t); null: no image
float* grad_img_of(ss_engine* e, const float* p, long R) {
    auto in = [&](const float* b, long cols) { return b && p >= b && p < b + R * cols; };
    if (in(e->d_act, e->CE)) return e->ioff(e->d_img, p - e->d_act);
    for (int i = 0; i < 2; ++i)
        if (in(e->d_act_l[i], e->CE)) return e->ioff(e->d_img_l[i], p - e->d_act_l[i]);
    if (in(e->d_xf, e->CE)) return e->ioff(e->d_img, p - e->d_xf);
    if (in(e->d_act_t, e->hp.dim_enc_2)) return e->ioff(e->d_img_t, p - e->d_act_t);
    return nullptr;
}

// y = relu(GN(conv5(x)))   x: slab view (ld), y: slab view
// gather (nullable): the resampling plan of the training forward -- GroupNorm + ReLU + gather in one kernel straight into gy / gy_img (the
// resampled slab and its image at the first real row and the block's first column); y is then not written
int conv_block_fwd(ss_engine* e, ConvBlk& cb, Slab x, Slab y, hipStream_t s, const InterpPlan* gather = nullptr, float* gy = nullptr, long gy_ld = 0,
                   float* gy_img = nullptr, hipEvent_t gather_ready = nullptr) {
    const int B = e->curB, T = e->curT;
    const long TP = T + 2 * HALO;
    GemmDesc d{};
    d.A = {x.p, x.ld, TP * x.ld, cb.Cp, x.ld};
    d.a_pre = x.img;
    d.a_pre_scale = x.scale;
    d.B = {cb.wf, 5L * cb.Cp, 0, 0, 0};
    d.b_pre = ((g_presplit & 1) && cb.img_ok()) ? cb.wf_img : nullptr;
    d.C = cb.cout + HALO * cb.Co;
    d.ldc = cb.Co;
    d.cstride = TP * cb.Co;
    d.bias = e->P + cb.b;
    d.M = T;
    d.N = cb.Co;
    d.K = 5 * cb.Cp;
    d.batch = B;
    d.ksplit = 1;
    d.want = g_conv_want;
    flatten_rows(d, B, T);
    PGEMM_FWD_ON(SS_PROF_CONV_FWD, d, s);
    if (gather) {
        if (gather_ready) HIPCHK(hipStreamWaitEvent(s, gather_ready, 0));
        { const int pa_ = prof_begin(e, SS_PROF_GN, s, 0.0);
        HIPCHK(gn_relu_gather(cb.cout, cb.Co, TP * cb.Co, gy, gy_ld, TP * gy_ld, gy_img, e->act_scale + cb.scale_i, e->P + cb.ga, e->P + cb.be, cb.stats,
                              *gather, B, T, cb.Co, s, e->img16()));
        prof_end(e, pa_, s); }
        return 0;
    }
    { const int pa_ = prof_begin(e, SS_PROF_GN, s, 0.0);
    HIPCHK(gn_relu_fwd(cb.cout, cb.Co, TP * cb.Co, y.p, y.ld, TP * y.ld, e->P + cb.ga, e->P + cb.be, cb.stats, B, T, cb.Co, s));
    prof_end(e, pa_, s); }
    return 0;
}

// dy: gradient of the block output (slab view, overwritten in place with the conv-output gradient);
// x: the block's forward input; dx: where to put the input gradient (p == nullptr: not needed)
// scatter / src (nullable): the block's output was resampled in the forward (training): src is the gradient of the RESAMPLED output (at its
// first real row and this block's first column, row stride src_ld); the gather's adjoint is taken inside the GroupNorm backward, which
// writes dy
// dws (nullable): the weight-gradient GEMM goes to that stream behind an event, the chain on `s` does not wait for it -- the caller keeps dy
// untouched until dws is joined
int conv_block_bwd(ss_engine* e, ConvBlk& cb, Slab dy, Slab x, Slab dx, hipStream_t s, const InterpPlan* scatter = nullptr, const float* src = nullptr,
                   long src_ld = 0, hipStream_t dws = nullptr) {
    const int B = e->curB, T = e->curT;
    const long TP = T + 2 * HALO, R = (long)B * TP;
    float* am = (g_bwd_f16x2 && cb.amax_i >= 0) ? e->amax + cb.amax_i : nullptr;
    const bool i16 = e->img16();
    // 16-bit data path: the GroupNorm backward writes the gradient's bf16 image itself (the image's halo rows are zero since the geometry
    // was planned and nobody writes them); fp32 mode: a split_image pass with the scale gn_relu_bwd has just measured
    float* im16 = (i16 && g_img && cb.Co % 8 == 0 && dy.ld % 8 == 0) ? grad_img_of(e, dy.p, R) : nullptr;
    { const int pa_ = prof_begin(e, SS_PROF_GN, s, 0.0);
    HIPCHK(gn_relu_bwd(cb.cout, cb.Co, TP * cb.Co, dy.p, dy.ld, TP * dy.ld, e->P + cb.ga, e->P + cb.be, cb.stats,
                       e->G + cb.ga, e->G + cb.be, e->G + cb.b, am, cb.part, B, T, cb.Co, s, scatter, src, src_ld, TP * src_ld, im16));
    prof_end(e, pa_, s); }
    // the conv-output gradient as an image for the image GEMM (scale: the power of two for the maximum gn_relu_bwd has just measured)
    const float* dimg = im16;
    const float* dsc = nullptr;
    if (!i16 && g_img && (g_img_mask & ((1 << SS_PROF_CONV_DW) | (1 << SS_PROF_CONV_DX))) && am && e->precision == SS_PRECISION_F32 && cb.Co % 8 == 0 && dy.ld % 8 == 0) {
        float* im = grad_img_of(e, dy.p, R);
        if (im) {
            HIPCHK(split_image(dy.p, dy.ld, R, cb.Co, am, 0.f, im, dy.ld, e->gscale + cb.amax_i, s));
            dimg = im;
            dsc = e->gscale + cb.amax_i;
        }
    }
    // weight gradient: one reduction over every slab row (halo rows of dy are zero); cb.gp was zeroed by zero_conv_grads
    GemmDesc d{};
    d.A = {dy.p + 2 * dy.ld, dy.ld, 0, 0, 0};
    if (dimg) {
        d.a_pre = e->ioff(dimg, 2 * dy.ld);
        d.a_pre_scale = dsc;
    }
    d.B = {x.p, x.ld, 0, cb.Cp, x.ld};
    d.b_pre = x.img;
    d.b_pre_scale = x.scale;
    d.C = cb.gp;
    d.ldc = 5L * cb.Cp;
    d.M = cb.Co;
    d.N = 5 * cb.Cp;
    d.K = (int)(R - 4);
    d.batch = 1;
    d.flags = GEMM_TA | GEMM_TB | GEMM_ACCUM | (am ? GEMM_F16X2 : 0);
    d.amax_a = am;                                  // gradient operand: measured scale; the block input is O(1)
    d.ksplit = pick_ksplit(d.M, d.N, d.K);
#ifdef SS_DIAG
    if (g_exp & 8) goto conv_dw_done;            // what-if timing run (WRONG gradients): without the conv weight-gradient GEMMs
#endif
    if (dws && dws != s) CHK(fork_join(e, s, dws));
    else dws = s;
    PGEMM_ON(SS_PROF_CONV_DW, d, dws);
#ifdef SS_DIAG
conv_dw_done:
#endif
    // packed [Co][5][Cp] -> the parameter's [Co][Ci][5]: nobody reads it before the optimiser (or, data parallel, the layer's bucket), so the
    // one-GPU step collects the blocks and unpacks them all in one launch at the end of the backward (backward_encoder) instead of seven
    // small launches on the trunk's dependent chain
    if (e->unpack_later && e->unpack.n < CONV_UNPACK_MAX) e->unpack.t[e->unpack.n++] = {cb.gp, e->G + cb.w, cb.Co, cb.Ci, cb.Cp};
    else HIPCHK(conv_unpack_grad(cb.gp, cb.Co, cb.Ci, cb.Cp, e->G + cb.w, dws));
    if (dx.p) {
        GemmDesc g{};
        g.A = {dy.p, dy.ld, TP * dy.ld, cb.Co, dy.ld};
        if (dimg) {
            g.a_pre = dimg;
            g.a_pre_scale = dsc;
        }
        g.B = {cb.wb, 5L * cb.Co, 0, 0, 0};
        g.b_pre = ((g_presplit & 1) && cb.img_ok()) ? cb.wb_img : nullptr;
        g.C = dx.p + HALO * dx.ld;
        g.ldc = dx.ld;
        g.cstride = TP * dx.ld;
        g.M = T;
        g.N = cb.Ci;
        g.K = 5 * cb.Co;
        g.batch = B;
        g.ksplit = 1;
        g.flags = am ? GEMM_F16X2 : 0;
        g.amax_a = am;
        g.want = g_conv_want;
        flatten_rows(g, B, T);
        PGEMM_ON(SS_PROF_CONV_DX, g, s);
    }
    return 0;
}

// ---- BLSTM block -------------------------------------------------------------------------------------------
struct Chain {
    int b0, nb;
    hipStream_t st;
};

int make_chains(ss_engine* e, int B, hipStream_t s, Chain ch[2]) {
    if (g_split && e->side2 && B >= 32) {
        const int bA = ((B / 2 + 15) / 16) * 16;
        ch[0] = {0, bA, s};
        ch[1] = {bA, B - bA, e->side2};
        return 2;
    }
    ch[0] = {0, B, s};
    return 1;
}

// Per-step re-layouts of one BLSTM block's weights: b_ih + b_hh per (layer, direction) and, for the decoder-size blocks,
// the fragment-major W_hh of the forward recurrence.  Independent of the activations, so the schedule runs it on a branch.
int lstm_prep(ss_engine* e, LstmBlk& lb, PrepTable& tb, hipStream_t s) {
    const int H = lb.H;
    const bool persist = lb.big() && g_persist && lstm_seq_supported(e->curB, H);
    for (int l = 0; l < lb.L; ++l) {
        for (int dir = 0; dir < 2; ++dir) {
            const LstmDir& pd = lb.pd[l * 2 + dir];
            const long n = 4L * H * lb.in_of(l);
            if (tb.n + 2 > PREP_MAX) return fail("lstm_prep: task table full");
            tb.t[tb.n++] = {e->P + pd.bih, e->P + pd.bhh, lb.bsum + ((long)l * 2 + dir) * 4 * H, 4L * H};       // summed biases
            tb.t[tb.n++] = {e->P + pd.wih, nullptr, lb.wcat[l] + dir * n, n, (lb.wcat_img[l] && lb.in_of(l) % 8 == 0) ? e->ioff(lb.wcat_img[l], dir * n) : nullptr};      // stacked W_ih (+ image: rows of whole groups of eight)
        }
        // fragment-major W_hh for the one-launch-per-step schedule (the persistent kernels read the parameters directly)
        if (lb.big() && !persist) HIPCHK(lstm_pack_w(e->P + lb.pd[l * 2].whh, e->P + lb.pd[l * 2 + 1].whh, lb.wfrag[l], H, 0, s));
    }
    if (lb.big() && persist) HIPCHK(hipMemsetAsync(lb.zf, 0, lb.zf_bytes, s));     // h(-1) = 0 and the group counters of every layer
    return 0;
}

// Decoder-size BLSTM (one launch per time step).  The batch is cut in two halves that run as independent chains on
// two streams through ALL layers: every operator is per-utterance, so nothing couples them until the head.
int lstm_big_fwd(ss_engine* e, LstmBlk& lb, Slab x, hipStream_t s) {
    const int B = e->curB, T = e->curT, H = lb.H;
    const long TP = T + 2 * HALO;
    Chain ch[2];
    const bool persist = g_persist && lstm_seq_supported(B, H);
    const int nch = persist ? 1 : make_chains(e, B, s, ch);
    if (persist) ch[0] = {0, B, s};
    if (nch == 2) CHK(fork_join(e, s, ch[1].st));
    // only fp16 x 2 GEMMs read the hidden states' pre-split images, and only the persistent recurrences write them
    lb.out_img_valid = persist && (g_presplit & 2) && ((e->precision == SS_PRECISION_F32 && g_fwd_f16x2) || e->img16()) && !lb.out_img.empty() && lb.out_img[0];
    for (int l = 0; l < lb.L; ++l) {
        const int In = lb.in_of(l);
        Slab xi = l == 0 ? x : Slab{lb.out[l - 1], 2L * H};
        const bool compact = l == 0 && lb.xf && persist && x.p == lb.xc;       // the input repeats in blocks of xf frames: one projection row per block
        for (int c = 0; c < nch; ++c) {
            const long r0 = (long)ch[c].b0 * TP;
            if (compact) {
                GemmDesc d{};
                d.A = {lb.xc, In, 0, 0, 0};
                d.B = {lb.wcat[l], In, 0, 0, 0};
                d.b_pre = (g_presplit & 1) ? lb.wimg(l) : nullptr;
                d.C = lb.xp0;
                d.ldc = 8L * H;
                d.bias = lb.bsum + (long)l * 8 * H;
                d.M = B * (T / lb.xf);
                d.N = 8 * H;
                d.K = In;
                d.batch = 1;
                d.ksplit = 1;
                PGEMM_FWD_ON(SS_PROF_DEC_PROJ0, d, ch[c].st);
            } else {   // both directions in one GEMM against the stacked W_ih / summed biases of lstm_prep (N = 8H)
                GemmDesc d{};
                d.A = {xi.p + (r0 + HALO) * xi.ld, xi.ld, TP * xi.ld, 0, 0};
                if (l > 0 && lb.out_img_valid) d.a_pre = e->ioff(lb.out_img[l - 1], (r0 + HALO) * xi.ld);       // written by the layer below's recurrence
                d.B = {lb.wcat[l], In, 0, 0, 0};
                d.b_pre = (g_presplit & 1) ? lb.wimg(l) : nullptr;
                d.C = lb.gates[l] + (r0 + HALO) * 8L * H;
                d.ldc = 8L * H;
                d.cstride = TP * 8L * H;
                d.bias = lb.bsum + (long)l * 8 * H;
                d.M = T;
                d.N = 8 * H;
                d.K = In;
                d.batch = ch[c].nb;
                d.ksplit = 1;
                if (nch == 1) flatten_rows(d, B, T);
                PGEMM_FWD_ON(l > 0 ? SS_PROF_DEC_PROJ : SS_PROF_DEC_PROJ0, d, ch[c].st);
            }
            if (!persist) {
                const long half = 2L * (((ch[c].nb + 15) / 16) * 16) * H;
                HIPCHK(hipMemsetAsync(lb.hf[c], 0, 2 * half * 4, ch[c].st));
            }
        }
        if (persist) {   // start state zeroed by lstm_prep
            // the input projections the GEMM has just written are read once more, in whole lines, beside the recurrence (lstm_seq.hip)
            // (worth it from ~48 utterances on: 64 x 128 -0.08 ms; at 32 and 16 the fork / join around the recurrence costs more than warm operands
            // gain: 32 x 128 bf16 3.00 vs 2.97 ms without, 16 x 128 3.28 vs 3.26)
            const bool pw = (g_prewarm & 2) && e->side3 && g_overlap && !compact && B > 32;
            if (pw) {
                CHK(fork_join(e, s, e->side3));
                HIPCHK(slab_prewarm(lb.gates[l], 8 * H, nullptr, nullptr, 2 * H, e->amax, B, T, false, e->side3));
            }
            const int pi = prof_begin(e, SS_PROF_REC_FWD, s, 2.0 * 2 * B * T * 4.0 * H * H);
            HIPCHK(lstm_seq_fwd(lb.gates[l], e->P + lb.pd[l * 2].whh, e->P + lb.pd[l * 2 + 1].whh, lb.hf_l(l), lb.out[l], lb.csave[l],
                                lb.sync_f(l), e->sticky, compact ? lb.xp0 : nullptr, compact ? lb.xf : 0, lb.out_img_valid ? lb.out_img[l] : nullptr, B, T, H,
                                false, false, s, (e->img16() ? 1 : 0) | (e->img16() && g_seq_hi ? 2 : 0)));
            prof_end(e, pi, s);
            if (pw) CHK(fork_join(e, e->side3, s));
            continue;
        }
        for (int st = 0; st < T; ++st)
            for (int c = 0; c < nch; ++c) {
                const long r0 = (long)ch[c].b0 * TP;
                const long half = 2L * (((ch[c].nb + 15) / 16) * 16) * H;
                HIPCHK(lstm_step_fwd(lb.gates[l] + r0 * 8L * H, lb.wfrag[l], lb.hf[c] + (st & 1) * half,
                                     lb.hf[c] + ((st & 1) ^ 1) * half, lb.out[l] + r0 * 2L * H, lb.csave[l] + r0 * 2L * H,
                                     ch[c].nb, T, H, st, ch[c].st));
            }
    }
    if (nch == 2) CHK(fork_join(e, ch[1].st, s));
    return 0;
}

// the decoder's layer 0 runs on the compact (one row per block of repeated frames) form of its input
static bool dec_compact(const ss_engine* e) { return g_compact0 && e->ld.xf > 0 && e->ld.big() && g_persist && lstm_seq_supported(e->curB, e->ld.H); }

int lstm_fwd(ss_engine* e, LstmBlk& lb, Slab x, hipStream_t s) {
    if (lb.big()) return lstm_big_fwd(e, lb, x, s);
    const int B = e->curB, T = e->curT, H = lb.H;
    const long TP = T + 2 * HALO;
    for (int l = 0; l < lb.L; ++l) {
        const int In = lb.in_of(l);
        Slab xi = l == 0 ? x : Slab{lb.out[l - 1], 2L * H};
        {   // both directions in one GEMM (stacked W_ih and summed biases from lstm_prep)
            GemmDesc d{};
            d.A = {xi.p + HALO * xi.ld, xi.ld, TP * xi.ld, 0, 0};
            d.a_pre_scale = xi.scale;                  // conv-block output (layer 0); hidden states |h| < 1 take the fixed 16
            d.B = {lb.wcat[l], In, 0, 0, 0};
            d.b_pre = (g_presplit & 1) ? lb.wimg(l) : nullptr;
            d.C = lb.gates[l] + HALO * 8L * H;
            d.ldc = 8L * H;
            d.cstride = TP * 8L * H;
            d.bias = lb.bsum + (long)l * 8 * H;
            d.M = T;
            d.N = 8 * H;
            d.K = In;
            d.batch = B;
            d.ksplit = 1;
            flatten_rows(d, B, T);
            PGEMM_FWD_ON(SS_PROF_ENC_LSTM, d, s);
        }
        { const int pa_ = prof_begin(e, SS_PROF_ENC_REC, s, 0.0);
        HIPCHK(lstm_small_fwd(lb.gates[l], e->P + lb.pd[l * 2].whh, e->P + lb.pd[l * 2 + 1].whh, lb.out[l], lb.csave[l], B, T, H,
                              s));
        prof_end(e, pa_, s); }
    }
    return 0;
}

// d_top: gradient slab of the last layer's output [B,TP,2H]; x: forward input; dx: input-gradient view or null
// weight / bias gradients of one layer from its finished pre-activation gradient slab (full batch, flat over all rows)
// W_ih gradient of decoder layer l (both directions, one launch) as a work-queue image GEMM, in two calls: lstm_wih_split makes the image
// of the layer's pre-activation gradients (split_image; as soon as its recurrence is through), lstm_wih_gemm_queued contracts it against
// the layer input's image (the hidden states of layer l - 1, written by the forward recurrence).  `gate` (nullable): the sync words of the
// persistent recurrence the launch is meant to run beside -- the GEMM is dispatched once that grid is resident (seq_gate).
bool lstm_wih_queue_ok(ss_engine* e, LstmBlk& lb, int l, const float* am) {
    if (!g_img || !am || &lb != &e->ld || l < 1 || l >= 3 || !e->dg_img[l] || e->precision != SS_PRECISION_F32 || !g_bwd_f16x2) return false;
    if (!lb.out_img_valid || !lb.out_img[l - 1] || !e->wq_pool || e->wq_next >= ss_engine::WQ_SLOTS) return false;
    return lb.pd[l * 2 + 1].wih > lb.pd[l * 2].wih;
}
int lstm_wih_split(ss_engine* e, LstmBlk& lb, int l, const float* am, hipStream_t ws) {
    const long TP = e->curT + 2 * HALO, R = (long)e->curB * TP;
    HIPCHK(split_image(lb.gates[l], 8L * lb.H, R, 8 * lb.H, am, 0.f, e->dg_img[l], 8L * lb.H, e->gscale + lb.amax0 + l, ws));
    return 0;
}
int lstm_wih_gemm_queued(ss_engine* e, LstmBlk& lb, int l, const float* am, const unsigned* gate, hipStream_t ws) {
    const int B = e->curB, T = e->curT, H = lb.H;
    const long TP = T + 2 * HALO, R = (long)B * TP;
    const int In = lb.in_of(l);
    const LstmDir &p0 = lb.pd[l * 2], &p1 = lb.pd[l * 2 + 1];
    if (gate) HIPCHK(seq_gate(gate, B, H, ws));
    GemmDesc a{};
    a.A = {lb.gates[l], 8L * H, 4L * H, 0, 0};
    a.B = {lb.out[l - 1], 2L * H, 0, 0, 0};
    a.a_pre = e->dg_img[l];
    a.a_pre_scale = e->gscale + lb.amax0 + l;
    a.b_pre = lb.out_img[l - 1];
    a.C = e->G + p0.wih;
    a.ldc = In;
    a.cstride = p1.wih - p0.wih;
    a.M = 4 * H;
    a.N = In;
    a.K = (int)R;
    a.batch = 2;
    a.flags = GEMM_TA | GEMM_TB | GEMM_ACCUM | GEMM_F16X2;
    a.amax_a = am;
    a.ksplit = 4;                   // > 1: the image path cuts the reduction into partial slabs
    a.queue = 1;
    const int pi = prof_begin(e, SS_PROF_DEC_DW, ws, 2.0 * a.M * a.N * (double)B * T * a.batch);
    g_cur_klass = SS_PROF_DEC_DW;
    const int r = try_img_gemm(e, a, ws);
    g_cur_klass = -1;
    prof_end(e, pi, ws);
    if (r < 0) return r;
    if (r == 0) return fail("lstm_wih_gemm_queued: the image GEMM refused a shape the engine planned for it");
    e->dec_ih_done |= 1 << l;
    return 0;
}

// part: 0 everything; 2 everything except the W_ih gradient (it went out through lstm_wih_grad_queued)
int lstm_weight_grads(ss_engine* e, LstmBlk& lb, int l, Slab xi, const float* am, bool bias_done, hipStream_t ws, int part = 0) {
#ifdef SS_DIAG      // what-if timing runs (WRONG gradients; -DSS_DIAG library only): the step without the decoder's / the encoder BLSTMs' weight-gradient launches
    if ((g_exp & 4) && lb.big()) return 0;
    if ((g_exp & 16) && !lb.big()) return 0;
#endif
    const int B = e->curB, T = e->curT, H = lb.H;
    const long TP = T + 2 * HALO, R = (long)B * TP;
    const int In = lb.in_of(l);
    float* dG = lb.gates[l];
    const LstmDir &p0 = lb.pd[l * 2], &p1 = lb.pd[l * 2 + 1];
    if (!lb.big() && g_wgrad_fused && H <= 32 && !bias_done && e->part && e->colsum_ctr && e->wg.n < WGRAD_MAX &&
        e->part_off + (long)(e->wg.tiles_total + lstm_small_wgrad_tiles(H, In)) * 16 * 4096 <= e->part_cap) {
        // encoder BLSTMs: one fused fp32 kernel for the weight and bias gradients of every layer (lstm_wgrad.hip)
        WgradTask& t = e->wg.t[e->wg.n];
        t = WgradTask{dG, xi.p, xi.ld, lb.out[l], e->G + p0.wih, e->G + p1.wih, e->G + p0.whh, e->G + p1.whh, e->G + p0.bih, e->G + p0.bhh, e->G + p1.bih,
                      e->G + p1.bhh, H, In, R, e->wg.tiles_total};
        e->wg.tiles_total += lstm_small_wgrad_tiles(H, In);
        ++e->wg.n;
        if (!e->wg_defer) CHK(wgrad_flush(e, ws));
        return 0;
    }
    const bool compact = l == 0 && lb.xf && xi.p == lb.xc;     // dW_ih from the block sums and one input row per block (K / xf)
    // the hidden-state slabs of a decoder-size block on the persistent kernels also exist as pre-split images (written by the forward)
    const bool img_ok = lb.out_img_valid && lb.out_img[l];
    // the decoder's pre-activation gradients as an image for the image GEMM (scale: the power of two for the maximum the recurrence measured)
    const float* dimg = nullptr;
    const float* dsc = nullptr;
    const bool i16 = e->img16();
    if (i16) {
        if (g_img && &lb == &e->ld && l < 3 && ((e->dg16_written >> l) & 1) && img_ok) dimg = e->dg_img[l];      // written by the backward recurrence's storing wave
    } else if (g_img && ((g_img_mask >> SS_PROF_DEC_DW) & 1) && am && &lb == &e->ld && l < 3 && e->dg_img[l] && e->precision == SS_PRECISION_F32 && img_ok) {
        HIPCHK(split_image(dG, 8L * H, R, 8 * H, am, 0.f, e->dg_img[l], 8L * H, e->gscale + lb.amax0 + l, ws));
        dimg = e->dg_img[l];
        dsc = e->gscale + lb.amax0 + l;
    }
    // Both directions in ONE launch each (batch = 2) when their parameters sit at one stride in the arena (PyTorch's order: they do).
    // dW_hh: h_prev is `out` one row earlier (forward) / later (reverse), so the forward direction reads dG one row later instead.
    // (With the image GEMM the decoder's launches are batched as well: 64 + 32 tile jobs per layer instead of 4 x (32 or 16) halve the
    // split-K factor, i.e. the partial-slab traffic and the number of reduce passes.)
    const bool img_batch = dimg && g_img_batch && (i16 ? ((g_bf16_img_mask >> SS_PROF_DEC_DW) & 1) != 0 : ((g_img_mask >> SS_PROF_DEC_DW) & 1) != 0);
    if ((!compact || img_batch) && (g_batch_dirs == 2 || (g_batch_dirs == 1 && !lb.big()) || img_batch) && p1.wih - p0.wih == p1.whh - p0.whh && p1.wih > p0.wih) {
        const long pstride = p1.wih - p0.wih;
        GemmDesc a{};
        a.A = {dG, 8L * H, 4L * H, 0, 0};
        a.B = {xi.p, xi.ld, 0, 0, 0};
        a.b_pre_scale = l == 0 ? xi.scale : nullptr;      // a conv block's output carries its own split scale (act_scales): right whichever product mode runs
        a.C = e->G + p0.wih;
        a.ldc = In;
        a.cstride = pstride;
        a.M = 4 * H;
        a.N = In;
        a.K = (int)R;
        a.batch = 2;
        a.flags = GEMM_TA | GEMM_TB | GEMM_ACCUM | (am ? GEMM_F16X2 : 0);
        a.amax_a = am;                              // gradient slab: measured scale; the layer input is O(1)
        a.ksplit = pick_ksplit(a.M, a.N, a.K);
        a.queue = e->wq_mode ? 1 : 0;
        if (dimg && l > 0) {
            a.a_pre = dimg;
            a.a_pre_scale = dsc;
            a.b_pre = lb.out_img[l - 1];
        }
        if (!compact && part != 2) PGEMM_ON(lb.big() ? SS_PROF_DEC_DW : SS_PROF_ENC_LSTM, a, ws);
        GemmDesc h{};
        h.A = {dG + 8L * H, 8L * H, 4L * H - 8L * H, 0, 0};                    // forward: rows 1 .., reverse: rows 0 .. of its own columns
        h.B = {lb.out[l], 2L * H, 2L * H + H, 0, 0};                           // forward: rows 0 .. of h_f, reverse: rows 1 .. of h_b
        if (dimg) {
            h.a_pre = e->ioff(dimg, 8L * H);
            h.a_pre_scale = dsc;
            h.b_pre = lb.out_img[l];
        }
        h.C = e->G + p0.whh;
        h.ldc = H;
        h.cstride = pstride;
        h.M = 4 * H;
        h.N = H;
        h.K = (int)(R - 1);
        h.batch = 2;
        h.flags = GEMM_TA | GEMM_TB | GEMM_ACCUM | (am ? GEMM_F16X2 : 0);
        h.amax_a = am;
        h.ksplit = pick_ksplit(h.M, h.N, h.K);
        h.queue = e->wq_mode ? 1 : 0;
        PGEMM_ON(lb.big() ? SS_PROF_DEC_DW : SS_PROF_ENC_LSTM, h, ws);
        if (!bias_done) {
            double* cpart;
            unsigned* cctr;
            colsum_scratch(e, 8 * H, &cpart, &cctr);
            HIPCHK(colsum_bias(dG, 8L * H, (int)R, 4 * H, e->G + p0.bih, e->G + p0.bhh, e->G + p1.bih, e->G + p1.bhh, cpart, cctr, ws));
        }
        if (!compact) return 0;
        // compact layer 0: dW_hh went out batched above, dW_ih comes from the block sums below, per direction
        for (int dir = 0; dir < 2; ++dir) {
            const LstmDir& pd = lb.pd[l * 2 + dir];
            GemmDesc c{};
            c.A = {lb.dgs + dir * 4L * H, 8L * H, 0, 0, 0};
            c.B = {xi.p, xi.ld, 0, 0, 0};
            c.C = e->G + pd.wih;
            c.ldc = In;
            c.M = 4 * H;
            c.N = In;
            c.K = B * (T / lb.xf);
            c.batch = 1;
            c.flags = GEMM_TA | GEMM_TB | GEMM_ACCUM | (am ? GEMM_F16X2 : 0);
            c.amax_a = am;
            c.ksplit = pick_ksplit(c.M, c.N, c.K);
            PGEMM_ON(SS_PROF_DEC_DW, c, ws);
            if (&lb == &e->ld && e->dp_on && part == 0 && pd.bhh + 4L * H > pd.wih) {      // data parallel: half a layer per collective (see below)
                CHK(dp_bucket(e, pd.wih, pd.bhh + 4L * H - pd.wih, ws));
                e->dp_dir_buckets = true;
            }
        }
        return 0;
    }
    for (int dir = 0; dir < 2; ++dir) {
        const LstmDir& pd = lb.pd[l * 2 + dir];
        const float* dGd = dG + dir * 4L * H;
        // dW_ih[n][k] = sum_r dG[r][n] * X[r][k]
        GemmDesc a{};
        a.A = {compact ? lb.dgs + dir * 4L * H : dGd, 8L * H, 0, 0, 0};
        if (dimg && !compact) {
            a.a_pre = e->ioff(dimg, a.A.p - dG);
            a.a_pre_scale = dsc;
        }
        a.B = {xi.p, xi.ld, 0, 0, 0};
        if (img_ok && l > 0) a.b_pre = lb.out_img[l - 1];
        a.C = e->G + pd.wih;
        a.ldc = In;
        a.M = 4 * H;
        a.N = In;
        a.K = compact ? B * (T / lb.xf) : (int)R;
        a.batch = 1;
        a.flags = GEMM_TA | GEMM_TB | GEMM_ACCUM | (am ? GEMM_F16X2 : 0);
        a.amax_a = am;                              // gradient slab: measured scale; the layer input is O(1)
        a.ksplit = pick_ksplit(a.M, a.N, a.K);
        hipStream_t wa = (dir == 1 && e->dw_over[0]) ? e->dw_over[0] : ws, wh = (dir == 1 && e->dw_over[1]) ? e->dw_over[1] : ws;
        if (part != 2 || compact) PGEMM_ON(lb.big() ? SS_PROF_DEC_DW : SS_PROF_ENC_LSTM, a, wa);
        // dW_hh[n][k] = sum_r dG[r][n] * h_prev[r][k];  h_prev = out one row earlier (fwd) / later (reverse)
        GemmDesc h{};
        h.A = {dir == 0 ? dGd + 8L * H : dGd, 8L * H, 0, 0, 0};
        if (dimg) {
            h.a_pre = e->ioff(dimg, h.A.p - dG);
            h.a_pre_scale = dsc;
        }
        h.B = {dir == 0 ? lb.out[l] : lb.out[l] + 2L * H + H, 2L * H, 0, 0, 0};
        if (img_ok) h.b_pre = dir == 0 ? lb.out_img[l] : e->ioff(lb.out_img[l], 2L * H + H);
        h.C = e->G + pd.whh;
        h.ldc = H;
        h.M = 4 * H;
        h.N = H;
        h.K = (int)(R - 1);
        h.batch = 1;
        h.flags = GEMM_TA | GEMM_TB | GEMM_ACCUM | (am ? GEMM_F16X2 : 0);
        h.amax_a = am;
        h.ksplit = pick_ksplit(h.M, h.N, h.K);
        PGEMM_ON(lb.big() ? SS_PROF_DEC_DW : SS_PROF_ENC_LSTM, h, wh);
        if (!bias_done) {     // the persistent backward kernel accumulates both bias gradients itself
            double* cpart;
            unsigned* cctr;
            colsum_scratch(e, 4 * H, &cpart, &cctr);
            HIPCHK(colsum_acc(dGd, 8L * H, (int)R, 4 * H, e->G + pd.bih, cpart, cctr, ws));
            HIPCHK(hipMemcpyAsync(e->G + pd.bhh, e->G + pd.bih, 4L * H * 4, hipMemcpyDeviceToDevice, ws));
        }
        // data parallel: this direction's parameters (W_ih, W_hh, b_ih, b_hh: contiguous in the arena) are final -- half a layer per
        // collective keeps the communication stream busy 140 us earlier than a bucket per layer (modelled N = 8: the last collective ends
        // closer to the backward's end)
        if (&lb == &e->ld && e->dp_on && part == 0 && pd.bhh + 4L * H > pd.wih) {
            if (wa != wh) return fail("internal: a direction's weight gradients on two streams under data parallelism");
            CHK(dp_bucket(e, pd.wih, pd.bhh + 4L * H - pd.wih, wh));
            e->dp_dir_buckets = true;
        }
    }
    return 0;
}

// input gradient of one layer for the slab rows [r0, r0 + nr):  dX = dG . W_ih  (both directions accumulate)
int lstm_input_grad(ss_engine* e, LstmBlk& lb, int l, Slab dxi, long r0, long nr, const float* am, hipStream_t st) {
    if (l == 0 && lb.xf && dxi.p == lb.d_xc) {
        // compact: d_xc[block] = (sum of dG over the block's frames) . W_ih, one row per block of repeated input frames
        const int H = lb.H, In = lb.in_of(0);
        const long R8 = (long)e->curB * (e->curT / lb.xf);
        GemmDesc g{};
        g.A = {lb.dgs, 8L * H, 0, 0, 0};
        g.B = {lb.wcat[0], In, 0, 0, 0};
        g.b_pre = (g_presplit & 1) ? lb.wimg(0) : nullptr;
        g.C = lb.d_xc;
        g.ldc = In;
        g.M = (int)R8;
        g.N = lb.xcols > 0 ? lb.xcols : In;        // the speaker columns (last in the row, model.py:308-309) are an input: nobody reads their gradient
        g.K = 8 * H;
        g.batch = 1;
        g.flags = GEMM_TB | GEMM_ACCUM | (am ? GEMM_F16X2 : 0);
        g.amax_a = am;                              // the block sums are at most xf times the slab's maximum: inside the scale's headroom (256x)
        g.ksplit = 8;
        HIPCHK(hipMemsetAsync(g.C, 0, R8 * In * 4, st));
        PGEMM_ON(SS_PROF_DEC_DX, g, st);
        return 0;
    }
    // dX[r][k] = sum over both directions' 8H gate units of dG[r][n] * W_ih[n][k]: ONE GEMM against the stacked weights
    // (lstm_prep).  A narrow input (the decoder's 164 columns) is cut along the reduction so the launch still fills the chip.
    const int H = lb.H, In = lb.in_of(l), T = e->curT;
    const long TPr = T + 2 * HALO;
    GemmDesc g{};
    g.A = {lb.gates[l] + r0 * 8L * H, 8L * H, 0, 0, 0};
    g.B = {lb.wcat[l], In, 0, 0, 0};
    // (round 2's kernel measured SLOWER with the weight image in format v2 on this transposing-read operand -- 573 us per step with it, 532
    // without -- so it only gets that image when asked to, presplit bit 3)
    g.b_pre = (((g_presplit & 8) || e->img16()) && (g_presplit & 1)) ? lb.wimg(l) : nullptr;
    const bool dg16 = e->img16() && &lb == &e->ld && l < 3 && ((e->dg16_written >> l) & 1);      // the gradient slab's bf16 image (backward recurrence's storing wave)
    if (dg16) g.a_pre = e->ioff(e->dg_img[l], r0 * 8L * H);
    g.C = dxi.p + r0 * dxi.ld;
    g.ldc = dxi.ld;
    g.M = (int)nr;
    g.N = In;
    g.K = 8 * H;
    g.batch = 1;
    g.flags = GEMM_TB | (am ? GEMM_F16X2 : 0);
    g.amax_a = am;                                  // gradient slab: measured scale; the stacked weights are O(1)
    g.ksplit = 1;
    const long tiles = (long)cdiv(g.M, 128) * cdiv(g.N, 64);
    if (tiles < 512 && g.K >= 1024 && dxi.ld == In) {          // split-K needs a zeroed, dense C
        g.ksplit = tiles < 256 ? 4 : 2;
        g.flags |= GEMM_ACCUM;
        HIPCHK(hipMemsetAsync(g.C, 0, nr * dxi.ld * 4, st));
    } else if (g_dx_batched && nr % TPr == 0 && r0 % TPr == 0 && T % 64 == 0) {
        // whole utterances: leave the halo rows out (nobody reads them in a gradient slab) -- one batch entry per
        // utterance, T rows each (T a multiple of 64), which at T = 128 also makes the 128 x 128 tiling come out at exactly 2 workgroups per CU for B = 64
        g.A = {lb.gates[l] + (r0 + HALO) * 8L * H, 8L * H, TPr * 8L * H, 0, 0};
        if (dg16) g.a_pre = e->ioff(e->dg_img[l], (r0 + HALO) * 8L * H);
        g.C = dxi.p + (r0 + HALO) * dxi.ld;
        g.cstride = TPr * dxi.ld;
        g.M = T;
        g.batch = (int)(nr / TPr);
        g.want = g_dx_batched == 2 ? 512 : 0;
    }
    PGEMM_ON(lb.big() ? SS_PROF_DEC_DX : SS_PROF_ENC_LSTM, g, st);
    return 0;
}

// d_top: gradient slab of the last layer's output [B,TP,2H]; x: forward input; dx: input-gradient view or null
// dx_ready (nullable): recorded on `s` right behind the input-gradient GEMM of layer 0, i.e. BEFORE this block's weight-gradient
// launches go to their branch stream.  A consumer that waits for it is not held up by whatever else shares a hardware queue with
// `s`: an event recorded later would sit in that queue behind every packet enqueued in between (measured: the conv trunk's
// backward idled 0.88 ms behind ~36 tiny weight-gradient launches of a sibling stream, profiles/r02/step_timeline_before.txt).
int lstm_late_weights(ss_engine* e, LstmBlk& lb, Slab x, hipStream_t ws, int l_hi = -1, int l_lo = 0);

// late_w: enqueue the recurrence chain and the input gradients only; the caller enqueues the block's weight gradients later
// (lstm_late_weights) -- after the phase's critical path -- on a stream it has ordered behind this chain.
int lstm_bwd(ss_engine* e, LstmBlk& lb, const float* d_top, Slab x, Slab dx, hipStream_t s, hipEvent_t dx_ready = nullptr, bool late_w = false) {
    const int B = e->curB, T = e->curT, H = lb.H;
    const long TP = T + 2 * HALO, R = (long)B * TP;
    const float* dcur = d_top;
    Chain ch[2];
    ch[0] = {0, B, s};
    const bool persist = lb.big() && g_persist && lstm_seq_supported(B, H);
    const int nch = (lb.big() && !persist) ? make_chains(e, B, s, ch) : 1;
    if (lb.big() && nch == 2) CHK(fork_join(e, s, ch[1].st));
    // see ss_engine::wq_pool: only with the weight gradients deferred (their usual schedule), on the decoder, where XCDs stay free
    const bool xcd = persist && &lb == &e->ld && g_xcd_dw && e->side && g_overlap && (g_defer_dw || late_w) && !g_deterministic &&
                     lstm_seq_free_xcds(B, H) >= 2;
    bool xcd_split[4] = {false, false, false, false};
    // 16-bit data path: the recurrence writes the gradient image itself and every weight-gradient GEMM of a layer is one image launch per
    // operand pair, so the WHOLE layer above goes out beside this layer's recurrence (work-queue form: 144 KB workgroups, one per free CU)
    const bool early16 = persist && &lb == &e->ld && g_early_dw && e->img16() && g_img && g_img_batch && e->side && g_overlap && (g_defer_dw || late_w) &&
                         !g_deterministic && !e->dp_on && e->wq_pool && lstm_seq_free_xcds(B, H) >= 2 && lb.out_img_valid;
    bool early_ready[4] = {false, false, false, false};
    for (int l = lb.L - 1; l >= 0; --l) {
        Slab xi = l == 0 ? x : Slab{lb.out[l - 1], 2L * H};
        Slab dxi = l == 0 ? dx : Slab{lb.dmid[l & 1], 2L * H};
        float* dG = lb.gates[l];
        hipStream_t ws = s;
        // fp16 x 2 gradient GEMMs need the slab's maximum, which only the persistent kernel measures
        float* am = (persist && g_bwd_f16x2 && lb.amax0 >= 0) ? e->amax + lb.amax0 + l : nullptr;
        // b_ih and b_hh gradients sit back to back in the arena (registration order): the persistent kernel fills both
        const bool bias_in_kernel = persist && !g_deterministic && lb.pd[l * 2].bhh == lb.pd[l * 2].bih + 4L * H &&
                                    lb.pd[l * 2 + 1].bhh == lb.pd[l * 2 + 1].bih + 4L * H;
        if (lb.big()) {
            for (int c = 0; c < nch && !persist; ++c) {
                const long half = 2L * (((ch[c].nb + 15) / 16) * 16) * 4 * H;
                HIPCHK(hipMemsetAsync(lb.gf[c], 0, 2 * half * 4, ch[c].st));
            }
            // persistent: start state zeroed by backward_decoder
            if (persist) {
                // activated gates and cell states of the forward pass are long gone from the caches: one streaming read beside the
                // recurrence puts them into the memory-side cache ahead of its 64-byte requests (lstm_seq.hip, slab_prewarm_kernel)
                const bool pw = (g_prewarm & 1) && e->side3 && g_overlap;
                if (pw) {
                    CHK(fork_join(e, s, e->side3));
                    HIPCHK(slab_prewarm(dG, 8 * H, lb.csave[l], dcur, 2 * H, e->amax, B, T, false, e->side3));
                }
                const bool dg16 = e->img16() && g_img && &lb == &e->ld && l < 3 && e->dg_img[l] && nch == 1;
                // only as bf16 when every reader takes the image: weight gradients and input gradient on the image GEMM (layers >= 1: aligned
                // shapes, both images present; layer 0: compact form -- dW_ih and dX from the block sums, dW_hh from the image), bias sums in the
                // kernel (not the deterministic mode's column sum over the fp32 slab), weight gradients deferred or not: same readers
                const bool compact0 = l == 0 && lb.xf && dx.p == lb.d_xc;
                const bool skip32 = dg16 && g_seq_skip32 && bias_in_kernel && lb.out_img_valid && lb.out_img[l] && g_img_batch &&
                                    ((g_bf16_img_mask >> SS_PROF_DEC_DW) & 1) && ((g_bf16_img_mask >> SS_PROF_DEC_DX) & 1) &&
                                    (l == 0 ? compact0 : (lb.wimg(l) != nullptr && (g_presplit & 1) && lb.in_of(l) % 64 == 0)) && H % 64 == 0;
                const int pi = prof_begin(e, SS_PROF_REC_BWD, s, 2.0 * 2 * B * T * 4.0 * H * H);
                HIPCHK(lstm_seq_bwd(dG, e->P + lb.pd[l * 2].whh, e->P + lb.pd[l * 2 + 1].whh, lb.px_l(l), dcur, lb.csave[l], lb.sync_b(l),
                                    e->sticky, am, bias_in_kernel ? e->G + lb.pd[l * 2].bih : nullptr,
                                    bias_in_kernel ? e->G + lb.pd[l * 2 + 1].bih : nullptr, (l == 0 && lb.xf && dx.p == lb.d_xc) ? lb.dgs : nullptr,
                                    (l == 0 && lb.xf && dx.p == lb.d_xc) ? lb.xf : 0, B, T, H, false, false, s, dg16 ? e->dg_img[l] : nullptr, (e->img16() && g_seq_hi ? 1 : 0) | (skip32 ? 4 : 0)));
                if (dg16) e->dg16_written |= 1 << l;
                if (skip32) e->dg32_skipped |= 1 << l;
                prof_end(e, pi, s);
                if (pw) CHK(fork_join(e, e->side3, s));
                if (early16) {
                    if (l + 1 < lb.L && early_ready[l + 1]) {
                        HIPCHK(seq_gate(lb.sync_b(l), B, H, e->side));          // dispatched once this recurrence's grid is resident
                        e->wq_mode = true;
                        const int rc = lstm_weight_grads(e, lb, l + 1, Slab{lb.out[l], 2L * H}, am ? e->amax + lb.amax0 + l + 1 : nullptr, true, e->side);
                        e->wq_mode = false;
                        CHK(rc);
                        e->dec_w_done |= 1 << (l + 1);
                    }
                    if (l >= 1 && l < 3 && dg16 && bias_in_kernel && lb.out_img[l] && lb.out_img[l - 1] && e->wq_next + 2 <= ss_engine::WQ_SLOTS) {
                        CHK(fork_join(e, s, e->side));        // layer l's slab and image are complete behind this point
                        early_ready[l] = true;
                        e->side_used = true;
                    }
                }
                // XCD-aware weight gradients: this recurrence leaves XCDs free, the layer above is through -- its W_ih gradient runs beside it
                if (xcd) {
                    if (l + 1 < lb.L && xcd_split[l + 1]) {
                        const float* am1 = e->amax + lb.amax0 + l + 1;
                        CHK(lstm_wih_gemm_queued(e, lb, l + 1, am1, lb.sync_b(l), e->side));
                    }
                    if (l >= 1 && lstm_wih_queue_ok(e, lb, l, am)) {       // this layer's turn comes beside the next recurrence: its image as soon as it is through
                        CHK(fork_join(e, s, e->side));
                        CHK(lstm_wih_split(e, lb, l, am, e->side));
                        xcd_split[l] = true;
                        e->side_used = true;
                    }
                }
            }
            for (int st = 0; st < T && !persist; ++st)
                for (int c = 0; c < nch; ++c) {
                    const long r0 = (long)ch[c].b0 * TP;
                    const long half = 2L * (((ch[c].nb + 15) / 16) * 16) * 4 * H;
                    HIPCHK(lstm_step_bwd(dG + r0 * 8L * H, lb.wfrag[l], lb.gf[c] + (st & 1) * half, lb.gf[c] + ((st & 1) ^ 1) * half,
                                         dcur + r0 * 2L * H, lb.csave[l] + r0 * 2L * H, lb.dc[c], ch[c].nb, T, H, st, ch[c].st));
                }
        } else {
            { const int pa_ = prof_begin(e, SS_PROF_ENC_REC, s, 0.0);
            HIPCHK(lstm_small_bwd(dG, e->P + lb.pd[l * 2].whh, e->P + lb.pd[l * 2 + 1].whh, dcur, lb.csave[l], B, T, H, s));
            prof_end(e, pa_, s); }
        }
        // Decoder on the persistent kernels: its weight-gradient GEMMs are held back until the whole recurrence chain
        // (layer L-1 .. 0 and the input gradients between them) is through.  Co-scheduled they do not fill idle cycles:
        // they stretch the latency-bound recurrence steps and halve the rate of the input-gradient GEMMs on the critical
        // path (measured: chain 2.95 ms with the GEMMs beside it against 1.93 ms alone + 0.78 ms of GEMMs), whereas the
        // encoder backward that follows is a string of small launches they can run beside.
        const bool defer = (persist && e->side && g_overlap && g_defer_dw) || (late_w && nch == 1);
        if (defer) {
            if (dxi.p) CHK(lstm_input_grad(e, lb, l, dxi, 0, R, am, s));
            if (l == 0 && dx_ready) HIPCHK(hipEventRecord(dx_ready, s));
            dcur = dxi.p;
            continue;
        }
        // the pre-activation gradients of this layer are complete once every chain has passed this point
        if (e->side && g_overlap) {
            // decoder: the side stream.  Encoder BLSTMs (a string of ~36 tiny split-K launches): the third branch stream, so
            // that they do not queue behind the decoder's weight gradients, which drain on the side stream until late
            ws = (!lb.big() && e->side3) ? e->side3 : e->side;
            for (int c = 0; c < nch; ++c)
                if (ch[c].st != ws) CHK(fork_join(e, ch[c].st, ws));
            if (ws == e->side) e->side_used = true;
        } else if (nch == 2) {
            CHK(fork_join(e, ch[1].st, s));        // weight gradients run on s and need both halves
        }
        // input gradient first (the next layer's recurrence needs it), per chain on its own rows
        if (dxi.p)
            for (int c = 0; c < nch; ++c)
                CHK(lstm_input_grad(e, lb, l, dxi, (long)ch[c].b0 * TP, nch == 2 ? (long)ch[c].nb * TP : R, am, ch[c].st));
        if (l == 0 && dx_ready && nch == 1) HIPCHK(hipEventRecord(dx_ready, ch[0].st));
        CHK(lstm_weight_grads(e, lb, l, xi, am, bias_in_kernel, ws));
        dcur = dxi.p;
    }
    if (nch == 2) CHK(fork_join(e, ch[1].st, s));
    if (persist && e->side && g_overlap && g_defer_dw && !late_w) {
        CHK(fork_join(e, s, e->side));
        e->side_used = true;
        CHK(lstm_late_weights(e, lb, x, e->side));
    }
    return 0;
}

// all layers' weight / bias gradients of a block whose chain ran with deferred weights (the decoder's deferred batch; every block under late_w)
int lstm_late_weights(ss_engine* e, LstmBlk& lb, Slab x, hipStream_t ws, int l_hi, int l_lo) {
    const int H = lb.H;
    const bool persist = lb.big() && g_persist && lstm_seq_supported(e->curB, H);
    for (int l = l_hi < 0 ? lb.L - 1 : l_hi; l >= l_lo; --l) {       // layers l_hi .. l_lo (default: all, last first)
        if (&lb == &e->ld && ((e->dec_w_done >> l) & 1)) continue;        // went out beside a recurrence (early_dw)
        Slab xi = l == 0 ? x : Slab{lb.out[l - 1], 2L * H};
        float* am = (persist && g_bwd_f16x2 && lb.amax0 >= 0) ? e->amax + lb.amax0 + l : nullptr;
        const bool bias_in_kernel = persist && !g_deterministic && lb.pd[l * 2].bhh == lb.pd[l * 2].bih + 4L * H && lb.pd[l * 2 + 1].bhh == lb.pd[l * 2 + 1].bih + 4L * H;
        e->dp_dir_buckets = false;
        CHK(lstm_weight_grads(e, lb, l, xi, am, bias_in_kernel, ws, (&lb == &e->ld && ((e->dec_ih_done >> l) & 1)) ? 2 : 0));
        // this layer's gradients (both directions: W_ih, W_hh, b_ih, b_hh -- contiguous in the arena) are final (unless each direction has
        // already gone out on its own)
        if (&lb == &e->ld && !e->dp_dir_buckets) CHK(dp_bucket(e, lb.pd[l * 2].wih, lb.pd[l * 2 + 1].bhh + 4L * H - lb.pd[l * 2].wih, ws));
    }
    return 0;
}

// every gradient enqueued on the side stream so far is complete on `s` after this
int join_side(ss_engine* e, hipStream_t s) {
    if (e->side_used) {
        CHK(fork_join(e, e->side, s));
        e->side_used = false;
    }
    return 0;
}

// Scale of the fp16 x 2 split of every conv block's output, from its GroupNorm affine (kernels.h act_scales): 16 for any sane
// parameters, smaller powers of two when a large gamma could push an activation past fp16's range -- the contractions that read the
// block's output (as an image or as fp32) take the scale from these words, so no parameter magnitude makes a forward product overflow.
int act_scales_all(ss_engine* e, hipStream_t s) {
    ActScaleTable tb{};
    ConvBlk* all[7] = {&e->c1[0], &e->c1[1], &e->c1[2], &e->c2[0], &e->c2[1], &e->c2[2], &e->ct};
    for (ConvBlk* cb : all) {
        const int i = cb->scale_i;
        if (i < 0 || i >= ACT_SCALE_MAX) return fail("act_scales_all: bad slot");
        tb.gamma[i] = cb->Co ? e->P + cb->ga : e->P;
        tb.beta[i] = cb->Co ? e->P + cb->be : e->P;
        tb.C[i] = cb->Co;                  // an absent block (Generator_6 has no convolutions_1) has C = 0: bound 0, scale 16
        if (i + 1 > tb.n) tb.n = i + 1;
    }
    HIPCHK(act_scales(tb, e->curT, e->act_scale, s));
    return 0;
}

// ---- whole-model schedules ---------------------------------------------------------------------------------
// Encoder_7 (G3) / Encoder_6 (G6) trunk + their LSTMs, Encoder_t, decoder, head.  Inputs already in in_mel/in_f0/org/emb.
int forward_core(ss_engine* e, bool training, const float* scales, const int* len_seg, int draw0, hipStream_t s) {
    const int B = e->curB, T = e->curT;
    e->part_off = 0;           // split-K scratch: every launch of a step gets its own region; the previous step is through (stream order) when this one's first kernel runs
    const long TP = T + 2 * HALO;
    const int CE = e->CE;
    const bool g3 = e->kind == SS_GENERATOR_3;
    const int S7 = e->plan[0].S;
    const int off2 = g3 ? e->hp.dim_enc : 0;       // first channel of the f0 stream inside the fused slab
    // Branches.  `s` carries the conv trunk; `b2` the per-step weight re-layouts followed by Encoder_t (depends on x_org
    // only); after the trunk the two encoder BLSTMs run side by side (`s`, `b1`).  Everything joins at the decoder input.
    const bool par = e->side && e->side2 && g_overlap;
    hipStream_t b1 = par ? e->side : s, b2 = par ? e->side2 : s;
    e->grads_zeroed = false;               // set again below when this forward belongs to a fused training step
    e->bwd_sync_zeroed = false;
    const bool pack_one = g_pack_one;
    if (pack_one) {
        ConvPackTable pt{};
        pt.img_bf16 = e->img16();
        ConvBlk* all[7] = {&e->c1[0], &e->c2[0], &e->c1[1], &e->c2[1], &e->c1[2], &e->c2[2], &e->ct};
        for (ConvBlk* cb : all) {
            if (!cb->Co) continue;
            const bool img = cb->img_ok();
            pt.t[pt.n++] = {e->P + cb->w, cb->wf, cb->wb, img ? cb->wf_img : nullptr, img ? cb->wb_img : nullptr, cb->Co, cb->Ci, cb->Cp};
        }
        { const int pa_ = prof_begin(e, SS_PROF_PREP, s, 0.0);
        HIPCHK(conv_pack_many(pt, s));
        prof_end(e, pa_, s); }
    } else {
        if (g3) CHK(conv_pack_all(e, e->c1[0], s));
        CHK(conv_pack_all(e, e->c2[0], s));
    }
    CHK(act_scales_all(e, s));             // before every branch forks: the scale words of the conv blocks' outputs
    if (par) CHK(fork_join(e, s, b2));
    if (e->late_org) {
        const ss_hparams& hh = e->hp;
        HIPCHK(copy_rows(e->late_org, hh.dim_freq, (long)T * hh.dim_freq, e->org + HALO * hh.dim_freq, hh.dim_freq, TP * hh.dim_freq, B, T,
                         hh.dim_freq, b2));
        if (e->late_emb) HIPCHK(hipMemcpyAsync(e->emb, e->late_emb, (long)B * hh.dim_spk_emb * 4, hipMemcpyDeviceToDevice, b2));
    }
    // Encoder_7's content (512 ch) and pitch (256 ch) stacks only share the random-resampling PLAN of each layer (model.py:199-206: one
    // warp applied to the concatenation), and the plans depend on the draws alone.  With g_trunk_indep the plans are computed first thing
    // on the branch stream and the two stacks run as independent chains on `s` and `b1` -- conv, GroupNorm, gather of their OWN columns --
    // down to their BLSTMs, instead of meeting before every gather.
    const bool indep = g3 && par && g_conv_par && g_trunk_indep;
    e->xf_img_valid = indep && training && (g_presplit & 4) && e->xf_img[0] && ((e->precision == SS_PRECISION_F32 && g_fwd_f16x2) || (e->img16() && g_gn_gather));     // only fp16 x 2 / 16-bit-path GEMMs read images
    hipEvent_t plans = nullptr;
    if (indep && training) {
        for (int i = 0; i < 3; ++i)
            HIPCHK(interp_plan(e->plan[draw0 + i], scales + (long)(draw0 + i) * B * S7, len_seg + (long)(draw0 + i) * B * S7, nullptr,
                               e->hp.max_len_pad, B, b2));
        plans = e->ev[e->ev_next];
        e->ev_next = (e->ev_next + 1) & 15;
        HIPCHK(hipEventRecord(plans, b2));
    }
    if (indep) CHK(fork_join(e, s, b1));
    for (int i = 1; i < 3 && !pack_one; ++i) {
        if (g3) CHK(conv_pack_all(e, e->c1[i], b2));
        CHK(conv_pack_all(e, e->c2[i], b2));
    }
    if (!pack_one) CHK(conv_pack_all(e, e->ct, b2));
    hipEvent_t packed = nullptr;
    if (par && !pack_one) {
        packed = e->ev[e->ev_next];
        e->ev_next = (e->ev_next + 1) & 15;
        HIPCHK(hipEventRecord(packed, b2));
    }
    // The branch stream's work that nobody needs before the trunk is through: bias sums / W_ih stackings, start state of the
    // persistent recurrences, the gradient-arena memset, the parameter guard, and Encoder_t (model.py:74-89).  With g_prio_order it
    // is ENQUEUED behind the trunk's three layers (it still runs beside them: the host is far ahead of the GPU), so that on a shared
    // hardware queue it can never sit in front of trunk launches.
    const bool prio_fwd = par && g_prio_order;
    auto branch_work = [&]() -> int {
        PrepTable tb;
        tb.n = 0;
        tb.img_bf16 = e->img16();
        if (g3) CHK(lstm_prep(e, e->l1, tb, b2));
        CHK(lstm_prep(e, e->l2, tb, b2));
        CHK(lstm_prep(e, e->lt, tb, b2));
        CHK(lstm_prep(e, e->ld, tb, b2));
        HIPCHK(prep_run(tb, b2));
        if (e->prezero) {        // nothing touches the gradient arena before the decoder backward; b2 is joined long before
            HIPCHK(hipMemsetAsync(e->G, 0, e->arena * 4, b2));
            HIPCHK(hipMemsetAsync(e->amax, 0, 16 * 4, b2));
            e->grads_zeroed = true;
            // the backward recurrences' group words and exchange tiles (nothing in the forward touches them): zeroed here, the backward
            // needs no fork / memset / join between the head's gradient and its first recurrence (two event hops on the critical path)
            if (e->ld.big() && g_persist && lstm_seq_supported(e->curB, e->ld.H) && !(g_exp & 32)) {      // (exp & 32: A/B switch, both schedules are correct)
                HIPCHK(hipMemsetAsync(e->ld.zb, 0, e->ld.zb_bytes, b2));
                if (e->wq_pool) HIPCHK(hipMemsetAsync(e->wq_pool, 0, ss_engine::WQ_SLOTS * 16, b2));
                e->bwd_sync_zeroed = true;
            }
        }
        // fp16 x 2 products scale WEIGHTS by a fixed 16 (forward and gradient contractions alike, and the persistent recurrences' W_hh):
        // fine up to |w| < 4094.  Activations carry their own scale (act_scales_all) and gradients their measured one, so the only thing
        // left to refuse is a parameter beyond 2048 in magnitude, or a non-finite one: the step is then marked invalid (status RANGE)
        // instead of silently overflowing to inf.
        if ((g_fwd_f16x2 || g_bwd_f16x2) && e->precision == SS_PRECISION_F32 && e->sticky) HIPCHK(param_guard(e->P, e->arena, 2048.0f, e->sticky, b2));
        // Encoder_t (model.py:74-89)
        CHK(conv_block_fwd(e, e->ct, Slab{e->org, e->hp.dim_freq}, Slab{e->act_t, e->hp.dim_enc_2, nullptr, e->act_scale + e->ct.scale_i}, b2));
        CHK(lstm_fwd(e, e->lt, Slab{e->act_t, e->hp.dim_enc_2, nullptr, e->act_scale + e->ct.scale_i}, b2));
        return 0;
    };
    for (int i = 0; i < 3; ++i) {
        float* y = training ? e->act : e->xf[i];
        if (i == 1) {
            // Issue order matters: a cross-stream wait on ROCm holds for everything the other stream had been handed when
            // the WAIT was issued, not only up to the recorded event (measured: the trunk stalled ~250 us behind the tiny
            // launches below).  So: first trunk layer, the wait for the re-layouts, and only then the rest of b2's work.
            if (packed) HIPCHK(hipStreamWaitEvent(s, packed, 0));
            if (packed && indep) HIPCHK(hipStreamWaitEvent(b1, packed, 0));
            if (!prio_fwd) CHK(branch_work());
        }
        if (indep) {
            const float* im = (e->xf_img_valid && i > 0) ? e->xf_img[i - 1] : nullptr;
            Slab x1 = i == 0 ? Slab{e->in_mel, e->hp.dim_freq} : Slab{e->xf[i - 1], CE, im, e->act_scale + e->c1[i - 1].scale_i};
            Slab x2 = i == 0 ? Slab{e->in_f0, e->f0p} : Slab{e->xf[i - 1] + off2, CE, im ? e->ioff(im, off2) : nullptr, e->act_scale + e->c2[i - 1].scale_i};
            if (training && g_gn_gather) {       // conv -> [GroupNorm + ReLU + gather] per stack: the normalised slab is never written
                InterpPlan& pl = e->plan[draw0 + i];
                float* gi = (e->xf_img_valid && e->xf_img[i]) ? e->ioff(e->xf_img[i], HALO * CE) : nullptr;
                CHK(conv_block_fwd(e, e->c2[i], x2, Slab{y + off2, CE}, b1, &pl, e->xf[i] + HALO * CE + off2, CE, gi ? e->ioff(gi, off2) : nullptr, i == 0 ? plans : nullptr));
                CHK(conv_block_fwd(e, e->c1[i], x1, Slab{y, CE}, s, &pl, e->xf[i] + HALO * CE, CE, gi, i == 0 ? plans : nullptr));
                continue;
            }
            CHK(conv_block_fwd(e, e->c2[i], x2, Slab{y + off2, CE}, b1));
            CHK(conv_block_fwd(e, e->c1[i], x1, Slab{y, CE}, s));
            if (training) {
                InterpPlan& pl = e->plan[draw0 + i];
                if (i == 0) {
                    HIPCHK(hipStreamWaitEvent(b1, plans, 0));
                    HIPCHK(hipStreamWaitEvent(s, plans, 0));
                }
                float* gi = (e->xf_img_valid && !e->img16() && e->xf_img[i]) ? e->xf_img[i] + HALO * CE : nullptr;      // (the separate gather writes format-v2 images only)
                HIPCHK(interp_gather(pl, e->act + HALO * CE + off2, CE, TP * CE, e->xf[i] + HALO * CE + off2, CE, TP * CE, CE - off2, B, b1,
                                     gi ? gi + off2 : nullptr, e->act_scale + e->c2[i].scale_i));
                HIPCHK(interp_gather(pl, e->act + HALO * CE, CE, TP * CE, e->xf[i] + HALO * CE, CE, TP * CE, off2, B, s, gi, e->act_scale + e->c1[i].scale_i));
            }
            continue;
        }
        // The content (512 ch) and pitch (256 ch) blocks of a layer are independent: with g_conv_par the pitch block and the layer's
        // resampling plan run on the first branch stream beside the content block.
        const bool cpar = g3 && par && g_conv_par;
        hipStream_t sp = cpar ? b1 : s;
        if (cpar) CHK(fork_join(e, s, b1));
        if (g3) {
            Slab x1 = i == 0 ? Slab{e->in_mel, e->hp.dim_freq} : Slab{e->xf[i - 1], CE, nullptr, e->act_scale + e->c1[i - 1].scale_i};
            if (!cpar) CHK(conv_block_fwd(e, e->c1[i], x1, Slab{y, CE}, s));
        }
        Slab x2 = i == 0 ? Slab{e->in_f0, e->f0p} : Slab{e->xf[i - 1] + off2, CE, nullptr, e->act_scale + e->c2[i - 1].scale_i};
        CHK(conv_block_fwd(e, e->c2[i], x2, Slab{y + off2, CE}, sp));
        if (training) {
            // one warp for both streams (model.py:202-206), len_seq = max_len_pad for every utterance (:105,157,203)
            InterpPlan& pl = e->plan[draw0 + i];
            HIPCHK(interp_plan(pl, scales + (long)(draw0 + i) * B * S7, len_seg + (long)(draw0 + i) * B * S7, nullptr,
                               e->hp.max_len_pad, B, sp));
        }
        if (cpar) {
            Slab x1 = i == 0 ? Slab{e->in_mel, e->hp.dim_freq} : Slab{e->xf[i - 1], CE, nullptr, e->act_scale + e->c1[i - 1].scale_i};
            CHK(conv_block_fwd(e, e->c1[i], x1, Slab{y, CE}, s));
            CHK(fork_join(e, b1, s));
        }
        if (training) {
            InterpPlan& pl = e->plan[draw0 + i];
            HIPCHK(interp_gather(pl, e->act + HALO * CE, CE, TP * CE, e->xf[i] + HALO * CE, CE, TP * CE, CE, B, s));
        }
    }
    if (prio_fwd) CHK(branch_work());
    if (par) {
        CHK(fork_join(e, b2, s));                  // bias sums of every block are ready (and Encoder_t is done)
        if (indep) CHK(fork_join(e, b2, b1));      // the pitch chain does not wait for the content chain
        else CHK(fork_join(e, s, b1));
    }
    CHK(lstm_fwd(e, e->l2, Slab{e->xf[2] + off2, CE, nullptr, e->act_scale + e->c2[2].scale_i}, b1));
    if (g3) CHK(lstm_fwd(e, e->l1, Slab{e->xf[2], CE, nullptr, e->act_scale + e->c1[2].scale_i}, s));
    if (par) CHK(fork_join(e, b1, s));
    // decoder input (model.py:301-309 / 341-347)
    CodeSrc src[3];
    int n = 0;
    const ss_hparams& h = e->hp;
    if (g3) {
        src[n++] = {e->l1.out[1], e->d_o1, h.dim_neck, h.freq, 0};
        src[n++] = {e->lt.out[0], e->d_ot, h.dim_neck_2, h.freq_2, 2 * h.dim_neck};
        src[n++] = {e->l2.out[0], e->d_o2, h.dim_neck_3, h.freq_3, 2 * h.dim_neck + 2 * h.dim_neck_2};
        if (dec_compact(e))
            HIPCHK(build_dec_in_compact(src, n, e->emb, h.dim_spk_emb, 2 * h.dim_neck + 2 * h.dim_neck_2 + 2 * h.dim_neck_3, e->ld.xc,
                                        e->dec_in_dim, B, T, e->ld.xf, s));
        else
            HIPCHK(build_dec_in(src, n, e->emb, h.dim_spk_emb, 2 * h.dim_neck + 2 * h.dim_neck_2 + 2 * h.dim_neck_3, e->dec_in,
                                e->dec_in_dim, B, T, s));
    } else {
        src[n++] = {e->lt.out[0], e->d_ot, h.dim_neck_2, h.freq_2, 0};
        src[n++] = {e->l2.out[0], e->d_o2, h.dim_neck_3, h.freq_3, 2 * h.dim_neck_2};
        if (dec_compact(e)) HIPCHK(build_dec_in_compact(src, n, nullptr, 0, e->dec_in_dim, e->ld.xc, e->dec_in_dim, B, T, e->ld.xf, s));
        else HIPCHK(build_dec_in(src, n, nullptr, 0, e->dec_in_dim, e->dec_in, e->dec_in_dim, B, T, s));
    }
    CHK(lstm_fwd(e, e->ld, dec_compact(e) ? Slab{e->ld.xc, e->dec_in_dim} : Slab{e->dec_in, e->dec_in_dim}, s));
    // LinearNorm head (model.py:253 / 277)
    const long HD = 2L * e->ld.H;
    GemmDesc d{};
    d.A = {e->ld.out[e->ld.L - 1] + HALO * HD, HD, TP * HD, 0, 0};
    if (e->ld.out_img_valid)
        d.a_pre = e->ioff(e->ld.out_img[e->ld.L - 1], HALO * HD);
    d.B = {e->P + e->head_w, HD, 0, 0, 0};
    d.C = e->out_slab + HALO * e->head_out;
    d.ldc = e->head_out;
    d.cstride = TP * e->head_out;
    d.bias = e->P + e->head_b;
    d.M = T;
    d.N = e->head_out;
    d.K = (int)HD;
    d.batch = B;
    d.ksplit = 1;
    flatten_rows(d, B, T);
    PGEMM_FWD_ON(SS_PROF_HEAD, d, s);
    e->fwd_training = training;
    e->enc_plan0 = draw0;
    e->have_fwd = true;
    return 0;
}

// gradient of the loss w.r.t. the head output is in d_out_slab (halo rows zero)
// weight and bias gradient of the LinearNorm head from d_out_slab and the last decoder layer's output (both untouched by the backward chain)
int head_weight_grads(ss_engine* e, hipStream_t st) {
    const long TP = e->curT + 2 * HALO, R = (long)e->curB * TP;
    const long HD = 2L * e->ld.H;
    const float* h3 = e->ld.out[e->ld.L - 1];
    GemmDesc a{};
    a.A = {e->d_out_slab, e->head_out, 0, 0, 0};
    a.B = {h3, HD, 0, 0, 0};
    a.C = e->G + e->head_w;
    a.ldc = HD;
    a.M = e->head_out;
    a.N = (int)HD;
    a.K = (int)R;
    a.batch = 1;
    a.flags = GEMM_TA | GEMM_TB | GEMM_ACCUM;
    a.ksplit = pick_ksplit(a.M, a.N, a.K);
    PGEMM_ON(SS_PROF_HEAD, a, st);
    {
        double* cpart;
        unsigned* cctr;
        colsum_scratch(e, e->head_out, &cpart, &cctr);
        HIPCHK(colsum_acc(e->d_out_slab, e->head_out, (int)R, e->head_out, e->G + e->head_b, cpart, cctr, st));
    }
    CHK(dp_bucket(e, e->head_w, e->status_off - e->head_w, st));       // (the status slot behind it rides the step's last bucket)
    return 0;
}

// late: only the critical chain is enqueued here (head input gradient, recurrences, input gradients); the decoder's and the head's
// weight gradients are enqueued by backward_encoder behind ITS critical path, ordered after the chain through ev_join[1]
int backward_decoder(ss_engine* e, hipStream_t s, bool late = false) {
    if (!e->have_fwd) return fail("backward without a preceding forward");
    const int B = e->curB, T = e->curT;
    const long TP = T + 2 * HALO, R = (long)B * TP;
    if (!e->grads_zeroed) {
        HIPCHK(hipMemsetAsync(e->G, 0, e->arena * 4, s));
        HIPCHK(hipMemsetAsync(e->amax, 0, 16 * 4, s));
    }
    e->grads_zeroed = false;
    // fragment-major W_hh^T of the decoder recurrences (overwrites the forward layout, no longer needed), beside the head
    const bool persist_dec = e->ld.big() && g_persist && lstm_seq_supported(e->curB, e->ld.H);
    const bool prezeroed = e->bwd_sync_zeroed && persist_dec;       // the fused step's forward has done it on its branch stream
    e->bwd_sync_zeroed = false;
    const bool par = e->side2 && g_overlap && !prezeroed;
    hipStream_t b2 = par ? e->side2 : s;
    if (par) CHK(fork_join(e, s, b2));
    e->dec_ih_done = 0;
    e->dec_w_done = 0;
    e->dg16_written = 0;
    e->dg32_skipped = 0;
    e->wq_next = 0;
    if (e->ld.big() && !prezeroed) {
        if (g_persist && lstm_seq_supported(e->curB, e->ld.H)) {
            // the group counters of every layer and their exchange tiles (tags start at 0)
            HIPCHK(hipMemsetAsync(e->ld.zb, 0, e->ld.zb_bytes, b2));
            if (e->wq_pool) HIPCHK(hipMemsetAsync(e->wq_pool, 0, ss_engine::WQ_SLOTS * 16, b2));
        } else {
            for (int l = 0; l < e->ld.L; ++l)
                HIPCHK(lstm_pack_w(e->P + e->ld.pd[l * 2].whh, e->P + e->ld.pd[l * 2 + 1].whh, e->ld.wfrag[l], e->ld.H, 1, b2));
        }
    }
    // head.  Only its input gradient is on the critical path; when the decoder's weight gradients are deferred to the side
    // stream (lstm_bwd), the head's weight / bias gradients go with them instead of running in front of the first recurrence.
    const long HD = 2L * e->ld.H;
    const bool defer_head = e->ld.big() && g_persist && lstm_seq_supported(B, e->ld.H) && e->side && g_overlap && g_defer_dw;
    late = late && defer_head;
    e->dec_w_pending = late;
    {
        if (!defer_head) CHK(head_weight_grads(e, s));
        GemmDesc g{};
        g.A = {e->d_out_slab, e->head_out, 0, 0, 0};
        g.B = {e->P + e->head_w, HD, 0, 0, 0};
        g.C = e->d_top;
        g.ldc = HD;
        g.M = (int)R;
        g.N = (int)HD;
        g.K = e->head_out;
        g.batch = 1;
        g.flags = GEMM_TB;
        g.ksplit = 1;
        PGEMM_ON(SS_PROF_HEAD, g, s);
    }
    if (par) CHK(fork_join(e, b2, s));
    if (dec_compact(e)) CHK(lstm_bwd(e, e->ld, e->d_top, Slab{e->ld.xc, e->dec_in_dim}, Slab{e->ld.d_xc, e->dec_in_dim}, s, nullptr, late));
    else CHK(lstm_bwd(e, e->ld, e->d_top, Slab{e->dec_in, e->dec_in_dim}, Slab{e->d_dec_in, e->dec_in_dim}, s, nullptr, late));
    if (late) HIPCHK(hipEventRecord(e->ev_join[1], s));          // the chain is through: what the side stream's batch waits for
    else if (defer_head) CHK(head_weight_grads(e, e->side));      // behind the decoder's weight gradients, ordered after the chain by lstm_bwd's fork
                                                            // (measured: on the third branch stream instead 6.395 vs 6.365 ms)
    // every persistent recurrence of the step is behind this point: publish this rank's status into the gradient arena's
    // status slot (part of the decoder bucket, so a data-parallel all-reduce carries it to every rank's Adam kernel)
    if (e->sticky) HIPCHK(status_publish(e->sticky, e->G + e->status_off, s));
    return 0;
}

// everything below the decoder input: code gradients, encoder BLSTMs, conv trunks.  Touches only gradient-arena
// offsets below the decoder's, so a data-parallel caller can all-reduce the decoder range meanwhile.
int backward_encoder(ss_engine* e, hipStream_t s) {
    e->unpack.n = 0;
    e->unpack_later = !e->dp_on && g_unpack_later;       // (data parallel: a trunk layer's bucket leaves right behind its block)
    struct UnpackOff {
        ss_engine* e;
        ~UnpackOff() { e->unpack_later = false; }
    } unpack_off{e};
    const int B = e->curB, T = e->curT;
    const long TP = T + 2 * HALO, R = (long)B * TP;
    const int CE = e->CE;
    const bool g3 = e->kind == SS_GENERATOR_3;
    const bool training = e->fwd_training;
    const ss_hparams& h = e->hp;
    CodeSrc src[3];
    int n = 0;
    if (g3) {
        src[n++] = {e->l1.out[1], e->d_o1, h.dim_neck, h.freq, 0};
        src[n++] = {e->lt.out[0], e->d_ot, h.dim_neck_2, h.freq_2, 2 * h.dim_neck};
        src[n++] = {e->l2.out[0], e->d_o2, h.dim_neck_3, h.freq_3, 2 * h.dim_neck + 2 * h.dim_neck_2};
    } else {
        src[n++] = {e->lt.out[0], e->d_ot, h.dim_neck_2, h.freq_2, 0};
        src[n++] = {e->l2.out[0], e->d_o2, h.dim_neck_3, h.freq_3, 2 * h.dim_neck_2};
    }
    if (dec_compact(e)) HIPCHK(dec_in_grad_compact(src, n, e->ld.d_xc, e->dec_in_dim, B, T, e->ld.xf, s));
    else HIPCHK(dec_in_grad(src, n, e->d_dec_in, e->dec_in_dim, B, T, s));
    const int off2 = g3 ? h.dim_enc : 0;
    // Three independent branches below the decoder input: lstm_1 (on `s`), lstm_2 (`b2`; writes the other columns of d_xf)
    // and Encoder_t (`b3`; joins at the end).  Their weight-gradient GEMMs share the side stream.
    const bool par = e->side2 && e->side3 && g_overlap;
    hipStream_t b2 = par ? e->side2 : s, b3 = par ? e->side3 : s;
    // prio: critical path first (g_prio_order) -- lstm_2's and lstm_1's chains and the conv trunk are enqueued before anything that only
    // has to be done by the end of the step; the latter (decoder + head weight gradients, the encoder BLSTMs' weight gradients,
    // Encoder_t's whole backward) follow below, each ordered behind its producer by an event that was recorded when the producer
    // was enqueued.
    const bool prio = par && g_prio_order && !e->l2.big() && !e->l1.big() && !e->lt.big();
    struct WgDefer {           // the small blocks' weight gradients of this backward: collected, launched once on b3 below (prio schedule)
        ss_engine* e;
        ~WgDefer() { e->wg_defer = false; e->wg.n = 0; e->wg.tiles_total = 0; }
    } wg_scope{e};
    e->wg.n = 0;
    e->wg.tiles_total = 0;
    e->wg_defer = prio;
    if (par) {
        if (prio) HIPCHK(hipEventRecord(e->ev_join[2], s));                 // dec_in_grad done: what Encoder_t's backward needs
        CHK(fork_join(e, s, b2));
        if (!prio) CHK(fork_join(e, s, b3));
    }
    CHK(zero_conv_grads(e, b2));                   // long done when the first conv weight gradient starts (b2 joins s, b3 forks after)
    if (prio && g_enc_t_first && b3 != b2) CHK(fork_join(e, b2, b3));      // Encoder_t's conv weight gradient accumulates into its zeroed image before b3 sees lstm_2's event
    if (par && !prio) CHK(fork_join(e, b2, b3));
    // encoder BLSTMs -> gradient of the last fused slab
    const bool early = par && !e->l2.big() && g_early_join;         // the join event of the lstm_2 branch is taken as soon as its last kernel is queued
    CHK(lstm_bwd(e, e->l2, e->d_o2, Slab{e->xf[2] + off2, CE, nullptr, e->act_scale + e->c2[2].scale_i}, Slab{e->d_xf + off2, CE}, b2, (early || prio) ? e->ev_join[0] : nullptr, prio));
    if (g3) {
        CHK(lstm_bwd(e, e->l1, e->d_o1, Slab{e->xf[2], CE, nullptr, e->act_scale + e->c1[2].scale_i}, Slab{e->d_xf, CE}, s, nullptr, prio));
        if (prio) HIPCHK(hipEventRecord(e->ev_join[3], s));                 // lstm_1's chain done: its weight gradients may start
    }
    if (!prio) {
        // Encoder_t
        CHK(lstm_bwd(e, e->lt, e->d_ot, Slab{e->act_t, h.dim_enc_2, nullptr, e->act_scale + e->ct.scale_i}, Slab{e->d_act_t, h.dim_enc_2}, b3));
        CHK(conv_block_bwd(e, e->ct, Slab{e->d_act_t, h.dim_enc_2}, Slab{e->org, h.dim_freq}, Slab{nullptr, 0}, b3));
    }
    if (early || prio) HIPCHK(hipStreamWaitEvent(s, e->ev_join[0], 0));
    else if (par) CHK(fork_join(e, b2, s));
    // backward_decoder(late): the decoder's (layers l_hi .. l_lo) and, with its layer 0, the head's weight gradients, behind the decoder chain
    int dec_next = e->dec_w_pending ? e->ld.L - 1 : -1;       // next decoder layer whose weight gradients are still to be enqueued
    hipStream_t dec_other[2] = {nullptr, nullptr};      // streams other than the side stream that carry decoder weight gradients (tail split)
    auto dec_late = [&](int l_lo, hipStream_t ws = nullptr, hipStream_t over_ih = nullptr, hipStream_t over_hh = nullptr) -> int {
        if (dec_next < l_lo) return 0;
        if (!ws) ws = e->side;
        for (hipStream_t o : {over_ih, over_hh})
            if (o && o != ws && o != e->side) {
                if (dec_other[0] != o && dec_other[1] != o) dec_other[dec_other[0] ? 1 : 0] = o;
                HIPCHK(hipStreamWaitEvent(o, e->ev_join[1], 0));
            }
        struct Over {
            ss_engine* e;
            ~Over() { e->dw_over[0] = e->dw_over[1] = nullptr; }
        } over_scope{e};
        e->dw_over[0] = over_ih;
        e->dw_over[1] = over_hh;
        if (dec_next == e->ld.L - 1) {
            if (g_exp & 1) CHK(fork_join(e, s, e->side));      // experiment: ... and behind the conv trunk's backward as well (the two chains run one after the other)
            HIPCHK(hipStreamWaitEvent(e->side, e->ev_join[1], 0));
        }
        if (ws != e->side) {
            if (dec_other[0] != ws && dec_other[1] != ws) dec_other[dec_other[0] ? 1 : 0] = ws;
            HIPCHK(hipStreamWaitEvent(ws, e->ev_join[1], 0));
        }
        CHK(lstm_late_weights(e, e->ld, dec_compact(e) ? Slab{e->ld.xc, e->dec_in_dim} : Slab{e->dec_in, e->dec_in_dim}, ws, dec_next, l_lo));
        dec_next = l_lo - 1;
        if (l_lo == 0) {
            CHK(head_weight_grads(e, ws));
            e->dec_w_pending = false;
            for (hipStream_t o : dec_other)
                if (o) CHK(fork_join(e, o, e->side));      // one join for the end of the step; the early optimiser update below reads what they wrote
            if (e->adam_early && g_adam_early && !e->dp_on && e->Mm && e->Vv) {
                // every persistent recurrence and the parameter guard ran before the event this stream waited for: the status word is final
                const long from = ss_grad_split(e);
                if (from % 4 == 0 && from < e->arena) {
                    HIPCHK(adam_prepare(e->adam, e->sticky, nullptr, e->side));
                    { const int pa_ = prof_begin(e, SS_PROF_ADAM, e->side, 0.0);
                    HIPCHK(adam_range(e->P + from, e->G + from, e->Mm + from, e->Vv + from, e->arena - from, e->adam, e->adam_early_gs, e->side));
                    prof_end(e, pa_, e->side); }
                    e->adam_early_from = from;
                }
            }
        }
        e->side_used = true;
        return 0;
    };
    // Data parallel: the communication stream executes its collectives in the order they are ENQUEUED, so the buckets are enqueued in the
    // order they become final -- decoder layer L-1, trunk layer 2, decoder layer L-2, trunk layer 1, the rest of the decoder + head --
    // instead of every decoder bucket in front of (or behind) every trunk bucket.  (One GPU: the decoder's weight gradients stay behind
    // the trunk in enqueue order, see prio_order.)
    if (e->dp_on) CHK(dec_late(e->ld.L - 1));
    // Encoder_7's content (512 ch) and pitch (256 ch) stacks are independent chains in the backward as in the forward (forward_core, trunk_indep):
    // each block's output gradient comes from its own BLSTM / its own upper block, its input gradient goes to its own lower block, and the two
    // write disjoint columns of the shared slabs.  With g_trunk_bwd_par the pitch chain never leaves the stream lstm_2's backward ran on.
    // Data parallel: that stream carries the collectives, so the pitch chain takes the third branch stream instead (behind the event of lstm_2's
    // input gradient; Encoder_t's backward and the fused weight gradients queue behind it there).
    const bool chain_par = g_trunk_bwd_par && training && g3 && par && (g_exp & 2) == 0 && g_gn_gather && (!e->dp_on || (prio && b3 != s && b3 != b2));
    hipStream_t cs = e->dp_on ? b3 : b2;
    if (chain_par && e->dp_on) HIPCHK(hipStreamWaitEvent(b3, e->ev_join[0], 0));      // d_xf's pitch columns (lstm_2's input gradient) and the zeroed conv images
    // weight gradients off the dependent chain (g_conv_dw_off): Generator_6's single chain, the second branch stream is idle behind lstm's backward
    const bool dw_off = g_conv_dw_off == 1 && training && !g3 && par && prio && !e->dp_on && g_gn_gather && e->unpack_later && b2 != s && e->d_act_l[0] && e->d_act_l[1];
    hipStream_t dw_s = dw_off ? b2 : nullptr;
    // conv trunk, last layer first
    for (int i = 2; i >= 0; --i) {
        float* dy = e->d_xf;
        // training: the adjoint of the layer's resampling, d_xf -> d_act; fused into each block's GroupNorm backward (g_gn_gather) or as a pass of its own
        const InterpPlan* sc = (training && g_gn_gather) ? &e->plan[e->enc_plan0 + i] : nullptr;
        const float* sc_src = e->d_xf + HALO * CE;
        if (training) {
            if (!sc) HIPCHK(interp_scatter(e->plan[e->enc_plan0 + i], e->d_xf + HALO * CE, CE, TP * CE, e->d_act + HALO * CE, CE, TP * CE, CE, B, s));
            dy = (dw_off && i > 0) ? e->d_act_l[i - 1] : e->d_act;
        }
        // input gradients of layer i become d_xf (the gradient of xf[i-1]); dy is consumed before it is overwritten
        // only when dy != d_xf, so in eval mode the input gradient goes through d_act instead.
        float* dxbuf = training ? e->d_xf : e->d_act;
        // the resampled activations also exist as pre-split images when the forward's gathers wrote them (training, independent trunk chains)
        const float* bim = (training && e->xf_img_valid && i > 0) ? e->xf_img[i - 1] : nullptr;
        if (!chain_par && i == 0 && g3 && par && !e->dp_on && (g_exp & 2) == 0) CHK(fork_join(e, s, b2));      // tail_par below: the pitch block's stream forks BEFORE the content block is enqueued
        if (g3) {
            Slab x1 = i == 0 ? Slab{e->in_mel, h.dim_freq} : Slab{e->xf[i - 1], CE, bim, e->act_scale + e->c1[i - 1].scale_i};
            CHK(conv_block_bwd(e, e->c1[i], Slab{dy, CE}, x1, i > 0 ? Slab{dxbuf, CE} : Slab{nullptr, 0}, s, sc, sc_src, CE));
        }
        Slab x2 = i == 0 ? Slab{e->in_f0, e->f0p} : Slab{e->xf[i - 1] + off2, CE, bim ? e->ioff(bim, off2) : nullptr, e->act_scale + e->c2[i - 1].scale_i};
        // Layer 0 is the step's tail: the decoder's weight gradients are through by then, and each of its two weight-gradient GEMMs alone
        // fills half the chip's workgroup slots -- the pitch block runs on the second branch stream beside the content block.  (Not under
        // data parallelism, where that stream carries the collectives.)
        const bool tail_par = chain_par || (i == 0 && g3 && par && !e->dp_on && (g_exp & 2) == 0);
        hipStream_t s2 = tail_par ? (chain_par ? cs : b2) : s;
        CHK(conv_block_bwd(e, e->c2[i], Slab{dy + off2, CE}, x2, i > 0 ? Slab{dxbuf + off2, CE} : Slab{nullptr, 0}, s2, sc, sc_src + off2, CE, (dw_off && i > 0) ? dw_s : nullptr));
        if (tail_par && (!chain_par || (i == 0 && !e->dp_on))) CHK(fork_join(e, b2, s));      // (chain_par: the two chains meet once, behind layer 0; data parallel: where the third branch stream joins below)
        if (i > 0) {           // the two wide layers' parameters (weight, bias, GroupNorm affine: contiguous) are final; layer 0 rides the last bucket
            if (g3) CHK(dp_bucket(e, e->c1[i].w, e->c1[i].be + e->c1[i].Co - e->c1[i].w, s));
            CHK(dp_bucket(e, e->c2[i].w, e->c2[i].be + e->c2[i].Co - e->c2[i].w, s2));
            if (e->dp_on) CHK(dec_late(i == 2 ? (e->ld.L >= 3 ? e->ld.L - 2 : 0) : 0));      // next decoder layer(s) behind this trunk layer's buckets
        }
        if (!training && i > 0) {
            // eval mode has no resampling between layers: the next (lower) layer reads its output gradient from d_xf
            HIPCHK(hipMemcpyAsync(e->d_xf, e->d_act, R * CE * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    // ---- everything that only has to be finished by the end of the step
    // (tail split: on the side stream alone the decoder's twelve weight-gradient GEMMs end ~450 us after every other stream -- its last layer and
    // the head go behind a stream that ends early instead)
    if (!e->dp_on && prio && chain_par && e->dec_w_pending && e->ld.L == 3 && g_dec_tail_split) {
        const int m = g_dec_tail_split;
        //                              m:        1        2        3    4        5        6
        hipStream_t for_l1 = m == 4 || m == 5 ? b3 : (m == 6 ? b2 : nullptr);                    // layer 1 (nullptr: side stream)
        hipStream_t for_l0 = m == 1 || m == 5 ? b2 : (m == 2 || m == 6 || m >= 7 ? b3 : (m == 3 ? s : nullptr));   // layer 0 + head
        // 7-9: as 2, and layer 1's reverse direction leaves the side stream as well: 7: dW_ih -> third, dW_hh -> pitch stream; 8: both -> third; 9: dW_hh -> third
        hipStream_t l1_ih = m == 7 || m == 8 ? b3 : nullptr, l1_hh = m == 7 ? b2 : (m == 8 || m == 9 ? b3 : nullptr);
        CHK(dec_late(2, nullptr, nullptr, m == 10 ? b3 : nullptr));      // 10: as 9, and layer 2's reverse dW_hh -> third as well
        if (m == 10) l1_hh = b3;
        CHK(dec_late(1, for_l1, l1_ih, l1_hh));
        CHK(dec_late(0, for_l0));
        if (for_l1 == b2 || for_l0 == b2 || l1_hh == b2) CHK(fork_join(e, b2, s));
    }
    CHK(dec_late(0));
    if (prio) {
        // Encoder_t's backward first: its input (d_ot from dec_in_grad) is the earliest thing this stream waits for, and it is a dependent chain of
        // five launches; the BLSTMs' weight gradients (all three blocks: ONE fused launch) only have to be done by the end of the step
        HIPCHK(hipStreamWaitEvent(b3, e->ev_join[2], 0));                   // d_ot from dec_in_grad
        if (!g_enc_t_first) {
            HIPCHK(hipStreamWaitEvent(b3, e->ev_join[0], 0));
            CHK(lstm_late_weights(e, e->l2, Slab{e->xf[2] + off2, CE, nullptr, e->act_scale + e->c2[2].scale_i}, b3));
            if (g3) {
                HIPCHK(hipStreamWaitEvent(b3, e->ev_join[3], 0));
                CHK(lstm_late_weights(e, e->l1, Slab{e->xf[2], CE, nullptr, e->act_scale + e->c1[2].scale_i}, b3));
            }
            CHK(wgrad_flush(e, b3));
        }
        CHK(lstm_bwd(e, e->lt, e->d_ot, Slab{e->act_t, h.dim_enc_2, nullptr, e->act_scale + e->ct.scale_i}, Slab{e->d_act_t, h.dim_enc_2}, b3));
        CHK(conv_block_bwd(e, e->ct, Slab{e->d_act_t, h.dim_enc_2}, Slab{e->org, h.dim_freq}, Slab{nullptr, 0}, b3));
        if (g_enc_t_first) {
            HIPCHK(hipStreamWaitEvent(b3, e->ev_join[0], 0));                   // lstm_2's pre-activation gradients (and the zeroed conv images)
            CHK(lstm_late_weights(e, e->l2, Slab{e->xf[2] + off2, CE, nullptr, e->act_scale + e->c2[2].scale_i}, b3));
            if (g3) {
                HIPCHK(hipStreamWaitEvent(b3, e->ev_join[3], 0));
                CHK(lstm_late_weights(e, e->l1, Slab{e->xf[2], CE, nullptr, e->act_scale + e->c1[2].scale_i}, b3));
            }
        }
        CHK(wgrad_flush(e, b3));                   // Encoder_t's, lstm_2's and both layers of lstm_1's: one launch
        e->wg_defer = false;
    }

    if (par) CHK(fork_join(e, b3, s));
    if (dw_off) CHK(fork_join(e, dw_s, s));
    CHK(join_side(e, s));
    if (e->unpack.n) {
        HIPCHK(conv_unpack_grads(e->unpack, s));
        e->unpack.n = 0;
    }
    return 0;
}

int backward_core(ss_engine* e, hipStream_t s) {
    CHK(backward_decoder(e, s, g_prio_order));
    return backward_encoder(e, s);
}

}  // namespace

// ================================================================================================ C ABI
namespace {

int stage_g3_inputs(ss_engine* e, const float* x_f0, const float* x_org, const float* c_trg, int B, int T, hipStream_t s) {
    const ss_hparams& h = e->hp;
    const long TP = T + 2 * HALO;
    const int CI = h.dim_freq + h.dim_f0;      // 337
    // split x_f0 [B,T,337] into the mel slab and the (channel-padded) f0 slab (model.py:196-197)
    HIPCHK(copy_rows(x_f0, CI, (long)T * CI, e->in_mel + HALO * h.dim_freq, h.dim_freq, TP * h.dim_freq, B, T, h.dim_freq, s));
    HIPCHK(copy_rows(x_f0 + h.dim_freq, CI, (long)T * CI, e->in_f0 + HALO * e->f0p, e->f0p, TP * e->f0p, B, T, h.dim_f0, s));
    HIPCHK(copy_rows(x_org, h.dim_freq, (long)T * h.dim_freq, e->org + HALO * h.dim_freq, h.dim_freq, TP * h.dim_freq, B, T,
                     h.dim_freq, s));
    HIPCHK(hipMemcpyAsync(e->emb, c_trg, (long)B * h.dim_spk_emb * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

int export_out(ss_engine* e, float* out, int B, int T, hipStream_t s) {
    const long TP = T + 2 * HALO;
    const int C = e->head_out;
    HIPCHK(copy_rows(e->out_slab + HALO * C, C, TP * C, out, C, (long)T * C, B, T, C, s));
    return 0;
}

int import_dout(ss_engine* e, const float* d_out, int B, int T, hipStream_t s) {
    const long TP = T + 2 * HALO;
    const int C = e->head_out;
    HIPCHK(copy_rows(d_out, C, (long)T * C, e->d_out_slab + HALO * C, C, TP * C, B, T, C, s));
    return 0;
}

}  // namespace

extern "C" {

int ss_comm_destroy(ss_engine* e);
const char* ss_last_error(void) { return g_err.c_str(); }
int ss_abi_version(void) { return 2; }      // 2: ss_profile(mask) / ss_profile_read, status word, RCCL entry points, SS_STEP_BUCKET

ss_engine* ss_create(int kind, const ss_hparams* hp, int max_batch, int max_frames) {
    if (!hp || (kind != SS_GENERATOR_3 && kind != SS_GENERATOR_6 && kind != SS_INTERP_ONLY)) {
        fail("ss_create: kind must be SS_GENERATOR_3, SS_GENERATOR_6 or SS_INTERP_ONLY");
        return nullptr;
    }
    if (hp->chs_grp != 16 || hp->dim_enc % 64 || hp->dim_enc_2 % 64 || hp->dim_enc_3 % 64) {
        fail("ss_create: this build needs chs_grp == 16 and conv widths that are multiples of 64");
        return nullptr;
    }
    if (2 * hp->max_len_seg > 64 || hp->max_len_pad > 512 || max_frames > 256 || max_frames < 8 || max_batch < 1) {
        fail("ss_create: limits are 2*max_len_seg <= 64, max_len_pad <= 512, 8 <= max_frames <= 256");
        return nullptr;
    }
    for (int hdim : {hp->dim_neck, hp->dim_neck_2, hp->dim_neck_3}) {
        if (hdim != 1 && hdim != 2 && hdim != 4 && hdim != 8 && hdim != 16 && hdim != 32) {
            fail("ss_create: bottleneck widths must be one of 1,2,4,8,16,32");
            return nullptr;
        }
    }
    ss_engine* e = new ss_engine();
    e->kind = kind;
    e->hp = *hp;
    e->bound_max_len_pad = hp->max_len_pad;
    e->maxB = max_batch;
    e->maxT = max_frames;
    build_table(e);
    return e;
}

void ss_destroy(ss_engine* e) {
    if (!e) return;
    if (e->comm) (void)ss_comm_destroy(e);
    if (e->sticky) {
        (void)hipDeviceSynchronize();
        (void)hipHostFree(e->sticky);
    }
    for (auto& ev : e->prof_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->side) {
        (void)hipStreamSynchronize(e->side);
        for (auto& ev : e->ev)
            if (ev) (void)hipEventDestroy(ev);
        (void)hipStreamDestroy(e->side);
        if (e->ev_comm) (void)hipEventDestroy(e->ev_comm);
        for (hipEvent_t ev : e->dp_ev_pool) (void)hipEventDestroy(ev);
        if (e->dp_bwd_end) (void)hipEventDestroy(e->dp_bwd_end);
        for (hipStream_t st : {e->side2, e->side3})
            if (st) {
                (void)hipStreamSynchronize(st);
                (void)hipStreamDestroy(st);
            }
        if (e->main_s) {
            (void)hipStreamSynchronize(e->main_s);
            (void)hipStreamDestroy(e->main_s);
        }
        {
            for (auto& ev : e->ev_io)
                if (ev) (void)hipEventDestroy(ev);
            for (auto& ev : e->ev_dec)
                if (ev) (void)hipEventDestroy(ev);
            for (auto& ev : e->ev_join)
                if (ev) (void)hipEventDestroy(ev);
        }
    }
    delete e;
}

int ss_num_params(const ss_engine* e) { return (int)e->params.size(); }

int ss_param_info(const ss_engine* e, int i, char* name, int cap, long* offset, int* ndim, long shape[3]) {
    if (i < 0 || i >= (int)e->params.size()) return fail("ss_param_info: index out of range");
    const ParamInfo& p = e->params[i];
    if (name && cap > 0) {
        std::strncpy(name, p.name.c_str(), cap - 1);
        name[cap - 1] = 0;
    }
    if (offset) *offset = p.offset;
    if (ndim) *ndim = p.ndim;
    if (shape)
        for (int k = 0; k < 3; ++k) shape[k] = p.shape[k];
    return 0;
}

long ss_arena_numel(const ss_engine* e) { return e->arena; }

// split-K scratch of the image GEMM behind the planned workspace (never zeroed, independent of the geometry): partial slabs have the
// size of weight tensors, so 16 arenas' worth holds a step's launches at ksplit <= 8 with room to spare
static long part_floats(const ss_engine* e) { return e->kind == SS_INTERP_ONLY ? 0 : 16 * ((e->arena + 63) & ~63L) + (32L << 20); }      // + 32 M floats: the small generator's deterministic mode, column sums, fused encoder weight gradients
static long plan_bytes(const ss_engine* e) {
    ss_engine tmp = *e;             // dry run on a copy: carve() assigns the slab pointers
    return (tmp.carve(e->maxB, e->maxT, false) + 255) & ~255L;
}
long ss_workspace_bytes(const ss_engine* e) { return plan_bytes(e) + part_floats(e) * 4; }

int ss_bind(ss_engine* e, float* params, float* grads, float* m, float* v, void* workspace, long ws_bytes, void* stream) {
    if (!workspace || (e->kind != SS_INTERP_ONLY && (!params || !grads))) return fail("ss_bind: null arena");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v | (uintptr_t)workspace) & 255)
        return fail("ss_bind: arenas must be 256-byte aligned");
    const long need = ss_workspace_bytes(e);
    if (ws_bytes < need) return fail("ss_bind: workspace smaller than ss_workspace_bytes()");
    e->P = params;
    e->G = grads;
    e->Mm = m;
    e->Vv = v;
    e->ws = (char*)workspace;
    e->ws_bytes = plan_bytes(e);
    e->part = (float*)(e->ws + e->ws_bytes);
    e->part_cap = part_floats(e);
    e->part_off = 0;
    e->curB = e->curT = 0;
    e->have_fwd = false;
    HIPCHK(hipMemsetAsync(e->ws, 0, e->ws_bytes, S(stream)));
    e->carve(e->maxB, e->maxT, true);
    if (!e->sticky) {       // host-coherent pinned word: kernels OR into it (system scope), the host reads it without a sync
        void* p = nullptr;
        HIPCHK(hipHostMalloc(&p, 64, hipHostMallocDefault));
        std::memset(p, 0, 64);
        e->sticky = (unsigned*)p;
    }
    if (!e->side && e->kind != SS_INTERP_ONLY) {
        // main + three branch streams, created back to back: HIP deals a process's streams onto its hardware queues in creation
        // order, so these four get one queue each no matter how many streams (PyTorch's, RCCL's) existed before
        if (g_own_streams) HIPCHK(hipStreamCreateWithFlags(&e->main_s, hipStreamNonBlocking));
        if (g_probe_queues && !g_side_prio) {
            // branch streams chosen by MEASURING which candidates share a hardware queue with the stream the step will be issued on
            CHK(pick_streams(e, (g_own_streams && e->main_s) ? e->main_s : S(stream)));
        } else {
            int least = 0, greatest = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, g_side_prio ? least : 0));
            HIPCHK(hipStreamCreateWithFlags(&e->side2, hipStreamNonBlocking));
            HIPCHK(hipStreamCreateWithFlags(&e->side3, hipStreamNonBlocking));
            e->stream_report = "branch streams as created (probe off)";
        }
        for (auto& ev : e->ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (auto& ev : e->ev_io) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (auto& ev : e->ev_dec) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (auto& ev : e->ev_join) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    AdamState st{};
    st.lr = 1e-4;
    st.beta1 = 0.9;
    st.beta2 = 0.999;
    st.eps = 1e-8;
    st.step = 0;
    HIPCHK(hipMemcpyAsync(e->adam, &st, sizeof(st), hipMemcpyHostToDevice, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    return 0;
}

int ss_set_adam(ss_engine* e, double lr, double b1, double b2, double eps, long step, void* stream) {
    if (!e->ws) return fail("engine is not bound");
    AdamState st{};
    st.lr = lr;
    st.beta1 = b1;
    st.beta2 = b2;
    st.eps = eps;
    st.step = step;
    HIPCHK(hipMemcpyAsync(e->adam, &st, sizeof(st), hipMemcpyHostToDevice, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    return 0;
}

// enqueue only: the host-side status check belongs to the ABI entry points, never to the middle of a step being enqueued
// (the device-side guard in adam_prepare_kernel is what protects the parameters)
static int adam_enqueue(ss_engine* e, float grad_scale, hipStream_t s) {
    if (!e->ws || !e->Mm || !e->Vv) return fail("ss_adam_step: Adam arenas are not bound");
    if (e->adam_early_from >= 0) {         // the decoder + head range went out beside the encoder backward (backward_encoder): the rest, same step state
        const long n = e->adam_early_from;
        e->adam_early_from = -1;
        { const int pa_ = prof_begin(e, SS_PROF_ADAM, s, 0.0);
        HIPCHK(adam_range(e->P, e->G, e->Mm, e->Vv, n, e->adam, grad_scale, s));
        prof_end(e, pa_, s); }
        return 0;
    }
    { const int pa_ = prof_begin(e, SS_PROF_ADAM, s, 0.0);
    HIPCHK(adam_step(e->P, e->G, e->Mm, e->Vv, e->arena, e->adam, grad_scale, e->sticky, e->G + e->status_off, s));
    prof_end(e, pa_, s); }
    return 0;
}
// a fused training step announces that Adam follows its backward inside the same call
struct AdamEarly {
    ss_engine* e;
    AdamEarly(ss_engine* e_, bool on, float gs) : e(e_) {
        e->adam_early = on;
        e->adam_early_gs = gs;
        e->adam_early_from = -1;
    }
    // A step that returns with an error behind the early range (decoder + head already updated on the side stream) must not leave the field
    // set: a later ss_adam_step would then update [0, from) only, with that step's prepared state.  The failed step's partial update stands
    // (its caller has an error in hand); the next optimiser call starts clean.
    ~AdamEarly() {
        e->adam_early = false;
        e->adam_early_from = -1;
    }
};

int ss_adam_step(ss_engine* e, float grad_scale, void* stream) {
    CHK(entry_check(e));
    e->adam_early_from = -1;               // a stand-alone optimiser step always covers the whole arena
    Own own(e, stream);
    return adam_enqueue(e, grad_scale, own.s);
}

int ss_zero_grads(ss_engine* e, void* stream) {
    if (!e->G) return fail("engine is not bound");
    Own own(e, stream);
    HIPCHK(hipMemsetAsync(e->G, 0, e->arena * 4, own.s));
    return 0;
}

int ss_g3_forward(ss_engine* e, const float* x_f0, const float* x_org, const float* c_trg, const float* scales,
                  const int* len_seg, int B, int T, int training, float* out, void* stream) {
    if (e->kind != SS_GENERATOR_3) return fail("ss_g3_forward on a Generator_6 engine");
    CHK(entry_check(e));
    Own own(e, stream);
    hipStream_t s = own.s;
    if (training && T != e->hp.max_len_pad)
        return fail("train-mode forward needs T == max_len_pad (InterpLnr pads to max_len_pad, model.py:370)");
    if (training && (!scales || !len_seg)) return fail("train-mode forward needs the InterpLnr draws");
    CHK(geometry(e, B, T, s));
    CHK(stage_g3_inputs(e, x_f0, x_org, c_trg, B, T, s));
    CHK(forward_core(e, training != 0, scales, len_seg, 0, s));
    if (out) CHK(export_out(e, out, B, T, s));
    return 0;
}

int ss_g3_backward(ss_engine* e, const float* d_out, void* stream) {
    if (e->kind != SS_GENERATOR_3) return fail("ss_g3_backward on a Generator_6 engine");
    Own own(e, stream);
    hipStream_t s = own.s;
    if (!e->have_fwd) return fail("backward without a preceding forward");
    CHK(import_dout(e, d_out, e->curB, e->curT, s));
    return backward_core(e, s);
}

int ss_g3_rhythm(ss_engine* e, const float* x_org, int B, int T, float* codes, void* stream) {
    if (e->kind != SS_GENERATOR_3) return fail("ss_g3_rhythm on a Generator_6 engine");
    Own own(e, stream);
    hipStream_t s = own.s;
    CHK(geometry(e, B, T, s));
    const ss_hparams& h = e->hp;
    const long TP = T + 2 * HALO;
    HIPCHK(copy_rows(x_org, h.dim_freq, (long)T * h.dim_freq, e->org + HALO * h.dim_freq, h.dim_freq, TP * h.dim_freq, B, T,
                     h.dim_freq, s));
    CHK(conv_pack_all(e, e->ct, s));
    PrepTable tb;
    tb.n = 0;
    tb.img_bf16 = e->img16();
    CHK(lstm_prep(e, e->lt, tb, s));
    HIPCHK(prep_run(tb, s));
    CHK(act_scales_all(e, s));
    CHK(conv_block_fwd(e, e->ct, Slab{e->org, h.dim_freq}, Slab{e->act_t, h.dim_enc_2, nullptr, e->act_scale + e->ct.scale_i}, s));
    CHK(lstm_fwd(e, e->lt, Slab{e->act_t, h.dim_enc_2, nullptr, e->act_scale + e->ct.scale_i}, s));
    // codes = cat(fwd[:, 7::8], bwd[:, ::8]) (model.py:84-87): reuse the decoder-input assembler on a 2H-wide row and pick t % freq == 0
    CodeSrc src{e->lt.out[0], nullptr, h.dim_neck_2, h.freq_2, 0};
    HIPCHK(build_dec_in(&src, 1, nullptr, 0, 2 * h.dim_neck_2, e->d_ot, 2 * h.dim_neck_2, B, T, s));
    const int W = 2 * h.dim_neck_2;
    HIPCHK(copy_rows(e->d_ot + HALO * W, (long)h.freq_2 * W, TP * W, codes, W, (long)(T / h.freq_2) * W, B, T / h.freq_2, W, s));
    e->have_fwd = false;
    return 0;
}

int ss_g6_forward(ss_engine* e, const float* x_org, const float* f0_trg, const float* scales, const int* len_seg, int B,
                  int T, int training, float* out, void* stream) {
    if (e->kind != SS_GENERATOR_6) return fail("ss_g6_forward on a Generator_3 engine");
    CHK(entry_check(e));
    Own own(e, stream);
    hipStream_t s = own.s;
    if (training && T != e->hp.max_len_pad) return fail("train-mode forward needs T == max_len_pad (model.py:370)");
    if (training && (!scales || !len_seg)) return fail("train-mode forward needs the InterpLnr draws");
    CHK(geometry(e, B, T, s));
    const ss_hparams& h = e->hp;
    const long TP = T + 2 * HALO;
    HIPCHK(copy_rows(x_org, h.dim_freq, (long)T * h.dim_freq, e->org + HALO * h.dim_freq, h.dim_freq, TP * h.dim_freq, B, T,
                     h.dim_freq, s));
    HIPCHK(copy_rows(f0_trg, h.dim_f0, (long)T * h.dim_f0, e->in_f0 + HALO * e->f0p, e->f0p, TP * e->f0p, B, T, h.dim_f0, s));
    CHK(forward_core(e, training != 0, scales, len_seg, 0, s));
    if (out) CHK(export_out(e, out, B, T, s));
    return 0;
}

int ss_g6_backward(ss_engine* e, const float* d_out, void* stream) {
    if (e->kind != SS_GENERATOR_6) return fail("ss_g6_backward on a Generator_3 engine");
    Own own(e, stream);
    hipStream_t s = own.s;
    if (!e->have_fwd) return fail("backward without a preceding forward");
    CHK(import_dout(e, d_out, e->curB, e->curT, s));
    return backward_core(e, s);
}

static int g3_step_body(ss_engine* e, const float* mel, const float* f0, const float* emb, const int* len_org,
                        const float* scales, const int* len_seg, int B, int T, float grad_scale, int flags, float* loss,
                        hipStream_t s) {
    const ss_hparams& h = e->hp;
    const long TP = T + 2 * HALO;
    prof_tick(e);
    // solver.py:160-163: resample [mel | f0] with the utterance lengths, re-quantise the f0 channel
    HIPCHK(interp_plan(e->plan[0], scales, len_seg, len_org, 0, B, s));
    HIPCHK(interp_quant(e->plan[0], mel, f0, h.dim_freq, e->in_mel + HALO * h.dim_freq, h.dim_freq, TP * h.dim_freq,
                        e->in_f0 + HALO * e->f0p, e->f0p, TP * e->f0p, h.dim_f0, e->qidx, B, s));
    // x_org and the speaker embedding are first read by Encoder_t / the decoder input: their copies ride on the branch stream
    e->late_org = mel;
    e->late_emb = emb;
    e->prezero = true;
    const int frc = forward_core(e, true, scales, len_seg, 1, s);                           // solver.py:165
    e->prezero = false;
    e->late_org = e->late_emb = nullptr;
    CHK(frc);
    const int C = e->head_out;
    HIPCHK(mse_loss(e->out_slab + HALO * C, C, TP * C, e->org + HALO * C, C, TP * C, e->d_out_slab + HALO * C, C, TP * C, B, T,
                    C, 1.0f, e->loss_part, loss, s));                                       // solver.py:166
    if (flags & SS_STEP_SPLIT_BACKWARD) {         // data parallel: stop once the decoder + head gradients are complete
        CHK(backward_decoder(e, s));
        if (!(flags & SS_STEP_SPLIT_NO_JOIN)) return join_side(e, s);
        // the weight-gradient GEMMs stay on the side stream (joined by backward_encoder, as in the one-call step);
        // ss_wait_decoder_grads() orders the consumer of the decoder range behind both parts
        HIPCHK(hipEventRecord(e->ev_dec[0], s));
        e->dec_pending = 1;
        if (e->side_used) {
            HIPCHK(hipEventRecord(e->ev_dec[1], e->side));
            e->dec_pending = 2;
        }
        return 0;
    }
    {
        AdamEarly ae(e, !(flags & SS_STEP_NO_ADAM), grad_scale);      // (its scope covers the optimiser call: the early range's bookkeeping ends with it)
        CHK(backward_core(e, s));                                                           // solver.py:170-171
        if (!(flags & SS_STEP_NO_ADAM)) CHK(adam_enqueue(e, grad_scale, s));                // solver.py:172
    }
    return 0;
}

// Length buckets (SS_STEP_BUCKET): a bucketed step runs with max_len_pad = T; a step WITHOUT the flag runs with the max_len_pad the
// engine was created with again, whatever bucket came before (a loader that mixes buckets with full-length batches sends the
// latter without the flag: speechsplit_amd/solver.py).
static int apply_bucket(ss_engine* e, int T, int flags) {
    int want = e->bound_max_len_pad;
    if (flags & SS_STEP_BUCKET) {      // this batch's length bucket: the step runs with max_len_pad = T
        if (T < 8 || T > e->maxT || T % 8) return fail("SS_STEP_BUCKET: T must be a multiple of 8 within the engine's max_frames");
        want = T;
    }
    if (e->hp.max_len_pad != want) {
        e->hp.max_len_pad = want;
        e->curB = e->curT = 0;         // InterpLnr plans are sized by max_len_pad: carve again
    }
    return 0;
}

int ss_g3_train_step(ss_engine* e, const float* mel, const float* f0, const float* emb, const int* len_org,
                     const float* scales, const int* len_seg, int B, int T, float grad_scale, int flags, float* loss,
                     void* stream) {
    if (e->kind != SS_GENERATOR_3) return fail("ss_g3_train_step on a Generator_6 engine");
    CHK(entry_check(e));
    Own own(e, stream);
    hipStream_t s = own.s;
    CHK(apply_bucket(e, T, flags));
    const ss_hparams& h = e->hp;
    if (T != h.max_len_pad) return fail("training needs T == max_len_pad (model.py:105,157,370); pass SS_STEP_BUCKET for a length-bucketed batch");
    CHK(geometry(e, B, T, s));
    flags &= ~SS_STEP_BUCKET;
    return g3_step_body(e, mel, f0, emb, len_org, scales, len_seg, B, T, grad_scale, flags, loss, s);
}

int ss_train_finish(ss_engine* e, float grad_scale, int flags, void* stream) {
    Own own(e, stream);
    hipStream_t s = own.s;
    if (!e->have_fwd) return fail("ss_train_finish without a preceding ss_*_train_step(SS_STEP_SPLIT_BACKWARD)");
    CHK(backward_encoder(e, s));
    if (!(flags & SS_STEP_NO_ADAM)) CHK(adam_enqueue(e, grad_scale, s));
    return 0;
}

int ss_wait_decoder_grads(ss_engine* e, void* consumer_stream) {
    if (!e->dec_pending) return fail("ss_wait_decoder_grads without a preceding SS_STEP_SPLIT_BACKWARD | SS_STEP_SPLIT_NO_JOIN step");
    hipStream_t c = S(consumer_stream);
    HIPCHK(hipStreamWaitEvent(c, e->ev_dec[0], 0));
    if (e->dec_pending == 2) HIPCHK(hipStreamWaitEvent(c, e->ev_dec[1], 0));
    e->dec_pending = 0;
    return 0;
}

void* ss_side_stream(ss_engine* e) { return (void*)e->side; }

long ss_grad_split(const ss_engine* e) {
    for (const auto& p : e->params)
        if (p.name.rfind("decoder.", 0) == 0) return p.offset;
    return e->arena;
}

int ss_g6_train_step(ss_engine* e, const float* mel, const float* f0_onehot, const int* target_idx, const float* scales,
                     const int* len_seg, int B, int T, float grad_scale, int flags, float* loss, void* stream) {
    if (e->kind != SS_GENERATOR_6) return fail("ss_g6_train_step on a Generator_3 engine");
    CHK(apply_bucket(e, T, flags));
    Own own(e, stream);
    hipStream_t s = own.s;
    prof_tick(e);
    CHK(ss_g6_forward(e, mel, f0_onehot, scales, len_seg, B, T, 1, nullptr, (void*)s));
    const long TP = T + 2 * HALO;
    const int C = e->head_out;
    HIPCHK(ce_loss(e->out_slab + HALO * C, C, TP * C, target_idx, e->d_out_slab + HALO * C, C, TP * C, B, T, C, 1.0f,
                   e->loss_part, loss, s));
    {
        AdamEarly ae(e, !(flags & SS_STEP_NO_ADAM), grad_scale);
        CHK(backward_core(e, s));
        if (!(flags & SS_STEP_NO_ADAM)) CHK(adam_enqueue(e, grad_scale, s));
    }
    return 0;
}

int ss_collate(const float* mel_cat, const float* f0_cat, const float* emb_tab, const long* row0, const int* len, const int* item,
               int B, int T, int n_mel, int emb_dim, float* mel, float* f0, float* emb, void* stream) {
    if (!mel_cat || !f0_cat || !emb_tab || !row0 || !len || !item || !mel || !f0 || !emb) return fail("ss_collate: null pointer");
    if (B <= 0 || T <= 0 || n_mel <= 0 || emb_dim <= 0) return fail("ss_collate: bad shape");
    HIPCHK(collate(mel_cat, f0_cat, emb_tab, row0, len, item, B, T, n_mel, emb_dim, mel, f0, emb, S(stream)));
    return 0;
}

int ss_melspec_frames(int n) { return n >= 513 ? (n + 256) / 256 : 0; }

int ss_melspec(const double* wav, int n, const double* mel_basis, int n_mels, float* out, void* stream) {
    if (!wav || !mel_basis || !out) return fail("ss_melspec: null pointer");
    if (n < 513) return fail("ss_melspec: at least 513 samples (reflect padding by 512)");
    HIPCHK(melspec(wav, n, mel_basis, n_mels, out, ss_melspec_frames(n), S(stream)));
    return 0;
}

int ss_f0_normalize(const double* f0, int n, float* out, void* stream) {
    if (!f0 || !out || n < 1) return fail("ss_f0_normalize: bad arguments");
    HIPCHK(f0_normalize(f0, n, out, S(stream)));
    return 0;
}

int ss_interp_forward(ss_engine* e, const float* x, const int* len_seq, const float* scales, const int* len_seg, int B, int T,
                      int C, float* y, int* i0, float* lam, int* counts, void* stream) {
    Own own(e, stream);
    hipStream_t s = own.s;
    if (!e->ws) return fail("engine is not bound");
    if (B > e->maxB || T > e->maxT) return fail("ss_interp_forward: batch / frames exceed the engine limits");
    if (!e->curB) CHK(geometry(e, e->maxB, e->maxT, s));
    InterpPlan pl = e->plan[3];     // standalone calls use the last plan slot with their own T
    pl.T = T;
    if ((long)B * (T + 1) > (long)e->curB * (e->curT + 1)) return fail("ss_interp_forward: plan storage too small");
    const int P = pl.P;
    HIPCHK(interp_plan(pl, scales, len_seg, len_seq, 0, B, s));
    HIPCHK(interp_gather(pl, x, C, (long)T * C, y, C, (long)P * C, C, B, s));
    if (i0) HIPCHK(hipMemcpyAsync(i0, pl.i0, (long)B * P * 4, hipMemcpyDeviceToDevice, s));
    if (lam) HIPCHK(hipMemcpyAsync(lam, pl.lam, (long)B * P * 4, hipMemcpyDeviceToDevice, s));
    if (counts) HIPCHK(hipMemcpyAsync(counts, pl.counts, (long)B * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

int ss_interp_backward(ss_engine* e, const float* dy, int B, int T, int C, float* dx, void* stream) {
    Own own(e, stream);
    hipStream_t s = own.s;
    if (!e->ws || !e->curB) return fail("ss_interp_backward without ss_interp_forward");
    InterpPlan pl = e->plan[3];
    pl.T = T;
    const int P = pl.P;
    HIPCHK(interp_scatter(pl, dy, C, (long)P * C, dx, C, (long)T * C, C, B, s));
    return 0;
}

int ss_op_gemm(const float* a, long lda, const float* b, long ldb, float* c, long ldc, const float* bias, int M, int N, int K,
               int flags, int ksplit, void* stream) {
    GemmDesc d{};
    d.A = {a, lda, 0, 0, 0};
    d.B = {b, ldb, 0, 0, 0};
    d.C = c;
    d.ldc = ldc;
    d.bias = bias;
    d.M = M;
    d.N = N;
    d.K = K;
    d.batch = 1;
    d.ksplit = ksplit < 1 ? 1 : ksplit;
    d.flags = (flags & 1 ? GEMM_TA : 0) | (flags & 2 ? GEMM_TB : 0) | (flags & 8 ? GEMM_BF16 : 0) | (flags & 16 ? GEMM_F16X2 : 0) | (d.ksplit > 1 ? GEMM_ACCUM : 0);
    HIPCHK(launch_gemm(d, S(stream)));
    return 0;
}

int ss_op_split_image(const float* src, long ld, long rows, int cols, float scale, float* img, long ldi, void* stream) {
    if (!src || !img) return fail("ss_op_split_image: null pointer");
    HIPCHK(split_image(src, ld, rows, cols, nullptr, scale, img, ldi, nullptr, S(stream)));
    return 0;
}

int ss_op_gemm_img(const float* a_img, long lda, const float* b_img, long ldb, float* c, long ldc, const float* bias, int M, int N, int K, int flags,
                   int ksplit, int cfg, float scale_a, float scale_b, int a_seglen, long a_segstride, float* part, const void* zeros, void* stream) {
    static float* sc = nullptr;           // the two scales as device words (test hook: one call at a time)
    if (!sc) HIPCHK(hipMalloc((void**)&sc, 256));
    const float h[2] = {scale_a, scale_b};
    HIPCHK(hipMemcpyAsync(sc, h, sizeof(h), hipMemcpyHostToDevice, S(stream)));
    ImgGemmDesc d{};
    d.A = {a_img, lda, 0, a_seglen, a_segstride};
    d.B = {b_img, ldb, 0, 0, 0};
    d.C = c;
    d.ldc = ldc;
    d.bias = bias;
    d.M = M;
    d.N = N;
    d.K = K;
    d.batch = 1;
    d.ksplit = ksplit < 1 ? 1 : ksplit;
    d.flags = (flags & 1 ? GEMM_TA : 0) | (flags & 2 ? GEMM_TB : 0) | (flags & 4 ? GEMM_ACCUM : 0);
    d.bf16 = (flags & 8) ? 1 : 0;           // single-piece form: the operands are plain bf16 matrices, ld in elements
    d.part = part;
    d.scale_a = d.bf16 ? nullptr : sc;
    d.scale_b = d.bf16 ? nullptr : sc + 1;
    d.zeros = zeros;
    d.cfg = cfg;
    d.diag = g_gemm_diag;
    if (g_img_xcc) {                      // test hook for the work-queue form: ss_tune("img_xcc", mask of allowed XCDs; 255 = queue without a filter)
        if (!g_img_wq) HIPCHK(hipMalloc((void**)&g_img_wq, IMG_WQ_BYTES));
        HIPCHK(hipMemsetAsync(g_img_wq, 0, (g_img_xcc & 0x100) ? IMG_WQ_BYTES : 8, S(stream)));
        d.wq = g_img_wq;
        d.xcc_allow = (unsigned)g_img_xcc;
    }
    if (!gemm_img_supported(d)) return fail("ss_op_gemm_img: shape / alignment not supported by the image GEMM");
    HIPCHK(launch_gemm_img(d, S(stream)));
    return 0;
}

int ss_set_precision(ss_engine* e, int precision) {
    if (!e) return fail("ss_set_precision: null engine");
    if (precision != SS_PRECISION_F32 && precision != SS_PRECISION_BF16) return fail("ss_set_precision: unknown precision");
    e->precision = precision;
    return 0;
}

int ss_profile(ss_engine* e, unsigned class_mask) {
    if (!e) return fail("ss_profile: null engine");
    if (class_mask) e->prof_n = 0;
    e->prof_mask = class_mask;
    e->prof_ctr = 0;
    e->prof_live = true;
    return 0;
}

int ss_profile_sample(ss_engine* e, int every_nth_step) {
    if (!e || every_nth_step < 1) return fail("ss_profile_sample: needs an engine and n >= 1");
    e->prof_every = every_nth_step;
    e->prof_ctr = 0;
    e->prof_live = true;
    return 0;
}

int ss_profile_read(ss_engine* e, int klass, int* launches, double* total_us, double* total_flops) {
    if (!e) return fail("ss_profile_read: null engine");
    int n = 0;
    double us = 0, fl = 0;
    for (int i = 0; i < e->prof_n; ++i) {
        if (e->prof_rec[i].klass != klass) continue;
        if (hipEventSynchronize(e->prof_ev[2 * i + 1]) != hipSuccess) return fail("ss_profile_read: event sync failed");
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]) != hipSuccess)
            return fail("ss_profile_read: elapsed time failed");
        us += ms * 1e3;
        fl += e->prof_rec[i].flops;
        ++n;
    }
    if (launches) *launches = n;
    if (total_us) *total_us = us;
    if (total_flops) *total_flops = fl;
    return 0;
}

// The brackets of ss_profile as a timeline: per record (class, start, end) in microseconds relative to the first record's start event --
// timestamps from the real run (no tracing tool slowing the host down), across the engine's streams.  3 doubles per record, enqueue order.
int ss_profile_timeline(ss_engine* e, double* out, int cap) {
    if (!e || !out || cap < 0) return fail("ss_profile_timeline: null argument");
    const int n = e->prof_n < cap ? e->prof_n : cap;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(e->prof_ev[2 * i + 1]) != hipSuccess) return fail("ss_profile_timeline: event sync failed");
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, e->prof_ev[0], e->prof_ev[2 * i]) != hipSuccess || hipEventElapsedTime(&b, e->prof_ev[0], e->prof_ev[2 * i + 1]) != hipSuccess)
            return fail("ss_profile_timeline: elapsed time failed");
        out[3 * i] = e->prof_rec[i].klass + 100 * e->prof_rec[i].stream;      // class + 100 x stream (0 main, 1 side, 2 / 3 branch streams)
        out[3 * i + 1] = a * 1e3;
        out[3 * i + 2] = b * 1e3;
    }
    return n;
}

// Where do the workgroups of a launch go?  n_wg workgroups of `threads` threads and lds_bytes of LDS each record (XCC_ID register, HW_ID
// register) in launch order; each then idles `hold_ticks` (100 MHz) so that the whole grid has to be resident at once.
__global__ void xcc_map_kernel(unsigned* out, long long hold_ticks) {
    extern __shared__ unsigned char xcc_map_lds[];
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        out[2 * blockIdx.x] = x;
        out[2 * blockIdx.x + 1] = h;
        xcc_map_lds[0] = (unsigned char)x;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
}
int ss_debug_xcc_map(int n_wg, int threads, int lds_bytes, int hold_us, unsigned* out_dev, void* stream) {
    if (!out_dev || n_wg < 1 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || hold_us < 0 || hold_us > 2000)
        return fail("ss_debug_xcc_map: bad arguments");
    if (lds_bytes > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)xcc_map_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(xcc_map_kernel, dim3(n_wg), dim3(threads), lds_bytes, S(stream), out_dev, (long long)hold_us * 100);
    HIPCHK(hipGetLastError());
    return 0;
}

int ss_debug_img_wq(unsigned* out_host, int words) {
    if (!out_host || words < 1 || words * 4L > IMG_WQ_BYTES || !g_img_wq) return fail("ss_debug_img_wq: nothing logged / bad size");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_host, g_img_wq, words * 4L, hipMemcpyDeviceToHost));
    return 0;
}

int ss_debug_gemm_phases(unsigned long long* out24, int reset) {
    if (!out24) return fail("ss_debug_gemm_phases: null pointer");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(gemm_phase_probe(out24, reset != 0));
    return 0;
}

int ss_tune(const char* key, int value) {
    const std::string k = key ? key : "";
    if (k == "lstm_nw" && (value == 4 || value == 8 || value == 16)) g_lstm_nw = value;
    else if (k == "lstm_g" && value >= 0 && value <= 16) g_lstm_g = value;
#ifdef SS_DIAG
    else if (k == "lstm_mode" && value >= 0 && value <= 4) g_lstm_mode = value;
    else if (k == "gemm_diag" && value >= 0 && value < 2048) g_gemm_diag = value;
    else if (k == "seq_prio" && value >= 0 && value < 65536) g_seq_prio = value;
#else
    else if (k == "seq_prio" && (value == 0 || value == 1)) g_seq_prio = value;
#endif
    else if (k == "seq_tag" && (value == 0 || value == 1)) g_seq_tag = value;
    else if (k == "op_time_major" && (value == 0 || value == 1)) g_op_time_major = value;
    else if (k == "seq_wlead" && value >= 0 && value < 32) g_seq_wlead = value;
    else if (k == "seq_var" && value >= 0 && value < 16) g_seq_var = value;
    else if (k == "prewarm" && value >= 0 && value <= 3) g_prewarm = value;
    else if (k == "batch_dirs" && value >= 0 && value <= 2) g_batch_dirs = value;
    else if (k == "compact0" && (value == 0 || value == 1)) g_compact0 = value;
    else if (k == "presplit" && value >= 0 && value <= 15) g_presplit = value;
    else if (k == "bf16_img" && (value == 0 || value == 1)) g_bf16_img = value;
    else if (k == "bf16_img_mask") g_bf16_img_mask = value;
    else if (k == "seq_hi" && (value == 0 || value == 1)) g_seq_hi = value;
    else if (k == "seq_skip32" && (value == 0 || value == 1)) g_seq_skip32 = value;
    else if (k == "pack_one" && (value == 0 || value == 1)) g_pack_one = value;
    else if (k == "trunk_bwd_par" && (value == 0 || value == 1)) g_trunk_bwd_par = value;
    else if (k == "wgrad_fused" && (value == 0 || value == 1)) g_wgrad_fused = value;
    else if (k == "img" && (value == 0 || value == 1)) g_img = value;
    else if (k == "img_mask" && value >= 0 && value < 2048) g_img_mask = value;
    else if (k == "img_batch" && (value == 0 || value == 1)) g_img_batch = value;
    else if (k == "dp_model" && value >= 0 && value <= 64) g_dp_model = value;
    else if (k == "dp_buckets" && (value == 0 || value == 1)) g_dp_buckets = value;
    else if (k == "img_dw_cfg" && value >= -1 && value <= 3) g_img_dw_cfg = value;
    else if (k == "img_dw_wgs" && value >= 32 && value <= 4096) g_img_dw_wgs = value;
    else if (k == "img_cfg" && value >= -1 && value <= 3) g_img_cfg = value;
    else if (k == "dw_wgs" && value >= 64 && value <= 4096) g_dw_wgs = value;
    else if (k == "trunk_indep" && (value == 0 || value == 1)) g_trunk_indep = value;
    else if (k == "branch_low" && (value == 0 || value == 1)) g_branch_low = value;
    else if (k == "gemm_ws" && value >= 0 && value <= 2) g_gemm_ws = value;
    else if (k == "seq_spin_log2" && value >= 0 && value <= 24) g_seq_spin_log2 = value;
    else if (k == "overlap" && (value == 0 || value == 1)) g_overlap = value;
    else if (k == "defer_dw" && (value == 0 || value == 1)) g_defer_dw = value;
    else if (k == "dx_batched" && value >= 0 && value <= 2) g_dx_batched = value;
    else if (k == "conv_want" && value >= 0) g_conv_want = value;
    else if (k == "conv_small_old" && (value == 0 || value == 1)) g_conv_small_old = value;
    else if (k == "small_lds" && value >= 0 && value <= 2) g_small_lds = value;
    else if (k == "small_prio" && (value == 0 || value == 1)) g_small_prio = value;
    else if (k == "gemm_tr" && value >= 0 && value <= 2) g_gemm_tr = value;
#ifdef SS_DIAG
#endif
    else if (k == "own_streams" && (value == 0 || value == 1)) g_own_streams = value;
    else if (k == "early_join" && (value == 0 || value == 1)) g_early_join = value;
    else if (k == "conv_par" && (value == 0 || value == 1)) g_conv_par = value;
    else if (k == "exp" && value >= 0) g_exp = value;
    else if (k == "adam_early" && (value == 0 || value == 1)) g_adam_early = value;
    else if (k == "enc_t_first" && (value == 0 || value == 1)) g_enc_t_first = value;
    else if (k == "conv_dw_off" && value >= 0 && value <= 3) g_conv_dw_off = value;
    else if (k == "dec_tail_split" && value >= 0 && value <= 10) g_dec_tail_split = value;
    else if (k == "early_dw" && (value == 0 || value == 1)) g_early_dw = value;
    else if (k == "xcd_dw" && (value == 0 || value == 1)) g_xcd_dw = value;
    else if (k == "dp_emulate" && (value == 0 || value == 1)) g_dp_emulate = value;
    else if (k == "gn_gather" && (value == 0 || value == 1)) g_gn_gather = value;
    else if (k == "unpack_later" && (value == 0 || value == 1)) g_unpack_later = value;
    else if (k == "part_splitk" && (value == 0 || value == 1)) g_part_splitk = value;
    else if (k == "gn_part" && (value == 0 || value == 1)) g_gn_part = value;
    else if (k == "img_xcc" && value >= 0 && value <= 511) g_img_xcc = value;      // bit 8: keep a placement log (ss_debug_img_wq)
    else if (k == "probe_queues" && (value == 0 || value == 1)) g_probe_queues = value;
    else if (k == "prio_order" && (value == 0 || value == 1)) g_prio_order = value;
    else if (k == "flat_rows" && (value == 0 || value == 1)) g_flat_rows = value;
    else if (k == "deterministic" && (value == 0 || value == 1)) g_deterministic = value;
    else if (k == "split" && (value == 0 || value == 1)) g_split = value;
    else if (k == "persist" && (value == 0 || value == 1)) g_persist = value;
    else if (k == "side_prio" && (value == 0 || value == 1)) g_side_prio = value;
    else if (k == "gemm_bk" && (value == 16 || value == 32)) g_gemm_bk = value;
    else if (k == "gemm_want" && value >= 1) g_gemm_want = value;
    else if (k == "gemm_mode" && (value == 0 || value == 1)) g_gemm_mode = value;
    else if (k == "fwd_f16x2" && (value == 0 || value == 1)) g_fwd_f16x2 = value;
    else if (k == "bwd_f16x2" && (value == 0 || value == 1)) g_bwd_f16x2 = value;
    else return fail("ss_tune: unknown key or bad value: " + k
#ifndef SS_DIAG
                     + " (the wrong-result timing modes lstm_mode / gemm_diag / seq_prio > 1 exist only in the -DSS_DIAG build: make diag)"
#endif
    );
    return 0;
}

int ss_op_lstm_fwd(float* gates, const float* whh_f, const float* whh_b, float* out, float* csave, float* scratch,
                   long scratch_floats, int B, int T, int H, void* stream) {
    hipStream_t s = S(stream);
    if (H > 32) {
        const long half = 2L * (((B + 15) / 16) * 16) * H, wn = 2L * 4 * H * H;
        if (!scratch || scratch_floats < wn + 2 * half) return fail("ss_op_lstm_fwd: scratch too small");
        float* hf = scratch + wn;
        if (g_persist && lstm_seq_supported(B, H) && wn * 4 >= lstm_seq_xbytes(B, H, false)) {
            // exchange buffer in the (unused) packed-weight area, counters behind it
            HIPCHK(lstm_seq_fwd(gates, whh_f, whh_b, scratch, out, csave, (unsigned*)hf, nullptr, nullptr, 0, nullptr, B, T, H, true, g_op_time_major != 0, s));
            return 0;
        }
        HIPCHK(lstm_pack_w(whh_f, whh_b, scratch, H, 0, s));
        HIPCHK(hipMemsetAsync(hf, 0, 2 * half * 4, s));
        for (int st = 0; st < T; ++st)
            HIPCHK(lstm_step_fwd(gates, scratch, hf + (st & 1) * half, hf + ((st & 1) ^ 1) * half, out, csave, B, T, H, st, s));
    } else {
        HIPCHK(lstm_small_fwd(gates, whh_f, whh_b, out, csave, B, T, H, s));
    }
    return 0;
}

int ss_op_lstm_bwd(float* gates, const float* whh_f, const float* whh_b, const float* d_out, const float* csave,
                   float* scratch, long scratch_floats, int B, int T, int H, void* stream) {
    hipStream_t s = S(stream);
    if (H > 32) {
        const long half = 2L * (((B + 15) / 16) * 16) * 4 * H, wn = 2L * 4 * H * H;
        if (!scratch || scratch_floats < wn + 2 * half + 2L * B * H) return fail("ss_op_lstm_bwd: scratch too small");
        float* gf = scratch + wn;
        float* dc = gf + 2 * half;
        const long xbytes = lstm_seq_xbytes(B, H, true);
        if (g_persist && lstm_seq_supported(B, H) && scratch_floats * 4 >= xbytes + 4L * LSTM_SEQ_SYNC_WORDS) {     // [exchange tiles][flags]
            HIPCHK(lstm_seq_bwd(gates, whh_f, whh_b, scratch, d_out, csave, (unsigned*)((char*)scratch + xbytes), nullptr, nullptr, nullptr, nullptr, nullptr, 0, B, T,
                                H, true, g_op_time_major != 0, s));
            return 0;
        }
        HIPCHK(lstm_pack_w(whh_f, whh_b, scratch, H, 1, s));
        HIPCHK(hipMemsetAsync(gf, 0, 2 * half * 4, s));
        for (int st = 0; st < T; ++st)
            HIPCHK(lstm_step_bwd(gates, scratch, gf + (st & 1) * half, gf + ((st & 1) ^ 1) * half, d_out, csave, dc, B, T, H, st, s));
    } else {
        HIPCHK(lstm_small_bwd(gates, whh_f, whh_b, d_out, csave, B, T, H, s));
    }
    return 0;
}

int ss_op_lstm_wgrad(const float* dg, const float* x, long x_ld, const float* hout, float* gwih, float* gwhh, float* gb, float* scratch, long scratch_floats,
                     long R, int H, int In, void* stream) {
    if (H < 1 || H > 32 || In < 1 || R < 1) return fail("ss_op_lstm_wgrad: H in 1..32, In >= 1");
    WgradTable w{};
    w.n = 1;
    w.tiles_total = lstm_small_wgrad_tiles(H, In);
    w.row_groups = 16;
    const long need = (long)w.tiles_total * w.row_groups * 4096, ctr_floats = (w.tiles_total + 63) & ~63L;
    if (!scratch || scratch_floats < need + ctr_floats) return fail("ss_op_lstm_wgrad: scratch too small");
    w.part = scratch;
    w.ctr = (unsigned*)(scratch + need);
    HIPCHK(hipMemsetAsync(w.ctr, 0, ctr_floats * 4, S(stream)));
    // gwih [2][4H][In], gwhh [2][4H][H], gb [2][2][4H] (b_ih then b_hh per direction): accumulated into, as the engine's gradient arena is
    w.t[0] = WgradTask{dg, x, x_ld, hout, gwih, gwih + 4L * H * In, gwhh, gwhh + 4L * H * H, gb, gb + 4L * H, gb + 8L * H, gb + 12L * H, H, In, R, 0};
    HIPCHK(lstm_small_wgrad(w, S(stream)));
    return 0;
}

int ss_check(ss_engine* e, void* stream) {
    HIPCHK(hipStreamSynchronize(S(stream)));
    HIPCHK(hipGetLastError());
    for (LstmBlk* lb : {&e->ld, &e->l1, &e->l2, &e->lt}) {
        if (!lb->zf) continue;
        for (int l = 0; l < 2 * lb->L; ++l) {
            unsigned flag = 0;
            HIPCHK(hipMemcpy(&flag, l < lb->L ? lb->sync_f(l) : lb->sync_b(l - lb->L), 4, hipMemcpyDeviceToHost));
            if (flag) return fail("persistent LSTM kernel gave up waiting for its group (bounded spin expired): results are invalid");
        }
    }
    return sticky_check(e);       // an abort in ANY earlier step (the per-launch words above are re-zeroed every step)
}

int ss_clear_abort(ss_engine* e, void* stream) {
    if (!e) return fail("ss_clear_abort: null engine");
    HIPCHK(hipStreamSynchronize(S(stream)));
    if (e->sticky) *(volatile unsigned*)e->sticky = 0u;
    return 0;
}

unsigned ss_status(const ss_engine* e) { return e && e->sticky ? *(volatile unsigned*)e->sticky : 0u; }

int ss_debug_buffer(ss_engine* e, const char* name, float** ptr, long* rows, long* cols) {
    auto it = e->dbg.find(name);
    if (it == e->dbg.end()) return fail(std::string("no such buffer: ") + name);
    *ptr = it->second.first;
    *rows = (long)e->curB * (e->curT + 2 * HALO);
    *cols = it->second.second;
    return 0;
}

int ss_debug_relu_mask(ss_engine* e, const char* block, float* mask, void* stream) {
    if (!e || !block || !mask) return fail("ss_debug_relu_mask: null argument");
    if (!e->curB) return fail("ss_debug_relu_mask: no forward has run");
    auto it = e->dbg.find(std::string(block) + ".conv");
    if (it == e->dbg.end()) return fail(std::string("ss_debug_relu_mask: no such conv block: ") + block);
    ConvBlk* all[7] = {&e->c1[0], &e->c1[1], &e->c1[2], &e->c2[0], &e->c2[1], &e->c2[2], &e->ct};
    for (ConvBlk* cb : all)
        if (cb->Co && cb->cout == it->second.first) {
            const long TP = e->curT + 2 * HALO;
            HIPCHK(gn_relu_mask(cb->cout, cb->Co, TP * cb->Co, e->P + cb->ga, e->P + cb->be, cb->stats, mask, e->curB, e->curT,
                                cb->Co, S(stream)));
            return 0;
        }
    return fail("ss_debug_relu_mask: block has no storage");
}

long ss_op_conv_block_scratch(int B, int T, int Ci, int Co) {
    const long Cp = align4(Ci), R = (long)B * (T + 2 * HALO);
    const long np = align4((long)Co * Ci * 5) + 3L * Co;
    return 2 * np + 2L * Co * 5 * Cp + (long)Ci * 5 * Co + R * (2 * Cp + 3L * Co) + 2L * B * (Co / 16) + 64;
}

int ss_op_conv_block(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, const float* dy,
                     float* y, float* dx, float* gw, float* gb, float* ggamma, float* gbeta, float* scratch, long scratch_floats,
                     int B, int T, int Ci, int Co, void* stream) {
    hipStream_t s = S(stream);
    if (!x || !w || !bias || !gamma || !beta || !y || !scratch) return fail("ss_op_conv_block: null pointer");
    if (Co % 64 || T < 1 || T > 256 || B < 1) return fail("ss_op_conv_block: needs Co % 64 == 0 and 1 <= T <= 256");
    if (scratch_floats < ss_op_conv_block_scratch(B, T, Ci, Co)) return fail("ss_op_conv_block: scratch too small");
    // a stack engine that holds nothing but this block: the product's own conv_block_fwd / conv_block_bwd do the work
    ss_engine e{};
    e.kind = SS_GENERATOR_3;
    e.curB = B;
    e.curT = T;
    const long TP = T + 2 * HALO, R = (long)B * TP;
    ConvBlk cb;
    cb.Ci = Ci;
    cb.Co = Co;
    cb.Cp = (int)align4(Ci);
    cb.w = 0;
    cb.b = align4((long)Co * Ci * 5);
    cb.ga = cb.b + Co;
    cb.be = cb.ga + Co;
    cb.need_dx = dx != nullptr;
    const long np = cb.be + Co;
    HIPCHK(hipMemsetAsync(scratch, 0, ss_op_conv_block_scratch(B, T, Ci, Co) * 4, s));
    float* p = scratch;
    auto take = [&](long n) {
        float* q = p;
        p += align4(n);
        return q;
    };
    e.P = take(np);
    e.G = take(np);
    cb.wf = take((long)Co * 5 * cb.Cp);
    cb.wb = take((long)Ci * 5 * Co);
    cb.gp = take((long)Co * 5 * cb.Cp);
    float* xs = take(R * cb.Cp);
    float* dxs = take(R * cb.Cp);
    cb.cout = take(R * Co);
    float* ys = take(R * Co);
    float* dys = take(R * Co);
    cb.stats = take(2L * B * (Co / 16));
    e.amax = take(16);
    cb.amax_i = 0;
    HIPCHK(hipMemcpyAsync(e.P + cb.w, w, (long)Co * Ci * 5 * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(e.P + cb.b, bias, Co * 4L, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(e.P + cb.ga, gamma, Co * 4L, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(e.P + cb.be, beta, Co * 4L, hipMemcpyDeviceToDevice, s));
    HIPCHK(copy_rows(x, Ci, (long)T * Ci, xs + HALO * cb.Cp, cb.Cp, TP * cb.Cp, B, T, Ci, s));
    CHK(conv_pack_all(&e, cb, s));
    CHK(conv_block_fwd(&e, cb, Slab{xs, cb.Cp}, Slab{ys, Co}, s));
    HIPCHK(copy_rows(ys + HALO * Co, Co, TP * Co, y, Co, (long)T * Co, B, T, Co, s));
    if (dy) {
        if (!gw || !gb || !ggamma || !gbeta) return fail("ss_op_conv_block: backward needs the gradient outputs");
        HIPCHK(copy_rows(dy, Co, (long)T * Co, dys + HALO * Co, Co, TP * Co, B, T, Co, s));
        CHK(conv_block_bwd(&e, cb, Slab{dys, Co}, Slab{xs, cb.Cp}, dx ? Slab{dxs, cb.Cp} : Slab{nullptr, 0}, s));
        HIPCHK(hipMemcpyAsync(gw, e.G + cb.w, (long)Co * Ci * 5 * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(gb, e.G + cb.b, Co * 4L, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(ggamma, e.G + cb.ga, Co * 4L, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(gbeta, e.G + cb.be, Co * 4L, hipMemcpyDeviceToDevice, s));
        if (dx) HIPCHK(copy_rows(dxs + HALO * cb.Cp, cb.Cp, TP * cb.Cp, dx, Ci, (long)T * Ci, B, T, Ci, s));
    }
    return 0;
}

const char* ss_stream_report(const ss_engine* e) { return e ? e->stream_report.c_str() : ""; }

int ss_debug_names(ss_engine* e, char* buf, int cap) {
    std::string all;
    for (auto& kv : e->dbg) all += kv.first + "\n";
    if (buf && cap > 0) {
        std::strncpy(buf, all.c_str(), cap - 1);
        buf[cap - 1] = 0;
    }
    return (int)all.size();
}

}  // extern "C"

// ================================================================================================ RCCL (data parallel)
// The reference is single-device (solver.py:38); this is the exchange step of SURVEY.md section 8(e).  RCCL is reached through
// dlopen: no link-time dependency, and inside a PyTorch process the copy PyTorch already loaded is reused (two RCCL copies in
// one process would each bring their own bootstrap state).
namespace {

struct NcclId {
    char internal[128];
};
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.h) return 0;
    void* h = nullptr;
    for (const char* n : {"librccl.so", "librccl.so.1"})
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);                     // already in the process (PyTorch's)?
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(std::string("RCCL not found: ") + dlerror());
    Rccl r;
    r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString || !r.GroupStart || !r.GroupEnd) return fail("RCCL: missing symbols");
    g_rccl = r;
    return 0;
}
#define NCCLCHK(x)                                                                          \
    do {                                                                                    \
        int _r = (x);                                                                       \
        if (_r != 0) return fail(std::string(#x) + ": " + g_rccl.GetErrorString(_r));       \
    } while (0)

// Stand-in for a collective when a data-parallel run is MODELLED on one GPU (ss_tune("dp_model", ranks)): 32 workgroups (RCCL's
// channels) stream the range once -- read and write it back, the memory traffic of a reduction -- and then hold their place until the
// modelled duration has passed.  Model (the conservative one): ring all-reduce, 2 (R - 1) / R of the bytes through
// one xGMI link of 153 GB/s, plus 25 us of launch / protocol latency.
__global__ __launch_bounds__(256) void dp_model_kernel(float* __restrict__ p, long n4, long long ticks) {
    const long long t0 = wall_clock64();                         // constant 100 MHz
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
        asm volatile("" : "+v"(v));
        reinterpret_cast<f32x4*>(p)[i] = v;
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
// ss_tune("dp_emulate", 1) with dp_model = N: the stand-in MULTIPLIES its range by N -- the sum N identical ranks would produce.  With the
// 1 / N of the Adam step behind it every element must come out as in the one-GPU step: an element reduced twice, or never, shows in the
// first moment (tests/test_gpu_buckets_dp.py).  At world 1 a real all-reduce is the identity, so nothing else on one GPU checks that the
// bucket schedule covers the arena exactly once.
__global__ __launch_bounds__(256) void dp_scale_kernel(float* __restrict__ p, long n, float mul) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] *= mul;
}
double dp_model_us(long bytes, int ranks) { return 25.0 + 2.0 * (ranks - 1) / ranks * (double)bytes / 153e9 * 1e6; }

int allreduce_range(ss_engine* e, long off, long count, hipStream_t st) {
    if (off < 0 || count < 0 || off + count > e->arena) return fail("ss_allreduce_grads: range outside the gradient arena");
    if (count == 0) return 0;
    if (g_dp_model > 1 && (!e->comm || e->comm_world == 1)) {
        if (g_dp_emulate) {
            hipLaunchKernelGGL(dp_scale_kernel, dim3(256), dim3(256), 0, st, e->G + off, count, (float)g_dp_model);
            HIPCHK(hipGetLastError());
            return 0;
        }
        const long a = (off + 3) & ~3L, b = (off + count) & ~3L;       // whole float4s inside the range
        hipLaunchKernelGGL(dp_model_kernel, dim3(32), dim3(256), 0, st, e->G + a, b > a ? (b - a) / 4 : 0, (long long)(dp_model_us(count * 4, g_dp_model) * 100.0));
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (!e->comm) return fail("no communicator: call ss_comm_init first");
    NCCLCHK(g_rccl.AllReduce(e->G + off, e->G + off, (size_t)count, /*ncclFloat*/ 7, /*ncclSum*/ 0, e->comm, st));
    return 0;
}

// ss_dp_profile bracket: a timing event from the pool (null when profiling is off or an event could not be created)
hipEvent_t dp_prof_event(ss_engine* e) {
    if (!e->dp_prof) return nullptr;
    if (e->dp_ev_used == (int)e->dp_ev_pool.size()) {
        hipEvent_t ev = nullptr;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        e->dp_ev_pool.push_back(ev);
    }
    return e->dp_ev_pool[e->dp_ev_used++];
}

// see the declaration above fork_join
int dp_bucket(ss_engine* e, long off, long count, hipStream_t producer) {
    if (!e->dp_on || !g_dp_buckets || count <= 0) return 0;
    if (off < 0 || off + count > e->arena) return fail("dp_bucket: range outside the gradient arena");
    for (auto& r : e->dp_done)          // a range reduced twice would be summed twice: at world 1 nobody would notice, so it is checked here
        if (off < r.second && r.first < off + count) return fail("dp_bucket: gradient range handed to a collective twice");
    HIPCHK(hipEventRecord(e->ev_comm, producer));
    HIPCHK(hipStreamWaitEvent(e->comm_s, e->ev_comm, 0));
    hipEvent_t pa = dp_prof_event(e), pb = pa ? dp_prof_event(e) : nullptr;
    if (pb) HIPCHK(hipEventRecord(pa, e->comm_s));
    CHK(allreduce_range(e, off, count, e->comm_s));
    if (pb) {
        HIPCHK(hipEventRecord(pb, e->comm_s));
        e->dp_rec.push_back({off, count, pa, pb});
    }
    e->dp_done.push_back({off, off + count});
    return 0;
}

// end of a data-parallel backward: everything not yet handed to a collective (layer-0 convolutions, the encoder BLSTMs, Encoder_t, the
// status slot -- a few MB, final only now), then the main stream waits for the communication stream
int dp_finish(ss_engine* e, hipStream_t s) {
    std::sort(e->dp_done.begin(), e->dp_done.end());
    std::vector<std::pair<long, long>> rest;
    long at = 0;
    for (auto& r : e->dp_done) {
        if (r.first > at) rest.push_back({at, r.first});
        if (r.second > at) at = r.second;
    }
    if (at < e->arena) rest.push_back({at, e->arena});
    // ONE launch for all of them (ncclGroupStart / End); modelled: one stand-in over their total size
    HIPCHK(hipEventRecord(e->ev_comm, s));
    HIPCHK(hipStreamWaitEvent(e->comm_s, e->ev_comm, 0));
    hipEvent_t pa = nullptr, pb = nullptr;
    if (e->dp_prof) {
        if (!e->dp_bwd_end) HIPCHK(hipEventCreate(&e->dp_bwd_end));
        HIPCHK(hipEventRecord(e->dp_bwd_end, s));           // the backward's last kernel on the main stream: time zero of the record
        pa = dp_prof_event(e);
        pb = pa ? dp_prof_event(e) : nullptr;
        if (pb) HIPCHK(hipEventRecord(pa, e->comm_s));
    }
    const bool model = g_dp_model > 1 && (!e->comm || e->comm_world == 1);
    if (model && g_dp_emulate) {
        for (auto& r : rest) CHK(allreduce_range(e, r.first, r.second - r.first, e->comm_s));
    } else if (model) {
        long tot = 0;
        for (auto& r : rest) tot += r.second - r.first;
        hipLaunchKernelGGL(dp_model_kernel, dim3(32), dim3(256), 0, e->comm_s, e->G, 0L, (long long)(dp_model_us(tot * 4, g_dp_model) * 100.0));
        HIPCHK(hipGetLastError());
    } else {
        // an open group must be closed whatever happens inside it: a failing member would otherwise leave every later RCCL call of the
        // process queued in a group that never ends
        const bool grouped = rest.size() > 1;
        if (grouped) NCCLCHK(g_rccl.GroupStart());
        int rc = 0;
        for (auto& r : rest)
            if ((rc = allreduce_range(e, r.first, r.second - r.first, e->comm_s)) != 0) break;
        if (grouped) {
            const std::string first_err = rc ? g_err : std::string();
            const int ge = g_rccl.GroupEnd();
            if (rc) return fail(first_err);
            NCCLCHK(ge);
        } else if (rc) return rc;
    }
    if (pb) {
        long tot = 0;
        for (auto& r : rest) tot += r.second - r.first;
        HIPCHK(hipEventRecord(pb, e->comm_s));
        e->dp_rec.push_back({-1, tot, pa, pb});             // offset -1: the grouped rest
    }
    e->dp_done.clear();
    HIPCHK(hipEventRecord(e->ev_comm, e->comm_s));
    HIPCHK(hipStreamWaitEvent(s, e->ev_comm, 0));
    return 0;
}
// The communication stream is the engine's SECOND BRANCH stream, not a fifth stream: HIP serves a process's streams from four hardware
// queues, so a fifth one shares a queue with one of the four the engine already runs on -- measured with modelled collectives
// (gpurun_out/r03/dp: it landed on the MAIN stream's queue, every collective sat in front of the encoder backward's launches and the
// backward took 1.16 ms longer).  The second branch stream has its own queue (pick_streams) and is idle from the first tenth of the
// encoder-backward phase on (it carries the pitch BLSTM's backward); whatever it still holds is simply in front of the first bucket.
int dp_streams(ss_engine* e) {
    e->comm_s = e->side2;
    if (!e->ev_comm) HIPCHK(hipEventCreateWithFlags(&e->ev_comm, hipEventDisableTiming));
    return 0;
}

}  // namespace

extern "C" {

int ss_comm_unique_id(char* id128) {
    if (!id128) return fail("ss_comm_unique_id: null pointer");
    CHK(rccl_load());
    NcclId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, id.internal, 128);
    return 0;
}

int ss_comm_init(ss_engine* e, const char* id128, int rank, int world) {
    if (!e || !id128 || world < 1 || rank < 0 || rank >= world) return fail("ss_comm_init: bad arguments");
    if (!e->G) return fail("ss_comm_init: engine is not bound");
    if (e->comm) return fail("ss_comm_init: the engine already has a communicator");
    CHK(rccl_load());
    NcclId id;
    std::memcpy(id.internal, id128, 128);
    void* c = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&c, world, id, rank));
    e->comm = c;
    e->comm_rank = rank;
    e->comm_world = world;
    if (world > 1) e->lockstep = true;
    return 0;
}

int ss_set_lockstep(ss_engine* e, int on) {
    if (!e) return fail("ss_set_lockstep: null engine");
    e->lockstep = on != 0;
    return 0;
}

int ss_comm_destroy(ss_engine* e) {
    if (e && e->comm) {
        (void)hipDeviceSynchronize();
        NCCLCHK(g_rccl.CommDestroy(e->comm));
        e->comm = nullptr;
        e->comm_world = 1;
    }
    return 0;
}

int ss_allreduce_grads(ss_engine* e, long offset, long count, void* stream) {
    Own own(e, stream);
    return allreduce_range(e, offset, count, own.s);
}

// common part of the data-parallel steps: `body` enqueues forward + loss + the whole backward (no Adam); the gradient arena is reduced
// in per-layer buckets on the communication stream while the backward is still running (dp_bucket: each of the decoder's layers as
// its four weight-gradient GEMMs retire, the head, the two wide trunk layers, the rest), Adam with the 1 / world mean folded in
static int dp_step(ss_engine* e, hipStream_t s, const std::function<int(hipStream_t)>& body) {
    const int world = (g_dp_model > 1 && (!e->comm || e->comm_world == 1)) ? g_dp_model : e->comm_world;
    const float gs = 1.0f / (float)world;
    CHK(dp_streams(e));
    if (!e->side || !e->side2 || !g_overlap || !g_dp_buckets) {        // no branch streams (or round 2's plan asked for): the plain step, then the arena
        CHK(body(s));
        if (g_dp_buckets) CHK(allreduce_range(e, 0, e->arena, s));
        else {
            const long k = ss_grad_split(e);
            CHK(allreduce_range(e, k, e->arena - k, s));
            CHK(allreduce_range(e, 0, k, s));
        }
        return adam_enqueue(e, gs, s);
    }
    e->dp_done.clear();
    e->dp_rec.clear();              // ss_dp_profile keeps the LAST step's record
    e->dp_ev_used = 0;
    e->dp_on = true;
    const int rc = body(s);
    e->dp_on = false;
    CHK(rc);
    CHK(dp_finish(e, s));
    return adam_enqueue(e, gs, s);                          // the mean is folded into the Adam kernel
}

int ss_g3_dp_train_step(ss_engine* e, const float* mel, const float* f0, const float* emb, const int* len_org, const float* scales,
                        const int* len_seg, int B, int T, int flags, float* loss, void* stream) {
    if (e->kind != SS_GENERATOR_3) return fail("ss_g3_dp_train_step on a Generator_6 engine");
    if (!e->comm && g_dp_model < 2) return fail("ss_g3_dp_train_step: call ss_comm_init first");
    CHK(apply_bucket(e, T, flags));      // every rank runs the same bucket (speechsplit_amd/buckets.py)
    if (T != e->hp.max_len_pad) return fail("training needs T == max_len_pad (model.py:105,157,370); pass SS_STEP_BUCKET for a length-bucketed batch");
    CHK(entry_check(e));
    Own own(e, stream);
    hipStream_t s = own.s;
    CHK(geometry(e, B, T, s));
    return dp_step(e, s, [&](hipStream_t st) { return g3_step_body(e, mel, f0, emb, len_org, scales, len_seg, B, T, 1.0f, SS_STEP_NO_ADAM, loss, st); });
}

long ss_scratch_fallbacks(const ss_engine* e) { return e ? e->scratch_fallbacks : -1; }

int ss_dp_profile(ss_engine* e, int on) {
    if (!e) return fail("ss_dp_profile: null engine");
    e->dp_prof = on != 0;
    e->dp_rec.clear();
    e->dp_ev_used = 0;
    return 0;
}

int ss_dp_profile_read(ss_engine* e, double* out, int cap) {
    if (!e) return fail("ss_dp_profile_read: null engine");
    const int n = (int)e->dp_rec.size();
    if (!out) return n;
    if (n && !e->dp_bwd_end) return fail("ss_dp_profile_read: no data-parallel step has run with the profile on");
    for (int i = 0; i < n && i < cap; ++i) {
        const auto& r = e->dp_rec[i];
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventSynchronize(e->dp_bwd_end) != hipSuccess) return fail("ss_dp_profile_read: event sync failed");
        float ta = 0, tb = 0;
        if (hipEventElapsedTime(&ta, e->dp_bwd_end, r.a) != hipSuccess || hipEventElapsedTime(&tb, e->dp_bwd_end, r.b) != hipSuccess)
            return fail("ss_dp_profile_read: elapsed time failed");
        out[4 * i + 0] = (double)r.off;
        out[4 * i + 1] = (double)r.count;
        out[4 * i + 2] = ta * 1e3;
        out[4 * i + 3] = tb * 1e3;
    }
    return n < cap ? n : cap;
}

int ss_g6_dp_train_step(ss_engine* e, const float* mel, const float* f0_onehot, const int* target_idx, const float* scales, const int* len_seg,
                        int B, int T, int flags, float* loss, void* stream) {
    if (e->kind != SS_GENERATOR_6) return fail("ss_g6_dp_train_step on a Generator_3 engine");
    if (!e->comm && g_dp_model < 2) return fail("ss_g6_dp_train_step: call ss_comm_init first");
    CHK(entry_check(e));
    Own own(e, stream);
    hipStream_t s = own.s;
    return dp_step(e, s, [&](hipStream_t st) {
        return ss_g6_train_step(e, mel, f0_onehot, target_idx, scales, len_seg, B, T, 1.0f, (flags & SS_STEP_BUCKET) | SS_STEP_NO_ADAM, loss, (void*)st);
    });
}

}  // extern "C"
