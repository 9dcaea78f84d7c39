/* speechsplit_amd.h -- C ABI of the MI355X (gfx950) SpeechSplit engine.
 *
 * The reference (biggytruck/SpeechSplit) is pure PyTorch and has no FFI of its own; the
 * boundary a maintainer binds is therefore derived from what its Python surface needs.  Each entry point names
 * the reference statement(s) it replaces.  Conventions:
 *   - plain C types only; every pointer marked "dev" is caller-owned device memory (e.g. tensor.data_ptr());
 *     the engine never frees caller memory and allocates nothing after ss_bind;
 *   - every call is asynchronous on the hipStream_t passed as `stream` (void*; NULL = default stream);
 *   - return value 0 = ok, negative = error, text via ss_last_error() (no C++ exceptions cross the ABI);
 *   - one engine per device per process; an engine is not thread-safe;
 *   - tensors are contiguous fp32, batch-first [B, T, C] exactly as the reference passes them (model.py:297,337);
 *   - T must be a multiple of the code down-sampling factor 8 (model.py:87,223-227) and <= max_frames;
 *   - the random-resampling draws of InterpLnr (model.py:392-393 rand(B*7)+0.5; :399-402 randint) are INPUTS:
 *     `scales` f32 and `len_seg` i32, one [B*7] row per InterpLnr call in call order.  Given equal draws the
 *     index path is bit-exact against the reference.
 */
#ifndef SPEECHSPLIT_AMD_H
#define SPEECHSPLIT_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ss_engine ss_engine;

/* reference hparams.py:9-32 (only the values the hot path reads) */
typedef struct ss_hparams {
    int freq, dim_neck, freq_2, dim_neck_2, freq_3, dim_neck_3;
    int dim_enc, dim_enc_2, dim_enc_3, dim_freq, dim_spk_emb, dim_f0, chs_grp;
    int min_len_seg, max_len_seg, max_len_seq, max_len_pad;
} ss_hparams;

#define SS_GENERATOR_3 3 /* model.py:283 Generator_3 = Encoder_7 | Encoder_t | Decoder_3 */
#define SS_GENERATOR_6 6 /* model.py:324 Generator_6 = Encoder_t | Encoder_6 | Decoder_4 */
#define SS_INTERP_ONLY 0 /* model.py:355 a bare InterpLnr module: no parameters, ss_interp_* only (arenas may be NULL) */

#define SS_STEP_NO_ADAM 1 /* ss_*_train_step: stop after backward (data-parallel: all-reduce grads, then ss_adam_step) */
#define SS_STEP_SPLIT_BACKWARD 2 /* ss_g3_train_step: return once the head + decoder gradients (arena offsets >=
                                    ss_grad_split(), 80 % of the bytes, produced first) are complete, so their all-reduce
                                    overlaps the encoder backward that ss_train_finish() then enqueues */
#define SS_STEP_SPLIT_NO_JOIN 4 /* with SS_STEP_SPLIT_BACKWARD: do not make `stream` wait for the decoder's weight-gradient
                                    GEMMs (they run on an engine stream beside the encoder backward, as in the one-call
                                    step); the consumer of the decoder range orders itself with ss_wait_decoder_grads() */

#define SS_STEP_BUCKET 16 /* ss_*_train_step: T is the length bucket of THIS batch and the step runs with max_len_pad = T (what the
                             reference does when its hparams.max_len_pad is set to the bucket length: InterpLnr pads to it, the
                             encoders' len_org equals it, model.py:105,157,370).  T % 8 == 0, T <= max_frames.  A
                             bucket change re-plans the workspace (one memset, 0.1 - 0.2 ms at batch 64). */

const char* ss_last_error(void);
int ss_abi_version(void);

/* model.py:285-295 / 327-334 (constructor).  No device memory is touched until ss_bind. */
ss_engine* ss_create(int kind, const ss_hparams* hp, int max_batch, int max_frames);
void ss_destroy(ss_engine* e);

/* state_dict() table, reference key names and shapes, in parameters() order.
 * `offset` is in floats into each of the four arenas (params / grads / Adam m / Adam v). */
int ss_num_params(const ss_engine* e);
int ss_param_info(const ss_engine* e, int index, char* name, int name_cap, long* offset, int* ndim, long shape[3]);
long ss_arena_numel(const ss_engine* e);     /* floats per arena, alignment gaps included */
long ss_workspace_bytes(const ss_engine* e); /* activation workspace for (max_batch, max_frames) */

/* .to(device) (solver.py:65): adopt caller-allocated device memory.  All four arenas hold ss_arena_numel floats.
 * The workspace is zero-filled here. */
int ss_bind(ss_engine* e, float* params_dev, float* grads_dev, float* adam_m_dev, float* adam_v_dev, void* workspace_dev,
            long workspace_bytes, void* stream);

/* ---- Generator_3 (model.py:297-320) ---- */
/* G(x_f0, x_org, c_trg): x_f0 [B,T,337] = [mel 80 | f0 one-hot 257], x_org [B,T,80], c_trg [B,82] -> out [B,T,80].
 * training != 0 applies InterpLnr after each of the three encoder conv layers (model.py:203): scales/len_seg [3][B*7]. */
int ss_g3_forward(ss_engine* e, const float* x_f0_dev, const float* x_org_dev, const float* c_trg_dev,
                  const float* scales_dev, const int* len_seg_dev, int B, int T, int training, float* out_dev,
                  void* stream);
/* loss.backward() for the last ss_g3_forward: d_out [B,T,80] -> grads arena (overwritten, not accumulated). */
int ss_g3_backward(ss_engine* e, const float* d_out_dev, void* stream);
/* G.rhythm(x_org) (model.py:316-320): codes [B, T/8, 2] */
int ss_g3_rhythm(ss_engine* e, const float* x_org_dev, int B, int T, float* codes_dev, void* stream);

/* ---- Generator_6 (model.py:337-351) ---- */
/* P(x_org, f0_trg): x_org [B,T,80], f0_trg [B,T,257] -> logits [B,T,257]; training: scales/len_seg [3][B*7] (model.py:128) */
int ss_g6_forward(ss_engine* e, const float* x_org_dev, const float* f0_trg_dev, const float* scales_dev,
                  const int* len_seg_dev, int B, int T, int training, float* out_dev, void* stream);
int ss_g6_backward(ss_engine* e, const float* d_out_dev, void* stream);

/* ---- Solver.train step body (solver.py:157-172), fused: cat(mel,f0) -> InterpLnr -> quantize_f0 -> G -> mse(mean)
 *      -> backward -> Adam.  mel [B,T,80], f0 [B,T,1] (-1e10 = unvoiced), emb [B,82], len_org i32[B];
 *      scales/len_seg [4][B*7] (outer call first).  T must equal hp.max_len_pad (model.py:105,157).
 *      grad_scale multiplies the gradients (1/world_size under data parallelism).  loss: one device float. */
int ss_g3_train_step(ss_engine* e, const float* mel_dev, const float* f0_dev, const float* emb_dev,
                     const int* len_org_dev, const float* scales_dev, const int* len_seg_dev, int B, int T,
                     float grad_scale, int flags, float* loss_dev, void* stream);
/* Generator_6 has no training loop in the reference; cross-entropy against target_idx i32[B,T] is this
 * engine's choice.  f0_onehot [B,T,257]; scales/len_seg [3][B*7]. */
int ss_g6_train_step(ss_engine* e, const float* mel_dev, const float* f0_onehot_dev, const int* target_idx_dev,
                     const float* scales_dev, const int* len_seg_dev, int B, int T, float grad_scale, int flags,
                     float* loss_dev, void* stream);

/* second half of a SS_STEP_SPLIT_BACKWARD step: encoder backward (+ Adam unless SS_STEP_NO_ADAM) */
int ss_train_finish(ss_engine* e, float grad_scale, int flags, void* stream);
/* After a SS_STEP_SPLIT_BACKWARD | SS_STEP_SPLIT_NO_JOIN step and BEFORE ss_train_finish is enqueued: make `consumer_stream`
 * (e.g. the stream the all-reduce of the decoder range is launched on) wait until the head + decoder gradients are final. */
int ss_wait_decoder_grads(ss_engine* e, void* consumer_stream);
/* the engine stream that carries the decoder's weight-gradient GEMMs (a hipStream_t; NULL if the engine runs without
 * branch streams).  A collective launched from it is ordered behind those GEMMs by construction. */
void* ss_side_stream(ss_engine* e);
/* How ss_bind chose the engine's branch streams.  The step forks three branches off the stream it is issued on; HIP maps a process's
 * streams onto 4 in-order hardware queues without saying which, and two streams on one queue cannot overlap (measured +14 % on the
 * step).  ss_bind therefore creates a pool of streams, MEASURES which of them share a queue with the caller's stream and with each
 * other (a 400 us spin kernel on one stream delays a trivial kernel on another only if they share a queue; ~10 ms once), keeps three
 * on queues of their own and destroys the rest.  Issue the steps on the stream that was passed to ss_bind. */
const char* ss_stream_report(const ss_engine* e);
/* first arena offset (floats) of the decoder + head parameters; [0, split) is the encoder */
long ss_grad_split(const ss_engine* e);

/* ---- data parallel over RCCL (nothing to mirror: the reference is single-device, solver.py:38) ----
 * One process per GPU.  Rank 0 obtains a 128-byte id (ss_comm_unique_id) and hands it to the other ranks by any means
 * (a file, MPI, torch.distributed's store); every rank then calls ss_comm_init.  RCCL is dlopen'd -- inside a process that
 * already loaded it (PyTorch) that copy is used. */
int ss_comm_unique_id(char* id128);
int ss_comm_init(ss_engine* e, const char* id128, int rank, int world);
int ss_comm_destroy(ss_engine* e);
/* in-place sum over the ranks of grads[offset, offset + count) (floats), enqueued on `stream` */
int ss_allreduce_grads(ss_engine* e, long offset, long count, void* stream);
/* The whole data-parallel step on this rank's shard (utterances [rank*B, (rank+1)*B) of the global batch and the matching
 * slices of the draws): the one-GPU step with the gradient arena all-reduced in BUCKETS on a communication stream of the engine's
 * own while the backward is still running -- each decoder layer (25 / 25 / 11 MB for Generator_3) as soon as its weight-gradient
 * GEMMs have retired (layer 2 first; they run beside the encoder backward), the head, the two wide layers of the conv trunk as the
 * trunk's backward passes them, and last the few MB that are final only at the end (layer-0 convolutions, encoder BLSTMs, Encoder_t,
 * the status slot); then Adam with the 1/world mean folded in.  flags: 0 or SS_STEP_BUCKET (every rank passes the same T).  loss: this
 * rank's local mean loss.  ss_g6_dp_train_step: the same for Generator_6 (cross-entropy step of ss_g6_train_step).
 * ss_tune("dp_model", R) MODELS an R-rank run on one GPU without a communicator: every collective is replaced by a stand-in kernel of
 * the modelled duration (ring all-reduce over one 153 GB/s xGMI link + 25 us), so a kernel trace shows where each bucket would sit
 * (tools/dp_timeline.sh); ss_tune("dp_buckets", 0) restores round 2's two buckets behind the whole backward. */
int ss_g3_dp_train_step(ss_engine* e, const float* mel_dev, const float* f0_dev, const float* emb_dev, const int* len_org_dev,
                        const float* scales_dev, const int* len_seg_dev, int B, int T, int flags, float* loss_dev, void* stream);
int ss_g6_dp_train_step(ss_engine* e, const float* mel_dev, const float* f0_onehot_dev, const int* target_idx_dev, const float* scales_dev,
                        const int* len_seg_dev, int B, int T, int flags, float* loss_dev, void* stream);
/* Where the collectives of a data-parallel step sit (for the multi-GPU scaling record): with ss_dp_profile(e, 1) every collective of
 * ss_g3_dp_train_step / ss_g6_dp_train_step is bracketed by hipEvents on the communication stream and one event marks the end of the
 * backward on the main stream.  ss_dp_profile_read (synchronises) returns the number of collectives of the LAST step and, for up to
 * `cap` of them in enqueue order, four doubles each: arena offset (-1: the grouped rest at the backward's end), element count, start
 * and end in microseconds RELATIVE TO THE BACKWARD'S END (negative = the collective ran beside the backward).  out == NULL: count only. */
/* Number of launches since ss_create that found the step's scratch (split-K partial slabs, column-sum partials, fused encoder weight gradients)
 * exhausted and took their slower fallback (atomics / unsplit / one workgroup per column block).  0 in a healthy run: an undersized scratch shows
 * up here instead of only in a profile. */
long ss_scratch_fallbacks(const ss_engine* e);
int ss_dp_profile(ss_engine* e, int on);
int ss_dp_profile_read(ss_engine* e, double* out, int cap);

/* torch.optim.Adam(G.parameters(), lr, [beta1, beta2]) (solver.py:62,172).  `step` = updates already applied. */
int ss_set_adam(ss_engine* e, double lr, double beta1, double beta2, double eps, long step, void* stream);
int ss_adam_step(ss_engine* e, float grad_scale, void* stream);
int ss_zero_grads(ss_engine* e, void* stream);

/* ---- InterpLnr as a standalone module (model.py:355-436; solver.py:59,161) ---- */
/* x [B,T,C], len_seq i32[B], draws [B*7] -> y [B,max_len_pad,C].  Optional outputs (may be NULL): i0 i32[B,P],
 * lam f32[B,P], counts i32[B] (un-truncated, model.py:418). */
int ss_interp_forward(ss_engine* e, const float* x_dev, const int* len_seq_dev, const float* scales_dev,
                      const int* len_seg_dev, int B, int T, int C, float* y_dev, int* i0_dev, float* lam_dev,
                      int* counts_dev, void* stream);
/* adjoint of the last ss_interp_forward: dy [B,P,C] -> dx [B,T,C] */
int ss_interp_backward(ss_engine* e, const float* dy_dev, int B, int T, int C, float* dx_dev, void* stream);

/* ---- offline feature extraction, the part that can be pinned without librosa / pysptk (make_spect_f0.py:57-71, utils.py:18-42) ----
 * wav: float64 [n] AFTER the host-side high-pass filtfilt and dither (make_spect_f0.py:53-54); mel_basis: float64 [513][n_mels]
 * (make_spect_f0.py:15 takes librosa's, transposed; here an input); out: float32 [ss_melspec_frames(n)][n_mels] =
 * (20 log10(max(1e-5, |STFT| . mel)) - 16 + 100) / 100 with the reference's reflect padding, periodic Hann window, 1024-point
 * transform and hop 256, computed in float64 as numpy does.  No engine needed. */
int ss_melspec_frames(int n);
int ss_melspec(const double* wav_dev, int n, const double* mel_basis_dev, int n_mels, float* out_dev, void* stream);
/* f0 float64 [n], -1e10 = unvoiced (RAPT's otype=2 convention, make_spect_f0.py:63-64) -> float32: (f0 - mean) / std / 4 clipped to
 * [-1, 1] and mapped to [0, 1] for voiced frames, mean / std over the voiced frames (utils.py:35-42). */
int ss_f0_normalize(const double* f0_dev, int n, float* out_dev, void* stream);

/* Batch producer on the GPU side (replaces the host loop of MyCollator.__call__, reference data_loader.py:101-128, for a
 * corpus kept resident in HBM): utterance b of the batch is rows [row0[b], row0[b] + len[b]) of the concatenated corpus
 * mel_cat [rows, n_mel] / f0_cat [rows]; the host only draws the crops (same generator calls, same order as the reference).
 * mel [B,T,n_mel] = clip(crop, 0, 1) zero-padded, f0 [B,T] = crop padded with -1e10, emb [B,emb_dim] = emb_tab[item[b]].
 * All pointers are device pointers; row0 is int64, len / item int32.  No engine needed. */
int ss_collate(const float* mel_cat_dev, const float* f0_cat_dev, const float* emb_tab_dev, const long* row0_dev,
               const int* len_dev, const int* item_dev, int B, int T, int n_mel, int emb_dim, float* mel_dev, float* f0_dev,
               float* emb_dev, void* stream);

/* Asynchronous failures.  The engine keeps a STATUS WORD that kernels set and no step clears:
 *   SS_STATUS_ABORT   a persistent recurrence kernel's bounded wait expired on this device (e.g. the GPU is shared and the
 *                     256-workgroup grid was not co-resident): that step's gradients are garbage;
 *   SS_STATUS_REMOTE  another data-parallel rank reported the same through the gradient arena's status slot (the last four
 *                     floats of the arena; the all-reduce sums it);
 *   SS_STATUS_RANGE   a parameter is not finite or reached |p| >= 2048, outside what the fixed scale of the WEIGHTS' fp16 x 2
 *                     split is valid for (activations and gradients carry data-dependent scales and have no such limit).
 * While it is non-zero the Adam kernel SKIPS the update on the device (parameters, moments and step counter untouched --
 * no host round trip is involved), every later ss_*_forward / ss_*_train_step / ss_adam_step returns an error without
 * enqueueing anything, and ss_check() keeps failing until ss_clear_abort().  ss_check synchronises `stream`;
 * ss_status() reads the word without synchronising.
 * LOCKSTEP mode (data parallel; switched on by ss_comm_init with world > 1, or by ss_set_lockstep for ranks that exchange their
 * gradients outside the engine): the entry points do NOT refuse -- the word is set asynchronously, ranks would notice it at different
 * iterations and the survivors would wait in a collective for ever.  Every rank keeps enqueueing (the device-side skip protects the
 * weights, the status slot carries the failure to every rank within the same step) and learns of it from ss_check(), which all
 * ranks call at the same iteration; all of them then either stop or call ss_clear_abort() and go on together. */
#define SS_STATUS_ABORT 1u
#define SS_STATUS_REMOTE 2u
#define SS_STATUS_RANGE 4u
int ss_check(ss_engine* e, void* stream);
unsigned ss_status(const ss_engine* e);
int ss_clear_abort(ss_engine* e, void* stream);
int ss_set_lockstep(ss_engine* e, int on);

/* ---- test / profiling hooks ---- */
/* C[M,N] = A . B^T style fp32 MFMA GEMM used by every contraction on the path (flags: 1 = A stored [K,M], 2 = B stored [K,N], 8 = bf16-rounded operands) */
int ss_op_gemm(const float* a_dev, long lda, const float* b_dev, long ldb, float* c_dev, long ldc, const float* bias_dev,
               int M, int N, int K, int flags, int ksplit, void* stream);
/* One bidirectional LSTM recurrence on haloed slabs (speechsplit_amd/csrc/kernels.h): gates [B,T+4,8H] holds
 * x.W_ih^T + b on entry and the activated gates on exit; out / csave [B,T+4,2H]; whh_* [4H,H].  H <= 32 runs the
 * single-launch kernel, H in {64,128,256,512} one launch per time step and needs scratch of at least
 * 8*H*H + 4*ceil16(B)*H floats (forward) / 8*H*H + 16*ceil16(B)*H + 2*B*H floats (backward).  H in {256,512} with
 * 2*ceil(B/16)*(H/16) <= 256 runs as ONE persistent launch (the engine's schedule) when, for the backward, scratch also
 * holds its exchange tiles and flags: 4*ceil(B/16)*(H/16)^2*1024 + 8192 bytes (the hook zeroes them: the tiles carry step tags). */
int ss_op_lstm_fwd(float* gates_dev, const float* whh_f_dev, const float* whh_b_dev, float* out_dev, float* csave_dev,
                   float* scratch_dev, long scratch_floats, int B, int T, int H, void* stream);
/* BPTT of the same: d_out [B,T+4,2H]; gates is replaced by the pre-activation gradients. */
int ss_op_lstm_bwd(float* gates_dev, const float* whh_f_dev, const float* whh_b_dev, const float* d_out_dev,
                   const float* csave_dev, float* scratch_dev, long scratch_floats, int B, int T, int H, void* stream);
/* Test hook for the fused weight / bias gradient kernel of the encoder BLSTMs (csrc/lstm_wgrad.hip; hidden size <= 32): from the pre-activation
 * gradients dg [R][8H], the layer input x [R][In] (row stride x_ld) and the layer output hout [R][2H] (haloed slabs flattened to R rows, halo
 * rows zero) ACCUMULATE dW_ih [2][4H][In], dW_hh [2][4H][H] and the bias gradients gb [2][2][4H] (b_ih, b_hh per direction).  scratch:
 * >= 16 * 4096 * tiles + 64 floats with tiles = ceil(8H / 64) * (ceil(In / 64) + (H >= 16 ? 1 : 2)). */
int ss_op_lstm_wgrad(const float* dg_dev, const float* x_dev, long x_ld, const float* hout_dev, float* gwih_dev, float* gwhh_dev, float* gb_dev,
                     float* scratch_dev, long scratch_floats, long R, int H, int In, void* stream);
/* The GEMM over operand images (speechsplit_amd/csrc/gemm_img.hip), the engine's default for every large contraction: an IMAGE has the
 * geometry of its fp32 matrix (4 bytes per element) with every aligned group of 8 elements along the contiguous axis replaced by 16 bytes of
 * hi pieces and 16 bytes of lo pieces of the fp16 x 2 split of scale * x (scale a power of two).
 *   ss_op_split_image: fp32 [rows][cols] (row stride ld) -> image (row stride ldi; cols % 8 == 0).
 *   ss_op_gemm_img:    C[M,N] (+)= sum_k A(m,k) B(n,k) (+ bias) over two images split with scale_a / scale_b.  flags: 1 = A stored [K,M],
 *                      2 = B stored [K,N], 4 = accumulate into C, 8 = single-piece form: both operands are plain bf16 matrices (2 bytes
 *                      per element, ld in elements, no scales; K % 64 == 0 for a K-contiguous operand) -- the 16-bit data path of
 *                      SS_PRECISION_BF16, where a slab stored in bf16 is its own image.  a_seglen / a_segstride: segmented K axis of a K-contiguous A (k = seg *
 *                      seglen + w lives at column seg * segstride + w: a k=5 convolution over a haloed slab), 0 = none.  ksplit > 1 needs
 *                      part_dev (ksplit * M * N floats of scratch; partial slabs, added in a fixed order).  zeros_dev: >= 1 KB of zero bytes
 *                      (needed when both operands are reduction-major and K % 32 != 0).  cfg: -1 = choose the tile, 0 = 256 x 256,
 *                      1 = 128 x 128, 2 = 256 x 128. */
int ss_op_split_image(const float* src_dev, long ld, long rows, int cols, float scale, float* img_dev, long ldi, void* stream);
int ss_op_gemm_img(const float* a_img_dev, long lda, const float* b_img_dev, long ldb, float* c_dev, long ldc, const float* bias_dev, int M, int N,
                   int K, int flags, int ksplit, int cfg, float scale_a, float scale_b, int a_seglen, long a_segstride, float* part_dev,
                   const void* zeros_dev, void* stream);
/* One relu(GroupNorm(ConvNorm(x))) block (model.py:61-67,76-77; 16 channels per group) through the engine's own block
 * routines, forward and -- when dy is given -- backward.  x [B,T,Ci], w [Co,Ci,5], bias/gamma/beta [Co], y/dy [B,T,Co],
 * dx [B,T,Ci] (nullable), gw [Co,Ci,5], gb/ggamma/gbeta [Co]; scratch of ss_op_conv_block_scratch() floats. */
long ss_op_conv_block_scratch(int B, int T, int Ci, int Co);
int ss_op_conv_block(const float* x_dev, const float* w_dev, const float* bias_dev, const float* gamma_dev, const float* beta_dev,
                     const float* dy_dev, float* y_dev, float* dx_dev, float* gw_dev, float* gb_dev, float* ggamma_dev,
                     float* gbeta_dev, float* scratch_dev, long scratch_floats, int B, int T, int Ci, int Co, void* stream);
/* The ReLU branch the engine took in conv block `block` ("enc1.c1_0" .. "enc1.c2_2", "enc3.c_0" .. "enc3.c_2", "enc2.c") of
 * the last forward: mask [B,T,Co] dense, 1.0f where the GroupNorm output is > 0.  A GroupNorm output within fp32 rounding
 * of 0 may fall on either side in two correct implementations; parity tests hand this mask to the oracle so that the
 * comparison of gradients does not depend on that coin flip. */
int ss_debug_relu_mask(ss_engine* e, const char* block, float* mask_dev, void* stream);
/* tuning knobs (process-global; every value leaves the results correct): "lstm_nw" 4|8|16, "lstm_g" 0..16, "gemm_bk" 16|32,
 * "gemm_want" >= 1, "overlap" 0|1, "persist" 0|1, "split" 0|1, "gemm_mode" 0 (fp32 MFMA) | 1 (split arithmetic on
 * the 16-bit pipe), "fwd_f16x2" / "bwd_f16x2" 0|1 (fp16 x 2 instead of bf16 x 3 for the forward / the scaled gradient
 * contractions), "seq_spin_log2" 0..24 (log2 of the persistent kernels' bounded wait; 0 makes it expire at once -- how the
 * tests exercise the abort path), "deterministic" 0|1, "seq_tag" 0|1 (forward persistent recurrence: step tag in the hand-off
 * payload where every group sits on one XCD | always the flag line), "seq_wlead" 0..31 (backward persistent recurrence: steps
 * between a warm-up read and the operand request it serves, 0 = the kernel's default), "seq_var" 0|2|4 (... form of those warm-up
 * reads: whole 1 KB runs | one dword per 128-byte line (default) | one per 64 bytes; results bit-identical), "gemm_ws" 0|1|2 (wave-specialised form of
 * the 128 x 128 fp16 x 2 GEMM: never | where it measured faster in isolation | always), "img" 0|1 and "img_mask" (bit = SS_PROF_* class: which contractions run on the image GEMM, csrc/gemm_img.hip; default decoder projections, conv forward, conv input gradients), "dp_model" 0|2..64 and "dp_buckets" 0|1 (data-parallel schedule, see ss_g3_dp_train_step), "presplit" 0..15 (operand images for the fp16 x 2 GEMMs: bit 0 weights, bit 1 decoder hidden states, bit 2 trunk
 * activations), "compact0" 0|1 (decoder layer 0 on one row per block of repeated input frames), "trunk_indep" 0|1, "batch_dirs" 0..2,
 * "prewarm" 0..3 (streaming pre-read of a decoder layer's operand slabs on a side stream beside its
 * persistent recurrence: bit 1 forward, bit 0 backward), "op_time_major" 0|1 (ss_op_lstm_fwd / _bwd
 * read their slabs as [T+4,B,C]; persistent kernels only -- a layout experiment); schedule of the end of the backward (round 4,
 * tools/real_timeline.py): "dec_tail_split" 0..6 (which stream takes the decoder's layer-0 / layer-1 and the head's weight gradients; default 2),
 * "enc_t_first" 0|1, "conv_dw_off" 0|1 (Generator_6: conv weight gradients off the trunk's dependent chain), "early_dw" 0|1 (16-bit mode: a decoder
 * layer's weight gradients beside the next backward recurrence, off).  The timing experiments that produce WRONG results ("lstm_mode",
 * "gemm_diag", "seq_prio" > 1) are compiled out of this library; `make -C speechsplit_amd/csrc diag` builds
 * libspeechsplit_hip_diag.so with them for tools/ (never loaded by the package unless SS_DIAG_LIB=1 is set). */
int ss_tune(const char* key, int value);
/* Arithmetic of the contractions (convolutions, LSTM input projections, all weight / input gradients, head).
 * SS_PRECISION_F32 (default) -- the 1e-4 parity mode.  Operands, accumulation and storage are fp32; every PRODUCT is formed
 *   on the 16-bit matrix pipe from a split of both fp32 operands:
 *     fp16 x 2 (default wherever an operand's magnitude is known): y = s*x = h + l, h = fp16(y), l = fp16(y - h), s a power of
 *       two; 3 v_mfma_f32_32x32x16_f16 per k-step (h.l, l.h, h.h): 22 significand bits relative to the operand's scaled
 *       maximum (elements more than 2^22 below it lose relative precision; the residual goes subnormal 2^-3 below the
 *       maximum's binade and flushes at 2^-33 of it).  Scales: WEIGHTS, hidden states (|h| < 1) and the network inputs use the fixed
 *       s = 16 (|x| < 4094 survives); the conv blocks' outputs use min(16, the power of two that keeps sqrt(16 T) max|gamma| +
 *       max|beta| inside fp16) computed on the device from their GroupNorm affine every step (a gamma of 100 or 1000 trains in this
 *       mode, at the same 1e-4 parity); gradient operands the power of two for the maximum their producer kernel measured.  What is
 *       refused -- status SS_STATUS_RANGE, the Adam update skipped on the device -- is a parameter that is not finite or reaches
 *       |p| >= 2048, checked inside every forward (there is no silent overflow and no automatic change of split).
 *     bf16 x 3 (head, encoder BLSTMs, unaligned shapes; everything with ss_tune("fwd_f16x2" / "bwd_f16x2", 0)): exact
 *       3-way truncation split x = h + m + l, 6 v_mfma_f32_32x32x16_bf16 per k-step, dropped terms <= 2^-24 relative.
 *     ss_tune("gemm_mode", 0): true fp32 MFMA (v_mfma_f32_32x32x2_f32), the A/B reference.
 *   Against fp64 the three measure 1.3-2.2e-6, 1.2e-6 and 1.3e-6 of max|C| (profiles/r02/f16x2_error.txt).
 *   ss_tune("gemm_mode", 0) together with ss_tune("persist", 0) (the decoder recurrences as one fp32-MFMA launch per time step) is the mode in
 *   which EVERY product of the step is fp32-wide -- the reference's arithmetic; bench.py times it as alt_precisions.all_fp32_mfma.
 * SS_PRECISION_BF16 (BASELINE configs 3-5) -- the 16-bit data path (round 4).  Whoever produces an operand of a large contraction also stores
 *   it as a plain bf16 tensor of the same geometry (round to nearest even), and the contraction runs from there on the single-piece form of
 *   the image GEMM (csrc/gemm_img.hip: one v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation, no scales): packed conv weights and
 *   stacked W_ih (per-step re-layouts), decoder hidden states (the forward recurrence's storing wave), resampled trunk activations (the
 *   fused GroupNorm + gather), pre-activation gradients (the backward recurrence's storing wave), conv-output gradients (the GroupNorm
 *   backward).  The persistent recurrences multiply the HIGH fp16 pieces of h and W_hh only (one v_mfma_f32_16x16x32_f16 per product, half the
 *   hand-off payload, 11 significand bits per operand).  fp32 throughout: master weights, gradients' accumulation and the gradient arena,
 *   cell state, gate pre-activations, GroupNorm statistics, losses, the resampling index path (bit-exact as in the fp32 mode), Adam.
 *   Contractions without images (layer-0 convolutions over the 80 / 264-channel inputs, the decoder's 164-column layer-0 input, the head's
 *   weights, the encoder BLSTM projections) round their fp32 operands to bf16 inside round 2's GEMM kernel; the encoder BLSTMs' weight
 *   gradients are exact fp32 sums in both modes (csrc/lstm_wgrad.hip).  ss_tune("bf16_img", 0) / ("seq_hi", 0) select round 3's form of the
 *   mode (fp32 slabs only, operands rounded inside the GEMM, fp16 x 2 recurrences) for A/B runs. */
#define SS_PRECISION_F32 0
#define SS_PRECISION_BF16 1
int ss_set_precision(ss_engine* e, int precision);
/* Live timing of the step's kernels inside a caller's own timed region.  ss_profile(e, mask) makes the engine bracket every
 * launch of the classes whose bit (1u << class) is set with hipEvents on the stream the launch goes to (a non-zero mask clears
 * the record, 0 stops recording; at most 8192 launches are kept).  A bracket costs 4-8 us of stream time (the launch behind it cannot
 * start before the one in front has retired and signalled): all 71 per step are +4.5 % on the step, the dominant class + the recurrences
 * (18 per step) +2.9 % (round 3; the step has become shorter, the brackets have not).  ss_profile_sample(e, n) therefore brackets only
 * every n-th training step (n = 1: every step; the count starts again with every ss_profile call).  ss_profile_read synchronises on the recorded events of one class and returns their count, their summed duration
 * and their summed algorithmic FLOPs (2*M*N*K per GEMM; 2 * 2 directions * B * T * 4H * H per recurrence launch). */
#define SS_PROF_DEC_PROJ 0  /* decoder input projection, layers >= 1 (NT, M = B*T, N = 8H, K = 2H) */
#define SS_PROF_DEC_PROJ0 1 /* decoder input projection, layer 0 (K = 164 / 66) */
#define SS_PROF_DEC_DW 2    /* decoder weight gradients dW_ih, dW_hh (TN, split-K) */
#define SS_PROF_DEC_DX 3    /* decoder input gradients (NN) */
#define SS_PROF_CONV_FWD 4  /* conv trunk forward (segmented-K NT) */
#define SS_PROF_CONV_DW 5   /* conv weight gradients (TN) */
#define SS_PROF_CONV_DX 6   /* conv input gradients */
#define SS_PROF_REC_FWD 7   /* persistent decoder recurrence, forward (one launch per layer) */
#define SS_PROF_REC_BWD 8   /* persistent decoder recurrence, backward */
#define SS_PROF_ENC_LSTM 9  /* encoder BLSTM projections / gradients (small GEMMs) */
#define SS_PROF_HEAD 10     /* LinearNorm head forward / gradients */
#define SS_PROF_CLASSES 11
/* timeline-only classes (no flops; ss_profile_timeline / tools/real_timeline.py): the non-GEMM launches on the step's dependent chains */
#define SS_PROF_ENC_REC 11  /* encoder BLSTM recurrences (one launch per layer and pass) */
#define SS_PROF_GN 12       /* GroupNorm + ReLU forward / gather / backward */
#define SS_PROF_WGRAD 13    /* fused encoder-BLSTM weight gradients */
#define SS_PROF_ADAM 14     /* optimiser launches */
#define SS_PROF_PREP 15     /* per-step weight re-layouts */
int ss_profile(ss_engine* e, unsigned class_mask);
int ss_profile_sample(ss_engine* e, int every_nth_step);
int ss_profile_read(ss_engine* e, int klass, int* launches, double* total_us, double* total_flops);
/* The same brackets as a timeline: out[3 * i .. 3 * i + 2] = (class + 100 x stream [0 main, 1 side, 2 / 3 branch streams], start, end) of record i in enqueue order, microseconds relative to the
 * first record's start; returns the number of records written (<= cap), negative on error.  (tools/real_timeline.py) */
int ss_profile_timeline(ss_engine* e, double* out, int cap);
/* timing experiment: with ss_tune("gemm_diag", 16) the 128x128 NT bf16x3 GEMM accumulates, for its first 64 workgroups, the
 * s_memtime ticks every wave spends per k-loop phase; out24 = [4 waves][split+store, barrier, load issue, fragments+MFMA,
 * barrier, k-tiles (wave 0 only)].  Synchronises the device. */
int ss_debug_gemm_phases(unsigned long long* out24, int reset);
/* Placement probe: n_wg workgroups (threads, lds_bytes each) write their (XCC_ID, HW_ID) hardware registers to out_dev[2 * n_wg] in launch
 * order and stay resident for hold_us microseconds.  (How workgroup indices map to XCDs is what the persistent recurrences' group slots
 * and the image GEMM's XCD filter rely on: tools/xcd_overlap_probe.py.) */
/* Placement log of ss_op_gemm_img's last work-queue launch made with ss_tune("img_xcc", mask | 0x100): words [0] tiles handed out,
 * [1] started flag, then per workgroup of the launch (from word 4) XCD (| 0x100: left without work), tiles taken, first and last tick (100 MHz). */
int ss_debug_img_wq(unsigned* out_host, int words);
int ss_debug_xcc_map(int n_wg, int threads, int lds_bytes, int hold_us, unsigned* out_dev, void* stream);
/* named internal slab of the last call ("enc1.xf2", "dec.out2", ...); layout [B, T+4, C], frame t at row t+2 */
int ss_debug_buffer(ss_engine* e, const char* name, float** ptr_dev, long* rows, long* cols);
int ss_debug_names(ss_engine* e, char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
