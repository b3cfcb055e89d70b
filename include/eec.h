/* eec.h -- C ABI of the MI355X-native early-exit Conformer encoder (libeec.so).
 *
 * The reference (augustgw/early-exit-transformer) has no FFI / plugin registry: its
 * drop-in boundary is the Python class `Early_conformer` / `full_conformer`
 * (models/model/early_exit.py:565-634, 637-800) as called from train.py:54,37 and
 * inference.py:66,45.  This library sits behind this repo's own nn.Module mirror of
 * that class (early_exit_transformer_amd/model.py); every entry point below states the
 * reference code it replaces.  Plain pointers and sizes only: all `const float*`,
 * `void* workspace` etc. are DEVICE pointers (HIP), `stream` is a hipStream_t passed as
 * void*.  Every function returns 0 on success, a non-zero hipError_t / EEC_ERR_* code
 * otherwise; eec_last_error() returns a thread-local message.  No entry point allocates,
 * frees or synchronises the device except create/destroy/pack.
 *
 * Devices: an eec_encoder handle belongs to the HIP device that was current in eec_encoder_create (its packed-weight
 * arena is a plain allocation on that device).  Every later call on the handle must be made with the same device
 * current and with parameters / inputs / workspace on it, else EEC_ERR_BAD_ARG.  The intended deployment is one
 * process per GPU (torch.distributed over RCCL); a process that does drive several devices creates one handle per
 * device (launch attributes are tracked per device).  Handles are not thread-safe; use one per thread or lock.
 */
#ifndef EEC_H_
#define EEC_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define EEC_ABI_VERSION 16
#define EEC_ERR_BAD_ARG 10001
#define EEC_ERR_UNSUPPORTED 10002
#define EEC_ERR_WORKSPACE 10003
#define EEC_ERR_NOT_PACKED 10004

/* Operand precision of the MFMA products (accumulation, residual stream, LayerNorm,
 * softmax and log-softmax are always fp32).  The reference computes everything in fp32. */
enum {
  EEC_PREC_F16X3 = 0, /* hi/lo-split fp16, 3 MFMA passes per GEMM: |dlogp| ~2e-4, parity mode   */
  EEC_PREC_MIXED = 1, /* feed-forward GEMMs single-pass fp16, all others split: |dlogp| ~1e-3  */
  EEC_PREC_F16 = 2,   /* every GEMM single-pass fp16: |dlogp| ~3e-3                            */
  EEC_PREC_F16F8 = 3  /* as F16X3, but the feed-forward GEMMs compute the two hi/lo correction products
                         with block-scaled fp8 (e5m2) MFMAs at twice the fp16 rate: |dlogp| ~3e-4 */
};

/* Constructor kwargs of Early_conformer that shape the encoder (early_exit.py:567-615;
 * flags util/conf.py --d_model --n_heads --d_feed_forward --depthwise_kernel_size
 * --n_enc_exits --n_enc_layers_per_exit --n_mels, dec_voc_size, --max_len). */
typedef struct eec_config {
  int32_t d_model;         /* 256 or 512 (64-row / 32-row tile geometry) */
  int32_t n_heads;         /* d_model / n_heads in {32, 64} */
  int32_t d_ff;            /* multiple of 32 */
  int32_t dw_kernel;       /* odd, <= 31 */
  int32_t n_exits;         /* E */
  int32_t layers_per_exit; /* L */
  int32_t n_mels;          /* features_length: 3*n_mels a multiple of 16, <= 384 (80 -> 240) */
  int32_t vocab;           /* dec_voc_size: multiple of 32, <= 256 */
  int32_t max_len;         /* rows of the positional-encoding table */
  int32_t arch;            /* EEC_ARCH_CONFORMER (Early_conformer / full_conformer) or EEC_ARCH_LEGACY (Early_encoder) */
} eec_config;

#define EEC_ARCH_CONFORMER 0
#define EEC_ARCH_LEGACY 1

/* fp32 parameters of one torchaudio ConformerLayer, by state_dict key suffix. */
typedef struct eec_layer_params {
  const float *ffn1_ln_w, *ffn1_ln_b;   /* ffn1.sequential.0.{weight,bias}          [D]      */
  const float *ffn1_w1, *ffn1_b1;       /* ffn1.sequential.1.{weight,bias}          [F,D],[F]*/
  const float *ffn1_w2, *ffn1_b2;       /* ffn1.sequential.4.{weight,bias}          [D,F],[D]*/
  const float *attn_ln_w, *attn_ln_b;   /* self_attn_layer_norm.{weight,bias}                */
  const float *attn_in_w, *attn_in_b;   /* self_attn.in_proj_{weight,bias}          [3D,D]   */
  const float *attn_out_w, *attn_out_b; /* self_attn.out_proj.{weight,bias}         [D,D]    */
  const float *conv_ln_w, *conv_ln_b;   /* conv_module.layer_norm.{weight,bias}              */
  const float *conv_pw1_w, *conv_pw1_b; /* conv_module.sequential.0.{weight,bias}   [2D,D,1] */
  const float *conv_dw_w, *conv_dw_b;   /* conv_module.sequential.2.{weight,bias}   [D,1,K]  */
  const float *conv_bn_w, *conv_bn_b;   /* conv_module.sequential.3.{weight,bias}            */
  const float *conv_bn_rm, *conv_bn_rv; /* conv_module.sequential.3.running_{mean,var}       */
  const float *conv_pw2_w, *conv_pw2_b; /* conv_module.sequential.5.{weight,bias}   [D,D,1]  */
  const float *ffn2_ln_w, *ffn2_ln_b, *ffn2_w1, *ffn2_b1, *ffn2_w2, *ffn2_b2; /* ffn2.sequential.* */
  const float *final_ln_w, *final_ln_b; /* final_layer_norm.{weight,bias}                    */
} eec_layer_params;

typedef struct eec_params {
  const float *sub0_w, *sub0_b; /* conv_subsample.sequential.0  [D, n_mels, 3], [D] */
  const float *sub1_w, *sub1_b; /* conv_subsample.sequential.1  [D, D, 3], [D]      */
  const float* pe;              /* positional_encoder.pe        [max_len, 1, D]     */
  const eec_layer_params* layers; /* HOST array of n_exits*layers_per_exit entries, exit-major */
  const float* const* head_w;   /* HOST array of n_exits device pointers: linears.e.weight [V, D] */
  const float* const* head_b;   /* HOST array of n_exits device pointers: linears.e.bias   [V]    */
} eec_params;

/* fp32 parameters of one legacy pre-norm transformer layer, models/blocks/encoder_layer.py:14-44 with
 * models/layers/multi_head_attention.py:11-29 (separate w_q/w_k/w_v/w_concat Linears) and
 * models/layers/position_wise_feed_forward.py:9-23 (Linear -> ReLU -> Linear); SURVEY.md 8a row a14. */
typedef struct eec_legacy_layer_params {
  const float *norm1_w, *norm1_b;                     /* norm1.{weight,bias}                 */
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo; /* attention.w_{q,k,v,concat}.{weight,bias} [D,D],[D] */
  const float *norm2_w, *norm2_b;                     /* norm2.{weight,bias}                 */
  const float *w1, *b1, *w2, *b2;                     /* ffn.linear{1,2}.{weight,bias} [F,D],[F],[D,F],[D] */
} eec_legacy_layer_params;

/* Early_encoder (models/model/early_exit.py:497-562): stem, PE, E x Encoder(L layers + layer_norm), E heads. */
typedef struct eec_legacy_params {
  const float *sub0_w, *sub0_b, *sub1_w, *sub1_b, *pe;
  const eec_legacy_layer_params* layers; /* HOST array, n_exits*layers_per_exit, exit-major: encoders.e.layers.l */
  const float* const* group_ln_w;        /* HOST arrays of n_exits device pointers: encoders.e.layer_norm.{weight,bias} */
  const float* const* group_ln_b;
  const float* const* head_w;            /* linears.e.{weight,bias} */
  const float* const* head_b;
} eec_legacy_params;

typedef struct eec_encoder eec_encoder;

const char* eec_last_error(void);
int eec_abi_version(void);

/* T' = ((T-3)/2+1 - 3)/2 + 1 : frames after the two stride-2 convs (early_exit.py:24-48). */
int eec_out_frames(int T);

/* The length -> key-mask arithmetic of Early_conformer.forward (early_exit.py:623) on its own:
 *   enc_len[b] = int32( min( float(lengths[b]) / 4, float(T') ) )      (true division in fp32, clamp, truncation)
 * keys t >= enc_len[b] are masked in every attention of the stack (torchaudio _lengths_to_padding_mask).  The forward
 * entry points compute it internally; this entry exposes the integers (tests compare them bit for bit). */
int eec_encoder_lengths(const int64_t* lengths, int B, int Tq, int32_t* enc_len, void* stream);
/* A small HOST int64 array (the `lengths` the reference's collate hands to forward() as a CPU tensor, train.py:34,54) -> device
 * memory through a kernel's argument block: stream-ordered like a copy, but no DMA and no cross-queue dependency in front of the
 * forward.  n <= eec_upload_i64_max() (480); the host array is read before the call returns. */
int eec_upload_i64_max(void);
int eec_upload_i64(const int64_t* host, int n, int64_t* dev, void* stream);


/* Replaces Early_conformer.__init__ (early_exit.py:567-615) for the encoder stack. */
int eec_encoder_create(const eec_config* cfg, eec_encoder** out);
void eec_encoder_destroy(eec_encoder* enc);

/* Re-packs the fp32 parameters (fp16 hi/lo MFMA fragments, BatchNorm folded into the depthwise
 * taps, conv weights transposed).  Call after load_state_dict / an optimizer step; eval-mode
 * BatchNorm semantics (running statistics).  Asynchronous on `stream`. */
int eec_encoder_pack(eec_encoder* enc, const eec_params* params, void* stream);

/* Same for an EEC_ARCH_LEGACY encoder (no mask, no convolution module; `lengths` of eec_encoder_forward is ignored:
 * Early_encoder.forward(src) passes mask=None, early_exit.py:549-554). */
int eec_encoder_pack_legacy(eec_encoder* enc, const eec_legacy_params* params, void* stream);

size_t eec_encoder_workspace_bytes(const eec_encoder* enc, int B, int T);

/* Replaces Early_conformer.forward(src, lengths) (early_exit.py:617-634) in eval mode:
 *   mel      [B, n_mels, T] fp32        lengths [B] int64 (device copy of the caller's tensor)
 *   out      [E, B, T', V] fp32 log-probabilities (written in place, no torch.cat)
 *   taps_opt [E, B, T', D] fp32 or NULL: pre-head activations after each exit group
 *            (what full_conformer._encoder_(src, lengths, n) returns, early_exit.py:719-737)
 *   stop_after: <0 = run everything with the production launch plan (Conformer: 3 launches per layer, the
 *            row-tile-local steps fused into one "chain" kernel); otherwise stop after that many
 *            sub-steps (0 = stem, then per layer: ffn1, attention, conv, ffn2; legacy: attention, ffn), each sub-step
 *            its own launch -- test hook; the current residual stream is then left in x_dbg_opt [B*T', D] if given. */
int eec_encoder_forward(eec_encoder* enc, const float* mel, const int64_t* lengths, int B, int T,
                        int precision, float* out, float* taps_opt, void* workspace, size_t workspace_bytes,
                        int stop_after, float* x_dbg_opt, void* stream);

/* Early exit: the same forward, stopped after the first n_groups exit groups (1 .. E) -- what the reference does with
 * full_conformer._encoder_(src, lengths, layer_n) (early_exit.py:719-737: the loop breaks after layer_n groups) and what
 * an early-exit deployment runs when it decodes from exit n_groups.  Production launch plan; cost is n_groups / E of a
 * full forward.
 *   out_opt   [n_groups, B, T', V] log-probs of the exits that were run, or NULL
 *   taps_opt  [n_groups, B, T', D] or NULL          x_out_opt [B, T', D]: encoder output after group n_groups, or NULL
 * At least one of the three must be given. */
int eec_encoder_forward_prefix(eec_encoder* enc, const float* mel, const int64_t* lengths, int B, int T, int precision,
                               int n_groups, float* out_opt, float* taps_opt, float* x_out_opt, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Building blocks for the reference's other encoder topologies built from the same Conformer groups (Splitformer,
 * early_exit.py:227-364: down-sampled parallel branches added to the main path at the first and last exit).
 * eec_encoder_pack may be called with the stem (sub0_w .. pe) and / or the heads (head_w, head_b) left NULL for an
 * encoder that is used through these entry points only.
 *   eec_encoder_group_forward: x [B, T', D] fp32, in place, through the layers_per_exit Conformer layers of `group`
 *       (= one torchaudio Conformer.forward(x, lengths), early_exit.py:603-615,627); key_len [B] int32 on the device:
 *       keys >= key_len[b] are masked (the caller applies the reference's length rule); production launch plan.
 *   eec_encoder_head_forward:  out [M, V] = log_softmax(x [M, D] . W_exit^T + b_exit)   (early_exit.py:629-631).
 *   eec_encoder_stem1_forward: x [B, T1, D] = Conv1d(k=3, s=2)(mel) + bias + pe[t1], T1 = (T - 3) / 2 + 1: the
 *       one-convolution stem of Early_zipformer (Conv1dSubampling_Zipformer, early_exit.py:80-95, 175-176); needs
 *       sub0_w, sub0_b and pe at pack time (sub1_* may be NULL). */
size_t eec_encoder_group_workspace_bytes(const eec_encoder* enc, int B, int Tq);
int eec_encoder_stem1_forward(eec_encoder* enc, const float* mel, int B, int T, float* x, void* stream);
int eec_encoder_group_forward(eec_encoder* enc, int group, float* x, const int32_t* key_len, int B, int Tq, int precision,
                              void* workspace, size_t workspace_bytes, void* stream);
int eec_encoder_head_forward(eec_encoder* enc, int exit, const float* x, int M, float* out, int precision, void* stream);

/* Measurement hook (no reference counterpart; the reference has no profiler hooks):
 * when enabled, every kernel launch of eec_encoder_forward is bracketed by hipEventRecord on the
 * launch stream; eec_encoder_profile_read synchronises the recorded events and returns the summed
 * milliseconds and launch counts per kernel class, index = EEC_KC_* (EEC_KC_CHAIN: the fused chain kernel of the
 * production plan; EEC_KC_FFN / _QKV / _DW_PW2 / _PROJ: the same steps as separate launches of the sub-step plan). */
enum { EEC_KC_STEM = 0, EEC_KC_FFN, EEC_KC_QKV, EEC_KC_ATTN, EEC_KC_PROJ_GLU, EEC_KC_PROJ, EEC_KC_DW_PW2, EEC_KC_HEAD, EEC_KC_CHAIN, EEC_KC_COUNT };
int eec_encoder_set_profiling(eec_encoder* enc, int enable, int max_launches);
int eec_encoder_profile_read(eec_encoder* enc, double* ms_by_class, long long* launches_by_class, int n_classes);

/* Replaces GreedyCTCDecoder.forward (util/beam_infer.py:9-24), batched over n_seq sequences:
 *   logp [n_seq, Tq, V] fp32 -> tokens [n_seq, Tq] int32 (first counts[s] entries valid), counts [n_seq]. */
int eec_greedy_ctc(const float* logp, int n_seq, int Tq, int V, int blank, int32_t* tokens, int32_t* counts,
                   void* stream);

/* Replaces the per-exit loss loop of the CTC training/eval step (train.py:53-65 with
 * nn.CTCLoss(blank, reduction='mean', zero_infinity=True), train.py:259), all exits and utterances in one launch:
 *   logp [E, B, T', V] fp32 log-probs (the encoder output as is; input length = T' for every utterance)
 *   targets [B, S] int64, target_len [B] int64 (device copies of the caller's tensors; len <= 255)
 *   nll_scratch [E*B] fp32, loss_per_exit [E] fp32 = batch mean of nll / max(len, 1); train.py's loss = their sum.
 * Input checking (nn.CTCLoss raises on these; a kernel cannot): a target_len outside [0, S] or a label outside [0, V)
 * among an utterance's first target_len labels makes that utterance's nll -- and the exit's loss -- NaN; NaN log-probs
 * propagate as NaN; only +inf (an infeasible alignment) is zeroed, as zero_infinity=True does. */
int eec_ctc_loss(const float* logp, const int64_t* targets, const int64_t* target_len, int E, int B, int Tq, int V, int S,
                 int blank, float* nll_scratch, float* loss_per_exit, void* stream);

/* Backward of the per-exit loss loop (train.py:60-68: loss.backward() through E nn.CTCLoss calls), first slice of the
 * training path: gradient with respect to the encoder's log-prob output.
 *   eec_ctc_loss_forward: as eec_ctc_loss, and additionally keeps every step's forward variables in `bwd_workspace`
 *       (eec_ctc_backward_workspace_bytes(E, B, T', S) bytes, 256-byte aligned device memory; nll [E*B] is an output
 *       the backward needs again).
 *   eec_ctc_loss_backward: dlogp [E, B, T', V] = d( sum_e grad_loss[e] * loss_e ) / d logp, what torch autograd returns
 *       for the same loop (for each lattice grad * (exp(logp) - state posteriors), zero for an infeasible lattice);
 *       consumes bwd_workspace (call once per forward).
 *   eec_logsoftmax_backward: grad_logits = grad_logp - exp(logp) * sum_c grad_logp  (rows of V <= 256 entries): the
 *       log-softmax half of the exit heads' backward (early_exit.py:629-631); the two plain GEMMs of the Linear's
 *       backward (dW = grad_logits^T . x, dx = grad_logits . W) are library GEMMs on the caller's side. */
size_t eec_ctc_backward_workspace_bytes(int E, int B, int Tq, int S);
int eec_ctc_loss_forward(const float* logp, const int64_t* targets, const int64_t* target_len, int E, int B, int Tq, int V, int S,
                         int blank, float* nll, float* loss_per_exit, void* bwd_workspace, void* stream);
int eec_ctc_loss_backward(const float* logp, const int64_t* targets, const int64_t* target_len, int E, int B, int Tq, int V, int S,
                          int blank, const float* nll, void* bwd_workspace, const float* grad_loss, float* dlogp, void* stream);
int eec_logsoftmax_backward(const float* logp, const float* grad_logp, int M, int V, float* grad_logits, void* stream);

/* CTC prefix beam search: replaces BeamInference.ctc_cuda_predict (util/beam_infer.py:79-80,102-112:
 * torchaudio cuda_ctc_decoder(tokens, nbest=1, beam_size=10, blank_skip_threshold=0.95) on the log-probs of one exit,
 * input length T' for every utterance), batched over n_seq sequences.  That decoder is third-party CUDA code outside the
 * reference tree: this is the published algorithm (prefix beam search without a language model; oracle/ctc_beam_ref.py),
 * parity with torchaudio's tie-breaking is unpinned.
 *   logp [n_seq, T', V] fp32 log-probs, blank label `blank` (0 in the reference), V <= 256, beam_size <= 16
 *   blank_skip_threshold in (0, 1): a frame with p(blank) above it is taken as a blank frame without expansion; >= 1 disables
 *   workspace: eec_ctc_beam_workspace_bytes(n_seq, T') bytes (back-pointers)
 *   tokens [n_seq, T'] int32 (first counts[s] valid), counts [n_seq], scores [n_seq] = log p of the best prefix. */
size_t eec_ctc_beam_workspace_bytes(int n_seq, int Tq);
int eec_ctc_beam_decode(const float* logp, int n_seq, int Tq, int V, int blank, int beam_size, float blank_skip_threshold,
                        void* workspace, int32_t* tokens, int32_t* counts, float* scores, void* stream);
/* What torchaudio's CUDA decoder does with a frame above blank_skip_threshold is not visible from the reference (third-party,
 * absent): eec_ctc_beam_decode takes the frame as a BLANK frame (every prefix's mass moves to "ending in blank": a label
 * repeated across the frame stays a repeat, "a _ a" -> "aa").  skip_drops_frame != 0 selects the other reading: the frame is
 * DROPPED, as if the sequence were one frame shorter (the repeat collapses, "a _ a" -> "a"; scores exclude the frame).  Both
 * are tested against the CPU statement (oracle/ctc_beam_ref.py); the default stays the first until a torchaudio vector pins it. */
int eec_ctc_beam_decode_ex(const float* logp, int n_seq, int Tq, int V, int blank, int beam_size, float blank_skip_threshold,
                           int skip_drops_frame, void* workspace, int32_t* tokens, int32_t* counts, float* scores, void* stream);

/* Mel front end: replaces util/data_loader.py:7-18 -- torchaudio Spectrogram(n_fft = 2 * args.n_fft = 1024,
 * hop_length 160, win_length 320; hann window, power 2, centred frames with reflect padding) followed by MelScale(sample_rate,
 * n_mels, n_stft = 513; htk scale, no normalisation), NO log -- on the device, as an exact-fp32 MFMA transform.
 *   wave [B, Lmax] fp32; lengths_opt [B] int64 valid samples per utterance or NULL (all Lmax)
 *   mel  [B, n_mels, 1 + Lmax / 160] fp32: utterance b fills its first 1 + lengths[b] / 160 frames (torch.stft, center=True),
 *        the rest is zero -- what the reference's collate (pad_sequence with 0) hands to the model.
 * The tables (window, DFT basis, filterbank) are built in eec_frontend_create; only the reference's geometry is served. */
typedef struct eec_frontend eec_frontend;
const char* eec_frontend_last_error(void);
int eec_frontend_create(int sample_rate, int n_fft, int win_length, int hop_length, int n_mels, eec_frontend** out);
void eec_frontend_destroy(eec_frontend* fe);
int eec_frontend_frames(int n_samples, int hop_length);
int eec_frontend_forward(eec_frontend* fe, const float* wave, const int64_t* lengths_opt, int B, int Lmax, float* mel, void* stream);

/* ---- Training step of the Early_conformer path (train.py:53-70) -------------------------------------------------
 * `enc_out = model(batch_0, valid_lengths)` in train mode and `loss.backward()` through it.  Parameters are read in place
 * as fp32 (eec_params: the nn.Parameter storages themselves, no packing); activations stay fp32 in HBM; GEMM operands are
 * split into bf16 hi / lo planes on the fly (`passes` 3: three MFMA products per GEMM, ~1e-5 relative; 1: plain bf16).
 * Semantics of the reference's train mode: BatchNorm1d normalises with the statistics of the batch (all B*T' frames,
 * padded ones included) -- `bn_batch_stats` [E*L][2][D] returns (mean, biased variance) per layer so the caller can update
 * running_mean / running_var (momentum 0.1, unbiased variance) --, dropout with probability `drop_prob` at the reference's
 * sites (after the positional encoding, inside and after each feed-forward module, on the attention probabilities, after
 * out_proj, after the convolution module) from a counter-based generator keyed by `seed` (its streams cannot match
 * torch's: parity with the reference is at drop_prob 0).
 * eec_train_forward records the activations the backward needs in `workspace` (eec_trainer_workspace_bytes; the caller
 * keeps it untouched until eec_train_backward); one recorded forward per trainer at a time.
 * eec_train_backward: `out` = the log-probs eec_train_forward returned, `grad_out` = dLoss/d out [E,B,T',V]; `grads` is
 * an eec_params whose pointers are WRITTEN (overwritten, not accumulated) with the gradient of the parameter at the same
 * position (pe / running_mean / running_var entries are ignored).  Gradient with respect to `mel` is not produced.
 * `taps` (optional, [E][B*T'][D]) returns the group outputs the heads read -- what full_conformer feeds its attention decoders
 * (early_exit.py:764-800); `grad_taps` (optional, same shape) is the gradient that arrived at them from outside the path.
 * A trainer is bound to the device that is current in its first eec_train_forward (eec_trainer_workspace_bytes is host
 * arithmetic and needs none), is not thread-safe, and holds ONE recorded forward at a time. */
typedef struct eec_trainer eec_trainer;
const char* eec_trainer_last_error(void);
int eec_trainer_create(const eec_config* cfg, eec_trainer** out);
void eec_trainer_destroy(eec_trainer* tr);
size_t eec_trainer_workspace_bytes(const eec_trainer* tr, int B, int T);
int eec_train_forward(eec_trainer* tr, const eec_params* params, const float* mel, const int64_t* lengths, int B, int T, int passes,
                      float drop_prob, uint64_t seed, float* out, float* taps, float* bn_batch_stats, void* workspace,
                      size_t workspace_bytes, void* stream);
int eec_train_backward(eec_trainer* tr, const eec_params* params, const eec_params* grads, const float* out, const float* grad_out,
                       const float* grad_taps, void* workspace, size_t workspace_bytes, void* stream);
/* The same backward, reporting its progress: the exit groups are differentiated last to first (train.py:60-68 sums the exit
 * losses, so the gradient of group e's parameters is final once the backward has passed group e), and `on_group(e, user)` is
 * called on the host right after the last launch that writes a gradient of exit group e (its layers and its head; e = E-1 ... 0),
 * then once with e = -1 after the stem.  Everything the callback enqueues on `stream` -- or on a stream that waits for it, as
 * torch.distributed's collectives do -- therefore runs behind those gradients and beside the backward of the earlier groups:
 * the hook that lets data-parallel training (BASELINE.json configs[3]) all-reduce bucket e under the backward of group e - 1.
 * The callback must not call into this library. */
typedef void (*eec_group_done_fn)(int group, void* user);
int eec_train_backward_ex(eec_trainer* tr, const eec_params* params, const eec_params* grads, const float* out, const float* grad_out,
                          const float* grad_taps, void* workspace, size_t workspace_bytes, void* stream, eec_group_done_fn on_group,
                          void* user);
/* ---- Building blocks of the training step: what `--model_type splitformer / zipformer` need (train.py:180-208) --------------
 * The same modules as eec_train_forward / _backward, cut where those models put their own glue (strided slices, repeats, adds
 * between Conformer groups: early_exit.py:117-224, 227-364).  Stateless: the activations a backward needs are recorded in the
 * caller's `workspace` (256-byte aligned, *_workspace_bytes; untouched until the matching backward, which takes the same
 * geometry, seed, drop_prob and site numbers).  A GROUP = n_layers ConformerLayers (torchaudio Conformer(num_layers=n_layers))
 * on rows x [B][T'][D] with key lengths key_len [B] (int32, device): x_out [B][T'][D], bn_batch_stats [n_layers][2][D] as in
 * eec_train_forward; the backward writes the gradient of every layer parameter (grads mirrors layers) and grad_in = dLoss/dx_in.
 * site_base numbers the group's dropout sites (7 per layer): calls of one step must use disjoint ranges.  The STEM =
 * Conv1d(k3, s2) [-> Conv1d(k3, s2) when sub1_* are given] -> + positional encoding -> dropout: x_out [B][To][D], To = T1 or T';
 * no gradient with respect to mel.  The HEAD = log_softmax(x . W^T + b) and the backward of exactly that. */
size_t eec_train_group_workspace_bytes(const eec_config* cfg, int n_layers, int B, int Tq);
int eec_train_group_forward(const eec_config* cfg, const eec_layer_params* layers, int n_layers, const float* x_in, const int32_t* key_len, int B,
                            int Tq, int passes, float drop_prob, uint64_t seed, uint32_t site_base, float* x_out, float* bn_batch_stats,
                            void* workspace, size_t workspace_bytes, void* stream);
int eec_train_group_backward(const eec_config* cfg, const eec_layer_params* layers, const eec_layer_params* grads, int n_layers, const float* x_in,
                             const int32_t* key_len, int B, int Tq, int passes, float drop_prob, uint64_t seed, uint32_t site_base,
                             const float* grad_out, float* grad_in, void* workspace, size_t workspace_bytes, void* stream);
size_t eec_train_stem_workspace_bytes(const eec_config* cfg, int B, int T, int two_convs);
int eec_train_stem_forward(const eec_config* cfg, const float* sub0_w, const float* sub0_b, const float* sub1_w, const float* sub1_b,
                           const float* pe, const float* mel, int B, int T, int passes, float drop_prob, uint64_t seed, uint32_t site, float* x_out,
                           void* workspace, size_t workspace_bytes, void* stream);
int eec_train_stem_backward(const eec_config* cfg, int two_convs, int B, int T, int passes, float drop_prob, uint64_t seed, uint32_t site,
                            const float* grad_x, float* g_sub0_w, float* g_sub0_b, float* g_sub1_w, float* g_sub1_b, void* workspace,
                            size_t workspace_bytes, void* stream);
int eec_train_head_forward(const float* x, const float* W, const float* b, int M, int V, int D, int passes, float* logp, float* scratch /* M*V */,
                           void* stream);
size_t eec_train_head_backward_scratch_floats(int M, int V, int D);
int eec_train_head_backward(const float* x, const float* W, const float* logp, const float* grad_logp, int M, int V, int D, int passes, float* dx,
                            float* dW, float* db, float* scratch, void* stream);
/* C = alpha * A . B^T (+ bias) on the training GEMM (test hook): A [M][K], B [N][K], C [M][N] fp32 row-major on the device */
int eec_train_gemm(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int passes, int a_transposed,
                   int b_transposed, void* stream);

/* ---- AED decoder forward (full_conformer._decoder_, early_exit.py:739-762; util/beam_infer.py:236-240) ----------------
 * out[Bm][S][V] = (log_softmax of) linears_2[e]( TransformerDecoder_e( positional_encoder_2(emb(trg)), memory = enc ) ) in eval
 * mode: n_layers x nn.TransformerDecoderLayer(batch_first, norm_first: causal + target-padding self-attention,
 * cross-attention over enc [Bm][Tq][D], ReLU feed-forward) and the shared final LayerNorm.  fp32 parameters are read in
 * place (state_dict tensors); arithmetic as the training GEMM (passes 3: bf16 hi/lo split, ~1e-5 of fp32).  trg: int64
 * [Bm][S]; positions equal to pad_idx are masked as keys.  The caller's beam search (util/beam_infer.py:198-307) stays
 * above this call: one call per decoding step on the whole prefix, as the reference (the step-wise form with a key / value
 * cache is eec_decoder_begin / eec_decoder_step below).  enc_shared != 0: every one of the Bm rows attends
 * to the SAME memory enc [1][Tq][D] (beam search expands one utterance over its beams): its keys / values are projected once. */
typedef struct eec_decoder_layer_params {
  const float *sa_in_w, *sa_in_b;   /* self_attn.in_proj_{weight,bias}      [3D,D],[3D] */
  const float *sa_out_w, *sa_out_b; /* self_attn.out_proj.{weight,bias}     [D,D],[D]   */
  const float *ca_in_w, *ca_in_b;   /* multihead_attn.in_proj_{weight,bias}             */
  const float *ca_out_w, *ca_out_b; /* multihead_attn.out_proj.{weight,bias}            */
  const float *w1, *b1, *w2, *b2;   /* linear1 [F,D],[F]; linear2 [D,F],[D]             */
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b;
} eec_decoder_layer_params;
typedef struct eec_decoder_params {
  const float* emb;                       /* emb.weight [V,D]                                   */
  const float* pe;                        /* positional_encoder_2.pe [max_len,1,D]              */
  const eec_decoder_layer_params* layers; /* HOST array of n_layers: decoders.e.layers.l        */
  int32_t n_layers, max_len;
  const float *norm_w, *norm_b;           /* layer_norm.{weight,bias} (shared final norm)       */
  const float *head_w, *head_b;           /* linears_2.e.{weight,bias} [V,D],[V]                */
} eec_decoder_params;
const char* eec_decoder_last_error(void);
size_t eec_decoder_workspace_bytes(int d_model, int n_heads, int d_ff, int vocab, int Bm, int S, int Tq);
int eec_decoder_forward(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, int pad_idx, const int64_t* trg,
                        const float* enc, int Bm, int S, int Tq, int enc_shared, int passes, int log_softmax, float* out,
                        void* workspace, size_t workspace_bytes, void* stream);


/* ---- Training step of the AED decoder (csrc/decoder_train.hip): what autograd does through the decoder half of
 * full_conformer.forward in train mode (early_exit.py:764-800; train.py:36-52, --decoder_mode aed) -----------------------------
 * eec_decoder_train_forward: out [Bm][S][V] = RAW logits (the reference comments the log_softmax out, early_exit.py:790) of exit
 * `exit_index`'s decoder on targets trg [Bm][S] (int64; pad_idx positions masked as keys) over the memory enc [Bm][Tq][D], with
 * dropout of probability drop_prob at the reference's sites (after the positional encoding -- site shared by the exits of a
 * forward, the reference embeds the targets once --, on both attention-probability tensors, after the three sub-modules, inside
 * the feed-forward), from the counter-based generator keyed by (seed, exit_index).  It records what the backward needs in
 * `workspace` (eec_decoder_train_workspace_bytes; 256-byte aligned; keep it untouched until the backward).
 * eec_decoder_train_backward (same geometry, trg, enc, seed, drop_prob, exit_index): grad_out = dLoss / d out; `grads` mirrors
 * `p` with pointers that are WRITTEN with the gradient of the parameter in the same position (emb [V][D]; pe ignored; layers a
 * host array); grad_enc [Bm][Tq][D] is written with the gradient of the memory (what flows on into the encoder's backward as
 * eec_train_backward's grad_taps).  The shared final LayerNorm and the embedding receive one such gradient per exit: the caller
 * sums them (autograd does).  Parity with the reference's modules is at drop_prob 0 (its dropout streams cannot match). */
const char* eec_decoder_train_last_error(void);
size_t eec_decoder_train_workspace_bytes(int d_model, int n_heads, int d_ff, int vocab, int n_layers, int Bm, int S, int Tq);
int eec_decoder_train_forward(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, int pad_idx, const int64_t* trg,
                              const float* enc, int Bm, int S, int Tq, int passes, float drop_prob, uint64_t seed, int exit_index, float* out,
                              void* workspace, size_t workspace_bytes, void* stream);
int eec_decoder_train_backward(const eec_decoder_params* p, const eec_decoder_params* grads, int d_model, int n_heads, int d_ff, int vocab,
                               const int64_t* trg, const float* enc, int Bm, int S, int Tq, int passes, float drop_prob, uint64_t seed,
                               int exit_index, const float* grad_out, float* grad_enc, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Step-wise AED decoding with a key / value cache (csrc/decoder_step.hip) ----------------------------------------------------
 * What util/beam_infer.py:233-240 needs from `_decoder_` is the LAST position's log-probs of every live beam; the reference gets
 * them by re-running the decoder over the whole prefix at every step.  A session here is a caller-owned device buffer `cache`
 * (eec_decoder_cache_bytes) holding the memory keys / values of ONE utterance's encoder output (projected once by
 * eec_decoder_begin), the self-attention keys / values of every (position, beam slot) decoded so far and each beam's ancestry.
 *   eec_decoder_begin(p, ..., enc [Tq][D], Tq, S_max, passes, cache, bytes, stream)
 *   eec_decoder_step (p, ..., pad_idx, last_tokens [R], parent [R] | NULL, R, R_prev, s, Tq, S_max, log_softmax, out [R][V], ...)
 * step s = 0, 1, 2, ... in order (s < S_max <= p->max_len): last_tokens[r] is beam r's token at position s, parent[r] the row of the
 * PREVIOUS step that beam r extends (NULL: r itself; ignored at s = 0), R_prev that step's beam count; 1 <= R <=
 * eec_decoder_step_max_beams() (16).  out[r] = log_softmax (or the raw logits) of the exit head at position s -- what
 * `_decoder_(prefix_r, enc, exit)[:, -1]` returns.  Plain fp32 arithmetic; the memory projection of _begin runs on the training
 * GEMM (`passes` as eec_decoder_forward).  EEC_ERR_UNSUPPORTED for geometries outside head dim 8 / 16 / 32 / 64, d_model <= 1024
 * and d_ff <= 2048 (multiples of 4): callers then stay on eec_decoder_forward. */
const char* eec_decoder_step_last_error(void);
int eec_decoder_step_max_beams(void);
size_t eec_decoder_cache_bytes(int d_model, int n_heads, int d_ff, int vocab, int n_layers, int S_max, int Tq);
int eec_decoder_begin(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, const float* enc, int Tq, int S_max,
                      int passes, void* cache, size_t cache_bytes, void* stream);
int eec_decoder_step(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, int pad_idx, const int64_t* last_tokens,
                     const int64_t* parent, int R, int R_prev, int s, int Tq, int S_max, int log_softmax, float* out, void* cache,
                     size_t cache_bytes, void* stream);
/* The same step for n <= 8 sessions at once -- the E exits of one utterance, which inference.py:44-51 decodes one after the other:
 * every launch of the step covers all sessions (one more grid dimension), so E searches cost the launches of one.  ps / caches: HOST
 * arrays of n pointers (sessions of one decoder geometry, each begun with eec_decoder_begin); last_tokens [n][R], parent [n][R] | NULL,
 * out [n][R][V]; R, R_prev and s are common to the sessions (they advance in lockstep). */
int eec_decoder_step_multi(int n, const eec_decoder_params* const* ps, int d_model, int n_heads, int d_ff, int vocab, int pad_idx,
                           const int64_t* last_tokens, const int64_t* parent, int R, int R_prev, int s, int Tq, int S_max, int log_softmax,
                           float* out, void* const* caches, size_t cache_bytes, void* stream);
/* The bookkeeping of one beam-search step (util/beam_infer.py:241-262) for n searches in lockstep, one launch: over the R live beams'
 * V next-token log-probs, cand = scores_in[r] + logp[r][v] / penalty; the K best, best first (ties: the lower r * V + v) ->
 * scores_out [n][K], parent [n][K] (= index / V), tok [n][K] (= index % V); tokens_new[i][b][0 .. len] = tokens_old[i][parent][0 .. len)
 * followed by tok.  Token buffers: [n][rows_ld][ld] int64, len tokens per beam so far.  R, K <= 16. */
int eec_beam_select(int n, int R, int V, int K, const float* logp, const float* scores_in, float penalty, float* scores_out, int64_t* parent,
                    int64_t* tok, const int64_t* tokens_old, int64_t* tokens_new, int len, int ld, int rows_ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EEC_H_ */
