// Internal (non-ABI) declarations: kernel argument blocks and host launchers.
#pragma once
#include "eec_device.h"

namespace eec {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, DEVICE): the attribute is per device, so a
// process-wide flag would leave the > 64 KiB launches of a second device failing.  Thread-safe.
hipError_t ensure_max_lds(const void* kernel, int bytes);
constexpr int kUploadMax = 480;  // int64 values that fit a kernel's argument block
hipError_t launch_upload_i64(const long long* host, int n, long long* dev, hipStream_t st);

struct FfnArgs {
  float* x;  // [M][D] fp32, updated in place
  int M, F, D;
  const float *ln_g, *ln_b;
  const uint4* w1p;  // packed log2(e) * W1 [F][256]
  const float* b1;   // log2(e) * b1
  const uint4* w2p;  // packed W2 / log2(e) [256][F]
  const float* b2;
  const float *fin_g, *fin_b;  // optional final LayerNorm (nullptr = none)
  float res_scale = 0.5f;      // x += res_scale * ffn(LN(x)): 0.5 Conformer half-step, 1.0 legacy layer
  bool relu = false;           // legacy layer: ReLU, weights packed without the log2(e) fold
  const uint4 *w1f8 = nullptr, *w2f8 = nullptr;  // the same two matrices in the NP == 8 stream layout (optional)
};
hipError_t launch_ffn(const FfnArgs& a, int np, hipStream_t st);

struct QkvArgs;
struct DwArgs;
struct ProjResArgs;

struct QkvArgs {
  const float* x;  // [M][D]
  int M, B, Tq, Tp, H, D;  // Tq = T' (frames per utterance), Tp = padded to 32
  const float *ln_g, *ln_b;
  const uint4* wp;  // packed in_proj_weight [3D][D]
  const float* bias;  // [3D]
  half_t *q, *k, *vt;  // q,k: [B][H][Tp][dh]; vt: [B][H][dh][Tp] (key order permuted per 16)
  half_t* vt_lo = nullptr;  // optional: fp16 residual V - fp16(V), same layout (the PV product then runs on hi + lo)
  const uint4* wf8 = nullptr;  // the same matrix as the f8 record stream (NP == 8)
  half_t *q_lo = nullptr, *k_lo = nullptr;  // optional (exact mode): fp16 residuals of the scaled Q and of K, same layouts
  // 1: Q, K, V^T (and their residuals) fragment-major -- [b][h][block of 32 frames][k-step of 16][lane][8 halves], i.e. every MFMA fragment
  // the fused attention reads (AttnArgs::vt_frag) is 1 KiB contiguous instead of 32 rows x 32 B.  Only the chain tail's whole-line store path
  // writes it (d_model 256, 8 heads, T' % 64 == 0, see qkv_body); the plan sets it exactly when the fused launch consumes the planes.
  int vt_frag = 0;
};
hipError_t launch_qkv(const QkvArgs& a, int np, hipStream_t st);

struct ProjResArgs {  // x += planes . W^T + bias   (attention out-proj; also the GEMM half of dw_pw2 / proj_glu)
  float* x;
  int M, D;
  const half_t *a_hi, *a_lo;  // [M][256]
  const uint4* wp;            // packed [D][D]
  const float* bias;
  const uint4* wf8 = nullptr;  // f8 record stream of the same matrix (NP == 8)
};
hipError_t launch_proj_residual(const ProjResArgs& a, int np, hipStream_t st);

struct GluArgs {  // g = GLU(LN(x) . W^T + b): value cols [0,D), gate cols [D,2D)
  const float* x;
  int M;
  const float *ln_g, *ln_b;
  const uint4* wp;  // packed [2D][D]
  const float* bias;
  half_t* g;  // [M][D] fp16
  const uint4* wf8 = nullptr;  // f8 record stream of the same matrix (NP == 8)
};
// fused: attention out-proj + residual -> conv LayerNorm -> pointwise-1 -> GLU (g.x is ignored: rows come from a.x)
hipError_t launch_proj_glu(const ProjResArgs& a, const GluArgs& g, int np, hipStream_t st);

struct AttnArgs;
// attention + out_proj + residual -> LN -> pointwise-1 -> GLU in ONE launch (the O planes never leave the CU); only for
// shapes where attn_fusable() holds (8 heads, T' a multiple of the row tile: every tile inside one utterance)
bool attn_fusable(const AttnArgs& at, int D);
hipError_t launch_attn_proj_glu(const AttnArgs& at, const ProjResArgs& a, const GluArgs& g, int np, hipStream_t st);

struct HeadArgs {  // out = log_softmax(x . W^T + b)
  const float* x;
  int M, V, D;
  const uint4* wp;  // packed [V][D]
  const float* bias;
  float* out;  // [M][V]
  const uint4* wf8 = nullptr;  // f8 record stream of the same matrix (NP == 8)
};
hipError_t launch_head(const HeadArgs& a, int np, hipStream_t st);

// All exit heads in ONE launch (north_star: "all exit logits come from one launch"): exit e reads its rows from
// x[e] ([M][256] fp32: the exit taps, the last exit straight from the residual stream) and writes out + e*M*V.
constexpr int kMaxHeadExits = 16;
struct HeadBatchArgs {
  const float* x[kMaxHeadExits];
  const uint4* wp[kMaxHeadExits];
  const uint4* wf8[kMaxHeadExits];
  const float* bias[kMaxHeadExits];
  float* out;  // [E][M][V]
  int M, V, E, D;
};
hipError_t launch_head_batch(const HeadBatchArgs& a, int np, hipStream_t st);

struct AttnArgs {
  const half_t *q, *k, *vt;
  const int* enc_len;  // [B] valid encoder frames (keys >= len are masked)
  int B, H, Tq, Tp, dh;
  half_t *o_hi, *o_lo;  // [M][256]
  const half_t* vt_lo = nullptr;  // optional residual plane of V^T (see QkvArgs)
  const half_t *q_lo = nullptr, *k_lo = nullptr;  // optional residual planes of Q and K: the score and PV products then run as three
                                                  // fp16 MFMA products each (the probabilities are split hi / lo in registers)
  int vt_frag = 0;  // Q / K / V^T planes are fragment-major (QkvArgs::vt_frag); the fused launch only
};
hipError_t launch_attention(const AttnArgs& a, int np, hipStream_t st);

struct DwArgs {
  const half_t* g;  // [B*Tq][D]
  int B, Tq;
  const float* wfold;  // [31][D] taps (BN folded, zero padded to 31, centred)
  const float* bfold;  // [D]
  half_t *o_hi, *o_lo;  // unused by the fused kernel (kept for layout compatibility of the argument block)
};
// fused depthwise conv + BN + SiLU -> pointwise-2 + residual (o_hi/o_lo of DwArgs are unused)
hipError_t launch_dw_pw2(const DwArgs& d, const ProjResArgs& a, int np, hipStream_t st);

// Chain kernel (ffn.hip): [depthwise + pointwise-2 front ->] 1 or 2 FFN stages [-> QKV tail] on one row tile.
struct FfnStage {
  const float *ln_g, *ln_b;
  const uint4* w1p;
  const float* b1;
  const uint4* w2p;
  const float* b2;
  const float *fin_g, *fin_b;  // optional LayerNorm after the residual (layer-final / group-final)
  const uint4 *w1f8, *w2f8;    // NP == 8 stream layout of the same matrices (optional)
  float res_scale;
  float* tap;                  // optional [M][256]: the stage's output rows are stored here too (exit taps)
};
// Training-step extras of a single-stage chain launch (the TR variants of ffn_chain_kernel; train.hip ffn_fwd): what the backward
// needs is recorded on the way, the two dropout sites of the module are applied with the training step's generator (eec_drop.h)
struct ChainTrain {
  float* y;                 // [M][D] output rows (x is only read)
  float *ln, *mean, *rstd;  // [M][D], [M], [M]: LayerNorm output and statistics of the input rows (forward only)
  float *pre, *act;         // [M][F]: W1 . LN(x) + b1, and drop(silu(pre)) -- both written by the forward.
                            // Backward (launch_ffn_train_bwd): `pre` is READ and `act` receives d(pre) = (dh . W2) * mask * silu'(pre)
  float p;                  // dropout probability (0: no dropout)
  unsigned long long seed;
  unsigned site_act, site_res;
  // forward, optional: the LayerNorm that consumes the output rows next runs in the same row pass -- ln2 = LN(y; ln2_g, ln2_b) [M][D] with
  // its statistics mean2 / rstd2 [M] (the attention module's LayerNorm after the first feed-forward, the layer's final one after the second)
  const float *ln2_g = nullptr, *ln2_b = nullptr;
  float *ln2 = nullptr, *mean2 = nullptr, *rstd2 = nullptr;
  // backward, optional: the module's own LayerNorm backward in the same row pass -- x_in = the module's input rows [M][D], mean / rstd
  // their statistics (read), st[0].ln_g the LayerNorm weight; the launch's x (the gradient of the module's output) then becomes the
  // gradient of the module's input IN PLACE (dx <- dx + LN'(d(LN(x)))), tr.y is not written, and ln_part [blocks][2][D] receives the
  // per-workgroup column sums of the weight / bias gradient (blocks = the launch's grid)
  const float* x_in = nullptr;
  float* ln_part = nullptr;
  // backward, optional: a LayerNorm backward IN FRONT of the module, in the prologue -- the launch's x rows are the gradient of that
  // LayerNorm's output and are replaced in place by the gradient of its input (pl_x: the LayerNorm's input rows, pl_mean / pl_rstd its
  // statistics, pl_g its weight); pl_part [8 x blocks][2][D]: one row of weight / bias gradient sums per wave
  const float *pl_x = nullptr, *pl_mean = nullptr, *pl_rstd = nullptr, *pl_g = nullptr;
  float* pl_part = nullptr;
};
struct ChainArgs {
  float* x;  // [M][D] residual stream, updated in place
  int M, F, nstage, D;
  int Tq;  // frames per utterance (0: unknown); only the EEC_FFN_ROT == 2 chunk order reads it
  FfnStage st[2];
  QkvArgs qkv;      // tail (x / M of this block are ignored)
  DwArgs dw;        // front
  ProjResArgs pw2;  // front (a_hi / a_lo unused)
  ChainTrain tr;    // TR variants only
};
// the feed-forward module of the training step's forward as one launch (plain-domain SiLU, weights packed with scale 1);
// np 3 (split pairs) or 1 (single fp16 / bf16 operands); d_model 256 / 512, F % 32 == 0
hipError_t launch_ffn_train_fwd(const ChainArgs& a, int np, hipStream_t st);
// ... and the data path of its backward as one launch: x = the gradient of the module's output [M][D]; dh = st[0].res_scale *
// dropmask(site_res) * x is what gets multiplied, and is stored to tr.ln (the operand of the W2 / b2 gradient); st[0].w1p = W2^T as
// [F][D] and st[0].w2p = W1^T as [D][F] (bf16 fragments: launch_pack_frags_bf16 / launch_pack_ffn_batch), tr.pre read,
// tr.act <- d(pre) (the operand of the W1 / b1 gradient), tr.y <- d(LN(x)) = d(pre) . W1; no LayerNorm, bias or residual
hipError_t launch_ffn_train_bwd(const ChainArgs& a, int np, hipStream_t st);
hipError_t launch_ffn_chain(const ChainArgs& a, int np, int np_front, int np_tail, bool front, bool tail, bool relu, hipStream_t st);

struct SubsampleArgs {
  const float* mel;  // [B][n_mels][T]
  int B, n_mels, T, T1, Tq, D;
  int* mid_e;        // [B*T1]: power-of-two exponent of every conv1 output row's scaled fp16 domain (scratch; unused by the one-conv stem)
  const uint4* w1p;       // packed conv1 weight as [256][n_mels*3] (its own [ci][j] flattening)
  const float* b1;
  const uint4* w2p;       // packed conv2 weight as [256][3*256], k ordered (j, ci)
  const float* b2;
  const float* pe;        // [max_len][256]
  half_t *mid_hi, *mid_lo;  // [B*T1][256] scratch planes (scaled by 2^-6)
  float* x;               // [B*Tq][256]
};
hipError_t launch_subsample(const SubsampleArgs& a, int np, hipStream_t st);
hipError_t launch_subsample_single(const SubsampleArgs& a, hipStream_t st);  // conv1 only: x = conv + bias + pe (mid / w2p / b2 unused)

// weight packing (device -> device)
hipError_t launch_pack_frags(const float* w, int N, int K, uint4* out, float scale, hipStream_t st);  // out may point into a larger matrix: n-tile nt0 of [N'][K] starts at out + nt0*(K/16)*128  // scale*W[N][K] -> fragments
hipError_t launch_pack_frags_bf16(const float* w, int N, int K, long ldn, long ldk, uint4* out, hipStream_t st);  // element (n, k) = w[n*ldn + k*ldk] -> bf16 hi / lo fragments
constexpr int kFfnPackModules = 12;
struct FfnPackJobs {  // see pack.hip pack_ffn_batch_kernel
  const float* w1[kFfnPackModules];   // [F][D]
  const float* w2[kFfnPackModules];   // [D][F]
  uint4* out[kFfnPackModules][4];     // F * D * 4 bytes each
  int n;
};
hipError_t launch_pack_ffn_batch(const FfnPackJobs& jb, int F, int D, int kind0, int nk, hipStream_t st);  // images kind0 .. kind0 + nk - 1
hipError_t launch_pack_frags_f8(const float* w, int N, int K, uint4* out, float scale, hipStream_t st);  // NP == 8 stream (K % 64 == 0)
hipError_t launch_scale_copy(const float* src, float* dst, int n, float scale, hipStream_t st);
hipError_t launch_fold_dw(const float* dw_w, const float* dw_b, const float* bn_w, const float* bn_b,
                          const float* bn_rm, const float* bn_rv, int ksize, int D, float* wfold, float* bfold,
                          hipStream_t st);
// conv weight [co][ci][3] -> fragments of the [co][3*ci_total] matrix with k ordered (j, ci)
hipError_t launch_pack_conv_jci(const float* w, int cout, int cin, uint4* out, hipStream_t st);
hipError_t launch_fill_int(int* dst, int n, int v, hipStream_t st);
hipError_t launch_enc_lengths(const long long* lengths, int B, int Tq, int* enc_len, hipStream_t st);

// greedy CTC (argmax -> unique_consecutive -> drop blank)
hipError_t launch_greedy_ctc(const float* logp, int n_seq, int Tq, int V, int blank, int* tokens, int* counts,
                             hipStream_t st);

// batched CTC forward: per-lattice negative log-likelihoods nll[E*B] and per-exit batch-mean losses out[E];
// astore (optional, ctc_store_floats() floats): the scaled alphas of every step, for launch_ctc_backward
hipError_t launch_ctc_loss(const float* logp, const long long* targets, const long long* target_len, int E, int B, int Tq,
                           int V, int S, int blank, float* nll, float* out, float* astore, hipStream_t st);
size_t ctc_store_floats(int E, int B, int Tq, int S);
// dlogp[E][B][Tq][V] = d( sum_e grad_loss[e] * loss_e ) / d logp  (astore is consumed: overwritten with state posteriors)
hipError_t launch_ctc_backward(const float* logp, const long long* targets, const long long* target_len, int E, int B, int Tq,
                               int V, int S, int blank, const float* nll, float* astore, const float* grad_loss, float* dlogp,
                               hipStream_t st);
hipError_t launch_logsoftmax_backward(const float* logp, const float* g, int M, int V, float* dlogits, hipStream_t st);

}  // namespace eec
