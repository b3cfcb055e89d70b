// Training step of the Early_conformer path behind the C ABI (include/eec.h, eec_trainer_*): the forward in train mode
// (batch-statistics BatchNorm, dropout at the reference's sites) that records what the backward needs, and the backward
// that produces the gradient of every parameter -- what `enc_out = model(...)` / `loss.backward()` do in train.py:53-68.
// Host code only; the kernels are in train_kernels.hip (+ the log-softmax backward of ctc.hip, the length kernel of pack.hip).
//
// Layer semantics follow torchaudio's ConformerLayer as restated in oracle/conformer_ref.py:
//   x += 0.5 * drop(W2 . drop(silu(W1 . LN(x))));  x += drop(out_proj(MHA(LN(x), attention-prob dropout)));
//   x += drop(pw2 . silu(BN_batch(dw(GLU(pw1 . LN(x))))));  x += 0.5 * ffn2(x);  x = LN(x)
// stem: Conv1d(k3, s2) o Conv1d(k3, s2) (no activation), + positional encoding, dropout (early_exit.py:24-48, 617-623).
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/eec.h"
#include "eec_kernels.h"
#include "eec_train.h"

using namespace eect;

namespace {

thread_local std::string g_terr;
int tfail(int code, const std::string& msg) {
  g_terr = msg;
  return code;
}

struct Bump {  // bump allocator; base == nullptr: sizes only
  char* base = nullptr;
  size_t off = 0, peak = 0, cap = ~(size_t)0;
  bool overflow = false;
  float* f(size_t n) {
    off = (off + 255) / 256 * 256;
    float* p = (float*)(base + off);
    off += n * sizeof(float);
    if (off > peak) peak = off;
    if (off > cap) overflow = true;  // checked by the entry points before anything is reported as done
    return p;
  }
  void reset(size_t to = 0) { off = to; }
};

struct FfnTape {
  float *x, *ln, *mean, *rstd, *pre, *act;
  uint32_t site_act, site_res;
  // fused path: this step's W1 / W2 as MFMA fragments (pack_ffn_weights): [0], [1] the forward's (fp16 pairs), [2], [3] the backward's
  // transposes (bf16 pairs)
  uint4* wp[4];
};
struct AttnTape {
  float *x, *ln, *mean, *rstd, *qkv, *P, *Pd, *ctx;  // Pd: drop(P), kept for the backward (== P when drop_prob is 0)
  float* lse;                                         // fused path (head dim 32 / 64): row log-sum-exp instead of P / Pd
  uint32_t site_p, site_res;
};
struct ConvTape {
  float *x, *ln, *mean, *rstd, *u, *g, *c, *stats, *s;
  uint32_t site_res;
};
struct LayerTape {
  FfnTape f1, f2;
  AttnTape at;
  ConvTape cv;
  float *x4, *fmean, *frstd, *out;
};

}  // namespace

struct eec_trainer {
  eec_config cfg;
  int device = 0;
  // geometry / settings of the recorded forward
  bool recorded = false;
  int B = 0, T = 0, T1 = 0, Tq = 0, M = 0, np = 3;
  float p = 0.0f;
  uint64_t seed = 0;
  std::vector<LayerTape> lt;
  float *a1 = nullptr, *out1 = nullptr, *w2p = nullptr;
  int32_t* key_len = nullptr;
  uint32_t site_pe = 0;
  size_t tape_bytes = 0;
  // weight-gradient GEMMs run on a second stream, beside the dX chain they do not feed (created at the first backward)
  hipStream_t side = nullptr;
  hipEvent_t ev_main = nullptr, ev_side = nullptr;
};

namespace {

struct Run {
  eec_trainer* tr;
  bool dry;
  hipStream_t st;
  Bump tape, scr, sscr;  // sscr: scratch of the side stream (split-K partials), never shared with the main stream's
  hipStream_t side = nullptr;
  bool side_busy = false;
  hipError_t err = hipSuccess;
  const char* where = "";
  uint32_t site = 1;
  void ok(hipError_t e, const char* w) {
    if (e != hipSuccess && err == hipSuccess) err = e, where = w;
  }
};
#define RUN(expr)                      \
  do {                                 \
    if (!r.dry) r.ok((expr), #expr);   \
  } while (0)

Drop drop_of(const Run& r, uint32_t site) { return Drop{r.tr->p, r.tr->seed, site}; }

// Side stream protocol.  A weight-gradient job reads its dY (main-stream scratch) and x (tape) operands: it starts after
// everything the main stream has enqueued so far (side_begin) and the main stream must not overwrite or recycle those
// scratch buffers before it has finished -- backward modules therefore never write into a buffer they handed to a job, and
// join_side() precedes every recycling of the main scratch (module starts) and the end of the backward.
void side_begin(Run& r) {
  if (r.dry || !r.side) return;
  RUN(hipEventRecord(r.tr->ev_main, r.st));
  RUN(hipStreamWaitEvent(r.side, r.tr->ev_main, 0));
}
void side_end(Run& r) {
  if (r.dry || !r.side) return;
  RUN(hipEventRecord(r.tr->ev_side, r.side));
  r.side_busy = true;
}
void join_side(Run& r) {
  if (r.dry || !r.side || !r.side_busy) return;
  RUN(hipStreamWaitEvent(r.st, r.tr->ev_side, 0));
  r.side_busy = false;
}
void bwd_scratch_reset(Run& r) {
  join_side(r);
  r.scr.reset();
}

// y[M][N] = x[M][K] . W[N][K]^T + bias
void linear_fwd(Run& r, const float* x, const float* W, const float* bias, float* y, int M, int N, int K) {
  GemmArgs g = gemm_args(x, K, 1, W, K, 1, y, N, M, N, K);
  g.bias = bias;
  RUN(launch_gemm(g, r.tr->np, r.st));
}
// y[M][N] = r + scale * drop(x[M][K] . W[N][K]^T + bias): a sub-module's last Linear with its residual connection
void linear_residual_fwd(Run& r, const float* x, const float* W, const float* bias, const float* res, float scale, uint32_t site, float* y, int M,
                         int N, int K) {
  GemmArgs g = gemm_args(x, K, 1, W, K, 1, y, N, M, N, K);
  g.bias = bias, g.epi = 4, g.aux = res, g.res_scale = scale, g.drop = drop_of(r, site);
  RUN(launch_gemm(g, r.tr->np, r.st));
}
// dx[M][K] (+)= dy[M][N] . W[N][K]
void linear_bwd_data(Run& r, const float* dy, const float* W, float* dx, int M, int N, int K, bool accumulate = false) {
  GemmArgs g = gemm_args(dy, N, 1, W, 1, K, dx, K, M, K, N);
  g.accumulate = accumulate;
  RUN(launch_gemm(g, r.tr->np, r.st));
}
// dW[N][K] = dy[M][N]^T . x[M][K] (split over the rows, partials summed in a fixed order); db[N] = column sums of dy
void linear_bwd_weight(Run& r, const float* dy, const float* x, float* dW, float* db, int M, int N, int K) {
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  // splits: enough workgroups for two per CU, but at least 8 k-tiles (256 rows) per split -- with 128-row splits a 256 x 256 weight
  // gradient was 128 partials of 263 KB (34 MB written and read back for a 2 GFLOP product) and 4 k-tiles per workgroup
  static const int wg_target = [] { const char* e = getenv("EEC_TRAIN_DW_WGS"); return e ? atoi(e) : 512; }();  // tuning knob
  int S = std::max(1, std::min(wg_target / tiles, M >= 4096 ? M / 256 : M / 64));
  int chunk = ((M + S - 1) / S + 31) / 32 * 32;
  S = (M + chunk - 1) / chunk;
  hipStream_t ws = r.side ? r.side : r.st;
  side_begin(r);
  r.sscr.reset();  // jobs are ordered on their stream: the previous one's partials are consumed before these are written
  GemmArgs g = gemm_args(dy, 1, N, x, 1, K, dW, K, N, K, chunk);
  // the bias gradient = row sums of dY^T, taken inside the same GEMM and reduced over the splits in the same launch as dW
  // (carved whether or not it is wanted: the sizing pass runs with null pointers and must see the same layout)
  if (S > 1) {
    const long pstride = (long)N * K + N;  // one split's partial: [N][K] weight gradient, then [N] bias gradient
    float* part = r.sscr.f((size_t)S * pstride);
    g.C = part, g.nz = S, g.zdiv = 1, g.ktot = M;
    g.a_z0 = (long)chunk * N, g.b_z0 = (long)chunk * K, g.c_z0 = pstride;
    if (db) g.rowsum = part + (long)N * K, g.rowsum_z = pstride;
    RUN(launch_gemm(g, r.tr->np, ws));
    if (db) RUN(launch_reduce_leading_split(part, S, pstride, pstride, (long)N * K, dW, db, ws));
    else RUN(launch_reduce_leading(part, S, pstride, (long)N * K, dW, ws));
  } else {
    g.K = M;
    if (db) g.rowsum = db, g.rowsum_z = 0;
    RUN(launch_gemm(g, r.tr->np, ws));
  }
  side_end(r);
}
// dx = dx + LN'(dln) in place; dg / db from the per-block partials
void ln_bwd(Run& r, const float* dln, const float* x, const float* g, const float* mean, const float* rstd, float* dx, bool add_res, float* dg,
            float* db, int M, int D) {
  const size_t mark = r.scr.off;
  const int nb = ln_bwd_blocks(M);
  float* part = r.scr.f((size_t)nb * 2 * D);
  RUN(launch_ln_bwd(dln, x, g, mean, rstd, add_res ? dx : nullptr, dx, part, M, D, r.st));
  RUN(launch_reduce_leading2(part, nb, D, dg, db, r.st));
  r.scr.reset(mark);
}

// ---- forward ---------------------------------------------------------------------------------------------------------
// the feed-forward module as one launch (a property of the configuration: the sizing pass carves the same way);
// EEC_TRAIN_FFN_FUSED=0 keeps the LayerNorm + two-GEMM path (A/B runs, and what other geometries take)
bool ffn_fused_fwd_supported(const eec_trainer* tr) {
  const char* e = getenv("EEC_TRAIN_FFN_FUSED");  // read per call: the tests flip it between steps of one process
  const bool off = e && atoi(e) == 0;
  const int D = tr->cfg.d_model, F = tr->cfg.d_ff;
  return !off && (D == 256 || D == 512) && F >= 32 && F % 32 == 0 && (tr->np == 1 || tr->np == 3);
}
bool ffn_fused_bwd_supported(const eec_trainer* tr) {
  const char* e = getenv("EEC_TRAIN_FFN_FUSED_BWD");
  const bool off = e && atoi(e) == 0;
  const int D = tr->cfg.d_model, F = tr->cfg.d_ff;
  return !off && (D == 256 || D == 512) && F >= 32 && F % 32 == 0 && (tr->np == 1 || tr->np == 3);
}
// Room for the fragment images of every feed-forward module's W1 / W2 (4 x F x D x 4 bytes per module), carved from the tape at the
// start of a forward by geometry alone (the tape layout must not depend on the operand mode of a particular step); a module's images
// for a direction are made right before that direction's fused launch (pack_ffn_module).
void carve_ffn_weights(Run& r, int n_layers) {
  eec_trainer* tr = r.tr;
  const int D = tr->cfg.d_model, F = tr->cfg.d_ff;
  if (!((D == 256 || D == 512) && F >= 32 && F % 32 == 0)) return;
  for (int l = 0; l < n_layers; ++l)
    for (int m = 0; m < 2; ++m) {
      FfnTape& t = m == 0 ? tr->lt[l].f1 : tr->lt[l].f2;
      for (int k = 0; k < 4; ++k) t.wp[k] = (uint4*)r.tape.f((size_t)F * D);
    }
}
void pack_ffn_module(Run& r, const FfnTape& t, const float* w1, const float* w2, int kind0) {  // kind0 0: forward images, 2: backward
  eec::FfnPackJobs jb{};
  jb.w1[0] = w1, jb.w2[0] = w2, jb.n = 1;
  for (int k = 0; k < 4; ++k) jb.out[0][k] = t.wp[k];
  RUN(eec::launch_pack_ffn_batch(jb, r.tr->cfg.d_ff, r.tr->cfg.d_model, kind0, 2, r.st));
}
// LayerNorm rows + statistics that a feed-forward module's fused launch writes for whoever normalises its output next
struct NextLn {
  const float *g = nullptr, *b = nullptr;       // in: the parameters
  float *ln = nullptr, *mean = nullptr, *rstd = nullptr;  // out: tape buffers, set (and filled) when the fused launch ran
};
float* ffn_fwd(Run& r, FfnTape& t, float* x, const float* ln_w, const float* ln_b, const float* w1, const float* b1, const float* w2, const float* b2,
               NextLn* next = nullptr) {
  const int M = r.tr->M, D = r.tr->cfg.d_model, F = r.tr->cfg.d_ff;
  t.x = x, t.ln = r.tape.f((size_t)M * D), t.mean = r.tape.f(M), t.rstd = r.tape.f(M), t.pre = r.tape.f((size_t)M * F);
  t.act = r.tape.f((size_t)M * F);
  t.site_act = r.site++, t.site_res = r.site++;
  if (ffn_fused_fwd_supported(r.tr)) {
    // ONE launch (ffn.hip, TR variants of the chain kernel): LayerNorm, both GEMMs, SiLU, both dropout sites and the residual; the
    // [M, F] tensors are written to the tape from the accumulators and never read back by the forward.  Its GEMMs run on split
    // fp16 fragments of THIS step's parameters, made right before it.
    float* y = r.tape.f((size_t)M * D);
    pack_ffn_module(r, t, w1, w2, 0);
    eec::ChainArgs a{};
    a.x = x, a.M = M, a.F = F, a.nstage = 1, a.D = D;
    a.st[0] = eec::FfnStage{ln_w, ln_b, t.wp[0], b1, t.wp[1], b2, nullptr, nullptr, nullptr, nullptr, 0.5f, nullptr};
    a.tr = eec::ChainTrain{y, t.ln, t.mean, t.rstd, t.pre, t.act, r.tr->p, (unsigned long long)r.tr->seed, t.site_act, t.site_res};
    if (next) {  // the consumer's LayerNorm rides in this launch's row pass (saves a launch and a pass over the rows)
      next->ln = r.tape.f((size_t)M * D), next->mean = r.tape.f(M), next->rstd = r.tape.f(M);
      a.tr.ln2_g = next->g, a.tr.ln2_b = next->b, a.tr.ln2 = next->ln, a.tr.mean2 = next->mean, a.tr.rstd2 = next->rstd;
    }
    RUN(eec::launch_ffn_train_fwd(a, r.tr->np, r.st));
    return y;
  }
  RUN(launch_ln_fwd(x, ln_w, ln_b, t.ln, t.mean, t.rstd, M, D, r.st));
  {  // pre = LN(x) . W1^T + b1 and, in the same epilogue, act = drop(silu(pre))
    GemmArgs g = gemm_args(t.ln, D, 1, w1, D, 1, t.pre, F, M, F, D);
    g.bias = b1, g.epi = 1, g.C2 = t.act, g.drop = drop_of(r, t.site_act);
    RUN(launch_gemm(g, r.tr->np, r.st));
  }
  r.scr.reset();
  float* y = r.tape.f((size_t)M * D);
  linear_residual_fwd(r, t.act, w2, b2, x, 0.5f, t.site_res, y, M, D, F);
  return y;
}

// batched-GEMM strides of the attention operands: z = b * H + h
struct AttnGeo {
  int B, H, Tq, D, dh;
  long qkv_b, qkv_h, p_b, p_h, x_b, x_h;
};
AttnGeo attn_geo(const eec_trainer* tr) {
  AttnGeo a{tr->B, tr->cfg.n_heads, tr->Tq, tr->cfg.d_model, tr->cfg.d_model / tr->cfg.n_heads};
  a.qkv_b = (long)a.Tq * 3 * a.D, a.qkv_h = a.dh;
  a.p_b = (long)a.H * a.Tq * a.Tq, a.p_h = (long)a.Tq * a.Tq;
  a.x_b = (long)a.Tq * a.D, a.x_h = a.dh;
  return a;
}
void batched(GemmArgs& g, const AttnGeo& a, long az0, long az1, long bz0, long bz1, long cz0, long cz1) {
  g.nz = a.B * a.H, g.zdiv = a.H;
  g.a_z0 = az0, g.a_z1 = az1, g.b_z0 = bz0, g.b_z1 = bz1, g.c_z0 = cz0, g.c_z1 = cz1;
}

float* attn_fwd(Run& r, AttnTape& t, float* x, const eec_layer_params& L, const NextLn* pre = nullptr) {
  const eec_trainer* tr = r.tr;
  const AttnGeo a = attn_geo(tr);
  const int M = tr->M, D = a.D, Tq = a.Tq;
  const bool fused = attn_fused_supported(D, a.H);  // a property of the configuration: the sizing pass carves the same way
  const bool have_ln = pre && pre->ln;  // the module's LayerNorm already ran in the preceding feed-forward launch
  t.x = x;
  if (have_ln) t.ln = pre->ln, t.mean = pre->mean, t.rstd = pre->rstd;
  else t.ln = r.tape.f((size_t)M * D), t.mean = r.tape.f(M), t.rstd = r.tape.f(M);
  t.qkv = r.tape.f((size_t)M * 3 * D);
  t.ctx = r.tape.f((size_t)M * D);
  t.P = t.Pd = t.lse = nullptr;
  if (fused) {
    t.lse = r.tape.f((size_t)a.B * a.H * Tq);
  } else {
    t.P = r.tape.f((size_t)a.B * a.H * Tq * Tq);
    float* pd = r.tape.f((size_t)a.B * a.H * Tq * Tq);  // carved whatever drop_prob is: the layout must not depend on the settings
    t.Pd = tr->p > 0.0f ? pd : t.P;
  }
  t.site_p = r.site++, t.site_res = r.site++;
  if (!have_ln) RUN(launch_ln_fwd(x, L.attn_ln_w, L.attn_ln_b, t.ln, t.mean, t.rstd, M, D, r.st));
  linear_fwd(r, t.ln, L.attn_in_w, L.attn_in_b, t.qkv, M, 3 * D, D);
  r.scr.reset();
  if (fused) {
    RUN(launch_attn_fwd_fused(t.qkv, tr->key_len, t.ctx, t.lse, a.B, a.H, Tq, D, tr->np, drop_of(r, t.site_p), r.st));
  } else {
    {  // S = Q . K^T
      GemmArgs g = gemm_args(t.qkv, 3 * D, 1, t.qkv + D, 3 * D, 1, t.P, Tq, Tq, Tq, a.dh);
      batched(g, a, a.qkv_b, a.qkv_h, a.qkv_b, a.qkv_h, a.p_b, a.p_h);
      RUN(launch_gemm(g, tr->np, r.st));
    }
    RUN(launch_softmax_fwd(t.P, tr->p > 0.0f ? t.Pd : nullptr, tr->key_len, a.B, a.H, Tq, 1.0f / sqrtf((float)a.dh), drop_of(r, t.site_p), r.st));
    {  // ctx = Pd . V
      GemmArgs g = gemm_args(t.Pd, Tq, 1, t.qkv + 2 * D, 1, 3 * D, t.ctx, D, Tq, a.dh, Tq);
      batched(g, a, a.p_b, a.p_h, a.qkv_b, a.qkv_h, a.x_b, a.x_h);
      RUN(launch_gemm(g, tr->np, r.st));
    }
  }
  float* y = r.tape.f((size_t)M * D);
  linear_residual_fwd(r, t.ctx, L.attn_out_w, L.attn_out_b, x, 1.0f, t.site_res, y, M, D, D);
  return y;
}

float* conv_fwd(Run& r, ConvTape& t, float* x, const eec_layer_params& L, float* bn_mv) {
  const eec_trainer* tr = r.tr;
  const int M = tr->M, D = tr->cfg.d_model, K = tr->cfg.dw_kernel;
  t.x = x, t.ln = r.tape.f((size_t)M * D), t.mean = r.tape.f(M), t.rstd = r.tape.f(M), t.u = r.tape.f((size_t)M * 2 * D);
  t.g = r.tape.f((size_t)M * D), t.c = r.tape.f((size_t)M * D), t.stats = r.tape.f(2 * D), t.s = r.tape.f((size_t)M * D);
  t.site_res = r.site++;
  RUN(launch_ln_fwd(x, L.conv_ln_w, L.conv_ln_b, t.ln, t.mean, t.rstd, M, D, r.st));
  linear_fwd(r, t.ln, L.conv_pw1_w, L.conv_pw1_b, t.u, M, 2 * D, D);
  RUN(launch_glu_fwd(t.u, t.g, M, D, r.st));
  RUN(launch_dw_fwd(t.g, L.conv_dw_w, L.conv_dw_b, t.c, tr->B, tr->Tq, D, K, r.st));
  r.scr.reset();
  float* part = r.scr.f((size_t)(colsum_blocks(M) + 1) * 2 * D);
  RUN(launch_bn_stats(t.c, M, D, part, t.stats, bn_mv, r.st));
  RUN(launch_bn_silu_fwd(t.c, t.stats, L.conv_bn_w, L.conv_bn_b, t.s, M, D, r.st));
  float* y = r.tape.f((size_t)M * D);
  linear_residual_fwd(r, t.s, L.conv_pw2_w, L.conv_pw2_b, x, 1.0f, t.site_res, y, M, D, D);
  return y;
}

// One Conformer layer (torchaudio ConformerLayer: ffn1, attention, conv module, ffn2, final LayerNorm).  When a feed-forward module runs
// as the fused launch, the LayerNorm that reads its output next -- the attention module's, the layer's final one -- is computed in that
// launch's row pass (ffn_fwd NextLn) instead of by its own kernel.
float* layer_fwd(Run& r, LayerTape& t, float* x, const eec_layer_params& L, float* bn_mv_layer) {
  const int M = r.tr->M, D = r.tr->cfg.d_model;
  const bool fuse_ln = ffn_fused_fwd_supported(r.tr);
  NextLn attn_ln{L.attn_ln_w, L.attn_ln_b}, fin_ln{L.final_ln_w, L.final_ln_b};
  x = ffn_fwd(r, t.f1, x, L.ffn1_ln_w, L.ffn1_ln_b, L.ffn1_w1, L.ffn1_b1, L.ffn1_w2, L.ffn1_b2, fuse_ln ? &attn_ln : nullptr);
  x = attn_fwd(r, t.at, x, L, &attn_ln);
  x = conv_fwd(r, t.cv, x, L, bn_mv_layer);
  x = ffn_fwd(r, t.f2, x, L.ffn2_ln_w, L.ffn2_ln_b, L.ffn2_w1, L.ffn2_b1, L.ffn2_w2, L.ffn2_b2, fuse_ln ? &fin_ln : nullptr);
  t.x4 = x;
  if (fin_ln.ln) {
    t.out = fin_ln.ln, t.fmean = fin_ln.mean, t.frstd = fin_ln.rstd;
  } else {
    t.fmean = r.tape.f(M), t.frstd = r.tape.f(M), t.out = r.tape.f((size_t)M * D);
    RUN(launch_ln_fwd(x, L.final_ln_w, L.final_ln_b, t.out, t.fmean, t.frstd, M, D, r.st));
  }
  return t.out;
}

void forward(Run& r, const eec_params* P, const float* mel, const int64_t* lengths, float* out, float* bn_mv, float* taps) {
  eec_trainer* tr = r.tr;
  const eec_config& c = tr->cfg;
  const int B = tr->B, T = tr->T, T1 = tr->T1, Tq = tr->Tq, M = tr->M, D = c.d_model, C = c.n_mels, V = c.vocab;
  const int nl = c.n_exits * c.layers_per_exit;
  tr->lt.assign(nl, LayerTape{});
  tr->key_len = (int32_t*)r.tape.f(B);
  tr->a1 = r.tape.f((size_t)B * T1 * 3 * C), tr->out1 = r.tape.f((size_t)B * T1 * D), tr->w2p = r.tape.f((size_t)D * 3 * D);
  float* x = r.tape.f((size_t)M * D);
  tr->site_pe = r.site++;
  RUN(eec::launch_enc_lengths((const long long*)lengths, B, Tq, tr->key_len, r.st));
  RUN(launch_im2col_mel(mel, tr->a1, B, C, T, T1, r.st));
  linear_fwd(r, tr->a1, P->sub0_w, P->sub0_b, tr->out1, B * T1, D, 3 * C);
  RUN(launch_permute_w3(P->sub1_w, tr->w2p, D, D, 1, r.st));
  {  // second conv: three consecutive rows of out1 are one contiguous K = 3D operand row
    GemmArgs g = gemm_args(tr->out1, 2 * D, 1, tr->w2p, 3 * D, 1, x, D, Tq, D, 3 * D);
    g.bias = P->sub1_b, g.nz = B, g.zdiv = 1, g.a_z0 = (long)T1 * D, g.c_z0 = (long)Tq * D;
    RUN(launch_gemm(g, tr->np, r.st));
  }
  RUN(launch_add_pe_drop(x, P->pe, B, Tq, D, drop_of(r, tr->site_pe), r.st));
  carve_ffn_weights(r, nl);
  for (int e = 0; e < c.n_exits; ++e) {
    for (int l = 0; l < c.layers_per_exit; ++l) {
      const int li = e * c.layers_per_exit + l;
      const eec_layer_params& L = P->layers[li];
      LayerTape& t = tr->lt[li];
      x = layer_fwd(r, t, x, L, bn_mv ? bn_mv + (size_t)li * 2 * D : nullptr);
    }
    if (taps) RUN(hipMemcpyAsync(taps + (size_t)e * M * D, x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, r.st));
    r.scr.reset();
    float* logits = r.scr.f((size_t)M * V);
    linear_fwd(r, x, P->head_w[e], P->head_b[e], logits, M, V, D);
    RUN(launch_logsoftmax_fwd(logits, out + (size_t)e * M * V, M, V, r.st));
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------
// a LayerNorm backward that sits in front of a feed-forward module's backward and has no residual path of its own (the layer-final
// LayerNorm before the second module): dx holds the gradient of its output and receives the gradient of its input.  The fused launch does
// it in its prologue; the GEMM path runs the LayerNorm-backward kernel first.
struct PreLnBwd {
  const float *x, *mean, *rstd, *g;
  float *dg, *db;
};
void ffn_bwd(Run& r, const FfnTape& t, float* dx, const float* ln_w, const float* w1, const float* w2, float* g_ln_w, float* g_ln_b, float* g_w1,
             float* g_b1, float* g_w2, float* g_b2, const PreLnBwd* pre = nullptr) {
  const int M = r.tr->M, D = r.tr->cfg.d_model, F = r.tr->cfg.d_ff;
  if (pre && !ffn_fused_bwd_supported(r.tr)) ln_bwd(r, dx, pre->x, pre->g, pre->mean, pre->rstd, dx, false, pre->dg, pre->db, M, D);
  bwd_scratch_reset(r);
  float* dh = r.scr.f((size_t)M * D);
  float* dpre = r.scr.f((size_t)M * F);
  float* dln = r.scr.f((size_t)M * D);  // its own buffer: dh is still being read by the dW2 job
  if (ffn_fused_bwd_supported(r.tr)) {
    // the data path as ONE launch (ffn.hip, TR = 2): dpre = (dh . W2) * dropmask * silu'(pre) chunk by chunk -- stored once, for the
    // W1 / b1 gradient -- and dln = dpre . W1 accumulated on chip; W2^T and W1^T as bf16 fragments of this step's parameters
    pack_ffn_module(r, t, w1, w2, 2);
    eec::ChainArgs a{};
    a.x = dx, a.M = M, a.F = F, a.nstage = 1, a.D = D;
    a.st[0] = eec::FfnStage{nullptr, nullptr, t.wp[2], nullptr, t.wp[3], nullptr, nullptr, nullptr, nullptr, nullptr, 0.5f, nullptr};
    // ... and the module's LayerNorm backward in the launch's row pass: dx becomes the gradient of the module's input in place, the
    // weight / bias gradient leaves as per-workgroup column sums for the usual reduce launch
    const int nb = (M + (D == 256 ? 64 : 32) - 1) / (D == 256 ? 64 : 32);  // the launch's grid: one workgroup per row tile
    float* part = r.scr.f((size_t)nb * 2 * D);
    a.st[0].ln_g = ln_w;
    a.tr = eec::ChainTrain{nullptr, dh, t.mean, t.rstd, t.pre, dpre, r.tr->p, (unsigned long long)r.tr->seed, t.site_act, t.site_res};
    a.tr.x_in = t.x, a.tr.ln_part = part;
    float* ppart = pre ? r.scr.f((size_t)nb * 8 * 2 * D) : nullptr;
    if (pre) a.tr.pl_x = pre->x, a.tr.pl_mean = pre->mean, a.tr.pl_rstd = pre->rstd, a.tr.pl_g = pre->g, a.tr.pl_part = ppart;
    RUN(eec::launch_ffn_train_bwd(a, r.tr->np, r.st));
    if (pre) RUN(launch_reduce_leading2(ppart, nb * 8, D, pre->dg, pre->db, r.st));
    RUN(launch_reduce_leading2(part, nb, D, g_ln_w, g_ln_b, r.st));
    linear_bwd_weight(r, dh, t.act, g_w2, g_b2, M, D, F);
    linear_bwd_weight(r, dpre, t.ln, g_w1, g_b1, M, F, D);
    return;
  }
  RUN(launch_scale_drop(dx, 0.5f, dh, (long)M * D, drop_of(r, t.site_res), r.st));
  linear_bwd_weight(r, dh, t.act, g_w2, g_b2, M, D, F);
  {  // dpre = (dh . W2) * dropmask * silu'(pre), the activation's backward in the GEMM epilogue
    GemmArgs g = gemm_args(dh, D, 1, w2, 1, F, dpre, F, M, F, D);
    g.epi = 2, g.aux = t.pre, g.drop = drop_of(r, t.site_act);
    RUN(launch_gemm(g, r.tr->np, r.st));
  }
  linear_bwd_weight(r, dpre, t.ln, g_w1, g_b1, M, F, D);
  linear_bwd_data(r, dpre, w1, dln, M, F, D);
  ln_bwd(r, dln, t.x, ln_w, t.mean, t.rstd, dx, true, g_ln_w, g_ln_b, M, D);
}

void attn_bwd(Run& r, const AttnTape& t, float* dx, const eec_layer_params& L, eec_layer_params& G) {
  const eec_trainer* tr = r.tr;
  const AttnGeo a = attn_geo(tr);
  const int M = tr->M, D = a.D, Tq = a.Tq;
  const long np_ = (long)a.B * a.H * Tq * Tq;
  const bool fused = attn_fused_supported(D, a.H);
  bwd_scratch_reset(r);
  float* d_o = r.scr.f((size_t)M * D);
  float* dln = r.scr.f((size_t)M * D);  // its own buffer: d_o is still being read by the out_proj weight-gradient job
  float* dctx = r.scr.f((size_t)M * D);
  float* dqkv = r.scr.f((size_t)M * 3 * D);
  float* dP = fused ? r.scr.f((size_t)a.B * a.H * Tq) : r.scr.f((size_t)np_);  // fused: the per-row delta instead
  RUN(launch_scale_drop(dx, 1.0f, d_o, (long)M * D, drop_of(r, t.site_res), r.st));
  linear_bwd_weight(r, d_o, t.ctx, (float*)G.attn_out_w, (float*)G.attn_out_b, M, D, D);
  linear_bwd_data(r, d_o, L.attn_out_w, dctx, M, D, D);
  if (fused) {
    RUN(launch_attn_bwd_fused(t.qkv, tr->key_len, t.ctx, dctx, t.lse, dP, dqkv, a.B, a.H, Tq, D, tr->np, drop_of(r, t.site_p), r.st));
  } else {
    const float* Pd = t.Pd;
    {  // dV[tk][d] = sum_tq Pd[tq][tk] dctx[tq][d]
      GemmArgs g = gemm_args(Pd, 1, Tq, dctx, 1, D, dqkv + 2 * D, 3 * D, Tq, a.dh, Tq);
      batched(g, a, a.p_b, a.p_h, a.x_b, a.x_h, a.qkv_b, a.qkv_h);
      RUN(launch_gemm(g, tr->np, r.st));
    }
    {  // dPd[tq][tk] = sum_d dctx[tq][d] V[tk][d]
      GemmArgs g = gemm_args(dctx, D, 1, t.qkv + 2 * D, 3 * D, 1, dP, Tq, Tq, Tq, a.dh);
      batched(g, a, a.x_b, a.x_h, a.qkv_b, a.qkv_h, a.p_b, a.p_h);
      RUN(launch_gemm(g, tr->np, r.st));
    }
    RUN(launch_softmax_bwd(t.P, dP, a.B, a.H, Tq, 1.0f / sqrtf((float)a.dh), drop_of(r, t.site_p), r.st));
    {  // dQ[tq][d] = sum_tk dS[tq][tk] K[tk][d]
      GemmArgs g = gemm_args(dP, Tq, 1, t.qkv + D, 1, 3 * D, dqkv, 3 * D, Tq, a.dh, Tq);
      batched(g, a, a.p_b, a.p_h, a.qkv_b, a.qkv_h, a.qkv_b, a.qkv_h);
      RUN(launch_gemm(g, tr->np, r.st));
    }
    {  // dK[tk][d] = sum_tq dS[tq][tk] Q[tq][d]
      GemmArgs g = gemm_args(dP, 1, Tq, t.qkv, 1, 3 * D, dqkv + D, 3 * D, Tq, a.dh, Tq);
      batched(g, a, a.p_b, a.p_h, a.qkv_b, a.qkv_h, a.qkv_b, a.qkv_h);
      RUN(launch_gemm(g, tr->np, r.st));
    }
  }
  linear_bwd_weight(r, dqkv, t.ln, (float*)G.attn_in_w, (float*)G.attn_in_b, M, 3 * D, D);
  linear_bwd_data(r, dqkv, L.attn_in_w, dln, M, 3 * D, D);
  ln_bwd(r, dln, t.x, L.attn_ln_w, t.mean, t.rstd, dx, true, (float*)G.attn_ln_w, (float*)G.attn_ln_b, M, D);
}

void conv_bwd(Run& r, const ConvTape& t, float* dx, const eec_layer_params& L, eec_layer_params& G) {
  const eec_trainer* tr = r.tr;
  const int M = tr->M, D = tr->cfg.d_model, K = tr->cfg.dw_kernel;
  bwd_scratch_reset(r);
  float* dv = r.scr.f((size_t)M * D);
  float* dln = r.scr.f((size_t)M * D);  // its own buffer: dv is still being read by the pointwise-2 weight-gradient job
  float* ds = r.scr.f((size_t)M * D);
  float* dc = r.scr.f((size_t)M * D);
  float* du = r.scr.f((size_t)M * 2 * D);
  float* sums = r.scr.f(2 * D);
  RUN(launch_scale_drop(dx, 1.0f, dv, (long)M * D, drop_of(r, t.site_res), r.st));
  linear_bwd_weight(r, dv, t.s, (float*)G.conv_pw2_w, (float*)G.conv_pw2_b, M, D, D);
  linear_bwd_data(r, dv, L.conv_pw2_w, ds, M, D, D);
  {
    const size_t mark = r.scr.off;
    float* part = r.scr.f((size_t)colsum_blocks(M) * 2 * D);
    RUN(launch_bn_silu_bwd(ds, t.c, t.stats, L.conv_bn_w, L.conv_bn_b, part, sums, dc, M, D, r.st));
    RUN(hipMemcpyAsync((void*)G.conv_bn_b, sums, D * sizeof(float), hipMemcpyDeviceToDevice, r.st));
    RUN(hipMemcpyAsync((void*)G.conv_bn_w, sums + D, D * sizeof(float), hipMemcpyDeviceToDevice, r.st));
    r.scr.reset(mark);
    // the depthwise taps' gradient is a weight-gradient job too: side stream, side scratch (dc is not written again in this module)
    side_begin(r);
    r.sscr.reset();
    float* wpart = r.sscr.f((size_t)(dw_bwd_weight_blocks(tr->B, tr->Tq) + 1) * (K + 1) * D);
    RUN(launch_dw_bwd_weight(dc, t.g, wpart, (float*)G.conv_dw_w, (float*)G.conv_dw_b, tr->B, tr->Tq, D, K, r.side ? r.side : r.st));
    side_end(r);
  }
  RUN(launch_dw_bwd_data(dc, L.conv_dw_w, ds, tr->B, tr->Tq, D, K, r.st));  // ds now holds d GLU-output
  RUN(launch_glu_bwd(ds, t.u, du, M, D, r.st));
  linear_bwd_weight(r, du, t.ln, (float*)G.conv_pw1_w, (float*)G.conv_pw1_b, M, 2 * D, D);
  linear_bwd_data(r, du, L.conv_pw1_w, dln, M, 2 * D, D);
  ln_bwd(r, dln, t.x, L.conv_ln_w, t.mean, t.rstd, dx, true, (float*)G.conv_ln_w, (float*)G.conv_ln_b, M, D);
}

void backward(Run& r, const eec_params* P, const eec_params* Gp, const float* out, const float* grad_out, const float* grad_taps,
              eec_group_done_fn on_group = nullptr, void* user = nullptr) {
  eec_trainer* tr = r.tr;
  const eec_config& c = tr->cfg;
  const int B = tr->B, T1 = tr->T1, Tq = tr->Tq, M = tr->M, D = c.d_model, C = c.n_mels, V = c.vocab;
  // the gradient of the residual stream lives at the start of the scratch region for the whole backward
  float* dx = r.tape.f((size_t)M * D);
  for (int e = c.n_exits - 1; e >= 0; --e) {
    const float* tap = tr->lt[(e + 1) * c.layers_per_exit - 1].out;
    bwd_scratch_reset(r);
    float* dlogits = r.scr.f((size_t)M * V);
    RUN(eec::launch_logsoftmax_backward(out + (size_t)e * M * V, grad_out + (size_t)e * M * V, M, V, dlogits, r.st));
    linear_bwd_weight(r, dlogits, tap, (float*)Gp->head_w[e], (float*)Gp->head_b[e], M, V, D);
    linear_bwd_data(r, dlogits, P->head_w[e], dx, M, V, D, e != c.n_exits - 1);
    if (grad_taps) RUN(launch_axpy(dx, grad_taps + (size_t)e * M * D, 1.0f, (long)M * D, r.st));  // what the caller did with the tap itself
    for (int l = c.layers_per_exit - 1; l >= 0; --l) {
      const int li = e * c.layers_per_exit + l;
      const eec_layer_params& L = P->layers[li];
      eec_layer_params G = Gp->layers[li];
      const LayerTape& t = tr->lt[li];
      const PreLnBwd fin{t.x4, t.fmean, t.frstd, L.final_ln_w, (float*)G.final_ln_w, (float*)G.final_ln_b};
      ffn_bwd(r, t.f2, dx, L.ffn2_ln_w, L.ffn2_w1, L.ffn2_w2, (float*)G.ffn2_ln_w, (float*)G.ffn2_ln_b, (float*)G.ffn2_w1, (float*)G.ffn2_b1,
              (float*)G.ffn2_w2, (float*)G.ffn2_b2, &fin);
      conv_bwd(r, t.cv, dx, L, G);
      attn_bwd(r, t.at, dx, L, G);
      ffn_bwd(r, t.f1, dx, L.ffn1_ln_w, L.ffn1_w1, L.ffn1_w2, (float*)G.ffn1_ln_w, (float*)G.ffn1_ln_b, (float*)G.ffn1_w1, (float*)G.ffn1_b1,
              (float*)G.ffn1_w2, (float*)G.ffn1_b2);
    }
    if (on_group && !r.dry) {
      // every gradient of exit group e (its layers and its head) is now enqueued; the weight-gradient jobs of the side
      // stream are joined first (the join the next group's scratch reset would make anyway), so "after everything on the
      // main stream so far" is a sufficient dependency for the caller's collective
      join_side(r);
      on_group(e, user);
    }
  }
  // stem
  bwd_scratch_reset(r);
  float* dx0 = r.scr.f((size_t)M * D);
  float* Gc = r.scr.f((size_t)M * 3 * D);
  float* dout1 = r.scr.f((size_t)B * T1 * D);
  float* dw2p = r.scr.f((size_t)D * 3 * D);
  RUN(launch_scale_drop(dx, 1.0f, dx0, (long)M * D, drop_of(r, tr->site_pe), r.st));
  {
    const size_t mark = r.scr.off;
    float* part = r.scr.f((size_t)B * D * 3 * D);
    GemmArgs g = gemm_args(dx0, 1, D, tr->out1, 1, 2 * D, part, 3 * D, D, 3 * D, Tq);
    g.nz = B, g.zdiv = 1, g.a_z0 = (long)Tq * D, g.b_z0 = (long)T1 * D, g.c_z0 = (long)D * 3 * D;
    RUN(launch_gemm(g, tr->np, r.st));
    RUN(launch_reduce_leading(part, B, (long)D * 3 * D, (long)D * 3 * D, dw2p, r.st));
    RUN(launch_permute_w3(dw2p, (float*)Gp->sub1_w, D, D, 0, r.st));
    r.scr.reset(mark);
    const int nb = colsum_blocks(M);
    float* bpart = r.scr.f((size_t)nb * D);
    RUN(launch_colsum_partial(dx0, M, D, bpart, r.st));
    RUN(launch_reduce_leading(bpart, nb, D, D, (float*)Gp->sub1_b, r.st));
    r.scr.reset(mark);
  }
  {  // G[m][(j, c)] = sum_o dx0[m][o] w2p[o][(j, c)]
    GemmArgs g = gemm_args(dx0, D, 1, tr->w2p, 1, 3 * D, Gc, 3 * D, M, 3 * D, D);
    RUN(launch_gemm(g, tr->np, r.st));
  }
  RUN(launch_col2im_stride2(Gc, dout1, B, T1, Tq, D, r.st));
  linear_bwd_weight(r, dout1, tr->a1, (float*)Gp->sub0_w, (float*)Gp->sub0_b, B * T1, D, 3 * C);
  join_side(r);  // every gradient is complete in main-stream order
  if (on_group && !r.dry) on_group(-1, user);
}

int check_trainer_cfg(const eec_config& c) {
  if (c.arch != EEC_ARCH_CONFORMER) return tfail(EEC_ERR_UNSUPPORTED, "the training step covers the Conformer architecture");
  if (c.d_model <= 0 || c.d_model > 1024 || c.n_heads <= 0 || c.d_model % c.n_heads) return tfail(EEC_ERR_BAD_ARG, "d_model <= 1024, divisible by n_heads");
  if (c.dw_kernel < 1 || c.dw_kernel > 31 || !(c.dw_kernel & 1)) return tfail(EEC_ERR_BAD_ARG, "depthwise kernel: odd, <= 31");
  if (c.d_ff <= 0 || c.n_exits <= 0 || c.layers_per_exit <= 0 || c.n_mels <= 0 || c.vocab <= 0) return tfail(EEC_ERR_BAD_ARG, "bad configuration");
  // the log-softmax backward and the CTC gradient hold a vocabulary row in one wave (csrc/ctc.hip): refuse here, before a
  // forward has recorded a multi-GB tape that loss.backward() could not use
  if (c.vocab > 256 || c.vocab % 4) return tfail(EEC_ERR_UNSUPPORTED, "the training step needs vocab <= 256 and a multiple of 4 (got " + std::to_string(c.vocab) + ")");
  return 0;
}

int set_geometry(eec_trainer* tr, int B, int T) {
  if (B <= 0 || T < 7) return tfail(EEC_ERR_BAD_ARG, "B >= 1, T >= 7");
  tr->B = B, tr->T = T, tr->T1 = (T - 3) / 2 + 1, tr->Tq = (tr->T1 - 3) / 2 + 1, tr->M = B * tr->Tq;
  if (tr->Tq > tr->cfg.max_len) return tfail(EEC_ERR_BAD_ARG, "T' exceeds max_len");
  return 0;
}

}  // namespace

extern "C" {

const char* eec_trainer_last_error(void) { return g_terr.c_str(); }

int eec_trainer_create(const eec_config* cfg, eec_trainer** out) {
  if (!cfg || !out) return tfail(EEC_ERR_BAD_ARG, "null argument");
  if (int rc = check_trainer_cfg(*cfg)) return rc;
  eec_trainer* tr = new eec_trainer();
  tr->cfg = *cfg;
  tr->device = -1;  // bound to the device that is current in the first eec_train_forward
  *out = tr;
  return 0;
}
void eec_trainer_destroy(eec_trainer* tr) {
  if (!tr) return;
  if (tr->side) (void)hipStreamDestroy(tr->side);
  if (tr->ev_main) (void)hipEventDestroy(tr->ev_main);
  if (tr->ev_side) (void)hipEventDestroy(tr->ev_side);
  delete tr;
}

size_t eec_trainer_workspace_bytes(const eec_trainer* tr_in, int B, int T) {
  if (!tr_in) return 0;
  eec_trainer tmp = *tr_in;
  if (set_geometry(&tmp, B, T)) return 0;
  Run r{&tmp, true, nullptr};
  std::vector<eec_layer_params> layers(tmp.cfg.n_exits * tmp.cfg.layers_per_exit);
  std::vector<const float*> heads(tmp.cfg.n_exits, nullptr);
  eec_params P{};
  P.layers = layers.data(), P.head_w = heads.data(), P.head_b = heads.data();
  forward(r, &P, nullptr, nullptr, nullptr, nullptr, nullptr);
  const size_t fwd_scr = r.scr.peak;
  r.scr = Bump{};
  backward(r, &P, &P, nullptr, nullptr, nullptr);
  return (r.tape.peak + 256) + (r.sscr.peak + 256) + std::max(fwd_scr, r.scr.peak) + 4096;
}

int eec_train_forward(eec_trainer* tr, const eec_params* params, const float* mel, const int64_t* lengths, int B, int T, int passes,
                      float drop_prob, uint64_t seed, float* out, float* taps, float* bn_batch_stats, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (!tr || !params || !mel || !lengths || !out || !workspace) return tfail(EEC_ERR_BAD_ARG, "null argument");
  if (passes != 1 && passes != 3) return tfail(EEC_ERR_BAD_ARG, "passes: 1 (bf16) or 3 (bf16x3)");
  if (!(drop_prob >= 0.0f && drop_prob < 1.0f)) return tfail(EEC_ERR_BAD_ARG, "drop_prob in [0, 1)");
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return tfail(EEC_ERR_BAD_ARG, "no current HIP device");
  if (tr->device < 0) tr->device = dev;
  if (dev != tr->device) return tfail(EEC_ERR_BAD_ARG, "the trainer belongs to another device");
  if (int rc = set_geometry(tr, B, T)) return rc;
  if (workspace_bytes < eec_trainer_workspace_bytes(tr, B, T)) return tfail(EEC_ERR_BAD_ARG, "workspace too small");
  tr->np = passes, tr->p = drop_prob, tr->seed = seed, tr->recorded = false;
  Run r{tr, false, (hipStream_t)stream};
  r.tape.base = (char*)workspace;
  // forward once with a null scratch to learn where the tape ends (pointer arithmetic only), then for real
  {
    eec_trainer tmp = *tr;
    Run d{&tmp, true, nullptr};
    forward(d, params, mel, lengths, out, bn_batch_stats, taps);
    tr->tape_bytes = (d.tape.peak + 255) / 256 * 256;
  }
  r.scr.base = (char*)workspace + tr->tape_bytes;
  r.tape.cap = tr->tape_bytes;
  if (workspace_bytes < tr->tape_bytes) return tfail(EEC_ERR_WORKSPACE, "workspace too small");
  r.scr.cap = workspace_bytes - tr->tape_bytes;
  {  // the scratch need of this forward, before any launch
    eec_trainer tmp = *tr;
    Run d{&tmp, true, nullptr};
    forward(d, params, mel, lengths, out, bn_batch_stats, taps);
    if (d.scr.peak > r.scr.cap) return tfail(EEC_ERR_WORKSPACE, "workspace too small for the forward scratch");
  }
  forward(r, params, mel, lengths, out, bn_batch_stats, taps);
  if (r.tape.overflow || r.scr.overflow) return tfail(EEC_ERR_WORKSPACE, "internal: workspace carve exceeded its size");
  if (r.err != hipSuccess) return tfail((int)r.err, std::string(r.where) + ": " + hipGetErrorString(r.err));
  tr->recorded = true;
  return 0;
}

int eec_train_backward(eec_trainer* tr, const eec_params* params, const eec_params* grads, const float* out, const float* grad_out,
                       const float* grad_taps, void* workspace, size_t workspace_bytes, void* stream) {
  return eec_train_backward_ex(tr, params, grads, out, grad_out, grad_taps, workspace, workspace_bytes, stream, nullptr, nullptr);
}

int eec_train_backward_ex(eec_trainer* tr, const eec_params* params, const eec_params* grads, const float* out, const float* grad_out,
                          const float* grad_taps, void* workspace, size_t workspace_bytes, void* stream, eec_group_done_fn on_group,
                          void* user) {
  if (!tr || !params || !grads || !out || !grad_out || !workspace) return tfail(EEC_ERR_BAD_ARG, "null argument");
  if (!tr->recorded) return tfail(EEC_ERR_BAD_ARG, "no recorded forward (eec_train_forward first, same workspace)");
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != tr->device) return tfail(EEC_ERR_BAD_ARG, "the trainer belongs to another device");
  if (workspace_bytes < eec_trainer_workspace_bytes(tr, tr->B, tr->T)) return tfail(EEC_ERR_BAD_ARG, "workspace too small");
  Run r{tr, false, (hipStream_t)stream};
  // the residual-stream gradient is carved from the head of the scratch region through `tape` (kept for the whole backward)
  r.tape.base = (char*)workspace + tr->tape_bytes;
  const size_t dx_bytes = ((size_t)tr->M * tr->cfg.d_model * sizeof(float) + 511) / 256 * 256;
  size_t side_bytes = 0;
  {  // the scratch needs of this backward, before any launch: [tape][dx][side-stream scratch][main scratch]
    eec_trainer tmp = *tr;
    Run d{&tmp, true, nullptr};
    backward(d, params, grads, out, grad_out, grad_taps);
    side_bytes = (d.sscr.peak + 255) / 256 * 256;
    if (workspace_bytes < tr->tape_bytes + dx_bytes + side_bytes || d.scr.peak > workspace_bytes - tr->tape_bytes - dx_bytes - side_bytes ||
        d.tape.peak > dx_bytes)
      return tfail(EEC_ERR_WORKSPACE, "workspace too small for the backward scratch: need " + std::to_string(d.scr.peak) + " + " +
                                          std::to_string(d.sscr.peak) + " + " + std::to_string(d.tape.peak) + " beside the tape of " +
                                          std::to_string(tr->tape_bytes) + ", workspace " + std::to_string(workspace_bytes));
  }
  const char* no_side = getenv("EEC_TRAIN_NO_SIDE");  // diagnostic: every job on the main stream (results are bit-identical)
  if (no_side && no_side[0] == '1') {
    r.side = nullptr;
  } else if (!tr->side) {  // a failure here only means the jobs run on the main stream
    if (hipStreamCreateWithFlags(&tr->side, hipStreamNonBlocking) != hipSuccess) tr->side = nullptr;
    if (tr->side && (hipEventCreateWithFlags(&tr->ev_main, hipEventDisableTiming) != hipSuccess ||
                     hipEventCreateWithFlags(&tr->ev_side, hipEventDisableTiming) != hipSuccess)) {
      (void)hipStreamDestroy(tr->side);
      tr->side = nullptr;
    }
  }
  if (!(no_side && no_side[0] == '1')) r.side = tr->side;
  r.sscr.base = (char*)workspace + tr->tape_bytes + dx_bytes, r.sscr.cap = side_bytes;
  r.scr.base = (char*)workspace + tr->tape_bytes + dx_bytes + side_bytes;
  r.tape.cap = dx_bytes, r.scr.cap = workspace_bytes - tr->tape_bytes - dx_bytes - side_bytes;
  backward(r, params, grads, out, grad_out, grad_taps, on_group, user);
  if (r.tape.overflow || r.scr.overflow || r.sscr.overflow) return tfail(EEC_ERR_WORKSPACE, "internal: workspace carve exceeded its size");
  if (r.err != hipSuccess) return tfail((int)r.err, std::string(r.where) + ": " + hipGetErrorString(r.err));
  return 0;
}

// ---- Building blocks of the training step (the other model types of train.py:180-208: Splitformer, Early_zipformer) ---------
// The same forward / backward modules as the monolithic step above, cut at the places where those models put their own glue
// (strided slices, repeats, adds: torch ops under autograd): a GROUP of Conformer layers on given rows, the STEM (one or two
// convolutions + positional encoding) and an exit HEAD.  The entries are stateless: the recorded activations live in the
// caller's workspace, whose layout is a function of the geometry, so the backward re-derives the pointers by a dry run of the
// forward's carve (same seed / drop_prob / site_base regenerate the masks).  Everything runs on the caller's stream.
}  // extern "C"

namespace {

struct BlockSizes {
  size_t tape, fwd_scr, bwd_tape, bwd_scr, side;
  size_t total() const { return tape + std::max(fwd_scr, bwd_tape + side + bwd_scr) + 1024; }
};
size_t up256(size_t n) { return (n + 255) / 256 * 256; }

void group_forward(Run& r, const eec_layer_params* layers, int n_layers, const float* x_in, float* x_out, float* bn_mv) {
  eec_trainer* tr = r.tr;
  const int M = tr->M, D = tr->cfg.d_model;
  tr->lt.assign(n_layers, LayerTape{});
  float* x = (float*)x_in;
  static const eec_layer_params kNone{};
  carve_ffn_weights(r, n_layers);
  for (int l = 0; l < n_layers; ++l) {
    const eec_layer_params& L = layers ? layers[l] : kNone;
    LayerTape& t = tr->lt[l];
    x = layer_fwd(r, t, x, L, bn_mv ? bn_mv + (size_t)l * 2 * D : nullptr);
  }
  if (x_out) RUN(hipMemcpyAsync(x_out, x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, r.st));
}
void group_backward(Run& r, const eec_layer_params* layers, const eec_layer_params* grads, int n_layers, const float* grad_out, float* grad_in) {
  eec_trainer* tr = r.tr;
  const int M = tr->M, D = tr->cfg.d_model;
  float* dx = r.tape.f((size_t)M * D);
  if (grad_out) RUN(hipMemcpyAsync(dx, grad_out, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, r.st));
  static const eec_layer_params kNone{};
  for (int l = n_layers - 1; l >= 0; --l) {
    const eec_layer_params& L = layers ? layers[l] : kNone;
    eec_layer_params G = grads ? grads[l] : kNone;
    const LayerTape& t = tr->lt[l];
    const PreLnBwd fin{t.x4, t.fmean, t.frstd, L.final_ln_w, (float*)G.final_ln_w, (float*)G.final_ln_b};
    ffn_bwd(r, t.f2, dx, L.ffn2_ln_w, L.ffn2_w1, L.ffn2_w2, (float*)G.ffn2_ln_w, (float*)G.ffn2_ln_b, (float*)G.ffn2_w1, (float*)G.ffn2_b1,
            (float*)G.ffn2_w2, (float*)G.ffn2_b2, &fin);
    conv_bwd(r, t.cv, dx, L, G);
    attn_bwd(r, t.at, dx, L, G);
    ffn_bwd(r, t.f1, dx, L.ffn1_ln_w, L.ffn1_w1, L.ffn1_w2, (float*)G.ffn1_ln_w, (float*)G.ffn1_ln_b, (float*)G.ffn1_w1, (float*)G.ffn1_b1,
            (float*)G.ffn1_w2, (float*)G.ffn1_b2);
  }
  join_side(r);
  if (grad_in) RUN(hipMemcpyAsync(grad_in, dx, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, r.st));
}
BlockSizes group_sizes(eec_trainer tmp, int n_layers) {
  Run d{&tmp, true, nullptr};
  group_forward(d, nullptr, n_layers, nullptr, nullptr, nullptr);
  BlockSizes z{};
  z.tape = up256(d.tape.peak), z.fwd_scr = up256(d.scr.peak);
  Run b{&tmp, true, nullptr};
  group_backward(b, nullptr, nullptr, n_layers, nullptr, nullptr);
  z.bwd_tape = up256(b.tape.peak), z.bwd_scr = up256(b.scr.peak), z.side = up256(b.sscr.peak);
  return z;
}
int block_trainer(eec_trainer& tr, const eec_config* cfg, int B, int Tq, int passes, float drop_prob, uint64_t seed, const int32_t* key_len) {
  if (!cfg) return tfail(EEC_ERR_BAD_ARG, "null argument");
  if (int rc = check_trainer_cfg(*cfg)) return rc;
  if (passes != 1 && passes != 3) return tfail(EEC_ERR_BAD_ARG, "passes: 1 (bf16) or 3 (bf16x3)");
  if (!(drop_prob >= 0.0f && drop_prob < 1.0f)) return tfail(EEC_ERR_BAD_ARG, "drop_prob in [0, 1)");
  if (B <= 0 || Tq <= 0 || Tq > cfg->max_len) return tfail(EEC_ERR_BAD_ARG, "B >= 1, 1 <= T' <= max_len");
  tr.cfg = *cfg;
  tr.B = B, tr.Tq = Tq, tr.M = B * Tq, tr.np = passes, tr.p = drop_prob, tr.seed = seed, tr.key_len = (int32_t*)key_len;
  return 0;
}

// stem: Conv1d(k3, s2) [-> Conv1d(k3, s2)] -> + positional encoding -> dropout; x [B][To][D], To = T1 (one conv) or T' (two)
struct StemGeo {
  int B, C, T, T1, To, D;
  bool two;
};
struct StemTape {
  float *a1, *out1, *w2p;
};
StemTape stem_carve(Bump& t, const StemGeo& g) {
  StemTape s{};
  s.a1 = t.f((size_t)g.B * g.T1 * 3 * g.C);
  if (g.two) s.out1 = t.f((size_t)g.B * g.T1 * g.D), s.w2p = t.f((size_t)g.D * 3 * g.D);
  return s;
}
void stem_forward(Run& r, const StemGeo& g, const float* w0, const float* b0, const float* w1, const float* b1, const float* pe, const float* mel,
                  float* x, uint32_t site) {
  const StemTape s = stem_carve(r.tape, g);
  RUN(launch_im2col_mel(mel, s.a1, g.B, g.C, g.T, g.T1, r.st));
  if (g.two) {
    linear_fwd(r, s.a1, w0, b0, s.out1, g.B * g.T1, g.D, 3 * g.C);
    RUN(launch_permute_w3(w1, s.w2p, g.D, g.D, 1, r.st));
    GemmArgs a = gemm_args(s.out1, 2 * g.D, 1, s.w2p, 3 * g.D, 1, x, g.D, g.To, g.D, 3 * g.D);
    a.bias = b1, a.nz = g.B, a.zdiv = 1, a.a_z0 = (long)g.T1 * g.D, a.c_z0 = (long)g.To * g.D;
    RUN(launch_gemm(a, r.tr->np, r.st));
  } else {
    linear_fwd(r, s.a1, w0, b0, x, g.B * g.T1, g.D, 3 * g.C);
  }
  RUN(launch_add_pe_drop(x, pe, g.B, g.To, g.D, drop_of(r, site), r.st));
}
void stem_backward(Run& r, const StemGeo& g, const float* grad_x, float* g_w0, float* g_b0, float* g_w1, float* g_b1, uint32_t site) {
  const StemTape s = stem_carve(r.tape, g);
  const int B = g.B, D = g.D, T1 = g.T1, To = g.To, M = B * To;
  r.scr.reset();
  float* dx0 = r.scr.f((size_t)M * D);
  RUN(launch_scale_drop(grad_x, 1.0f, dx0, (long)M * D, drop_of(r, site), r.st));
  if (!g.two) {
    linear_bwd_weight(r, dx0, s.a1, g_w0, g_b0, M, D, 3 * g.C);
    join_side(r);
    return;
  }
  float* Gc = r.scr.f((size_t)M * 3 * D);
  float* dout1 = r.scr.f((size_t)B * T1 * D);
  float* dw2p = r.scr.f((size_t)D * 3 * D);
  {
    const size_t mark = r.scr.off;
    float* part = r.scr.f((size_t)B * D * 3 * D);
    GemmArgs a = gemm_args(dx0, 1, D, s.out1, 1, 2 * D, part, 3 * D, D, 3 * D, To);
    a.nz = B, a.zdiv = 1, a.a_z0 = (long)To * D, a.b_z0 = (long)T1 * D, a.c_z0 = (long)D * 3 * D;
    RUN(launch_gemm(a, r.tr->np, r.st));
    RUN(launch_reduce_leading(part, B, (long)D * 3 * D, (long)D * 3 * D, dw2p, r.st));
    RUN(launch_permute_w3(dw2p, g_w1, D, D, 0, r.st));
    r.scr.reset(mark);
    const int nb = colsum_blocks(M);
    float* bpart = r.scr.f((size_t)nb * D);
    RUN(launch_colsum_partial(dx0, M, D, bpart, r.st));
    RUN(launch_reduce_leading(bpart, nb, D, D, g_b1, r.st));
    r.scr.reset(mark);
  }
  {  // G[m][(j, c)] = sum_o dx0[m][o] w2p[o][(j, c)]
    GemmArgs a = gemm_args(dx0, D, 1, s.w2p, 1, 3 * D, Gc, 3 * D, M, 3 * D, D);
    RUN(launch_gemm(a, r.tr->np, r.st));
  }
  RUN(launch_col2im_stride2(Gc, dout1, B, T1, To, D, r.st));
  linear_bwd_weight(r, dout1, s.a1, g_w0, g_b0, B * T1, D, 3 * g.C);
  join_side(r);
}
int stem_geo(StemGeo& g, const eec_config* cfg, int B, int T, int two) {
  if (!cfg || B <= 0 || T < (two ? 7 : 3)) return tfail(EEC_ERR_BAD_ARG, "B >= 1 and T >= 3 (one convolution) / 7 (two)");
  const int T1 = (T - 3) / 2 + 1;
  g = StemGeo{B, cfg->n_mels, T, T1, two ? (T1 - 3) / 2 + 1 : T1, cfg->d_model, two != 0};
  if (g.To > cfg->max_len) return tfail(EEC_ERR_BAD_ARG, "output frames exceed max_len");
  return 0;
}
struct StemSizes {
  size_t tape, scr, side;
};
StemSizes stem_sizes(eec_trainer tmp, const StemGeo& g) {
  Run d{&tmp, true, nullptr};
  stem_backward(d, g, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  return StemSizes{up256(d.tape.peak), up256(d.scr.peak), up256(d.sscr.peak)};
}
int finish(const Run& r) {
  if (r.tape.overflow || r.scr.overflow || r.sscr.overflow) return tfail(EEC_ERR_WORKSPACE, "internal: workspace carve exceeded its size");
  if (r.err != hipSuccess) return tfail((int)r.err, std::string(r.where) + ": " + hipGetErrorString(r.err));
  return 0;
}

}  // namespace

extern "C" {

size_t eec_train_group_workspace_bytes(const eec_config* cfg, int n_layers, int B, int Tq) {
  eec_trainer tmp;
  if (n_layers <= 0 || n_layers > 64 || block_trainer(tmp, cfg, B, Tq, 3, 0.0f, 0, nullptr)) return 0;
  return group_sizes(tmp, n_layers).total();
}

int eec_train_group_forward(const eec_config* cfg, const eec_layer_params* layers, int n_layers, const float* x_in, const int32_t* key_len, int B,
                            int Tq, int passes, float drop_prob, uint64_t seed, uint32_t site_base, float* x_out, float* bn_batch_stats,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!layers || !x_in || !key_len || !x_out || !workspace || n_layers <= 0 || n_layers > 64) return tfail(EEC_ERR_BAD_ARG, "bad argument");
  if (((uintptr_t)workspace & 255) != 0) return tfail(EEC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  eec_trainer tr;
  if (int rc = block_trainer(tr, cfg, B, Tq, passes, drop_prob, seed, key_len)) return rc;
  const BlockSizes z = group_sizes(tr, n_layers);
  if (workspace_bytes < z.total()) return tfail(EEC_ERR_WORKSPACE, "workspace too small");
  Run r{&tr, false, (hipStream_t)stream};
  r.site = site_base;
  r.tape.base = (char*)workspace, r.tape.cap = z.tape;
  r.scr.base = (char*)workspace + z.tape, r.scr.cap = workspace_bytes - z.tape;
  group_forward(r, layers, n_layers, x_in, x_out, bn_batch_stats);
  return finish(r);
}

int eec_train_group_backward(const eec_config* cfg, const eec_layer_params* layers, const eec_layer_params* grads, int n_layers, const float* x_in,
                             const int32_t* key_len, int B, int Tq, int passes, float drop_prob, uint64_t seed, uint32_t site_base,
                             const float* grad_out, float* grad_in, void* workspace, size_t workspace_bytes, void* stream) {
  if (!layers || !grads || !x_in || !key_len || !grad_out || !grad_in || !workspace || n_layers <= 0 || n_layers > 64)
    return tfail(EEC_ERR_BAD_ARG, "bad argument");
  if (((uintptr_t)workspace & 255) != 0) return tfail(EEC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  eec_trainer tr;
  if (int rc = block_trainer(tr, cfg, B, Tq, passes, drop_prob, seed, key_len)) return rc;
  const BlockSizes z = group_sizes(tr, n_layers);
  if (workspace_bytes < z.total()) return tfail(EEC_ERR_WORKSPACE, "workspace too small");
  {  // the recorded tape's pointers: the forward's carve again, without launches
    Run d{&tr, true, nullptr};
    d.site = site_base;
    d.tape.base = (char*)workspace;
    group_forward(d, layers, n_layers, x_in, nullptr, nullptr);
  }
  Run r{&tr, false, (hipStream_t)stream};
  char* base = (char*)workspace + z.tape;
  r.tape.base = base, r.tape.cap = z.bwd_tape;
  r.sscr.base = base + z.bwd_tape, r.sscr.cap = z.side;
  r.scr.base = base + z.bwd_tape + z.side, r.scr.cap = workspace_bytes - z.tape - z.bwd_tape - z.side;
  group_backward(r, layers, grads, n_layers, grad_out, grad_in);
  return finish(r);
}

size_t eec_train_stem_workspace_bytes(const eec_config* cfg, int B, int T, int two_convs) {
  StemGeo g;
  eec_trainer tmp;
  if (stem_geo(g, cfg, B, T, two_convs) || block_trainer(tmp, cfg, B, g.To, 3, 0.0f, 0, nullptr)) return 0;
  const StemSizes z = stem_sizes(tmp, g);
  return z.tape + z.scr + z.side + 1024;
}

int eec_train_stem_forward(const eec_config* cfg, const float* sub0_w, const float* sub0_b, const float* sub1_w, const float* sub1_b,
                           const float* pe, const float* mel, int B, int T, int passes, float drop_prob, uint64_t seed, uint32_t site, float* x_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (!sub0_w || !sub0_b || !pe || !mel || !x_out || !workspace || (sub1_w != nullptr) != (sub1_b != nullptr)) return tfail(EEC_ERR_BAD_ARG, "bad argument");
  if (((uintptr_t)workspace & 255) != 0) return tfail(EEC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  StemGeo g;
  if (int rc = stem_geo(g, cfg, B, T, sub1_w != nullptr)) return rc;
  eec_trainer tr;
  if (int rc = block_trainer(tr, cfg, B, g.To, passes, drop_prob, seed, nullptr)) return rc;
  const StemSizes z = stem_sizes(tr, g);
  if (workspace_bytes < z.tape + z.scr + z.side) return tfail(EEC_ERR_WORKSPACE, "workspace too small");
  Run r{&tr, false, (hipStream_t)stream};
  r.tape.base = (char*)workspace, r.tape.cap = z.tape;
  r.scr.base = (char*)workspace + z.tape, r.scr.cap = workspace_bytes - z.tape;
  stem_forward(r, g, sub0_w, sub0_b, sub1_w, sub1_b, pe, mel, x_out, site);
  return finish(r);
}

int eec_train_stem_backward(const eec_config* cfg, int two_convs, int B, int T, int passes, float drop_prob, uint64_t seed, uint32_t site,
                            const float* grad_x, float* g_sub0_w, float* g_sub0_b, float* g_sub1_w, float* g_sub1_b, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!grad_x || !g_sub0_w || !g_sub0_b || !workspace || (two_convs && (!g_sub1_w || !g_sub1_b))) return tfail(EEC_ERR_BAD_ARG, "bad argument");
  if (((uintptr_t)workspace & 255) != 0) return tfail(EEC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  StemGeo g;
  if (int rc = stem_geo(g, cfg, B, T, two_convs)) return rc;
  eec_trainer tr;
  if (int rc = block_trainer(tr, cfg, B, g.To, passes, drop_prob, seed, nullptr)) return rc;
  const StemSizes z = stem_sizes(tr, g);
  if (workspace_bytes < z.tape + z.scr + z.side) return tfail(EEC_ERR_WORKSPACE, "workspace too small");
  Run r{&tr, false, (hipStream_t)stream};
  r.tape.base = (char*)workspace, r.tape.cap = z.tape;
  r.sscr.base = (char*)workspace + z.tape, r.sscr.cap = z.side;
  r.scr.base = (char*)workspace + z.tape + z.side, r.scr.cap = workspace_bytes - z.tape - z.side;
  stem_backward(r, g, grad_x, g_sub0_w, g_sub0_b, g_sub1_w, g_sub1_b, site);
  return finish(r);
}

/* exit head: logp = log_softmax(x . W^T + b); scratch = M * V floats */
int eec_train_head_forward(const float* x, const float* W, const float* b, int M, int V, int D, int passes, float* logp, float* scratch, void* stream) {
  if (!x || !W || !b || !logp || !scratch || M <= 0 || V <= 0 || D <= 0 || (passes != 1 && passes != 3)) return tfail(EEC_ERR_BAD_ARG, "bad argument");
  GemmArgs g = gemm_args(x, D, 1, W, D, 1, scratch, V, M, V, D);
  g.bias = b;
  hipStream_t st = (hipStream_t)stream;
  if (hipError_t e = launch_gemm(g, passes, st); e != hipSuccess) return tfail((int)e, hipGetErrorString(e));
  if (hipError_t e = launch_logsoftmax_fwd(scratch, logp, M, V, st); e != hipSuccess) return tfail((int)e, hipGetErrorString(e));
  return 0;
}
/* dx (optional) = dlogits . W, dW = dlogits^T . x, db = column sums of dlogits, with dlogits = grad_logp - exp(logp) * rowsum(grad_logp);
 * scratch: eec_train_head_backward_scratch_floats(M, V, D) floats */
size_t eec_train_head_backward_scratch_floats(int M, int V, int D) {
  if (M <= 0 || V <= 0 || D <= 0) return 0;
  eec_trainer tmp;
  Run d{&tmp, true, nullptr};
  linear_bwd_weight(d, nullptr, nullptr, nullptr, nullptr, M, V, D);
  return (size_t)M * V + 64 + d.sscr.peak / sizeof(float) + 64;
}
int eec_train_head_backward(const float* x, const float* W, const float* logp, const float* grad_logp, int M, int V, int D, int passes, float* dx,
                            float* dW, float* db, float* scratch, void* stream) {
  if (!x || !W || !logp || !grad_logp || !dW || !db || !scratch || M <= 0 || D <= 0 || (passes != 1 && passes != 3)) return tfail(EEC_ERR_BAD_ARG, "bad argument");
  if (V <= 0 || V > 256 || V % 4) return tfail(EEC_ERR_UNSUPPORTED, "the log-softmax backward needs vocab <= 256, a multiple of 4");
  eec_trainer tr;
  tr.np = passes;
  Run r{&tr, false, (hipStream_t)stream};
  float* dlogits = scratch;
  r.sscr.base = (char*)(scratch + (((size_t)M * V + 63) / 64) * 64);
  RUN(eec::launch_logsoftmax_backward(logp, grad_logp, M, V, dlogits, r.st));
  linear_bwd_weight(r, dlogits, x, dW, db, M, V, D);
  if (dx) linear_bwd_data(r, dlogits, W, dx, M, V, D);
  return finish(r);
}

int eec_train_gemm(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int passes, int a_transposed,
                   int b_transposed, void* stream) {
  if (!A || !B || !C || (passes != 1 && passes != 3)) return tfail(EEC_ERR_BAD_ARG, "bad argument");
  // a_transposed: A is stored [K][M]; b_transposed: B is stored [K][N]
  GemmArgs g = gemm_args(A, a_transposed ? 1 : K, a_transposed ? M : 1, B, b_transposed ? 1 : K, b_transposed ? N : 1, C, N, M, N, K);
  g.bias = bias;
  if (hipError_t e = launch_gemm(g, passes, (hipStream_t)stream); e != hipSuccess) return tfail((int)e, hipGetErrorString(e));
  return 0;
}

}  // extern "C"
