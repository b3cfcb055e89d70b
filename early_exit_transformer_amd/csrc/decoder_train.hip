// Training step of the AED decoder behind the C ABI (include/eec.h, eec_decoder_train_*): what autograd does through
// `full_conformer.forward`'s decoder half in train mode (models/model/early_exit.py:764-800, called from train.py:36-52 with
// --decoder_mode aed) -- token embedding + positional encoding + dropout, n_dec_layers x nn.TransformerDecoderLayer(batch_first,
// norm_first):
//     x += drop1(out_proj(MHA(LN1(x); causal + target-padding mask; attention-probability dropout)))
//     x += drop2(out_proj(MHA(LN2(x), memory = encoder output; attention-probability dropout)))
//     x += drop3(W2 . drop(relu(W1 . LN3(x))))
// the decoders' shared final LayerNorm and the exit's output Linear (raw logits: the reference comments the log_softmax out,
// early_exit.py:790) -- and its backward: the gradient of every decoder parameter, of the embedding table and of the encoder
// output the decoder attended to.  Same design as the encoder's training step (train.hip): fp32 tensors in HBM, parameters read
// in place, activations the backward needs recorded in a caller-owned workspace, every GEMM on the bf16-split MFMA kernel of
// train_kernels.hip (passes 3: hi.hi + hi.lo + lo.hi, ~fp32 results; 1: plain bf16), weight gradients split over the rows and
// summed in a fixed order, no atomics.  The entries are stateless: the tape's layout is a function of the geometry alone, so the
// backward re-derives it (same seed / drop_prob / exit_index as the forward regenerate the dropout masks).  Host code only.
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/eec.h"
#include "eec_train.h"

using namespace eect;

namespace {

thread_local std::string g_dterr;
int dtfail(int code, const std::string& msg) {
  g_dterr = msg;
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
    if (off > cap) overflow = true;
    return p;
  }
  void reset(size_t to = 0) { off = to; }
};

struct Geo {
  int D, H, F, V, L, Bm, S, Tq, dh;
  long M, Mk;
};

struct LayerTape {
  float *x0, *ln1, *m1, *r1, *qkv, *Psa, *Pdsa, *ctxsa;
  float *x1, *ln2, *m2, *r2, *q, *kv, *Pca, *Pdca, *ctxca;
  float *x2, *ln3, *m3, *r3, *pre, *act;
};
struct Tape {
  unsigned char* pad;
  std::vector<LayerTape> lt;
  float *xf, *lnf, *mf, *rf;
};

// the recorded activations; identical for the sizing pass, the forward and the backward (nothing depends on the settings: the
// dropped probability copies are carved whatever drop_prob is)
Tape carve_tape(Bump& t, const Geo& g, bool drop) {
  Tape tp;
  const size_t M = g.M, Mk = g.Mk, D = g.D;
  const size_t psa = (size_t)g.Bm * g.H * g.S * g.S, pca = (size_t)g.Bm * g.H * g.S * g.Tq;
  tp.pad = (unsigned char*)t.f((M + 3) / 4);
  tp.lt.resize(g.L);
  float* x = t.f(M * D);
  for (int l = 0; l < g.L; ++l) {
    LayerTape& a = tp.lt[l];
    a.x0 = x;
    a.ln1 = t.f(M * D), a.m1 = t.f(M), a.r1 = t.f(M), a.qkv = t.f(M * 3 * D), a.Psa = t.f(psa);
    float* pd = t.f(psa);
    a.Pdsa = drop ? pd : a.Psa;
    a.ctxsa = t.f(M * D), a.x1 = t.f(M * D);
    a.ln2 = t.f(M * D), a.m2 = t.f(M), a.r2 = t.f(M), a.q = t.f(M * D), a.kv = t.f(Mk * 2 * D), a.Pca = t.f(pca);
    pd = t.f(pca);
    a.Pdca = drop ? pd : a.Pca;
    a.ctxca = t.f(M * D), a.x2 = t.f(M * D);
    a.ln3 = t.f(M * D), a.m3 = t.f(M), a.r3 = t.f(M), a.pre = t.f(M * g.F), a.act = t.f(M * g.F);
    x = t.f(M * D);  // the layer's output = the next layer's x0
  }
  tp.xf = x, tp.lnf = t.f(M * D), tp.mf = t.f(M), tp.rf = t.f(M);
  return tp;
}

// dropout sites: 0 = positional encoding (the reference embeds the targets ONCE for all exits: the same mask for every exit of a
// forward); the others are private to (exit, layer, place)
enum { kSitePsa = 0, kSiteRes1, kSitePca, kSiteRes2, kSiteAct, kSiteRes3, kSitesPerLayer };
uint32_t site_of(int exit_index, int layer, int place) { return 1u + (uint32_t)exit_index * 1024u + (uint32_t)layer * kSitesPerLayer + (uint32_t)place; }

struct Run {
  Geo g;
  int np;
  float p;
  uint64_t seed;
  int exit_index;
  bool dry;
  hipStream_t st;
  Bump scr;
  hipError_t err = hipSuccess;
  const char* where = "";
  void ok(hipError_t e, const char* w) {
    if (e != hipSuccess && err == hipSuccess) err = e, where = w;
  }
  Drop drop(int layer, int place) const { return Drop{p, seed, site_of(exit_index, layer, place)}; }
  Drop drop_pe() const { return Drop{p, seed, 0u}; }
};
#define RUN(expr)                    \
  do {                               \
    if (!r.dry) r.ok((expr), #expr); \
  } while (0)

void linear_fwd(Run& r, const float* x, const float* W, const float* bias, float* y, long M, int N, int K) {
  GemmArgs g = gemm_args(x, K, 1, W, K, 1, y, N, (int)M, N, K);
  g.bias = bias;
  RUN(launch_gemm(g, r.np, r.st));
}
// y = res + drop(x . W^T + bias)
void linear_residual_fwd(Run& r, const float* x, const float* W, const float* bias, const float* res, Drop d, float* y, long M, int N, int K) {
  GemmArgs g = gemm_args(x, K, 1, W, K, 1, y, N, (int)M, N, K);
  g.bias = bias, g.epi = 4, g.aux = res, g.res_scale = 1.0f, g.drop = d;
  RUN(launch_gemm(g, r.np, r.st));
}
// dx[M][K] (+)= dy[M][N] . W[N][K]
void linear_bwd_data(Run& r, const float* dy, const float* W, float* dx, long M, int N, int K, bool accumulate = false) {
  GemmArgs g = gemm_args(dy, N, 1, W, 1, K, dx, K, (int)M, K, N);
  g.accumulate = accumulate;
  RUN(launch_gemm(g, r.np, r.st));
}
// dW[N][K] = dy[M][N]^T . x[M][K] (split over the rows, partials summed in a fixed order); db[N] = column sums of dy
void linear_bwd_weight(Run& r, const float* dy, const float* x, float* dW, float* db, long Ml, int N, int K) {
  const int M = (int)Ml;
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  int S = std::max(1, std::min(512 / tiles, M / 64));
  int chunk = ((M + S - 1) / S + 31) / 32 * 32;
  S = (M + chunk - 1) / chunk;
  const size_t mark = r.scr.off;
  GemmArgs g = gemm_args(dy, 1, N, x, 1, K, dW, K, N, K, chunk);
  if (S > 1) {
    const long pstride = (long)N * K + N;  // one split's partial: [N][K] weight gradient, then [N] bias gradient
    float* part = r.scr.f((size_t)S * pstride);
    g.C = part, g.nz = S, g.zdiv = 1, g.ktot = M;
    g.a_z0 = (long)chunk * N, g.b_z0 = (long)chunk * K, g.c_z0 = pstride;
    g.rowsum = part + (long)N * K, g.rowsum_z = pstride;
    RUN(launch_gemm(g, r.np, r.st));
    RUN(launch_reduce_leading_split(part, S, pstride, pstride, (long)N * K, dW, db, r.st));
  } else {
    g.K = M;
    g.rowsum = db, g.rowsum_z = 0;
    RUN(launch_gemm(g, r.np, r.st));
  }
  r.scr.reset(mark);
}
// dx = (add_res ? dx : 0) + LN'(dln); dg / db from the per-block partials
void ln_bwd(Run& r, const float* dln, const float* x, const float* g, const float* mean, const float* rstd, float* dx, bool add_res, float* dg,
            float* db, long M, int D) {
  const size_t mark = r.scr.off;
  const int nb = ln_bwd_blocks((int)M);
  float* part = r.scr.f((size_t)nb * 2 * D);
  RUN(launch_ln_bwd(dln, x, g, mean, rstd, add_res ? dx : nullptr, dx, part, (int)M, D, r.st));
  RUN(launch_reduce_leading2(part, nb, D, dg, db, r.st));
  r.scr.reset(mark);
}

// operands of one attention: queries [Bm*S] rows of stride q_m (batch stride q_b), keys / values [Bm*Tk] rows of stride kv_m
struct AttnOps {
  const float *q, *k, *v;
  long q_m, q_b, kv_m, kv_b;
  int Tk;
};
void batched(GemmArgs& a, const Geo& g, long az0, long az1, long bz0, long bz1, long cz0, long cz1) {
  a.nz = g.Bm * g.H, a.zdiv = g.H;
  a.a_z0 = az0, a.a_z1 = az1, a.b_z0 = bz0, a.b_z1 = bz1, a.c_z0 = cz0, a.c_z1 = cz1;
}

void attention_fwd(Run& r, const AttnOps& o, float* P, float* Pd, float* ctx, int causal, const unsigned char* pad, Drop d) {
  const Geo& g = r.g;
  const long pb = (long)g.H * g.S * o.Tk, ph = (long)g.S * o.Tk;
  {  // scores[z][tq][tk] = Q . K^T
    GemmArgs a = gemm_args(o.q, o.q_m, 1, o.k, o.kv_m, 1, P, o.Tk, g.S, o.Tk, g.dh);
    batched(a, g, o.q_b, g.dh, o.kv_b, g.dh, pb, ph);
    RUN(launch_gemm(a, r.np, r.st));
  }
  RUN(launch_softmax_masked_drop(P, Pd != P ? Pd : nullptr, g.Bm, g.H, g.S, o.Tk, 1.0f / sqrtf((float)g.dh), causal, pad, d, r.st));
  {  // ctx = Pd . V
    GemmArgs a = gemm_args(Pd, o.Tk, 1, o.v, 1, o.kv_m, ctx, g.D, g.S, g.dh, o.Tk);
    batched(a, g, pb, ph, o.kv_b, g.dh, (long)g.S * g.D, g.dh);
    RUN(launch_gemm(a, r.np, r.st));
  }
}
// gradients of one attention: dq rows (stride q_m), dk / dv rows (stride kv_m); dP is [Bm*H][S][Tk] scratch
void attention_bwd(Run& r, const AttnOps& o, const float* P, const float* Pd, const float* dctx, float* dP, float* dq, float* dk, float* dv, Drop d) {
  const Geo& g = r.g;
  const int S = g.S, Tk = o.Tk, dh = g.dh, D = g.D;
  const long pb = (long)g.H * S * Tk, ph = (long)S * Tk, xb = (long)S * D;
  {  // dV[tk][d] = sum_tq Pd[tq][tk] dctx[tq][d]
    GemmArgs a = gemm_args(Pd, 1, Tk, dctx, 1, D, dv, o.kv_m, Tk, dh, S);
    batched(a, g, pb, ph, xb, dh, o.kv_b, dh);
    RUN(launch_gemm(a, r.np, r.st));
  }
  {  // dPd[tq][tk] = sum_d dctx[tq][d] V[tk][d]
    GemmArgs a = gemm_args(dctx, D, 1, o.v, o.kv_m, 1, dP, Tk, S, Tk, dh);
    batched(a, g, xb, dh, o.kv_b, dh, pb, ph);
    RUN(launch_gemm(a, r.np, r.st));
  }
  RUN(launch_softmax_bwd_rows(P, dP, (long)g.Bm * g.H * S, Tk, 1.0f / sqrtf((float)dh), d, r.st));
  {  // dQ[tq][d] = sum_tk dS[tq][tk] K[tk][d]
    GemmArgs a = gemm_args(dP, Tk, 1, o.k, 1, o.kv_m, dq, o.q_m, S, dh, Tk);
    batched(a, g, pb, ph, o.kv_b, dh, o.q_b, dh);
    RUN(launch_gemm(a, r.np, r.st));
  }
  {  // dK[tk][d] = sum_tq dS[tq][tk] Q[tq][d]
    GemmArgs a = gemm_args(dP, 1, Tk, o.q, 1, o.q_m, dk, o.kv_m, Tk, dh, S);
    batched(a, g, pb, ph, o.q_b, dh, o.kv_b, dh);
    RUN(launch_gemm(a, r.np, r.st));
  }
}

void forward(Run& r, Bump& tb, const eec_decoder_params* p, int pad_idx, const int64_t* trg, const float* enc, float* out) {
  const Geo& g = r.g;
  const int D = g.D, S = g.S, Tq = g.Tq;
  const long M = g.M, Mk = g.Mk;
  Tape t = carve_tape(tb, g, r.p > 0.0f);
  RUN(launch_embed_pe_drop((const long long*)trg, p->emb, p->pe, r.dry ? nullptr : t.lt[0].x0, t.pad, M, S, D, g.V, pad_idx, r.drop_pe(), r.st));
  for (int l = 0; l < g.L; ++l) {
    const eec_decoder_layer_params& L = p->layers ? p->layers[l] : eec_decoder_layer_params{};
    const LayerTape& a = t.lt[l];
    // self-attention: causal + target key padding
    RUN(launch_ln_fwd(a.x0, L.norm1_w, L.norm1_b, a.ln1, a.m1, a.r1, (int)M, D, r.st));
    linear_fwd(r, a.ln1, L.sa_in_w, L.sa_in_b, a.qkv, M, 3 * D, D);
    attention_fwd(r, AttnOps{a.qkv, a.qkv + D, a.qkv + 2 * D, 3L * D, (long)S * 3 * D, 3L * D, (long)S * 3 * D, S}, a.Psa, a.Pdsa, a.ctxsa, 1, t.pad,
                  r.drop(l, kSitePsa));
    linear_residual_fwd(r, a.ctxsa, L.sa_out_w, L.sa_out_b, a.x0, r.drop(l, kSiteRes1), a.x1, M, D, D);
    // cross-attention over the encoder output (memory), no mask
    RUN(launch_ln_fwd(a.x1, L.norm2_w, L.norm2_b, a.ln2, a.m2, a.r2, (int)M, D, r.st));
    linear_fwd(r, a.ln2, L.ca_in_w, L.ca_in_b, a.q, M, D, D);
    linear_fwd(r, enc, L.ca_in_w ? L.ca_in_w + (size_t)D * D : nullptr, L.ca_in_b ? L.ca_in_b + D : nullptr, a.kv, Mk, 2 * D, D);
    attention_fwd(r, AttnOps{a.q, a.kv, a.kv + D, (long)D, (long)S * D, 2L * D, (long)Tq * 2 * D, Tq}, a.Pca, a.Pdca, a.ctxca, 0, nullptr,
                  r.drop(l, kSitePca));
    linear_residual_fwd(r, a.ctxca, L.ca_out_w, L.ca_out_b, a.x1, r.drop(l, kSiteRes2), a.x2, M, D, D);
    // feed-forward: pre = W1 . LN3(x) + b1 and, in the same epilogue, act = drop(relu(pre))
    RUN(launch_ln_fwd(a.x2, L.norm3_w, L.norm3_b, a.ln3, a.m3, a.r3, (int)M, D, r.st));
    {
      GemmArgs ga = gemm_args(a.ln3, D, 1, L.w1, D, 1, a.pre, g.F, (int)M, g.F, D);
      ga.bias = L.b1, ga.epi = 5, ga.C2 = a.act, ga.drop = r.drop(l, kSiteAct);
      RUN(launch_gemm(ga, r.np, r.st));
    }
    float* xn = l + 1 < g.L ? t.lt[l + 1].x0 : t.xf;
    linear_residual_fwd(r, a.act, L.w2, L.b2, a.x2, r.drop(l, kSiteRes3), xn, M, D, g.F);
  }
  RUN(launch_ln_fwd(t.xf, p->norm_w, p->norm_b, t.lnf, t.mf, t.rf, (int)M, D, r.st));
  linear_fwd(r, t.lnf, p->head_w, p->head_b, out, M, g.V, D);
}

void backward(Run& r, Bump& tb, const eec_decoder_params* p, const eec_decoder_params* G, const int64_t* trg, const float* enc,
              const float* grad_out, float* grad_enc) {
  const Geo& g = r.g;
  const int D = g.D, S = g.S, Tq = g.Tq, F = g.F;
  const long M = g.M, Mk = g.Mk;
  Tape t = carve_tape(tb, g, r.p > 0.0f);
  r.scr.reset();
  float* dx = r.scr.f(M * D);
  float* dy = r.scr.f(M * D);      // the (masked) gradient of a sub-module's output
  float* dln = r.scr.f(M * D);
  float* dctx = r.scr.f(M * D);
  float* dqkv = r.scr.f(M * 3 * D);  // self-attention: dQ | dK | dV rows; cross-attention: its first D columns' worth as dq [M][D]
  float* dkv = r.scr.f(Mk * 2 * D);
  float* dP = r.scr.f((size_t)g.Bm * g.H * S * std::max(S, Tq));
  float* dpre = r.scr.f(M * F);
  // head and the shared final LayerNorm
  linear_bwd_weight(r, grad_out, t.lnf, (float*)G->head_w, (float*)G->head_b, M, g.V, D);
  linear_bwd_data(r, grad_out, p->head_w, dln, M, g.V, D);
  ln_bwd(r, dln, t.xf, p->norm_w, t.mf, t.rf, dx, false, (float*)G->norm_w, (float*)G->norm_b, M, D);
  for (int l = g.L - 1; l >= 0; --l) {
    const eec_decoder_layer_params& L = p->layers ? p->layers[l] : eec_decoder_layer_params{};
    const eec_decoder_layer_params Gl = G->layers ? G->layers[l] : eec_decoder_layer_params{};
    const LayerTape& a = t.lt[l];
    // ---- feed-forward
    RUN(launch_scale_drop(dx, 1.0f, dy, M * D, r.drop(l, kSiteRes3), r.st));
    linear_bwd_weight(r, dy, a.act, (float*)Gl.w2, (float*)Gl.b2, M, D, F);
    {  // dpre = (dy . W2) * dropmask * relu'(pre)
      GemmArgs ga = gemm_args(dy, D, 1, L.w2, 1, F, dpre, F, (int)M, F, D);
      ga.epi = 6, ga.aux = a.pre, ga.drop = r.drop(l, kSiteAct);
      RUN(launch_gemm(ga, r.np, r.st));
    }
    linear_bwd_weight(r, dpre, a.ln3, (float*)Gl.w1, (float*)Gl.b1, M, F, D);
    linear_bwd_data(r, dpre, L.w1, dln, M, F, D);
    ln_bwd(r, dln, a.x2, L.norm3_w, a.m3, a.r3, dx, true, (float*)Gl.norm3_w, (float*)Gl.norm3_b, M, D);
    // ---- cross-attention
    RUN(launch_scale_drop(dx, 1.0f, dy, M * D, r.drop(l, kSiteRes2), r.st));
    linear_bwd_weight(r, dy, a.ctxca, (float*)Gl.ca_out_w, (float*)Gl.ca_out_b, M, D, D);
    linear_bwd_data(r, dy, L.ca_out_w, dctx, M, D, D);
    {
      float* dq = dqkv;  // [M][D]
      attention_bwd(r, AttnOps{a.q, a.kv, a.kv + D, (long)D, (long)S * D, 2L * D, (long)Tq * 2 * D, Tq}, a.Pca, a.Pdca, dctx, dP, dq, dkv, dkv + D,
                    r.drop(l, kSitePca));
      float* gw = (float*)Gl.ca_in_w;
      float* gb = (float*)Gl.ca_in_b;
      linear_bwd_weight(r, dq, a.ln2, gw, gb, M, D, D);                                                         // the query third of in_proj
      linear_bwd_weight(r, dkv, enc, gw ? gw + (size_t)D * D : nullptr, gb ? gb + D : nullptr, Mk, 2 * D, D);  // the key / value thirds
      linear_bwd_data(r, dq, L.ca_in_w, dln, M, D, D);
      linear_bwd_data(r, dkv, L.ca_in_w ? L.ca_in_w + (size_t)D * D : nullptr, grad_enc, Mk, 2 * D, D, l != g.L - 1);  // into the memory
    }
    ln_bwd(r, dln, a.x1, L.norm2_w, a.m2, a.r2, dx, true, (float*)Gl.norm2_w, (float*)Gl.norm2_b, M, D);
    // ---- self-attention
    RUN(launch_scale_drop(dx, 1.0f, dy, M * D, r.drop(l, kSiteRes1), r.st));
    linear_bwd_weight(r, dy, a.ctxsa, (float*)Gl.sa_out_w, (float*)Gl.sa_out_b, M, D, D);
    linear_bwd_data(r, dy, L.sa_out_w, dctx, M, D, D);
    attention_bwd(r, AttnOps{a.qkv, a.qkv + D, a.qkv + 2 * D, 3L * D, (long)S * 3 * D, 3L * D, (long)S * 3 * D, S}, a.Psa, a.Pdsa, dctx, dP, dqkv,
                  dqkv + D, dqkv + 2 * D, r.drop(l, kSitePsa));
    linear_bwd_weight(r, dqkv, a.ln1, (float*)Gl.sa_in_w, (float*)Gl.sa_in_b, M, 3 * D, D);
    linear_bwd_data(r, dqkv, L.sa_in_w, dln, M, 3 * D, D);
    ln_bwd(r, dln, a.x0, L.norm1_w, a.m1, a.r1, dx, true, (float*)Gl.norm1_w, (float*)Gl.norm1_b, M, D);
  }
  // embedding table (the positional encoding is a buffer)
  RUN(launch_embed_bwd((const long long*)trg, dx, (float*)G->emb, M, D, g.V, r.drop_pe(), r.st));
}

int check_geo(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, int Bm, int S, int Tq, int passes, float drop_prob) {
  if (!p) return dtfail(EEC_ERR_BAD_ARG, "null argument");
  if (passes != 1 && passes != 3) return dtfail(EEC_ERR_BAD_ARG, "passes: 1 (bf16) or 3 (bf16x3)");
  if (!(drop_prob >= 0.0f && drop_prob < 1.0f)) return dtfail(EEC_ERR_BAD_ARG, "drop_prob in [0, 1)");
  if (d_model <= 0 || d_model > 1024 || n_heads <= 0 || d_model % n_heads || d_ff <= 0 || vocab <= 0 || vocab > 1024 || p->n_layers <= 0 ||
      p->n_layers > 64 || Bm <= 0 || S <= 0 || Tq <= 0 || S > p->max_len)
    return dtfail(EEC_ERR_BAD_ARG, "bad geometry");
  return 0;
}
Geo make_geo(int d_model, int n_heads, int d_ff, int vocab, int n_layers, int Bm, int S, int Tq) {
  return Geo{d_model, n_heads, d_ff, vocab, n_layers, Bm, S, Tq, d_model / n_heads, (long)Bm * S, (long)Bm * Tq};
}
struct Sizes {
  size_t tape, scratch;
};
Sizes sizes_of(const Geo& g) {
  Bump tb;
  carve_tape(tb, g, true);
  Run r{g, 3, 0.0f, 0, 0, true, nullptr};
  Bump tb2;
  eec_decoder_params none{};
  none.n_layers = g.L;
  backward(r, tb2, &none, &none, nullptr, nullptr, nullptr, nullptr);
  return Sizes{(tb.peak + 255) / 256 * 256, (r.scr.peak + 255) / 256 * 256};
}

}  // namespace

extern "C" {

const char* eec_decoder_train_last_error(void) { return g_dterr.c_str(); }

size_t eec_decoder_train_workspace_bytes(int d_model, int n_heads, int d_ff, int vocab, int n_layers, int Bm, int S, int Tq) {
  if (d_model <= 0 || n_heads <= 0 || d_model % n_heads || d_ff <= 0 || vocab <= 0 || n_layers <= 0 || n_layers > 64 || Bm <= 0 || S <= 0 || Tq <= 0)
    return 0;
  const Sizes s = sizes_of(make_geo(d_model, n_heads, d_ff, vocab, n_layers, Bm, S, Tq));
  return s.tape + s.scratch + 512;
}

int eec_decoder_train_forward(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, int pad_idx, const int64_t* trg,
                              const float* enc, int Bm, int S, int Tq, int passes, float drop_prob, uint64_t seed, int exit_index, float* out,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (!p || !trg || !enc || !out || !workspace || !p->layers) return dtfail(EEC_ERR_BAD_ARG, "null argument");
  if (int rc = check_geo(p, d_model, n_heads, d_ff, vocab, Bm, S, Tq, passes, drop_prob)) return rc;
  if (((uintptr_t)workspace & 255) != 0) return dtfail(EEC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  const Geo g = make_geo(d_model, n_heads, d_ff, vocab, p->n_layers, Bm, S, Tq);
  const Sizes sz = sizes_of(g);
  if (workspace_bytes < sz.tape + sz.scratch) return dtfail(EEC_ERR_WORKSPACE, "workspace too small");
  Run r{g, passes, drop_prob, seed, exit_index, false, (hipStream_t)stream};
  Bump tb;
  tb.base = (char*)workspace, tb.cap = sz.tape;
  r.scr.base = (char*)workspace + sz.tape, r.scr.cap = workspace_bytes - sz.tape;
  forward(r, tb, p, pad_idx, trg, enc, out);
  if (tb.overflow || r.scr.overflow) return dtfail(EEC_ERR_WORKSPACE, "internal: workspace carve exceeded its size");
  if (r.err != hipSuccess) return dtfail((int)r.err, std::string(r.where) + ": " + hipGetErrorString(r.err));
  return 0;
}

int eec_decoder_train_backward(const eec_decoder_params* p, const eec_decoder_params* grads, int d_model, int n_heads, int d_ff, int vocab,
                               const int64_t* trg, const float* enc, int Bm, int S, int Tq, int passes, float drop_prob, uint64_t seed,
                               int exit_index, const float* grad_out, float* grad_enc, void* workspace, size_t workspace_bytes, void* stream) {
  if (!p || !grads || !trg || !enc || !grad_out || !grad_enc || !workspace || !p->layers || !grads->layers)
    return dtfail(EEC_ERR_BAD_ARG, "null argument");
  if (int rc = check_geo(p, d_model, n_heads, d_ff, vocab, Bm, S, Tq, passes, drop_prob)) return rc;
  if (grads->n_layers != p->n_layers) return dtfail(EEC_ERR_BAD_ARG, "grads must mirror params");
  if (((uintptr_t)workspace & 255) != 0) return dtfail(EEC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  const Geo g = make_geo(d_model, n_heads, d_ff, vocab, p->n_layers, Bm, S, Tq);
  const Sizes sz = sizes_of(g);
  if (workspace_bytes < sz.tape + sz.scratch) return dtfail(EEC_ERR_WORKSPACE, "workspace too small");
  Run r{g, passes, drop_prob, seed, exit_index, false, (hipStream_t)stream};
  Bump tb;
  tb.base = (char*)workspace, tb.cap = sz.tape;
  r.scr.base = (char*)workspace + sz.tape, r.scr.cap = workspace_bytes - sz.tape;
  backward(r, tb, p, grads, trg, enc, grad_out, grad_enc);
  if (tb.overflow || r.scr.overflow) return dtfail(EEC_ERR_WORKSPACE, "internal: workspace carve exceeded its size");
  if (r.err != hipSuccess) return dtfail((int)r.err, std::string(r.where) + ": " + hipGetErrorString(r.err));
  return 0;
}

}  // extern "C"
