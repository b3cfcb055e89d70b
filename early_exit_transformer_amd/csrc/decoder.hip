// AED decoder forward behind the C ABI (include/eec.h, eec_decoder_*): `full_conformer._decoder_` of the reference
// (models/model/early_exit.py:739-762) -- token embedding + positional encoding, n_dec_layers x nn.TransformerDecoderLayer
// (batch_first, norm_first: x += SA(LN1(x), causal + target-padding mask); x += CA(LN2(x), memory); x += W2.relu(W1.LN3(x))),
// the shared final LayerNorm, the exit's output Linear and (optionally) log_softmax -- in eval mode, on the fp32 / bf16-split
// GEMM and row kernels of train_kernels.hip.  Host code only.
#include <algorithm>
#include <string>

#include "../../include/eec.h"
#include "eec_train.h"

using namespace eect;

namespace {

thread_local std::string g_derr;
int dfail(int code, const std::string& msg) {
  g_derr = msg;
  return code;
}

struct Carve {
  char* base = nullptr;
  size_t off = 0, cap = ~(size_t)0;
  bool overflow = false;
  float* f(size_t n) {
    off = (off + 255) / 256 * 256;
    float* p = (float*)(base + off);
    off += n * sizeof(float);
    if (off > cap) overflow = true;
    return p;
  }
};

struct Geo {
  int D, H, F, V, L, Bm, S, Tq, dh;
};

struct Bufs {
  float *x, *ln, *qkv, *kv, *P, *ctx, *h, *o, *logits;
  unsigned char* pad;
};
Bufs carve(Carve& c, const Geo& g) {
  Bufs b{};
  const size_t M = (size_t)g.Bm * g.S, Mk = (size_t)g.Bm * g.Tq;
  b.x = c.f(M * g.D), b.ln = c.f(M * g.D), b.qkv = c.f(M * 3 * g.D), b.kv = c.f(Mk * 2 * g.D);
  b.P = c.f((size_t)g.Bm * g.H * g.S * std::max(g.S, g.Tq));
  b.ctx = c.f(M * g.D), b.h = c.f(M * g.F), b.o = c.f(M * g.D), b.logits = c.f(M * g.V);
  b.pad = (unsigned char*)c.f((M + 3) / 4);
  return b;
}

#define DRUN(expr)                                                                          \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) return dfail((int)_e, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

}  // namespace

extern "C" {

const char* eec_decoder_last_error(void) { return g_derr.c_str(); }

size_t eec_decoder_workspace_bytes(int d_model, int n_heads, int d_ff, int vocab, int Bm, int S, int Tq) {
  if (d_model <= 0 || n_heads <= 0 || d_model % n_heads || d_ff <= 0 || vocab <= 0 || Bm <= 0 || S <= 0 || Tq <= 0) return 0;
  Carve c;
  carve(c, Geo{d_model, n_heads, d_ff, vocab, 0, Bm, S, Tq, d_model / n_heads});
  return c.off + 256;
}

int eec_decoder_forward(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, int pad_idx, const int64_t* trg,
                        const float* enc, int Bm, int S, int Tq, int enc_shared, int passes, int log_softmax, float* out,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!p || !trg || !enc || !out || !workspace || !p->layers) return dfail(EEC_ERR_BAD_ARG, "null argument");
  if (passes != 1 && passes != 3) return dfail(EEC_ERR_BAD_ARG, "passes: 1 (bf16) or 3 (bf16x3)");
  if (d_model <= 0 || d_model > 1024 || n_heads <= 0 || d_model % n_heads || vocab <= 0 || vocab > 1024 || p->n_layers <= 0 || Bm <= 0 || S <= 0 ||
      Tq <= 0 || S > p->max_len)
    return dfail(EEC_ERR_BAD_ARG, "bad geometry");
  const Geo g{d_model, n_heads, d_ff, vocab, p->n_layers, Bm, S, Tq, d_model / n_heads};
  Carve c;
  c.base = (char*)workspace, c.cap = workspace_bytes;
  const Bufs b = carve(c, g);
  if (c.overflow) return dfail(EEC_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int D = g.D, H = g.H, dh = g.dh, M = Bm * S, Mk = (enc_shared ? 1 : Bm) * Tq;
  const long mem_b = enc_shared ? 0 : (long)Tq * 2 * D;  // batch stride of the memory keys / values
  const float scale = 1.0f / sqrtf((float)dh);
  float* stats = b.h;  // LayerNorm statistics are not kept: [2][M] scratch in the h buffer, which is free whenever a LayerNorm runs
  auto linear = [&](const float* x, const float* W, const float* bias, float* y, int m, int n, int k, int epi, bool accumulate) {
    GemmArgs a = gemm_args(x, k, 1, W, k, 1, y, n, m, n, k);
    a.bias = bias, a.epi = epi, a.accumulate = accumulate ? 1 : 0;
    return launch_gemm(a, passes, st);
  };
  auto attention = [&](const float* q, long q_m, long q_b, const float* k, const float* v, long kv_m, long kv_b, int Tk, int causal,
                       const unsigned char* pad) -> hipError_t {
    {  // scores[z][tq][tk] = Q . K^T
      GemmArgs a = gemm_args(q, q_m, 1, k, kv_m, 1, b.P, Tk, S, Tk, dh);
      a.nz = Bm * H, a.zdiv = H, a.a_z0 = q_b, a.a_z1 = dh, a.b_z0 = kv_b, a.b_z1 = dh, a.c_z0 = (long)H * S * Tk, a.c_z1 = (long)S * Tk;
      if (hipError_t e = launch_gemm(a, passes, st); e != hipSuccess) return e;
    }
    if (hipError_t e = launch_softmax_masked(b.P, Bm, H, S, Tk, scale, causal, pad, st); e != hipSuccess) return e;
    GemmArgs a = gemm_args(b.P, Tk, 1, v, 1, kv_m, b.ctx, D, S, dh, Tk);
    a.nz = Bm * H, a.zdiv = H, a.a_z0 = (long)H * S * Tk, a.a_z1 = (long)S * Tk, a.b_z0 = kv_b, a.b_z1 = dh, a.c_z0 = (long)S * D, a.c_z1 = dh;
    return launch_gemm(a, passes, st);
  };
  DRUN(launch_embed_pe((const long long*)trg, p->emb, p->pe, b.x, b.pad, (long)M, S, D, vocab, pad_idx, st));
  for (int l = 0; l < p->n_layers; ++l) {
    const eec_decoder_layer_params& L = p->layers[l];
    // self-attention: causal + target key padding
    DRUN(launch_ln_fwd(b.x, L.norm1_w, L.norm1_b, b.ln, stats, stats + M, M, D, st));
    DRUN(linear(b.ln, L.sa_in_w, L.sa_in_b, b.qkv, M, 3 * D, D, 0, false));
    DRUN(attention(b.qkv, 3 * D, (long)S * 3 * D, b.qkv + D, b.qkv + 2 * D, 3 * D, (long)S * 3 * D, S, 1, b.pad));
    DRUN(linear(b.ctx, L.sa_out_w, L.sa_out_b, b.x, M, D, D, 0, true));  // x += out_proj(ctx)
    // cross-attention over the encoder output (memory), no mask
    DRUN(launch_ln_fwd(b.x, L.norm2_w, L.norm2_b, b.ln, stats, stats + M, M, D, st));
    DRUN(linear(b.ln, L.ca_in_w, L.ca_in_b, b.o, M, D, D, 0, false));                                   // q
    DRUN(linear(enc, L.ca_in_w + (size_t)D * D, L.ca_in_b + D, b.kv, Mk, 2 * D, D, 0, false));         // k | v of the memory
    DRUN(attention(b.o, D, (long)S * D, b.kv, b.kv + D, 2 * D, mem_b, Tq, 0, nullptr));
    DRUN(linear(b.ctx, L.ca_out_w, L.ca_out_b, b.x, M, D, D, 0, true));
    // feed-forward, ReLU
    DRUN(launch_ln_fwd(b.x, L.norm3_w, L.norm3_b, b.ln, stats, stats + M, M, D, st));
    DRUN(linear(b.ln, L.w1, L.b1, b.h, M, g.F, D, 3, false));
    DRUN(linear(b.h, L.w2, L.b2, b.x, M, D, g.F, 0, true));
  }
  DRUN(launch_ln_fwd(b.x, p->norm_w, p->norm_b, b.ln, stats, stats + M, M, D, st));
  if (log_softmax) {
    DRUN(linear(b.ln, p->head_w, p->head_b, b.logits, M, vocab, D, 0, false));
    DRUN(launch_logsoftmax_fwd(b.logits, out, M, vocab, st));
  } else {
    DRUN(linear(b.ln, p->head_w, p->head_b, out, M, vocab, D, 0, false));
  }
  return 0;
}

}  // extern "C"
