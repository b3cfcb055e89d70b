// C-ABI of libeec.so (include/eec.h): parameter packing, workspace carving and the
// launch plan of one encoder forward.  Host code only; kernels live in the other .hip files.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/eec.h"
#include <cstdlib>
#include <cstring>

#include "eec_kernels.h"

using namespace eec;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return (int)e;
}
#define EEC_HIP(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return hip_fail(_e, #expr);  \
  } while (0)

struct PackedLayer {
  // device pointers into the encoder's arena
  float *ffn1_ln_w, *ffn1_ln_b, *ffn1_b1, *ffn1_b2;
  uint4 *ffn1_w1p, *ffn1_w2p, *ffn1_w1f8, *ffn1_w2f8;
  float *attn_ln_w, *attn_ln_b, *attn_in_b, *attn_out_b;
  uint4 *attn_in_p, *attn_out_p, *attn_in_f8, *attn_out_f8;
  float *conv_ln_w, *conv_ln_b, *conv_pw1_b, *conv_pw2_b, *dw_wfold, *dw_bfold;
  uint4 *conv_pw1_p, *conv_pw2_p, *conv_pw1_f8, *conv_pw2_f8;
  float *ffn2_ln_w, *ffn2_ln_b, *ffn2_b1, *ffn2_b2;
  uint4 *ffn2_w1p, *ffn2_w2p, *ffn2_w1f8, *ffn2_w2f8;
  float *final_ln_w, *final_ln_b;
};

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Arena {  // bump allocator over one hipMalloc
  char* base = nullptr;
  size_t cap = 0, off = 0;
  template <typename T>
  T* take(size_t count) {
    off = align_up(off);
    T* p = (T*)(base ? base + off : nullptr);
    off += count * sizeof(T);
    return p;
  }
};

size_t frag_u4(int N, int K) { return (size_t)((N + 31) / 32) * (K / 16) * 128; }
size_t f8_u4(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 63) / 64) * kF8Rec; }

}  // namespace

enum KernelClass { KC_STEM = 0, KC_FFN, KC_QKV, KC_ATTN, KC_PROJ_GLU, KC_PROJ, KC_DW_PW2, KC_HEAD, KC_CHAIN, KC_COUNT };

struct eec_encoder {
  eec_config cfg;
  int device = -1;  // the HIP device the packed-weight arena lives on: every entry point must run with it current
  Arena arena;
  bool packed = false;
  bool has_stem = false, has_stem1 = false, has_heads = false;  // eec_encoder_pack may be given layers only (building-block use)
  // optional per-kernel-class timing with HIP events on the launch stream (bench/roofline only)
  bool profiling = false;
  std::vector<hipEvent_t> ev;       // pairs: start, stop
  std::vector<int> ev_class;        // class of each recorded pair
  size_t ev_used = 0;
  double prof_ms[KC_COUNT] = {0};
  long long prof_n[KC_COUNT] = {0};
  std::vector<PackedLayer> layers;
  uint4 *sub_w1p, *sub_w2p;
  float *sub_b1, *sub_b2, *pe;
  std::vector<uint4*> head_p, head_f8;
  std::vector<float*> head_b;

  void carve() {
    const int D = cfg.d_model, F = cfg.d_ff, nl = cfg.n_exits * cfg.layers_per_exit;
    layers.assign(nl, PackedLayer());
    for (auto& L : layers) {
      L.ffn1_ln_w = arena.take<float>(D);
      L.ffn1_ln_b = arena.take<float>(D);
      L.ffn1_b1 = arena.take<float>(F);
      L.ffn1_b2 = arena.take<float>(D);
      L.ffn1_w1p = arena.take<uint4>(frag_u4(F, D));
      L.ffn1_w2p = arena.take<uint4>(frag_u4(D, F));
      L.ffn1_w1f8 = arena.take<uint4>(f8_u4(F, D));
      L.ffn1_w2f8 = arena.take<uint4>(f8_u4(D, F));
      L.attn_ln_w = arena.take<float>(D);
      L.attn_ln_b = arena.take<float>(D);
      L.attn_in_b = arena.take<float>(3 * D);
      L.attn_out_b = arena.take<float>(D);
      L.attn_in_p = arena.take<uint4>(frag_u4(3 * D, D));
      L.attn_out_p = arena.take<uint4>(frag_u4(D, D));
      L.attn_in_f8 = arena.take<uint4>(f8_u4(3 * D, D));
      L.attn_out_f8 = arena.take<uint4>(f8_u4(D, D));
      L.conv_ln_w = arena.take<float>(D);
      L.conv_ln_b = arena.take<float>(D);
      L.conv_pw1_b = arena.take<float>(2 * D);
      L.conv_pw2_b = arena.take<float>(D);
      L.dw_wfold = arena.take<float>(31 * D);
      L.dw_bfold = arena.take<float>(D);
      L.conv_pw1_p = arena.take<uint4>(frag_u4(2 * D, D));
      L.conv_pw2_p = arena.take<uint4>(frag_u4(D, D));
      L.conv_pw1_f8 = arena.take<uint4>(f8_u4(2 * D, D));
      L.conv_pw2_f8 = arena.take<uint4>(f8_u4(D, D));
      L.ffn2_ln_w = arena.take<float>(D);
      L.ffn2_ln_b = arena.take<float>(D);
      L.ffn2_b1 = arena.take<float>(F);
      L.ffn2_b2 = arena.take<float>(D);
      L.ffn2_w1p = arena.take<uint4>(frag_u4(F, D));
      L.ffn2_w2p = arena.take<uint4>(frag_u4(D, F));
      L.ffn2_w1f8 = arena.take<uint4>(f8_u4(F, D));
      L.ffn2_w2f8 = arena.take<uint4>(f8_u4(D, F));
      L.final_ln_w = arena.take<float>(D);
      L.final_ln_b = arena.take<float>(D);
    }
    sub_w1p = arena.take<uint4>(frag_u4(D, cfg.n_mels * 3));
    sub_b1 = arena.take<float>(D);
    sub_w2p = arena.take<uint4>(frag_u4(D, 3 * D));
    sub_b2 = arena.take<float>(D);
    pe = arena.take<float>((size_t)cfg.max_len * D);
    head_p.assign(cfg.n_exits, nullptr);
    head_f8.assign(cfg.n_exits, nullptr);
    head_b.assign(cfg.n_exits, nullptr);
    for (int e = 0; e < cfg.n_exits; ++e) {
      head_p[e] = arena.take<uint4>(frag_u4(cfg.vocab, D));
      head_f8[e] = arena.take<uint4>(f8_u4(cfg.vocab, D));
      head_b[e] = arena.take<float>(cfg.vocab);
    }
    arena.off = align_up(arena.off);
  }
};

namespace {

struct Workspace {
  float* x;
  float* y;  // [(E-1)][M][256]: exit rows for the batched head launch when the caller passes no tap buffer
  half_t *mid_hi, *mid_lo, *q, *k, *vt, *p_hi, *p_lo, *g;
  int *enc_len, *mid_e;
  size_t bytes;
};

Workspace carve_ws(const eec_config& c, int B, int T, char* base) {
  const int T1 = (T - 3) / 2 + 1, Tq = (T1 - 3) / 2 + 1, Tp = (Tq + 31) / 32 * 32;
  const size_t M = (size_t)B * Tq, D = c.d_model;
  Arena a;
  a.base = base;
  Workspace w;
  w.x = a.take<float>(M * D);
  w.y = a.take<float>((size_t)(c.n_exits > 1 ? c.n_exits - 1 : 0) * M * D);
  w.mid_hi = a.take<half_t>((size_t)B * T1 * D);
  w.mid_lo = a.take<half_t>((size_t)B * T1 * D);
  w.q = a.take<half_t>((size_t)2 * B * Tp * D);   // hi plane, then the residual plane (exact mode f16x3 only)
  w.k = a.take<half_t>((size_t)2 * B * Tp * D);
  w.vt = a.take<half_t>((size_t)2 * B * Tp * D);  // hi plane, then the residual plane
  w.p_hi = a.take<half_t>(M * D);
  w.p_lo = a.take<half_t>(M * D);
  w.g = a.take<half_t>(M * D);
  w.enc_len = a.take<int>(B);
  w.mid_e = a.take<int>((size_t)B * T1);
  w.bytes = align_up(a.off);
  return w;
}

// One device per encoder handle: the arena (and the caller's workspace) are plain device allocations, never peer-mapped.
int check_device(const eec_encoder* enc) {
  int dev = -1;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return hip_fail(e, "hipGetDevice");
  if (dev != enc->device)
    return fail(EEC_ERR_BAD_ARG, "encoder was created on device " + std::to_string(enc->device) + " but device " +
                                     std::to_string(dev) + " is current (one handle per device; re-create it after a move)");
  return 0;
}

int check_cfg(const eec_config& c) {
  if (c.arch != EEC_ARCH_CONFORMER && c.arch != EEC_ARCH_LEGACY) return fail(EEC_ERR_BAD_ARG, "unknown arch");
  if (c.d_model != 256 && c.d_model != 512) return fail(EEC_ERR_UNSUPPORTED, "d_model must be 256 or 512 in this build");
  if (c.n_heads <= 0 || c.d_model % c.n_heads) return fail(EEC_ERR_BAD_ARG, "n_heads must divide d_model");
  const int dh = c.d_model / c.n_heads;
  if (dh != 32 && dh != 64) return fail(EEC_ERR_UNSUPPORTED, "head dim must be 32 or 64");
  if (c.d_ff <= 0 || c.d_ff % 32) return fail(EEC_ERR_UNSUPPORTED, "d_ff must be a positive multiple of 32");
  if (c.arch == EEC_ARCH_CONFORMER && (c.dw_kernel < 1 || c.dw_kernel > 31 || !(c.dw_kernel & 1)))
    return fail(EEC_ERR_UNSUPPORTED, "depthwise kernel must be odd and <= 31");
  if (c.vocab <= 0 || c.vocab > 256 || c.vocab % 32) return fail(EEC_ERR_UNSUPPORTED, "vocab must be a multiple of 32, <= 256");
  if ((c.n_mels * 3) % 16 || c.n_mels * 3 > 384)
    return fail(EEC_ERR_UNSUPPORTED, "n_mels*3 must be a multiple of 16 and <= 384 (n_mels in {16,32,48,64,80,96,112,128})");
  if (c.n_exits <= 0 || c.layers_per_exit <= 0 || c.n_mels <= 0 || c.max_len <= 0)
    return fail(EEC_ERR_BAD_ARG, "n_exits, layers_per_exit, n_mels, max_len must be positive");
  return 0;
}

}  // namespace

// TIMED(class, launch-expression): brackets the launch with events when profiling is on (uses `enc` and `st` of the caller)
#define TIMED(cls, expr)                                                        \
  do {                                                                          \
    const bool _p = enc->profiling && (enc->ev_used + 1) * 2 <= enc->ev.size(); \
    if (_p) EEC_HIP(hipEventRecord(enc->ev[enc->ev_used * 2], st));             \
    EEC_HIP(expr);                                                              \
    if (_p) {                                                                   \
      EEC_HIP(hipEventRecord(enc->ev[enc->ev_used * 2 + 1], st));               \
      enc->ev_class[enc->ev_used++] = (cls);                                    \
    }                                                                           \
  } while (0)

// The production launch plan of Conformer layers [l0, l1) on rows x (in place): 3 launches per layer.  Everything that
// is local to a 64-row tile runs in ONE chain kernel per layer boundary,
//   [depthwise + pointwise-2 of layer l] -> ffn2(l) + final LN (+ tap) -> ffn1(l+1) -> in_proj(l+1),
// and only the two steps with cross-tile dependencies keep their own launch: attention (all keys of the utterance) and
// out_proj + LN + pointwise-1 + GLU (whose output the depthwise conv reads with a +-15 frame halo).  tap_of(li) says
// where layer li's output is stored besides x (nullptr: nowhere; x itself moves on to ffn1 of layer li + 1).
struct LayerBufs {
  float* x;
  half_t *q, *k, *vt, *vt_lo, *p_hi, *p_lo, *g;
  const int* key_len;
  half_t *q_lo = nullptr, *k_lo = nullptr;  // exact mode (f16x3): Q and K keep their fp16 residuals, attention runs as its own launch
};
struct LayerFormats {
  int ffn, front, qkv, att, glu;  // operand formats of the GEMM groups (1, 3 or 8)
};
template <typename TapFn>
static int run_layer_plan(eec_encoder* enc, int l0, int l1, const LayerBufs& b, int B, int Tq, const LayerFormats& np,
                          TapFn tap_of, hipStream_t st) {
  const eec_config& c = enc->cfg;
  const int Tp = (Tq + 31) / 32 * 32, M = B * Tq, D = c.d_model, H = c.n_heads;
  // the attention of a row tile runs in the prologue of the out_proj / GLU launch when the shape allows it -- a property of the
  // call, not of the layer (exact mode: Q / K residual planes present -> its three-product form, which needs glu format 3); the
  // chain tail then writes V^T fragment-major for it
  AttnArgs at_shape{b.q, b.k, b.vt, b.key_len, B, H, Tq, Tp, D / H, b.p_hi, b.p_lo, b.vt_lo};
  const bool fused_attn = attn_fusable(at_shape, D) && (np.att != 1) == (np.glu != 1) && (!b.q_lo || np.glu == 3);
  auto qkv_args = [&](const PackedLayer& L) {
    QkvArgs q{b.x, M, B, Tq, Tp, H, D, L.attn_ln_w, L.attn_ln_b, L.attn_in_p, L.attn_in_b, b.q, b.k, b.vt, b.vt_lo, L.attn_in_f8};
    q.q_lo = b.q_lo, q.k_lo = b.k_lo;
    q.vt_frag = fused_attn && D == 256 && Tq % 64 == 0 ? 1 : 0;  // qkv_body's whole-line store path is taken for exactly these shapes
    return q;
  };
  auto stage1 = [&](const PackedLayer& L) {
    return FfnStage{L.ffn1_ln_w, L.ffn1_ln_b, L.ffn1_w1p, L.ffn1_b1, L.ffn1_w2p, L.ffn1_b2, nullptr, nullptr,
                    L.ffn1_w1f8, L.ffn1_w2f8, 0.5f, nullptr};
  };
  {
    ChainArgs ca{};
    ca.x = b.x, ca.M = M, ca.F = c.d_ff, ca.D = D, ca.nstage = 1, ca.Tq = Tq;
    ca.st[0] = stage1(enc->layers[l0]);
    ca.qkv = qkv_args(enc->layers[l0]);
    TIMED(KC_CHAIN, launch_ffn_chain(ca, np.ffn, np.front, np.qkv, false, true, false, st));
  }
  for (int li = l0; li < l1; ++li) {
    const PackedLayer& L = enc->layers[li];
    const bool last = li + 1 == l1;
    AttnArgs at{b.q, b.k, b.vt, b.key_len, B, H, Tq, Tp, D / H, b.p_hi, b.p_lo, b.vt_lo};
    at.q_lo = b.q_lo, at.k_lo = b.k_lo;
    at.vt_frag = fused_attn && D == 256 && Tq % 64 == 0 ? 1 : 0;
    ProjResArgs pr{b.x, M, D, b.p_hi, b.p_lo, L.attn_out_p, L.attn_out_b, L.attn_out_f8};
    GluArgs ga{b.x, M, L.conv_ln_w, L.conv_ln_b, L.conv_pw1_p, L.conv_pw1_b, b.g, L.conv_pw1_f8};
    if (fused_attn) {
      // two launches per layer: the attention of a row tile runs in the prologue of the out_proj / GLU kernel
      TIMED(KC_PROJ_GLU, launch_attn_proj_glu(at, pr, ga, np.glu, st));
    } else {
      TIMED(KC_ATTN, launch_attention(at, np.att == 1 ? 1 : 3, st));
      TIMED(KC_PROJ_GLU, launch_proj_glu(pr, ga, np.glu, st));
    }
    ChainArgs ca{};
    ca.x = b.x, ca.M = M, ca.F = c.d_ff, ca.D = D, ca.nstage = last ? 1 : 2, ca.Tq = Tq;
    ca.dw = DwArgs{b.g, B, Tq, L.dw_wfold, L.dw_bfold, b.p_hi, b.p_lo};
    ca.pw2 = ProjResArgs{b.x, M, D, nullptr, nullptr, L.conv_pw2_p, L.conv_pw2_b, L.conv_pw2_f8};
    ca.st[0] = FfnStage{L.ffn2_ln_w, L.ffn2_ln_b, L.ffn2_w1p, L.ffn2_b1, L.ffn2_w2p, L.ffn2_b2, L.final_ln_w, L.final_ln_b,
                        L.ffn2_w1f8, L.ffn2_w2f8, 0.5f, tap_of(li)};
    if (!last) {
      ca.st[1] = stage1(enc->layers[li + 1]);
      ca.qkv = qkv_args(enc->layers[li + 1]);
    }
    TIMED(KC_CHAIN, launch_ffn_chain(ca, np.ffn, np.front, np.qkv, true, !last, false, st));
  }
  return 0;
}


extern "C" {

const char* eec_last_error(void) { return g_err.c_str(); }
int eec_abi_version(void) { return EEC_ABI_VERSION; }

int eec_out_frames(int T) {
  if (T < 7) return 0;
  const int T1 = (T - 3) / 2 + 1;
  return (T1 - 3) / 2 + 1;
}

int eec_encoder_lengths(const int64_t* lengths, int B, int Tq, int32_t* enc_len, void* stream) {
  if (!lengths || !enc_len || B <= 0 || Tq <= 0) return fail(EEC_ERR_BAD_ARG, "bad argument");
  EEC_HIP(launch_enc_lengths((const long long*)lengths, B, Tq, enc_len, (hipStream_t)stream));
  return 0;
}

int eec_upload_i64_max(void) { return kUploadMax; }

int eec_upload_i64(const int64_t* host, int n, int64_t* dev, void* stream) {
  if (!host || !dev || n <= 0 || n > kUploadMax) return fail(EEC_ERR_BAD_ARG, "eec_upload_i64: 1 .. 480 values");
  EEC_HIP(launch_upload_i64((const long long*)host, n, (long long*)dev, (hipStream_t)stream));
  return 0;
}

int eec_encoder_create(const eec_config* cfg, eec_encoder** out) {
  if (!cfg || !out) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (int rc = check_cfg(*cfg)) return rc;
  eec_encoder* enc = new eec_encoder();
  enc->cfg = *cfg;
  if (hipError_t e = hipGetDevice(&enc->device); e != hipSuccess) {
    delete enc;
    return hip_fail(e, "hipGetDevice");
  }
  enc->carve();  // dry run: sizes only
  const size_t need = enc->arena.off;
  void* mem = nullptr;
  hipError_t e = hipMalloc(&mem, need);
  if (e != hipSuccess) {
    delete enc;
    return hip_fail(e, "hipMalloc(packed weights)");
  }
  enc->arena = Arena();
  enc->arena.base = (char*)mem;
  enc->arena.cap = need;
  enc->carve();
  *out = enc;
  return 0;
}

void eec_encoder_destroy(eec_encoder* enc) {
  if (!enc) return;
  for (hipEvent_t e : enc->ev) (void)hipEventDestroy(e);
  if (enc->arena.base) (void)hipFree(enc->arena.base);
  delete enc;
}

int eec_encoder_pack(eec_encoder* enc, const eec_params* p, void* stream) {
  if (!enc || !p || !p->layers) return fail(EEC_ERR_BAD_ARG, "null argument");
  if ((p->head_w == nullptr) != (p->head_b == nullptr)) return fail(EEC_ERR_BAD_ARG, "head_w and head_b go together");
  if (enc->cfg.arch != EEC_ARCH_CONFORMER) return fail(EEC_ERR_BAD_ARG, "use eec_encoder_pack_legacy for EEC_ARCH_LEGACY");
  if (int rc = check_device(enc)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const eec_config& c = enc->cfg;
  const int D = c.d_model, F = c.d_ff;
  auto cp = [&](float* dst, const float* src, size_t n) {
    return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
  };
  for (size_t i = 0; i < enc->layers.size(); ++i) {
    const eec_layer_params& s = p->layers[i];
    PackedLayer& L = enc->layers[i];
    EEC_HIP(cp(L.ffn1_ln_w, s.ffn1_ln_w, D));
    EEC_HIP(cp(L.ffn1_ln_b, s.ffn1_ln_b, D));
    EEC_HIP(launch_scale_copy(s.ffn1_b1, L.ffn1_b1, F, kLog2e, st));
    EEC_HIP(cp(L.ffn1_b2, s.ffn1_b2, D));
    EEC_HIP(launch_pack_frags(s.ffn1_w1, F, D, L.ffn1_w1p, kLog2e, st));
    EEC_HIP(launch_pack_frags(s.ffn1_w2, D, F, L.ffn1_w2p, 1.0f / kLog2e, st));
    if (F % 128 == 0) {
      EEC_HIP(launch_pack_frags_f8(s.ffn1_w1, F, D, L.ffn1_w1f8, kLog2e, st));
      EEC_HIP(launch_pack_frags_f8(s.ffn1_w2, D, F, L.ffn1_w2f8, 1.0f / kLog2e, st));
    }
    EEC_HIP(cp(L.attn_ln_w, s.attn_ln_w, D));
    EEC_HIP(cp(L.attn_ln_b, s.attn_ln_b, D));
    EEC_HIP(cp(L.attn_in_b, s.attn_in_b, 3 * D));
    EEC_HIP(cp(L.attn_out_b, s.attn_out_b, D));
    EEC_HIP(launch_pack_frags(s.attn_in_w, 3 * D, D, L.attn_in_p, 1.0f, st));
    EEC_HIP(launch_pack_frags(s.attn_out_w, D, D, L.attn_out_p, 1.0f, st));
    EEC_HIP(launch_pack_frags_f8(s.attn_in_w, 3 * D, D, L.attn_in_f8, 1.0f, st));
    EEC_HIP(launch_pack_frags_f8(s.attn_out_w, D, D, L.attn_out_f8, 1.0f, st));
    EEC_HIP(cp(L.conv_ln_w, s.conv_ln_w, D));
    EEC_HIP(cp(L.conv_ln_b, s.conv_ln_b, D));
    EEC_HIP(cp(L.conv_pw1_b, s.conv_pw1_b, 2 * D));
    EEC_HIP(cp(L.conv_pw2_b, s.conv_pw2_b, D));
    EEC_HIP(launch_pack_frags(s.conv_pw1_w, 2 * D, D, L.conv_pw1_p, 1.0f, st));
    EEC_HIP(launch_pack_frags(s.conv_pw2_w, D, D, L.conv_pw2_p, 1.0f, st));
    EEC_HIP(launch_pack_frags_f8(s.conv_pw1_w, 2 * D, D, L.conv_pw1_f8, 1.0f, st));
    EEC_HIP(launch_pack_frags_f8(s.conv_pw2_w, D, D, L.conv_pw2_f8, 1.0f, st));
    EEC_HIP(launch_fold_dw(s.conv_dw_w, s.conv_dw_b, s.conv_bn_w, s.conv_bn_b, s.conv_bn_rm, s.conv_bn_rv,
                           c.dw_kernel, D, L.dw_wfold, L.dw_bfold, st));
    EEC_HIP(cp(L.ffn2_ln_w, s.ffn2_ln_w, D));
    EEC_HIP(cp(L.ffn2_ln_b, s.ffn2_ln_b, D));
    EEC_HIP(launch_scale_copy(s.ffn2_b1, L.ffn2_b1, F, kLog2e, st));
    EEC_HIP(cp(L.ffn2_b2, s.ffn2_b2, D));
    EEC_HIP(launch_pack_frags(s.ffn2_w1, F, D, L.ffn2_w1p, kLog2e, st));
    EEC_HIP(launch_pack_frags(s.ffn2_w2, D, F, L.ffn2_w2p, 1.0f / kLog2e, st));
    if (F % 128 == 0) {
      EEC_HIP(launch_pack_frags_f8(s.ffn2_w1, F, D, L.ffn2_w1f8, kLog2e, st));
      EEC_HIP(launch_pack_frags_f8(s.ffn2_w2, D, F, L.ffn2_w2f8, 1.0f / kLog2e, st));
    }
    EEC_HIP(cp(L.final_ln_w, s.final_ln_w, D));
    EEC_HIP(cp(L.final_ln_b, s.final_ln_b, D));
  }
  enc->has_stem = p->sub0_w && p->sub0_b && p->sub1_w && p->sub1_b && p->pe;
  enc->has_stem1 = p->sub0_w && p->sub0_b && p->pe;  // enough for the one-convolution stem (eec_encoder_stem1_forward)
  if (enc->has_stem1 && !enc->has_stem) {
    EEC_HIP(launch_pack_frags(p->sub0_w, D, c.n_mels * 3, enc->sub_w1p, 1.0f, st));
    EEC_HIP(cp(enc->sub_b1, p->sub0_b, D));
    EEC_HIP(cp(enc->pe, p->pe, (size_t)c.max_len * D));
  }
  if (enc->has_stem) {
    EEC_HIP(launch_pack_frags(p->sub0_w, D, c.n_mels * 3, enc->sub_w1p, 1.0f, st));
    EEC_HIP(cp(enc->sub_b1, p->sub0_b, D));
    EEC_HIP(launch_pack_conv_jci(p->sub1_w, D, D, enc->sub_w2p, st));
    EEC_HIP(cp(enc->sub_b2, p->sub1_b, D));
    EEC_HIP(cp(enc->pe, p->pe, (size_t)c.max_len * D));
  }
  enc->has_heads = p->head_w != nullptr;
  if (enc->has_heads)
    for (int e = 0; e < c.n_exits; ++e) {
      if (!p->head_w[e] || !p->head_b[e]) return fail(EEC_ERR_BAD_ARG, "null head parameter");
      EEC_HIP(launch_pack_frags(p->head_w[e], c.vocab, D, enc->head_p[e], 1.0f, st));
      EEC_HIP(launch_pack_frags_f8(p->head_w[e], c.vocab, D, enc->head_f8[e], 1.0f, st));
      EEC_HIP(cp(enc->head_b[e], p->head_b[e], c.vocab));
    }
  enc->packed = true;
  return 0;
}

int eec_encoder_pack_legacy(eec_encoder* enc, const eec_legacy_params* p, void* stream) {
  if (!enc || !p || !p->layers || !p->group_ln_w || !p->group_ln_b || !p->head_w || !p->head_b)
    return fail(EEC_ERR_BAD_ARG, "null argument");
  if (enc->cfg.arch != EEC_ARCH_LEGACY) return fail(EEC_ERR_BAD_ARG, "encoder was not created with EEC_ARCH_LEGACY");
  if (int rc = check_device(enc)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const eec_config& c = enc->cfg;
  const int D = c.d_model, F = c.d_ff;
  auto cp = [&](float* dst, const float* src, size_t n) {
    return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
  };
  const size_t tile8 = (size_t)(D / 32) * (D / 16) * 128;  // the D/32 n-tiles (D output columns) of one of w_q / w_k / w_v
  for (size_t i = 0; i < enc->layers.size(); ++i) {
    const eec_legacy_layer_params& s = p->layers[i];
    PackedLayer& L = enc->layers[i];
    const int e = (int)i / c.layers_per_exit;
    EEC_HIP(cp(L.attn_ln_w, s.norm1_w, D));
    EEC_HIP(cp(L.attn_ln_b, s.norm1_b, D));
    // w_q / w_k / w_v become one [768][256] in_proj
    EEC_HIP(launch_pack_frags(s.wq, D, D, L.attn_in_p, 1.0f, st));
    EEC_HIP(launch_pack_frags(s.wk, D, D, L.attn_in_p + tile8, 1.0f, st));
    EEC_HIP(launch_pack_frags(s.wv, D, D, L.attn_in_p + 2 * tile8, 1.0f, st));
    EEC_HIP(cp(L.attn_in_b, s.bq, D));
    EEC_HIP(cp(L.attn_in_b + D, s.bk, D));
    EEC_HIP(cp(L.attn_in_b + 2 * D, s.bv, D));
    EEC_HIP(launch_pack_frags(s.wo, D, D, L.attn_out_p, 1.0f, st));
    EEC_HIP(cp(L.attn_out_b, s.bo, D));
    EEC_HIP(cp(L.ffn2_ln_w, s.norm2_w, D));
    EEC_HIP(cp(L.ffn2_ln_b, s.norm2_b, D));
    EEC_HIP(launch_pack_frags(s.w1, F, D, L.ffn2_w1p, 1.0f, st));
    EEC_HIP(cp(L.ffn2_b1, s.b1, F));
    EEC_HIP(launch_pack_frags(s.w2, D, F, L.ffn2_w2p, 1.0f, st));
    if (F % 128 == 0) {
      EEC_HIP(launch_pack_frags_f8(s.w1, F, D, L.ffn2_w1f8, 1.0f, st));
      EEC_HIP(launch_pack_frags_f8(s.w2, D, F, L.ffn2_w2f8, 1.0f, st));
    }
    EEC_HIP(cp(L.ffn2_b2, s.b2, D));
    EEC_HIP(cp(L.final_ln_w, p->group_ln_w[e], D));
    EEC_HIP(cp(L.final_ln_b, p->group_ln_b[e], D));
  }
  enc->has_stem = p->sub0_w && p->sub0_b && p->sub1_w && p->sub1_b && p->pe;
  enc->has_stem1 = p->sub0_w && p->sub0_b && p->pe;  // enough for the one-convolution stem (eec_encoder_stem1_forward)
  if (enc->has_stem1 && !enc->has_stem) {
    EEC_HIP(launch_pack_frags(p->sub0_w, D, c.n_mels * 3, enc->sub_w1p, 1.0f, st));
    EEC_HIP(cp(enc->sub_b1, p->sub0_b, D));
    EEC_HIP(cp(enc->pe, p->pe, (size_t)c.max_len * D));
  }
  if (enc->has_stem) {
    EEC_HIP(launch_pack_frags(p->sub0_w, D, c.n_mels * 3, enc->sub_w1p, 1.0f, st));
    EEC_HIP(cp(enc->sub_b1, p->sub0_b, D));
    EEC_HIP(launch_pack_conv_jci(p->sub1_w, D, D, enc->sub_w2p, st));
    EEC_HIP(cp(enc->sub_b2, p->sub1_b, D));
    EEC_HIP(cp(enc->pe, p->pe, (size_t)c.max_len * D));
  }
  enc->has_heads = p->head_w != nullptr;
  if (enc->has_heads)
    for (int e = 0; e < c.n_exits; ++e) {
      if (!p->head_w[e] || !p->head_b[e]) return fail(EEC_ERR_BAD_ARG, "null head parameter");
      EEC_HIP(launch_pack_frags(p->head_w[e], c.vocab, D, enc->head_p[e], 1.0f, st));
      EEC_HIP(launch_pack_frags_f8(p->head_w[e], c.vocab, D, enc->head_f8[e], 1.0f, st));
      EEC_HIP(cp(enc->head_b[e], p->head_b[e], c.vocab));
    }
  enc->packed = true;
  return 0;
}

size_t eec_encoder_workspace_bytes(const eec_encoder* enc, int B, int T) {
  if (!enc || B <= 0 || T < 7) return 0;
  return carve_ws(enc->cfg, B, T, nullptr).bytes;
}

// n_groups: exit groups to run (1 .. E); the production plan then ends after the last layer of group n_groups
static int forward_impl(eec_encoder* enc, const float* mel, const int64_t* lengths, int B, int T, int precision,
                        float* out, float* taps_opt, void* workspace, size_t workspace_bytes, int stop_after,
                        float* x_dbg_opt, int n_groups, void* stream) {
  if (!enc || !mel || !lengths || !workspace) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (!out && !taps_opt && !x_dbg_opt && stop_after < 0) return fail(EEC_ERR_BAD_ARG, "no output buffer given");
  if (n_groups < 1 || n_groups > enc->cfg.n_exits) return fail(EEC_ERR_BAD_ARG, "n_groups must be in 1 .. n_exits");
  if (enc->packed && enc->cfg.arch == EEC_ARCH_CONFORMER && (!enc->has_stem || (out && !enc->has_heads)))
    return fail(EEC_ERR_NOT_PACKED, "this encoder was packed without stem / head parameters");
  if (!enc->packed) return fail(EEC_ERR_NOT_PACKED, "eec_encoder_pack has not been called");
  if (int rc = check_device(enc)) return rc;
  if (B <= 0 || T < 7) return fail(EEC_ERR_BAD_ARG, "need B > 0 and T >= 7 (two k=3 s=2 convs)");
  if (precision < EEC_PREC_F16X3 || precision > EEC_PREC_F16F8) return fail(EEC_ERR_BAD_ARG, "unknown precision");
  const eec_config& c = enc->cfg;
  const int T1 = (T - 3) / 2 + 1, Tq = (T1 - 3) / 2 + 1, Tp = (Tq + 31) / 32 * 32;
  if (Tq > c.max_len) return fail(EEC_ERR_BAD_ARG, "T' exceeds the positional-encoding table (max_len)");
  if (((uintptr_t)workspace & 255) != 0) return fail(EEC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  Workspace ws = carve_ws(c, B, T, (char*)workspace);
  if (workspace_bytes < ws.bytes) return fail(EEC_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int np_ffn = precision == EEC_PREC_F16X3 ? 3 : (precision == EEC_PREC_F16F8 ? 8 : 1);
  const int np_o = precision == EEC_PREC_F16 ? 1 : 3;
  // f16f8 (default): besides the feed-forward, out_proj + pointwise-1, pointwise-2 and (round 3) in_proj run on the f8
  // stream (fp16 + two fp8 correction products); the exit heads (their error is the logit error) keep the exact 3-pass
  // split.  Measured on the default model, 3 weight / input seeds (profiles/r02_np_budget.txt): max |dlogp| 3.6e-4 (2.5e-4
  // with in_proj on the 3-pass split: its error is amplified by the softmax, but the chain kernel's in_proj tail is
  // MFMA-bound at three passes: -8 us of 169 per launch, profiles/r03_ab_chain_knobs.txt), heads too 4.0e-4
  // (d_model 512 keeps the fragment formats there: its f8 projection kernels do not fit the register file yet)
  const int np_p = (precision == EEC_PREC_F16F8 && c.arch == EEC_ARCH_CONFORMER && c.d_model == 256 && c.d_ff % 128 == 0) ? 8 : np_o;
  // operand format per GEMM group of the production plan (a diagnostic build can override them one by one)
  int np_qkv = np_p, np_att = np_o, np_glu = np_p, np_front = np_p, np_head = np_o;
  half_t* const vt_lo = np_o == 3 ? ws.vt + (size_t)B * Tp * c.d_model : nullptr;  // V keeps its fp16 residual in the split modes
  // exact parity mode (f16x3): Q, K and the attention probabilities keep their fp16 residuals as well (three MFMA products per
  // attention product, attention as its own launch): what remains of the log-prob error is the GLU output's fp16 rounding
  const bool exact_attn = precision == EEC_PREC_F16X3 && c.arch == EEC_ARCH_CONFORMER;
  half_t* const q_lo = exact_attn ? ws.q + (size_t)B * Tp * c.d_model : nullptr;
  half_t* const k_lo = exact_attn ? ws.k + (size_t)B * Tp * c.d_model : nullptr;
#ifdef EEC_NP_EXPERIMENT
  if (const char* ov = getenv("EEC_NP_OVERRIDE")) {  // e.g. "qkv=1,glu=1": error-budget experiments (tools/np_budget.py)
    auto pick = [&](const char* key, int& dst) {
      const char* p = strstr(ov, key);
      if (p) dst = atoi(p + strlen(key));
    };
    pick("qkv=", np_qkv), pick("att=", np_att), pick("glu=", np_glu), pick("front=", np_front), pick("head=", np_head);
  }
#endif
  const int M = B * Tq, D = c.d_model, H = c.n_heads;
  int step = 0;
  auto done = [&](void) -> bool { return stop_after >= 0 && step > stop_after; };
  auto finish_dbg = [&]() -> int {
    if (x_dbg_opt) EEC_HIP(hipMemcpyAsync(x_dbg_opt, ws.x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
    return 0;
  };


  EEC_HIP(launch_enc_lengths((const long long*)lengths, B, Tq, ws.enc_len, st));
  if (Tp != Tq) EEC_HIP(hipMemsetAsync(ws.vt, 0, (size_t)2 * B * Tp * D * sizeof(half_t), st));
  {
    SubsampleArgs a{mel, B, c.n_mels, T, T1, Tq, D, ws.mid_e, enc->sub_w1p, enc->sub_b1, enc->sub_w2p, enc->sub_b2, enc->pe, ws.mid_hi, ws.mid_lo, ws.x};
    TIMED(KC_STEM, launch_subsample(a, 3, st));  // raw power mel: always hi/lo split (1 % of the flops)
  }
  ++step;
  if (done()) return finish_dbg();

  if (c.arch == EEC_ARCH_LEGACY) {
    // Early_encoder: every key is valid (mask=None): overwrite the clamp(lengths/4) result with T'
    EEC_HIP(launch_fill_int(ws.enc_len, B, Tq, st));
    for (int e = 0; e < n_groups; ++e) {
      for (int l = 0; l < c.layers_per_exit; ++l) {
        const PackedLayer& L = enc->layers[e * c.layers_per_exit + l];
        {
          QkvArgs a{ws.x, M, B, Tq, Tp, H, D, L.attn_ln_w, L.attn_ln_b, L.attn_in_p, L.attn_in_b, ws.q, ws.k, ws.vt, vt_lo};
          TIMED(KC_QKV, launch_qkv(a, np_o, st));
          AttnArgs at{ws.q, ws.k, ws.vt, ws.enc_len, B, H, Tq, Tp, D / H, ws.p_hi, ws.p_lo, vt_lo};
          TIMED(KC_ATTN, launch_attention(at, np_o, st));
          ProjResArgs pr{ws.x, M, D, ws.p_hi, ws.p_lo, L.attn_out_p, L.attn_out_b};
          TIMED(KC_PROJ, launch_proj_residual(pr, np_o, st));
        }
        ++step;
        if (done()) return finish_dbg();
        {
          const bool last = l == c.layers_per_exit - 1;  // Encoder.layer_norm closes the group
          FfnArgs a{ws.x, M, c.d_ff, D, L.ffn2_ln_w, L.ffn2_ln_b, L.ffn2_w1p, L.ffn2_b1, L.ffn2_w2p, L.ffn2_b2,
                    last ? L.final_ln_w : nullptr, last ? L.final_ln_b : nullptr, 1.0f, true};
          a.w1f8 = L.ffn2_w1f8, a.w2f8 = L.ffn2_w2f8;
          TIMED(KC_FFN, launch_ffn(a, np_ffn, st));
        }
        ++step;
        if (done()) return finish_dbg();
      }
      if (out) {
        HeadArgs h{ws.x, M, c.vocab, D, enc->head_p[e], enc->head_b[e], out + (size_t)e * M * c.vocab};
        TIMED(KC_HEAD, launch_head(h, np_o, st));
      }
      if (taps_opt)
        EEC_HIP(hipMemcpyAsync(taps_opt + (size_t)e * M * D, ws.x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
    }
    return finish_dbg();
  }

  if (stop_after < 0) {
    // production plan (run_layer_plan); the sub-step hook (stop_after >= 0) uses the unfused plan below; both are parity-tested
    const int n_layers = n_groups * c.layers_per_exit;
    const bool batch_heads = out && n_groups <= kMaxHeadExits;  // all exit heads in ONE launch after the last layer
    // where exit e's encoder output goes besides x: the caller's tap buffer, else (if a head needs it later) workspace rows
    auto tap_of = [&](int li) -> float* {
      if ((li + 1) % c.layers_per_exit != 0) return nullptr;
      const int e = li / c.layers_per_exit;
      if (taps_opt) return taps_opt + (size_t)e * M * D;
      return (out && li + 1 != n_layers) ? ws.y + (size_t)e * M * D : nullptr;
    };
    LayerBufs bufs{ws.x, ws.q, ws.k, ws.vt, vt_lo, ws.p_hi, ws.p_lo, ws.g, ws.enc_len};
    bufs.q_lo = q_lo, bufs.k_lo = k_lo;
    const LayerFormats nps{np_ffn, np_front, np_qkv, np_att, np_glu};
    if (int rc = run_layer_plan(enc, 0, n_layers, bufs, B, Tq, nps, tap_of, st)) return rc;
    if (out) {
      HeadBatchArgs hb{};
      hb.out = out, hb.M = M, hb.V = c.vocab, hb.E = n_groups, hb.D = D;
      for (int e = 0; e < n_groups; ++e) {
        const int li = (e + 1) * c.layers_per_exit - 1;
        const float* rows = li + 1 == n_layers ? ws.x : tap_of(li);
        if (batch_heads) {
          hb.x[e] = rows, hb.wp[e] = enc->head_p[e], hb.wf8[e] = enc->head_f8[e], hb.bias[e] = enc->head_b[e];
        } else {
          HeadArgs h{rows, M, c.vocab, D, enc->head_p[e], enc->head_b[e], out + (size_t)e * M * c.vocab, enc->head_f8[e]};
          TIMED(KC_HEAD, launch_head(h, np_head, st));
        }
      }
      if (batch_heads) TIMED(KC_HEAD, launch_head_batch(hb, np_head, st));
    }
    return finish_dbg();
  }

  for (int e = 0; e < n_groups; ++e) {
    for (int l = 0; l < c.layers_per_exit; ++l) {
      const PackedLayer& L = enc->layers[e * c.layers_per_exit + l];
      {
        FfnArgs a{ws.x, M, c.d_ff, D, L.ffn1_ln_w, L.ffn1_ln_b, L.ffn1_w1p, L.ffn1_b1, L.ffn1_w2p, L.ffn1_b2, nullptr, nullptr};
        a.w1f8 = L.ffn1_w1f8, a.w2f8 = L.ffn1_w2f8;
        TIMED(KC_FFN, launch_ffn(a, np_ffn, st));
      }
      ++step;
      if (done()) return finish_dbg();
      {
        QkvArgs a{ws.x, M, B, Tq, Tp, H, D, L.attn_ln_w, L.attn_ln_b, L.attn_in_p, L.attn_in_b, ws.q, ws.k, ws.vt, vt_lo, L.attn_in_f8};
        a.q_lo = q_lo, a.k_lo = k_lo;
        TIMED(KC_QKV, launch_qkv(a, np_qkv, st));
        AttnArgs at{ws.q, ws.k, ws.vt, ws.enc_len, B, H, Tq, Tp, D / H, ws.p_hi, ws.p_lo, vt_lo};
        at.q_lo = q_lo, at.k_lo = k_lo;
        TIMED(KC_ATTN, launch_attention(at, np_o, st));
        ProjResArgs pr{ws.x, M, D, ws.p_hi, ws.p_lo, L.attn_out_p, L.attn_out_b, L.attn_out_f8};
        GluArgs ga{ws.x, M, L.conv_ln_w, L.conv_ln_b, L.conv_pw1_p, L.conv_pw1_b, ws.g, L.conv_pw1_f8};
        // out-proj + residual and the conv module's LN -> pointwise-1 -> GLU share one launch
        // (the sub-step hook stops between them, so it falls back to the two separate kernels)
        const bool split_here = stop_after >= 0 && step + 1 > stop_after;
        if (split_here) {
          TIMED(KC_PROJ, launch_proj_residual(pr, np_o, st));
        } else {
          TIMED(KC_PROJ_GLU, launch_proj_glu(pr, ga, np_p, st));
        }
        ++step;
        if (done()) return finish_dbg();
      }
      {
        DwArgs da{ws.g, B, Tq, L.dw_wfold, L.dw_bfold, ws.p_hi, ws.p_lo};
        ProjResArgs pr{ws.x, M, D, nullptr, nullptr, L.conv_pw2_p, L.conv_pw2_b, L.conv_pw2_f8};
        TIMED(KC_DW_PW2, launch_dw_pw2(da, pr, np_p, st));
      }
      ++step;
      if (done()) return finish_dbg();
      {
        FfnArgs a{ws.x, M, c.d_ff, D, L.ffn2_ln_w, L.ffn2_ln_b, L.ffn2_w1p, L.ffn2_b1, L.ffn2_w2p, L.ffn2_b2, L.final_ln_w, L.final_ln_b};
        a.w1f8 = L.ffn2_w1f8, a.w2f8 = L.ffn2_w2f8;
        TIMED(KC_FFN, launch_ffn(a, np_ffn, st));
      }
      ++step;
      if (done()) return finish_dbg();
    }
    if (out) {
      HeadArgs h{ws.x, M, c.vocab, D, enc->head_p[e], enc->head_b[e], out + (size_t)e * M * c.vocab, enc->head_f8[e]};
      TIMED(KC_HEAD, launch_head(h, np_o, st));
    }
    if (taps_opt)
      EEC_HIP(hipMemcpyAsync(taps_opt + (size_t)e * M * D, ws.x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
  }
  return finish_dbg();
}

int eec_encoder_forward(eec_encoder* enc, const float* mel, const int64_t* lengths, int B, int T, int precision,
                        float* out, float* taps_opt, void* workspace, size_t workspace_bytes, int stop_after,
                        float* x_dbg_opt, void* stream) {
  if (!enc) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (!out && stop_after < 0) return fail(EEC_ERR_BAD_ARG, "out is null");
  return forward_impl(enc, mel, lengths, B, T, precision, out, taps_opt, workspace, workspace_bytes, stop_after, x_dbg_opt,
                      enc->cfg.n_exits, stream);
}

int eec_encoder_forward_prefix(eec_encoder* enc, const float* mel, const int64_t* lengths, int B, int T, int precision,
                               int n_groups, float* out_opt, float* taps_opt, float* x_out_opt, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!enc) return fail(EEC_ERR_BAD_ARG, "null argument");
  return forward_impl(enc, mel, lengths, B, T, precision, out_opt, taps_opt, workspace, workspace_bytes, -1, x_out_opt,
                      n_groups, stream);
}

// ---------------------------------------------------------------------------
// Building blocks: one exit group / one head on caller-owned rows.  They let a host compose the reference's
// other encoder topologies (Splitformer early_exit.py:227-364: parallel down-sampled branches) out of the same
// kernels; Early_conformer itself goes through eec_encoder_forward.
struct GroupWs {
  half_t *q, *k, *vt, *p_hi, *p_lo, *g;
  size_t bytes;
};
static GroupWs carve_group_ws(const eec_config& c, int B, int Tq, char* base) {
  const int Tp = (Tq + 31) / 32 * 32;
  const size_t M = (size_t)B * Tq, D = c.d_model;
  Arena a;
  a.base = base;
  GroupWs w;
  w.q = a.take<half_t>((size_t)2 * B * Tp * D);   // hi plane, then the residual plane (exact mode f16x3 only)
  w.k = a.take<half_t>((size_t)2 * B * Tp * D);
  w.vt = a.take<half_t>((size_t)2 * B * Tp * D);  // hi plane, then the residual plane
  w.p_hi = a.take<half_t>(M * D);
  w.p_lo = a.take<half_t>(M * D);
  w.g = a.take<half_t>(M * D);
  w.bytes = align_up(a.off);
  return w;
}

size_t eec_encoder_group_workspace_bytes(const eec_encoder* enc, int B, int Tq) {
  if (!enc || B <= 0 || Tq <= 0) return 0;
  return carve_group_ws(enc->cfg, B, Tq, nullptr).bytes;
}

int eec_encoder_group_forward(eec_encoder* enc, int group, float* x, const int32_t* key_len, int B, int Tq, int precision,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (!enc || !x || !key_len || !workspace) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (!enc->packed) return fail(EEC_ERR_NOT_PACKED, "eec_encoder_pack has not been called");
  if (int rc = check_device(enc)) return rc;
  const eec_config& c = enc->cfg;
  if (c.arch != EEC_ARCH_CONFORMER) return fail(EEC_ERR_UNSUPPORTED, "group forward is built for EEC_ARCH_CONFORMER");
  if (group < 0 || group >= c.n_exits) return fail(EEC_ERR_BAD_ARG, "group out of range");
  if (B <= 0 || Tq <= 0) return fail(EEC_ERR_BAD_ARG, "need B > 0 and T' > 0");
  if (precision < EEC_PREC_F16X3 || precision > EEC_PREC_F16F8) return fail(EEC_ERR_BAD_ARG, "unknown precision");
  if (((uintptr_t)workspace & 255) != 0) return fail(EEC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  const GroupWs ws = carve_group_ws(c, B, Tq, (char*)workspace);
  if (workspace_bytes < ws.bytes) return fail(EEC_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int np_ffn = precision == EEC_PREC_F16X3 ? 3 : (precision == EEC_PREC_F16F8 ? 8 : 1);
  const int np_o = precision == EEC_PREC_F16 ? 1 : 3;
  const int Tp = (Tq + 31) / 32 * 32, D = c.d_model;
  half_t* const vt_lo = np_o == 3 ? ws.vt + (size_t)B * Tp * D : nullptr;
  if (Tp != Tq) EEC_HIP(hipMemsetAsync(ws.vt, 0, (size_t)2 * B * Tp * D * sizeof(half_t), st));
  LayerBufs bufs{x, ws.q, ws.k, ws.vt, vt_lo, ws.p_hi, ws.p_lo, ws.g, key_len};
  if (precision == EEC_PREC_F16X3) bufs.q_lo = ws.q + (size_t)B * Tp * D, bufs.k_lo = ws.k + (size_t)B * Tp * D;  // exact mode (forward_impl)
  const int np_p = (precision == EEC_PREC_F16F8 && c.d_model == 256 && c.d_ff % 128 == 0) ? 8 : np_o;
  const LayerFormats nps{np_ffn, np_p, np_p, np_o, np_p};  // {ffn, front, qkv, att, glu}
  const int l0 = group * c.layers_per_exit;
  return run_layer_plan(enc, l0, l0 + c.layers_per_exit, bufs, B, Tq, nps, [](int) -> float* { return nullptr; }, st);
}

int eec_encoder_stem1_forward(eec_encoder* enc, const float* mel, int B, int T, float* x, void* stream) {
  if (!enc || !mel || !x) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (!enc->packed || !enc->has_stem1) return fail(EEC_ERR_NOT_PACKED, "no packed stem parameters");
  if (int rc = check_device(enc)) return rc;
  if (B <= 0 || T < 3) return fail(EEC_ERR_BAD_ARG, "need B > 0 and T >= 3");
  const eec_config& c = enc->cfg;
  const int T1 = (T - 3) / 2 + 1;
  if (T1 > c.max_len) return fail(EEC_ERR_BAD_ARG, "T1 exceeds the positional-encoding table (max_len)");
  hipStream_t st = (hipStream_t)stream;
  SubsampleArgs a{mel, B, c.n_mels, T, T1, T1, c.d_model, nullptr, enc->sub_w1p, enc->sub_b1, nullptr, nullptr, enc->pe, nullptr, nullptr, x};
  TIMED(KC_STEM, launch_subsample_single(a, st));
  return 0;
}

int eec_encoder_head_forward(eec_encoder* enc, int exit, const float* x, int M, float* out, int precision, void* stream) {
  if (!enc || !x || !out) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (!enc->packed || !enc->has_heads) return fail(EEC_ERR_NOT_PACKED, "no packed head parameters");
  if (int rc = check_device(enc)) return rc;
  if (exit < 0 || exit >= enc->cfg.n_exits || M <= 0) return fail(EEC_ERR_BAD_ARG, "exit / M out of range");
  if (precision < EEC_PREC_F16X3 || precision > EEC_PREC_F16F8) return fail(EEC_ERR_BAD_ARG, "unknown precision");
  HeadArgs h{x, M, enc->cfg.vocab, enc->cfg.d_model, enc->head_p[exit], enc->head_b[exit], out, enc->head_f8[exit]};
  hipStream_t st = (hipStream_t)stream;
  TIMED(KC_HEAD, launch_head(h, precision == EEC_PREC_F16 ? 1 : 3, st));
  return 0;
}

int eec_ctc_loss(const float* logp, const int64_t* targets, const int64_t* target_len, int E, int B, int Tq, int V, int S,
                 int blank, float* nll_scratch, float* loss_per_exit, void* stream) {
  if (!logp || !targets || !target_len || !nll_scratch || !loss_per_exit) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (E <= 0 || B <= 0 || Tq <= 0 || V <= 0 || S <= 0) return fail(EEC_ERR_BAD_ARG, "bad size");
  if (blank < 0 || blank >= V) return fail(EEC_ERR_BAD_ARG, "blank must be a label in [0, V)");
  if (2 * S + 1 > 512) return fail(EEC_ERR_UNSUPPORTED, "target length above 255");
  EEC_HIP(launch_ctc_loss(logp, (const long long*)targets, (const long long*)target_len, E, B, Tq, V, S, blank, nll_scratch,
                          loss_per_exit, nullptr, (hipStream_t)stream));
  return 0;
}

size_t eec_ctc_backward_workspace_bytes(int E, int B, int Tq, int S) {
  if (E <= 0 || B <= 0 || Tq <= 0 || S <= 0) return 0;
  return ctc_store_floats(E, B, Tq, S) * sizeof(float);
}

int eec_ctc_loss_forward(const float* logp, const int64_t* targets, const int64_t* target_len, int E, int B, int Tq, int V, int S,
                         int blank, float* nll, float* loss_per_exit, void* bwd_workspace, void* stream) {
  if (!logp || !targets || !target_len || !nll || !loss_per_exit || !bwd_workspace) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (E <= 0 || B <= 0 || Tq <= 0 || V <= 0 || S <= 0) return fail(EEC_ERR_BAD_ARG, "bad size");
  if (blank < 0 || blank >= V) return fail(EEC_ERR_BAD_ARG, "blank must be a label in [0, V)");
  if (2 * S + 1 > 512) return fail(EEC_ERR_UNSUPPORTED, "target length above 255");
  EEC_HIP(launch_ctc_loss(logp, (const long long*)targets, (const long long*)target_len, E, B, Tq, V, S, blank, nll,
                          loss_per_exit, (float*)bwd_workspace, (hipStream_t)stream));
  return 0;
}

int eec_ctc_loss_backward(const float* logp, const int64_t* targets, const int64_t* target_len, int E, int B, int Tq, int V, int S,
                          int blank, const float* nll, void* bwd_workspace, const float* grad_loss, float* dlogp, void* stream) {
  if (!logp || !targets || !target_len || !nll || !bwd_workspace || !grad_loss || !dlogp) return fail(EEC_ERR_BAD_ARG, "null argument");
  if (E <= 0 || B <= 0 || Tq <= 0 || V <= 0 || S <= 0) return fail(EEC_ERR_BAD_ARG, "bad size");
  if (V > 256 || V % 4) return fail(EEC_ERR_UNSUPPORTED, "vocab must be a multiple of 4, <= 256");
  if (2 * S + 1 > 512) return fail(EEC_ERR_UNSUPPORTED, "target length above 255");
  EEC_HIP(launch_ctc_backward(logp, (const long long*)targets, (const long long*)target_len, E, B, Tq, V, S, blank, nll,
                              (float*)bwd_workspace, grad_loss, dlogp, (hipStream_t)stream));
  return 0;
}

int eec_logsoftmax_backward(const float* logp, const float* grad_logp, int M, int V, float* grad_logits, void* stream) {
  if (!logp || !grad_logp || !grad_logits || M <= 0 || V <= 0) return fail(EEC_ERR_BAD_ARG, "bad argument");
  if (V > 256 || V % 4) return fail(EEC_ERR_UNSUPPORTED, "vocab must be a multiple of 4, <= 256");
  EEC_HIP(launch_logsoftmax_backward(logp, grad_logp, M, V, grad_logits, (hipStream_t)stream));
  return 0;
}

int eec_encoder_set_profiling(eec_encoder* enc, int enable, int max_launches) {
  if (!enc) return fail(EEC_ERR_BAD_ARG, "null argument");
  for (hipEvent_t e : enc->ev) (void)hipEventDestroy(e);
  enc->ev.clear();
  enc->ev_class.clear();
  enc->ev_used = 0;
  for (int i = 0; i < KC_COUNT; ++i) enc->prof_ms[i] = 0, enc->prof_n[i] = 0;
  enc->profiling = enable != 0;
  if (enc->profiling) {
    if (max_launches <= 0) max_launches = 4096;
    enc->ev.resize((size_t)max_launches * 2);
    enc->ev_class.assign(max_launches, 0);
    for (auto& e : enc->ev) EEC_HIP(hipEventCreate(&e));
  }
  return 0;
}

int eec_encoder_profile_read(eec_encoder* enc, double* ms_by_class, long long* launches_by_class, int n_classes) {
  if (!enc || !ms_by_class || !launches_by_class) return fail(EEC_ERR_BAD_ARG, "null argument");
  for (size_t i = 0; i < enc->ev_used; ++i) {
    EEC_HIP(hipEventSynchronize(enc->ev[2 * i + 1]));
    float ms = 0.f;
    EEC_HIP(hipEventElapsedTime(&ms, enc->ev[2 * i], enc->ev[2 * i + 1]));
    enc->prof_ms[enc->ev_class[i]] += ms;
    enc->prof_n[enc->ev_class[i]] += 1;
  }
  enc->ev_used = 0;
  for (int i = 0; i < n_classes && i < KC_COUNT; ++i) {
    ms_by_class[i] = enc->prof_ms[i];
    launches_by_class[i] = enc->prof_n[i];
  }
  return 0;
}

int eec_greedy_ctc(const float* logp, int n_seq, int Tq, int V, int blank, int32_t* tokens, int32_t* counts,
                   void* stream) {
  if (!logp || !tokens || !counts || n_seq <= 0 || Tq <= 0 || V <= 0) return fail(EEC_ERR_BAD_ARG, "bad argument");
  EEC_HIP(launch_greedy_ctc(logp, n_seq, Tq, V, blank, tokens, counts, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
