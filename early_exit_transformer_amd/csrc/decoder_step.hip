// Step-wise AED decoding with a key / value cache behind the C ABI (include/eec.h, eec_decoder_begin / eec_decoder_step).
// The reference's beam search (util/beam_infer.py:233-240) calls `_decoder_` on the whole prefix at every step: O(S^2)
// decoder work per utterance and, at 10 beams, launches that each cover a handful of rows.  Here one step computes ONLY the new
// position of every beam: the self-attention keys / values of earlier positions come from a cache indexed through the beams'
// ancestry (a beam re-ordering copies 4-byte slot numbers, not keys), the memory keys / values of the utterance are projected
// once per (utterance, exit) in eec_decoder_begin.  Same arithmetic as models/model/early_exit.py:739-762 in eval mode
// (norm_first nn.TransformerDecoderLayer, causal + target-padding key mask, shared final LayerNorm, exit Linear, log_softmax);
// plain fp32 FMA, fp32 weights straight from the module's parameters.
//
// A step has at most kRows = 16 live beams.  8 launches per decoder layer:
//   skinny_linear (LN1 -> in_proj)  step_attn<self>  skinny_linear (out_proj, += x)
//   skinny_linear (LN2 -> q)        step_attn<cross> skinny_linear (out_proj, += x)
//   skinny_linear (LN3 -> linear1 -> ReLU)           skinny_linear (linear2, += x)
// A step is bound by that launch count (a dependent launch costs 4 - 6 us here), not by its work, so every kernel takes a
// GROUP of up to 8 sessions -- the exits of one utterance, whose searches are independent -- as one more grid dimension
// (eec_decoder_step_multi): E searches for the launches of one.
#include <algorithm>
#include <string>

#include "../../include/eec.h"
#include "eec_train.h"

namespace eec {
hipError_t ensure_max_lds(const void* kernel, int bytes);  // pack.hip
}

using namespace eect;

namespace {

constexpr int kRows = 16;   // live beams per step (rows of every activation of a step)
constexpr int kGroup = 8;   // sessions (exits of one utterance) advanced by the same launches: one more grid dimension
template <typename A>
struct Group {
  A a[kGroup];
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Y[r][n] (+)= act( LN?(X[r]) . W[n] + bias[n] ), r < R <= 16.  Threads are (column c = tid / KL, k-lane j = tid % KL):
// a workgroup owns 256 / KL output columns (KL = 16: 16 columns, for wide outputs; KL = 64: a column per wave, so that a
// narrow output still spreads over >= 64 workgroups); the (normalised) rows sit in LDS; every weight is read once, as
// float4, in batches of kBatch loads that are all in flight before the first one is used -- the first batch is requested
// before the rows are staged.  The kernel is a latency chain (launch, rows, weights, reduce), not a bandwidth problem:
// 16 x 2048 fp32 rows and a 2 MB weight matrix per call at most.
// ---------------------------------------------------------------------------------------------------------------------
struct SkinnyArgs {
  const float* X;
  long ldx;
  const float *ln_g, *ln_b;  // LayerNorm over K <= 1024 (eps 1e-5) in the prologue, or null
  const float *W, *bias;     // [N][K], [N]
  float* Y;
  long ldy;
  int R, N, K, relu, accumulate;
};

#define EECS_DPP_ADD(v, ctrl, rmask) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, rmask, 0xf, false)))
// sum over the 16 lanes of a DPP row, left in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v = EECS_DPP_ADD(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
  v = EECS_DPP_ADD(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
  v = EECS_DPP_ADD(v, 0x141, 0xf);  // row_half_mirror
  v = EECS_DPP_ADD(v, 0x140, 0xf);  // row_mirror
  return v;
}
// sum over the wave, wave-uniform
__device__ __forceinline__ float wave64_sum(float v) {
  v = row16_sum(v);
  v = EECS_DPP_ADD(v, 0x142, 0xa);  // row_bcast15 -> rows 1, 3
  v = EECS_DPP_ADD(v, 0x143, 0xc);  // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr int kBatch = 8;

// The kernel runs ONCE per workgroup, so its instruction stream is fetched cold: code size is latency.  Hence rolled loops
// (the unrolled first version was 22 KB of straight-line code and took 9 us per call whatever the shape).
template <int KL>
__global__ __launch_bounds__(256) void skinny_linear_kernel(Group<SkinnyArgs> grp) {
  const SkinnyArgs& a = grp.a[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [kRows + 2][K]: the rows, then LayerNorm gain and bias
  const int tid = threadIdx.x, K = a.K, per_row = K >> 2;
  const int c = tid / KL, j = tid % KL, n = blockIdx.x * (256 / KL) + c;
  const bool col = n < a.N;
  const float* wr = a.W + (long)(col ? n : 0) * K;
  float4 wv[kBatch];
  auto fetch = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kBatch; ++i) {
      const int k = kb + (i * KL + j) * 4;
      wv[i] = k < K ? *reinterpret_cast<const float4*>(wr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  fetch(0);
  // each output (r = j, n) has one writer: its bias and, when accumulating, its old value are requested up front
  const bool writer = col && j < a.R;
  float* yp = a.Y + (writer ? j : 0) * a.ldy + (col ? n : 0);
  const float bias = (writer && a.bias) ? a.bias[n] : 0.0f;
  const float yold = (writer && a.accumulate) ? *yp : 0.0f;
  // ---- live rows (and the LayerNorm parameters) -> LDS ----
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = g4;
  const bool ln = a.ln_g != nullptr, ln_lane = ln && tid < per_row;  // K <= 1024: one float4 of gain / bias per thread
  if (ln_lane) g4 = reinterpret_cast<const float4*>(a.ln_g)[tid], b4 = reinterpret_cast<const float4*>(a.ln_b)[tid];
  {
    int r = tid / per_row, k4 = tid - r * per_row;
    const int dr = 256 / per_row, dk = 256 - dr * per_row;  // 256 float4 further on
    while (r < a.R) {
      float4 t[8];  // the rows were written by another XCD: every round trip goes to memory, so keep many in flight
      int at[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool on = r < a.R;
        at[i] = on ? r * per_row + k4 : -1;
        t[i] = *reinterpret_cast<const float4*>(a.X + (on ? r * a.ldx + 4 * k4 : 0));
        k4 += dk, r += dr;
        if (k4 >= per_row) k4 -= per_row, ++r;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (at[i] >= 0) reinterpret_cast<float4*>(xs)[at[i]] = t[i];
    }
  }
  if (ln_lane) {
    reinterpret_cast<float4*>(xs)[kRows * per_row + tid] = g4;
    reinterpret_cast<float4*>(xs)[(kRows + 1) * per_row + tid] = b4;
  }
  __syncthreads();
  if (ln) {  // in place, a row per 16-lane DPP row
    const int r = tid >> 4, l16 = tid & 15;
    if (r < a.R) {
      float4* xr = reinterpret_cast<float4*>(xs) + r * per_row;
      const float4* gs = reinterpret_cast<const float4*>(xs) + kRows * per_row;
      const float4* bs = gs + per_row;
      float sum = 0.0f;
      for (int q = l16; q < per_row; q += 16) {
        const float4 v = xr[q];
        sum += (v.x + v.y) + (v.z + v.w);
      }
      const float mu = row16_sum(sum) / K;
      float sq = 0.0f;
      for (int q = l16; q < per_row; q += 16) {
        const float4 v = xr[q];
        const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
        sq += dx * dx + dy * dy + dz * dz + dw * dw;
      }
      const float rs = rsqrtf(row16_sum(sq) / K + 1e-5f);
      for (int q = l16; q < per_row; q += 16) {
        float4 v = xr[q];
        const float4 g = gs[q], b = bs[q];
        v.x = (v.x - mu) * rs * g.x + b.x, v.y = (v.y - mu) * rs * g.y + b.y;
        v.z = (v.z - mu) * rs * g.z + b.z, v.w = (v.w - mu) * rs * g.w + b.w;
        xr[q] = v;
      }
    }
    __syncthreads();
  }
  float acc[kRows];
#pragma unroll
  for (int r = 0; r < kRows; ++r) acc[r] = 0.0f;
  for (int kb = 0; kb < K; kb += kBatch * KL * 4) {
    if (kb) fetch(kb);
#pragma unroll 1
    for (int i = 0; i < kBatch; ++i) {
      const int ku = kb + i * KL * 4;
      if (ku >= K) break;
      const float4 w0 = wv[0];
#pragma unroll
      for (int q = 0; q + 1 < kBatch; ++q) wv[q] = wv[q + 1];  // rotate: the loop stays rolled, the batch stays in registers
      const float4* xk = reinterpret_cast<const float4*>(xs + ku + j * 4);
      if (ku + j * 4 < K) {
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          if (r < a.R) {  // uniform
            const float4 xv = xk[r * per_row];
            acc[r] += w0.x * xv.x + w0.y * xv.y + w0.z * xv.z + w0.w * xv.w;
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    if (KL == 64) acc[r] = wave64_sum(acc[r]);
    else {
      acc[r] = row16_sum(acc[r]);
      if (KL == 32) acc[r] += __shfl_xor(acc[r], 16, 64);
    }
  }
  if (!writer) return;
  float v = 0.0f;
#pragma unroll
  for (int r = 0; r < kRows; ++r)
    if (j == r) v = acc[r];
  v += bias;
  if (a.relu) v = fmaxf(v, 0.0f);
  *yp = yold + v;
}

// the n sessions of a group share every shape (N, K, R): only pointers differ
hipError_t skinny_linear(const Group<SkinnyArgs>& g, int n, hipStream_t st) {
  const SkinnyArgs& a = g.a[0];
  const size_t lds = (size_t)(a.ln_g ? kRows + 2 : a.R) * a.K * sizeof(float);
  // k-lanes per column: 64 (a column per wave) spreads a narrow output over many workgroups -- the per-workgroup chain of
  // memory round trips is what a call costs (measured on the d_ff-long product of six sessions: 31 / 21 / 17 us at 96 / 192 /
  // 384 workgroups) -- down to 16 when the group of sessions would exceed about two workgroups per CU
  int kl = a.N >= 1024 ? 16 : 64;
  const int limit = 512;
  while (kl > 16 && n * ((a.N + 256 / kl - 1) / (256 / kl)) > limit) kl >>= 1;
#define EECS_SKINNY(KLv)                                                                                                  \
  do {                                                                                                                    \
    if (hipError_t e = eec::ensure_max_lds((const void*)skinny_linear_kernel<KLv>, (int)lds); e != hipSuccess) return e;  \
    hipLaunchKernelGGL(skinny_linear_kernel<KLv>, dim3((a.N + 256 / KLv - 1) / (256 / KLv), n), dim3(256), lds, st, g);  \
  } while (0)
  if (kl == 64) EECS_SKINNY(64);
  else if (kl == 32) EECS_SKINNY(32);
  else EECS_SKINNY(16);
#undef EECS_SKINNY
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// New position s of every live beam: x[r] = emb[token[r]] + pe[s]; pad flag of (s, r); ancestry of beam r = ancestry of
// its parent in the previous step + its own slot r at position s.
// ---------------------------------------------------------------------------------------------------------------------
struct EmbedArgs {
  const long long *tok, *parent;
  const float *emb, *pe;
  float* x;
  unsigned char* pad;
  const int* anc_old;
  int* anc_new;
};
__global__ __launch_bounds__(256) void step_embed_kernel(Group<EmbedArgs> grp, int s, int S_max, int D, int V, int pad_idx, int R_prev) {
  const EmbedArgs& a = grp.a[blockIdx.y];
  const int r = blockIdx.x;
  const long long t = a.tok[r];
  const long long tc = t < 0 ? 0 : (t >= V ? V - 1 : t);  // nn.Embedding would raise; stay in bounds
  for (int c = threadIdx.x; c < D; c += 256) a.x[(long)r * D + c] = a.emb[tc * D + c] + a.pe[(long)s * D + c];
  if (threadIdx.x == 0) a.pad[s * kRows + r] = t == pad_idx;
  int p = 0;
  if (s > 0) {
    const long long pp = a.parent ? a.parent[r] : r;
    p = (int)(pp < 0 ? 0 : (pp >= R_prev ? R_prev - 1 : pp));
  }
  for (int i = threadIdx.x; i <= s; i += 256) a.anc_new[r * S_max + i] = i < s ? a.anc_old[p * S_max + i] : r;
}

// log_softmax of the exit head's logits, a wave per (beam, session)
struct LsmArgs {
  const float* logits;
  float* out;
};
__global__ __launch_bounds__(64) void step_logsoftmax_kernel(Group<LsmArgs> grp, int V) {
  const LsmArgs& a = grp.a[blockIdx.y];
  const int lane = threadIdx.x;
  const float* xr = a.logits + (long)blockIdx.x * V;
  float mx = -INFINITY;
  for (int k = lane; k < V; k += 64) mx = fmaxf(mx, xr[k]);
  mx = wmax(mx);
  float sum = 0.0f;
  for (int k = lane; k < V; k += 64) sum += expf(xr[k] - mx);
  const float lse = mx + logf(wsum(sum));
  for (int k = lane; k < V; k += 64) a.out[(long)blockIdx.x * V + k] = xr[k] - lse;
}

// ---------------------------------------------------------------------------------------------------------------------
// One query row per (beam r, head h) workgroup of 4 waves: softmax(q . K^T * scale) . V over the cached keys.
// SELF: keys 0 .. s of the beam's ancestry (position s itself comes from the step's in_proj output and is appended to the
// cache here), keys whose token is the padding index masked.  Cross: the Tk memory keys, no mask.
// ---------------------------------------------------------------------------------------------------------------------
struct StepAttnArgs {
  const float* q;  // row r, head h at q + r * ldq + h * dh
  long ldq;
  const float *kn, *vn;  // SELF: this step's key / value rows (same row stride ldq)
  float* kv;             // SELF: cache [S_max][kRows][2D] of the layer; cross: memory [Tk][2D]
  const int* anc;        // [kRows][S_max]
  const unsigned char* pad;
  float* ctx;  // [R][D]
  int s, S_max, Tk, D, dh;
  float scale;
};

template <bool SELF>
__global__ __launch_bounds__(256) void step_attn_kernel(Group<StepAttnArgs> grp) {
  const StepAttnArgs& a = grp.a[blockIdx.z];
  extern __shared__ float lds[];
  __shared__ float red[4][64];
  __shared__ float stat[2][4];
  const int r = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int dh = a.dh, D = a.D, nk = SELF ? a.s + 1 : a.Tk;
  float* sc = lds;                  // [nk] scores, then probabilities
  int* slot = (int*)(lds + nk);     // [nk] SELF: cache slot of key t
  const float* q = a.q + r * a.ldq + h * dh;
  if (SELF && tid < dh) {  // append the new position: cache row (s, r)
    float* dst = a.kv + ((long)a.s * kRows + r) * 2 * D + h * dh;
    dst[tid] = a.kn[r * a.ldq + h * dh + tid];
    dst[D + tid] = a.vn[r * a.ldq + h * dh + tid];
  }
  float mx = -INFINITY;
  for (int t = tid; t < nk; t += 256) {
    const float* kp;
    bool live = true;
    if (SELF) {
      const int sl = t == a.s ? r : a.anc[r * a.S_max + t];
      slot[t] = sl;
      live = !a.pad[t * kRows + sl];
      kp = t == a.s ? a.kn + r * a.ldq + h * dh : a.kv + ((long)t * kRows + sl) * 2 * D + h * dh;
    } else {
      kp = a.kv + (long)t * 2 * D + h * dh;
    }
    float d = 0.0f;
    for (int i = 0; i < dh; i += 4) {
      const float4 kv4 = *reinterpret_cast<const float4*>(kp + i);
      const float4 qv = *reinterpret_cast<const float4*>(q + i);
      d += qv.x * kv4.x + qv.y * kv4.y + qv.z * kv4.z + qv.w * kv4.w;
    }
    d = live ? d * a.scale : -INFINITY;
    sc[t] = d;
    mx = fmaxf(mx, d);
  }
  mx = wmax(mx);
  if (lane == 0) stat[0][w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(stat[0][0], stat[0][1]), fmaxf(stat[0][2], stat[0][3]));
  float sum = 0.0f;
  for (int t = tid; t < nk; t += 256) {
    const float p = sc[t] == -INFINITY ? 0.0f : __expf(sc[t] - mx);
    sc[t] = p;
    sum += p;
  }
  sum = wsum(sum);
  if (lane == 0) stat[1][w] = sum;
  __syncthreads();
  const float inv = 1.0f / (stat[1][0] + stat[1][1] + stat[1][2] + stat[1][3]);  // no live key: nan, as torch
  // probabilities . V: lane = (feature d, part): the 4 * (64 / dh) (wave, part) pairs interleave the keys
  const int parts = 64 / dh, d = lane % dh, part = lane / dh, stride = 4 * parts;
  float acc = 0.0f;
  for (int t0 = w * parts + part; t0 < nk; t0 += 8 * stride) {  // 8 value rows in flight per lane
    float pv[8], vv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = t0 + i * stride;
      pv[i] = 0.0f, vv[i] = 0.0f;
      if (t < nk) {
        const float* vp;
        if (SELF) vp = t == a.s ? a.vn + r * a.ldq + h * dh : a.kv + ((long)t * kRows + slot[t]) * 2 * D + D + h * dh;
        else vp = a.kv + (long)t * 2 * D + D + h * dh;
        pv[i] = sc[t], vv[i] = vp[d];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += pv[i] * vv[i];
  }
  for (int m = dh; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
  red[w][lane] = acc;
  __syncthreads();
  if (tid < dh) a.ctx[(long)r * D + h * dh + tid] = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) * inv;
}

template <bool SELF>
hipError_t step_attn(const Group<StepAttnArgs>& g, int n, int R, int H, hipStream_t st) {
  const int nk = SELF ? g.a[0].s + 1 : g.a[0].Tk;
  const size_t lds = (size_t)nk * 8;
  if (hipError_t e = eec::ensure_max_lds((const void*)step_attn_kernel<SELF>, (int)lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(step_attn_kernel<SELF>, dim3(R, H, n), dim3(256), lds, st, g);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// The beam search's bookkeeping of one step (util/beam_infer.py:241-262) for every session of a group in one launch:
//   cand = scores[r] + logp[r][v] / penalty over all (r, v);  the K best, best first (ties: the lower flat index);
//   parent = index / V, token = index % V;  tokens_new[b] = tokens_old[parent[b]][0 .. len) + token[b].
// One workgroup per session; the R * V candidates sit in LDS and are searched K times.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beam_select_kernel(const float* __restrict__ logp, const float* __restrict__ scores_in, float penalty, int R,
                                                          int V, int K, float* __restrict__ scores_out, long long* __restrict__ parent,
                                                          long long* __restrict__ tok, const long long* __restrict__ tokens_old,
                                                          long long* __restrict__ tokens_new, int len, int ld, int rows_ld) {
  extern __shared__ float cand[];  // [R * V]
  __shared__ float wv[4];
  __shared__ int wi[4];
  __shared__ int chosen[kRows];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = R * V;
  const float* lp = logp + (long)e * n;
  for (int i = tid; i < n; i += 256) cand[i] = scores_in[e * R + i / V] + lp[i] / penalty;
  __syncthreads();
  for (int b = 0; b < K; ++b) {
    float best = -INFINITY;
    int at = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
      const float v = cand[i];
      if (v > best) best = v, at = i;  // ascending i: the first of equal values stays
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float ov = __shfl_xor(best, m, 64);
      const int oi = __shfl_xor(at, m, 64);
      if (ov > best || (ov == best && oi < at)) best = ov, at = oi;
    }
    if (lane == 0) wv[w] = best, wi[w] = at;
    __syncthreads();
    if (tid == 0) {
      float bv = wv[0];
      int bi = wi[0];
      for (int k = 1; k < 4; ++k)
        if (wv[k] > bv || (wv[k] == bv && wi[k] < bi)) bv = wv[k], bi = wi[k];
      if (bi < 0 || bi >= n) bi = 0;  // nothing comparable (all NaN / -inf): stay in bounds
      chosen[b] = bi;
      scores_out[e * K + b] = bv;
      parent[e * K + b] = bi / V;
      tok[e * K + b] = bi % V;
      cand[bi] = -INFINITY;
    }
    __syncthreads();
  }
  const long long* told = tokens_old + (long)e * rows_ld * ld;
  long long* tnew = tokens_new + (long)e * rows_ld * ld;
  for (int i = tid; i < K * (len + 1); i += 256) {
    const int b = i / (len + 1), j = i - b * (len + 1);
    tnew[(long)b * ld + j] = j < len ? told[(long)(chosen[b] / V) * ld + j] : (long long)(chosen[b] % V);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
thread_local std::string g_serr;
int sfail(int code, const std::string& msg) {
  g_serr = msg;
  return code;
}

struct Geo {
  int D, H, F, V, L, S_max, Tq;
};
struct Cache {
  float *mem, *kv, *x, *qkv, *q, *ctx, *h, *logits;
  int* anc;
  unsigned char* pad;
  size_t bytes;
};
Cache carve(char* base, const Geo& g) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    off = (off + 255) / 256 * 256;
    char* p = base + off;
    off += bytes;
    return p;
  };
  Cache c{};
  const size_t f = sizeof(float);
  c.mem = (float*)take((size_t)g.L * g.Tq * 2 * g.D * f);
  c.kv = (float*)take((size_t)g.L * g.S_max * kRows * 2 * g.D * f);
  c.x = (float*)take((size_t)kRows * g.D * f);
  c.qkv = (float*)take((size_t)kRows * 3 * g.D * f);
  c.q = (float*)take((size_t)kRows * g.D * f);
  c.ctx = (float*)take((size_t)kRows * g.D * f);
  c.h = (float*)take((size_t)kRows * g.F * f);
  c.logits = (float*)take((size_t)kRows * g.V * f);
  c.anc = (int*)take((size_t)2 * kRows * g.S_max * sizeof(int));
  c.pad = (unsigned char*)take((size_t)g.S_max * kRows);
  c.bytes = off + 256;
  return c;
}

bool geometry_ok(int d_model, int n_heads, int d_ff, int vocab, int n_layers, int S_max, int Tq) {
  if (d_model <= 0 || n_heads <= 0 || d_model % n_heads || d_ff <= 0 || vocab <= 0 || n_layers <= 0 || S_max <= 0 || Tq <= 0) return false;
  const int dh = d_model / n_heads;
  if (dh != 8 && dh != 16 && dh != 32 && dh != 64) return false;  // a head's features on a power-of-two fraction of a wave
  if (d_model % 4 || d_ff % 4) return false;                        // float4 weight rows
  if (d_model > 1024 || d_ff > 2048) return false;                  // LayerNorm rows in registers; 16 rows of d_ff in LDS (128 KB)
  if ((size_t)std::max(S_max, Tq) * 8 > 60000) return false;        // scores + slots of one query row in LDS
  return true;
}

#define SRUN(expr)                                                                          \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) return sfail((int)_e, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

}  // namespace

extern "C" {

const char* eec_decoder_step_last_error(void) { return g_serr.c_str(); }

int eec_decoder_step_max_beams(void) { return kRows; }

size_t eec_decoder_cache_bytes(int d_model, int n_heads, int d_ff, int vocab, int n_layers, int S_max, int Tq) {
  if (!geometry_ok(d_model, n_heads, d_ff, vocab, n_layers, S_max, Tq)) return 0;
  return carve(nullptr, Geo{d_model, n_heads, d_ff, vocab, n_layers, S_max, Tq}).bytes;
}

int eec_decoder_begin(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, const float* enc, int Tq, int S_max,
                      int passes, void* cache, size_t cache_bytes, void* stream) {
  if (!p || !p->layers || !enc || !cache) return sfail(EEC_ERR_BAD_ARG, "null argument");
  if (passes != 1 && passes != 3) return sfail(EEC_ERR_BAD_ARG, "passes: 1 (bf16) or 3 (bf16x3)");
  if (!geometry_ok(d_model, n_heads, d_ff, vocab, p->n_layers, S_max, Tq) || S_max > p->max_len)
    return sfail(EEC_ERR_UNSUPPORTED, "geometry not served by the step-wise decoder (use eec_decoder_forward)");
  const Geo g{d_model, n_heads, d_ff, vocab, p->n_layers, S_max, Tq};
  const Cache c = carve((char*)cache, g);
  if (c.bytes > cache_bytes) return sfail(EEC_ERR_WORKSPACE, "cache too small");
  hipStream_t st = (hipStream_t)stream;
  const int D = d_model;
  for (int l = 0; l < p->n_layers; ++l) {  // memory keys | values of every layer: enc . W[D:3D]^T + b[D:3D]
    const eec_decoder_layer_params& L = p->layers[l];
    GemmArgs a = gemm_args(enc, D, 1, L.ca_in_w + (size_t)D * D, D, 1, c.mem + (size_t)l * Tq * 2 * D, 2 * D, Tq, 2 * D, D);
    a.bias = L.ca_in_b + D;
    SRUN(launch_gemm(a, passes, st));
  }
  return 0;
}

int eec_decoder_step_multi(int n, const eec_decoder_params* const* ps, int d_model, int n_heads, int d_ff, int vocab, int pad_idx,
                           const int64_t* last_tokens, const int64_t* parent, int R, int R_prev, int s, int Tq, int S_max, int log_softmax,
                           float* out, void* const* caches, size_t cache_bytes, void* stream) {
  if (n <= 0 || n > kGroup) return sfail(EEC_ERR_BAD_ARG, "1 .. 8 sessions per call");
  if (!ps || !last_tokens || !out || !caches) return sfail(EEC_ERR_BAD_ARG, "null argument");
  for (int i = 0; i < n; ++i) {
    if (!ps[i] || !ps[i]->layers || !caches[i]) return sfail(EEC_ERR_BAD_ARG, "null argument");
    if (ps[i]->n_layers != ps[0]->n_layers) return sfail(EEC_ERR_BAD_ARG, "the sessions of a call share one decoder geometry");
    if (S_max > ps[i]->max_len) return sfail(EEC_ERR_UNSUPPORTED, "S_max beyond the positional-encoding table");
  }
  const int n_layers = ps[0]->n_layers;
  if (!geometry_ok(d_model, n_heads, d_ff, vocab, n_layers, S_max, Tq))
    return sfail(EEC_ERR_UNSUPPORTED, "geometry not served by the step-wise decoder (use eec_decoder_forward)");
  if (R <= 0 || R > kRows) return sfail(EEC_ERR_BAD_ARG, "1 .. 16 live beams per step");
  if (s < 0 || s >= S_max) return sfail(EEC_ERR_BAD_ARG, "step index outside the cache (S_max)");
  if (s > 0 && (R_prev <= 0 || R_prev > kRows)) return sfail(EEC_ERR_BAD_ARG, "R_prev: the previous step's beam count");
  const Geo g{d_model, n_heads, d_ff, vocab, n_layers, S_max, Tq};
  Cache c[kGroup];
  for (int i = 0; i < n; ++i) {
    c[i] = carve((char*)caches[i], g);
    if (c[i].bytes > cache_bytes) return sfail(EEC_ERR_WORKSPACE, "cache too small");
  }
  hipStream_t st = (hipStream_t)stream;
  const int D = d_model, H = n_heads, dh = D / H, F = d_ff;
  const float scale = 1.0f / sqrtf((float)dh);
  const size_t anc_new = (size_t)(s & 1) * kRows * S_max, anc_old = (size_t)((s + 1) & 1) * kRows * S_max;
  {
    Group<EmbedArgs> e{};
    for (int i = 0; i < n; ++i)
      e.a[i] = EmbedArgs{(const long long*)last_tokens + (size_t)i * R, parent ? (const long long*)parent + (size_t)i * R : nullptr, ps[i]->emb,
                         ps[i]->pe, c[i].x, c[i].pad, c[i].anc + anc_old, c[i].anc + anc_new};
    hipLaunchKernelGGL(step_embed_kernel, dim3(R, n), dim3(256), 0, st, e, s, S_max, D, vocab, pad_idx, R_prev);
    SRUN(hipGetLastError());
  }
  // one skinny_linear launch for all sessions: session i's operands through f(i)
  auto linear = [&](auto f, int N, int K, int relu, int accumulate) {
    Group<SkinnyArgs> grp{};
    for (int i = 0; i < n; ++i) {
      grp.a[i] = f(i);
      grp.a[i].R = R, grp.a[i].N = N, grp.a[i].K = K, grp.a[i].relu = relu, grp.a[i].accumulate = accumulate;
    }
    return skinny_linear(grp, n, st);
  };
  for (int l = 0; l < n_layers; ++l) {
    auto L = [&](int i) -> const eec_decoder_layer_params& { return ps[i]->layers[l]; };
    // self-attention over the beam's own prefix
    SRUN(linear([&](int i) { return SkinnyArgs{c[i].x, D, L(i).norm1_w, L(i).norm1_b, L(i).sa_in_w, L(i).sa_in_b, c[i].qkv, 3L * D}; }, 3 * D, D, 0, 0));
    {
      Group<StepAttnArgs> sa{};
      for (int i = 0; i < n; ++i)
        sa.a[i] = StepAttnArgs{c[i].qkv, 3L * D, c[i].qkv + D, c[i].qkv + 2 * D, c[i].kv + (size_t)l * S_max * kRows * 2 * D, c[i].anc + anc_new, c[i].pad,
                               c[i].ctx, s, S_max, 0, D, dh, scale};
      SRUN(step_attn<true>(sa, n, R, H, st));
    }
    SRUN(linear([&](int i) { return SkinnyArgs{c[i].ctx, D, nullptr, nullptr, L(i).sa_out_w, L(i).sa_out_b, c[i].x, D}; }, D, D, 0, 1));
    // cross-attention over the utterance's memory
    SRUN(linear([&](int i) { return SkinnyArgs{c[i].x, D, L(i).norm2_w, L(i).norm2_b, L(i).ca_in_w, L(i).ca_in_b, c[i].q, D}; }, D, D, 0, 0));
    {
      Group<StepAttnArgs> ca{};
      for (int i = 0; i < n; ++i)
        ca.a[i] = StepAttnArgs{c[i].q, (long)D, nullptr, nullptr, c[i].mem + (size_t)l * Tq * 2 * D, nullptr, nullptr, c[i].ctx, s, S_max, Tq, D, dh, scale};
      SRUN(step_attn<false>(ca, n, R, H, st));
    }
    SRUN(linear([&](int i) { return SkinnyArgs{c[i].ctx, D, nullptr, nullptr, L(i).ca_out_w, L(i).ca_out_b, c[i].x, D}; }, D, D, 0, 1));
    // feed-forward, ReLU
    SRUN(linear([&](int i) { return SkinnyArgs{c[i].x, D, L(i).norm3_w, L(i).norm3_b, L(i).w1, L(i).b1, c[i].h, F}; }, F, D, 1, 0));
    SRUN(linear([&](int i) { return SkinnyArgs{c[i].h, F, nullptr, nullptr, L(i).w2, L(i).b2, c[i].x, D}; }, D, F, 0, 1));
  }
  const size_t out_stride = (size_t)R * vocab;
  if (log_softmax) {
    SRUN(linear([&](int i) { return SkinnyArgs{c[i].x, D, ps[i]->norm_w, ps[i]->norm_b, ps[i]->head_w, ps[i]->head_b, c[i].logits, vocab}; }, vocab, D, 0, 0));
    Group<LsmArgs> lg{};
    for (int i = 0; i < n; ++i) lg.a[i] = LsmArgs{c[i].logits, out + i * out_stride};
    hipLaunchKernelGGL(step_logsoftmax_kernel, dim3(R, n), dim3(64), 0, st, lg, vocab);
    SRUN(hipGetLastError());
  } else {
    SRUN(linear([&](int i) { return SkinnyArgs{c[i].x, D, ps[i]->norm_w, ps[i]->norm_b, ps[i]->head_w, ps[i]->head_b, out + i * out_stride, vocab}; }, vocab, D, 0, 0));
  }
  return 0;
}

int eec_beam_select(int n, int R, int V, int K, const float* logp, const float* scores_in, float penalty, float* scores_out, int64_t* parent,
                    int64_t* tok, const int64_t* tokens_old, int64_t* tokens_new, int len, int ld, int rows_ld, void* stream) {
  if (!logp || !scores_in || !scores_out || !parent || !tok || !tokens_old || !tokens_new) return sfail(EEC_ERR_BAD_ARG, "null argument");
  if (n <= 0 || R <= 0 || R > kRows || V <= 0 || K <= 0 || K > kRows || K > (long)R * V || rows_ld < std::max(R, K) || len < 0 || ld < len + 1 ||
      !(penalty > 0.0f))
    return sfail(EEC_ERR_BAD_ARG, "eec_beam_select: 1 .. 16 beams in and out, token rows of at least len + 1");
  const size_t lds = (size_t)R * V * sizeof(float);
  if (lds > 150000) return sfail(EEC_ERR_UNSUPPORTED, "eec_beam_select: R * V candidates must fit the LDS");
  SRUN(eec::ensure_max_lds((const void*)beam_select_kernel, (int)lds));
  hipLaunchKernelGGL(beam_select_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, logp, scores_in, penalty, R, V, K, scores_out,
                     (long long*)parent, (long long*)tok, (const long long*)tokens_old, (long long*)tokens_new, len, ld, rows_ld);
  SRUN(hipGetLastError());
  return 0;
}

int eec_decoder_step(const eec_decoder_params* p, int d_model, int n_heads, int d_ff, int vocab, int pad_idx, const int64_t* last_tokens,
                     const int64_t* parent, int R, int R_prev, int s, int Tq, int S_max, int log_softmax, float* out, void* cache,
                     size_t cache_bytes, void* stream) {
  return eec_decoder_step_multi(1, &p, d_model, n_heads, d_ff, vocab, pad_idx, last_tokens, parent, R, R_prev, s, Tq, S_max, log_softmax, out, &cache,
                                cache_bytes, stream);
}

}  // extern "C"
