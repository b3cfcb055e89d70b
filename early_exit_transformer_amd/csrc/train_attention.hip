// Fused self-attention of the training step (head dim 32 or 64): forward without the [B, H, T', T'] score / probability
// tensors in HBM -- a wave owns 32 query frames, walks the key tiles with an online softmax and keeps only the row
// log-sum-exp for the backward -- and a backward in two passes that recompute the probabilities from Q, K and that
// log-sum-exp: one over query tiles (dQ), one over key tiles (dK, dV).  Operands are read as fp32 from the packed
// in_proj output [M][3D] and split into bf16 hi / lo in registers (3 MFMA products per contraction, or 1: eec_train.h).
//
// Orientation trick (as the inference kernel): score tiles are computed TRANSPOSED where the next product contracts over
// their rows, so that an accumulator lane already holds what the next MFMA wants as its operand: accumulator register i
// of lane half h is tile row 16*(i/8) + 8*((i%8)/4) + 4*h + i%4, i.e. registers 8*ks .. 8*ks+7 are the eight k-slots of
// k-step ks; the partner operand is gathered from memory with the same slot -> row map.
#include "eec_drop.h"

namespace eec {
hipError_t ensure_max_lds(const void* kernel, int bytes);  // pack.hip: per-device MaxDynamicSharedMemorySize attribute
}

namespace eect {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef EECT_ABLATE_ATTN_STORES
#define EECT_ABLATE_ATTN_STORES 0  // timing-only builds: 1 = the output tiles are computed and not stored
#endif
namespace {

typedef DropState DropState2;  // the training step's generator (eec_drop.h)

struct Frag {
  bf16x8 hi, lo;
};
template <int NP>
__device__ __forceinline__ Frag split8(const f32x8 v) {
  Frag f;
  f.hi = __builtin_convertvector(v, bf16x8);
  if (NP == 3) f.lo = __builtin_convertvector(v - __builtin_convertvector(f.hi, f32x8), bf16x8);
  else f.lo = f.hi;
  return f;
}
// c += a . b with the operands' bf16 hi / lo parts
template <int NP>
__device__ __forceinline__ f32x16 mfma3(const Frag& a, const Frag& b, f32x16 c) {
  if (NP == 3) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
}
// tile row of k-slot j (0..7) of k-step ks (0, 1) for lane half hh: the accumulator layout read as an operand
__device__ __forceinline__ int slot_row(int ks, int j, int hh) { return 16 * ks + 8 * (j >> 2) + 4 * hh + (j & 3); }
__device__ __forceinline__ int acc_row(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

// operand fragment of row `row` (8 consecutive fp32 at p); zeros when !ok.  p is 32-byte aligned (head dim % 8 == 0).
template <int NP>
__device__ __forceinline__ Frag load_kc(const float* __restrict__ p, bool ok) {
  f32x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ok) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    v = (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
  return split8<NP>(v);
}
// operand fragment gathered over the ROWS of X: slot j <- X[(row0 + slot_row(ks, j, hh)) * ld + col]; rows >= nrows give zeros
template <int NP>
__device__ __forceinline__ Frag load_gather(const float* __restrict__ X, long ld, int row0, int nrows, int ks, int hh, int col) {
  f32x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = row0 + slot_row(ks, j, hh);
    v[j] = r < nrows ? X[(long)r * ld + col] : 0.0f;
  }
  return split8<NP>(v);
}
// the eight registers 8*ks .. 8*ks+7 of an accumulator as an operand fragment
template <int NP>
__device__ __forceinline__ Frag acc_frag(const f32x16& a, int ks) {
  f32x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a[8 * ks + j];
  return split8<NP>(v);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

constexpr float kNegBig = -1.0e30f;

struct AttnGeoK {  // pointers of one (b, h): rows are frames, ld = 3D for q / k / v, D for o / do
  const float *q, *k, *v;
  long ld;
  int Tq, len;
  long pbase;  // index of P[z][0][0] in the flat [B*H][Tq][Tq] tensor (dropout stream position)
};
__device__ __forceinline__ AttnGeoK attn_geo(const float* qkv, const int32_t* key_len, int H, int Tq, int D, int DH, int z) {
  const int b = z / H, h = z % H;
  AttnGeoK g;
  g.q = qkv + (long)b * Tq * 3 * D + h * DH, g.k = g.q + D, g.v = g.q + 2 * D;
  g.ld = 3L * D, g.Tq = Tq, g.len = min(key_len[b], Tq);
  g.pbase = (long)z * Tq * Tq;
  return g;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Tiles shared by the four waves of a workgroup: a 32-row x DH block of an fp32 matrix is loaded and split ONCE per
// workgroup into LDS, as bf16 hi / lo planes in the row layout ([32][DH + 8]: ds_read_b128 = the 8 consecutive-d operand
// of a row) and / or the slot layout ([DH][40]: ds_read_b128 = the 8 k-slots of column d, rows stored at slot_pos(row)).
// Two buffers: the global loads of tile t + 1 are issued before tile t is consumed and written to LDS after it.
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int DH, int NP, bool ROWL, bool SLOTL>
struct TileLds {
  static constexpr int kLdR = DH + 8, kLdT = 40, kPl = NP == 3 ? 2 : 1;  // planes per layout
  static constexpr int kRow = 32 * kLdR, kSlot = DH * kLdT;
  static constexpr int kRowLo = kRow, kSlotHi = ROWL ? kPl * kRow : 0, kSlotLo = kSlotHi + kSlot;
  static constexpr int kElems = (ROWL ? kPl * kRow : 0) + (SLOTL ? kPl * kSlot : 0);
};
__device__ __forceinline__ int slot_pos(int row) {  // inverse of slot_row: where tile row `row` sits among the 32 slots
  const int x = row & 15;
  return (row & 16) + ((x >> 2) & 1) * 8 + ((x >> 3) << 2) + (x & 3);
}
template <int DH>
struct StageRegs {
  f32x4 v[DH / 32];
};
template <int DH>
__device__ __forceinline__ void stage_issue(StageRegs<DH>& sr, const float* __restrict__ X, long ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int it = 0; it < DH / 32; ++it) {
    const int idx = it * 256 + tid, row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
    const int rr = row0 + row;
    sr.v[it] = rr < nrows ? *(const f32x4*)(X + (long)rr * ld + c4) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  }
}
template <int DH, int NP, bool ROWL, bool SLOTL>
__device__ __forceinline__ void stage_commit(const StageRegs<DH>& sr, bf16* __restrict__ tile, int tid) {
  using T = TileLds<DH, NP, ROWL, SLOTL>;
#pragma unroll
  for (int it = 0; it < DH / 32; ++it) {
    const int idx = it * 256 + tid, row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
    const bf16x4 h = __builtin_convertvector(sr.v[it], bf16x4);
    bf16x4 l = h;
    if (NP == 3) l = __builtin_convertvector(sr.v[it] - __builtin_convertvector(h, f32x4), bf16x4);
    if (ROWL) {
      *(bf16x4*)(tile + row * T::kLdR + c4) = h;
      if (NP == 3) *(bf16x4*)(tile + T::kRowLo + row * T::kLdR + c4) = l;
    }
    if (SLOTL) {
      const int pos = slot_pos(row);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tile[T::kSlotHi + (c4 + e) * T::kLdT + pos] = h[e];
        if (NP == 3) tile[T::kSlotLo + (c4 + e) * T::kLdT + pos] = l[e];
      }
    }
  }
}
template <int DH, int NP, bool ROWL, bool SLOTL>
__device__ __forceinline__ Frag row_frag(const bf16* __restrict__ tile, int row, int ks, int hh) {
  using T = TileLds<DH, NP, ROWL, SLOTL>;
  Frag f;
  f.hi = *(const bf16x8*)(tile + row * T::kLdR + ks * 16 + 8 * hh);
  f.lo = NP == 3 ? *(const bf16x8*)(tile + T::kRowLo + row * T::kLdR + ks * 16 + 8 * hh) : f.hi;
  return f;
}
template <int DH, int NP, bool ROWL, bool SLOTL>
__device__ __forceinline__ Frag slot_frag(const bf16* __restrict__ tile, int col, int ks, int hh) {
  using T = TileLds<DH, NP, ROWL, SLOTL>;
  Frag f;
  f.hi = *(const bf16x8*)(tile + T::kSlotHi + col * T::kLdT + ks * 16 + 8 * hh);
  f.lo = NP == 3 ? *(const bf16x8*)(tile + T::kSlotLo + col * T::kLdT + ks * 16 + 8 * hh) : f.hi;
  return f;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: ctx[q][h*DH + d] = sum_k drop(softmax_k(scale * q.k))[q][k] v[k][d];  lse[z][q] = log sum_k exp(scale * q.k)
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int NP>
__global__ __launch_bounds__(256, DH == 32 ? 2 : 1) void attn_fwd_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ key_len, float* __restrict__ ctx,
                                                          float* __restrict__ lse, int H, int Tq, int D, float scale, Drop drop) {
  constexpr int KSQ = DH / 16, DT = DH / 32;
  using TK = TileLds<DH, NP, true, false>;   // K: row layout
  using TV = TileLds<DH, NP, false, true>;   // V: slot layout
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* const lds0 = (bf16*)smem;
  bf16* kt_lds[2];
  bf16* vt_lds[2];
  kt_lds[0] = lds0, kt_lds[1] = lds0 + TK::kElems, vt_lds[0] = lds0 + 2 * TK::kElems, vt_lds[1] = lds0 + 2 * TK::kElems + TV::kElems;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int z = blockIdx.y, q = blockIdx.x * 128 + w * 32 + r;
  const AttnGeoK g = attn_geo(qkv, key_len, H, Tq, D, DH, z);
  const DropState2 ds(drop);
  const float scale2 = scale * 1.4426950408889634f;  // scores in the exp2 domain; the row log-sum-exp is kept in log2 units
  Frag qf[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) qf[ks] = load_kc<NP>(g.q + (long)q * g.ld + ks * 16 + 8 * hh, q < Tq);
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
  float m_run = kNegBig, l_run = 0.0f;
  const int nkt = (g.len + 31) / 32;  // uniform over the workgroup
  StageRegs<DH> sk, sv;
  if (nkt > 0) {
    stage_issue<DH>(sk, g.k, g.ld, 0, Tq, tid);
    stage_issue<DH>(sv, g.v, g.ld, 0, Tq, tid);
    stage_commit<DH, NP, true, false>(sk, kt_lds[0], tid);
    stage_commit<DH, NP, false, true>(sv, vt_lds[0], tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const bf16* kt_t = kt_lds[kt & 1];
    const bf16* vt_t = vt_lds[kt & 1];
    if (kt + 1 < nkt) {
      stage_issue<DH>(sk, g.k, g.ld, (kt + 1) * 32, Tq, tid);
      stage_issue<DH>(sv, g.v, g.ld, (kt + 1) * 32, Tq, tid);
    }
    f32x16 sc = zero16();
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) sc = mfma3<NP>(row_frag<DH, NP, true, false>(kt_t, r, ks, hh), qf[ks], sc);
    // the softmax runs in the exp2 domain (scale2 = scale * log2 e: one v_exp_f32 per score, no multiply in front of it) and a key tile
    // that lies inside the utterance -- all but the last one -- takes the form without the per-element key tests
    const bool full = kt * 32 + 32 <= g.len;  // uniform
    float tmax = kNegBig;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sc[i] = (full || kt * 32 + acc_row(i, hh) < g.len) ? sc[i] * scale2 : kNegBig;
      tmax = fmaxf(tmax, sc[i]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax), alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.0f;
    f32x16 pd;
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {  // register quad gk = four consecutive keys: one set of index products for their four masks
      float dm[4];
      ds.mul4((uint64_t)(g.pbase + (long)q * Tq + kt * 32 + 8 * gk + 4 * hh), dm);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * gk + e;
        const float p = (full || kt * 32 + acc_row(i, hh) < g.len) ? __builtin_amdgcn_exp2f(sc[i] - m_new) : 0.0f;
        psum += p;
        pd[i] = p * dm[e];
      }
    }
    l_run = l_run * alpha + psum;
    Frag pf[2];
    pf[0] = acc_frag<NP>(pd, 0), pf[1] = acc_frag<NP>(pd, 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) o[dt] = mfma3<NP>(slot_frag<DH, NP, false, true>(vt_t, dt * 32 + r, ks, hh), pf[ks], o[dt]);
    }
    if (kt + 1 < nkt) {
      stage_commit<DH, NP, true, false>(sk, kt_lds[(kt + 1) & 1], tid);
      stage_commit<DH, NP, false, true>(sv, vt_lds[(kt + 1) & 1], tid);
    }
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = g.len > 0 ? 1.0f / l_tot : NAN;  // no key at all: nan, as torch's masked softmax
  if (q < Tq && !(EECT_ABLATE_ATTN_STORES && H > 0)) {
    float* dst = ctx + ((long)(z / H) * Tq + q) * D + (z % H) * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *(f32x4*)(dst + dt * 32 + 8 * gq + 4 * hh) = (f32x4){o[dt][4 * gq] * inv, o[dt][4 * gq + 1] * inv, o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv};
    if (hh == 0) lse[(long)z * Tq + q] = g.len > 0 ? m_run + __builtin_amdgcn_logf(l_tot) : NAN;  // log2 of the row sum of exp2(scale2 * s): what the backward subtracts
  }
}

// delta[z][q] = sum_d dO[q][h*DH + d] * O[q][h*DH + d].  A lane takes four consecutive columns of a row, so a wave-instruction reads
// whole rows (1 KiB contiguous); the DH / 4 lanes of a head (8 or 16) add up through the DPP row.  (First form: a thread per (z, q)
// reading its 128 B piece in 8 loads, every load instruction touching 64 different rows: 25 us per launch against 8 us of traffic.)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ delta, int H, int Tq,
                                                         int D, int DH, long n4 /* rows * D / 4 */) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = D >> 2, per_head = DH >> 2;
  float s = 0.0f;
  long row = 0;
  int c4 = 0;
  if (i < n4) {
    row = i / per_row, c4 = (int)(i - row * per_row);
    const f32x4 a = *(const f32x4*)(o + i * 4), b = *(const f32x4*)(d_o + i * 4);
    s = (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
  }
  for (int off = 1; off < per_head; off <<= 1) s += __shfl_xor(s, off, 64);  // per_head is 8 or 16: stays inside the head's lanes
  if (i < n4 && (c4 & (per_head - 1)) == 0) {
    const long bb = row / Tq;
    const int q = (int)(row - bb * Tq), h = c4 / per_head;
    delta[(bb * H + h) * Tq + q] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, pass over query tiles: dQ[q][d] = sum_k dS[q][k] K[k][d],  dS = scale * P * (dP_drop * mask - delta)
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int NP>
__global__ __launch_bounds__(256, DH == 32 ? 2 : 1) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ key_len, const float* __restrict__ d_o,
                                                             const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
                                                             int H, int Tq, int D, float scale, Drop drop) {
  constexpr int KSQ = DH / 16, DT = DH / 32;
  using TK = TileLds<DH, NP, true, true>;    // K: row layout (scores) and slot layout (dQ)
  using TV = TileLds<DH, NP, true, false>;   // V: row layout (dP)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* const lds0 = (bf16*)smem;
  bf16* kt_lds[2];
  bf16* vt_lds[2];
  kt_lds[0] = lds0, kt_lds[1] = lds0 + TK::kElems, vt_lds[0] = lds0 + 2 * TK::kElems, vt_lds[1] = lds0 + 2 * TK::kElems + TV::kElems;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int z = blockIdx.y, q = blockIdx.x * 128 + w * 32 + r;
  const AttnGeoK g = attn_geo(qkv, key_len, H, Tq, D, DH, z);
  const DropState2 ds(drop);
  const float scale2 = scale * 1.4426950408889634f;  // scores in the exp2 domain; the row log-sum-exp is kept in log2 units
  const float* dob = d_o + (long)(z / H) * Tq * D + (z % H) * DH;
  Frag qf[KSQ], dof[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) {
    qf[ks] = load_kc<NP>(g.q + (long)q * g.ld + ks * 16 + 8 * hh, q < Tq);
    dof[ks] = load_kc<NP>(dob + (long)q * D + ks * 16 + 8 * hh, q < Tq);
  }
  // (a frame past the end: a log-sum-exp of +1e30 makes every one of its probabilities exp2(-1e30) = 0 without a per-element test)
  const float lse_q = q < Tq ? lse[(long)z * Tq + q] : 1.0e30f, del_q = q < Tq ? delta[(long)z * Tq + q] : 0.0f;
  f32x16 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dq[dt] = zero16();
  const int nkt = (g.len + 31) / 32;
  StageRegs<DH> sk, sv;
  if (nkt > 0) {
    stage_issue<DH>(sk, g.k, g.ld, 0, Tq, tid);
    stage_issue<DH>(sv, g.v, g.ld, 0, Tq, tid);
    stage_commit<DH, NP, true, true>(sk, kt_lds[0], tid);
    stage_commit<DH, NP, true, false>(sv, vt_lds[0], tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const bf16* kt_t = kt_lds[kt & 1];
    const bf16* vt_t = vt_lds[kt & 1];
    if (kt + 1 < nkt) {
      stage_issue<DH>(sk, g.k, g.ld, (kt + 1) * 32, Tq, tid);
      stage_issue<DH>(sv, g.v, g.ld, (kt + 1) * 32, Tq, tid);
    }
    f32x16 sc = zero16(), dp = zero16();
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      sc = mfma3<NP>(row_frag<DH, NP, true, true>(kt_t, r, ks, hh), qf[ks], sc);
      dp = mfma3<NP>(row_frag<DH, NP, true, false>(vt_t, r, ks, hh), dof[ks], dp);
    }
    f32x16 dsv;
    const bool full = kt * 32 + 32 <= g.len;  // uniform: only the utterance's last key tile tests its keys
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      float dm[4];
      ds.mul4((uint64_t)(g.pbase + (long)q * Tq + kt * 32 + 8 * gk + 4 * hh), dm);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * gk + e;
        const float p = (full || kt * 32 + acc_row(i, hh) < g.len) ? __builtin_amdgcn_exp2f(sc[i] * scale2 - lse_q) : 0.0f;
        dsv[i] = scale * p * (dp[i] * dm[e] - del_q);
      }
    }
    Frag dsf[2];
    dsf[0] = acc_frag<NP>(dsv, 0), dsf[1] = acc_frag<NP>(dsv, 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dq[dt] = mfma3<NP>(slot_frag<DH, NP, true, true>(kt_t, dt * 32 + r, ks, hh), dsf[ks], dq[dt]);
    if (kt + 1 < nkt) {
      stage_commit<DH, NP, true, true>(sk, kt_lds[(kt + 1) & 1], tid);
      stage_commit<DH, NP, true, false>(sv, vt_lds[(kt + 1) & 1], tid);
    }
    __syncthreads();
  }
  if (q < Tq && !(EECT_ABLATE_ATTN_STORES && H > 0)) {
    float* dst = dqkv + ((long)(z / H) * Tq + q) * 3 * D + (z % H) * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *(f32x4*)(dst + dt * 32 + 8 * gq + 4 * hh) = (f32x4){dq[dt][4 * gq], dq[dt][4 * gq + 1], dq[dt][4 * gq + 2], dq[dt][4 * gq + 3]};
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, pass over key tiles: dV[k][d] = sum_q P_drop[q][k] dO[q][d],  dK[k][d] = sum_q dS[q][k] Q[q][d]
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int NP>
__global__ __launch_bounds__(256, DH == 32 ? 2 : 1) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ key_len, const float* __restrict__ d_o,
                                                              const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
                                                              int H, int Tq, int D, float scale, Drop drop) {
  constexpr int KSQ = DH / 16, DT = DH / 32;
  using T = TileLds<DH, NP, true, true>;  // Q and dO: both layouts
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* const lds0 = (bf16*)smem;
  bf16* qt_lds[2];
  bf16* dot_lds[2];
  qt_lds[0] = lds0, qt_lds[1] = lds0 + T::kElems, dot_lds[0] = lds0 + 2 * T::kElems, dot_lds[1] = lds0 + 3 * T::kElems;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int z = blockIdx.y, key = blockIdx.x * 128 + w * 32 + r;  // the key this lane owns as an accumulator COLUMN
  const AttnGeoK g = attn_geo(qkv, key_len, H, Tq, D, DH, z);
  const DropState2 ds(drop);
  const float scale2 = scale * 1.4426950408889634f;  // scores in the exp2 domain; the row log-sum-exp is kept in log2 units
  const float* dob = d_o + (long)(z / H) * Tq * D + (z % H) * DH;
  Frag kf[KSQ], vf[KSQ];
#pragma unroll
  for (int ks = 0; ks < KSQ; ++ks) {
    kf[ks] = load_kc<NP>(g.k + (long)key * g.ld + ks * 16 + 8 * hh, key < Tq);
    vf[ks] = load_kc<NP>(g.v + (long)key * g.ld + ks * 16 + 8 * hh, key < Tq);
  }
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dk[dt] = zero16(), dv[dt] = zero16();
  const bool live = key < g.len;  // masked keys keep zero gradients
  const int nqt = (int)blockIdx.x * 128 < g.len ? (Tq + 31) / 32 : 0;  // uniform over the workgroup: a fully masked key block skips the walk
  StageRegs<DH> sq, sd;
  if (nqt > 0) {
    stage_issue<DH>(sq, g.q, g.ld, 0, Tq, tid);
    stage_issue<DH>(sd, dob, D, 0, Tq, tid);
    stage_commit<DH, NP, true, true>(sq, qt_lds[0], tid);
    stage_commit<DH, NP, true, true>(sd, dot_lds[0], tid);
  }
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const bf16* q_t = qt_lds[qt & 1];
    const bf16* do_t = dot_lds[qt & 1];
    if (qt + 1 < nqt) {
      stage_issue<DH>(sq, g.q, g.ld, (qt + 1) * 32, Tq, tid);
      stage_issue<DH>(sd, dob, D, (qt + 1) * 32, Tq, tid);
    }
    f32x16 sc = zero16(), dp = zero16();
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      sc = mfma3<NP>(row_frag<DH, NP, true, true>(q_t, r, ks, hh), kf[ks], sc);
      dp = mfma3<NP>(row_frag<DH, NP, true, true>(do_t, r, ks, hh), vf[ks], dp);
    }
    // lse / delta of the 16 query rows of this lane: rows come in runs of four (acc_row), one 16-byte load per run when the
    // tile is inside the utterance and T' % 4 == 0
    float lse_r[16], del_r[16];
    const bool vec_rows = (Tq & 3) == 0 && qt * 32 + 32 <= Tq;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const long o4 = (long)z * Tq + qt * 32 + 8 * gq + 4 * hh;
      if (vec_rows) {
        const f32x4 a = *(const f32x4*)(lse + o4), b = *(const f32x4*)(delta + o4);
#pragma unroll
        for (int e = 0; e < 4; ++e) lse_r[4 * gq + e] = a[e], del_r[4 * gq + e] = b[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool in = qt * 32 + 8 * gq + 4 * hh + e < Tq;
          lse_r[4 * gq + e] = in ? lse[o4 + e] : 1.0e30f, del_r[4 * gq + e] = in ? delta[o4 + e] : 0.0f;  // past the end: p = exp2(-1e30) = 0
        }
      }
    }
    f32x16 pdv, dsv;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {  // register quad gq = four consecutive queries of this lane's key: indices Tq apart
      float dm[4];
      ds.mul4s((uint64_t)(g.pbase + (long)(qt * 32 + 8 * gq + 4 * hh) * Tq + key), (uint32_t)Tq, dm);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * gq + e;
        const float p = live ? __builtin_amdgcn_exp2f(sc[i] * scale2 - lse_r[i]) : 0.0f;  // rows past the end carry lse = 1e30
        pdv[i] = p * dm[e];
        dsv[i] = scale * p * (dp[i] * dm[e] - del_r[i]);
      }
    }
    Frag pdf[2], dsf[2];
    pdf[0] = acc_frag<NP>(pdv, 0), pdf[1] = acc_frag<NP>(pdv, 1), dsf[0] = acc_frag<NP>(dsv, 0), dsf[1] = acc_frag<NP>(dsv, 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        dv[dt] = mfma3<NP>(slot_frag<DH, NP, true, true>(do_t, dt * 32 + r, ks, hh), pdf[ks], dv[dt]);
        dk[dt] = mfma3<NP>(slot_frag<DH, NP, true, true>(q_t, dt * 32 + r, ks, hh), dsf[ks], dk[dt]);
      }
    if (qt + 1 < nqt) {
      stage_commit<DH, NP, true, true>(sq, qt_lds[(qt + 1) & 1], tid);
      stage_commit<DH, NP, true, true>(sd, dot_lds[(qt + 1) & 1], tid);
    }
    __syncthreads();
  }
  if (key < Tq && !(EECT_ABLATE_ATTN_STORES && H > 0)) {
    float* dst = dqkv + ((long)(z / H) * Tq + key) * 3 * D + (z % H) * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        *(f32x4*)(dst + D + dt * 32 + 8 * gq + 4 * hh) = (f32x4){dk[dt][4 * gq], dk[dt][4 * gq + 1], dk[dt][4 * gq + 2], dk[dt][4 * gq + 3]};
        *(f32x4*)(dst + 2 * D + dt * 32 + 8 * gq + 4 * hh) = (f32x4){dv[dt][4 * gq], dv[dt][4 * gq + 1], dv[dt][4 * gq + 2], dv[dt][4 * gq + 3]};
      }
  }
}

bool attn_fused_supported(int D, int H) {
  const int dh = H > 0 ? D / H : 0;
  return H > 0 && D % H == 0 && (dh == 32 || dh == 64) && D % 8 == 0;
}

template <typename K, typename... Args>
static hipError_t launch_with_lds(K kernel, dim3 grid, size_t lds, hipStream_t st, Args... args) {
  if (hipError_t e = eec::ensure_max_lds((const void*)kernel, (int)lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, args...);
  return hipGetLastError();
}
template <int DH, int NP>
static size_t fwd_lds() { return 2 * (TileLds<DH, NP, true, false>::kElems + TileLds<DH, NP, false, true>::kElems) * sizeof(bf16); }
template <int DH, int NP>
static size_t dq_lds() { return 2 * (TileLds<DH, NP, true, true>::kElems + TileLds<DH, NP, true, false>::kElems) * sizeof(bf16); }
template <int DH, int NP>
static size_t dkv_lds() { return 4 * TileLds<DH, NP, true, true>::kElems * sizeof(bf16); }

hipError_t launch_attn_fwd_fused(const float* qkv, const int32_t* key_len, float* ctx, float* lse, int B, int H, int Tq, int D, int np, Drop d,
                                 hipStream_t st) {
  if (!attn_fused_supported(D, H)) return hipErrorInvalidValue;
  const int dh = D / H;
  const float scale = 1.0f / sqrtf((float)dh);
  const dim3 grid((Tq + 127) / 128, B * H);
#define EECT_AF(DHv, NPv) return launch_with_lds(attn_fwd_kernel<DHv, NPv>, grid, fwd_lds<DHv, NPv>(), st, qkv, key_len, ctx, lse, H, Tq, D, scale, d)
  if (dh == 32) { if (np == 1) EECT_AF(32, 1); else EECT_AF(32, 3); }
  else { if (np == 1) EECT_AF(64, 1); else EECT_AF(64, 3); }
#undef EECT_AF
}

hipError_t launch_attn_bwd_fused(const float* qkv, const int32_t* key_len, const float* ctx, const float* d_ctx, const float* lse, float* delta,
                                 float* dqkv, int B, int H, int Tq, int D, int np, Drop d, hipStream_t st) {
  if (!attn_fused_supported(D, H)) return hipErrorInvalidValue;
  const int dh = D / H;
  const float scale = 1.0f / sqrtf((float)dh);
  const long n4 = (long)B * Tq * (D / 4);
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, ctx, d_ctx, delta, H, Tq, D, dh, n4);
  const dim3 grid((Tq + 127) / 128, B * H);
  const float* cdelta = delta;
#define EECT_AB(DHv, NPv)                                                                                                                     \
  do {                                                                                                                                        \
    if (hipError_t e = launch_with_lds(attn_bwd_dq_kernel<DHv, NPv>, grid, dq_lds<DHv, NPv>(), st, qkv, key_len, d_ctx, lse, cdelta, dqkv, H, \
                                       Tq, D, scale, d);                                                                                      \
        e != hipSuccess)                                                                                                                      \
      return e;                                                                                                                               \
    return launch_with_lds(attn_bwd_dkv_kernel<DHv, NPv>, grid, dkv_lds<DHv, NPv>(), st, qkv, key_len, d_ctx, lse, cdelta, dqkv, H, Tq, D,    \
                           scale, d);                                                                                                         \
  } while (0)
  if (dh == 32) { if (np == 1) EECT_AB(32, 1); else EECT_AB(32, 3); }
  else { if (np == 1) EECT_AB(64, 1); else EECT_AB(64, 3); }
#undef EECT_AB
}

}  // namespace eect
