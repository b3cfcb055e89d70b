// K = D projection kernels of a Conformer layer.  One 512-thread workgroup (8 waves) owns a row tile (Geo<D>: 64 rows at
// D = 256, 32 at D = 512); wave w owns the NW = D / 256 adjacent 32-wide output column tiles [NW w, NW (w + 1)) of every
// D-wide output.  All products run in the swapped orientation (frame on the lane, 4 consecutive output features per
// register quad), so every epilogue stores straight from the accumulators - no LDS round trip, no barrier after the
// prologue.  Weight fragments stream through the register ring of eec_device.h; the ring of a pass
// is filled during the previous pass (or the prologue).  Accumulators start at the bias.
//   qkv_kernel           LN -> in_proj (N=3D) -> Q (pre-scaled), K, V^T in attention layouts     
//   proj_residual_kernel x += A . W^T + b   (attention out_proj a6; conv pointwise-2 a7)
//   proj_glu_kernel      out_proj + residual -> LN -> pointwise-1 (N=2D) -> GLU -> fp16, one launch (a6, a7)
//   head_kernel          exit head: Linear(D,V) -> log_softmax -> fp32 log-probs                    (a9)
#include "eec_blocks.h"

namespace eec {

EEC_TL_DEFINE(qkv)
EEC_TL_DEFINE(glu)
// ---------------------------------------------------------------------------
template <int D, int NP>
__global__ __launch_bounds__(kLinThreads, 2) void qkv_kernel(QkvArgs a) {
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = wave_id();
  const int row0 = row_tile_index() * G::kRows;
  ProjStream<NP, kLPF, G::kNW> rq;
  EEC_TL_STAMP(qkv, 0);
  rows_f32_to_planes<D, NP, true>(smem, a.x, row0, a.M, a.ln_g, a.ln_b, [&]() {
    proj_fill<D, NP, kLPF, G::kNW>(rq, WMat{a.wp, a.wf8}, G::kNW * w);
  });
  EEC_TL_STAMP(qkv, 1);
  __syncthreads();
  EEC_TL_STAMP(qkv, 2);
  qkv_body<D, NP>(smem, a, row0, rq);
}

template <int D>
static hipError_t launch_qkv_d(const QkvArgs& a, int np, hipStream_t st) {
  using G = Geo<D>;
  auto k = np == 8 ? qkv_kernel<D, 8> : np == 3 ? qkv_kernel<D, 3> : qkv_kernel<D, 1>;
  hipError_t e = ensure_max_lds((const void*)k, kLinLds<D>);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + G::kRows - 1) / G::kRows), dim3(kLinThreads), kLinLds<D>, st, a);
  return hipGetLastError();
}
hipError_t launch_qkv(const QkvArgs& a, int np, hipStream_t st) {
  return a.D == 512 ? launch_qkv_d<512>(a, np, st) : launch_qkv_d<256>(a, np, st);
}

// ---------------------------------------------------------------------------
// fp16 planes in global ([M][D] hi, [M][D] lo) -> LDS planes, 512 threads.
template <int D, int NP>
struct PlaneRegs {
  static constexpr int IT = Geo<D>::kRows * (D / 8) / kLinThreads;  // 16-byte pieces per thread per plane: 4
  uint4 h[IT], l[NP != 1 ? IT : 1];
};
// issue: all global loads of the tile's pieces (rows x D/8 sixteen-byte pieces);
// commit: the LDS writes.  Split so that a caller can queue further loads (weight ring, residual rows) behind
// them and so that no LDS write sits between two loads (one memory round trip for the tile, not four).
template <int D, int NP>
__device__ __forceinline__ void planes_issue512(PlaneRegs<D, NP>& pr, const half_t* __restrict__ hi,
                                                const half_t* __restrict__ lo, int row0, int M) {
  constexpr int PPR = D / 8;
#pragma unroll
  for (int it = 0; it < PlaneRegs<D, NP>::IT; ++it) {
    const int piece = it * kLinThreads + threadIdx.x, rl = piece / PPR, c16 = piece % PPR, row = row0 + rl;
    pr.h[it] = make_uint4(0, 0, 0, 0);
    if (NP != 1) pr.l[it] = make_uint4(0, 0, 0, 0);
    if (row < M) {
      pr.h[it] = *(const uint4*)(hi + (size_t)row * D + c16 * 8);
      if (NP != 1) pr.l[it] = *(const uint4*)(lo + (size_t)row * D + c16 * 8);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int D, int NP>
__device__ __forceinline__ void planes_commit512(char* lds_act, const PlaneRegs<D, NP>& pr) {
  using G = Geo<D>;
  constexpr int PPR = D / 8;
#pragma unroll
  for (int it = 0; it < PlaneRegs<D, NP>::IT; ++it) {
    const int piece = it * kLinThreads + threadIdx.x, rl = piece / PPR, c16 = piece % PPR;
    *(uint4*)(lds_act + rl * G::kALd + c16 * 16) = pr.h[it];
    if (NP == 3) *(uint4*)(lds_act + G::kAPlane + rl * G::kALd + c16 * 16) = pr.l[it];
    if (NP == 8)  // the residual plane arrives as fp16: its top bytes (e5m2) go to the permuted byte plane, 8 k at a time
    {
      typedef _Float16 h8v __attribute__((ext_vector_type(8)));
      const h8v lg = __builtin_bit_cast(h8v, pr.l[it]) * (half_t)kF8ALoGain;  // gain-compensated (eec_device.h, kF8ALoGain)
      *(uint2*)(lds_act + G::kAPlane + rl * G::kA8Ld + lo8_pos(c16 * 8)) = top_bytes(__builtin_bit_cast(uint4, lg));
      if (EEC_X_HI8) *(uint2*)(lds_act + G::kAPlane + rl * G::kA8Ld + G::kA8Hi + lo8_pos(c16 * 8)) = top_bytes(pr.h[it]);
    }
  }
}

// x += A . Wo^T + b on 32-row tiles (the sub-step plan and the legacy layer; these K = D kernels are latency-bound, so
// two small workgroups per CU beat one large one).
template <int D, int NP>
__global__ __launch_bounds__(kLinThreads, 2) void proj_residual_kernel(ProjResArgs a) {
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = 32, NW = G::kNW;
  static_assert(NP != 8, "the 32-row projection keeps the fragment formats (sub-step plan, legacy layer)");
  constexpr int PLANE = ROWS * G::kALd;
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int row0 = blockIdx.x * ROWS;
  const char* a_lane = smem + (lane & 31) * G::kALd + hh * 16;
  WRing<NP, kLPF, NW> r;
  ring_fill<NP, kLPF, NW>(r, wfrag_lane<G::kKS>(a.wp, NW * w), (size_t)G::kKS * 128, G::kKS);
  for (int piece = threadIdx.x; piece < ROWS * (D / 8); piece += kLinThreads) {
    const int rl = piece / (D / 8), c16 = piece % (D / 8), row = row0 + rl;
    uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
    if (row < a.M) {
      vh = *(const uint4*)(a.a_hi + (size_t)row * D + c16 * 8);
      if (NP == 3) vl = *(const uint4*)(a.a_lo + (size_t)row * D + c16 * 8);
    }
    *(uint4*)(smem + rl * G::kALd + c16 * 16) = vh;
    if (NP == 3) *(uint4*)(smem + PLANE + rl * G::kALd + c16 * 16) = vl;
  }
  __syncthreads();
  f32x16 acc[1][NW];
  acc_init_bias_np<NP, 1, NW>(acc, a.bias + 32 * NW * w);
  gemm_ring<NP, G::kKS, NW, true, kLPF, NoSide, 0, 1, !kMfma16For<NP>, true>(acc, a_lane, G::kALd, PLANE, wfrag_lane<G::kKS>(a.wp, NW * w),
                                                      (size_t)G::kKS * 128, r);
  acc_swapped_add_rows<D, 1, NW>(a.x, row0, a.M, acc, 32 * NW * w);
}

template <int D>
static hipError_t launch_proj_residual_d(const ProjResArgs& a, int np, hipStream_t st) {
  constexpr int ROWS = 32, LDS = 2 * ROWS * Geo<D>::kALd;
  auto k = np == 3 ? proj_residual_kernel<D, 3> : proj_residual_kernel<D, 1>;
  hipError_t e = ensure_max_lds((const void*)k, LDS);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + ROWS - 1) / ROWS), dim3(kLinThreads), LDS, st, a);
  return hipGetLastError();
}
hipError_t launch_proj_residual(const ProjResArgs& a, int np, hipStream_t st) {
  return a.D == 512 ? launch_proj_residual_d<512>(a, np, st) : launch_proj_residual_d<256>(a, np, st);
}

// ---------------------------------------------------------------------------
// Fused attention tail + conv-module head (one launch instead of two):
//   x += O . Wo^T + bo ;  g = GLU( LN_conv(x) . Wpw1^T + bpw1 )
// The out-proj result crosses from "wave owns NW column tiles" to "wave owns kRPW rows" through an fp32
// tile in LDS; the updated rows are LayerNorm'ed while still in registers and become the planes of
// the GLU product.  The GLU weight stream starts before the tile exchange.
template <int D>
constexpr int kProjGluLds = 2 * Geo<D>::kAPlane + Geo<D>::kETile;  // 134144 / 132608

// Attention of one row tile straight into the activation planes of the out-proj product (the fused launch below).
// Preconditions (checked by the launcher): 8 heads -- wave w IS head w --, T' a multiple of the tile's rows, so all of
// the tile's queries belong to ONE utterance and T' needs no key padding.  The products are those of attn_kernel
// (attention.hip): S^T = K . Q^T and O^T += V^T . P^T with the query on the MFMA lane, online softmax over blocks of
// 64 keys, keys >= enc_len masked; but K and V^T fragments come straight from global memory (they are contiguous 2 KiB /
// 64-B-per-row pieces, read by the four workgroups of the utterance out of L2), double-buffered in registers: there is
// no LDS staging, no barrier, and the normalised O never leaves the CU -- it is split hi / lo and written into the
// LDS planes at columns [32 w .. ) of the tile's rows.
// X3 (exact mode f16x3, round 4): Q and K arrive with fp16 residual planes and the probabilities are split hi / lo in registers,
// so both products run as three fp16 MFMA products like every GEMM of the mode (what attn_kernel<.., 3, true> does in its own
// launch, with K and its residual staged in LDS).  The second fragment set costs the registers of one key tile: blocks of 32 keys
// instead of 64.
#ifndef EEC_ATTN_LAZY
#define EEC_ATTN_LAZY 0  // experiment (8 measured: 42.4 -> 42.2 us, not adopted): log2 units a block maximum may exceed the running maximum before the accumulators are rescaled (0: every block)
#endif
template <int D, int NP, bool X3 = false>
__device__ __forceinline__ void attn_tile_to_planes(char* smem, const AttnArgs& a, int row0) {
  using G = Geo<D>;
  constexpr int MT = G::kMT, DH = D / 8, KSQ = DH / 16, DT = DH / 32, KB = X3 ? 1 : 2;
  constexpr float kNegBig = -1.0e30f;
  const int lane = lane_id(), w = wave_id(), r = lane & 31, hh = lane >> 5;
  const int b = row0 / a.Tq, q0 = row0 - b * a.Tq, bh = b * a.H + w;
  const int len_raw = a.enc_len[b];  // requested first, consumed after the first K / V requests have left
  int len = 0, nkt = 0;
  const bool v2 = a.vt_lo != nullptr;  // uniform
  h8 qf[MT][KSQ];
  [[maybe_unused]] h8 qfl[MT][KSQ];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      // fragment-major planes (a.vt_frag, head dim 32: [b][h][32-frame block][k-step][lane][8]): 1 KiB contiguous per fragment
      const size_t qoff = a.vt_frag ? (size_t)bh * a.Tp * DH + ((size_t)(((q0 >> 5) + mt) * 2 + ks) * 64 + lane) * 8
                                    : ((size_t)bh * a.Tp + q0 + mt * 32 + r) * DH + 8 * hh + ks * 16;
      qf[mt][ks] = *(const h8*)(a.q + qoff);
      if constexpr (X3) qfl[mt][ks] = *(const h8*)(a.q_lo + qoff);
    }
  }
  f32x16 o[MT][DT];
  float m_run[MT], l_run[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    m_run[mt] = kNegBig, l_run[mt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[mt][dt][i] = 0.f;
  }
  const half_t* kbase = a.k + (size_t)bh * a.Tp * DH;
  const half_t* vbase = a.vt + (size_t)bh * DH * a.Tp;
  const half_t* vlbase = v2 ? a.vt_lo + (size_t)bh * DH * a.Tp : vbase;
  [[maybe_unused]] const half_t* klbase = X3 ? a.k_lo + (size_t)bh * a.Tp * DH : kbase;
  struct Blk {  // K and V^T fragments of one block of KB key tiles (X3: K's residual too)
    h8 k[KB][KSQ], v[KB][DT][2], vl[KB][DT][2], kl[X3 ? KB : 1][X3 ? KSQ : 1];
  };
  const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // `last`: the last tile that may be read (a tile past the utterance's keys is re-read and masked away below).  The FIRST
  // block is fetched against the padded frame count, which is a kernel argument: its requests leave together with the Q
  // fragments', before the utterance's length has arrived (a dependent load: ~2 us at kernel start, measured 7 k cycles from
  // kernel entry to the first K / V request)
  auto fetch = [&](Blk& f, int kt0, int last) {
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const int kt = min(kt0 + j, last);
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) {
        const size_t koff = a.vt_frag ? ((size_t)(kt * 2 + ks) * 64 + lane) * 8 : ((size_t)kt * 32 + r) * DH + ks * 16 + 8 * hh;
        f.k[j][ks] = *(const h8*)(kbase + koff);
        if constexpr (X3) f.kl[j][ks] = *(const h8*)(klbase + koff);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          // fragment-major planes (a.vt_frag; head dim 32): the fragment is 1 KiB contiguous in lane order
          const size_t off = a.vt_frag ? ((size_t)(kt * 2 + ks) * 64 + lane) * 8 : (size_t)(dt * 32 + r) * a.Tp + kt * 32 + ks * 16 + 8 * hh;
          f.v[j][dt][ks] = *(const h8*)(vbase + off);
          f.vl[j][dt][ks] = v2 ? *(const h8*)(vlbase + off) : zero8;
        }
    }
  };
  auto block = [&](const Blk& f, int kt0) {
    f32x16 sc[KB][MT];
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[j][mt][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
          if constexpr (X3) {  // the two correction products first, the main product last (as the GEMMs of this mode)
            sc[j][mt] = mfma16(f.kl[j][ks], qf[mt][ks], sc[j][mt]);
            sc[j][mt] = mfma16(f.k[j][ks], qfl[mt][ks], sc[j][mt]);
          }
          sc[j][mt] = mfma16(f.k[j][ks], qf[mt][ks], sc[j][mt]);
        }
      }
    const int key0 = kt0 * 32;
    if (key0 + KB * 32 > len) {  // block touches the masked tail (or runs past the last tile): wave-uniform test
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (key0 + j * 32 + acc_row(i, lane) >= len || kt0 + j >= nkt) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) sc[j][mt][i] = kNegBig;
          }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float tmax = kNegBig;
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, sc[j][mt][i]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      // Lazy running maximum: it moves (and the row sums and the O accumulators are rescaled) only when some row of the tile sees a score
      // more than kLazyMax above it -- a wave-uniform test; otherwise the block's probabilities are taken against the standing maximum
      // (<= 2^kLazyMax: exact in the fp16 pair / fp32 sums all the same) and the exp2, the 16 DT + 1 multiplies per row tile are skipped.
      const bool bump = EEC_ATTN_LAZY == 0 || __builtin_amdgcn_ballot_w64(tmax > m_run[mt] + (float)EEC_ATTN_LAZY) != 0;
      if (bump) {
        const float m_up = fmaxf(m_run[mt], tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run[mt] - m_up);
        m_run[mt] = m_up;
        l_run[mt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[mt][dt][i] *= alpha;
      }
      const float m_new = m_run[mt];
      float psum = 0.f;
      h8 pf[KB][2];
      [[maybe_unused]] h8 pfl[KB][2];
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const float p0 = __builtin_amdgcn_exp2f(sc[j][mt][i] - m_new), p1 = __builtin_amdgcn_exp2f(sc[j][mt][i + 1] - m_new);
          psum += p0 + p1;
          if constexpr (X3) {  // hi / lo pair in 4 instructions (the softmax VALU, not the MFMAs, bounds this phase in the exact mode)
            const hl2_t sp = split2_mix(p0, p1);
            pf[j][i >> 3][i & 7] = sp.hi[0], pf[j][i >> 3][(i & 7) + 1] = sp.hi[1];
            pfl[j][i >> 3][i & 7] = sp.lo[0], pfl[j][i >> 3][(i & 7) + 1] = sp.lo[1];
          } else {
            pf[j][i >> 3][i & 7] = (half_t)p0, pf[j][i >> 3][(i & 7) + 1] = (half_t)p1;
          }
        }
      l_run[mt] += psum;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            if (v2) o[mt][dt] = mfma16(f.vl[j][dt][ks], pf[j][ks], o[mt][dt]);
            if constexpr (X3) o[mt][dt] = mfma16(f.v[j][dt][ks], pfl[j][ks], o[mt][dt]);
            o[mt][dt] = mfma16(f.v[j][dt][ks], pf[j][ks], o[mt][dt]);
          }
      }
    }
  };
  Blk fa, fb;
#ifdef EEC_ATTN_LATE_FETCH  // A/B knob: the round-2 order (first fetch behind the length)
  len = min(len_raw, a.Tq);
  nkt = (len + 31) / 32;
  if (nkt > 0) fetch(fa, 0, nkt - 1);
#else
  fetch(fa, 0, a.Tp / 32 - 1);
  __builtin_amdgcn_sched_barrier(0);
  len = min(len_raw, a.Tq);
  nkt = (len + 31) / 32;
#endif
  // static priority for the younger half of the workgroup: waves 4-7 lose every issue arbitration against their SIMD partners
  // (wave 7 reached its first key block 4.4 k cycles after wave 0 and stayed behind: the tile's barrier waited 6.8 k cycles for it;
  // profiles/r03_proj_glu_timeline.txt).  One setprio, no per-phase flips; A/B 37.7 -> 36.6 us per launch.
#ifndef EEC_ATTN_PRIO
#define EEC_ATTN_PRIO 1
#endif
  if (EEC_ATTN_PRIO > 0 && w >= 4) __builtin_amdgcn_s_setprio(EEC_ATTN_PRIO);
  EEC_TL_STAMP(glu, 11);
  for (int kt0 = 0; kt0 < nkt; kt0 += 2 * KB) {
    if (kt0 + KB < nkt) fetch(fb, kt0 + KB, nkt - 1);
    block(fa, kt0);
    EEC_TL_STAMP(glu, kt0 == 0 ? 12 : 14);
    if (kt0 + KB < nkt) {
      if (kt0 + 2 * KB < nkt) fetch(fa, kt0 + 2 * KB, nkt - 1);
      block(fb, kt0 + KB);
    }
    EEC_TL_STAMP(glu, kt0 == 0 ? 13 : 15);
  }
  if (EEC_ATTN_PRIO > 0) __builtin_amdgcn_s_setprio(0);
  // normalise and write the planes.  No valid key at all (length 0): the installed torch returns zeros ("safe softmax").
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float l_tot = l_run[mt] + __shfl_xor(l_run[mt], 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int rl = mt * 32 + r;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = w * DH + dt * 32 + 8 * g + 4 * hh;
        const hl2_t s0 = split2<(NP == 1 ? 1 : 3)>(o[mt][dt][4 * g] * inv, o[mt][dt][4 * g + 1] * inv);
        const hl2_t s1 = split2<(NP == 1 ? 1 : 3)>(o[mt][dt][4 * g + 2] * inv, o[mt][dt][4 * g + 3] * inv);
        h4 hi, lo;
        hi.xy = s0.hi, hi.zw = s1.hi, lo.xy = s0.lo, lo.zw = s1.lo;
        *(h4*)(smem + rl * G::kALd + col * 2) = hi;
        if (NP == 3) *(h4*)(smem + G::kAPlane + rl * G::kALd + col * 2) = lo;
        if (NP == 8) {
          h4 lg;
          lg.xy = lo8_gain(s0.lo), lg.zw = lo8_gain(s1.lo);
          const uint2 lb = __builtin_bit_cast(uint2, lg);
          *(unsigned*)(smem + G::kAPlane + rl * G::kA8Ld + lo8_pos(col)) = __builtin_amdgcn_perm(lb.y, lb.x, 0x07050301u);
          if (EEC_X_HI8) {
            const uint2 hb = __builtin_bit_cast(uint2, hi);
            *(unsigned*)(smem + G::kAPlane + rl * G::kA8Ld + G::kA8Hi + lo8_pos(col)) = __builtin_amdgcn_perm(hb.y, hb.x, 0x07050301u);
          }
        }
      }
  }
}

// FUSED: the tile's attention runs in the prologue (attn_tile_to_planes) instead of loading the O planes a separate
// attention launch wrote: two launches per Conformer layer instead of three, no O round trip through HBM.
// FUSED: 0 = the O planes come from a separate attention launch; 1 = fused attention; 2 = fused attention of the exact mode (X3)
template <int D, int NP, int FUSED>
__global__ __launch_bounds__(kLinThreads, 2) void proj_glu_kernel(ProjResArgs a, GluArgs gl, AttnArgs at) {
  using G = Geo<D>;
  constexpr int MT = G::kMT, NW = G::kNW, RPW = G::kRPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_e = smem + 2 * G::kAPlane;
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int row0 = row_tile_index() * G::kRows;
  ProjStream<NP, kLPF, NW> r, rv;
  const WMat wo{a.wp, a.wf8}, wg{gl.wp, gl.wf8};
  EEC_TL_STAMP(glu, 0);
  RowV<G::kQ> xres[RPW];
  if constexpr (FUSED != 0) {
    attn_tile_to_planes<D, NP, FUSED == 2>(smem, at, row0);
    proj_fill<D, NP, kLPF, NW>(r, wo, NW * w);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = row0 + w * RPW + i;
      xres[i] = zero_row<G::kQ>();
      if (row < a.M) xres[i] = load_row<D>(a.x + (size_t)row * D, lane);
    }
  } else {
    PlaneRegs<D, NP> pr;
    planes_issue512<D, NP>(pr, a.a_hi, a.a_lo, row0, a.M);  // needed first: queued ahead of the weight ring
    proj_fill<D, NP, kLPF, NW>(r, wo, NW * w);
    // residual rows of this wave: requested now, consumed after the out-proj GEMM (their latency hides under it)
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = row0 + w * RPW + i;
      xres[i] = zero_row<G::kQ>();
      if (row < a.M) xres[i] = load_row<D>(a.x + (size_t)row * D, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    planes_commit512<D, NP>(smem, pr);
  }
  EEC_TL_STAMP(glu, 1);
  __syncthreads();
  EEC_TL_STAMP(glu, 2);
  {
    f32x16 acc[MT][NW];
    acc_init_bias_np<NP, MT, NW>(acc, a.bias + 32 * NW * w);
    proj_gemm<D, NP, kLPF, NW, MT>(acc, smem, r, wo, NW * w);
    EEC_TL_STAMP(glu, 3);
    proj_fill<D, NP, kLPF, NW>(rv, wg, NW * w);  // GLU value weights: in flight during the exchange
    acc_swapped_to_etile<MT, NW>(lds_e, G::kELd, acc, 32 * NW * w);
  }
  EEC_TL_STAMP(glu, 4);
  __syncthreads();  // tile complete; every wave is done reading the O planes
  EEC_TL_STAMP(glu, 5);
  {
    const RowV<G::kQ> g = load_row<D>(gl.ln_g, lane), bt = load_row<D>(gl.ln_b, lane);
    RowV<G::kQ> v[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rl = w * RPW + i, row = row0 + rl;
      v[i] = xres[i];
#pragma unroll
      for (int q = 0; q < G::kQ; ++q) {
        const float4 e = *(const float4*)(lds_e + rl * G::kELd + (q * 256 + lane * 4) * 4);
        v[i].p[q].x += e.x, v[i].p[q].y += e.y, v[i].p[q].z += e.z, v[i].p[q].w += e.w;
      }
      if (row < a.M) store_row<D>(a.x + (size_t)row * D, v[i], lane);
    }
    layer_norm_rows<D, RPW>(v, g, bt);
    rows_to_planes<D, NP, RPW>(smem, v, w * RPW, row0, a.M, true, lane);
  }
  EEC_TL_STAMP(glu, 6);
  __syncthreads();
  EEC_TL_STAMP(glu, 7);
  constexpr int TQ = D / 32;  // column tiles of the value half; the gate half follows
  ProjStream<NP, kLPF, NW> rg;
  proj_fill<D, NP, kLPF, NW>(rg, wg, TQ + NW * w);
  f32x16 av[MT][NW], ag[MT][NW];
  acc_init_bias_np<NP, MT, NW>(av, gl.bias + 32 * NW * w);
  proj_gemm<D, NP, kLPF, NW, MT>(av, smem, rv, wg, NW * w);
  EEC_TL_STAMP(glu, 8);
  acc_init_bias_np<NP, MT, NW>(ag, gl.bias + D + 32 * NW * w);
  proj_gemm<D, NP, kLPF, NW, MT>(ag, smem, rg, wg, TQ + NW * w);
  EEC_TL_STAMP(glu, 9);
  if constexpr (D == 256) {
    // the wave's [64 rows][32 columns] of GLU output through its own staging area (the exchange tile is dead by now): 16 rows x 64 B
    // per store instruction instead of 32 rows x 16 B (see qkv_body's whole-line stores, eec_blocks.h)
    char* const stg = lds_e + w * kQkvStageBytes;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gate = ag[mt][0][4 * g + i];
          o[i] = to_half_sat(av[mt][0][4 * g + i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * gate)));
        }
        *(h4*)(stg + (mt * 32 + (lane & 31)) * kQkvStageLd + (8 * g + 4 * hh) * 2) = o;
      }
    asm volatile("" ::: "memory");  // the wave's own area, served in order: only the compiler has to keep the order
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rl = (lane >> 2) + 16 * k;
      const uint4 v = *(const uint4*)(stg + rl * kQkvStageLd + (lane & 3) * 16);
      if (row0 + rl < a.M) *(uint4*)(gl.g + (size_t)(row0 + rl) * D + 32 * w + (lane & 3) * 8) = v;
    }
    EEC_TL_STAMP(glu, 10);
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = row0 + mt * 32 + (lane & 31);
    if (row < a.M) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        half_t* dst = gl.g + (size_t)row * D + 32 * (NW * w + j) + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float gate = ag[mt][j][4 * g + i];
            o[i] = to_half_sat(av[mt][j][4 * g + i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * gate)));
          }
          store_maybe_nt<EEC_NT_G != 0>((h4*)(dst + 8 * g), o);
        }
      }
    }
  }
  EEC_TL_STAMP(glu, 10);
}

template <int D, int FUSED>
static hipError_t launch_proj_glu_d(const ProjResArgs& a, const GluArgs& g, const AttnArgs& at, int np, hipStream_t st) {
  auto k = proj_glu_kernel<D, 3, FUSED>;
  if constexpr (FUSED == 2) {
    if (np != 3) return hipErrorInvalidValue;  // the exact mode only
  } else {
    k = np == 8 ? proj_glu_kernel<D, 8, FUSED> : np == 3 ? proj_glu_kernel<D, 3, FUSED> : proj_glu_kernel<D, 1, FUSED>;
  }
  hipError_t e = ensure_max_lds((const void*)k, kProjGluLds<D>);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + Geo<D>::kRows - 1) / Geo<D>::kRows), dim3(kLinThreads), kProjGluLds<D>, st, a, g, at);
  return hipGetLastError();
}
hipError_t launch_proj_glu(const ProjResArgs& a, const GluArgs& g, int np, hipStream_t st) {
  const AttnArgs none{};
  return a.D == 512 ? launch_proj_glu_d<512, 0>(a, g, none, np, st) : launch_proj_glu_d<256, 0>(a, g, none, np, st);
}
// attention + out_proj + LN + pointwise-1 + GLU in one launch; usable when attn_fusable() holds
bool attn_fusable(const AttnArgs& at, int D) {
  // d_model 512 (head dim 64) is left to the separate attention launch: two double-buffered K / V^T fragment sets of that
  // size do not fit the register file beside the accumulators
  return D == 256 && at.H == 8 && at.dh * 8 == D && at.Tq % Geo<256>::kRows == 0 && at.Tp == at.Tq;
}
hipError_t launch_attn_proj_glu(const AttnArgs& at, const ProjResArgs& a, const GluArgs& g, int np, hipStream_t st) {
  if (!attn_fusable(at, a.D)) return hipErrorInvalidValue;
  if (at.q_lo || at.k_lo) {  // exact mode: Q, K, V all with residual planes, three products per attention product
    if (np != 3 || !at.q_lo || !at.k_lo || !at.vt_lo) return hipErrorInvalidValue;
    return launch_proj_glu_d<256, 2>(a, g, at, np, st);
  }
  return launch_proj_glu_d<256, 1>(a, g, at, np, st);
}

// ---------------------------------------------------------------------------
// Exit head.  V <= 256, V % 32 == 0.  Wave w owns vocabulary tile w of the GEMM (idle if 32w >= V).
constexpr int kHeadELd = (256 + 4) * 4;  // row stride of the [rows][V <= 256] fp32 logit tile
template <int D>
constexpr int kHeadLds = kLinLds<D> > Geo<D>::kRows * kHeadELd ? kLinLds<D> : Geo<D>::kRows * kHeadELd;

template <int D, int NP>
__device__ __forceinline__ void head_body(char* smem, const HeadArgs& a) {
  using G = Geo<D>;
  constexpr int MT = G::kMT, RPW = G::kRPW;
  const int lane = lane_id(), w = wave_id();
  const int row0 = row_tile_index() * G::kRows;
  const bool active = 32 * w < a.V;  // wave-uniform
  const WMat wm{a.wp, a.wf8};
  ProjStream<NP, kLPF, 1> r;
  rows_f32_to_planes<D, NP, false>(smem, a.x, row0, a.M, nullptr, nullptr,
                                   [&]() { if (active) proj_fill<D, NP, kLPF, 1>(r, wm, w); });
  __syncthreads();
  f32x16 acc[MT][1];
  if (active) {
    acc_init_bias_np<NP, MT, 1>(acc, a.bias + 32 * w);
    proj_gemm<D, NP, kLPF, 1, MT>(acc, smem, r, wm, w);
  }
  // logits -> fp32 exchange tile (over the dead activation planes) -> each wave finishes RPW whole frames: the
  // log-sum-exp is a wave reduction and every output row leaves as one contiguous store (V * 4 bytes)
  __syncthreads();  // every wave is done reading the planes
  if (active) acc_swapped_to_etile<MT, 1>(smem, kHeadELd, acc, 32 * w);
  __syncthreads();
  const int c0 = lane * 4;  // this lane's 4 vocabulary entries
  const bool has = c0 < a.V;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int rl = w * RPW + i, row = row0 + rl;
    float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (has) v = *(const float4*)(smem + rl * kHeadELd + c0 * 4);
    const float mx = wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    const float sm = wave_sum(has ? __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx) : 0.f);
    const float lse = mx + __logf(sm);
    if (has && row < a.M) store_maybe_nt<EEC_NT_HEAD != 0>((f32x4*)(a.out + (size_t)row * a.V + c0), (f32x4){v.x - lse, v.y - lse, v.z - lse, v.w - lse});
  }
}

template <int D, int NP>
__global__ __launch_bounds__(kLinThreads, 2) void head_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  head_body<D, NP>(smem, a);
}

// blockIdx.y = exit: the per-exit pointers are picked out of the argument block with a wave-uniform index
template <int D, int NP>
__global__ __launch_bounds__(kLinThreads, 2) void head_batch_kernel(HeadBatchArgs b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int e = blockIdx.y;
  const HeadArgs a{b.x[e], b.M, b.V, b.D, b.wp[e], b.bias[e], b.out + (size_t)e * b.M * b.V, b.wf8[e]};
  head_body<D, NP>(smem, a);
}

template <int D>
static hipError_t launch_head_d(const HeadArgs& a, int np, hipStream_t st) {
  auto k = np == 8 ? head_kernel<D, 8> : np == 3 ? head_kernel<D, 3> : head_kernel<D, 1>;
  hipError_t e = ensure_max_lds((const void*)k, kHeadLds<D>);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + Geo<D>::kRows - 1) / Geo<D>::kRows), dim3(kLinThreads), kHeadLds<D>, st, a);
  return hipGetLastError();
}
hipError_t launch_head(const HeadArgs& a, int np, hipStream_t st) {
  return a.D == 512 ? launch_head_d<512>(a, np, st) : launch_head_d<256>(a, np, st);
}

template <int D>
static hipError_t launch_head_batch_d(const HeadBatchArgs& a, int np, hipStream_t st) {
  auto k = np == 8 ? head_batch_kernel<D, 8> : np == 3 ? head_batch_kernel<D, 3> : head_batch_kernel<D, 1>;
  hipError_t e = ensure_max_lds((const void*)k, kHeadLds<D>);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + Geo<D>::kRows - 1) / Geo<D>::kRows, a.E), dim3(kLinThreads), kHeadLds<D>, st, a);
  return hipGetLastError();
}
hipError_t launch_head_batch(const HeadBatchArgs& a, int np, hipStream_t st) {
  if (a.E < 1 || a.E > kMaxHeadExits) return hipErrorInvalidValue;
  return a.D == 512 ? launch_head_batch_d<512>(a, np, st) : launch_head_batch_d<256>(a, np, st);
}

}  // namespace eec
