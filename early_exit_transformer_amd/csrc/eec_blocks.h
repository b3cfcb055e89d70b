// Building blocks shared by the row-tile kernels (linear.hip, conv.hip) and the fused chain kernel (ffn.hip):
// every block works on one row tile (Geo<D>::kRows rows) of the flattened (utterance, frame) axis with 512 threads.
#pragma once
#include "eec_kernels.h"

namespace eec {

constexpr int kLinThreads = 512;
template <int D>
constexpr int kLinLds = 2 * Geo<D>::kAPlane;  // the activation planes only
#ifndef EEC_LPF
#define EEC_LPF 4
#endif
constexpr int kLPF = EEC_LPF;  // weight fragments (1 KiB each per plane) a wave keeps in flight per column tile

// per-wave LDS staging of qkv_body's whole-line stores: [64 frames][80 B] for Q / K, [32 dims][144 B] for V^T (rows padded by 16 B)
constexpr int kQkvStageLd = 80, kQkvStageLdV = 144, kQkvStageBytes = 64 * kQkvStageLd;
static_assert(32 * kQkvStageLdV <= kQkvStageBytes, "V^T staging fits the Q / K staging area");
constexpr int kQkvFragLd = 1056;  // staged fragment stride (fragment-major V^T): 2 x (512 B + 16 B)
static_assert(4 * kQkvFragLd <= kQkvStageBytes && (kQkvFragLd / 2) % 16 == 0, "fragment-major V^T staging");
__device__ __forceinline__ int vt_perm(int t) {  // swap bits 2 and 3: MFMA k-order of an accumulator-fed operand
  return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
}

// acc[mt][nt][4g + j] <- bias[n0 + 32 nt + 8g + 4hh + j]  (swapped orientation: register = output feature)
template <int MT, int NT>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[MT][NT], const float* __restrict__ bias_n0, float scale = 1.0f) {
  const int hh = lane_id() >> 5;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *(const float4*)(bias_n0 + 32 * nt + 8 * g + 4 * hh);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][nt][4 * g + 0] = bb.x * scale;
        acc[mt][nt][4 * g + 1] = bb.y * scale;
        acc[mt][nt][4 * g + 2] = bb.z * scale;
        acc[mt][nt][4 * g + 3] = bb.w * scale;
      }
    }
}

// The same start values in the layout the format's k-loop wants: the quadrant layout for the split format on the 16x16x32 shape
// (eec_device.h, EEC_MFMA16: register 4 (2 ra + cb) + j <-> output feature 16 ra + 8 hh + 4 u + j, both frame blocks cb), so that
// the product needs no conversion on the way in; the standard layout otherwise.
template <int NP, int MT, int NT>
__device__ __forceinline__ void acc_init_bias_np(f32x16 (&acc)[MT][NT], const float* __restrict__ bias_n0, float scale = 1.0f) {
  if constexpr (kMfma16For<NP>) {
    const int lane = lane_id(), hh = lane >> 5, u = (lane >> 4) & 1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ra = 0; ra < 2; ++ra) {
        const float4 bb = *(const float4*)(bias_n0 + 32 * nt + 16 * ra + 8 * hh + 4 * u);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            acc[mt][nt][4 * (2 * ra + cb) + 0] = bb.x * scale;
            acc[mt][nt][4 * (2 * ra + cb) + 1] = bb.y * scale;
            acc[mt][nt][4 * (2 * ra + cb) + 2] = bb.z * scale;
            acc[mt][nt][4 * (2 * ra + cb) + 3] = bb.w * scale;
          }
      }
  } else {
    acc_init_bias<MT, NT>(acc, bias_n0, scale);
  }
}

// first fragment (this lane) of n-tile nt of a packed [N][16 KS] matrix
template <int KS>
__device__ __forceinline__ const uint4* wfrag_lane(const uint4* wp, int nt) {
  return wp + (size_t)nt * KS * 128 + lane_id();
}

// ---------------------------------------------------------------------------
// One K = D product of a row tile against NT adjacent 32-wide column tiles of a packed weight matrix, swapped orientation
// (lane = frame), activation planes in LDS in the NP format:
//   NP = 1 / 3: fp16 hi (/ + lo) planes, weights as hi (/ lo) fragments (wp);
//   NP = 8    : fp16 hi plane + e5m2 lo byte plane, weights as the f8 record stream (wf8): fp16 main product + two
//               block-scaled fp8 correction products (eec_device.h, gemm_ring_f8) -- 2 pass-equivalents, 3 B / weight.
// A ProjStream holds the weight registers in flight; proj_fill starts a tile set's stream (one product ahead of its use).
constexpr int kProjNWB = 2;  // lo8 + scale group buffers of an f8 stream (rolling)
template <int NP, int PF, int NT>
struct ProjStream {
  WRing<(NP == 8 ? 1 : NP), PF, NT> r;
  WGroupF8<NT> wg[NP == 8 ? kProjNWB : 1];
};
struct WMat {  // the two packings of one weight matrix [N][D]
  const uint4* wp;   // fragment planes (NP = 1, 3)
  const uint4* wf8;  // f8 record stream (NP = 8)
};
template <int D, int NP, int PF, int NT>
__device__ __forceinline__ void proj_fill(ProjStream<NP, PF, NT>& st, const WMat& w, int t0) {
  constexpr int KS = D / 16, NG = D / 64;
  if constexpr (NP == 8) {
    const uint4* rec = w.wf8 + (size_t)t0 * NG * kF8Rec + lane_id();
    ring_fill_f8<PF, NT>(st.r, rec, (size_t)NG * kF8Rec);
#pragma unroll
    for (int g = 0; g < kProjNWB; ++g) f8_group_load<NT>(st.wg[g], rec + (size_t)g * kF8Rec, (size_t)NG * kF8Rec);
  } else {
    ring_fill<NP, PF, NT>(st.r, wfrag_lane<KS>(w.wp, t0), (size_t)KS * 128, KS);
  }
}
template <int D, int NP, int PF, int NT, int MT>
__device__ __forceinline__ void proj_gemm(f32x16 (&acc)[MT][NT], const char* smem, ProjStream<NP, PF, NT>& st, const WMat& w, int t0,
                                          int row_stride_mul = 1) {
  using G = Geo<D>;
  constexpr int KS = D / 16, NG = D / 64;
  const int lane = lane_id(), hh = lane >> 5;
  const char* a_lane = smem + (lane & 31) * G::kALd + hh * 16;
  if constexpr (NP == 8) {
    const char* a8_lane = smem + G::kAPlane + (lane & 31) * G::kA8Ld + hh * 32;
    const uint4* rec = w.wf8 + (size_t)t0 * NG * kF8Rec + lane;
    gemm_ring_f8<NG, NT, true, PF, NoSide, 0, kProjNWB, 0, MT, false, (EEC_X_HI8 ? G::kA8Hi : 0)>(acc, a_lane, G::kALd, a8_lane, G::kA8Ld, rec,
                                                                                                    (size_t)NG * kF8Rec, st.r, st.wg);
  } else {
    // (the accumulators come from acc_init_bias_np: already in the layout this format's k-loop wants)
    gemm_ring<NP, KS, NT, true, PF, NoSide, 0, MT, !kMfma16For<NP>, true>(acc, a_lane, G::kALd, G::kAPlane, wfrag_lane<KS>(w.wp, t0), (size_t)KS * 128, st.r);
  }
  (void)row_stride_mul;
}

// Q / K / V products of one row tile whose LayerNormed activation planes are in LDS (NP format), the
// planes being complete and visible (caller has passed a workgroup barrier).  `rq` holds the first kLPF
// k-steps of this wave's Q weight tiles (filled by the caller, ahead of time).  Wave w owns output columns
// [32 NW w, 32 NW (w + 1)) of each of Q, K, V.  (the in_proj of nn.MultiheadAttention.)
template <int D, int NP>
__device__ __forceinline__ void qkv_body(char* smem, const QkvArgs& a, int row0, ProjStream<NP, kLPF, Geo<D>::kNW>& rq, char* stage = nullptr) {
  using G = Geo<D>;
  constexpr int MT = G::kMT, NW = G::kNW;
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int dh = D / a.H;
  // Whole-line stores (d_model 256, 8 heads, T' a multiple of the 64-row tile, `stage` = kQkvStageBytes of LDS per wave): a wave owns one
  // head, its 64 frames x 32 dims of Q (and K) are one contiguous 4 KB block per plane and its V^T block is 32 rows of 128 B -- the
  // accumulator tile is transposed through the wave's own staging area and leaves as 16-byte stores of consecutive lanes (1 KB
  // contiguous per instruction for Q / K, 8 rows x 128 B for V^T) instead of 8-byte pieces of 32 rows (Q, K) and 2-byte stores (V^T):
  // the stores cost 8 us of a 216 us chain launch before (same-box ablation, profiles/r04_ab_qkv_stores.txt)
  const bool fast = D == 256 && stage != nullptr && dh == 32 && a.Tq % G::kRows == 0 && a.Tp == a.Tq;  // uniform
  char* const stg = stage ? stage + w * kQkvStageBytes : nullptr;
  // hi / lo halves of a finished [64 frames][32 dims] accumulator tile -> the two planes at element offset `off` (frame-major rows of 32)
  [[maybe_unused]] auto store_rows_fast = [&](const f32x16 (&acc)[MT][NW], float scale, half_t* p_hi, half_t* p_lo, size_t off) {
    h4 lo4[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = acc[mt][0][4 * g + i] * scale;
          o[i] = to_half_sat(v);
          lo4[mt][g][i] = (half_t)(v - (float)o[i]);
        }
        *(h4*)(stg + (mt * 32 + (lane & 31)) * kQkvStageLd + (8 * g + 4 * hh) * 2) = o;
      }
    // the staging area is the wave's own and the LDS serves a wave's accesses in order: no barrier, but the COMPILER must keep the
    // accesses of different types in program order
    asm volatile("" ::: "memory");
    // the staged tile out: frame-major rows of 64 B, or (a.vt_frag) the four MFMA fragments of the tile -- (32-frame block j, 16 dims ks),
    // lane (frame r, dim half hh) -- each 1 KiB contiguous; either way 4 KB contiguous at `off`
    auto tile_out = [&](half_t* dstp) {
      if (a.vt_frag) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          *(uint4*)(dstp + off + k * 512 + lane * 8) = *(const uint4*)(stg + ((k >> 1) * 32 + (lane & 31)) * kQkvStageLd + (k & 1) * 32 + hh * 16);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = (lane >> 2) + 16 * k;
          *(uint4*)(dstp + off + (size_t)row * 32 + (lane & 3) * 8) = *(const uint4*)(stg + row * kQkvStageLd + (lane & 3) * 16);
        }
      }
    };
    tile_out(p_hi);
    asm volatile("" ::: "memory");
    if (p_lo) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) *(h4*)(stg + (mt * 32 + (lane & 31)) * kQkvStageLd + (8 * g + 4 * hh) * 2) = lo4[mt][g];
      asm volatile("" ::: "memory");
      tile_out(p_lo);
      asm volatile("" ::: "memory");
    }
  };
  const WMat wm{a.wp, a.wf8};
  ProjStream<NP, kLPF, NW> rk;
  // row -> (utterance, frame) of this lane's frames
  int rb[MT], rt[MT];
  bool ok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = row0 + mt * 32 + (lane & 31);
    ok[mt] = row < a.M;
    rb[mt] = row / a.Tq;
    rt[mt] = row - rb[mt] * a.Tq;
  }
  const int t0 = NW * w;            // first of this wave's column tiles inside Q (and K, V)
  constexpr int TQ = D / 32;        // column tiles per part
  int hd[NW], d0[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int n0 = 32 * (t0 + j);
    hd[j] = n0 / dh;
    d0[j] = n0 - hd[j] * dh + 4 * hh;
  }

  f32x16 acc[MT][NW];
  // ---- Q ----
  proj_fill<D, NP, kLPF, NW>(rk, wm, TQ + t0);
  acc_init_bias_np<NP, MT, NW>(acc, a.bias + 32 * t0);
  proj_gemm<D, NP, kLPF, NW, MT>(acc, smem, rq, wm, t0);
  proj_fill<D, NP, kLPF, NW>(rq, wm, 2 * TQ + t0);  // V weights, in flight during the K pass
  if (fast) {
    if constexpr (D == 256) {
      const int b0 = row0 / a.Tq, tt0 = row0 - b0 * a.Tq;
      store_rows_fast(acc, kLog2e * rsqrtf((float)dh), a.q, a.q_lo, ((size_t)(b0 * a.H + w) * a.Tp + tt0) * 32);
    }
  } else {
    const float scale = kLog2e * rsqrtf((float)dh);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (ok[mt]) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const size_t qoff = ((size_t)(rb[mt] * a.H + hd[j]) * a.Tp + rt[mt]) * dh + d0[j];
          half_t* dst = a.q + qoff;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            h4 o, ol;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = acc[mt][j][4 * g + i] * scale;
              o[i] = to_half_sat(v);
              ol[i] = (half_t)(v - (float)o[i]);
            }
            store_maybe_nt<EEC_NT_QKV != 0>((h4*)(dst + 8 * g), o);
            if (a.q_lo) store_maybe_nt<EEC_NT_QKV != 0>((h4*)(a.q_lo + qoff + 8 * g), ol);
          }
        }
      }
  }
  // ---- K ----
  acc_init_bias_np<NP, MT, NW>(acc, a.bias + D + 32 * t0);
  proj_gemm<D, NP, kLPF, NW, MT>(acc, smem, rk, wm, TQ + t0);
  if (fast) {
    if constexpr (D == 256) {
      const int b0 = row0 / a.Tq, tt0 = row0 - b0 * a.Tq;
      store_rows_fast(acc, 1.0f, a.k, a.k_lo, ((size_t)(b0 * a.H + w) * a.Tp + tt0) * 32);
    }
  } else
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    if (ok[mt]) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const size_t koff = ((size_t)(rb[mt] * a.H + hd[j]) * a.Tp + rt[mt]) * dh + d0[j];
        half_t* dst = a.k + koff;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h4 o, ol;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[mt][j][4 * g + i];
            o[i] = to_half_sat(v);
            ol[i] = (half_t)(v - (float)o[i]);
          }
          store_maybe_nt<EEC_NT_QKV != 0>((h4*)(dst + 8 * g), o);
          if (a.k_lo) store_maybe_nt<EEC_NT_QKV != 0>((h4*)(a.k_lo + koff + 8 * g), ol);
        }
      }
    }
  // ---- V: V^T[b][h][d][perm(t)], 2-byte stores contiguous along t ----
  acc_init_bias_np<NP, MT, NW>(acc, a.bias + 2 * D + 32 * t0);
  proj_gemm<D, NP, kLPF, NW, MT>(acc, smem, rq, wm, 2 * TQ + t0);
  if (fast) {
    if constexpr (D == 256) {
      // V^T[b][h][d][perm(t)]: the tile's 64 frames are 128 contiguous bytes of each of the head's 32 rows d
      const int b0 = row0 / a.Tq, tt0 = row0 - b0 * a.Tq;
      const size_t vbase = ((size_t)(b0 * a.H + w) * 32) * a.Tp + tt0;
      // staging offset of element (d, tcol) = base(tcol) + d * dstride in either layout: rows [d][tcol] (144-B rows), or fragment-major
      // -- fragment (key tile tcol / 32, k-step (tcol / 16) % 2), lane d + 32 ((tcol / 8) % 2), element tcol % 8, staged with 16 B of
      // padding per 512-B half fragment (the four pieces a store instruction touches would otherwise share their banks)
      const int dstride = a.vt_frag ? 16 : kQkvStageLdV;
      int sbase[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int tcol = vt_perm(mt * 32 + (lane & 31));
        sbase[mt] = a.vt_frag ? ((tcol >> 5) * 2 + ((tcol >> 4) & 1)) * kQkvFragLd + ((tcol >> 3) & 1) * (kQkvFragLd / 2) + (tcol & 7) * 2 : tcol * 2;
      }
      auto plane = [&](half_t* dstp, bool lo_plane) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = acc[mt][0][4 * g + i];
              const half_t hi = to_half_sat(v);
              *(half_t*)(stg + sbase[mt] + (8 * g + 4 * hh + i) * dstride) = lo_plane ? (half_t)(v - (float)hi) : hi;
            }
        asm volatile("" ::: "memory");
        if (a.vt_frag) {  // the tile's four fragments are 4 KB contiguous: [b][h][key tile][k-step][lane][8]
          half_t* const fb = dstp + (size_t)(b0 * a.H + w) * 32 * a.Tp + (size_t)(tt0 >> 5) * 1024;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            *(uint4*)(fb + k * 512 + lane * 8) = *(const uint4*)(stg + k * kQkvFragLd + (lane >> 5) * (kQkvFragLd / 2) + (lane & 31) * 16);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int d = (lane >> 3) + 8 * k;
            *(uint4*)(dstp + vbase + (size_t)d * a.Tp + (lane & 7) * 8) = *(const uint4*)(stg + d * kQkvStageLdV + (lane & 7) * 16);
          }
        }
        asm volatile("" ::: "memory");
      };
      plane(a.vt, false);
      if (a.vt_lo) plane(a.vt_lo, true);
    }
  } else
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    if (ok[mt]) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const size_t voff = ((size_t)(rb[mt] * a.H + hd[j]) * dh + d0[j]) * a.Tp + vt_perm(rt[mt]);
        half_t* dst = a.vt + voff;
        // V's own fp16 rounding is the largest single contribution of the attention path to the log-prob error
        // (CPU emulation: 1.7e-4 of 2.5e-4 on the default model, 1.3e-3 on a stress configuration): keep the residual
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[mt][j][4 * g + i];
            const half_t hi = to_half_sat(v);
            store_maybe_nt<EEC_NT_QKV != 0>(dst + (size_t)(8 * g + i) * a.Tp, hi);
            if (a.vt_lo) store_maybe_nt<EEC_NT_QKV != 0>(a.vt_lo + voff + (size_t)(8 * g + i) * a.Tp, (half_t)(v - (float)hi));
          }
      }
    }
}

// ---------------------------------------------------------------------------
// Conv-module tail, first half: depthwise Conv1d(K<=31, 'same', zero padding at UTTERANCE
// ends) + folded BatchNorm(eval) + SiLU of the tile's rows, written as activation planes (NP format) for
// the pointwise-2 GEMM.  LDS: planes at smem, staged GLU rows (+ the taps at D = 256) behind them (DwGeo<D>::kLds in
// total; at D = 512 the [31][512] fp32 taps no longer fit beside them and every thread reads its own 31 tap pairs from
// global memory, 63 KiB shared by every workgroup of the launch: L1 / L2 hits).
// The caller must barrier before the planes are read.
// ---------------------------------------------------------------------------
constexpr int kDwTaps = 31;
constexpr int kDwHalo = (kDwTaps - 1) / 2;
constexpr int kDwFrames = 16;
constexpr int kDwWin = kDwFrames + kDwTaps - 1;            // 46
constexpr int kDPF = 4;
template <int D>
struct DwGeo {
  using G = Geo<D>;
  static constexpr bool kTapsLds = D == 256;
  static constexpr int kGRows = G::kRows + kDwTaps - 1;    // 94 / 62 staged rows
  static constexpr int kGLd = D * 2;                       // bytes per staged row
  static constexpr int kLds = 2 * G::kAPlane + kGRows * kGLd + (kTapsLds ? kDwTaps * D * 4 : 0);  // 147456 / 130048
  static_assert((D / 2) * (G::kRows / kDwFrames) == 512, "thread = 2 channels x 16 frames");
};

template <int D, int NP>
__device__ __forceinline__ void dw_front(char* smem, const DwArgs& d, int M, int row0) {
  using G = Geo<D>;
  using DG = DwGeo<D>;
  constexpr int kGRows = DG::kGRows, kGLd = DG::kGLd;
  char* lds_g = smem + 2 * G::kAPlane;
  float* lds_w = (float*)(lds_g + kGRows * kGLd);
  const int Tq = d.Tq;
  const int c = (threadIdx.x % (D / 2)) * 2, tg = threadIdx.x / (D / 2);  // 2 channels x frames [16 tg, 16 tg + 16)
  typedef float f32x2 __attribute__((ext_vector_type(2)));  // two adjacent channels: v_pk_fma_f32
  [[maybe_unused]] f32x2 taps[DG::kTapsLds ? 1 : kDwTaps];
  {  // stage the GLU rows (and the folded tap rows): every global load is issued before the first LDS write
    constexpr int PPR = D / 8;  // 16-byte pieces per staged row
    constexpr int GIT = (kGRows * PPR + 511) / 512, WIT = DG::kTapsLds ? (kDwTaps * D / 4 + 511) / 512 : 1;
    uint4 gv[GIT];
    [[maybe_unused]] float4 wv[WIT];
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int p = it * 512 + threadIdx.x, rl = p / PPR, c16 = p % PPR, row = row0 - kDwHalo + rl;
      gv[it] = make_uint4(0, 0, 0, 0);
      if (rl < kGRows && row >= 0 && row < M) gv[it] = *(const uint4*)(d.g + (size_t)row * D + c16 * 8);
    }
    if constexpr (DG::kTapsLds) {
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const int p = it * 512 + threadIdx.x;
        wv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < kDwTaps * D / 4) wv[it] = ((const float4*)d.wfold)[p];
      }
    } else {
      static_range<0, kDwTaps>([&](auto jt) { taps[decltype(jt)::value] = *(const f32x2*)(d.wfold + decltype(jt)::value * D + c); });
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int p = it * 512 + threadIdx.x, rl = p / PPR, c16 = p % PPR;
      if (rl < kGRows) *(uint4*)(lds_g + rl * kGLd + c16 * 16) = gv[it];
    }
    if constexpr (DG::kTapsLds) {
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const int p = it * 512 + threadIdx.x;
        if (p < kDwTaps * D / 4) ((float4*)lds_w)[p] = wv[it];
      }
    }
  }
  __syncthreads();
  {
    const int m0 = row0 + tg * kDwFrames;  // first output row of this thread
    const f32x2 bias = *(const f32x2*)(d.bfold + c);
    f32x2 acc[kDwFrames];
#pragma unroll
    for (int i = 0; i < kDwFrames; ++i) acc[i] = bias;
    // window rows are flattened rows m0-15 .. m0+30.  When the 16 output frames lie in one utterance
    // (always, if T' % 16 == 0) the "same"-padding zeros are applied ONCE to the window (rows of the
    // neighbouring utterances, or outside [0, M)) and the tap loop stays branch-free; only a frame
    // group that straddles two utterances needs the per-output validity test.
    const int first = m0 - kDwHalo, last = m0 + kDwFrames - 1 + kDwHalo;
    const int m_last = min(m0 + kDwFrames - 1, M - 1);
    const int b0 = min(m0, M - 1) / Tq;
    const bool one_utt = b0 == m_last / Tq;  // wave-uniform (tg is uniform per wave)
    auto window = [&](int k) -> f32x2 {
      const h2 g = *(const h2*)(lds_g + (tg * kDwFrames + k) * kGLd + c * 2);
      return (f32x2){(float)g[0], (float)g[1]};
    };
    if constexpr (DG::kTapsLds) {
      // D = 256: the 46-frame window lives in registers, the taps are read from LDS once per tap
      f32x2 win[kDwWin];
#pragma unroll
      for (int k = 0; k < kDwWin; ++k) win[k] = window(k);
      if (one_utt) {
        const bool interior = first >= b0 * Tq && last < (b0 + 1) * Tq;
        if (!interior) {
          const int klo = b0 * Tq - first, khi = (b0 + 1) * Tq - 1 - first;
#pragma unroll
          for (int k = 0; k < kDwWin; ++k)
            if (k < klo || k > khi) win[k] = (f32x2){0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < kDwTaps; ++j) {
          const f32x2 wv = *(const f32x2*)(lds_w + j * D + c);
#pragma unroll
          for (int i = 0; i < kDwFrames; ++i) acc[i] = __builtin_elementwise_fma(wv, win[i + j], acc[i]);
        }
      } else {
        // per output row i the valid window slots are [klo, khi]: same utterance as the output row
        int klo[kDwFrames], khi[kDwFrames];
#pragma unroll
        for (int i = 0; i < kDwFrames; ++i) {
          const int m = min(m0 + i, M - 1), b = m / Tq;
          klo[i] = b * Tq - first;
          khi[i] = (b + 1) * Tq - 1 - first;
        }
#pragma unroll
        for (int j = 0; j < kDwTaps; ++j) {
          const f32x2 wv = *(const f32x2*)(lds_w + j * D + c);
#pragma unroll
          for (int i = 0; i < kDwFrames; ++i) {
            const bool ok = (i + j) >= klo[i] && (i + j) <= khi[i];
            acc[i].x = fmaf(ok ? wv.x : 0.f, win[i + j].x, acc[i].x);
            acc[i].y = fmaf(ok ? wv.y : 0.f, win[i + j].y, acc[i].y);
          }
        }
      }
    } else {
      // D = 512: the 31 tap pairs live in registers (loaded from global above), the window is walked frame by frame:
      // window frame k feeds output i through tap j = k - i.  Per output the taps still arrive in the order
      // j = 0 .. 30, so the sums are the same as the tap-major loop's.
      if (one_utt) {
        const int klo = b0 * Tq - first, khi = (b0 + 1) * Tq - 1 - first;
        static_range<0, kDwWin>([&](auto kt) {
          constexpr int k = decltype(kt)::value;
          f32x2 g = window(k);
          if (k < klo || k > khi) g = (f32x2){0.f, 0.f};
          static_range<(k - kDwTaps + 1 > 0 ? k - kDwTaps + 1 : 0), (k + 1 < kDwFrames ? k + 1 : kDwFrames)>([&](auto it) {
            constexpr int i = decltype(it)::value;
            acc[i] = __builtin_elementwise_fma(taps[k - i], g, acc[i]);
          });
        });
      } else {
        int klo[kDwFrames], khi[kDwFrames];
#pragma unroll
        for (int i = 0; i < kDwFrames; ++i) {
          const int m = min(m0 + i, M - 1), b = m / Tq;
          klo[i] = b * Tq - first;
          khi[i] = (b + 1) * Tq - 1 - first;
        }
        static_range<0, kDwWin>([&](auto kt) {
          constexpr int k = decltype(kt)::value;
          const f32x2 g = window(k);
          static_range<(k - kDwTaps + 1 > 0 ? k - kDwTaps + 1 : 0), (k + 1 < kDwFrames ? k + 1 : kDwFrames)>([&](auto it) {
            constexpr int i = decltype(it)::value;
            const bool ok = k >= klo[i] && k <= khi[i];
            acc[i].x = fmaf(ok ? taps[k - i].x : 0.f, g.x, acc[i].x);
            acc[i].y = fmaf(ok ? taps[k - i].y : 0.f, g.y, acc[i].y);
          });
        });
      }
    }
#pragma unroll
    for (int i = 0; i < kDwFrames; ++i) {
      const int rl = tg * kDwFrames + i;
      float vx = silu_f(acc[i].x), vy = silu_f(acc[i].y);
      if (row0 + rl >= M) vx = vy = 0.f;
      const hl2_t sp = split2<(NP == 1 ? 1 : 3)>(vx, vy);
      *(h2*)(smem + rl * G::kALd + c * 2) = sp.hi;
      if (NP == 3) *(h2*)(smem + G::kAPlane + rl * G::kALd + c * 2) = sp.lo;
      if (NP == 8) {  // e5m2 bytes of the two residuals (adjacent channels are adjacent in the permuted byte plane)
        const unsigned lb = __builtin_bit_cast(unsigned, lo8_gain(sp.lo));
        *(unsigned short*)(smem + G::kAPlane + rl * G::kA8Ld + lo8_pos(c)) = (unsigned short)(((lb >> 8) & 0xffu) | ((lb >> 16) & 0xff00u));
        if (EEC_X_HI8) {
          const unsigned hb = __builtin_bit_cast(unsigned, sp.hi);
          *(unsigned short*)(smem + G::kAPlane + rl * G::kA8Ld + G::kA8Hi + lo8_pos(c)) = (unsigned short)(((hb >> 8) & 0xffu) | ((hb >> 16) & 0xff00u));
        }
      }
    }
  }
}

// pointwise-2 of the conv module on the planes dw_front left: acc2[mt][j] = bias + planes . W^T for this wave's
// NW column tiles [32 NW w, ..) (swapped orientation: lane = frame).  `r` holds the first kDPF k-steps.
template <int D, int NP>
__device__ __forceinline__ void pw2_gemm(f32x16 (&acc2)[Geo<D>::kMT][Geo<D>::kNW], const char* smem, const ProjResArgs& a,
                                         ProjStream<NP, kDPF, Geo<D>::kNW>& r) {
  using G = Geo<D>;
  const int w = wave_id();
  acc_init_bias_np<NP, G::kMT, G::kNW>(acc2, a.bias + 32 * G::kNW * w);
  proj_gemm<D, NP, kDPF, G::kNW, G::kMT>(acc2, smem, r, WMat{a.wp, a.wf8}, G::kNW * w);
}

// swapped-orientation accumulators (lane = frame, register quad = 4 consecutive columns of this wave's tiles)
// -> fp32 tile [rows][e_ld bytes] in LDS; the wave's first column is col0.
template <int MT, int NT>
__device__ __forceinline__ void acc_swapped_to_etile(char* lds_e, int e_ld, const f32x16 (&acc)[MT][NT], int col0) {
  const int lane = lane_id(), hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      char* dst = lds_e + (mt * 32 + (lane & 31)) * e_ld + (col0 + 32 * nt + 4 * hh) * 4;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(dst + g * 32) = make_float4(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
    }
}

// swapped-orientation accumulators added to the fp32 rows x[row][col0 + ..] in place (residual epilogue)
template <int D, int MT, int NT>
__device__ __forceinline__ void acc_swapped_add_rows(float* __restrict__ x, int row0, int M, const f32x16 (&acc)[MT][NT], int col0) {
  const int lane = lane_id(), hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = row0 + mt * 32 + (lane & 31);
    if (row < M) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float* xr = x + (size_t)row * D + col0 + 32 * nt + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = *(const float4*)(xr + 8 * g);
          v.x += acc[mt][nt][4 * g + 0];
          v.y += acc[mt][nt][4 * g + 1];
          v.z += acc[mt][nt][4 * g + 2];
          v.w += acc[mt][nt][4 * g + 3];
          *(float4*)(xr + 8 * g) = v;
        }
      }
    }
  }
}

}  // namespace eec
