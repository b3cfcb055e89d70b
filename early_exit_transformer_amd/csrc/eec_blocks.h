// Building blocks shared by the row-tile kernels (linear.hip, conv.hip) and the fused chain kernel (ffn.hip):
// every block works on one 64-row tile of the flattened (utterance, frame) axis with 512 threads.
#pragma once
#include "eec_kernels.h"

namespace eec {

constexpr int kLinThreads = 512;
constexpr int kLinLds = 2 * kAPlane;  // 67584: the activation planes only
#ifndef EEC_LPF
#define EEC_LPF 4
#endif
constexpr int kLPF = EEC_LPF;  // weight fragments (1 KiB each per plane) a wave keeps in flight

__device__ __forceinline__ int vt_perm(int t) {  // swap bits 2 and 3: MFMA k-order of an accumulator-fed operand
  return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
}

// acc[mt][0][4g + j] <- bias[n0 + 8g + 4hh + j]  (swapped orientation: register = output feature)
template <int MT>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[MT][1], const float* __restrict__ bias_n0) {
  const int hh = lane_id() >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bb = *(const float4*)(bias_n0 + 8 * g + 4 * hh);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt][0][4 * g + 0] = bb.x;
      acc[mt][0][4 * g + 1] = bb.y;
      acc[mt][0][4 * g + 2] = bb.z;
      acc[mt][0][4 * g + 3] = bb.w;
    }
  }
}

__device__ __forceinline__ const uint4* wfrag_lane(const uint4* wp, int nt) {
  return wp + (size_t)nt * (kD / 16) * 128 + lane_id();
}

// Q / K / V products of one 64-row tile whose LayerNormed activation planes are in LDS (NP format), the
// planes being complete and visible (caller has passed a workgroup barrier).  `rq` holds the first kLPF
// k-steps of this wave's Q weight tile (filled by the caller, ahead of time).  Wave w owns output columns
// [32w, 32w+32) of each of Q, K, V.  (the in_proj of nn.MultiheadAttention.)
template <int NP>
__device__ __forceinline__ void qkv_body(char* smem, const QkvArgs& a, int row0, WRing<NP, kLPF>& rq) {
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int dh = kD / a.H;
  const char* a_lane = smem + (lane & 31) * kALd + hh * 16;
  WRing<NP, kLPF> rk;
  // row -> (utterance, frame) of this lane's two frames
  int rb[2], rt[2];
  bool ok[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = row0 + mt * 32 + (lane & 31);
    ok[mt] = row < a.M;
    rb[mt] = row / a.Tq;
    rt[mt] = row - rb[mt] * a.Tq;
  }
  const int n0 = 32 * w, hd = n0 / dh, d0 = n0 - hd * dh + 4 * hh;

  f32x16 acc[2][1];
  // ---- Q ----
  ring_fill<NP, kLPF, 1>(rk, wfrag_lane(a.wp, 8 + w), 0, kD / 16);
  acc_init_bias<2>(acc, a.bias + n0);
  gemm_ring<NP, kD / 16, 1, true, kLPF>(acc, a_lane, kALd, kAPlane, wfrag_lane(a.wp, w), 0, rq);
  ring_fill<NP, kLPF, 1>(rq, wfrag_lane(a.wp, 16 + w), 0, kD / 16);  // V weights, in flight during the K pass
  {
    const float scale = kLog2e * rsqrtf((float)dh);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      if (ok[mt]) {
        half_t* dst = a.q + ((size_t)(rb[mt] * a.H + hd) * a.Tp + rt[mt]) * dh + d0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = to_half_sat(acc[mt][0][4 * g + j] * scale);
          *(h4*)(dst + 8 * g) = o;
        }
      }
  }
  // ---- K ----
  acc_init_bias<2>(acc, a.bias + kD + n0);
  gemm_ring<NP, kD / 16, 1, true, kLPF>(acc, a_lane, kALd, kAPlane, wfrag_lane(a.wp, 8 + w), 0, rk);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    if (ok[mt]) {
      half_t* dst = a.k + ((size_t)(rb[mt] * a.H + hd) * a.Tp + rt[mt]) * dh + d0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = to_half_sat(acc[mt][0][4 * g + j]);
        *(h4*)(dst + 8 * g) = o;
      }
    }
  // ---- V: V^T[b][h][d][perm(t)], 2-byte stores contiguous along t ----
  acc_init_bias<2>(acc, a.bias + 2 * kD + n0);
  gemm_ring<NP, kD / 16, 1, true, kLPF>(acc, a_lane, kALd, kAPlane, wfrag_lane(a.wp, 16 + w), 0, rq);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    if (ok[mt]) {
      const size_t voff = ((size_t)(rb[mt] * a.H + hd) * dh + d0) * a.Tp + vt_perm(rt[mt]);
      half_t* dst = a.vt + voff;
      // V's own fp16 rounding is the largest single contribution of the attention path to the log-prob error
      // (CPU emulation: 1.7e-4 of 2.5e-4 on the default model, 1.3e-3 on a stress configuration): keep the residual
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[mt][0][4 * g + j];
          const half_t hi = to_half_sat(v);
          dst[(size_t)(8 * g + j) * a.Tp] = hi;
          if (a.vt_lo) a.vt_lo[voff + (size_t)(8 * g + j) * a.Tp] = (half_t)(v - (float)hi);
        }
    }
}

// ---------------------------------------------------------------------------
// Conv-module tail, first half: depthwise Conv1d(K<=31, 'same', zero padding at UTTERANCE
// ends) + folded BatchNorm(eval) + SiLU of the tile's 64 rows, written as activation planes (NP format) for
// the pointwise-2 GEMM.  LDS: planes at smem, staged GLU rows + taps behind them (kDwLds in total).
// The caller must barrier before the planes are read.
// ---------------------------------------------------------------------------
constexpr int kDwTaps = 31;
constexpr int kDwHalo = (kDwTaps - 1) / 2;
constexpr int kDwFrames = 16;
constexpr int kDwWin = kDwFrames + kDwTaps - 1;            // 46
constexpr int kGRows = kTileRows + kDwTaps - 1;             // 94 staged rows
constexpr int kGLd = kD * 2;                                // 512 B per staged row
constexpr int kDwLds = 2 * kAPlane + kGRows * kGLd + kDwTaps * kD * 4;  // 67584 + 48128 + 31744 = 147456
constexpr int kDPF = 4;


template <int NP>
__device__ __forceinline__ void dw_front(char* smem, const DwArgs& d, int M, int row0) {
  char* lds_g = smem + 2 * kAPlane;
  float* lds_w = (float*)(lds_g + kGRows * kGLd);
  const int Tq = d.Tq;
  {  // stage the 94 GLU rows and the 31 folded tap rows: every global load is issued before the first LDS write
    constexpr int GIT = (kGRows * 32 + 511) / 512, WIT = (kDwTaps * kD / 4 + 511) / 512;
    uint4 gv[GIT];
    float4 wv[WIT];
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int p = it * 512 + threadIdx.x, rl = p >> 5, c16 = p & 31, row = row0 - kDwHalo + rl;
      gv[it] = make_uint4(0, 0, 0, 0);
      if (rl < kGRows && row >= 0 && row < M) gv[it] = *(const uint4*)(d.g + (size_t)row * kD + c16 * 8);
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int p = it * 512 + threadIdx.x;
      wv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < kDwTaps * kD / 4) wv[it] = ((const float4*)d.wfold)[p];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int p = it * 512 + threadIdx.x, rl = p >> 5, c16 = p & 31;
      if (rl < kGRows) *(uint4*)(lds_g + rl * kGLd + c16 * 16) = gv[it];
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int p = it * 512 + threadIdx.x;
      if (p < kDwTaps * kD / 4) ((float4*)lds_w)[p] = wv[it];
    }
  }
  __syncthreads();
  {
    const int c = (threadIdx.x & 127) * 2, tg = threadIdx.x >> 7;  // 2 channels x frames [16 tg, 16 tg + 16)
    const int m0 = row0 + tg * kDwFrames;                           // first output row of this thread
    typedef float f32x2 __attribute__((ext_vector_type(2)));  // two adjacent channels: v_pk_fma_f32
    f32x2 win[kDwWin];
#pragma unroll
    for (int k = 0; k < kDwWin; ++k) {
      const h2 g = *(const h2*)(lds_g + (tg * kDwFrames + k) * kGLd + c * 2);
      win[k] = (f32x2){(float)g[0], (float)g[1]};
    }
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
    const bool one_utt = b0 == m_last / Tq;  // wave-uniform (tg is per wave pair)
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
        const f32x2 wv = *(const f32x2*)(lds_w + j * kD + c);
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
        const f32x2 wv = *(const f32x2*)(lds_w + j * kD + c);
#pragma unroll
        for (int i = 0; i < kDwFrames; ++i) {
          const bool ok = (i + j) >= klo[i] && (i + j) <= khi[i];
          acc[i].x = fmaf(ok ? wv.x : 0.f, win[i + j].x, acc[i].x);
          acc[i].y = fmaf(ok ? wv.y : 0.f, win[i + j].y, acc[i].y);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kDwFrames; ++i) {
      const int rl = tg * kDwFrames + i;
      float vx = silu_f(acc[i].x), vy = silu_f(acc[i].y);
      if (row0 + rl >= M) vx = vy = 0.f;
      const hl2_t sp = split2<NP>(vx, vy);
      *(h2*)(smem + rl * kALd + c * 2) = sp.hi;
      if (NP == 3) *(h2*)(smem + kAPlane + rl * kALd + c * 2) = sp.lo;
    }
  }
}

// pointwise-2 of the conv module on the planes dw_front left: acc2[mt][0] = bias + planes . W^T for this wave's
// 32 output columns [32w, 32w+32) (swapped orientation: lane = frame).  `r` holds the first kDPF k-steps.
template <int NP>
__device__ __forceinline__ void pw2_gemm(f32x16 (&acc2)[2][1], const char* smem, const ProjResArgs& a,
                                         WRing<NP, kDPF, 1>& r) {
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const uint4* w_lane = a.wp + (size_t)w * (kD / 16) * 128 + lane;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bb = *(const float4*)(a.bias + 32 * w + 8 * g + 4 * hh);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      acc2[mt][0][4 * g + 0] = bb.x;
      acc2[mt][0][4 * g + 1] = bb.y;
      acc2[mt][0][4 * g + 2] = bb.z;
      acc2[mt][0][4 * g + 3] = bb.w;
    }
  }
  const char* a_lane = smem + (lane & 31) * kALd + hh * 16;
  gemm_ring<NP, kD / 16, 1, true, kDPF>(acc2, a_lane, kALd, kAPlane, w_lane, 0, r);
}

// swapped-orientation accumulators (lane = frame, register quad = 4 consecutive columns of this wave's 32)
// -> fp32 tile [64][kELd] in LDS
__device__ __forceinline__ void acc_swapped_to_etile(char* lds_e, const f32x16 (&acc)[2][1]) {
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    char* dst = lds_e + (mt * 32 + (lane & 31)) * kELd + (32 * w + 4 * hh) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(float4*)(dst + g * 32) = make_float4(acc[mt][0][4 * g], acc[mt][0][4 * g + 1], acc[mt][0][4 * g + 2], acc[mt][0][4 * g + 3]);
  }
}

}  // namespace eec
