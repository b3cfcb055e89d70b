// Conv-module tail: depthwise Conv1d(K<=31, groups=D, 'same') + BatchNorm(eval, folded) + SiLU fused into
// the pointwise-2 GEMM + residual.  Layout everywhere: [utterance][frame][channel], channel fastest.
#include "eec_kernels.h"

namespace eec {

// ---------------------------------------------------------------------------
// Fused conv-module tail:
//     x += PW2( SiLU( BN( DW_K(g) ) ) ) + b      g = GLU output, fp16 [B*T'][256]
// One 512-thread workgroup = 64 consecutive rows of the flattened (utterance, frame) axis.
//   1. stage g rows [row0-15, row0+79) (94 x 512 B) and the folded taps [31][256] in LDS;
//   2. depthwise conv + folded BatchNorm + SiLU: thread = 2 adjacent channels x 16 frames, window read
//      from LDS (lanes = consecutive channels: conflict-free); zero padding at UTTERANCE ends only
//      (reference behaviour: padded frames inside the batch tensor do leak into valid frames);
//      result split hi/lo and written straight into the A-plane layout of the GEMM - the conv
//      output never touches HBM;
//   3. pointwise-2 as the ring-pipelined MFMA GEMM (wave w -> columns [32w, 32w+32)), residual add
//      straight from the accumulators.
// ---------------------------------------------------------------------------
constexpr int kDwTaps = 31;
constexpr int kDwHalo = (kDwTaps - 1) / 2;
constexpr int kDwFrames = 16;
constexpr int kDwWin = kDwFrames + kDwTaps - 1;            // 46
constexpr int kGRows = kTileRows + kDwTaps - 1;             // 94 staged rows
constexpr int kGLd = kD * 2;                                // 512 B per staged row
constexpr int kDwLds = 2 * kAPlane + kGRows * kGLd + kDwTaps * kD * 4;  // 67584 + 48128 + 31744 = 147456
constexpr int kDPF = 4;

EEC_TL_DEFINE(dw)
template <int NP>
__global__ __launch_bounds__(512, 2) void dw_pw2_kernel(DwArgs d, ProjResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_g = smem + 2 * kAPlane;
  float* lds_w = (float*)(lds_g + kGRows * kGLd);
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int row0 = blockIdx.x * kTileRows;
  const int M = a.M, Tq = d.Tq;

  EEC_TL_STAMP(dw, 0);
  WRing<NP, kDPF, 1> r;
  const uint4* w_lane = a.wp + (size_t)w * (kD / 16) * 128 + lane;
  ring_fill<NP, kDPF, 1>(r, w_lane, 0, kD / 16);
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
  EEC_TL_STAMP(dw, 1);
  __syncthreads();
  EEC_TL_STAMP(dw, 2);
  {
    const int c = (threadIdx.x & 127) * 2, tg = threadIdx.x >> 7;  // 2 channels x frames [16 tg, 16 tg + 16)
    const int m0 = row0 + tg * kDwFrames;                           // first output row of this thread
    float2 win[kDwWin];
#pragma unroll
    for (int k = 0; k < kDwWin; ++k) {
      const h2 g = *(const h2*)(lds_g + (tg * kDwFrames + k) * kGLd + c * 2);
      win[k] = make_float2((float)g[0], (float)g[1]);
    }
    const float2 bias = *(const float2*)(d.bfold + c);
    float2 acc[kDwFrames];
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
          if (k < klo || k > khi) win[k] = make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < kDwTaps; ++j) {
        const float2 wv = *(const float2*)(lds_w + j * kD + c);
#pragma unroll
        for (int i = 0; i < kDwFrames; ++i) {
          acc[i].x = fmaf(wv.x, win[i + j].x, acc[i].x);
          acc[i].y = fmaf(wv.y, win[i + j].y, acc[i].y);
        }
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
        const float2 wv = *(const float2*)(lds_w + j * kD + c);
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
  EEC_TL_STAMP(dw, 3);
  __syncthreads();
  EEC_TL_STAMP(dw, 4);
  f32x16 acc2[2][1];
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
  EEC_TL_STAMP(dw, 5);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = row0 + mt * 32 + (lane & 31);
    if (row < M) {
      float* xr = a.x + (size_t)row * kD + 32 * w + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = *(const float4*)(xr + 8 * g);
        v.x += acc2[mt][0][4 * g + 0];
        v.y += acc2[mt][0][4 * g + 1];
        v.z += acc2[mt][0][4 * g + 2];
        v.w += acc2[mt][0][4 * g + 3];
        *(float4*)(xr + 8 * g) = v;
      }
    }
  }
  EEC_TL_STAMP(dw, 6);
}

hipError_t launch_dw_pw2(const DwArgs& d, const ProjResArgs& a, int np, hipStream_t st) {
  static bool d3 = false, d1 = false;
  auto k = np == 3 ? dw_pw2_kernel<3> : dw_pw2_kernel<1>;
  bool& done = np == 3 ? d3 : d1;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kDwLds);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows), dim3(512), kDwLds, st, d, a);
  return hipGetLastError();
}

// BN(eval) folded into the depthwise taps, taps zero-padded/centred to 31:
//   y = ((sum_j w_j x + b) - rm) * s + beta,  s = gamma / sqrt(rv + 1e-5)
__global__ void fold_dw_kernel(const float* dw_w, const float* dw_b, const float* bn_w, const float* bn_b,
                               const float* bn_rm, const float* bn_rv, int ksize, float* wfold, float* bfold) {
  const int c = threadIdx.x;
  const float s = bn_w[c] / sqrtf(bn_rv[c] + 1e-5f);
  const int shift = (kDwTaps - 1) / 2 - (ksize - 1) / 2;
  for (int j = 0; j < kDwTaps; ++j) {
    const int jj = j - shift;
    wfold[j * kD + c] = (jj >= 0 && jj < ksize) ? dw_w[c * ksize + jj] * s : 0.f;
  }
  bfold[c] = (dw_b[c] - bn_rm[c]) * s + bn_b[c];
}

hipError_t launch_fold_dw(const float* dw_w, const float* dw_b, const float* bn_w, const float* bn_b,
                          const float* bn_rm, const float* bn_rv, int ksize, float* wfold, float* bfold,
                          hipStream_t st) {
  hipLaunchKernelGGL(fold_dw_kernel, dim3(1), dim3(kD), 0, st, dw_w, dw_b, bn_w, bn_b, bn_rm, bn_rv, ksize, wfold,
                     bfold);
  return hipGetLastError();
}

}  // namespace eec
