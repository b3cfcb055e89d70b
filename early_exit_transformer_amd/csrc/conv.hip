// Conv-module tail: depthwise Conv1d(K<=31, groups=D, 'same') + BatchNorm(eval, folded) + SiLU fused into
// the pointwise-2 GEMM + residual.  Layout everywhere: [utterance][frame][channel], channel fastest.
#include "eec_blocks.h"

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
EEC_TL_DEFINE(dw)
template <int NP>
__global__ __launch_bounds__(512, 2) void dw_pw2_kernel(DwArgs d, ProjResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int row0 = blockIdx.x * kTileRows;
  const int M = a.M;
  EEC_TL_STAMP(dw, 0);
  WRing<NP, kDPF, 1> r;
  ring_fill<NP, kDPF, 1>(r, a.wp + (size_t)w * (kD / 16) * 128 + lane, 0, kD / 16);
  dw_front<NP>(smem, d, M, row0);
  EEC_TL_STAMP(dw, 3);
  __syncthreads();
  EEC_TL_STAMP(dw, 4);
  f32x16 acc2[2][1];
  pw2_gemm<NP>(acc2, smem, a, r);
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
  auto k = np == 3 ? dw_pw2_kernel<3> : dw_pw2_kernel<1>;
  if (hipError_t e = ensure_max_lds((const void*)k, kDwLds); e != hipSuccess) return e;
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
