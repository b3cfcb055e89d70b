// Convolution-side kernels of the path (all HBM/VALU-bound, fp32 arithmetic):
//   dwconv_kernel      depthwise Conv1d(K<=31, groups=D, 'same') + BatchNorm(eval, folded) + SiLU   
// Layout everywhere: [utterance][frame][channel], channel fastest, so lanes run over
// channels and every global access is a contiguous row segment.
#include "eec_kernels.h"

namespace eec {

// ---------------------------------------------------------------------------
// Depthwise conv: thread = 2 adjacent channels x 16 output frames; the 46-frame input window
// lives in registers.  Zero padding only at utterance ends (reference behaviour: padded
// frames inside the batch tensor do leak into valid frames).
// ---------------------------------------------------------------------------
constexpr int kDwTaps = 31;
constexpr int kDwFrames = 16;

template <int NP>
__global__ __launch_bounds__(kThreads) void dwconv_kernel(DwArgs a) {
  const int c2 = threadIdx.x & 127, half = threadIdx.x >> 7;
  const int b = blockIdx.y;
  const int t0 = (blockIdx.x * 2 + half) * kDwFrames;
  if (t0 >= a.Tq) return;
  const int c = c2 * 2;
  float2 win[kDwFrames + kDwTaps - 1];
#pragma unroll
  for (int j = 0; j < kDwFrames + kDwTaps - 1; ++j) {
    const int t = t0 - (kDwTaps - 1) / 2 + j;
    float2 v = make_float2(0.f, 0.f);
    if (t >= 0 && t < a.Tq) {
      const h2 g = *(const h2*)(a.g + ((size_t)b * a.Tq + t) * kD + c);
      v = make_float2((float)g[0], (float)g[1]);
    }
    win[j] = v;
  }
  const float2 bias = *(const float2*)(a.bfold + c);
  float2 acc[kDwFrames];
#pragma unroll
  for (int i = 0; i < kDwFrames; ++i) acc[i] = bias;
#pragma unroll
  for (int j = 0; j < kDwTaps; ++j) {
    const float2 wv = *(const float2*)(a.wfold + j * kD + c);
#pragma unroll
    for (int i = 0; i < kDwFrames; ++i) {
      acc[i].x = fmaf(wv.x, win[i + j].x, acc[i].x);
      acc[i].y = fmaf(wv.y, win[i + j].y, acc[i].y);
    }
  }
#pragma unroll
  for (int i = 0; i < kDwFrames; ++i) {
    const int t = t0 + i;
    if (t < a.Tq) {
      h2 hi, lo;
      EEC_SPLIT(silu_f(acc[i].x), hi, lo, 0);
      EEC_SPLIT(silu_f(acc[i].y), hi, lo, 1);
      const size_t off = ((size_t)b * a.Tq + t) * kD + c;
      *(h2*)(a.o_hi + off) = hi;
      if (NP == 3) *(h2*)(a.o_lo + off) = lo;
    }
  }
}

hipError_t launch_dwconv(const DwArgs& a, int np, hipStream_t st) {
  auto k = np == 3 ? dwconv_kernel<3> : dwconv_kernel<1>;
  const int tiles = (a.Tq + 2 * kDwFrames - 1) / (2 * kDwFrames);
  hipLaunchKernelGGL(k, dim3(tiles, a.B), dim3(kThreads), 0, st, a);
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
