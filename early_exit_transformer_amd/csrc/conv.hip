// Convolution-side kernels of the path (all HBM/VALU-bound, fp32 arithmetic):
//   dwconv_kernel      depthwise Conv1d(K<=31, groups=D, 'same') + BatchNorm(eval, folded) + SiLU   
//   subsample1/2       Conv1dSubampling: two Conv1d(k=3, s=2, p=0), then + sinusoid PE               (a1, a2)
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

// ---------------------------------------------------------------------------
// Subsampling, fp32 FMA (the input is un-logged power mel: large dynamic range).
// conv1: mid[b][t1][c] = b1[c] + sum_{ci,j} w1[c][ci][j] * mel[b][ci][2 t1 + j]
// Block = 32 output frames of one utterance; thread = output channel; the mel tile
// [n_mels][68] is shared through LDS and read as broadcast 16-byte pieces.
// ---------------------------------------------------------------------------
constexpr int kS1Frames = 32;
constexpr int kS1Ld = 68;  // 2*32+1 = 65 inputs, padded to a multiple of 4

__global__ __launch_bounds__(kThreads) void subsample1_kernel(SubsampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = (float*)smem;
  const int b = blockIdx.y, t10 = blockIdx.x * kS1Frames, c = threadIdx.x;
  const int in0 = 2 * t10;
  for (int p = threadIdx.x; p < a.n_mels * kS1Ld; p += kThreads) {
    const int ci = p / kS1Ld, j = p - ci * kS1Ld;
    const int t = in0 + j;
    tile[p] = (t < a.T) ? a.mel[((size_t)b * a.n_mels + ci) * a.T + t] : 0.f;
  }
  __syncthreads();
  float acc[kS1Frames];
  const float bias = a.b1[c];
#pragma unroll
  for (int i = 0; i < kS1Frames; ++i) acc[i] = bias;
  for (int ci = 0; ci < a.n_mels; ++ci) {
    float v[kS1Ld];
#pragma unroll
    for (int q = 0; q < kS1Ld / 4; ++q) {
      const float4 t4 = *(const float4*)(tile + ci * kS1Ld + q * 4);
      v[4 * q] = t4.x;
      v[4 * q + 1] = t4.y;
      v[4 * q + 2] = t4.z;
      v[4 * q + 3] = t4.w;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float wv = a.w1t[(ci * 3 + j) * kD + c];
#pragma unroll
      for (int i = 0; i < kS1Frames; ++i) acc[i] = fmaf(wv, v[2 * i + j], acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < kS1Frames; ++i) {
    const int t1 = t10 + i;
    if (t1 < a.T1) a.mid[((size_t)b * a.T1 + t1) * kD + c] = acc[i];
  }
}

// conv2 + PE: x[b*Tq + t][c] = b2[c] + pe[t][c] + sum_{ci,j} w2[c][ci][j] * mid[b][2t + j][ci]
constexpr int kS2Frames = 16;
constexpr int kS2Ld = 36;  // 33 inputs padded to a multiple of 4

__global__ __launch_bounds__(kThreads) void subsample2_kernel(SubsampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = (float*)smem;  // [256 ci][36]
  const int b = blockIdx.y, t0 = blockIdx.x * kS2Frames, c = threadIdx.x;
  for (int j = 0; j < 2 * kS2Frames + 1; ++j) {
    const int t1 = 2 * t0 + j;
    tile[c * kS2Ld + j] = (t1 < a.T1) ? a.mid[((size_t)b * a.T1 + t1) * kD + c] : 0.f;
  }
  __syncthreads();
  float acc[kS2Frames];
  const float bias = a.b2[c];
#pragma unroll
  for (int i = 0; i < kS2Frames; ++i) acc[i] = bias;
  for (int ci = 0; ci < kD; ++ci) {
    float v[kS2Ld];
#pragma unroll
    for (int q = 0; q < kS2Ld / 4; ++q) {
      const float4 t4 = *(const float4*)(tile + ci * kS2Ld + q * 4);
      v[4 * q] = t4.x;
      v[4 * q + 1] = t4.y;
      v[4 * q + 2] = t4.z;
      v[4 * q + 3] = t4.w;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float wv = a.w2t[(ci * 3 + j) * kD + c];
#pragma unroll
      for (int i = 0; i < kS2Frames; ++i) acc[i] = fmaf(wv, v[2 * i + j], acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < kS2Frames; ++i) {
    const int t = t0 + i;
    if (t < a.Tq) a.x[((size_t)b * a.Tq + t) * kD + c] = acc[i] + a.pe[(size_t)t * kD + c];
  }
}

hipError_t launch_subsample(const SubsampleArgs& a, hipStream_t st) {
  const int lds1 = a.n_mels * kS1Ld * 4;
  hipLaunchKernelGGL(subsample1_kernel, dim3((a.T1 + kS1Frames - 1) / kS1Frames, a.B), dim3(kThreads), lds1, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(subsample2_kernel, dim3((a.Tq + kS2Frames - 1) / kS2Frames, a.B), dim3(kThreads),
                     kD * kS2Ld * 4, st, a);
  return hipGetLastError();
}

// [co][ci][j] -> [(ci*ks + j)][co]
__global__ void transpose_conv_kernel(const float* w, int cout, int cin, int ks, float* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cout * cin * ks) return;
  const int co = idx / (cin * ks), rem = idx - co * cin * ks;
  out[(size_t)rem * cout + co] = w[idx];
}

hipError_t launch_transpose_conv(const float* w, int cout, int cin, int ks, float* out, hipStream_t st) {
  const int n = cout * cin * ks;
  hipLaunchKernelGGL(transpose_conv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, cout, cin, ks, out);
  return hipGetLastError();
}

}  // namespace eec
