// Conv1dSubampling + positional encoding (reference early_exit.py:24-48,620-621):
// two Conv1d(k=3, stride=2, pad=0) back to back (no activation between), transpose to
// [utterance][frame][channel], + sinusoid PE.  Both convolutions run as MFMA GEMMs on the ring
// pipeline of eec_device.h, with hi/lo-split fp16 operands:
//   conv1: rows = (b, t1), K = n_mels*3 in the weight tensor's own [ci][j] order,
//          A[row][3ci + j] = mel[b][ci][2 t1 + j]                       -> mid (scaled, fp16 planes)
//   conv2: rows = (b, t'), K = 3*256 as three K=256 passes j = 0..2,
//          A_j[row][ci] = mid[b][2 t' + j][ci]  (129 staged frames, lane row stride 2)
// The input is un-logged power mel (large dynamic range): it is multiplied by 2^-6 (exact) on the
// way in, mid stays in that scaled domain, and the final accumulators are multiplied by 2^6.
#include "eec_kernels.h"

namespace eec {

constexpr int kStemThreads = 512;
constexpr float kMelScale = 1.0f / 64.0f;
constexpr int kSPF = 4;

// ---------------------------------------------------------------------------
// conv1.  K1 = n_mels * 3 (multiple of 16, <= 384); plane row stride (K1 + 8) halves.
// DIRECT: the one-convolution stem of Early_zipformer (Conv1dSubampling_Zipformer, early_exit.py:80-95): the result
// leaves as fp32 x[b][t1][:] = conv + bias + pe[t1] instead of the scaled fp16 planes that feed conv2.
template <int NP, bool DIRECT>
__global__ __launch_bounds__(kStemThreads, 2) void stem_conv1_kernel(SubsampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int K1 = a.n_mels * 3, ks = K1 / 16;
  const int ld = (K1 + 8) * 2, plane = kTileRows * ld;
  const int M1 = a.B * a.T1;
  const int row0 = blockIdx.x * kTileRows;

  WRing<NP, kSPF, 1> r;
  const uint4* w_lane = a.w1p + (size_t)w * ks * 128 + lane;
  ring_fill<NP, kSPF, 1>(r, w_lane, 0, ks);
  // stage A: thread = (row r, channel group); 3 taps per (row, ci)
  {
    const int rr = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int row = row0 + rr;
    const bool ok = row < M1;
    const int b = ok ? row / a.T1 : 0, t1 = ok ? row - b * a.T1 : 0;
    const float* src = a.mel + (size_t)b * a.n_mels * a.T + 2 * t1;
    for (int ci = cg; ci < a.n_mels; ci += 8) {
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (ok) {
        const float* p = src + (size_t)ci * a.T;
        v0 = p[0] * kMelScale, v1 = p[1] * kMelScale, v2 = p[2] * kMelScale;
      }
      const hl2_t s01 = split2<NP>(v0, v1), s2 = split2<NP>(v2, 0.f);
      half_t* dh = (half_t*)(smem + rr * ld) + ci * 3;
      dh[0] = s01.hi[0], dh[1] = s01.hi[1], dh[2] = s2.hi[0];
      if (NP == 3) {
        half_t* dl = (half_t*)(smem + plane + rr * ld) + ci * 3;
        dl[0] = s01.lo[0], dl[1] = s01.lo[1], dl[2] = s2.lo[0];
      }
    }
  }
  __syncthreads();
  f32x16 acc[2][1];
  {  // accumulators start at bias / 64 (the scaled domain)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *(const float4*)(a.b1 + 32 * w + 8 * g + 4 * hh);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        acc[mt][0][4 * g + 0] = bb.x * kMelScale;
        acc[mt][0][4 * g + 1] = bb.y * kMelScale;
        acc[mt][0][4 * g + 2] = bb.z * kMelScale;
        acc[mt][0][4 * g + 3] = bb.w * kMelScale;
      }
    }
  }
  const char* a_lane = smem + (lane & 31) * ld + hh * 16;
  gemm_plain_ring<NP, kSPF>(acc, a_lane, ld, plane, w_lane, ks, r);
  if constexpr (DIRECT) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = row0 + mt * 32 + (lane & 31);
      if (row < M1) {
        const int t1 = row % a.T1;
        float* dst = a.x + (size_t)row * kD + 32 * w + 4 * hh;
        const float* pe = a.pe + (size_t)t1 * kD + 32 * w + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 p = *(const float4*)(pe + 8 * g);
          *(float4*)(dst + 8 * g) = make_float4(acc[mt][0][4 * g + 0] * 64.f + p.x, acc[mt][0][4 * g + 1] * 64.f + p.y,
                                                acc[mt][0][4 * g + 2] * 64.f + p.z, acc[mt][0][4 * g + 3] * 64.f + p.w);
        }
      }
    }
    return;
  }
  // mid planes [B*T1][256] (scaled domain): lane = row, register quad = 4 consecutive channels
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = row0 + mt * 32 + (lane & 31);
    if (row < M1) {
      const size_t off = (size_t)row * kD + 32 * w + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const hl2_t s0 = split2<NP>(acc[mt][0][4 * g + 0], acc[mt][0][4 * g + 1]);
        const hl2_t s1 = split2<NP>(acc[mt][0][4 * g + 2], acc[mt][0][4 * g + 3]);
        h4 hi, lo;
        hi.xy = s0.hi, hi.zw = s1.hi, lo.xy = s0.lo, lo.zw = s1.lo;
        *(h4*)(a.mid_hi + off + 8 * g) = hi;
        if (NP == 3) *(h4*)(a.mid_lo + off + 8 * g) = lo;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// conv2 + PE.  One workgroup = 64 consecutive output frames of one utterance.
constexpr int kS2Ld = 520;                 // staged frame stride (bytes): lane row stride 1040 B is conflict-free
constexpr int kS2Rows = 2 * kTileRows + 1; // 129 staged mid frames
constexpr int kS2Plane = kS2Rows * kS2Ld;  // 67080
constexpr int kStem2Lds = 2 * ((kS2Plane + 15) / 16 * 16);

template <int NP>
__global__ __launch_bounds__(kStemThreads, 2) void stem_conv2_kernel(SubsampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int plane = kStem2Lds / 2;
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * kTileRows;
  const int ks_total = 3 * kD / 16;  // 48 k-steps, ordered (j, ci)
  WRing<NP, kSPF, 1> r;
  const uint4* w_lane = a.w2p + (size_t)w * ks_total * 128 + lane;
  ring_fill<NP, kSPF, 1>(r, w_lane, 0, kD / 16);
  // stage mid frames [2 t0, 2 t0 + 128] of utterance b: 129 rows x 32 sixteen-byte pieces per plane
  for (int p = threadIdx.x; p < kS2Rows * 32; p += kStemThreads) {
    const int fr = p >> 5, c16 = p & 31;
    const int t1 = 2 * t0 + fr;
    uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
    if (t1 < a.T1) {
      const size_t off = ((size_t)b * a.T1 + t1) * kD + c16 * 8;
      vh = *(const uint4*)(a.mid_hi + off);
      if (NP == 3) vl = *(const uint4*)(a.mid_lo + off);
    }
    *(uint4*)(smem + fr * kS2Ld + c16 * 16) = vh;
    if (NP == 3) *(uint4*)(smem + plane + fr * kS2Ld + c16 * 16) = vl;
  }
  __syncthreads();
  f32x16 acc[2][1];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bb = *(const float4*)(a.b2 + 32 * w + 8 * g + 4 * hh);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      acc[mt][0][4 * g + 0] = bb.x * kMelScale;
      acc[mt][0][4 * g + 1] = bb.y * kMelScale;
      acc[mt][0][4 * g + 2] = bb.z * kMelScale;
      acc[mt][0][4 * g + 3] = bb.w * kMelScale;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const char* a_lane = smem + (2 * (lane & 31) + j) * kS2Ld + hh * 16;
    gemm_ring<NP, kD / 16, 1, true, kSPF>(acc, a_lane, 2 * kS2Ld, plane, w_lane + (size_t)j * (kD / 16) * 128, 0, r);
    if (j < 2) ring_fill<NP, kSPF, 1>(r, w_lane + (size_t)(j + 1) * (kD / 16) * 128, 0, kD / 16);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int t = t0 + mt * 32 + (lane & 31);
    if (t < a.Tq) {
      float* dst = a.x + ((size_t)b * a.Tq + t) * kD + 32 * w + 4 * hh;
      const float* pe = a.pe + (size_t)t * kD + 32 * w + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 p = *(const float4*)(pe + 8 * g);
        *(float4*)(dst + 8 * g) = make_float4(acc[mt][0][4 * g + 0] * 64.f + p.x, acc[mt][0][4 * g + 1] * 64.f + p.y,
                                              acc[mt][0][4 * g + 2] * 64.f + p.z, acc[mt][0][4 * g + 3] * 64.f + p.w);
      }
    }
  }
}

hipError_t launch_subsample(const SubsampleArgs& a, int np, hipStream_t st) {
  const int K1 = a.n_mels * 3;
  if (K1 % 16 || K1 > 384) return hipErrorInvalidValue;
  const int lds1 = 2 * kTileRows * (K1 + 8) * 2;
  auto k1 = np == 3 ? stem_conv1_kernel<3, false> : stem_conv1_kernel<1, false>;
  auto k2 = np == 3 ? stem_conv2_kernel<3> : stem_conv2_kernel<1>;
  if (hipError_t e = ensure_max_lds((const void*)k1, 2 * kTileRows * (384 + 8) * 2); e != hipSuccess) return e;
  if (hipError_t e = ensure_max_lds((const void*)k2, kStem2Lds); e != hipSuccess) return e;
  const int M1 = a.B * a.T1;
  hipLaunchKernelGGL(k1, dim3((M1 + kTileRows - 1) / kTileRows), dim3(kStemThreads), lds1, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k2, dim3((a.Tq + kTileRows - 1) / kTileRows, a.B), dim3(kStemThreads), kStem2Lds, st, a);
  return hipGetLastError();
}

// one Conv1d(k=3, s=2) + PE: x [B][T1][256] fp32 (the Early_zipformer stem); hi/lo split operands
hipError_t launch_subsample_single(const SubsampleArgs& a, hipStream_t st) {
  const int K1 = a.n_mels * 3;
  if (K1 % 16 || K1 > 384) return hipErrorInvalidValue;
  auto k1 = stem_conv1_kernel<3, true>;
  if (hipError_t e = ensure_max_lds((const void*)k1, 2 * kTileRows * (384 + 8) * 2); e != hipSuccess) return e;
  const int M1 = a.B * a.T1;
  hipLaunchKernelGGL(k1, dim3((M1 + kTileRows - 1) / kTileRows), dim3(kStemThreads), 2 * kTileRows * (K1 + 8) * 2, st, a);
  return hipGetLastError();
}

}  // namespace eec
