// K=256 projection kernels of a Conformer layer, one 64-row tile per workgroup:
//   qkv_kernel          LN -> in_proj (N=768) -> Q (pre-scaled), K, V^T in attention layouts   
//   proj_residual_kernel x += A . W^T + b   (attention out_proj a6; conv pointwise-2 a7)
//   pw1_glu_kernel      LN -> pointwise-1 (N=512) -> GLU -> fp16                                  (a7)
//   head_kernel         exit head: Linear(D,V) -> log_softmax -> fp32 log-probs                    (a9)
// All share: activation planes in LDS (eec_device.h), weights streamed as packed
// fragments, results staged through an fp32 [64][256] LDS tile so that global
// stores are whole rows.
#include "eec_kernels.h"

namespace eec {

constexpr int kLinLds = 2 * kAPlane + kETile;  // 134144

__device__ __forceinline__ int vt_perm(int t) {  // swap bits 2 and 3: MFMA k-order of an accumulator-fed operand
  return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
}

// ---------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(kThreads, 1) void qkv_kernel(QkvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_e = smem + 2 * kAPlane;
  const int lane = lane_id(), w = wave_id();
  const int row0 = blockIdx.x * kTileRows;
  const int dh = kD / a.H;
  rows_f32_to_planes<NP, true>(lds_a, a.x, row0, a.M, a.ln_g, a.ln_b);
  __syncthreads();

  // Q (blk 0) and K (blk 1): normal orientation -> E tile -> row-wise 8-byte stores
  for (int blk = 0; blk < 2; ++blk) {
    f32x16 acc[2][2];
    zero_acc(acc);
    gemm_stage<NP, kD / 16, 2, 2, false>(acc, lds_a, kALd, kAPlane, a.wp, kD / 16, blk * 8 + 2 * w, 0);
    if (blk) __syncthreads();  // previous row pass finished reading the E tile
    acc_to_etile<2>(lds_e, acc, w * 64, a.bias + blk * kD);
    __syncthreads();
    half_t* dst = blk == 0 ? a.q : a.k;
    const float scale = blk == 0 ? kLog2e * rsqrtf((float)dh) : 1.0f;
    const int col = lane * 4, hd = col / dh, d = col % dh;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int rl = w * 16 + i, row = row0 + rl;
      if (row >= a.M) break;
      const int b = row / a.Tq, t = row - b * a.Tq;
      const float4 e = *(const float4*)(lds_e + rl * kELd + lane * 16);
      h4 o;
      o[0] = to_half_sat(e.x * scale);
      o[1] = to_half_sat(e.y * scale);
      o[2] = to_half_sat(e.z * scale);
      o[3] = to_half_sat(e.w * scale);
      *(h4*)(dst + ((size_t)(b * a.H + hd) * a.Tp + t) * dh + d) = o;
    }
  }
  // V: swapped orientation (frames on lanes) -> V^T[b][h][d][perm(t)], 2-byte stores contiguous along t
  {
    f32x16 acc[2][2];
    zero_acc(acc);
    gemm_stage<NP, kD / 16, 2, 2, true>(acc, lds_a, kALd, kAPlane, a.wp, kD / 16, 16 + 2 * w, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = row0 + mt * 32 + (lane & 31);
      if (row < a.M) {
        const int b = row / a.Tq, t = row - b * a.Tq;
        const int tp = vt_perm(t);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int n = w * 64 + nt * 32 + acc_row(i, lane);
            const int hd = n / dh, d = n - hd * dh;
            a.vt[((size_t)(b * a.H + hd) * dh + d) * a.Tp + tp] =
                to_half_sat(acc[mt][nt][i] + a.bias[2 * kD + n]);
          }
      }
    }
  }
}

hipError_t launch_qkv(const QkvArgs& a, int np, hipStream_t st) {
  auto k = np == 3 ? qkv_kernel<3> : qkv_kernel<1>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows), dim3(kThreads), kLinLds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(kThreads, 1) void proj_residual_kernel(ProjResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_e = smem + 2 * kAPlane;
  const int lane = lane_id(), w = wave_id();
  const int row0 = blockIdx.x * kTileRows;
  rows_planes_to_lds<NP>(lds_a, a.a_hi, a.a_lo, row0, a.M);
  __syncthreads();
  f32x16 acc[2][2];
  zero_acc(acc);
  gemm_stage<NP, kD / 16, 2, 2, false>(acc, lds_a, kALd, kAPlane, a.wp, kD / 16, 2 * w, 0);
  acc_to_etile<2>(lds_e, acc, w * 64, a.bias);
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int rl = w * 16 + i, row = row0 + rl;
    if (row >= a.M) break;
    const float4 e = *(const float4*)(lds_e + rl * kELd + lane * 16);
    float4 v = ((const float4*)(a.x + (size_t)row * kD))[lane];
    v.x += e.x;
    v.y += e.y;
    v.z += e.z;
    v.w += e.w;
    ((float4*)(a.x + (size_t)row * kD))[lane] = v;
  }
}

hipError_t launch_proj_residual(const ProjResArgs& a, int np, hipStream_t st) {
  auto k = np == 3 ? proj_residual_kernel<3> : proj_residual_kernel<1>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows), dim3(kThreads), kLinLds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(kThreads, 1) void pw1_glu_kernel(GluArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_e = smem + 2 * kAPlane;
  const int lane = lane_id(), w = wave_id();
  const int row0 = blockIdx.x * kTileRows;
  rows_f32_to_planes<NP, true>(lds_a, a.x, row0, a.M, a.ln_g, a.ln_b);
  __syncthreads();
  f32x16 av[2][2], ag[2][2];
  zero_acc(av);
  zero_acc(ag);
  gemm_stage<NP, kD / 16, 2, 2, false>(av, lds_a, kALd, kAPlane, a.wp, kD / 16, 2 * w, 0);
  gemm_stage<NP, kD / 16, 2, 2, false>(ag, lds_a, kALd, kAPlane, a.wp, kD / 16, 8 + 2 * w, 0);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = w * 64 + nt * 32 + (lane & 31);
    const float bv = a.bias[col], bg = a.bias[kD + col];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = mt * 32 + acc_row(i, lane);
        *(float*)(lds_e + r * kELd + col * 4) = (av[mt][nt][i] + bv) * sigmoid_f(ag[mt][nt][i] + bg);
      }
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int rl = w * 16 + i, row = row0 + rl;
    if (row >= a.M) break;
    const float4 e = *(const float4*)(lds_e + rl * kELd + lane * 16);
    h4 o;
    o[0] = to_half_sat(e.x);
    o[1] = to_half_sat(e.y);
    o[2] = to_half_sat(e.z);
    o[3] = to_half_sat(e.w);
    *(h4*)(a.g + (size_t)row * kD + lane * 4) = o;
  }
}

hipError_t launch_pw1_glu(const GluArgs& a, int np, hipStream_t st) {
  auto k = np == 3 ? pw1_glu_kernel<3> : pw1_glu_kernel<1>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows), dim3(kThreads), kLinLds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Exit head.  V <= 256, V % 32 == 0; columns >= V are padding (-inf before the softmax).
template <int NP>
__global__ __launch_bounds__(kThreads, 1) void head_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_e = smem + 2 * kAPlane;
  const int lane = lane_id(), w = wave_id();
  const int row0 = blockIdx.x * kTileRows;
  rows_f32_to_planes<NP, false>(lds_a, a.x, row0, a.M, nullptr, nullptr);
  __syncthreads();
  const int ntiles = a.V / 32;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nt = 2 * w + j;  // wave-uniform
    if (nt < ntiles) {
      f32x16 acc[2][1];
      zero_acc(acc);
      gemm_stage<NP, kD / 16, 2, 1, false>(acc, lds_a, kALd, kAPlane, a.wp, kD / 16, nt, 0);
      const int col = nt * 32 + (lane & 31);
      const float b = a.bias[col];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          *(float*)(lds_e + (mt * 32 + acc_row(i, lane)) * kELd + col * 4) = acc[mt][0][i] + b;
    }
  }
  __syncthreads();
  const bool have = lane * 4 < a.V;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int rl = w * 16 + i, row = row0 + rl;
    if (row >= a.M) break;
    float4 e = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (have) e = *(const float4*)(lds_e + rl * kELd + lane * 16);
    const float mx = wave_max(fmaxf(fmaxf(e.x, e.y), fmaxf(e.z, e.w)));
    float s = 0.f;
    if (have) s = __expf(e.x - mx) + __expf(e.y - mx) + __expf(e.z - mx) + __expf(e.w - mx);
    const float lse = mx + __logf(wave_sum(s));
    if (have) {
      e.x -= lse;
      e.y -= lse;
      e.z -= lse;
      e.w -= lse;
      *(float4*)(a.out + (size_t)row * a.V + lane * 4) = e;
    }
  }
}

hipError_t launch_head(const HeadArgs& a, int np, hipStream_t st) {
  auto k = np == 3 ? head_kernel<3> : head_kernel<1>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLinLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows), dim3(kThreads), kLinLds, st, a);
  return hipGetLastError();
}

}  // namespace eec
