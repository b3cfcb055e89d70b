// Fused Conformer feed-forward half-step (torchaudio _FeedForwardModule
// + the 0.5*y + x residual of ConformerLayer; a8 for the optional final LayerNorm):
//
//     x <- [LN_final]( 0.5 * ( W2 . silu( W1 . LN(x) + b1 ) + b2 ) + x )
//
// One workgroup = 64 rows of x; the [64, F] hidden activation never leaves the CU:
// F is walked in chunks of 128 columns, each chunk is produced by GEMM1 (swapped
// orientation, so a lane owns one frame and 4 consecutive hidden features per
// register quad), SiLU'd, written to LDS as the next A operand (ds_write_b64,
// row-major [frame][hidden]) and consumed by GEMM2 into the [64, 256] output
// accumulators that stay in registers for all chunks.  Weights stream from L2/HBM
// as pre-packed 1-KiB fragments straight into registers (each wave owns distinct
// weight rows, so there is nothing to share through LDS).
//
// Algorithmic work: 2*2*D*F flop per row = 2.097 MFLOP (D=256, F=2048);
// bound: MFMA.  HBM traffic per launch: x read+write 2*M*1 KiB, weights 2 MiB
// fp16 per plane (L2-resident after the first workgroups).
#include "eec_kernels.h"

namespace eec {

constexpr int kFC = 128;                          // hidden columns per chunk
constexpr int kHLd = (kFC + 8) * 2;               // 272
constexpr int kHPlane = kTileRows * kHLd;         // 17408
constexpr int kFfnLds = 2 * kAPlane + 4 * kHPlane;  // 137216

template <int NP, bool FINAL_LN>
__global__ __launch_bounds__(kThreads, 1) void ffn_kernel(float* __restrict__ x, int M,
                                                          const float* __restrict__ ln_g,
                                                          const float* __restrict__ ln_b,
                                                          const uint4* __restrict__ w1p,
                                                          const float* __restrict__ b1,
                                                          const uint4* __restrict__ w2p,
                                                          const float* __restrict__ b2, int F,
                                                          const float* __restrict__ fin_g,
                                                          const float* __restrict__ fin_b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_h = smem + 2 * kAPlane;
  const int lane = lane_id(), w = wave_id();
  const int row0 = blockIdx.x * kTileRows;

  rows_f32_to_planes<NP, true>(lds_a, x, row0, M, ln_g, ln_b);
  __syncthreads();

  f32x16 acc2[2][2];
  zero_acc(acc2);
  const int nchunk = F / kFC;
  const int ks2_total = F / 16;
  for (int c = 0; c < nchunk; ++c) {
    char* hb = lds_h + (c & 1) * 2 * kHPlane;
    f32x16 acc1[2][1];
    zero_acc(acc1);
    gemm_stage<NP, kD / 16, 2, 1, true>(acc1, lds_a, kALd, kAPlane, w1p, kD / 16, c * 4 + w, 0);
    // bias + SiLU + split -> H[frame][hidden] (lane = frame, register quad = 4 consecutive hidden)
    const int fl0 = w * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *(const float4*)(b1 + c * kFC + fl0 + 8 * g);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = mt * 32 + (lane & 31);
        h4 hi, lo;
        EEC_SPLIT(silu_f(acc1[mt][0][4 * g + 0] + bb.x), hi, lo, 0);
        EEC_SPLIT(silu_f(acc1[mt][0][4 * g + 1] + bb.y), hi, lo, 1);
        EEC_SPLIT(silu_f(acc1[mt][0][4 * g + 2] + bb.z), hi, lo, 2);
        EEC_SPLIT(silu_f(acc1[mt][0][4 * g + 3] + bb.w), hi, lo, 3);
        *(h4*)(hb + m * kHLd + (fl0 + 8 * g) * 2) = hi;
        if (NP == 3) *(h4*)(hb + kHPlane + m * kHLd + (fl0 + 8 * g) * 2) = lo;
      }
    }
    __syncthreads();
    gemm_stage<NP, kFC / 16, 2, 2, false>(acc2, hb, kHLd, kHPlane, w2p, ks2_total, 2 * w, c * (kFC / 16));
  }
  __syncthreads();  // every wave is done with the A planes and H buffers
  acc_to_etile<2>(smem, acc2, w * 64, b2);
  __syncthreads();

  float4 g = make_float4(1.f, 1.f, 1.f, 1.f), bt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FINAL_LN) {
    g = ((const float4*)fin_g)[lane];
    bt = ((const float4*)fin_b)[lane];
  }
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int rl = w * 16 + i, row = row0 + rl;
    if (row >= M) break;  // wave-uniform
    const float4 e = *(const float4*)(smem + rl * kELd + lane * 16);
    float4 v = ((const float4*)(x + (size_t)row * kD))[lane];
    v.x += 0.5f * e.x;
    v.y += 0.5f * e.y;
    v.z += 0.5f * e.z;
    v.w += 0.5f * e.w;
    if (FINAL_LN) {
      const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / kD);
      const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
      const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / kD);
      const float rs = rsqrtf(var + kLnEps);
      v.x = dx * rs * g.x + bt.x;
      v.y = dy * rs * g.y + bt.y;
      v.z = dz * rs * g.z + bt.z;
      v.w = dw * rs * g.w + bt.w;
    }
    ((float4*)(x + (size_t)row * kD))[lane] = v;
  }
}

template <int NP, bool FL>
static hipError_t launch_ffn_t(const FfnArgs& a, hipStream_t st) {
  auto k = ffn_kernel<NP, FL>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLds);
  if (e != hipSuccess) return e;
  const int grid = (a.M + kTileRows - 1) / kTileRows;
  hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), kFfnLds, st, a.x, a.M, a.ln_g, a.ln_b, a.w1p, a.b1, a.w2p,
                     a.b2, a.F, a.fin_g, a.fin_b);
  return hipGetLastError();
}

hipError_t launch_ffn(const FfnArgs& a, int np, hipStream_t st) {
  const bool fl = a.fin_g != nullptr;
  if (np == 3) return fl ? launch_ffn_t<3, true>(a, st) : launch_ffn_t<3, false>(a, st);
  return fl ? launch_ffn_t<1, true>(a, st) : launch_ffn_t<1, false>(a, st);
}

}  // namespace eec
