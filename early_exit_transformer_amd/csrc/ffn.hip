// Fused Conformer feed-forward half-step (torchaudio _FeedForwardModule
// + the 0.5*y + x residual of ConformerLayer; a8 for the optional final LayerNorm):
//
//     x <- [LN_final]( 0.5 * ( W2 . silu( W1 . LN(x) + b1 ) + b2 ) + x )
//
// One 512-thread workgroup (8 waves, two per SIMD) owns 64 rows of x; the [64, F] hidden
// activation never leaves the CU.  F is walked in chunks of 256 hidden units:
//   GEMM1 (swapped orientation: a lane owns one frame, a register quad 4 consecutive hidden
//          units) - wave w produces hidden units [32w, 32w+32) of the chunk for all 64 rows;
//          accumulators start at the bias; SiLU in the exp2 domain (log2 e folded into the
//          packed W1/b1, 1/log2 e into W2); hi/lo split; ds_write_b64 into H[frame][hidden].
//   GEMM2 (normal orientation) - wave w accumulates output columns [32w, 32w+32) over the chunk.
// Each wave streams its own, disjoint weight fragments from L2 straight into a register ring
// (1 KiB per load, lane-linear); the ring of a stage is filled during the PREVIOUS stage, so no
// stage starts with an empty pipeline.  Two waves per SIMD overlap one wave's SiLU / LDS / VMEM
// latency with the other's MFMAs.  Two workgroup barriers per chunk (H is single-buffered: the
// LDS holds the LN(x) planes, 66 KiB, and the H planes, 66 KiB).
//
// Algorithmic work: 4*D*F flop per row (2.097 MFLOP at D=256, F=2048); bound: MFMA.
// Executed MFMA work is NP x that.  HBM/L2 traffic per launch: x read+write 2 KiB/row; each
// workgroup streams all 2*D*F*2 B (x2 planes when NP=3) of weights once from L2.
#include "eec_kernels.h"

namespace eec {

constexpr int kFfnThreads = 512;
constexpr int kFC = 256;                 // hidden units per chunk
constexpr int kFfnLds = 4 * kAPlane;     // A hi/lo + H hi/lo planes, all [64][264] fp16 = 135168 B
constexpr int kPF = 4;                   // k-steps of weights kept in flight per wave

template <int NP, bool FINAL_LN>
__global__ __launch_bounds__(kFfnThreads, 2) void ffn_kernel(float* __restrict__ x, int M,
                                                             const float* __restrict__ ln_g,
                                                             const float* __restrict__ ln_b,
                                                             const uint4* __restrict__ w1p,
                                                             const float* __restrict__ b1s,
                                                             const uint4* __restrict__ w2p,
                                                             const float* __restrict__ b2, int F,
                                                             const float* __restrict__ fin_g,
                                                             const float* __restrict__ fin_b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_h = smem + 2 * kAPlane;
  const int lane = lane_id(), w = wave_id();
  const int hh = lane >> 5;
  const int row0 = blockIdx.x * kTileRows;

  rows_f32_to_planes<NP, true, 8>(lds_a, x, row0, M, ln_g, ln_b);

  const int nft = F / 32;                // 32-wide hidden tiles
  const int nchunk = (nft + 7) / 8;
  const int ks2_total = F / 16;
  const char* a_lane = lds_a + (lane & 31) * kALd + hh * 16;
  const char* h_lane = lds_h + (lane & 31) * kALd + hh * 16;

  WRing<NP, kPF> r1, r2;
  if (w < nft) ring_fill<NP, kPF>(r1, w1p + (size_t)w * (kD / 16) * 128 + lane, kD / 16);
  __syncthreads();

  f32x16 acc2[2][1];
  zero_acc(acc2);
  for (int c = 0; c < nchunk; ++c) {
    const int ft = c * 8 + w;
    const bool active1 = ft < nft;                       // wave-uniform
    const int ks2 = min(kFC / 16, ks2_total - c * (kFC / 16));
    const uint4* w2_lane = w2p + ((size_t)w * ks2_total + c * (kFC / 16)) * 128 + lane;
    ring_fill<NP, kPF>(r2, w2_lane, ks2);                // in flight during GEMM1

    f32x16 acc1[2][1];
    if (active1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bb = *(const float4*)(b1s + ft * 32 + 8 * g + 4 * hh);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc1[mt][0][4 * g + 0] = bb.x;
          acc1[mt][0][4 * g + 1] = bb.y;
          acc1[mt][0][4 * g + 2] = bb.z;
          acc1[mt][0][4 * g + 3] = bb.w;
        }
      }
      gemm_ring<NP, kD / 16, true, kPF>(acc1, a_lane, kALd, kAPlane, w1p + (size_t)ft * (kD / 16) * 128 + lane, r1);
    }
    __syncthreads();  // every wave has finished reading H of the previous chunk
    if (active1) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        char* dst = lds_h + (mt * 32 + (lane & 31)) * kALd + (w * 32 + 4 * hh) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const hl2_t s0 = split2<NP>(silu_exp2(acc1[mt][0][4 * g + 0]), silu_exp2(acc1[mt][0][4 * g + 1]));
          const hl2_t s1 = split2<NP>(silu_exp2(acc1[mt][0][4 * g + 2]), silu_exp2(acc1[mt][0][4 * g + 3]));
          h4 hi, lo;
          hi.xy = s0.hi, hi.zw = s1.hi, lo.xy = s0.lo, lo.zw = s1.lo;
          *(h4*)(dst + g * 16) = hi;
          if (NP == 3) *(h4*)(dst + kAPlane + g * 16) = lo;
        }
      }
    }
    __syncthreads();  // H of this chunk is complete
    if (ft + 8 < nft) ring_fill<NP, kPF>(r1, w1p + (size_t)(ft + 8) * (kD / 16) * 128 + lane, kD / 16);
    if (ks2 == kFC / 16)
      gemm_ring<NP, kFC / 16, false, kPF>(acc2, h_lane, kALd, kAPlane, w2_lane, r2);
    else
      gemm_plain<NP, false>(acc2, h_lane, kALd, kAPlane, w2_lane, ks2);
  }
  __syncthreads();  // all waves are done with the A planes (the fp32 tile below aliases them)
  acc_to_etile<1>(smem, acc2, w * 32, b2);
  __syncthreads();

  float4 g = make_float4(1.f, 1.f, 1.f, 1.f), bt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FINAL_LN) {
    g = ((const float4*)fin_g)[lane];
    bt = ((const float4*)fin_b)[lane];
  }
#pragma unroll 4
  for (int i = 0; i < 8; ++i) {
    const int rl = w * 8 + i, row = row0 + rl;
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
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int grid = (a.M + kTileRows - 1) / kTileRows;
  hipLaunchKernelGGL(k, dim3(grid), dim3(kFfnThreads), kFfnLds, st, a.x, a.M, a.ln_g, a.ln_b, a.w1p, a.b1, a.w2p,
                     a.b2, a.F, a.fin_g, a.fin_b);
  return hipGetLastError();
}

hipError_t launch_ffn(const FfnArgs& a, int np, hipStream_t st) {
  const bool fl = a.fin_g != nullptr;
  if (np == 3) return fl ? launch_ffn_t<3, true>(a, st) : launch_ffn_t<3, false>(a, st);
  return fl ? launch_ffn_t<1, true>(a, st) : launch_ffn_t<1, false>(a, st);
}

}  // namespace eec
