// Conv-module tail: depthwise Conv1d(K<=31, groups=D, 'same') + BatchNorm(eval, folded) + SiLU fused into
// the pointwise-2 GEMM + residual.  Layout everywhere: [utterance][frame][channel], channel fastest.
#include "eec_blocks.h"

namespace eec {

// ---------------------------------------------------------------------------
// Fused conv-module tail:
//     x += PW2( SiLU( BN( DW_K(g) ) ) ) + b      g = GLU output, fp16 [B*T'][256]
// One 512-thread workgroup = one row tile (64 rows at D = 256, 32 at D = 512) of the flattened (utterance, frame) axis.
//   1. stage g rows [row0-15, row0+rows+15) and (D = 256) the folded taps [31][D] in LDS;
//   2. depthwise conv + folded BatchNorm + SiLU: thread = 2 adjacent channels x 16 frames, window read
//      from LDS (lanes = consecutive channels: conflict-free); zero padding at UTTERANCE ends only
//      (reference behaviour: padded frames inside the batch tensor do leak into valid frames);
//      result split hi/lo and written straight into the A-plane layout of the GEMM - the conv
//      output never touches HBM;
//   3. pointwise-2 as the ring-pipelined MFMA GEMM (wave w -> its NW column tiles), residual add
//      straight from the accumulators.
// ---------------------------------------------------------------------------
EEC_TL_DEFINE(dw)
template <int D, int NP>
__global__ __launch_bounds__(512, 2) void dw_pw2_kernel(DwArgs d, ProjResArgs a) {
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = wave_id();
  const int row0 = row_tile_index() * G::kRows;
  const int M = a.M;
  EEC_TL_STAMP(dw, 0);
  ProjStream<NP, kDPF, G::kNW> r;
  proj_fill<D, NP, kDPF, G::kNW>(r, WMat{a.wp, a.wf8}, G::kNW * w);
  dw_front<D, NP>(smem, d, M, row0);
  EEC_TL_STAMP(dw, 3);
  __syncthreads();
  EEC_TL_STAMP(dw, 4);
  f32x16 acc2[G::kMT][G::kNW];
  pw2_gemm<D, NP>(acc2, smem, a, r);
  EEC_TL_STAMP(dw, 5);
  acc_swapped_add_rows<D, G::kMT, G::kNW>(a.x, row0, M, acc2, 32 * G::kNW * w);
  EEC_TL_STAMP(dw, 6);
}

template <int D>
static hipError_t launch_dw_pw2_d(const DwArgs& d, const ProjResArgs& a, int np, hipStream_t st) {
  auto k = np == 8 ? dw_pw2_kernel<D, 8> : np == 3 ? dw_pw2_kernel<D, 3> : dw_pw2_kernel<D, 1>;
  if (hipError_t e = ensure_max_lds((const void*)k, DwGeo<D>::kLds); e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + Geo<D>::kRows - 1) / Geo<D>::kRows), dim3(512), DwGeo<D>::kLds, st, d, a);
  return hipGetLastError();
}
hipError_t launch_dw_pw2(const DwArgs& d, const ProjResArgs& a, int np, hipStream_t st) {
  return a.D == 512 ? launch_dw_pw2_d<512>(d, a, np, st) : launch_dw_pw2_d<256>(d, a, np, st);
}

// BN(eval) folded into the depthwise taps, taps zero-padded/centred to 31:
//   y = ((sum_j w_j x + b) - rm) * s + beta,  s = gamma / sqrt(rv + 1e-5)
__global__ void fold_dw_kernel(const float* dw_w, const float* dw_b, const float* bn_w, const float* bn_b,
                               const float* bn_rm, const float* bn_rv, int ksize, int D, float* wfold, float* bfold) {
  const int c = threadIdx.x;
  const float s = bn_w[c] / sqrtf(bn_rv[c] + 1e-5f);
  const int shift = (kDwTaps - 1) / 2 - (ksize - 1) / 2;
  for (int j = 0; j < kDwTaps; ++j) {
    const int jj = j - shift;
    wfold[j * D + c] = (jj >= 0 && jj < ksize) ? dw_w[c * ksize + jj] * s : 0.f;
  }
  bfold[c] = (dw_b[c] - bn_rm[c]) * s + bn_b[c];
}

hipError_t launch_fold_dw(const float* dw_w, const float* dw_b, const float* bn_w, const float* bn_b,
                          const float* bn_rm, const float* bn_rv, int ksize, int D, float* wfold, float* bfold,
                          hipStream_t st) {
  hipLaunchKernelGGL(fold_dw_kernel, dim3(1), dim3(D), 0, st, dw_w, dw_b, bn_w, bn_b, bn_rm, bn_rv, ksize, D, wfold,
                     bfold);
  return hipGetLastError();
}

}  // namespace eec
