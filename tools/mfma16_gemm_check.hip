// Device check of the 16x16x32 GEMM forms (eec_device.h, EEC_MFMA16) against the 32x32x16 forms on the same LDS planes, packed
// weights and rings: both orientations, one / two row tiles, one / two column tiles, ring refills, and the layout conversions.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I early_exit_transformer_amd/csrc tools/mfma16_gemm_check.hip -o tools/mfma16_gemm_check
#include <stdio.h>
#include <stdlib.h>
#include "eec_device.h"
using namespace eec;
template <bool SW, int MT, int NT, int KS, int PF>
__global__ void k(const half_t* act_hi, const half_t* act_lo, const uint4* wp, float* out32, float* out16) {
  constexpr int K = KS * 16, LD = (K + 8) * 2, PLANE = 64 * LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 64 * K; i += 64) {
    const int r = i / K, c = i % K;
    *(half_t*)(smem + r * LD + c * 2) = act_hi[i];
    *(half_t*)(smem + PLANE + r * LD + c * 2) = act_lo[i];
  }
  __syncthreads();
  const int lane = threadIdx.x, hh = lane >> 5;
  const char* a_lane = smem + (lane & 31) * LD + hh * 16;
  f32x16 a32[MT][NT], a16[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) a32[mt][nt][i] = a16[mt][nt][i] = 0.01f * i + mt + 3 * nt;
  WRing<3, PF, NT> r;
  const size_t nts = (size_t)KS * 128;
  ring_fill_32<3, PF, NT>(r, wp + lane, nts, KS);
  gemm_ring_32<3, KS, NT, SW, PF, NoSide, 0, MT>(a32, a_lane, LD, PLANE, wp + lane, nts, r);
  ring_fill_16<3, PF, NT>(r, wp + lane, nts, KS);
  gemm_ring_16<3, KS, NT, SW, PF, NoSide, 0, MT, true, true>(a16, a_lane, LD, PLANE, wp + lane, nts, r);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        out32[((mt * NT + nt) * 16 + i) * 64 + lane] = a32[mt][nt][i];
        out16[((mt * NT + nt) * 16 + i) * 64 + lane] = a16[mt][nt][i];
      }
}
static half_t *ah, *al; static uint4* wp; static float *o32, *o16;
template <bool SW, int MT, int NT, int KS, int PF>
static int run(const char* name) {
  hipLaunchKernelGGL((k<SW, MT, NT, KS, PF>), dim3(1), dim3(64), 2 * 64 * (KS * 16 + 8) * 2, 0, ah, al, wp, o32, o16);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return 1; }
  double worst = 0; int bad = 0;
  for (int i = 0; i < MT * NT * 16 * 64; ++i) { const double d = fabs(o32[i] - o16[i]); if (d > worst) worst = d; if (d > 1e-4) ++bad; }
  printf("%-36s max |32x32x16 - 16x16x32| = %.3e, %d of %d differ  %s\n", name, worst, bad, MT * NT * 1024, bad ? "FAIL" : "OK");
  return bad != 0;
}
int main() {
  const int KMAX = 512;
  (void)hipMallocManaged(&ah, 64 * KMAX * 2); (void)hipMallocManaged(&al, 64 * KMAX * 2); (void)hipMallocManaged(&wp, 2 * (KMAX / 16) * 2 * 64 * 16);
  (void)hipMallocManaged(&o32, 4 * 16 * 64 * 4); (void)hipMallocManaged(&o16, 4 * 16 * 64 * 4);
  srand(1);
  for (int i = 0; i < 64 * KMAX; ++i) { ah[i] = (half_t)((rand() % 200 - 100) / 64.0f); al[i] = (half_t)((rand() % 200 - 100) / 65536.0f); }
  half_t* w = (half_t*)wp;
  for (int i = 0; i < 2 * (KMAX / 16) * 2 * 64 * 8; ++i) w[i] = (half_t)((rand() % 200 - 100) / 128.0f);
  int fails = 0;
  fails += run<false, 2, 1, 4, 4>("normal  MT2 NT1 KS4  PF4");
  fails += run<true, 2, 1, 4, 4>("swapped MT2 NT1 KS4  PF4");
  fails += run<true, 1, 1, 4, 4>("swapped MT1 NT1 KS4  PF4");
  fails += run<true, 1, 1, 16, 4>("swapped MT1 NT1 KS16 PF4");
  fails += run<true, 2, 1, 16, 8>("swapped MT2 NT1 KS16 PF8");
  fails += run<true, 1, 2, 32, 4>("swapped MT1 NT2 KS32 PF4");
  fails += run<false, 2, 2, 8, 4>("normal  MT2 NT2 KS8  PF4");
  fails += run<false, 1, 2, 8, 2>("normal  MT1 NT2 KS8  PF2");
  return fails;
}
