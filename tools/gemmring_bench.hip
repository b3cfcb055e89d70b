// Microbenchmark of eec::gemm_ring in isolation: shader cycles per k-step for the FFN consumer
// (NT=2, normal) and producer (NT=1, swapped) configurations, 1 or 2 waves per SIMD.
#include "../early_exit_transformer_amd/csrc/eec_device.h"
#include <stdio.h>
#include <stdlib.h>
using namespace eec;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NP, int KS, int NT, bool SWAP, int PF>
__global__ __launch_bounds__(512, 2) void kern(const uint4* w, float* out, unsigned long long* cyc, int reps, int nwaves) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 2 * kAPlane / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv >= nwaves) return;
  const char* a_lane = smem + (lane & 31) * kALd + (lane >> 5) * 16;
  const size_t nt_stride = (size_t)KS * 128;
  const uint4* w_lane = w + (size_t)wv * NT * nt_stride + lane;
  f32x16 acc[2][NT];
  zero_acc(acc);
  WRing<NP, PF, NT> r;
  ring_fill<NP, PF, NT>(r, w_lane, nt_stride, KS);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; ++it) {
    gemm_ring<NP, KS, NT, SWAP, PF>(acc, a_lane, kALd, kAPlane, w_lane, nt_stride, r);
    ring_fill<NP, PF, NT>(r, w_lane, nt_stride, KS);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
  float s = 0;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < NT; ++b) s += acc[a][b][3];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

template <typename K>
void run(const char* nm, K k, int ks, int mf_per_step, int nwaves, uint4* w, float* out, unsigned long long* cyc) {
  const int reps = 200;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kAPlane));
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 2 * kAPlane, 0, w, out, cyc, reps, nwaves);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 2 * kAPlane, 0, w, out, cyc, reps, nwaves);
  CK(hipDeviceSynchronize());
  static unsigned long long h[2048];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double s = 0; int n = 0;
  for (int b = 0; b < 256; ++b) for (int v = 0; v < nwaves; ++v) { s += h[b * 8 + v]; ++n; }
  const double per_step = s / n / reps / ks;
  printf("%-58s %7.1f cycles/k-step  (%d MFMAs -> ideal %d alone, %d shared)  %5.1f cycles/MFMA\n", nm, per_step, mf_per_step,
         mf_per_step * 32, mf_per_step * 64, per_step / mf_per_step);
}

int main() {
  uint4* w; float* out; unsigned long long* cyc;
  CK(hipMalloc(&w, 64 << 20)); CK(hipMemset(w, 0x11, 64 << 20));
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 2048 * 8));
  run("consumer NP3 KS8 NT2 normal PF4, 4 waves (1/SIMD)", kern<3, 8, 2, false, 4>, 8, 12, 4, w, out, cyc);
  run("consumer NP3 KS8 NT2 normal PF4, 8 waves (2/SIMD)", kern<3, 8, 2, false, 4>, 8, 12, 8, w, out, cyc);
  run("producer NP3 KS16 NT1 swapped PF8, 4 waves (1/SIMD)", kern<3, 16, 1, true, 8>, 16, 6, 4, w, out, cyc);
  run("producer NP3 KS16 NT1 swapped PF8, 8 waves (2/SIMD)", kern<3, 16, 1, true, 8>, 16, 6, 8, w, out, cyc);
  run("consumer NP1 KS8 NT2 normal PF8, 4 waves (1/SIMD)", kern<1, 8, 2, false, 8>, 8, 4, 4, w, out, cyc);
  run("producer NP1 KS16 NT1 swapped PF12, 4 waves (1/SIMD)", kern<1, 16, 1, true, 12>, 16, 2, 4, w, out, cyc);
  return 0;
}
