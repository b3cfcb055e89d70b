// Microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate per SIMD for the wave arrangements the
// FFN kernel uses.  ./mfmabw -> TFLOP/s per case (256 workgroups, operands in registers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// mode 0: every wave MFMA (NACC independent accumulators, CHAIN dependent MFMAs in a row per accumulator)
// mode 1: waves 0-3 MFMA, waves 4-7 VALU transcendental loop (silu-like)
template <int NACC, int CHAIN, int MODE, int PRIO_MFMA = 0, int VALU_ILP = 8>
__global__ __launch_bounds__(512, 2) void k(const h8* in, float* out, int iters, int mfma_waves, unsigned long long* cyc) {
  const int w = threadIdx.x >> 6;
  h8 a = in[threadIdx.x], b = in[threadIdx.x + 512];
  if (MODE == 0 || w < mfma_waves) {
    if (w >= mfma_waves) return;
    if (PRIO_MFMA) __builtin_amdgcn_s_setprio(PRIO_MFMA);
    f16v acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) out[threadIdx.x] = s;
  } else {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
    for (int it = 0; it < iters * NACC * CHAIN * 4; ++it) {  // long enough to outlast the MFMA waves
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v[j]));
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) out[threadIdx.x] = s;
  }
}

template <typename K>
void run(const char* name, K kern, int mfma_waves, int per_iter, h8* in, float* out) {
  static unsigned long long* cyc = nullptr; if (!cyc) CK(hipMalloc(&cyc, 256 * 8 * 8));
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, in, out, iters, mfma_waves, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, in, out, iters, mfma_waves, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double nm = 256.0 * mfma_waves * iters * per_iter * 5;
  static unsigned long long hc[2048]; CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
  double sc = 0; int nn = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < mfma_waves; ++w) { sc += hc[b * 8 + w]; ++nn; }
  printf("[%6.1f shader-cycles per MFMA per wave] ", sc / nn / (iters * per_iter));
  printf("%-62s %7.1f TFLOP/s   %6.1f cycles/MFMA/SIMD @2.0GHz-equivalent\n", name, nm * 32768 / (ms * 1e-3) / 1e12,
         (ms * 1e-3 / 5) * 2.0e9 / (iters * per_iter * (mfma_waves > 4 ? 2 : 1)));
}

int main() {
  h8* in; float* out;
  CK(hipMalloc(&in, 1024 * 16)); CK(hipMalloc(&out, 4096));
  _Float16 host[8192];
  for (int i = 0; i < 8192; ++i) host[i] = (_Float16)(((i * 37) % 200 - 100) / 100.0f);
  CK(hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice));
  run("8 waves (2/SIMD), 4 acc x chain 1", k<4, 1, 0>, 8, 4, in, out);
  run("8 waves (2/SIMD), 4 acc x chain 3 (NP=3 pattern)", k<4, 3, 0>, 8, 12, in, out);
  run("8 waves (2/SIMD), 2 acc x chain 3", k<2, 3, 0>, 8, 6, in, out);
  run("8 waves (2/SIMD), 1 acc x chain 8 (fully dependent)", k<1, 8, 0>, 8, 8, in, out);
  run("4 waves (1/SIMD), 4 acc x chain 1", k<4, 1, 0>, 4, 4, in, out);
  run("4 waves (1/SIMD), 1 acc x chain 8 (fully dependent)", k<1, 8, 0>, 4, 8, in, out);
  run("4 MFMA waves + 4 VALU(exp/rcp) waves, 4 acc x chain 3", k<4, 3, 1>, 4, 12, in, out);
  run("  same, MFMA waves at s_setprio 3", k<4, 3, 1, 3>, 4, 12, in, out);
  run("  same, MFMA waves at s_setprio 1", k<4, 3, 1, 1>, 4, 12, in, out);
  return 0;
}
