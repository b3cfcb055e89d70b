// Which fp16 MFMA shape does the chip run faster on RANDOM data when the clock, not the issue rate, is the limit?
// (MI355X_MICROARCH.md, DVFS give-back item 7: the 16x16x32 shape held ~1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP.)
// Same output tile per wave (64 x 64 fp32 = 64 accumulator registers), operands in registers, 1 or 2 waves per SIMD, every CU busy;
// reports wall TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_bench.hip -o tools/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_stamps[4];

template <int SHAPE>  // 0: 32x32x16 (4 tiles of 16 regs), 1: 16x16x32 (16 tiles of 4 regs)
__global__ __launch_bounds__(512, 2) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  // operand fragments: 4 A + 4 B register quads of random fp16 (|x| < 1), fixed for the whole loop
  h8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(h8, src[(i * 1024 + threadIdx.x) & 8191]);
    b[i] = __builtin_bit_cast(h8, src[((i + 4) * 1024 + threadIdx.x) & 8191]);
  }
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0) t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)  // K = 32 per iteration: two 16-deep steps
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ks + i], b[2 * ks + j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += acc[i][j][k];
  } else {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)  // K = 32 per iteration: one 32-deep step
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    g_stamps[0] = t0, g_stamps[1] = r0, g_stamps[2] = __builtin_amdgcn_s_memtime(), g_stamps[3] = __builtin_amdgcn_s_memrealtime();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main() {
  uint4* src;
  float* out;
  CK(hipMalloc(&src, 8192 * 16));
  CK(hipMalloc(&out, 256 * 512 * 4 * 2));
  {
    unsigned short* h = (unsigned short*)malloc(8192 * 16);
    unsigned long long s = 12345;
    for (int i = 0; i < 8192 * 8; ++i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned r = (unsigned)(s >> 33);
      h[i] = (unsigned short)((r & 0x8000u) | (0x3000u + (r & 0x0bffu)));  // random sign, |x| in [2^-3, 1): full mantissa toggling
    }
    CK(hipMemcpy(src, h, 8192 * 16, hipMemcpyHostToDevice));
    free(h);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 40000;  // x 64 x 64 x 32 x 2 flop per wave
  for (int wpb = 256; wpb <= 512; wpb += 256)
    for (int shape = 0; shape < 2; ++shape) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(256), dim3(wpb), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(mfma_loop<1>, dim3(256), dim3(wpb), 0, 0, src, out, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[4];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
        const double flop = 2.0 * 64 * 64 * 32 * (double)iters * (wpb / 64) * 256;
        if (rep == 2)
          printf("%s, %d waves/SIMD: %.3f ms, %.0f TFLOP/s, in-kernel clock %.2f GHz, %.1f cycles per 64x64x32 step\n",
                 shape == 0 ? "32x32x16" : "16x16x32", wpb / 256, ms, flop / (ms * 1e-3) / 1e12,
                 (double)(st[2] - st[0]) / ((double)(st[3] - st[1]) * 10.0), (double)(st[2] - st[0]) / iters);
      }
    }
  return 0;
}
