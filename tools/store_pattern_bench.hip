// Microbenchmark (round 3): how fast can the training GEMM's OUTPUT alone be written?  2048 workgroups of 256 threads each store one
// 128 x 128 fp32 tile of a [16384][2048] matrix (134 MB) with 16-byte stores, in the epilogue's two orders:
//   rows    a wave-instruction covers 2 rows x 512 B (the slab epilogue: every thread 16 contiguous bytes of a row)
//   lanes   a wave-instruction covers 32 rows x 32 B (accumulator order of the operand-swapped form)
// with the GEMM's XCD-aware tile map.   hipcc -O3 --offload-arch=gfx950 tools/store_pattern_bench.hip -o /tmp/spb && /tmp/spb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 3) void store_kernel(float* __restrict__ C, int N, float v) {
  unsigned bx = blockIdx.x, by = blockIdx.y;
  const unsigned nbx = gridDim.x, total = nbx * gridDim.y;
  const unsigned l = bx + nbx * by, t = (l & 7u) * (total >> 3) + (l >> 3);
  bx = t % nbx, by = t / nbx;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float* base = C + (size_t)by * 128 * N + bx * 128;
  const f32x4 val = {v, v + 1.0f, v + 2.0f, v + (float)tid};
  if (MODE == 0) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int idx = p * 256 + tid, row = idx / 32, c4 = (idx % 32) * 4;
      *(f32x4*)(base + (size_t)row * N + c4) = val;
    }
  } else {
    const int wm = w >> 1, wn = w & 1;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)(base + (size_t)((wm * 2 + mt) * 32 + (lane & 31)) * N + (wn * 2 + nt) * 32 + 8 * q + 4 * (lane >> 5)) = val;
  }
}

int main() {
  const int M = 16384, N = 2048;
  float* C;
  CK(hipMalloc(&C, (size_t)M * N * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int round = 0; round < 2; ++round)
    for (int mode = 0; mode < 2; ++mode) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(N / 128, M / 128), dim3(256), 0, 0, C, N, 1.0f);
        else hipLaunchKernelGGL(store_kernel<1>, dim3(N / 128, M / 128), dim3(256), 0, 0, C, N, 1.0f);
      };
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 20; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-6s %7.1f us per 134 MB  %6.2f TB/s\n", mode ? "lanes" : "rows", ms * 1e3 / 20, (double)M * N * 4 * 20 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
