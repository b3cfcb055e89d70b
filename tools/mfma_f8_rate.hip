// Microbenchmark: issue-to-issue cycles of v_mfma_scale_f32_32x32x64_f8f6f4 per operand format (bf8 / fp6 / fp4),
// against v_mfma_f32_32x32x16_f16 (32 cycles).  One wave per SIMD, 4 independent accumulators.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f8_rate.hip -o tools/mfma_f8_rate && ./tools/mfma_f8_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int FMT, int FMTB = FMT>  // 0: fp8 e4m3, 1: bf8 e5m2, 2: fp6 e2m3, 4: fp4 e2m1, -1: fp16 32x32x16
__global__ __launch_bounds__(256) void k(const int* in, float* out, int iters, unsigned long long* cyc) {
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x * 8 + i], b[i] = in[2048 + threadIdx.x * 8 + i];
  h8 ha = __builtin_bit_cast(h8, (int __attribute__((ext_vector_type(4)))){a[0], a[1], a[2], a[3]});
  h8 hb = __builtin_bit_cast(h8, (int __attribute__((ext_vector_type(4)))){b[0], b[1], b[2], b[3]});
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (FMT < 0)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i], 0, 0, 0);
      else
        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], FMT, FMTB, 0, 127, 0, 127);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][9];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int FMT, int FMTB = FMT>
void run(const char* name, int* in, float* out, unsigned long long* cyc) {
  const int iters = 4000;
  hipLaunchKernelGGL((k<FMT, FMTB>), dim3(256), dim3(256), 0, 0, in, out, iters, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<FMT, FMTB>), dim3(256), dim3(256), 0, 0, in, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
  const double flop = (FMT < 0 ? 32768.0 : 131072.0) * 256 * 4 * iters * 4.0;
  printf("%-28s %6.1f shader cycles per MFMA per wave   %8.1f TFLOP/s\n", name, s / 1024 / (iters * 4.0), flop / (ms * 1e-3) / 1e12);
}

int main() {
  int* in; float* out; unsigned long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096); hipMalloc(&cyc, 1024 * 8);
  static int host[4096];
  for (int i = 0; i < 4096; ++i) host[i] = 0x38383838 + (i % 7) * 0x01010101;
  hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice);
  run<-1>("f16 32x32x16", in, out, cyc);
  run<1>("scale bf8 x bf8 32x32x64", in, out, cyc);
  run<0>("scale fp8 x fp8 32x32x64", in, out, cyc);
  run<2>("scale fp6 x fp6 32x32x64", in, out, cyc);
  run<4>("scale fp4 x fp4 32x32x64", in, out, cyc);
  run<1, 4>("scale bf8 x fp4 32x32x64", in, out, cyc);
  run<4, 1>("scale fp4 x bf8 32x32x64", in, out, cyc);
  run<1, 2>("scale bf8 x fp6 32x32x64", in, out, cyc);
  return 0;
}
