// Microbenchmark: how fast can every CU stream the SAME weight buffer out of L2, as a function of
// how the 8 waves of a workgroup spread their 1-KiB loads over the address space?
//   ./l2bw   -> table of GB/s per access pattern (256 workgroups x 512 threads, 4 MiB buffer)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// each wave: for s in [0, nsteps): load 16 B/lane at base + wave*wave_stride + ((s + rot) % nsteps)*step_stride + piece*1024
template <int PIECES, int DEPTH>
__global__ __launch_bounds__(512, 2) void stream_kernel(const uint4* __restrict__ buf, size_t wave_stride_u4,
                                                        size_t step_stride_u4, int nsteps, int rot_mul, uint4* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint4* p = buf + w * wave_stride_u4 + lane;
  const int rot = (blockIdx.x * rot_mul) % nsteps;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint4 q[DEPTH][PIECES];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int s = (d + rot) % nsteps;
#pragma unroll
    for (int k = 0; k < PIECES; ++k) q[d][k] = p[s * step_stride_u4 + k * 64];
  }
  for (int s0 = 0; s0 < nsteps; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int k = 0; k < PIECES; ++k) {
        acc.x ^= q[d][k].x; acc.y ^= q[d][k].y; acc.z ^= q[d][k].z; acc.w ^= q[d][k].w;
      }
      const int sn = s0 + d + DEPTH;
      if (sn < nsteps) {
        const int s = (sn + rot) % nsteps;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) q[d][k] = p[s * step_stride_u4 + k * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (acc.x == 0x12345678u) sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
  const size_t bytes = 64u << 20;
  uint4 *buf, *sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 256 * 512 * 16));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { const char* name; size_t wave_stride, step_stride; int nsteps, rot_mul, grid; };
  // every wave reads nsteps * 2 KiB; 8 waves -> 4 MiB per workgroup when nsteps = 256
  std::vector<Case> cases = {
      {"W1-like: wave stride 32K, step 2K (tile-major)      ", 32 << 10, 2 << 10, 16, 0, 256},
      {"long   : wave stride 512K, step 2K                   ", 512 << 10, 2 << 10, 256, 0, 256},
      {"W2-like: wave stride 256K, step 2K                   ", 256 << 10, 2 << 10, 128, 0, 256},
      {"interleaved: wave stride 2K, step 16K                ", 2 << 10, 16 << 10, 256, 0, 256},
      {"interleaved + per-WG rotation                        ", 2 << 10, 16 << 10, 256, 37, 256},
      {"long + per-WG rotation                               ", 512 << 10, 2 << 10, 256, 37, 256},
      {"long, skewed wave stride 512K+2K                     ", (512 << 10) + (2 << 10), 2 << 10, 255, 0, 256},
      {"long, skewed wave stride 512K+4K                     ", (512 << 10) + (4 << 10), 2 << 10, 254, 0, 256},
      {"long, skewed wave stride 512K+256                    ", (512 << 10) + 256, 2 << 10, 255, 0, 256},
      {"long, 1024 WGs                                       ", 512 << 10, 2 << 10, 256, 0, 1024},
  };
  for (auto& c : cases) {
    auto k = stream_kernel<2, 4>;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(c.grid), dim3(512), 0, 0, buf, c.wave_stride / 16, c.step_stride / 16, c.nsteps, c.rot_mul, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k, dim3(c.grid), dim3(512), 0, 0, buf, c.wave_stride / 16, c.step_stride / 16, c.nsteps, c.rot_mul, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double gb = (double)c.grid * 8 * c.nsteps * 2048.0 * reps / 1e9;
    printf("%s  %8.1f us/launch  %8.1f GB/s  (%.1f MB per WG)\n", c.name, ms / reps * 1e3, gb / (ms * 1e-3), 8.0 * c.nsteps * 2048 / 1e6);
  }
  // deeper prefetch / single piece variants on the 'long' pattern
  {
    auto run = [&](const char* nm, auto k, int pieces) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, buf, (512 << 10) / 16, (size_t)(pieces * 1024) / 16, 256 * 2 / pieces, 0, sink);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, buf, (512 << 10) / 16, (size_t)(pieces * 1024) / 16, 256 * 2 / pieces, 0, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s  %8.1f us/launch  %8.1f GB/s\n", nm, ms / 20 * 1e3, 256.0 * 8 * 512 * 1024 * 20 / 1e9 / (ms * 1e-3));
    };
    run("long, depth 8 x 2 pieces                            ", stream_kernel<2, 8>, 2);
    run("long, depth 4 x 4 pieces                            ", stream_kernel<4, 4>, 4);
    run("long, depth 8 x 1 piece                             ", stream_kernel<1, 8>, 1);
  }
  return 0;
}
