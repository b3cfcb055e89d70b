// Microbenchmark (round 3): does a weight stream that reaches the MFMA waves through an LDS ring filled by dedicated loader
// waves (LDS-DMA, `global_load_lds_dwordx4`) run a chain-kernel slot faster than the production form, where every compute
// wave streams its own fragments straight into a register ring?
//
// The model is one steady-state slot of ffn_chain_kernel<256, 8, ...> (f16f8): 8 compute waves per CU (2 per SIMD), each
// consuming 24 KiB of fragment-major weights per slot (16 KiB fp16 + 8 KiB e5m2 residual; 192 KiB per CU and slot, every CU
// reading the SAME bytes out of L2) for 32 fp16 MFMAs (32x32x16) and 16 block-scaled fp8 MFMAs (32x32x64) whose other operand
// comes from LDS planes; the producers' SiLU is modelled by VALU filler.  A slot is walked in 8 sub-steps of 3 KiB per wave.
//
//   reg<PF>    every compute wave: `global_load_dwordx4` into a register ring PF sub-steps deep (production form)
//   dma<NB>    4 loader waves (one per SIMD) fill an LDS ring of NB sub-step buffers (24 KiB each) by LDS-DMA, compute waves
//              read their fragments with ds_read_b128; one workgroup barrier per sub-step; NB = 2: `vmcnt(0)` before the barrier,
//              NB = 3: one sub-step stays in flight across the barrier (counted vmcnt, raw s_barrier)
//
//   hipcc -O3 --offload-arch=gfx950 tools/dma_ring_bench.hip -o /tmp/dma_ring_bench && /tmp/dma_ring_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSub = 8;                          // sub-steps per slot
constexpr int kFrag = 3;                         // 1-KiB fragments per compute wave and sub-step: 2 fp16 + half an e5m2 one
constexpr int kSubCU = 8 * kFrag * 1024;         // 24 KiB per CU and sub-step
constexpr int kSlot = kSub * kSubCU;             // 192 KiB
constexpr int kALd = 528, kA8Ld = 272;           // activation planes as in the chain kernel: [64][256 + 8] fp16, [64][256 + 16] bytes
constexpr int kAHi = 64 * kALd, kABytes = kAHi + 64 * kA8Ld;  // 51 200

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma8(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// the arithmetic of one sub-step of one compute wave: 4 fp16 MFMAs, and on odd sub-steps 4 fp8 MFMAs; `filler` VALU ops
template <int FILL>
__device__ __forceinline__ void sub_step(f32x16 (&acc)[2], const char* a_lane, const char* a8_lane, int t, u32x4 w0, u32x4 w1, u32x4 wl,
                                         u32x4& wl_prev, float& fill) {
  const int ks = (t * 2) & 15;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const h8 b = __builtin_bit_cast(h8, j ? w1 : w0);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const h8 a = *(const h8*)(a_lane + m * 32 * kALd + (ks + j) * 32);
      acc[m] = mfma16(a, b, acc[m]);
    }
  }
  if (t & 1) {
    const i32x8 blo = {(int)wl_prev[0], (int)wl_prev[1], (int)wl_prev[2], (int)wl_prev[3], (int)wl[0], (int)wl[1], (int)wl[2], (int)wl[3]};
    const i32x8 bhi = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
    const int k8 = (t >> 1) & 3;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const u32x4 p = *(const u32x4*)(a8_lane + m * 32 * kA8Ld + k8 * 64), q = *(const u32x4*)(a8_lane + m * 32 * kA8Ld + k8 * 64 + 16);
      const i32x8 a8 = {(int)p[0], (int)p[1], (int)p[2], (int)p[3], (int)q[0], (int)q[1], (int)q[2], (int)q[3]};
      acc[m] = mfma8(a8, blo, acc[m]);
      acc[m] = mfma8(a8, bhi, acc[m]);
    }
  }
  wl_prev = wl;
#pragma unroll
  for (int i = 0; i < FILL; ++i) fill = __builtin_fmaf(fill, 1.0001f, 0.25f);
}

__device__ __forceinline__ void init_planes(char* smem, int tid, int nthreads) {
  for (int i = tid; i < kABytes / 4; i += nthreads) ((unsigned*)smem)[i] = 0x2c002c00u + (i & 3);
}

// ---- production form: every compute wave streams its own fragments into registers ----
template <int PF, int FILL>
__global__ __launch_bounds__(512, 2) void reg_kernel(const uint4* __restrict__ W, float* out, unsigned long long* cyc, int nslots, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  init_planes(smem, threadIdx.x, 512);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const char* a_lane = smem + (lane & 31) * kALd + (lane >> 5) * 16;
  const char* a8_lane = smem + kAHi + (lane & 31) * kA8Ld + (lane >> 5) * 32;
  const uint4* wl_base = W + (size_t)w * kFrag * 64 + lane;  // + t * kSubCU / 16 + f * 64
  const int T = nslots * kSub;
  f32x16 acc[2] = {};
  float fill = (float)lane;
  u32x4 wl_prev = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    uint4 r[PF][kFrag];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
      for (int f = 0; f < kFrag; ++f) r[d][f] = wl_base[(size_t)d * (kSubCU / 16) + f * 64];
    for (int tb = 0; tb < T; tb += PF) {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        const int t = tb + d;
        const u32x4 w0 = __builtin_bit_cast(u32x4, r[d][0]), w1 = __builtin_bit_cast(u32x4, r[d][1]), wl = __builtin_bit_cast(u32x4, r[d][2]);
        const int tn = t + PF < T ? t + PF : t;  // the tail re-reads: keeps the loop uniform
#pragma unroll
        for (int f = 0; f < kFrag; ++f) r[d][f] = wl_base[(size_t)tn * (kSubCU / 16) + f * 64];
        __builtin_amdgcn_sched_barrier(0);
        if (w < 4) sub_step<FILL>(acc, a_lane, a8_lane, t, w0, w1, wl, wl_prev, fill);
        else sub_step<0>(acc, a_lane, a8_lane, t, w0, w1, wl, wl_prev, fill);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (((tb / PF) & 7) == 7) __syncthreads();  // the chain kernel's one barrier per slot (PF-independent enough)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
  float s = fill;
  for (int m = 0; m < 2; ++m) s += acc[m][3];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

// ---- loader waves + LDS ring ----
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int NB, int FILL>
__global__ __launch_bounds__(768) void dma_kernel(const uint4* __restrict__ W, float* out, unsigned long long* cyc, int nslots, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // planes, then the ring: ONE shared object (a second one makes hipcc drain vmcnt)
  init_planes(smem, threadIdx.x, 768);
  __syncthreads();
  char* ring = smem + kABytes;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int T = nslots * kSub;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (w >= 8) {
    // loader L fills fragments [6 L, 6 L + 6) of every sub-step buffer
    const int L = w - 8;
    const uint4* src = W + (size_t)L * 6 * 64 + lane;
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int f = 0; f < 6; ++f)
        __builtin_amdgcn_global_load_lds((glb_void*)(src + (size_t)t * (kSubCU / 16) + f * 64), (lds_void*)(ring + buf * kSubCU + (L * 6 + f) * 1024), 16, 0, 0);
    };
    for (int rep = 0; rep < reps; ++rep) {
      // prologue: NB - 1 sub-steps ahead
      for (int d = 0; d < NB - 1; ++d) issue(d, d);
      if (NB == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int t = 0; t < T; ++t) {
        // buffer (t + NB - 1) % NB was read during sub-step t - 1: free since the barrier that ended it
        const int tn = t + NB - 1 < T ? t + NB - 1 : T - 1;
        issue(tn, (t + NB - 1) % NB);
        if (NB == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // sub-step t + 1 landed
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // sub-step t + 1 landed, t + 2 in flight across the barrier
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    const char* a_lane = smem + (lane & 31) * kALd + (lane >> 5) * 16;
    const char* a8_lane = smem + kAHi + (lane & 31) * kA8Ld + (lane >> 5) * 32;
    const char* my = ring + (w * kFrag) * 1024 + lane * 16;
    f32x16 acc[2] = {};
    float fill = (float)lane;
    u32x4 wl_prev = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int t = 0; t < T; ++t) {
        const char* b = my + (t % NB) * kSubCU;
        const u32x4 w0 = *(const u32x4*)b, w1 = *(const u32x4*)(b + 1024), wl = *(const u32x4*)(b + 2048);
        if (w < 4) sub_step<FILL>(acc, a_lane, a8_lane, t, w0, w1, wl, wl_prev, fill);
        else sub_step<0>(acc, a_lane, a8_lane, t, w0, w1, wl, wl_prev, fill);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    float s = fill;
    for (int m = 0; m < 2; ++m) s += acc[m][3];
    if (s == 1.2345f) out[threadIdx.x] = s;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
}

template <typename K>
static void run(const char* name, K k, int threads, size_t lds, const uint4* W, float* out, unsigned long long* cyc, int nslots) {
  const int reps = 4;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), lds, 0, W, out, cyc, nslots, reps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int launches = 5;
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(threads), lds, 0, W, out, cyc, nslots, reps);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  static unsigned long long h[256 * 16];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double s = 0;
  for (int b = 0; b < 256; ++b) s += (double)h[b * 16];
  const double slots = (double)nslots * reps;
  printf("%-44s %7.0f cycles/slot  %6.2f us/slot  %6.1f TB/s L2->CU  (MFMA floor 4096 cycles/slot)\n", name, s / 256 / slots, ms * 1e3 / launches / slots,
         256.0 * kSlot * slots * launches / (ms * 1e-3) / 1e12);
}

int main() {
  const int nslots = 32;  // 6.3 MB of weights per pass, as one two-stage launch
  uint4* W; float* out; unsigned long long* cyc;
  CK(hipMalloc(&W, (size_t)(nslots + 2) * kSlot)); CK(hipMemset(W, 0x11, (size_t)(nslots + 2) * kSlot));
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 256 * 16 * 8));
  const size_t a = kABytes;
  for (int round = 0; round < 2; ++round) {
    run("reg ring 3 sub-steps (9 KiB/wave)", reg_kernel<3, 0>, 512, a, W, out, cyc, nslots);
    run("reg ring 4 sub-steps (12 KiB/wave)", reg_kernel<4, 0>, 512, a, W, out, cyc, nslots);
    run("reg ring 6 sub-steps (18 KiB/wave)", reg_kernel<6, 0>, 512, a, W, out, cyc, nslots);
    run("reg ring 4 + SiLU filler 100 VALU/sub-step", reg_kernel<4, 100>, 512, a, W, out, cyc, nslots);
    run("dma ring 2 buffers (48 KiB)", dma_kernel<2, 0>, 768, a + 2 * kSubCU, W, out, cyc, nslots);
    run("dma ring 3 buffers (72 KiB)", dma_kernel<3, 0>, 768, a + 3 * kSubCU, W, out, cyc, nslots);
    run("dma ring 2 + SiLU filler", dma_kernel<2, 100>, 768, a + 2 * kSubCU, W, out, cyc, nslots);
    run("dma ring 3 + SiLU filler", dma_kernel<3, 100>, 768, a + 3 * kSubCU, W, out, cyc, nslots);
  }
  return 0;
}
