// Microbenchmark for DESIGN.md section 9 item 1: can two workgroups of ONE kernel that run on different XCDs hand a
// tile to each other, and what does the hand-over cost?  Workgroup i writes 64 KiB, publishes a flag with an
// agent-scope release, its partner i ^ 1 (blockIdx % 8 differs: another XCD, another non-coherent L2) waits for the
// flag with agent-scope acquires and reads the tile back.  Every wait is BOUNDED (the kernel always drains).
// hipcc --offload-arch=gfx950 -O3 tools/xcd_exchange.hip -o tools/xcd_exchange && ./tools/xcd_exchange
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kTileWords = 16384;  // 64 KiB per workgroup
constexpr int kIters = 200;
constexpr int kMaxSpin = 2000000;

// Variant B: no cache-wide maintenance.  The tile is written with write-through stores (sc0 sc1) and read with loads that
// bypass the non-coherent L2 (sc0 sc1); the flags are relaxed device-scope atomics ordered by s_waitcnt vmcnt(0).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(unsigned* p, uint4 v) {
  const u32x4 r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ uint4 load_bypass(const unsigned* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return make_uint4(r[0], r[1], r[2], r[3]);
}
__global__ __launch_bounds__(256) void exchange_b(unsigned* tiles, unsigned* flags, unsigned* bad, unsigned long long* cyc, int* timeouts) {
  const int wg = blockIdx.x, partner = wg ^ 1, t = threadIdx.x;
  unsigned* mine = tiles + (size_t)wg * kTileWords;
  const unsigned* theirs = tiles + (size_t)partner * kTileWords;
  unsigned errors = 0;
  unsigned long long waited = 0;
  for (int it = 1; it <= kIters; ++it) {
    for (int i = t * 4; i < kTileWords; i += 1024) {
      const unsigned b = (unsigned)(wg * 1000003 + it * 7919 + i);
      store_wt(mine + i, make_uint4(b, b + 1, b + 2, b + 3));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores have been acknowledged
    __syncthreads();
    if (t == 0) {
      __hip_atomic_store(&flags[wg], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      int spin = 0;
      while (__hip_atomic_load(&flags[partner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)it && ++spin < kMaxSpin) __builtin_amdgcn_s_sleep(2);
      if (spin >= kMaxSpin) atomicAdd(timeouts, 1);
      waited += __builtin_amdgcn_s_memtime() - t0;
    }
    __syncthreads();
    for (int i = t * 4; i < kTileWords; i += 1024) {
      const uint4 v = load_bypass(theirs + i);
      const unsigned b = (unsigned)(partner * 1000003 + it * 7919 + i);
      errors += (v.x != b) + (v.y != b + 1) + (v.z != b + 2) + (v.w != b + 3);
    }
    __syncthreads();
    if (t == 0) {
      __hip_atomic_store(&flags[256 + wg], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      int spin = 0;
      while (__hip_atomic_load(&flags[256 + partner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)it && ++spin < kMaxSpin) __builtin_amdgcn_s_sleep(2);
      if (spin >= kMaxSpin) atomicAdd(timeouts, 1);
    }
    __syncthreads();
  }
  if (errors) atomicAdd(bad, errors);
  if (t == 0) cyc[wg] = waited;
}

__global__ __launch_bounds__(256) void exchange(unsigned* tiles, unsigned* flags, unsigned* bad, unsigned long long* cyc, int* timeouts) {
  const int wg = blockIdx.x, partner = wg ^ 1, t = threadIdx.x;
  unsigned* mine = tiles + (size_t)wg * kTileWords;
  const unsigned* theirs = tiles + (size_t)partner * kTileWords;
  unsigned errors = 0;
  unsigned long long waited = 0;
  for (int it = 1; it <= kIters; ++it) {
    for (int i = t; i < kTileWords; i += 256) mine[i] = (unsigned)(wg * 1000003 + it * 7919 + i);
    __syncthreads();  // all of this workgroup's stores are issued ...
    if (t == 0) {     // ... and made visible device-wide by the release below
      __hip_atomic_store(&flags[wg], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      int spin = 0;
      while (__hip_atomic_load(&flags[partner], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it && ++spin < kMaxSpin) __builtin_amdgcn_s_sleep(2);
      if (spin >= kMaxSpin) atomicAdd(timeouts, 1);
      waited += __builtin_amdgcn_s_memtime() - t0;
    }
    __syncthreads();
    // the acquire was done by one thread: the others need their own (cheap) acquire before reading the partner's tile
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // workgroup+ scope per HIP; use the agent-scope builtin below
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int i = t; i < kTileWords; i += 256)
      errors += theirs[i] != (unsigned)(partner * 1000003 + it * 7919 + i);
    __syncthreads();
    // second hand-shake: nobody overwrites a tile that its partner is still reading
    if (t == 0) {
      __hip_atomic_store(&flags[256 + wg], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spin = 0;
      while (__hip_atomic_load(&flags[256 + partner], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it && ++spin < kMaxSpin) __builtin_amdgcn_s_sleep(2);
      if (spin >= kMaxSpin) atomicAdd(timeouts, 1);
    }
    __syncthreads();
  }
  if (errors) atomicAdd(bad, errors);
  if (t == 0) cyc[wg] = waited;
}

int main() {
  unsigned *tiles, *flags, *bad; unsigned long long* cyc; int* timeouts;
  CK(hipMalloc(&tiles, (size_t)256 * kTileWords * 4)); CK(hipMalloc(&flags, 512 * 4)); CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&cyc, 256 * 8)); CK(hipMalloc(&timeouts, 4));
  CK(hipMemset(flags, 0, 512 * 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(timeouts, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int rc = 0;
  for (int variant = 0; variant < 2; ++variant) {
    CK(hipMemset(flags, 0, 512 * 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(timeouts, 0, 4));
    CK(hipEventRecord(e0));
    if (variant == 0)
      hipLaunchKernelGGL(exchange, dim3(256), dim3(256), 0, 0, tiles, flags, bad, cyc, timeouts);
    else
      hipLaunchKernelGGL(exchange_b, dim3(256), dim3(256), 0, 0, tiles, flags, bad, cyc, timeouts);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned hbad; int hto; static unsigned long long hc[256];
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hto, timeouts, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
    double w = 0; for (int i = 0; i < 256; ++i) w += hc[i];
    printf("%s\n  256 workgroups x %d hand-overs of 64 KiB with the partner on another XCD: %.1f us per round (write + publish + wait + read + ack)\n",
           variant == 0 ? "A: agent-scope release / acquire fences (whole-L2 write-back and invalidate)" : "B: write-through stores + L2-bypassing loads (sc0 sc1), relaxed flags",
           kIters, ms * 1e3 / kIters);
    printf("  stale words read: %u   bounded waits that ran out: %d   mean wait for the partner's flag: %.0f cycles\n", hbad, hto, w / 256 / kIters);
    rc |= (hbad || hto);
  }
  return rc;
}
