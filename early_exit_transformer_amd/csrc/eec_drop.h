// Dropout generator of the training step, device side (the sites and the Drop record are in eec_train.h): shared by the
// training kernels (train_kernels.hip) and the fused feed-forward forward of the training step (ffn.hip, TR variants), which
// must produce the masks the backward regenerates.
#pragma once
#include "eec_train.h"

namespace eect {

// counter-based: 32-bit mix (lowbias32) of the element index keyed by (seed, site)
__device__ __forceinline__ uint32_t drop_key(const Drop& d) {
  uint64_t x = d.seed * 0x9E3779B97F4A7C15ull + ((uint64_t)d.site << 32 | d.site);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)(x >> 32) ^ (uint32_t)x;
}
__device__ __forceinline__ bool drop_keep(uint32_t key, uint64_t i, uint32_t thr) {
  uint32_t h = (uint32_t)i * 0x9E3779B1u + (uint32_t)(i >> 32) * 0x85EBCA77u + key;
  h ^= h >> 16, h *= 0x7FEB352Du, h ^= h >> 15, h *= 0x846CA68Bu, h ^= h >> 16;
  return h >= thr;
}
__device__ __forceinline__ uint32_t drop_thr(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
// per-thread dropout state of one site; mul(i) = multiplier of element i (1 when p == 0)
struct DropState {
  uint32_t key, thr;
  float inv_keep;
  bool on;
  __device__ __forceinline__ explicit DropState(const Drop& d)
      : key(drop_key(d)), thr(drop_thr(d.p)), inv_keep(d.p > 0.0f ? 1.0f / (1.0f - d.p) : 1.0f), on(d.p > 0.0f) {}
  __device__ __forceinline__ float mul(uint64_t i) const { return !on ? 1.0f : (drop_keep(key, i, thr) ? inv_keep : 0.0f); }
  // the multipliers of elements i .. i + 3 (the same values as mul(i + j)): the index products of the hash are shared -- h(i + j) =
  // h(i) + j * C1 (+ C2 when the low word wraps) -- so four elements cost 10 quarter-rate 32-bit multiplies instead of 16
  // ... and of elements i, i + stride, i + 2 stride, i + 3 stride (3 * stride < 2^32): the attention kernels that hold a key per lane walk
  // the virtual [B H, T', T'] tensor down a column
  __device__ __forceinline__ void mul4s(uint64_t i, uint32_t stride, float (&m)[4]) const {
    if (!on) {
      m[0] = m[1] = m[2] = m[3] = 1.0f;
      return;
    }
    const uint32_t lo = (uint32_t)i, h0 = lo * 0x9E3779B1u + (uint32_t)(i >> 32) * 0x85EBCA77u + key, s1 = stride * 0x9E3779B1u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t h = h0 + (uint32_t)j * s1 + ((lo + (uint32_t)j * stride) < lo ? 0x85EBCA77u : 0u);
      h ^= h >> 16, h *= 0x7FEB352Du, h ^= h >> 15, h *= 0x846CA68Bu, h ^= h >> 16;
      m[j] = h >= thr ? inv_keep : 0.0f;
    }
  }
  __device__ __forceinline__ void mul4(uint64_t i, float (&m)[4]) const {
    if (!on) {
      m[0] = m[1] = m[2] = m[3] = 1.0f;
      return;
    }
    const uint32_t lo = (uint32_t)i, h0 = lo * 0x9E3779B1u + (uint32_t)(i >> 32) * 0x85EBCA77u + key;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t h = h0 + (uint32_t)j * 0x9E3779B1u + ((lo + (uint32_t)j) < lo ? 0x85EBCA77u : 0u);
      h ^= h >> 16, h *= 0x7FEB352Du, h ^= h >> 15, h *= 0x846CA68Bu, h ^= h >> 16;
      m[j] = h >= thr ? inv_keep : 0.0f;
    }
  }
};

}  // namespace eect
