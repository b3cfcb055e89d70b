// Weight packing, length conversion and greedy CTC decode.
#include <map>
#include <mutex>
#include <utility>

#include "eec_kernels.h"

namespace eec {

hipError_t ensure_max_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;  // (kernel, device) -> bytes already granted
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({kernel, dev});
  if (it != done.end() && it->second >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done[{kernel, dev}] = bytes;
  return e;
}

// W[N][K] fp32 (torch Linear / 1x1-conv layout) -> MFMA fragments, hi and lo fp16 planes.
// out[((nt*KS + s)*2 + plane)*64 + lane] = 8 halves W[32nt + (lane&31)][16s + 8(lane>>5) + j]
__global__ void pack_frags_kernel(const float* __restrict__ w, int N, int K, uint4* __restrict__ out, int total,
                                  float scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, frag = idx >> 6;
  const int KS = K / 16;
  const int nt = frag / KS, s = frag - nt * KS;
  const int n = nt * 32 + (lane & 31), k0 = s * 16 + 8 * (lane >> 5);
  h8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (n < N) ? w[(size_t)n * K + k0 + j] * scale : 0.f;
    EEC_SPLIT(v, hi, lo, j);
  }
  out[(size_t)frag * 128 + lane] = __builtin_bit_cast(uint4, hi);
  out[(size_t)frag * 128 + 64 + lane] = __builtin_bit_cast(uint4, lo);
}

// The same fragment layout from a strided source, element (n, k) = w[n * ldn + k * ldk], as bf16 hi / lo planes (hi = RNE(x),
// lo = RNE(x - hi)): the operands of the training step's fused feed-forward backward, which multiplies by W2^T and W1^T.
__global__ void pack_frags_bf16_kernel(const float* __restrict__ w, int N, int K, long ldn, long ldk, uint4* __restrict__ out, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, frag = idx >> 6;
  const int KS = K / 16;
  const int nt = frag / KS, s = frag - nt * KS;
  const int n = nt * 32 + (lane & 31), k0 = s * 16 + 8 * (lane >> 5);
  bf8_t hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (n < N) ? w[(size_t)n * ldn + (size_t)(k0 + j) * ldk] : 0.f;
    const __bf16 h = (__bf16)v;
    hi[j] = h;
    lo[j] = (__bf16)(v - (float)h);
  }
  out[(size_t)frag * 128 + lane] = __builtin_bit_cast(uint4, hi);
  out[(size_t)frag * 128 + 64 + lane] = __builtin_bit_cast(uint4, lo);
}
hipError_t launch_pack_frags_bf16(const float* w, int N, int K, long ldn, long ldk, uint4* out, hipStream_t st) {
  if (K % 16) return hipErrorInvalidValue;
  const int total = ((N + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(pack_frags_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, N, K, ldn, ldk, out, total);
  return hipGetLastError();
}

// Fragment images of up to kFfnPackModules feed-forward modules in ONE launch.  (The training step packs a module's two images for a
// direction right before that direction's fused launch: packed for the whole model at the start of the forward, 192 MB, they have left
// the Infinity Cache by the time they are streamed -- the forward's fused launch then takes 192 instead of 136 us.)  Per module: out[0] = W1 as [F][D] and out[1] = W2 as
// [D][F], fp16 hi / lo (the forward's operands); out[2] = W2^T as [F][D] and out[3] = W1^T as [D][F], bf16 hi / lo (the backward's).
__global__ void pack_ffn_batch_kernel(FfnPackJobs jb, int F, int D, int kind0, int nk) {  // images kind0 .. kind0 + nk - 1 of every module
  const int bpj = F * D / 8 / 256;  // blocks per image (every image has F * D / 8 threads)
  const int job = blockIdx.x / bpj, mod = job / nk, kind = kind0 + job - mod * nk;
  const int idx = (blockIdx.x - job * bpj) * 256 + threadIdx.x;
  const int N = (kind & 1) ? D : F, K = (kind & 1) ? F : D;
  const float* __restrict__ w = (kind == 0 || kind == 3) ? jb.w1[mod] : jb.w2[mod];
  uint4* __restrict__ out = jb.out[mod][kind];
  const long ldn = kind < 2 ? K : 1, ldk = kind < 2 ? 1 : N;
  const int lane = idx & 63, frag = idx >> 6, KS = K / 16;
  const int nt = frag / KS, s = frag - nt * KS;
  const int n = nt * 32 + (lane & 31), k0 = s * 16 + 8 * (lane >> 5);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = w[(size_t)n * ldn + (size_t)(k0 + j) * ldk];
  if (kind < 2) {
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) EEC_SPLIT(v[j], hi, lo, j);
    out[(size_t)frag * 128 + lane] = __builtin_bit_cast(uint4, hi);
    out[(size_t)frag * 128 + 64 + lane] = __builtin_bit_cast(uint4, lo);
  } else {
    bf8_t hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 h = (__bf16)v[j];
      hi[j] = h;
      lo[j] = (__bf16)(v[j] - (float)h);
    }
    out[(size_t)frag * 128 + lane] = __builtin_bit_cast(uint4, hi);
    out[(size_t)frag * 128 + 64 + lane] = __builtin_bit_cast(uint4, lo);
  }
}
hipError_t launch_pack_ffn_batch(const FfnPackJobs& jb, int F, int D, int kind0, int nk, hipStream_t st) {
  if (jb.n < 1 || jb.n > kFfnPackModules || F % 32 || D % 32 || (F * D / 8) % 256 || kind0 < 0 || nk < 1 || kind0 + nk > 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_ffn_batch_kernel, dim3((unsigned)(jb.n * nk * (F * D / 8 / 256))), dim3(256), 0, st, jb, F, D, kind0, nk);
  return hipGetLastError();
}

hipError_t launch_pack_frags(const float* w, int N, int K, uint4* out, float scale, hipStream_t st) {
  if (K % 16) return hipErrorInvalidValue;
  const int total = ((N + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(pack_frags_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, N, K, out, total, scale);
  return hipGetLastError();
}

// "f8" stream for the fp16 + 2 x fp8-correction product (eec_device.h, NP == 8):
// per (n-tile, 64-k group) a record of kF8Rec uint4 = [4 fp16 hi fragments][lo8: 32 e5m2 bytes per lane]
// [E8M0 scale dword per lane].  lo = scale*W - fp16(scale*W) is quantised per 32-k block (two k-steps) to
// e5m2 after division by 2^(floor(log2 max|lo|) - 14); lane r + 32 b carries the scale of block b of row r.
__global__ void pack_frags_f8_kernel(const float* __restrict__ w, int N, int K, uint4* __restrict__ out, int total, float scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, rec = idx >> 6;
  const int NG = K / 64;
  const int nt = rec / NG, g = rec - nt * NG;
  const int r = lane & 31, h = lane >> 5, n = nt * 32 + r;
  uint4* o = out + (size_t)rec * kF8Rec;
  float lo[32];  // this lane's 32 slots: slot p <-> k = 64g + 16(p/8) + 8h + p%8
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h8 hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 64 * g + 16 * q + 8 * h + j;
      const float v = (n < N) ? w[(size_t)n * K + k] * scale : 0.f;
      hi[j] = to_half_sat(v);
      lo[8 * q + j] = (v - (float)hi[j]) * kF8WLoGain;  // gain: compensates the truncation of its partner a_hi8 (eec_device.h)
    }
    o[q * 64 + lane] = __builtin_bit_cast(uint4, hi);
  }
  // block b = slots [16b, 16b+16) of BOTH lane halves = k in [64g + 32b, 64g + 32b + 32): recompute the other half's residuals
  int ebias[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float m = 0.f;
    for (int kk = 0; kk < 32; ++kk) {
      const int k = 64 * g + 32 * b + kk;
      const float v = (n < N) ? w[(size_t)n * K + k] * scale : 0.f;
      m = fmaxf(m, fabsf(v - (float)to_half_sat(v)) * kF8WLoGain);
    }
    int e = 0;
    if (m > 0.f) e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127 - 14;  // scaled max lands in [2^14, 2^15)
    ebias[b] = min(max(e + 127, 0), 254);
  }
  unsigned bytes[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) bytes[d] = 0u;
#pragma unroll
  for (int p = 0; p < 32; ++p) {
    const float v = ldexpf(lo[p], 127 - ebias[p >> 4]);
    unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, (half_t)v);
    hb = min((hb & 0x7fffu) + 0x80u, 0x7b00u) | (hb & 0x8000u);  // round to nearest on the dropped byte, clamp to e5m2 max
    bytes[p >> 2] |= (hb >> 8) << (8 * (p & 3));
  }
  o[256 + 2 * lane] = make_uint4(bytes[0], bytes[1], bytes[2], bytes[3]);
  o[256 + 2 * lane + 1] = make_uint4(bytes[4], bytes[5], bytes[6], bytes[7]);
  ((int*)(o + 384))[lane] = ebias[h];  // lane r + 32 b supplies the scale of block b
}

hipError_t launch_pack_frags_f8(const float* w, int N, int K, uint4* out, float scale, hipStream_t st) {
  if (K % 64) return hipErrorInvalidValue;
  const int total = ((N + 31) / 32) * (K / 64) * 64;
  hipLaunchKernelGGL(pack_frags_f8_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, N, K, out, total, scale);
  return hipGetLastError();
}

// conv2 weight [co][ci][3] -> fragments of W'[co][k], k = j*cin + ci  (frame-major im2col order)
__global__ void pack_conv_jci_kernel(const float* __restrict__ w, int cout, int cin, uint4* __restrict__ out, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, frag = idx >> 6;
  const int K = 3 * cin, KS = K / 16;
  const int nt = frag / KS, s = frag - nt * KS;
  const int n = nt * 32 + (lane & 31), k0 = s * 16 + 8 * (lane >> 5);
  h8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k0 + i, j = k / cin, ci = k - j * cin;
    const float v = (n < cout) ? w[((size_t)n * cin + ci) * 3 + j] : 0.f;
    EEC_SPLIT(v, hi, lo, i);
  }
  out[(size_t)frag * 128 + lane] = __builtin_bit_cast(uint4, hi);
  out[(size_t)frag * 128 + 64 + lane] = __builtin_bit_cast(uint4, lo);
}

hipError_t launch_pack_conv_jci(const float* w, int cout, int cin, uint4* out, hipStream_t st) {
  if ((3 * cin) % 16) return hipErrorInvalidValue;
  const int total = ((cout + 31) / 32) * (3 * cin / 16) * 64;
  hipLaunchKernelGGL(pack_conv_jci_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, cout, cin, out, total);
  return hipGetLastError();
}

__global__ void scale_copy_kernel(const float* src, float* dst, int n, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * scale;
}

hipError_t launch_scale_copy(const float* src, float* dst, int n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, n, scale);
  return hipGetLastError();
}

// enc_len[b] = int(clamp(float(len) / 4, max = T'))   (reference early_exit.py:623: true division, truncation)
__global__ void enc_lengths_kernel(const long long* lengths, int B, int Tq, int* enc_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) enc_len[b] = (int)fminf((float)lengths[b] / 4.0f, (float)Tq);
}

hipError_t launch_enc_lengths(const long long* lengths, int B, int Tq, int* enc_len, hipStream_t st) {
  hipLaunchKernelGGL(enc_lengths_kernel, dim3((B + 63) / 64), dim3(64), 0, st, lengths, B, Tq, enc_len);
  return hipGetLastError();
}

// Small host arrays (the collate's `lengths`, train.py:34) reach the device inside a kernel's ARGUMENT block: no DMA and no
// cross-queue dependency on the stream (a host-to-device copy in front of the forward leaves a ~35 us hole behind it).
struct UploadArgs {
  long long v[kUploadMax];
};
__global__ void upload_i64_kernel(UploadArgs a, int n, long long* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a.v[i];
}
hipError_t launch_upload_i64(const long long* host, int n, long long* dev, hipStream_t st) {
  if (n < 0 || n > kUploadMax) return hipErrorInvalidValue;
  UploadArgs a;
  for (int i = 0; i < n; ++i) a.v[i] = host[i];
  for (int i = n; i < kUploadMax; ++i) a.v[i] = 0;
  hipLaunchKernelGGL(upload_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, n, dev);
  return hipGetLastError();
}

__global__ void fill_int_kernel(int* dst, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}

hipError_t launch_fill_int(int* dst, int n, int v, hipStream_t st) {
  hipLaunchKernelGGL(fill_int_kernel, dim3((n + 63) / 64), dim3(64), 0, st, dst, n, v);
  return hipGetLastError();
}

// Greedy CTC (reference util/beam_infer.py:9-24): argmax over labels, collapse repeats, drop blank.
// One wave per sequence; frames are walked 64 at a time, kept tokens compacted with a ballot.
// Ties in the argmax resolve to the lowest label index (torch.argmax on CPU).
__global__ __launch_bounds__(64) void greedy_ctc_kernel(const float* __restrict__ logp, int Tq, int V, int blank,
                                                        int* __restrict__ tokens, int* __restrict__ counts) {
  const int seq = blockIdx.x, lane = threadIdx.x;
  const float* base = logp + (size_t)seq * Tq * V;
  int* out = tokens + (size_t)seq * Tq;
  int n_out = 0, prev_last = -1;
  for (int t0 = 0; t0 < Tq; t0 += 64) {
    const int t = t0 + lane;
    int best = -1;
    if (t < Tq) {
      const float* row = base + (size_t)t * V;
      float bv = row[0];
      best = 0;
      for (int v = 1; v < V; ++v) {
        const float x = row[v];
        if (x > bv) {
          bv = x;
          best = v;
        }
      }
    }
    int prev = __shfl_up(best, 1, 64);
    if (lane == 0) prev = prev_last;
    const bool keep = (t < Tq) && (best != prev) && (best != blank);
    const unsigned long long mask = __ballot(keep);
    const int pos = __popcll(mask & ((1ull << lane) - 1ull));
    if (keep) out[n_out + pos] = best;
    n_out += __popcll(mask);
    const int last_lane = min(63, Tq - 1 - t0);
    prev_last = __shfl(best, last_lane, 64);
  }
  if (lane == 0) counts[seq] = n_out;
}

hipError_t launch_greedy_ctc(const float* logp, int n_seq, int Tq, int V, int blank, int* tokens, int* counts,
                             hipStream_t st) {
  hipLaunchKernelGGL(greedy_ctc_kernel, dim3(n_seq), dim3(64), 0, st, logp, Tq, V, blank, tokens, counts);
  return hipGetLastError();
}

}  // namespace eec
