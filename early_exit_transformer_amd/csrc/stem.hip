// Conv1dSubampling + positional encoding (reference early_exit.py:24-48,620-621):
// two Conv1d(k=3, stride=2, pad=0) back to back (no activation between), transpose to
// [utterance][frame][channel], + sinusoid PE.  Both convolutions run as MFMA GEMMs on the ring
// pipeline of eec_device.h, with hi/lo-split fp16 operands:
//   conv1: rows = (b, t1), K = n_mels*3 in the weight tensor's own [ci][j] order,
//          A[row][3ci + j] = mel[b][ci][2 t1 + j]                       -> mid (scaled, fp16 planes)
//   conv2: rows = (b, t'), K = 3*D as three K=D passes j = 0..2,
//          A_j[row][ci] = mid[b][2 t' + j][ci]  (2 R + 1 staged frames, lane row stride 2)
// The input is un-logged power mel (util/data_loader.py:7-18: heavy-tailed, no upper bound, 5+ decades between the loud
// and the quiet frames of one utterance).  fp16 operands need a bounded domain, so every ROW of a product gets its own
// power-of-two scale (exact): a conv1 row (the 3-frame window of one half-rate frame) is multiplied by 2^-e,
// e = exponent(max |window|) - 15, and its fp32 accumulators by 2^e; the conv1 output row is stored as fp16 hi/lo planes
// in a scaled domain of ITS own (exponent array mid_e[B*T1]); conv2 sums three mid rows per output frame, one K = D pass
// each, and rescales the accumulators (lane = output frame) by 2^(e_j - e_(j+1)) between the passes.  One outlier bin or
// one loud utterance therefore costs no other frame its precision, and nothing saturates up to fp32's own range.
#include "eec_kernels.h"
#include <algorithm>

#include "eec_blocks.h"

namespace eec {

constexpr int kStemThreads = 512;
constexpr int kStemRows1 = 64;  // conv1 row tile: (utterance, frame) rows of the half-rate sequence
constexpr int kSPF = 4;

// exponent e such that |m| * 2^-e < 2^15 (m = a row's maximum magnitude); 0 for an all-zero / non-finite row
__device__ __forceinline__ int row_exponent(float m) {
  if (!(m > 0.f) || !(m < INFINITY)) return 0;
  int ex;
  (void)frexpf(m, &ex);  // m = f * 2^ex, f in [0.5, 1)
  return ex - 15;
}

// ---------------------------------------------------------------------------
// conv1.  K1 = n_mels * 3 (multiple of 16, <= 384); plane row stride (K1 + 8) halves.
// DIRECT: the one-convolution stem of Early_zipformer (Conv1dSubampling_Zipformer, early_exit.py:80-95): the result
// leaves as fp32 x[b][t1][:] = conv + bias + pe[t1] instead of the scaled fp16 planes that feed conv2.
// where the staged output tile of conv1 starts: inside the (dead) input planes when they hold it, else behind the row maxima
__host__ __device__ constexpr int stem1_stage_offset(int planes_bytes, int stage_bytes) {
  return planes_bytes >= stage_bytes ? 0 : (planes_bytes + 2 * kStemRows1 * 4 + 15) / 16 * 16;
}
template <int D, int NP, bool DIRECT>
__global__ __launch_bounds__(kStemThreads, 2) void stem_conv1_kernel(SubsampleArgs a) {
  constexpr int NW = Geo<D>::kNW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int K1 = a.n_mels * 3, ks = K1 / 16;
  const int ld = (K1 + 8) * 2, plane = kStemRows1 * ld;
  const int M1 = a.B * a.T1;
  const int row0 = blockIdx.x * kStemRows1;

  WRing<NP, kSPF, NW> r;
  const uint4* w_lane = a.w1p + (size_t)(NW * w) * ks * 128 + lane;
  const size_t nts = (size_t)ks * 128;
  ring_fill_32<NP, kSPF, NW>(r, w_lane, nts, ks);  // consumed by gemm_plain_ring (32x32x16 fragments)
  unsigned* row_max = (unsigned*)(smem + 2 * plane);  // [64] input rows, [64] output rows: |max| as fp32 bit patterns
  if (threadIdx.x < 2 * kStemRows1) row_max[threadIdx.x] = 0u;
  // stage A: thread = (row rr, channel group cg); 3 taps per (row, ci); the row's scale comes from its own maximum
  {
    const int rr = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int row = row0 + rr;
    const bool ok = row < M1;
    const int b = ok ? row / a.T1 : 0, t1 = ok ? row - b * a.T1 : 0;
    const float* src = a.mel + (size_t)b * a.n_mels * a.T + 2 * t1;
    constexpr int CI = 16;  // channels per thread: n_mels <= 128
    float v[CI][3];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < CI; ++i) {
      const int ci = cg + 8 * i;
      v[i][0] = v[i][1] = v[i][2] = 0.f;
      if (ok && ci < a.n_mels) {
        const float* p = src + (size_t)ci * a.T;
        v[i][0] = p[0], v[i][1] = p[1], v[i][2] = p[2];
      }
      m = fmaxf(m, fmaxf(fabsf(v[i][0]), fmaxf(fabsf(v[i][1]), fabsf(v[i][2]))));
    }
    __syncthreads();  // row_max zeroed
    atomicMax(&row_max[rr], __builtin_bit_cast(unsigned, m));  // non-negative floats order like their bit patterns
    __syncthreads();
    const float sc = ldexpf(1.0f, -row_exponent(__builtin_bit_cast(float, row_max[rr])));
#pragma unroll
    for (int i = 0; i < CI; ++i) {
      const int ci = cg + 8 * i;
      if (ci < a.n_mels) {
        const hl2_t s01 = split2<NP>(v[i][0] * sc, v[i][1] * sc), s2 = split2<NP>(v[i][2] * sc, 0.f);
        half_t* dh = (half_t*)(smem + rr * ld) + ci * 3;
        dh[0] = s01.hi[0], dh[1] = s01.hi[1], dh[2] = s2.hi[0];
        if (NP == 3) {
          half_t* dl = (half_t*)(smem + plane + rr * ld) + ci * 3;
          dl[0] = s01.lo[0], dl[1] = s01.lo[1], dl[2] = s2.lo[0];
        }
      }
    }
  }
  __syncthreads();
  // this lane's two rows and the exponents of their scaled input domains
  int rowl[2], el[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    rowl[mt] = row0 + mt * 32 + (lane & 31);
    el[mt] = row_exponent(__builtin_bit_cast(float, row_max[mt * 32 + (lane & 31)]));
  }
  f32x16 acc[2][NW];
  // accumulators start at bias * 2^-e (the row's scaled domain)
#pragma unroll
  for (int nt = 0; nt < NW; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *(const float4*)(a.b1 + 32 * (NW * w + nt) + 8 * g + 4 * hh);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float sc = ldexpf(1.0f, -el[mt]);
        acc[mt][nt][4 * g + 0] = bb.x * sc;
        acc[mt][nt][4 * g + 1] = bb.y * sc;
        acc[mt][nt][4 * g + 2] = bb.z * sc;
        acc[mt][nt][4 * g + 3] = bb.w * sc;
      }
    }
  const char* a_lane = smem + (lane & 31) * ld + hh * 16;
  gemm_plain_ring<NP, kSPF, NW>(acc, a_lane, ld, plane, w_lane, nts, ks, r);
  if constexpr (DIRECT) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = rowl[mt];
      if (row < M1) {
        const int t1 = row % a.T1;
        const float up = ldexpf(1.0f, el[mt]);
#pragma unroll
        for (int nt = 0; nt < NW; ++nt) {
          const int c0 = 32 * (NW * w + nt) + 4 * hh;
          float* dst = a.x + (size_t)row * D + c0;
          const float* pe = a.pe + (size_t)t1 * D + c0;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 p = *(const float4*)(pe + 8 * g);
            *(float4*)(dst + 8 * g) = make_float4(acc[mt][nt][4 * g + 0] * up + p.x, acc[mt][nt][4 * g + 1] * up + p.y,
                                                  acc[mt][nt][4 * g + 2] * up + p.z, acc[mt][nt][4 * g + 3] * up + p.w);
          }
        }
      }
    }
    return;
  }
  // the output row's own scaled domain: maximum over its D channels (8 waves x 2 lane halves hold parts of a row)
  unsigned* out_max = row_max + kStemRows1;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    float m = 0.f;
#pragma unroll
    for (int nt = 0; nt < NW; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) m = fmaxf(m, fabsf(acc[mt][nt][i]));
    atomicMax(&out_max[mt * 32 + (lane & 31)], __builtin_bit_cast(unsigned, m));
  }
  __syncthreads();
  // mid planes [B*T1][D]: lane = row, register quad = 4 consecutive channels; mid_e[row] = exponent of the row's domain.
  // The tile leaves through LDS (the input planes are dead; the staging area sits in them, or behind the row maxima when they are
  // too small): straight from the accumulators a store instruction is 32 rows x 16 B, from the staged tile it is 8 KB of whole rows.
  constexpr int kSLd = D * 2 + 16;  // staged row stride (bytes)
  char* const stg = smem + stem1_stage_offset(2 * plane, kStemRows1 * kSLd);
  h4 lo4[2][NW][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = rowl[mt];
    const int eo = row_exponent(__builtin_bit_cast(float, out_max[mt * 32 + (lane & 31)]));  // relative to the input domain
    if (row < M1 && w == 0 && hh == 0) a.mid_e[row] = el[mt] + eo;
    const float sc = ldexpf(1.0f, -eo);
#pragma unroll
    for (int nt = 0; nt < NW; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const hl2_t s0 = split2<NP>(acc[mt][nt][4 * g + 0] * sc, acc[mt][nt][4 * g + 1] * sc);
        const hl2_t s1 = split2<NP>(acc[mt][nt][4 * g + 2] * sc, acc[mt][nt][4 * g + 3] * sc);
        h4 hi;
        hi.xy = s0.hi, hi.zw = s1.hi, lo4[mt][nt][g].xy = s0.lo, lo4[mt][nt][g].zw = s1.lo;
        *(h4*)(stg + (mt * 32 + (lane & 31)) * kSLd + (32 * (NW * w + nt) + 8 * g + 4 * hh) * 2) = hi;
      }
  }
  auto rows_out = [&](half_t* dstp) {
    constexpr int PPR = D / 8;  // 16-byte pieces per row
    for (int p = threadIdx.x; p < kStemRows1 * PPR; p += kStemThreads) {
      const int rl = p / PPR, c16 = p - rl * PPR;
      if (row0 + rl < M1) *(uint4*)(dstp + (size_t)(row0 + rl) * D + c16 * 8) = *(const uint4*)(stg + rl * kSLd + c16 * 16);
    }
  };
  __syncthreads();
  rows_out(a.mid_hi);
  if constexpr (NP == 3) {
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NW; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(h4*)(stg + (mt * 32 + (lane & 31)) * kSLd + (32 * (NW * w + nt) + 8 * g + 4 * hh) * 2) = lo4[mt][nt][g];
    __syncthreads();
    rows_out(a.mid_lo);
  }
}

// ---------------------------------------------------------------------------
// conv2 + PE.  One workgroup = Geo<D>::kRows consecutive output frames of one utterance.
template <int D>
struct Stem2Geo {
  static constexpr int kLd = D * 2 + 8;                       // staged frame stride (bytes): lane row stride 2 kLd is conflict-free
  static constexpr int kRowsIn = 2 * Geo<D>::kRows + 1;        // 129 / 65 staged mid frames
  static constexpr int kPlane = (kRowsIn * kLd + 15) / 16 * 16;
  static constexpr int kLds = 2 * kPlane;                      // 134176 / 134176
};

template <int D, int NP>
__global__ __launch_bounds__(kStemThreads, 2) void stem_conv2_kernel(SubsampleArgs a) {
  using G = Geo<D>;
  using S = Stem2Geo<D>;
  constexpr int MT = G::kMT, NW = G::kNW, KS = G::kKS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int plane = S::kPlane;
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * G::kRows;
  constexpr int ks_total = 3 * KS;  // k-steps per n-tile, ordered (j, ci)
  constexpr size_t nts = (size_t)ks_total * 128;
  WRing<NP, kSPF, NW> r;
  const uint4* w_lane = a.w2p + (size_t)(NW * w) * ks_total * 128 + lane;
  ring_fill<NP, kSPF, NW>(r, w_lane, nts, KS);
  // stage mid frames [2 t0, 2 t0 + 2 R] of utterance b: rows x D/8 sixteen-byte pieces per plane
  constexpr int PPR = D / 8;
  for (int p = threadIdx.x; p < S::kRowsIn * PPR; p += kStemThreads) {
    const int fr = p / PPR, c16 = p % PPR;
    const int t1 = 2 * t0 + fr;
    uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
    if (t1 < a.T1) {
      const size_t off = ((size_t)b * a.T1 + t1) * D + c16 * 8;
      vh = *(const uint4*)(a.mid_hi + off);
      if (NP == 3) vl = *(const uint4*)(a.mid_lo + off);
    }
    *(uint4*)(smem + fr * S::kLd + c16 * 16) = vh;
    if (NP == 3) *(uint4*)(smem + plane + fr * S::kLd + c16 * 16) = vl;
  }
  __syncthreads();
  // exponents of this lane's three input rows per output frame (mid rows 2 t + j of utterance b)
  int ej[MT][3];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t = t0 + mt * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 3; ++j) ej[mt][j] = (2 * t + j < a.T1) ? a.mid_e[(size_t)b * a.T1 + 2 * t + j] : 0;
  }
  f32x16 acc[MT][NW];
#pragma unroll
  for (int nt = 0; nt < NW; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *(const float4*)(a.b2 + 32 * (NW * w + nt) + 8 * g + 4 * hh);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float sc = ldexpf(1.0f, -ej[mt][0]);
        acc[mt][nt][4 * g + 0] = bb.x * sc;
        acc[mt][nt][4 * g + 1] = bb.y * sc;
        acc[mt][nt][4 * g + 2] = bb.z * sc;
        acc[mt][nt][4 * g + 3] = bb.w * sc;
      }
    }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const char* a_lane = smem + (2 * (lane & 31) + j) * S::kLd + hh * 16;
    gemm_ring<NP, KS, NW, true, kSPF, NoSide, 0, MT>(acc, a_lane, 2 * S::kLd, plane, w_lane + (size_t)j * KS * 128, nts, r);
    if (j < 2) {
      ring_fill<NP, kSPF, NW>(r, w_lane + (size_t)(j + 1) * KS * 128, nts, KS);
      // into the next row's domain (exact: powers of two; the clamp keeps the shift inside ldexp's exact range)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int d = max(-120, min(120, ej[mt][j] - ej[mt][j + 1]));
#pragma unroll
        for (int nt = 0; nt < NW; ++nt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mt][nt][i] = ldexpf(acc[mt][nt][i], d);
      }
    }
  }
  // The fp32 tile leaves through LDS (the staged mid frames are dead once every wave is past its three passes): a wave then adds the
  // positional encoding to, and stores, whole rows (1 KiB per instruction) -- from the accumulators a store instruction and a load of
  // the encoding are 32 rows x 32 B each.
  constexpr int kSLd2 = D * 4 + 16;
  static_assert(G::kRows * kSLd2 <= S::kLds, "the staged output tile fits the input planes");
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float up = ldexpf(1.0f, ej[mt][2]);
#pragma unroll
    for (int nt = 0; nt < NW; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(smem + (mt * 32 + (lane & 31)) * kSLd2 + (32 * (NW * w + nt) + 8 * g + 4 * hh) * 4) =
            make_float4(acc[mt][nt][4 * g + 0] * up, acc[mt][nt][4 * g + 1] * up, acc[mt][nt][4 * g + 2] * up, acc[mt][nt][4 * g + 3] * up);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < G::kRPW; ++i) {
    const int rl = w * G::kRPW + i, t = t0 + rl;
    if (t < a.Tq) {
      RowV<G::kQ> v = load_row<D>((const float*)(smem + rl * kSLd2), lane);
      const RowV<G::kQ> p = load_row<D>(a.pe + (size_t)t * D, lane);
#pragma unroll
      for (int q = 0; q < G::kQ; ++q) v.p[q].x += p.p[q].x, v.p[q].y += p.p[q].y, v.p[q].z += p.p[q].z, v.p[q].w += p.p[q].w;
      store_row<D>(a.x + ((size_t)b * a.Tq + t) * D, v, lane);
    }
  }
}

template <int D>
static hipError_t launch_subsample_d(const SubsampleArgs& a, int np, hipStream_t st) {
  const int K1 = a.n_mels * 3;
  // input planes + row maxima, and the staged output tile (inside the planes when they are large enough, behind the maxima otherwise)
  constexpr int kStage = kStemRows1 * (D * 2 + 16);
  const int planes = 2 * kStemRows1 * (K1 + 8) * 2;
  const int lds1 = std::max(planes + 2 * kStemRows1 * 4, stem1_stage_offset(planes, kStage) + kStage);
  auto k1 = np == 3 ? stem_conv1_kernel<D, 3, false> : stem_conv1_kernel<D, 1, false>;
  auto k2 = np == 3 ? stem_conv2_kernel<D, 3> : stem_conv2_kernel<D, 1>;
  if (hipError_t e = ensure_max_lds((const void*)k1, std::max(2 * kStemRows1 * (384 + 8) * 2 + 2 * kStemRows1 * 4, 2 * kStage + 1024)); e != hipSuccess) return e;
  if (hipError_t e = ensure_max_lds((const void*)k2, Stem2Geo<D>::kLds); e != hipSuccess) return e;
  const int M1 = a.B * a.T1;
  hipLaunchKernelGGL(k1, dim3((M1 + kStemRows1 - 1) / kStemRows1), dim3(kStemThreads), lds1, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k2, dim3((a.Tq + Geo<D>::kRows - 1) / Geo<D>::kRows, a.B), dim3(kStemThreads), Stem2Geo<D>::kLds, st, a);
  return hipGetLastError();
}
hipError_t launch_subsample(const SubsampleArgs& a, int np, hipStream_t st) {
  const int K1 = a.n_mels * 3;
  if (K1 % 16 || K1 > 384 || !a.mid_e) return hipErrorInvalidValue;
  return a.D == 512 ? launch_subsample_d<512>(a, np, st) : launch_subsample_d<256>(a, np, st);
}

// one Conv1d(k=3, s=2) + PE: x [B][T1][D] fp32 (the Early_zipformer stem); hi/lo split operands
template <int D>
static hipError_t launch_subsample_single_d(const SubsampleArgs& a, hipStream_t st) {
  const int K1 = a.n_mels * 3;
  auto k1 = stem_conv1_kernel<D, 3, true>;
  if (hipError_t e = ensure_max_lds((const void*)k1, 2 * kStemRows1 * (384 + 8) * 2 + 2 * kStemRows1 * 4); e != hipSuccess) return e;
  const int M1 = a.B * a.T1;
  hipLaunchKernelGGL(k1, dim3((M1 + kStemRows1 - 1) / kStemRows1), dim3(kStemThreads), 2 * kStemRows1 * (K1 + 8) * 2 + 2 * kStemRows1 * 4, st, a);
  return hipGetLastError();
}
hipError_t launch_subsample_single(const SubsampleArgs& a, hipStream_t st) {
  const int K1 = a.n_mels * 3;
  if (K1 % 16 || K1 > 384) return hipErrorInvalidValue;
  return a.D == 512 ? launch_subsample_single_d<512>(a, st) : launch_subsample_single_d<256>(a, st);
}

}  // namespace eec
