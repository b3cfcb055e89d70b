// Fused Conformer feed-forward half-step (torchaudio _FeedForwardModule
// + the 0.5*y + x residual of ConformerLayer; a8 for the optional final LayerNorm):
//
//     x <- [LN_final]( 0.5 * ( W2 . silu( W1 . LN(x) + b1 ) + b2 ) + x )
//
// One 512-thread workgroup owns 64 rows of x; the [64, F] hidden activation never leaves the CU.
// F is walked in chunks of 128 hidden units.  The 8 waves are specialised (one wave of each kind
// per SIMD):
//   producers (waves 4-7): GEMM1 in the swapped orientation (lane = frame, register quad = 4
//       consecutive hidden units; wave wl makes hidden [32wl, 32wl+32) of the chunk for all 64
//       rows; accumulators start at the bias), SiLU in the exp2 domain (log2 e folded into W1/b1,
//       1/log2 e into W2), hi/lo split, ds_write_b64 into H[c & 1][frame][hidden];
//   consumers (waves 0-3): GEMM2 (normal orientation) of chunk c-1 from H[(c-1) & 1], wave wl
//       accumulating output columns [64wl, 64wl+64) in registers that live across all chunks.
// In slot s the producers multiply chunk s and, inside that k-loop (two values per k-step, in the
// shadow of the step's MFMAs), SiLU chunk s-1 into H[(s-1) & 1]; the consumers eat chunk s-2.  Both
// waves of a SIMD therefore run continuous MFMA streams of 32*NP MFMAs per slot, with ONE workgroup
// barrier per slot.  (A stand-alone SiLU phase halves the partner wave's MFMA rate and is pure
// critical path: measured 9.8k cycles per slot against 6.1k of MFMA work.)
// Each wave streams its own, disjoint weight fragments from L2 straight into a register ring
// (1 KiB per load) that runs PF k-steps ahead, across chunk boundaries.
//
// Algorithmic work: 4*D*F flop per row (2.097 MFLOP at D=256, F=2048); bound: MFMA.
// Executed MFMA work is NP x that.  HBM/L2 traffic per launch: x read+write 2 KiB/row; each
// workgroup streams all 2*D*F*2 B (x2 planes when NP=3) of weights once from L2.
#include "eec_kernels.h"

namespace eec {

constexpr int kFfnThreads = 512;
constexpr int kFC = 128;                          // hidden units per chunk (4 waves x 32)
constexpr int kHLd = (kFC + 8) * 2;               // 272
constexpr int kHPlane = kTileRows * kHLd;         // 17408
constexpr int kFfnLds = 2 * kAPlane + 4 * kHPlane;  // A hi/lo + per-group H hi/lo = 137216
// k-steps of W1 / W2 fragments a producer / consumer wave keeps in flight.  The shared weight stream
// out of L2 is latency x concurrency bound (tools/l2bw.hip: 64 KiB in flight per CU -> 18 TB/s,
// 128 KiB -> 28 TB/s), so the rings are as deep as the register budget allows.
#ifndef EEC_PF1_NP3
#define EEC_PF1_NP3 8
#define EEC_PF2_NP3 4
#define EEC_PF1_NP1 12
#define EEC_PF2_NP1 8
#endif
template <int NP> struct FfnPf { static constexpr int P1 = EEC_PF1_NP3, P2 = EEC_PF2_NP3; };
template <> struct FfnPf<1> { static constexpr int P1 = EEC_PF1_NP1, P2 = EEC_PF2_NP1; };
#ifndef EEC_PF1_NP8
#define EEC_PF1_NP8 6
#define EEC_PF2_NP8 3
#endif
template <> struct FfnPf<8> { static constexpr int P1 = EEC_PF1_NP8, P2 = EEC_PF2_NP8; };  // hi fragments only ride the ring in the f8 stream
constexpr int kH8Ld = kFC + 16;  // 144: H lo8 byte plane row stride (NP == 8)

#ifdef EEC_TIMELINE
// Diagnostic build only: s_memtime stamps of wave 0 (producer) and wave 4 (consumer) of the first
// 8 workgroups, written to a buffer nothing else reads.  Layout: [block][role][stamp], 64 stamps.
__device__ unsigned long long* g_timeline = nullptr;
__device__ __forceinline__ void tl_stamp(int& idx) {
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0 && (w == 0 || w == 4) && blockIdx.x < 8 && g_timeline && idx < 64)
    g_timeline[(blockIdx.x * 2 + (w >> 2)) * 64 + idx] = __builtin_amdgcn_s_memtime();
  ++idx;
}
#define TL_STAMP() tl_stamp(tl_idx)
#ifdef EEC_KSTEP_STAMPS
__device__ int g_ks_idx;  // stamps of consumer wave 0, block 0 only, slots 3..5
__device__ unsigned long long g_ks[256];
__device__ void eec_kstep_stamp() {
  if (threadIdx.x == 0 && blockIdx.x == 0 && g_ks_idx < 256) g_ks[g_ks_idx++] = __builtin_amdgcn_s_memtime();
}
extern "C" int eec_debug_ksteps(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  int zero = 0;
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ks), 256 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ks_idx), &zero, 4);
  return (int)e;
}
#endif
#else
#define TL_STAMP()
#endif

// ACT 0: SiLU in the exp2 domain (Conformer; log2 e folded into the packed weights); ACT 1: ReLU (the
// legacy pre-norm transformer layer, models/layers/position_wise_feed_forward.py:9-23, plain weights).
template <int NP, bool FINAL_LN, int ACT>
__global__ __launch_bounds__(kFfnThreads, 2) void ffn_kernel(float* __restrict__ x, int M, float res_scale,
                                                             const float* __restrict__ ln_g,
                                                             const float* __restrict__ ln_b,
                                                             const uint4* __restrict__ w1p,
                                                             const float* __restrict__ b1s,
                                                             const uint4* __restrict__ w2p,
                                                             const float* __restrict__ b2, int F,
                                                             const float* __restrict__ fin_g,
                                                             const float* __restrict__ fin_b,
                                                             const uint4* __restrict__ w1f8,
                                                             const uint4* __restrict__ w2f8) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kPF1 = FfnPf<NP>::P1, kPF2 = FfnPf<NP>::P2;
  const int lane = lane_id(), w = wave_id();
  const int hh = lane >> 5, wl = w & 3;
  const bool producer = w >= 4;  // wave-uniform; consumers are the OLDER waves (issue arbitration: priority, then age)
  const int row0 = blockIdx.x * kTileRows;
  char* lds_h = smem + 2 * kAPlane;  // H[buf][plane][64][136]
  const char* a_lane = smem + (lane & 31) * kALd + hh * 16;

  const int nft = F / 32;  // 32-wide hidden tiles
  const int nchunk = (nft + 3) / 4;
  const int ks2_total = F / 16;
  // The hidden chunks are summed in a ROTATED order that differs between the workgroups of an XCD
  // (blockIdx % 8 picks the XCD, blockIdx / 8 the slot within it): all 32 CUs of an XCD stream the
  // same weights out of the same L2, and walking them in lockstep makes every CU ask the same L2
  // channel for the same line at the same time.
#ifndef EEC_FFN_ROT
#define EEC_FFN_ROT 0
#endif
  const int rot = (EEC_FFN_ROT && (nft & 3) == 0) ? (int)((blockIdx.x >> 3) % (unsigned)nchunk) : 0;
  auto phys = [&](int c) { const int p = c + rot; return p >= nchunk ? p - nchunk : p; };
  const size_t w2_nt_stride = (size_t)ks2_total * 128;

#ifdef EEC_TIMELINE
  int tl_idx = 0;
#endif
  TL_STAMP();  // 0: kernel entry
  constexpr int RNP = NP == 8 ? 1 : NP;  // the f8 stream keeps only the hi fragments in the ring
  WRing<RNP, kPF1, 1> r1;
  WRing<RNP, kPF2, 2> r2;
  WGroupF8<1> wg1[4];  // NP == 8: lo8 + scales of a whole GEMM1 stage (K = 256 = 4 groups)
  WGroupF8<2> wg2[2];  //          ... of a whole GEMM2 stage (128 hidden = 2 groups, 2 n-tiles)
  const size_t w2f8_nt = (size_t)(F / 64) * kF8Rec;
  auto w1f8_lane = [&](int ft) { return w1f8 + (size_t)ft * 4 * kF8Rec + lane; };
  auto w2f8_lane = [&](int c) { return w2f8 + ((size_t)(2 * wl) * (F / 64) + 2 * c) * kF8Rec + lane; };
  auto fill1 = [&](int ft) {  // start the W1 stream of hidden tile ft
    if constexpr (NP == 8) {
      ring_fill_f8<kPF1, 1>(r1, w1f8_lane(ft), 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) f8_group_load<1>(wg1[g], w1f8_lane(ft) + (size_t)g * kF8Rec, 0);
    } else {
      ring_fill<RNP, kPF1, 1>(r1, w1p + (size_t)ft * (kD / 16) * 128 + lane, 0, kD / 16);
    }
  };
  auto fill2 = [&](int c) {  // start the W2 stream of chunk c
    if constexpr (NP == 8) {
      ring_fill_f8<kPF2, 2>(r2, w2f8_lane(c), w2f8_nt);
#pragma unroll
      for (int g = 0; g < 2; ++g) f8_group_load<2>(wg2[g], w2f8_lane(c) + (size_t)g * kF8Rec, w2f8_nt);
    } else {
      ring_fill<RNP, kPF2, 2>(r2, w2p + ((size_t)(2 * wl) * ks2_total + c * (kFC / 16)) * 128 + lane, w2_nt_stride,
                             min(kFC / 16, ks2_total - c * (kFC / 16)));
    }
  };
  // both weight streams start inside the LayerNorm prologue, right behind the x-row loads
  auto start_streams = [&]() {
    if (producer) {
      if (wl < nft) fill1(phys(0) * 4 + wl);
    } else {
      fill2(phys(0));
    }
  };
  rows_f32_to_planes<NP, true, 8>(smem, x, row0, M, ln_g, ln_b, start_streams);
  TL_STAMP();  // 1: prologue done
  __syncthreads();
  TL_STAMP();  // 2: after prologue barrier

  // The two roles run separate loops (so neither carries the other's registers); both execute
  // exactly nslots workgroup barriers.  Pipeline: chunk c is multiplied (GEMM1) in slot c, SiLU'd
  // and written to H[c & 1] in slot c+1 -- inside the k-loop of GEMM1(c+1), two values per k-step in
  // the shadow of that step's MFMAs -- and consumed (GEMM2) in slot c+2.
  const int nslots = nchunk + 2;
  f32x16 acc2c[2][2];  // consumers' [64 x 64] output accumulators (unused by producers)
  if (producer) {
    // SiLU + hi/lo split + ds_write of values [2q, 2q+1] of tile mt of a finished accumulator
    auto silu_pair = [&](const f32x16 (&acc)[2][1], char* hb, int step, h2& keep_hi, h2& keep_lo) {
      const int mt = step >> 3, q = step & 7;
      const float u0 = acc[mt][0][2 * q], u1 = acc[mt][0][2 * q + 1];
      constexpr int SNP = NP == 1 ? 1 : 3;
      const hl2_t sp = ACT == 0 ? split2<SNP>(silu_exp2(u0), silu_exp2(u1)) : split2<SNP>(fmaxf(u0, 0.f), fmaxf(u1, 0.f));
      if ((q & 1) == 0) {
        keep_hi = sp.hi;
        keep_lo = sp.lo;
      } else {
        char* dst = hb + (mt * 32 + (lane & 31)) * kHLd + (wl * 32 + 4 * hh) * 2 + (q >> 1) * 16;
        h4 hi, lo;
        hi.xy = keep_hi, hi.zw = sp.hi, lo.xy = keep_lo, lo.zw = sp.lo;
        *(h4*)dst = hi;
        if (NP == 3) *(h4*)(dst + kHPlane) = lo;
        if (NP == 8) {
          const uint2 lb = __builtin_bit_cast(uint2, lo);
          *(unsigned*)(hb + kHPlane + (mt * 32 + (lane & 31)) * kH8Ld + lo8_pos(wl * 32 + 4 * hh + (q >> 1) * 8)) =
              __builtin_amdgcn_perm(lb.y, lb.x, 0x07050301u);
        }
      }
    };
    auto init_bias = [&](f32x16 (&acc)[2][1], int ft) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bb = *(const float4*)(b1s + ft * 32 + 8 * g + 4 * hh);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc[mt][0][4 * g + 0] = bb.x;
          acc[mt][0][4 * g + 1] = bb.y;
          acc[mt][0][4 * g + 2] = bb.z;
          acc[mt][0][4 * g + 3] = bb.w;
        }
      }
    };
    // one slot: GEMM1 of chunk s into `cur` while the SiLU of chunk s-1 (held in `prev`) rides along
    auto slot = [&](int s, f32x16 (&cur)[2][1], f32x16 (&prev)[2][1]) {
      const int ft = (s < nchunk ? phys(s) : s) * 4 + wl;  // s >= nchunk: no GEMM1 (ft is out of range)
      const bool do_gemm = s < nchunk && ft < nft;
      const bool do_silu = s >= 1 && s - 1 < nchunk && phys(s - 1) * 4 + wl < nft;
      char* hb_prev = lds_h + ((s - 1) & 1) * 2 * kHPlane;
      h2 khi, klo;
      if (do_gemm) {
        init_bias(cur, ft);
        const uint4* w1_lane = w1p + (size_t)ft * (kD / 16) * 128 + lane;
        const char* a8_lane = smem + kAPlane + (lane & 31) * kA8Ld + hh * 32;
        if (do_silu) {
          auto side = [&](int st) { silu_pair(prev, hb_prev, st, khi, klo); };
          if constexpr (NP == 8)
            gemm_ring_f8<4, 1, true, kPF1, decltype(side), 5>(cur, a_lane, kALd, a8_lane, kA8Ld, w1f8_lane(ft), 0, r1, wg1, side);
          else
            gemm_ring<RNP, kD / 16, 1, true, kPF1, decltype(side), (NP == 3 ? 3 : 7)>(cur, a_lane, kALd, kAPlane, w1_lane,
                                                                                     0, r1, side);
        } else {
          if constexpr (NP == 8)
            gemm_ring_f8<4, 1, true, kPF1>(cur, a_lane, kALd, a8_lane, kA8Ld, w1f8_lane(ft), 0, r1, wg1);
          else
            gemm_ring<RNP, kD / 16, 1, true, kPF1>(cur, a_lane, kALd, kAPlane, w1_lane, 0, r1);
        }
        if (s + 1 < nchunk && phys(s + 1) * 4 + wl < nft) fill1(phys(s + 1) * 4 + wl);  // next chunk's W1 stream
      } else if (do_silu) {  // the last chunk's SiLU has no GEMM1 to hide under
#pragma unroll
        for (int st = 0; st < 16; ++st) silu_pair(prev, hb_prev, st, khi, klo);
      }
      TL_STAMP();  // producer: slot work done
      __syncthreads();
      TL_STAMP();  // producer: barrier passed
    };
    f32x16 accA[2][1], accB[2][1];
    for (int s = 0; s < nslots; s += 2) {
      slot(s, accA, accB);
      if (s + 1 < nslots) slot(s + 1, accB, accA);
    }
  } else {
    f32x16 (&acc2)[2][2] = acc2c;
    zero_acc(acc2);
    for (int s = 0; s < nslots; ++s) {
      if (s >= 2) {
        const int cl = s - 2, c = phys(cl);  // logical slot chunk (picks the H buffer) / physical hidden chunk
        const char* h_lane = lds_h + (cl & 1) * 2 * kHPlane + (lane & 31) * kHLd + hh * 16;
        const int ks2 = min(kFC / 16, ks2_total - c * (kFC / 16));
        const uint4* w2_lane = w2p + ((size_t)(2 * wl) * ks2_total + c * (kFC / 16)) * 128 + lane;
        if constexpr (NP == 8) {  // the launcher guarantees F % 128 == 0 for this stream
          const char* h8_lane = lds_h + (cl & 1) * 2 * kHPlane + kHPlane + (lane & 31) * kH8Ld + hh * 32;
          gemm_ring_f8<2, 2, false, kPF2>(acc2, h_lane, kHLd, h8_lane, kH8Ld, w2f8_lane(c), w2f8_nt, r2, wg2);
        } else if (ks2 == kFC / 16) {
          gemm_ring<RNP, kFC / 16, 2, false, kPF2>(acc2, h_lane, kHLd, kHPlane, w2_lane, w2_nt_stride, r2);
        } else {
          gemm_plain<RNP, 2, false>(acc2, h_lane, kHLd, kHPlane, w2_lane, w2_nt_stride, ks2);
        }
        if (cl + 1 < nchunk) fill2(phys(cl + 1));  // next chunk's W2 stream: in flight across the barrier
      }
      TL_STAMP();  // consumer: slot work done
      __syncthreads();
      TL_STAMP();  // consumer: barrier passed
    }
  }
  // residual rows of this wave: issued now, consumed after the tile exchange below
  float4 xr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = row0 + w * 8 + i;
    xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < M) xr[i] = ((const float4*)(x + (size_t)row * kD))[lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  // the consumers hold the [64, 256] result: stage it through the fp32 tile (aliases the A planes;
  // the last barrier of the loops guarantees no producer still reads them)
  if (!producer) acc_to_etile<2>(smem, acc2c, wl * 64, b2);
  __syncthreads();

  float4 g = make_float4(1.f, 1.f, 1.f, 1.f), bt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FINAL_LN) {
    g = ((const float4*)fin_g)[lane];
    bt = ((const float4*)fin_b)[lane];
  }
  {
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = xr[i];
      const float4 e = *(const float4*)(smem + (w * 8 + i) * kELd + lane * 16);
      v[i].x += res_scale * e.x;
      v[i].y += res_scale * e.y;
      v[i].z += res_scale * e.z;
      v[i].w += res_scale * e.w;
    }
    if (FINAL_LN) layer_norm_rows<8>(v, g, bt);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = row0 + w * 8 + i;
      if (row < M) ((float4*)(x + (size_t)row * kD))[lane] = v[i];
    }
  }
  TL_STAMP();  // last: epilogue done
}

#ifdef EEC_TIMELINE
extern "C" int eec_debug_timeline(unsigned long long* host_out, int n) {
  static unsigned long long* dev = nullptr;
  if (!dev) {
    if (hipMalloc(&dev, 8 * 2 * 64 * 8) != hipSuccess) return 1;
    (void)hipMemset(dev, 0, 8 * 2 * 64 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &dev, sizeof(dev));
    return 0;
  }
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(host_out, dev, (size_t)n * 8, hipMemcpyDeviceToHost);
}
#endif

template <int NP, bool FL, int ACT>
static hipError_t launch_ffn_t(const FfnArgs& a, hipStream_t st) {
  auto k = ffn_kernel<NP, FL, ACT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int grid = (a.M + kTileRows - 1) / kTileRows;
  hipLaunchKernelGGL(k, dim3(grid), dim3(kFfnThreads), kFfnLds, st, a.x, a.M, a.res_scale, a.ln_g, a.ln_b, a.w1p, a.b1, a.w2p,
                     a.b2, a.F, a.fin_g, a.fin_b, a.w1f8, a.w2f8);
  return hipGetLastError();
}

hipError_t launch_ffn(const FfnArgs& a, int np_in, hipStream_t st) {
  int np = np_in;
  const bool fl = a.fin_g != nullptr;
  if (np == 8 && (a.F % kFC != 0 || !a.w1f8 || !a.w2f8)) np = 3;  // the f8 stream needs whole 128-wide chunks
  if (np == 8) {
    if (a.relu) return fl ? launch_ffn_t<8, true, 1>(a, st) : launch_ffn_t<8, false, 1>(a, st);
    return fl ? launch_ffn_t<8, true, 0>(a, st) : launch_ffn_t<8, false, 0>(a, st);
  }
  if (a.relu) {
    if (np == 3) return fl ? launch_ffn_t<3, true, 1>(a, st) : launch_ffn_t<3, false, 1>(a, st);
    return fl ? launch_ffn_t<1, true, 1>(a, st) : launch_ffn_t<1, false, 1>(a, st);
  }
  if (np == 3) return fl ? launch_ffn_t<3, true, 0>(a, st) : launch_ffn_t<3, false, 0>(a, st);
  return fl ? launch_ffn_t<1, true, 0>(a, st) : launch_ffn_t<1, false, 0>(a, st);
}

}  // namespace eec
