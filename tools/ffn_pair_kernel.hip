// The pair-split feed-forward stage (d_model 256).  SURVEY 8a row a5 (torchaudio _FeedForwardModule inside ConformerLayer):
//
//     y = W2 . silu( W1 . LN(x) + b1 )            (b2, the 0.5 residual scale and the optional final LayerNorm are applied by
//                                                  whoever consumes the two partial sums: the next launch's prologue)
//
// Why this form ("pair kernel").  In the chain kernel (ffn.hip) a CU owns 64 rows and streams EVERY weight of
// the stage out of L2: in the exact mode (4 B per weight) that stream, at what an XCD's L2 delivers to 32 CUs reading the same
// bytes, is longer than the MFMA work it feeds.  Here TWO workgroups share a 128-row tile and each takes HALF of the hidden units:
//   * a CU streams half the weights for twice the rows -- L2 -> CU bytes per row are halved;
//   * the weights reach the CU ONCE, by LDS-DMA (`global_load_lds_dwordx4`: no VGPRs) into a ring of six 16-KiB entries that four
//     waves read -- every byte is used for 128 rows;
//   * the rows never touch LDS as planes: wave w < 4 ("G1") owns the 32 rows of row tile w and keeps their LayerNormed
//     activations in registers (the B operand of GEMM1: 128 VGPRs as fp16 hi + lo); its GEMM1 accumulator tile [32 hidden x 32
//     rows], SiLU'd and split in place, IS GEMM2's B operand (MFMA accumulator-as-operand) and is handed, 4 KiB per chunk, through
//     LDS to wave w + 4 ("G2", same SIMD), which keeps the [256 x 32] output accumulators (128 VGPRs) of the same rows.  Both
//     roles issue 24 MFMAs per half-slot; the SiLU / split VALU, the LDS-DMA issue (~80 cycles per piece for the issuing wave)
//     and the LDS reads of one wave run under the MFMAs of its SIMD partner (the four-wave form of this kernel, one wave per SIMD
//     doing everything, was measured first: nothing hides, profiles/r04_micro_ffn_pair.txt);
//   * one workgroup barrier per half-slot hands over ring entries and hidden chunks;
//   * the two halves' partial sums go to HBM (fp32, 2 x 128 KiB per tile, in the lane order of the accumulators so that stores
//     and loads are whole 1-KiB wave instructions) and are summed, in the fixed order P0 + P1, by the prologue of the launch that
//     needs the rows next.  No inter-workgroup synchronisation inside a launch.
//
// Weights come in the "pair format" (pack_ffn_pair_kernel): per 32-unit hidden chunk c one 64-KiB block
//   A(c) = 16 k-steps x {hi, lo} fragments of W1 rows [32c, 32c + 32)        (32 KiB)
//   B(c) =  8 n-tiles x 2 k-steps x {hi, lo} fragments of W2 columns [32c, 32c + 32)   (32 KiB)
// with the k-slots of every fragment in ACCUMULATOR order: slot j of lane half h of k-step s is k = 16 s + 8 (j >> 2) + 4 h + (j & 3)
// -- the order in which a 32 x 32 accumulator tile, converted in place, presents its rows to the next MFMA.  The same order is
// used for the LayerNormed rows, so a lane's 32 float4 pieces of its row are features [8 g + 4 h, + 4), g = 0 .. 31.
//
// Algorithmic work per launch: 2 * M * 2 * D * F flop (34.4 GFLOP at M = 16384); executed MFMA work 3 x that (hi / lo split, NP = 3).
#include "eec_kernels.h"

namespace eec {

// Pair-split feed-forward stage (this file; NOT part of libeec.so: measured against the chain kernel and not faster, profiles/r04_micro_ffn_pair.txt): 128-row tiles x two hidden halves, partial sums to HBM.
struct PairArgs {
  const float* x_in;     // [M][256] residual rows
  float* x_out;          // optional: the rows after the prologue (x_in + scale_in * (P0 + P1), final LayerNorm) are stored here
  float* tap;            // optional second copy of the same rows (exit tap)
  const float* part_in;  // optional: the previous stage's partial sums (two planes, pair layout: see ffn_pair.hip, pair_part_stride)
  const float* b2;       // [256] bias of THIS stage's second Linear: added into plane 0 of part_out (consumers sum the planes only)
  float scale_in;        // residual scale of the previous stage (0.5)
  const float *fin_g, *fin_b;  // optional LayerNorm applied to the rows before they are stored (layer-final)
  const float *ln_g, *ln_b;    // this stage's LayerNorm
  const uint4* wk;       // pair-format weights of this stage (pack_ffn_pair_kernel)
  const float* b1;       // [F], scaled like W1
  float* part_out;       // two planes (pair layout): this stage's partial sums W2[:, half] . silu(W1[half] . LN(x) + b1[half])
  int M, F;
  long long part_stride = 0;  // float4 elements between the two planes (set by launch_ffn_pair)
};
size_t pair_part_stride(int M);  // float4 elements per plane of a partial-sum buffer (rows rounded up to whole 128-row tiles)
hipError_t launch_ffn_pair(const PairArgs& a, hipStream_t st);
hipError_t launch_pack_ffn_pair(const float* w1, const float* w2, int F, uint4* out, float s1, float s2, hipStream_t st);


constexpr int kPairThreads = 512;
constexpr int kPairRows = 128;
constexpr int kPairEntry = 16 * 1024;                // bytes per ring entry (half of an A or B block)
constexpr int kPairSlots = 6;                        // ring entries: two being read, four in flight
constexpr int kPairRing = kPairSlots * kPairEntry;   // 98304
constexpr int kPairHBuf = 4 * 4 * 1024;              // one hidden chunk as GEMM2 operands: 4 row tiles x {t0 hi, t0 lo, t1 hi, t1 lo} x 1 KiB
constexpr int kPairHOff = kPairRing;                 // two such buffers
constexpr int kPairLnOff = kPairRing + 2 * kPairHBuf;  // 131072: LayerNorm parameters {fin_g, fin_b, ln_g, ln_b}, 1 KiB each
constexpr int kPairB1Off = kPairLnOff + 4096;          // 135168: this half's b1 (F / 2 floats)
constexpr int kPairChunkU4 = 4096;                   // uint4 per hidden chunk in the pair format (A then B)

// (128-row tile, hidden half) of this workgroup.  Same XCD affinity as row_tile_index(): inside each group of 32 blocks XCD x
// (= b % 8) gets the 256 rows [256 (8 g + x), + 256) -- the four 64-row tiles the other row-tile kernels give that XCD -- as two
// 128-row tiles x two halves, so a tile's partial sums, residual rows and conv halo are served by one L2.  Speed only.
__device__ __forceinline__ void pair_tile_index(int& tile, int& half) {
  const int b = blockIdx.x;
  const int full = (int)(gridDim.x & ~31u);
  if (b < full) {
    const int x = b & 7, i = b >> 3, g = i >> 2, r = i & 3;
    tile = 2 * (8 * g + x) + (r >> 1);
    half = r & 1;
  } else {
    tile = b >> 1;
    half = b & 1;
  }
}

// LayerNorm of a row held as 32 float4 pieces per lane (half a row; the other half is in lane ^ 32), in two parts: the
// statistics (v is left centred; returns 1 / sqrt(var + eps)) and the per-piece scale + shift.  The pieces are finished in
// groups of four with the scheduler fenced between groups: left alone, hipcc hoists all 64 gamma / beta loads of a row in
// front of the arithmetic (256 registers of loads in flight next to the 128 of the row: the row's operands end up in scratch).
__device__ __forceinline__ float pair_ln_stats(float4 (&v)[32]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  s += __shfl_xor(s, 32);
  const float mean = s * (1.0f / 256.0f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    v[i].x -= mean, v[i].y -= mean, v[i].z -= mean, v[i].w -= mean;
    sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  sq += __shfl_xor(sq, 32);
  return rsqrtf(sq * (1.0f / 256.0f) + kLnEps);
}
// (g, b: the parameter vectors staged in LDS -- a row's 64 parameter loads from global memory are latency and registers the
// prologue does not have)
__device__ __forceinline__ float4 pair_ln_piece(const float4& c, float rs, const float* g, const float* b, int i, int h) {
  const float4 gg = *(const float4*)(g + 8 * i + 4 * h), bb = *(const float4*)(b + 8 * i + 4 * h);
  return make_float4(c.x * rs * gg.x + bb.x, c.y * rs * gg.y + bb.y, c.z * rs * gg.z + bb.z, c.w * rs * gg.w + bb.w);
}
__device__ __forceinline__ void pair_layer_norm(float4 (&v)[32], const float* g, const float* b, int h) {
  const float rs = pair_ln_stats(v);
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = pair_ln_piece(v[i], rs, g, b, i, h);
}

// Half-slot hand-over: this wave's pieces of the two ring entries about to be read have landed (all but its 4 youngest LDS-DMA
// pieces -- those of the next half-slot -- are done; `drain`: nothing is issued behind them any more), every LDS access of the
// previous half-slot has completed, then the workgroup barrier: the entries and the hidden chunk written before it are visible
// to every wave, and the ring entries and hidden buffer read before it are free.
__device__ __forceinline__ void pair_handover(bool drain) {
  __builtin_amdgcn_sched_barrier(0);
#ifndef PAIR_ABL_NOBAR  // (timing-only build: no hand-over)
  if (!drain) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  __builtin_amdgcn_sched_barrier(0);
}
#ifdef PAIR_STAMPS  // diagnostic build (tools/ffn_pair_bench): s_memtime / s_memrealtime of wave 0 of every 16th block
__device__ unsigned long long g_pair_stamps[16 * 8];
#define PAIR_STAMP(i)                                                                                  \
  do {                                                                                                 \
    if ((threadIdx.x & 511) == 0 && (blockIdx.x & 15) == 0 && blockIdx.x < 256) {                      \
      g_pair_stamps[(blockIdx.x >> 4) * 8 + 2 * (i)] = __builtin_amdgcn_s_memtime();                   \
      g_pair_stamps[(blockIdx.x >> 4) * 8 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();           \
    }                                                                                                  \
  } while (0)
#else
#define PAIR_STAMP(i)
#endif
// hi / lo split of two fp32 values as split2<3> makes it (hi = rtz(x), lo = rtz(x - hi)) in 4 instructions instead of 6:
// v_fma_mix_f32 reads the fp16 halves of `hi` directly (x - hi is exact in fp32 either way: identical results).
__device__ __forceinline__ hl2_t pair_split2(float a, float b) {
  hl2_t r;
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
  float la, lb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
  r.hi = __builtin_bit_cast(h2, hi);
  r.lo = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(la, lb));
  return r;
}
template <bool B>
struct PairTag {
  static constexpr bool value = B;
};

// HAS_PART: the rows are x_in + scale_in * (part_in[0] + part_in[1]) (the previous stage's partial sums, its bias included);
// fin_g != null: a LayerNorm on that (layer-final); x_out / tap != null: the rows are stored (by ONE of the two halves).
//
// Schedule.  Slot s = 0 .. C+1 (C = F / 64 hidden chunks of this half), two half-slots hs each, one barrier per half-slot:
//   G1 (waves 0-3)  slot s < C : GEMM1 of chunk s, k-steps [8 hs, 8 hs + 8), from ring entry A(s).hs; in its shadow the SiLU +
//                                split of values [8 hs, 8 hs + 8) of chunk s-1 -> GEMM2 operand k-step hs -> hidden buffer (s-1) & 1
//                   slot s = C : the SiLU of chunk C-1 alone
//   G2 (waves 4-7)  slot s >= 2: GEMM2 of chunk s-2, n-tiles [4 hs, 4 hs + 4), from ring entry B(s-2).hs; its operands are read
//                                from hidden buffer s & 1 at hs = 0
// Ring: half-slot n = 2 s + hs reads entries 2n (A part) and 2n + 1 (B part) in ring positions 2n % 6, (2n + 1) % 6; right after
// its barrier every wave issues its 2 + 2 pieces of half-slot n + 2's entries into the positions half-slot n - 1 has just freed
// (where a part does not exist -- B in slots 0, 1; A in slots C, C+1 -- a dummy block is fetched so that the piece count a
// `vmcnt` wait relies on never changes).
template <bool HAS_PART>
__global__ __launch_bounds__(kPairThreads, 2) void ffn_pair_kernel(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rt = w & 3;  // row tile of this wave (G1 and G2 alike)
  int tile, half;
  pair_tile_index(tile, half);
  const int M = a.M, C = a.F / 64;  // hidden chunks of this half
  const int NH = 2 * (C + 2);       // half-slots
  const uint4* wsrc = a.wk + (size_t)half * C * kPairChunkU4 + lane;
  float* b1s = (float*)(smem + kPairB1Off);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // This wave's 2 + 2 pieces of the entries of half-slot n.  The LDS-DMA instructions are issued from inline assembly: hipcc
  // treats a `global_load_lds` it knows about as a pending LDS store that every later ds_read may alias and puts
  // `s_waitcnt vmcnt(0)` in front of the next LDS read -- the ring would drain at every step.  Hidden in asm, the only waits on
  // the ring are the counted ones of pair_handover().  m0 (LDS base of a piece) is saved and restored around the group.
  auto issue = [&](int n) {
    if (n >= NH) return;
#ifdef PAIR_ABL_NODMA  // timing-only build: no weight stream at all (the ring holds garbage)
    return;
#endif
    const int s = n >> 1, hs = n & 1;
    const size_t offA = s < C ? (size_t)s * kPairChunkU4 + hs * 1024 : 0;                    // uint4 units (16 KiB = 1024 uint4)
    const size_t offB = (s >= 2 && s < C + 2) ? (size_t)(s - 2) * kPairChunkU4 + 2048 + hs * 1024 : 0;
    const uint4* srcA = wsrc + offA + (size_t)(2 * w) * 64;
    const uint4* srcB = wsrc + offB + (size_t)(2 * w) * 64;
    const unsigned dstA = lds_base + (unsigned)(((2 * n) % kPairSlots) * kPairEntry + 2 * w * 1024);
    const unsigned dstB = lds_base + (unsigned)(((2 * n + 1) % kPairSlots) * kPairEntry + 2 * w * 1024);
    unsigned m0_save;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(m0_save)
        : "v"(srcA), "v"(srcA + 64), "v"(srcB), "v"(srcB + 64), "s"(dstA), "s"(dstB)
        : "memory", "scc");
  };

  PAIR_STAMP(0);
  issue(0);
  issue(1);
  float* lnp = (float*)(smem + kPairLnOff);  // {fin_g, fin_b, ln_g, ln_b} x 256
  const size_t prow = ((size_t)tile * 4 + rt) * 32 * 64 + lane;  // float4 index of this lane's piece 0 in a partial-sum plane

#if defined(PAIR_ONLY_ROLE)  // register-budget diagnostics: compile one role only (never run such a build)
  if (PAIR_ONLY_ROLE == 0) {
#else
  if (w < 4) {
#endif
    // =================================== G1: rows -> LayerNorm -> GEMM1 -> SiLU -> hidden operands ===================================
    const int row = tile * kPairRows + rt * 32 + r;
    const bool ok = row < M;
    const int row_c = ok ? row : M - 1;  // rows beyond M: load a valid row unconditionally (a per-element "load or zero" select makes
                                         // hipcc branch around every load and wait for each one), never store it
    float4 v[32];
    {
      const float* xr = a.x_in + (size_t)row_c * 256 + 4 * h;
#pragma unroll
      for (int g = 0; g < 32; ++g) v[g] = *(const float4*)(xr + 8 * g);
    }
    if constexpr (HAS_PART) {
      // the two partial-sum planes, in batches of four pieces with the next batch's loads issued before this one is added
      // (the row is 128 registers: everything in flight at once does not fit beside it)
      const float4* p0 = (const float4*)a.part_in + prow;
      const float4* p1 = p0 + (size_t)a.part_stride;
      const float sc = a.scale_in;
      float4 q0[2][4], q1[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) q0[0][i] = p0[i * 64], q1[0][i] = p1[i * 64];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b + 1 < 8) {
#pragma unroll
          for (int i = 0; i < 4; ++i) q0[(b + 1) & 1][i] = p0[(4 * (b + 1) + i) * 64], q1[(b + 1) & 1][i] = p1[(4 * (b + 1) + i) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4& t = v[4 * b + i];
          const float4 u0 = q0[b & 1][i], u1 = q1[b & 1][i];
          t.x += sc * (u0.x + u1.x), t.y += sc * (u0.y + u1.y), t.z += sc * (u0.z + u1.z), t.w += sc * (u0.w + u1.w);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the LayerNorm parameters and b1 are in LDS once the G2 waves have passed this barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (a.fin_g) pair_layer_norm(v, lnp, lnp + 256, h);
    if (ok && (rt & 1) == half) {  // both halves hold identical rows: each stores two of the tile's four 32-row groups
      if (a.x_out) {
        float* xo = a.x_out + (size_t)row * 256 + 4 * h;
#pragma unroll
        for (int g = 0; g < 32; ++g) *(float4*)(xo + 8 * g) = v[g];
      }
      if (a.tap) {
        float* xo = a.tap + (size_t)row * 256 + 4 * h;
#pragma unroll
        for (int g = 0; g < 32; ++g) *(float4*)(xo + 8 * g) = v[g];
      }
    }
    // this stage's LayerNorm, finished piece by piece straight into the B operands of GEMM1: k-step s = pieces 2 s (slots
    // 0..3) and 2 s + 1 (slots 4..7); the fp32 pieces die as their operands are made
    h8 xh[16], xl[16];
    {
      const float rs = pair_ln_stats(v);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float4 u0 = pair_ln_piece(v[2 * s], rs, lnp + 512, lnp + 768, 2 * s, h), u1 = pair_ln_piece(v[2 * s + 1], rs, lnp + 512, lnp + 768, 2 * s + 1, h);
        const hl2_t p0 = pair_split2(u0.x, u0.y), p1 = pair_split2(u0.z, u0.w), p2 = pair_split2(u1.x, u1.y), p3 = pair_split2(u1.z, u1.w);
        xh[s] = (h8){p0.hi[0], p0.hi[1], p1.hi[0], p1.hi[1], p2.hi[0], p2.hi[1], p3.hi[0], p3.hi[1]};
        xl[s] = (h8){p0.lo[0], p0.lo[1], p1.lo[0], p1.lo[1], p2.lo[0], p2.lo[1], p3.lo[0], p3.lo[1]};
      }
    }
    PAIR_STAMP(1);

    // values [8 hs, 8 hs + 8) of a finished tile -> SiLU -> hi / lo -> GEMM2 operand k-step hs of hidden buffer `hb`
    // (lane-linear 1-KiB blocks [rt][t][plane])
    auto store_h = [&](char* hb, int hs, const h8& hi, const h8& lo) {
      char* dst = hb + (rt * 4 + hs * 2) * 1024 + lane * 16;
      *(h8*)dst = hi;
      *(h8*)(dst + 1024) = lo;
    };
    // one half-slot of GEMM1: k-steps [8 hs, 8 hs + 8) of chunk c into acc; SIDE: the SiLU of `prev` rides along
    auto gemm1_half = [&](int c, int hs, int pos, f32x16& acc, const f32x16& prev, char* hb, auto hs_tag, auto side_tag) {
      constexpr bool SIDE = decltype(side_tag)::value;
      constexpr int HS = decltype(hs_tag)::value;
      (void)hs;
      const char* base = smem + pos * kPairEntry + lane * 16;
      if constexpr (HS == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bb = *(const float4*)(b1s + 32 * c + 8 * g + 4 * h);
          acc[4 * g + 0] = bb.x, acc[4 * g + 1] = bb.y, acc[4 * g + 2] = bb.z, acc[4 * g + 3] = bb.w;
        }
      }
      h8 wh[3], wl[3];  // fragment registers: two k-steps ahead of the MFMAs
      wh[0] = *(const h8*)(base);
      wl[0] = *(const h8*)(base + 1024);
      wh[1] = *(const h8*)(base + 2048);
      wl[1] = *(const h8*)(base + 2048 + 1024);
      h8 Hh, Hl;
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int s = 8 * HS + i, cur = i % 3, nxt = (i + 2) % 3;
        if (i + 2 < 8) {
          wh[nxt] = *(const h8*)(base + (i + 2) * 2048);
          wl[nxt] = *(const h8*)(base + (i + 2) * 2048 + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = mfma16(wh[cur], xl[s], acc);
        acc = mfma16(wl[cur], xh[s], acc);
        acc = mfma16(wh[cur], xh[s], acc);
        if constexpr (SIDE) {
#ifdef PAIR_ABL_NOSILU  // timing-only build: the hidden chunk is passed on without activation work
          if (i & 1) Hh[i - 1] = (half_t)prev[s - 1], Hh[i] = (half_t)prev[s], Hl[i - 1] = (half_t)0.f, Hl[i] = (half_t)0.f;
#else
          const float f = silu_exp2(prev[s]);
          if ((i & 1) == 0) {
            keep = f;
          } else {
            const hl2_t sp = pair_split2(keep, f);
            Hh[i - 1] = sp.hi[0], Hh[i] = sp.hi[1];
            Hl[i - 1] = sp.lo[0], Hl[i] = sp.lo[1];
          }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SIDE) store_h(hb, HS, Hh, Hl);
    };
    auto silu_half = [&](const f32x16& prev, char* hb, auto hs_tag) {
      constexpr int HS = decltype(hs_tag)::value;
      h8 Hh, Hl;
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const hl2_t sp = pair_split2(silu_exp2(prev[8 * HS + i]), silu_exp2(prev[8 * HS + i + 1]));
        Hh[i] = sp.hi[0], Hh[i + 1] = sp.hi[1];
        Hl[i] = sp.lo[0], Hl[i + 1] = sp.lo[1];
      }
      store_h(hb, HS, Hh, Hl);
    };
    f32x16 accA, accB;
    // one slot of G1 (chunk s into `cur`, SiLU of chunk s-1 from `prev`)
    auto slot = [&](int s, f32x16& cur, f32x16& prev) {
      char* hb = smem + kPairHOff + ((s - 1) & 1) * kPairHBuf;
      static_range<0, 2>([&](auto hs_tag) {
        constexpr int HS = decltype(hs_tag)::value;
        const int n = 2 * s + HS;
        pair_handover(n + 1 >= NH);
        issue(n + 2);
        if (s == 0) gemm1_half(s, HS, (2 * n) % kPairSlots, cur, prev, hb, hs_tag, PairTag<false>{});
        else if (s < C) gemm1_half(s, HS, (2 * n) % kPairSlots, cur, prev, hb, hs_tag, PairTag<true>{});
        else if (s == C) silu_half(prev, hb, hs_tag);
      });
    };
    for (int s = 0; s < C + 2; s += 2) {  // C is even (launch_ffn_pair): accA holds the even chunks, accB the odd ones
      slot(s, accA, accB);
      slot(s + 1, accB, accA);
    }
    PAIR_STAMP(2);
  } else {
    // =================================== G2: hidden operands -> GEMM2 -> partial sums ===================================
    // while the G1 waves fetch their rows: stage b1 and the LayerNorm parameters in LDS
    {
      const int t = threadIdx.x - 256;
      for (int i = t; i < a.F / 8; i += 256) ((float4*)b1s)[i] = ((const float4*)(a.b1 + (size_t)half * (a.F / 2)))[i];
      if (t < 64) {
        if (a.fin_g) ((float4*)lnp)[t] = ((const float4*)a.fin_g)[t], ((float4*)lnp)[64 + t] = ((const float4*)a.fin_b)[t];
        ((float4*)lnp)[128 + t] = ((const float4*)a.ln_g)[t], ((float4*)lnp)[192 + t] = ((const float4*)a.ln_b)[t];
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    PAIR_STAMP(1);
    f32x16 acc2[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[nt][i] = 0.f;
    h8 Hh[2], Hl[2];
    for (int s = 0; s < C + 2; ++s) {
      static_range<0, 2>([&](auto hs_tag) {
        constexpr int HS = decltype(hs_tag)::value;
        const int n = 2 * s + HS;
        pair_handover(n + 1 >= NH);
        issue(n + 2);
        if (s >= 2) {
          if constexpr (HS == 0) {
            const char* hb = smem + kPairHOff + (s & 1) * kPairHBuf + rt * 4 * 1024 + lane * 16;
            Hh[0] = *(const h8*)(hb);
            Hl[0] = *(const h8*)(hb + 1024);
            Hh[1] = *(const h8*)(hb + 2048);
            Hl[1] = *(const h8*)(hb + 3072);
          }
          const char* base = smem + ((2 * n + 1) % kPairSlots) * kPairEntry + lane * 16;
          h8 wh[3], wl[3];
          wh[0] = *(const h8*)(base);
          wl[0] = *(const h8*)(base + 1024);
          wh[1] = *(const h8*)(base + 2048);
          wl[1] = *(const h8*)(base + 2048 + 1024);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int nt = 4 * HS + (i >> 1), t = i & 1, cur = i % 3, nxt = (i + 2) % 3;
            if (i + 2 < 8) {
              wh[nxt] = *(const h8*)(base + (i + 2) * 2048);
              wl[nxt] = *(const h8*)(base + (i + 2) * 2048 + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc2[nt] = mfma16(wh[cur], Hl[t], acc2[nt]);
            acc2[nt] = mfma16(wl[cur], Hh[t], acc2[nt]);
            acc2[nt] = mfma16(wh[cur], Hh[t], acc2[nt]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      });
    }
    PAIR_STAMP(2);
    // ---- epilogue: this half's partial sums in accumulator lane order: piece g = 4 nt + gg of lane (r, h) = features 8 g + 4 h .. ----
    // (plane 0 carries the stage's bias b2, so that consumers only sum the two planes)
    float4* po = (float4*)a.part_out + (size_t)half * a.part_stride + prow;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o = make_float4(acc2[nt][4 * g], acc2[nt][4 * g + 1], acc2[nt][4 * g + 2], acc2[nt][4 * g + 3]);
        if (half == 0) {
          const float4 bb = *(const float4*)(a.b2 + 32 * nt + 8 * g + 4 * h);
          o.x += bb.x, o.y += bb.y, o.z += bb.z, o.w += bb.w;
        }
        po[(4 * nt + g) * 64] = o;
      }
  }
  PAIR_STAMP(3);
}

// float4 elements between the two halves' planes of a partial-sum buffer for M rows (whole 128-row tiles)
size_t pair_part_stride(int M) { return (size_t)((M + kPairRows - 1) / kPairRows) * kPairRows * 64; }

hipError_t launch_ffn_pair(const PairArgs& a_in, hipStream_t st) {
  PairArgs a = a_in;
  if (a.F % 128 || a.F < 256 || a.M <= 0) return hipErrorInvalidValue;  // an even number (>= 2) of 32-unit chunks per half
  a.part_stride = (long long)pair_part_stride(a.M);
  const int lds = kPairB1Off + (a.F / 2) * 4;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int tiles = (a.M + kPairRows - 1) / kPairRows;
  if (a.part_in) {
    auto k = ffn_pair_kernel<true>;
    if (hipError_t e = ensure_max_lds((const void*)k, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(2 * tiles), dim3(kPairThreads), lds, st, a);
  } else {
    auto k = ffn_pair_kernel<false>;
    if (hipError_t e = ensure_max_lds((const void*)k, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(2 * tiles), dim3(kPairThreads), lds, st, a);
  }
  return hipGetLastError();
}

// W1[F][256] (scaled by s1), W2[256][F] (scaled by s2) -> pair format (see the file comment).
__global__ void pack_ffn_pair_kernel(const float* __restrict__ w1, const float* __restrict__ w2, int F, uint4* __restrict__ out,
                                     float s1, float s2) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (chunk, fragment pair, lane)
  const int total = (F / 32) * 32 * 64;
  if (idx >= total) return;
  const int lane = idx & 63, f = (idx >> 6) & 31, c = idx >> 11, r = lane & 31, h = lane >> 5;
  h8 hi, lo;
  uint4* o = out + (size_t)c * kPairChunkU4;
  if (f < 16) {  // A: k-step f of W1 rows [32c, 32c+32)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * f + 8 * (j >> 2) + 4 * h + (j & 3);
      EEC_SPLIT(w1[(size_t)(32 * c + r) * 256 + k] * s1, hi, lo, j);
    }
    o[(f * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    o[(f * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
  } else {  // B: (n-tile nt, k-step t) of W2 columns [32c, 32c+32)
    const int i = f - 16, nt = i >> 1, t = i & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int hid = 32 * c + 16 * t + 8 * (j >> 2) + 4 * h + (j & 3);
      EEC_SPLIT(w2[(size_t)(32 * nt + r) * F + hid] * s2, hi, lo, j);
    }
    o[2048 + (i * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    o[2048 + (i * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
  }
}

hipError_t launch_pack_ffn_pair(const float* w1, const float* w2, int F, uint4* out, float s1, float s2, hipStream_t st) {
  if (F % 32) return hipErrorInvalidValue;
  const int total = (F / 32) * 32 * 64;
  hipLaunchKernelGGL(pack_ffn_pair_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w1, w2, F, out, s1, s2);
  return hipGetLastError();
}

}  // namespace eec
