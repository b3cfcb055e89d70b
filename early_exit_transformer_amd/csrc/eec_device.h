// Shared device-side building blocks for the gfx950 (CDNA4, wave64) encoder kernels.
//
// Conventions used by every kernel in this directory
// ---------------------------------------------------
// * A "row tile" is Geo<D>::kRows consecutive rows of the flattened (utterance x frame)
//   axis M = B*T': 64 rows at d_model 256, 32 rows at d_model 512 (the tile's activation
//   planes are 64 KiB of LDS either way).  One 512-thread workgroup (8 waves, two per SIMD)
//   owns one row tile and all of the output columns of it; at the headline shape (D = 256)
//   M/64 = 256 tiles = one workgroup per CU.
// * 16-bit operands are fp16.  "NP" = number of MFMA passes per product:
//     NP=1  a_hi*w_hi                                   (plain fp16 operands)
//     NP=3  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi           (hi/lo split, ~2^-21)
//   with x = hi + lo, hi = fp16(x), lo = fp16(x - hi).  The reference computes
//   in fp32; single-pass 16-bit operands miss its 1e-3 log-prob
//   tolerance by 2.7x (fp16) / 25x (bf16), the split meets it with margin.
// * MFMA shape: v_mfma_f32_32x32x16_f16.  Lane l: r = l & 31, h = l >> 5.
//     A operand: lane holds A[row r][k = 8h + j], j = 0..7
//     B operand: lane holds B[k = 8h + j][col r]
//     C/D      : col = r, row = (reg & 3) + 8 * (reg >> 2) + 4 * h, reg = 0..15
//   Both operand fragments of an activation tile Act[m][k] and of a weight
//   matrix W[n][k] (torch Linear layout, k contiguous) are therefore "row r,
//   8 consecutive k starting at 16*s + 8*h": the same 16-byte read.  Passing
//   (act, w) gives acc[m][n] ("normal"); passing (w, act) gives acc[n][m]
//   ("swapped": frames on lanes, output features in registers).
// * Packed weights: fragment (nt, s) of W[N][K] is the 32 rows [32nt, 32nt+32)
//   x 16 k [16s, 16s+16) stored lane-linear: uint4 index
//       ((nt * KS + s) * 2 + plane) * 64 + lane,   KS = K / 16, plane 0 = hi, 1 = lo
//   so a wave reads one fragment as one contiguous 1 KiB global_load_dwordx4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eec {

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// EEC_OPERAND_BF16 (a translation-unit switch; only the training step's fused feed-forward BACKWARD sets it, ffn.hip): the split
// operands are bf16 hi / lo pairs (2^-16 per product, the fp32 exponent range: gradients) on v_mfma_*_bf16 instead of fp16 pairs.
// Fragments keep their h8 / h2 storage types -- only the conversions and the MFMA builtins differ.
#ifndef EEC_OPERAND_BF16
#define EEC_OPERAND_BF16 0
#endif
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Streaming-store experiments (profiles/r04_ab_nt_stores.txt): which of the large write-once outputs of the inference kernels carry the
// non-temporal hint, so that they do not push the weight streams out of the L2.  NT = a compile-time flag per site.
template <bool NT, typename T>
__device__ __forceinline__ void store_maybe_nt(T* p, T v) {
#ifdef EEC_ABLATE_QKV_STORES  // timing-only build: what the Q / K / V / GLU / head stores cost (the value stays live, nothing is written)
  asm volatile("" ::"v"(v), "v"(p));
  return;
#endif
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
#ifndef EEC_NT_QKV
#define EEC_NT_QKV 0
#endif
#ifndef EEC_NT_HEAD
#define EEC_NT_HEAD 0
#endif
#ifndef EEC_NT_G
#define EEC_NT_G 0
#endif

constexpr int kWave = 64;
constexpr float kLnEps = 1e-5f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kHalfMax = 65504.0f;

// Tile geometry by d_model.  D = 256: 64-row tiles, a wave owns ONE 32-wide column tile of a D-wide output and two
// 32-row tiles (acc[2][1]); D = 512: 32-row tiles, a wave owns TWO adjacent column tiles and one row tile (acc[1][2]).
// LDS geometry in bytes; row pads of one 16-B slot make ds_read_b128 of "32 different rows, same column"
// conflict-free (stride = 4 banks mod 64).
// EEC_X_HI8 (experiment knob, off): the activations' e5m2 hi bytes (the a_hi8 operand of the weight-residual correction product)
// are written ONCE by whoever writes the planes and read as MFMA operands straight from LDS, instead of being re-made with two
// v_perm per fragment and k-step by every wave that multiplies the tile (chain kernel producers: -64 VALU per slot and wave).
// Bit-identical results; same-box A/B: 171.6 us with, 171.0 us without (profiles/r03_ab_chain_knobs.txt) -- the producers' VALU
// issue is not what bounds the slot.
#ifndef EEC_X_HI8
#define EEC_X_HI8 0
#endif
template <int D>
struct Geo {
  static_assert(D == 256 || D == 512, "d_model must be 256 or 512");
  static constexpr int kRows = 16384 / D;       // rows of M per workgroup: 64 / 32
  static constexpr int kMT = kRows / 32;        // 32-row MFMA tiles per workgroup: 2 / 1
  static constexpr int kNW = D / 256;           // 32-wide column tiles of a D-wide output per wave (8 waves): 1 / 2
  static constexpr int kKS = D / 16;            // k-steps of a K = D product
  static constexpr int kQ = D / 256;            // float4 pieces of a row per lane (piece q = columns 256 q + 4 lane ..)
  static constexpr int kRPW = kRows / 8;        // rows per wave in a row pass: 8 / 4
  static constexpr int kALd = (D + 8) * 2;      // 528 / 1040 : [rows][D] fp16 activation plane
  static constexpr int kAPlane = kRows * kALd;  // 33792 / 33280
  // NP == 8 byte plane, per row: [D e5m2 residual bytes][D e5m2 top bytes of the hi halves (EEC_X_HI8)][16 pad], both in the
  // permuted MX slot order (lo8_pos).  With the hi bytes the row stride equals the fp16 plane's (conflict-free ds_read_b128) and the
  // plane fills exactly the region the fp16 lo plane of the split format has: no extra LDS.
  static constexpr int kA8Ld = EEC_X_HI8 ? 2 * D + 16 : D + 16;  // 528 / 1040 (272 / 528 without the hi bytes)
  static constexpr int kA8Hi = D;               // byte offset of the hi8 half inside a row
  static constexpr int kELd = (D + 4) * 4;      // 1040 / 2064: [rows][D] fp32 exchange tile
  static constexpr int kETile = kRows * kELd;   // 66560 / 66048
};

// Row tile of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an XCD and its
// L2); consecutive row tiles -- the four tiles of an utterance at T' = 256 -- exchange data between launches (K / V of the
// utterance, the +-15-frame halo of the depthwise conv) and every tile re-reads its own residual rows in the next launch.
// EEC_XCD_TILES: inside each group of 32 blocks, XCD x (= b % 8) gets the four CONSECUTIVE tiles 4x .. 4x+3, and the same
// map is used by every row-tile kernel, so that traffic is served by the XCD's own L2 instead of the fabric.  Speed only:
// correctness does not depend on the placement (kernel boundaries order all global traffic).
#ifndef EEC_XCD_TILES
#define EEC_XCD_TILES 1
#endif
__device__ __forceinline__ int row_tile_index() {
  const int b = blockIdx.x;
#if EEC_XCD_TILES
  const int full = (int)(gridDim.x & ~31u);  // blocks in whole groups of 32; a ragged tail keeps the identity map
  if (b < full) return (b & ~31) + (b & 7) * 4 + ((b >> 3) & 3);
#endif
  return b;
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
// Lane index recomputed from nothing (v_mbcnt of the full mask) behind an optimisation barrier: code after a long
// register-starved loop derives its lane / row indices from THIS value, so the compiler recomputes them (a few
// VALU ops) instead of keeping what it computed before the loop alive -- i.e. spilled to scratch and reloaded
// (from DRAM: the weight stream has evicted the scratch lines) at the loop exit.  The wave index for such code
// comes from an SGPR copy made at kernel entry (wave_id_sgpr()).
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ int wave_id_sgpr() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Wave-wide reductions on the DPP crossbar (no LDS traffic, unlike __shfl_xor = ds_bpermute): xor-1 / xor-2
// inside each quad, half-row and row mirrors -> every lane of a 16-lane row holds the row total; row_bcast15
// into rows 1,3 and row_bcast31 into rows 2,3 -> lane 63 holds the wave total, returned wave-uniform.
#define EEC_DPP_ADD(v, ctrl, rmask) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, rmask, 0xf, false)))
#define EEC_DPP_MAX(v, ctrl, rmask) \
  fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), ctrl, rmask, 0xf, false)))
__device__ __forceinline__ float wave_sum(float v) {
  v = EEC_DPP_ADD(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
  v = EEC_DPP_ADD(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
  v = EEC_DPP_ADD(v, 0x141, 0xf);  // row_half_mirror
  v = EEC_DPP_ADD(v, 0x140, 0xf);  // row_mirror
  v = EEC_DPP_ADD(v, 0x142, 0xa);  // row_bcast15 -> rows 1, 3
  v = EEC_DPP_ADD(v, 0x143, 0xc);  // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = EEC_DPP_MAX(v, 0xB1, 0xf);
  v = EEC_DPP_MAX(v, 0x4E, 0xf);
  v = EEC_DPP_MAX(v, 0x141, 0xf);
  v = EEC_DPP_MAX(v, 0x140, 0xf);
  v = EEC_DPP_MAX(v, 0x142, 0xa);
  v = EEC_DPP_MAX(v, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ half_t to_half_sat(float x) {
  return (half_t)fminf(fmaxf(x, -kHalfMax), kHalfMax);
}
// hi/lo split of one fp32 value.
struct hl_t {
  half_t hi, lo;
};
__device__ __forceinline__ hl_t split_hl(float x) {
  hl_t r;
  r.hi = to_half_sat(x);
  r.lo = (half_t)(x - (float)r.hi);
  return r;
}
// vector elements cannot bind to references: assign through a temporary
#define EEC_SPLIT(val, HI, LO, idx)       \
  do {                                    \
    const hl_t _t = split_hl(val);        \
    (HI)[idx] = _t.hi;                    \
    (LO)[idx] = _t.lo;                    \
  } while (0)

// packed hi/lo split of two values (6 VALU per pair in the split form).
struct hl2_t {
  h2 hi, lo;
};
// NP == 3: hi = rtz(x) (never overflows to inf), lo = rtz(x - hi): x = hi + lo to ~2^-21.
// NP == 1: there is no lo plane, so hi must be round-to-nearest (truncation doubles the error).
template <int NP>
__device__ __forceinline__ hl2_t split2(float a, float b) {
  hl2_t r;
#if EEC_OPERAND_BF16
  {
    const f32x2_t x = {a, b};
    const bf2_t hi = __builtin_convertvector(x, bf2_t);
    r.hi = __builtin_bit_cast(h2, hi);
    r.lo = NP == 3 ? __builtin_bit_cast(h2, __builtin_convertvector(x - __builtin_convertvector(hi, f32x2_t), bf2_t)) : r.hi;
    return r;
  }
#endif
  if (NP == 3) {
    r.hi = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
    r.lo = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a - (float)r.hi[0], b - (float)r.hi[1]));
  } else {
    r.hi[0] = to_half_sat(a);
    r.hi[1] = to_half_sat(b);
    r.lo = r.hi;
  }
  return r;
}
// The same split as split2<3> (hi = rtz(x), lo = rtz(x - hi)) in 4 instructions instead of 6: v_fma_mix_f32 reads the fp16 halves
// of `hi` directly (x - hi is exact in fp32 either way: identical results).
__device__ __forceinline__ hl2_t split2_mix(float a, float b) {
  hl2_t r;
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
  float la, lb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
  r.hi = __builtin_bit_cast(h2, hi);
  r.lo = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(la, lb));
  return r;
}
// SiLU in the exp2 domain: u = log2(e) * x  ->  u / (1 + 2^-u) = log2(e) * silu(x).
// (log2(e) is folded into W1/b1 and 1/log2(e) into W2 at pack time.)  4 VALU, 2 transcendental.
__device__ __forceinline__ float silu_exp2(float u) {
  return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u));
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * x)); }

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
#if EEC_OPERAND_BF16
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
// row of accumulator register i for this lane (within a 32x32 tile)
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------
// Register ring of weight fragments: the first PF k-steps of a stage are loaded by the caller
// one stage AHEAD (so no stage starts with an empty pipeline), the stage refills it PF ahead.
// w_lane points at fragment (nt0, s0), plane 0, this lane; consecutive k-steps are 128 uint4
// apart, consecutive n-tiles nt_stride uint4 apart.
// ---------------------------------------------------------------------------
#ifdef EEC_KSTEP_STAMPS
__device__ void eec_kstep_stamp();
#endif
// weight-stream load (experiment knob EEC_W_NT: non-temporal = L1-bypassing loads for the streamed fragments)
#ifndef EEC_W_NT
#define EEC_W_NT 0
#endif
__device__ __forceinline__ uint4 wload(const uint4* p) {
#if EEC_W_NT
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(uint4, __builtin_nontemporal_load((const u32x4_t*)p));
#else
  return *p;
#endif
}
template <int NP, int PF, int NT = 1>
struct WRing {
  uint4 q[PF][NT][(NP == 3) ? 2 : 1];
};

template <int NP, int PF, int NT>
__device__ __forceinline__ void ring_fill_32(WRing<NP, PF, NT>& r, const uint4* __restrict__ w_lane, size_t nt_stride,
                                          int steps_avail) {
#ifdef EEC_ABLATE_W
  if (threadIdx.x > 100000)  // timing-only build: no weight loads at all
#endif
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < steps_avail) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        r.q[p][nt][0] = w_lane[nt * nt_stride + (size_t)p * 128];
#ifdef EEC_X3_LO_SKIP  // timing-only build: the lo fragments are not loaded (what a 2 B / weight stream would cost)
        if (NP == 3) r.q[p][nt][(NP == 3) ? 1 : 0] = r.q[p][nt][0];
#else
        if (NP == 3) r.q[p][nt][(NP == 3) ? 1 : 0] = w_lane[nt * nt_stride + (size_t)p * 128 + 64];
#endif
      }
    }
  __builtin_amdgcn_sched_barrier(0);  // keep these loads HERE: one stage ahead of their consumer
}

// acc[2][NT] += Act[64 rows][16*KS] (LDS planes) x Wfrag stream; MT = 2 row tiles, NT 32-wide n-tiles.
// Software pipeline, pinned with sched_barrier(0) at every k-step boundary (hipcc otherwise sinks
// the prefetch loads down to their first use and every step pays a full L2 round trip):
//   step s:  ds_read A(s+1)  |  MFMAs of step s from ring slot s%PF  |  global_load W(s+PF) -> slot s%PF
struct NoSide {
  __device__ __forceinline__ void operator()(int) const {}
};
// `side(s)` is called once per k-step (s is a constant after unrolling): VALU work that is issued
// in the shadow of that step's MFMAs (e.g. the SiLU of the previous chunk).
// SIDE_VALU > 0: after side(s), pin the order "1 MFMA, SIDE_VALU VALU" for the step, so the side
// work's dependent VALU chain is spread over the MFMA issue gaps (an in-order wave otherwise runs it
// as one serial chain after the MFMAs, ~150-250 exposed cycles per step).
template <int NP, int KS, int NT, bool SWAP, int PF, typename Side = NoSide, int SIDE_VALU = 0, int MT = 2>
__device__ __forceinline__ void gemm_ring_32(f32x16 (&acc)[MT][NT], const char* a_lane, int ld_bytes, int plane_bytes,
                                             const uint4* __restrict__ w_lane, size_t nt_stride,
                                             WRing<NP, PF, NT>& r, Side side = Side()) {
  constexpr int LO = (NP == 3) ? 1 : 0;
  h8 ah[2][MT], al[2][MT];  // [buffer][mt]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    ah[0][mt] = *(const h8*)(a_lane + mt * 32 * ld_bytes);
    if (NP == 3) al[0][mt] = *(const h8*)(a_lane + plane_bytes + mt * 32 * ld_bytes);
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
#ifdef EEC_ABLATE_A
    if (s == 0) {  // timing-only build: one LDS fragment read per stage, reused for every k-step
#else
    if (s + 1 < KS) {
#endif
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[nxt][mt] = *(const h8*)(a_lane + mt * 32 * ld_bytes + (s + 1) * 32);
        if (NP == 3) al[nxt][mt] = *(const h8*)(a_lane + plane_bytes + mt * 32 * ld_bytes + (s + 1) * 32);
      }
      __builtin_amdgcn_sched_barrier(0);  // issue next step's LDS reads BEFORE this step's MFMAs
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const h8 bh = __builtin_bit_cast(h8, r.q[s % PF][nt][0]);
      const h8 bl = __builtin_bit_cast(h8, r.q[s % PF][nt][LO]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (!SWAP) {
          if (NP == 3) {
            acc[mt][nt] = mfma16(al[cur][mt], bh, acc[mt][nt]);
            acc[mt][nt] = mfma16(ah[cur][mt], bl, acc[mt][nt]);
          }
          acc[mt][nt] = mfma16(ah[cur][mt], bh, acc[mt][nt]);
        } else {
          if (NP == 3) {
            acc[mt][nt] = mfma16(bh, al[cur][mt], acc[mt][nt]);
            acc[mt][nt] = mfma16(bl, ah[cur][mt], acc[mt][nt]);
          }
          acc[mt][nt] = mfma16(bh, ah[cur][mt], acc[mt][nt]);
        }
      }
    }
#ifdef EEC_ABLATE_W
    if (false) {  // timing-only build: no in-loop weight loads
#else
    if (s + PF < KS) {
#endif
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        r.q[s % PF][nt][0] = w_lane[nt * nt_stride + (size_t)(s + PF) * 128];
#ifdef EEC_X3_LO_SKIP
        if (NP == 3) r.q[s % PF][nt][LO] = r.q[s % PF][nt][0];
#else
        if (NP == 3) r.q[s % PF][nt][LO] = w_lane[nt * nt_stride + (size_t)(s + PF) * 128 + 64];
#endif
      }
    }
    side(s);
    if (SIDE_VALU > 0) {
#pragma unroll
      for (int i = 0; i < MT * NT * NP; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, SIDE_VALU, 0);  // then SIDE_VALU VALU (incl. transcendental)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef EEC_KSTEP_STAMPS
    if (NT == 2) eec_kstep_stamp();
#endif
  }
}

// Same product with a runtime k-step count and no ring (ragged tails only).
template <int NP, int NT, bool SWAP, int MT = 2>
__device__ __forceinline__ void gemm_plain_32(f32x16 (&acc)[MT][NT], const char* a_lane, int ld_bytes, int plane_bytes,
                                           const uint4* __restrict__ w_lane, size_t nt_stride, int ks) {
  for (int s = 0; s < ks; ++s) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const h8 bh = __builtin_bit_cast(h8, w_lane[nt * nt_stride + (size_t)s * 128]);
      h8 bl = bh;
      if (NP == 3) bl = __builtin_bit_cast(h8, w_lane[nt * nt_stride + (size_t)s * 128 + 64]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const h8 ah = *(const h8*)(a_lane + mt * 32 * ld_bytes + s * 32);
        if (NP == 3) {
          const h8 al = *(const h8*)(a_lane + plane_bytes + mt * 32 * ld_bytes + s * 32);
          acc[mt][nt] = SWAP ? mfma16(bh, al, acc[mt][nt]) : mfma16(al, bh, acc[mt][nt]);
          acc[mt][nt] = SWAP ? mfma16(bl, ah, acc[mt][nt]) : mfma16(ah, bl, acc[mt][nt]);
        }
        acc[mt][nt] = SWAP ? mfma16(bh, ah, acc[mt][nt]) : mfma16(ah, bh, acc[mt][nt]);
      }
    }
  }
}

// Ring-pipelined product with a RUNTIME k-step count (swapped orientation): the loop is
// unrolled by PF so ring slots stay statically indexed.  The ring must hold steps 0..PF-1.
template <int NP, int PF, int NT = 1>
__device__ __forceinline__ void gemm_plain_ring(f32x16 (&acc)[2][NT], const char* a_lane, int ld_bytes, int plane_bytes,
                                                const uint4* __restrict__ w_lane, size_t nt_stride, int ks, WRing<NP, PF, NT>& r) {
  constexpr int LO = (NP == 3) ? 1 : 0;
  for (int s0 = 0; s0 < ks; s0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int s = s0 + p;
      if (s < ks) {
        h8 ah[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          ah[mt] = *(const h8*)(a_lane + mt * 32 * ld_bytes + s * 32);
          if (NP == 3) al[mt] = *(const h8*)(a_lane + plane_bytes + mt * 32 * ld_bytes + s * 32);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const h8 bh = __builtin_bit_cast(h8, r.q[p][nt][0]);
          const h8 bl = __builtin_bit_cast(h8, r.q[p][nt][LO]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            if (NP == 3) {
              acc[mt][nt] = mfma16(bh, al[mt], acc[mt][nt]);
              acc[mt][nt] = mfma16(bl, ah[mt], acc[mt][nt]);
            }
            acc[mt][nt] = mfma16(bh, ah[mt], acc[mt][nt]);
          }
          if (s + PF < ks) {
            r.q[p][nt][0] = w_lane[nt * nt_stride + (size_t)(s + PF) * 128];
            if (NP == 3) r.q[p][nt][LO] = w_lane[nt * nt_stride + (size_t)(s + PF) * 128 + 64];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// ===========================================================================
// The same products on v_mfma_f32_16x16x32_f16 (EEC_MFMA16, the default).  The two fp16 shapes have the same MACs per cycle, but
// under load the chip holds a higher clock on 16x16x32 (tools/mfma_shape_bench.hip: 2.02-2.08 GHz against 1.69-1.74 GHz in bare
// loops on random data; timing-only swap inside these kernels: chain launch -7.7 %, profiles/r04_micro_mfma_shape_clock.txt).
// Everything around the k-loops keeps its layouts; only the lane <-> element map of the loads and of the accumulators differs:
//   * lane l = 32 hh + 16 u + c.  An operand fragment of row block rb (16 rows) and DOUBLE k-step S (32 k) holds, in lane l,
//     row 16 rb + c, k = 32 S + 8 (2 hh + u) + j.  From an LDS plane that is just another address; in the packed weights (1-KiB
//     fragments per 16-deep k-step, lane-linear) it is fragment 2 S + hh, slot 16 rb + c + 32 u -- another per-lane pointer into
//     the SAME packing.  A ring entry p is (S = p / 2, rb = p % 2): the ring holds the same bytes as before, differently dealt.
//   * a 32 x 32 accumulator tile is four 16 x 16 quadrants (ra, cb) in registers 4 (2 ra + cb) + i: m = 16 ra + 4 (2 hh + u) + i,
//     n = 16 cb + c ("quadrant layout").  The standard layout every epilogue expects (n = 16 u + c, m = (i & 3) + 8 (i >> 2) + 4 hh)
//     is restored with v_permlane16_swap + v_permlane32_swap on the register pairs (4 (2 ra) + i, 4 (2 ra + 1) + i): 32 cross-lane
//     instructions per tile, once per accumulation (acc_q_to_std / acc_std_to_q).
// ===========================================================================
#ifndef EEC_MFMA16
#define EEC_MFMA16 1
#endif
// (a translation-unit switch, off: the single-product format on the 16x16x32 shape too -- slower in the inference kernels; an experiment of
// the training step's fused feed-forward, whose tape stores want the quadrant layout's 16 rows x 64 B per instruction)
#ifndef EEC_MFMA16_NP1
#define EEC_MFMA16_NP1 0
#endif
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mfma32(h8 a, h8 b, f32x4 c) {
#if EEC_OPERAND_BF16
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}
// quadrant (ra, cb) of a tile += A[ra] . B[cb]^T for ra, cb in {0, 1}
__device__ __forceinline__ void tile_mac16(f32x16& acc, const h8 (&a)[2], const h8 (&b)[2]) {
#pragma unroll
  for (int ra = 0; ra < 2; ++ra)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int q = 4 * (2 * ra + cb);
      f32x4 t = {acc[q], acc[q + 1], acc[q + 2], acc[q + 3]};
      t = mfma32(a[ra], b[cb], t);
      acc[q] = t[0], acc[q + 1] = t[1], acc[q + 2] = t[2], acc[q + 3] = t[3];
    }
}
// The lane exchanges of one tile as ONE block of in-place asm (both registers of a swap are read and written; a chain of the
// builtins loses its second result in hipcc 7.2).  The eight pairs are independent, so inside the block no swap reads a register
// written fewer than seven instructions earlier; the leading s_nop covers a VALU write right in front of the block.
#define EEC_SWAP8(OP)                                                                                                        \
  "v_permlane" OP "_swap_b32 %0, %4\n\tv_permlane" OP "_swap_b32 %1, %5\n\tv_permlane" OP "_swap_b32 %2, %6\n\t"            \
  "v_permlane" OP "_swap_b32 %3, %7\n\tv_permlane" OP "_swap_b32 %8, %12\n\tv_permlane" OP "_swap_b32 %9, %13\n\t"          \
  "v_permlane" OP "_swap_b32 %10, %14\n\tv_permlane" OP "_swap_b32 %11, %15\n\t"
__device__ __forceinline__ void acc_q_to_std(f32x16& acc) {
  float r0 = acc[0], r1 = acc[1], r2 = acc[2], r3 = acc[3], r4 = acc[4], r5 = acc[5], r6 = acc[6], r7 = acc[7];
  float r8 = acc[8], r9 = acc[9], r10 = acc[10], r11 = acc[11], r12 = acc[12], r13 = acc[13], r14 = acc[14], r15 = acc[15];
  asm volatile("s_nop 1\n\t" EEC_SWAP8("16") EEC_SWAP8("32") "s_nop 1"
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(r8), "+v"(r9), "+v"(r10), "+v"(r11),
                 "+v"(r12), "+v"(r13), "+v"(r14), "+v"(r15));
  acc = (f32x16){r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15};
}
__device__ __forceinline__ void acc_std_to_q(f32x16& acc) {
  float r0 = acc[0], r1 = acc[1], r2 = acc[2], r3 = acc[3], r4 = acc[4], r5 = acc[5], r6 = acc[6], r7 = acc[7];
  float r8 = acc[8], r9 = acc[9], r10 = acc[10], r11 = acc[11], r12 = acc[12], r13 = acc[13], r14 = acc[14], r15 = acc[15];
  asm volatile("s_nop 1\n\t" EEC_SWAP8("32") EEC_SWAP8("16") "s_nop 1"
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(r8), "+v"(r9), "+v"(r10), "+v"(r11),
                 "+v"(r12), "+v"(r13), "+v"(r14), "+v"(r15));
  acc = (f32x16){r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15};
}
template <int MT, int NT>
__device__ __forceinline__ void accs_q_to_std(f32x16 (&acc)[MT][NT]) {
  // the swaps below are inline asm: hipcc does not pad the MFMA-result -> VALU-read hazard in front of them (with a single tile
  // the last MFMA's quadrant was read stale: tools/mfma16_gemm_check.hip).  19 wait states cover a 16-pass MFMA.
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc_q_to_std(acc[a][b]);
}
template <int MT, int NT>
__device__ __forceinline__ void accs_std_to_q(f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc_std_to_q(acc[a][b]);
}
// per-lane bases of the 16x16x32 fragments, derived from the 32x32x16 ones the callers pass (a_lane = plane + (lane & 31) * ld +
// 16 hh;  w_lane = fragment base + lane)
__device__ __forceinline__ const char* a_lane16(const char* a_lane, int ld_bytes) {
  const int lane = lane_id(), hh = lane >> 5, u = (lane >> 4) & 1;
  return a_lane + u * (16 - 16 * ld_bytes) + hh * 16;
}
__device__ __forceinline__ const uint4* w_lane16(const uint4* w_lane) {
  const int lane = lane_id(), hh = lane >> 5, u = (lane >> 4) & 1;
  return w_lane + 96 * hh + 16 * u;
}

// EEC_W_SADDR (experiment, off): the weight stream's loads as SGPR base + 32-bit lane offset (`global_load_dwordx4 v, v_off,
// s[base:base+1]`) instead of a 64-bit address per lane -- half the address payload per load instruction.  Measured SLOWER: chain
// launch 207.8 -> 217.6 us same-box (the uniform part has to be pinned in SGPRs per load with readfirstlane, or hipcc folds it back
// into one 64-bit VGPR address): the ~75 cycles a fragment load costs its wave are not address traffic.
#ifndef EEC_W_SADDR
#define EEC_W_SADDR 0
#endif
struct WAddr16 {
  const char* base;  // wave-uniform
  unsigned voff;     // this lane's byte offset: slot (lane & 15) + 32 u of fragment hh
  __device__ __forceinline__ uint4 load(size_t uniform_u4) const {
#if EEC_W_SADDR
    // the uniform part is pinned in SGPRs (readfirstlane is opaque to the optimiser: left alone it folds base + voff into ONE 64-bit
    // VGPR address and adds the uniform offsets to that)
    const size_t a = (size_t)base + uniform_u4 * 16;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return *(const uint4*)((const char*)(((size_t)hi << 32) | lo) + voff);
#else
    return *(const uint4*)(base + uniform_u4 * 16 + voff);
#endif
  }
};
__device__ __forceinline__ WAddr16 w_addr16(const uint4* w_lane) {
  const int lane = lane_id(), hh = lane >> 5, u = (lane >> 4) & 1;
#if EEC_W_SADDR
  const uint4* b = w_lane - lane;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)b), hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)b >> 32));
  return WAddr16{(const char*)(((size_t)hi << 32) | lo), (unsigned)(lane + 96 * hh + 16 * u) * 16u};
#else
  return WAddr16{(const char*)(w_lane + 96 * hh + 16 * u), 0u};
#endif
}

template <int NP, int PF, int NT>
__device__ __forceinline__ void ring_fill_16(WRing<NP, PF, NT>& r, const uint4* __restrict__ w_lane, size_t nt_stride, int steps_avail) {
  static_assert(PF % 2 == 0, "the ring is dealt in double k-steps");
  const WAddr16 w16 = w_addr16(w_lane);
#ifdef EEC_ABLATE_W
  if (threadIdx.x > 100000)  // timing-only build: no weight loads at all
#endif
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < steps_avail) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        r.q[p][nt][0] = w16.load(nt * nt_stride + 16 * (p & 1) + (size_t)(p >> 1) * 256);
        if (NP == 3) r.q[p][nt][(NP == 3) ? 1 : 0] = w16.load(nt * nt_stride + 16 * (p & 1) + (size_t)(p >> 1) * 256 + 64);
      }
    }
  __builtin_amdgcn_sched_barrier(0);  // keep these loads HERE: one stage ahead of their consumer
}

// gemm_ring on the 16x16x32 shape.  IN_STD / OUT_STD: the accumulators arrive / leave in the standard layout (converted here);
// false: they arrive / stay in the quadrant layout (a caller that accumulates over several calls converts once at the end).
// The loop walks HALF double-steps p = 2 S + rb: step p multiplies row block rb of the activation operand (one fragment set: the
// same registers per step as the 32x32x16 loop's k-step) with both row blocks of the weight operand (ring entries 2 S, 2 S + 1),
// i.e. the two quadrants of each tile that row block rb of the activations belongs to.
__device__ __forceinline__ void quad_mac16(f32x16& acc, int ra, int cb, h8 a, h8 b) {  // ra, cb constants after unrolling
  const int q = 4 * (2 * ra + cb);
  f32x4 t = {acc[q], acc[q + 1], acc[q + 2], acc[q + 3]};
  t = mfma32(a, b, t);
  acc[q] = t[0], acc[q + 1] = t[1], acc[q + 2] = t[2], acc[q + 3] = t[3];
}
template <int NP, int KS, int NT, bool SWAP, int PF, typename Side = NoSide, int SIDE_VALU = 0, int MT = 2, bool IN_STD = true, bool OUT_STD = true>
__device__ __forceinline__ void gemm_ring_16(f32x16 (&acc)[MT][NT], const char* a_lane, int ld_bytes, int plane_bytes,
                                             const uint4* __restrict__ w_lane, size_t nt_stride, WRing<NP, PF, NT>& r, Side side = Side()) {
  static_assert(KS % 2 == 0 && PF % 2 == 0, "double k-steps");
  constexpr int LO = (NP == 3) ? 1 : 0;
  const char* a16 = a_lane16(a_lane, ld_bytes);
  const WAddr16 w16 = w_addr16(w_lane);
  if constexpr (IN_STD) accs_std_to_q<MT, NT>(acc);
  h8 ah[2][MT], al[2][MT];  // [buffer][mt]: row block p & 1 of double step p >> 1
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    ah[0][mt] = *(const h8*)(a16 + mt * 32 * ld_bytes);
    if (NP == 3) al[0][mt] = *(const h8*)(a16 + plane_bytes + mt * 32 * ld_bytes);
  }
#pragma unroll
  for (int p = 0; p < KS; ++p) {
    const int S = p >> 1, rb = p & 1, cur = p & 1, nxt = cur ^ 1;
#ifdef EEC_ABLATE_A
    if (p == 0) {
#else
    if (p + 1 < KS) {
#endif
      const int off = (16 * ((p + 1) & 1)) * ld_bytes + ((p + 1) >> 1) * 64;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[nxt][mt] = *(const h8*)(a16 + mt * 32 * ld_bytes + off);
        if (NP == 3) al[nxt][mt] = *(const h8*)(a16 + plane_bytes + mt * 32 * ld_bytes + off);
      }
      __builtin_amdgcn_sched_barrier(0);  // issue the next step's LDS reads BEFORE this step's MFMAs
    }
    // weight row block wb outermost: ring entry (S, wb) is finished -- and refilled PF / 2 double steps ahead -- as soon as its
    // products of the step with rb == 1 are issued, half a step before the double step ends.  (With the refill at the end of the
    // double step a two-entry ring had NO lead at all: the loads of double step S + 1 were requested when S + 1 began.)
#pragma unroll
    for (int wb = 0; wb < 2; ++wb) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const h8 bh = __builtin_bit_cast(h8, r.q[(2 * S + wb) % PF][nt][0]);
        const h8 bl = __builtin_bit_cast(h8, r.q[(2 * S + wb) % PF][nt][LO]);
#pragma unroll
        for (int pr = (NP == 3 ? 0 : 2); pr < 3; ++pr)  // a_lo . w_hi, a_hi . w_lo, a_hi . w_hi
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const h8 a = pr == 0 ? al[cur][mt] : ah[cur][mt], w = pr == 1 ? bl : bh;
            if (!SWAP) quad_mac16(acc[mt][nt], rb, wb, a, w);  // A = activations (row block rb), B = weights (row block wb)
            else quad_mac16(acc[mt][nt], wb, rb, w, a);        // A = weights, B = activations
          }
        // this (row block, column tile) of the ring is finished: refill it right away -- the loads leave in pairs spread over the
        // step instead of in one burst at its end (an in-order wave stalls at issue while the CU's load path is backed up)
        if (rb == 1) {
#ifdef EEC_ABLATE_W
          if (false) {
#else
          if (2 * S + PF < KS) {
#endif
            r.q[(2 * S + wb) % PF][nt][0] = w16.load(nt * nt_stride + 16 * wb + (size_t)(S + PF / 2) * 256);
#ifdef EEC_X3_LO_SKIP
            if (NP == 3) r.q[(2 * S + wb) % PF][nt][LO] = r.q[(2 * S + wb) % PF][nt][0];
#else
            if (NP == 3) r.q[(2 * S + wb) % PF][nt][LO] = w16.load(nt * nt_stride + 16 * wb + (size_t)(S + PF / 2) * 256 + 64);
#endif
          }
        }
      }
    }
    side(p);
    if (SIDE_VALU > 0) {
#pragma unroll
      for (int i = 0; i < MT * NT * NP; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);          // two 16x16x32 MFMAs (= one 32x32x16)
        __builtin_amdgcn_sched_group_barrier(0x002, SIDE_VALU, 0);  // then SIDE_VALU VALU (incl. transcendental)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef EEC_KSTEP_STAMPS
    if (NT == 2) eec_kstep_stamp();
#endif
  }
  if constexpr (OUT_STD) accs_q_to_std<MT, NT>(acc);
}

// runtime (even) k-step count, no ring: ragged tails only
template <int NP, int NT, bool SWAP, int MT = 2, bool IN_STD = true, bool OUT_STD = true>
__device__ __forceinline__ void gemm_plain_16(f32x16 (&acc)[MT][NT], const char* a_lane, int ld_bytes, int plane_bytes,
                                              const uint4* __restrict__ w_lane, size_t nt_stride, int ks) {
  const char* a16 = a_lane16(a_lane, ld_bytes);
  const uint4* w16 = w_lane16(w_lane);
  if constexpr (IN_STD) accs_std_to_q<MT, NT>(acc);
  for (int S = 0; S < ks / 2; ++S) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      h8 bh[2], bl[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        bh[rb] = __builtin_bit_cast(h8, w16[nt * nt_stride + 16 * rb + (size_t)S * 256]);
        bl[rb] = bh[rb];
        if (NP == 3) bl[rb] = __builtin_bit_cast(h8, w16[nt * nt_stride + 16 * rb + (size_t)S * 256 + 64]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        h8 ah[2], al[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          ah[rb] = *(const h8*)(a16 + (mt * 32 + 16 * rb) * ld_bytes + S * 64);
          al[rb] = ah[rb];
          if (NP == 3) al[rb] = *(const h8*)(a16 + plane_bytes + (mt * 32 + 16 * rb) * ld_bytes + S * 64);
        }
        if (NP == 3) {
          if (SWAP) tile_mac16(acc[mt][nt], bh, al), tile_mac16(acc[mt][nt], bl, ah);
          else tile_mac16(acc[mt][nt], al, bh), tile_mac16(acc[mt][nt], ah, bl);
        }
        if (SWAP) tile_mac16(acc[mt][nt], bh, ah);
        else tile_mac16(acc[mt][nt], ah, bh);
      }
    }
  }
  if constexpr (OUT_STD) accs_q_to_std<MT, NT>(acc);
}

// The names the kernels use: the 16x16x32 forms for the split format NP = 3 (-DEEC_MFMA16=0: the 32x32x16 forms everywhere).
// NP = 1 keeps 32x32x16: with a third of the MFMAs per SiLU the producers are VALU-bound, and a 16x16x32 MFMA holds the SIMD's
// vector issue for 8 of its 16 cycles instead of 8 of 32 (measured: `mixed` 27.0 -> 22.9 M, `f16` 30.3 -> 23.9 M frames/s).
// IN_STD / OUT_STD only matter for the 16x16x32 forms (the 32x32x16 accumulators are always in the standard layout).
template <int NP>
constexpr bool kMfma16For = (EEC_MFMA16 != 0) && (NP == 3 || (EEC_MFMA16_NP1 != 0 && NP == 1));
template <int NP, int PF, int NT>
__device__ __forceinline__ void ring_fill(WRing<NP, PF, NT>& r, const uint4* __restrict__ w_lane, size_t nt_stride, int steps_avail) {
  if constexpr (kMfma16For<NP>) ring_fill_16<NP, PF, NT>(r, w_lane, nt_stride, steps_avail);
  else ring_fill_32<NP, PF, NT>(r, w_lane, nt_stride, steps_avail);
}
template <int NP, int KS, int NT, bool SWAP, int PF, typename Side = NoSide, int SIDE_VALU = 0, int MT = 2, bool IN_STD = true, bool OUT_STD = true>
__device__ __forceinline__ void gemm_ring(f32x16 (&acc)[MT][NT], const char* a_lane, int ld_bytes, int plane_bytes,
                                          const uint4* __restrict__ w_lane, size_t nt_stride, WRing<NP, PF, NT>& r, Side side = Side()) {
  if constexpr (kMfma16For<NP>) gemm_ring_16<NP, KS, NT, SWAP, PF, Side, SIDE_VALU, MT, IN_STD, OUT_STD>(acc, a_lane, ld_bytes, plane_bytes, w_lane, nt_stride, r, side);
  else gemm_ring_32<NP, KS, NT, SWAP, PF, Side, SIDE_VALU, MT>(acc, a_lane, ld_bytes, plane_bytes, w_lane, nt_stride, r, side);
}
template <int NP, int NT, bool SWAP, int MT = 2, bool IN_STD = true, bool OUT_STD = true>
__device__ __forceinline__ void gemm_plain(f32x16 (&acc)[MT][NT], const char* a_lane, int ld_bytes, int plane_bytes,
                                           const uint4* __restrict__ w_lane, size_t nt_stride, int ks) {
  if constexpr (kMfma16For<NP>) gemm_plain_16<NP, NT, SWAP, MT, IN_STD, OUT_STD>(acc, a_lane, ld_bytes, plane_bytes, w_lane, nt_stride, ks);
  else gemm_plain_32<NP, NT, SWAP, MT>(acc, a_lane, ld_bytes, plane_bytes, w_lane, nt_stride, ks);
}

// ===========================================================================
// NP == 8: "fp16 + 2 x fp8-correction" product.
//   a.w  ~=  a_hi.w_hi  (v_mfma_f32_32x32x16_f16)
//          + a_lo8.w_hi8 + a_hi8.w_lo8  (v_mfma_scale_f32_32x32x64_f8f6f4, bf8 = e5m2 operands, 2x the fp16 rate)
// The two correction terms are ~2^-11 of the main term, so 3 significant bits in their operands keep the
// product good to ~13-14 bits: measured |dlogp| 2.9e-4 (FFN only) against 2.4e-4 for the exact 3-pass split.
// e5m2 is the top byte of an fp16, so a_hi8 / w_hi8 are byte-permutes of the fp16 fragments already in
// registers; a_lo8 is a byte plane in LDS; w_lo8 is packed per weight with an E8M0 block scale (the weight
// residuals are fp16-subnormal: the block scale of the MX instruction is what makes them representable).
// Hardware layout of the K=64 instruction (tools/mx_probe.hip): lane l holds row/col l%32; its 32 bytes pair
// element-wise between A and B, so any consistent k <-> byte-slot map works; bytes 0-15 / 16-31 of a lane form
// scale blocks 0 / 1 (together with the same bytes of lane l^32), and the scale of block b comes from lane r+32b.
// Slot map used here: byte p of lane half h of a 64-k group  <->  k = 16*(p/8) + 8*h + p%8, i.e. the concatenation
// of the lane's four fp16 k-step fragments; a scale block is then two consecutive k-steps (32 consecutive k).
// ===========================================================================
typedef int i32x8 __attribute__((ext_vector_type(8)));
constexpr int kE8M0One = 127;
// The e5m2 operands derived inside the kernels are TRUNCATIONS (the top byte of an fp16): hi8 = trunc(hi) is low by 8.3 % on
// average (uniform mantissas), so each correction product would come out ~8 % (weight-residual term) / ~16 % (activation-
// residual term: both of its operands are truncated) short, coherently.  The PARTNER operand of each product carries the
// compensating gain: the packed weight residuals are scaled by kF8WLoGain before their e5m2 rounding (pack.hip), the
// activation residuals by kF8ALoGain before truncation.  CPU emulation on N(0,1) x U(+-0.06), K = 256: rms error of the
// two terms 1.9e-5 / 1.1e-5 -> 0.86e-5 / 0.76e-5 (the whole product 2.2e-5 -> 1.15e-5; 3-pass split 3e-8, 1-pass 1.6e-4).
constexpr float kF8WLoGain = 1.09f;
constexpr float kF8ALoGain = 1.17f;
// e5m2 bytes (fp16 top bytes) of a pair of residual halves, gain applied: two such pairs fill one dword of the lo8 plane
__device__ __forceinline__ h2 lo8_gain(h2 lo) {
  const h2 g = {(half_t)kF8ALoGain, (half_t)kF8ALoGain};
  return lo * g;  // v_pk_mul_f16
}

// top bytes (e5m2) of the 8 halves of an fp16 fragment -> 8 bytes
__device__ __forceinline__ uint2 top_bytes(uint4 f) {
  return make_uint2(__builtin_amdgcn_perm(f.y, f.x, 0x07050301u), __builtin_amdgcn_perm(f.w, f.z, 0x07050301u));
}
__device__ __forceinline__ f32x16 mfma_f8(i32x8 a, i32x8 b, f32x16 c, int scale_a, int scale_b) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1 /*A: bf8*/, 1 /*B: bf8*/, 0, scale_a, 0, scale_b);
}
// byte position of activation column k inside the permuted lo8 plane row (64-k groups, see slot map)
__device__ __forceinline__ int lo8_pos(int k) {
  return (k & ~63) + 32 * ((k >> 3) & 1) + 8 * ((k >> 4) & 3) + (k & 7);
}

// Packed "f8" weight stream: per (n-tile, 64-k group) one record of 400 uint4:
//   [4 hi fragments x 64 lanes][lo8: 64 lanes x 2 uint4][scale: 64 lanes x 1 dword (64 B used of 256)]
constexpr int kF8Rec = 400;
template <int NT>
struct WGroupF8 {      // lo8 + scale of ONE 64-k group for NT n-tiles
  uint4 lo[NT][2];
  int sc[NT];
};
template <int NT>
__device__ __forceinline__ void f8_group_load(WGroupF8<NT>& g, const uint4* __restrict__ rec_lane, size_t nt_stride) {
  // rec_lane = record base + lane
#ifdef EEC_ABLATE_W
  if (threadIdx.x > 100000)  // timing-only build: no weight loads at all
#endif
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const uint4* r = rec_lane + nt * nt_stride - lane_id();
    g.lo[nt][0] = wload(r + 256 + 2 * lane_id());
#ifdef EEC_LO_HALF  // timing-only build: half the residual bytes (what an fp4 residual would stream)
    g.lo[nt][1] = g.lo[nt][0];
#else
    g.lo[nt][1] = wload(r + 256 + 2 * lane_id() + 1);
#endif
    g.sc[nt] = ((const int*)(r + 384))[lane_id()];
  }
}

// acc[2][NT] += Act x W over NG 64-k groups.  a_lane / a8_lane: this lane's fp16-hi and byte-lo8 LDS pointers
// (row l%32, half l/32); rec_lane: first record (+lane); consecutive groups are kF8Rec uint4 apart.
// hi fragments run through the register ring `r` (PF k-steps ahead, as gemm_ring); the lo8/scale blocks of
// the whole stage are in `wg` (loaded by the caller one stage ahead).
// NW: number of lo8 group buffers.  NW == NG: the whole stage is resident (loaded by the caller one stage ahead);
// NW < NG: rolling buffers, group g lives in wg[g % NW] and is refilled with group g + NW right after its use
// (the caller preloads groups 0 .. NW-1 of the next stage).
// CONT: the stream CONTINUES into another product of the same shape whose first record (+lane) is `next_lane` (wave-uniform,
// may be null): the last PF k-steps refill the ring -- and the last NW groups their lo8 buffers -- with that product's first
// steps / groups, so the caller starts it with a full pipeline and no separate fill.
// A8HI > 0: the activation tile's byte plane also holds the hi bytes, A8HI bytes behind the residual bytes of the same row
// (Geo::kA8Hi; EEC_X_HI8): they are read as the group's operand instead of being permuted out of the fp16 fragments.
template <int NG, int NT, bool SWAP, int PF, typename Side = NoSide, int SIDE_VALU = 0, int NW = NG, int DROP = 0, int MT = 2, bool CONT = false,
          int A8HI = 0>
__device__ __forceinline__ void gemm_ring_f8(f32x16 (&acc)[MT][NT], const char* a_lane, int ld_bytes, const char* a8_lane,
                                             int ld8_bytes, const uint4* __restrict__ rec_lane, size_t nt_stride,
                                             WRing<1, PF, NT>& r, WGroupF8<NT> (&wg)[NW], Side side = Side(),
                                             const uint4* __restrict__ next_lane = nullptr) {
  constexpr int KS = 4 * NG;
  static_assert(!CONT || (KS % PF == 0 && NG % NW == 0), "a continued stream keeps its ring phase: PF | k-steps, NW | groups");
  auto hi_addr = [&](int s) { return (size_t)(s >> 2) * kF8Rec + (size_t)(s & 3) * 64; };
  h8 ah[2][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ah[0][mt] = *(const h8*)(a_lane + mt * 32 * ld_bytes);
  [[maybe_unused]] uint2 a8[MT][4];  // [mt][step in group]: e5m2 of the activation hi fragments (A8HI == 0)
  [[maybe_unused]] i32x8 ahi8[MT];   // ... or the group's hi bytes read from the byte plane (A8HI > 0)
  uint2 w8[NT][4];  // [nt][step in group]: e5m2 of the weight hi fragments
  i32x8 alo[MT];    // activation lo8 of the current group
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int cur = s & 1, nxt = cur ^ 1, q = s & 3, g = s >> 2;
#ifdef EEC_ABLATE_A
    if (s == 0) {  // timing-only build: one LDS fragment read per stage
#else
    if (s + 1 < KS) {
#endif
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) ah[nxt][mt] = *(const h8*)(a_lane + mt * 32 * ld_bytes + (s + 1) * 32);
    }
#ifdef EEC_ABLATE_A
    if (s == 0) {
#else
    if (q == 0) {
#endif
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const uint4 l0 = *(const uint4*)(a8_lane + mt * 32 * ld8_bytes + g * 64);
        const uint4 l1 = *(const uint4*)(a8_lane + mt * 32 * ld8_bytes + g * 64 + 16);
        alo[mt] = (i32x8){(int)l0.x, (int)l0.y, (int)l0.z, (int)l0.w, (int)l1.x, (int)l1.y, (int)l1.z, (int)l1.w};
        if constexpr (A8HI > 0) {
          const uint4 h0 = *(const uint4*)(a8_lane + A8HI + mt * 32 * ld8_bytes + g * 64);
          const uint4 h1 = *(const uint4*)(a8_lane + A8HI + mt * 32 * ld8_bytes + g * 64 + 16);
          ahi8[mt] = (i32x8){(int)h0.x, (int)h0.y, (int)h0.z, (int)h0.w, (int)h1.x, (int)h1.y, (int)h1.z, (int)h1.w};
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (A8HI == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a8[mt][q] = top_bytes(__builtin_bit_cast(uint4, ah[cur][mt]));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const uint4 wq = r.q[s % PF][nt][0];
      w8[nt][q] = top_bytes(wq);
      const h8 bh = __builtin_bit_cast(h8, wq);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = SWAP ? mfma16(bh, ah[cur][mt], acc[mt][nt]) : mfma16(ah[cur][mt], bh, acc[mt][nt]);
    }
#ifdef EEC_ABLATE_W
    if (false) {
#else
    if (s + PF < KS) {
#endif
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) r.q[s % PF][nt][0] = wload(rec_lane + nt * nt_stride + hi_addr(s + PF));
    } else if (CONT && next_lane) {
#ifndef EEC_ABLATE_W
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) r.q[s % PF][nt][0] = wload(next_lane + nt * nt_stride + hi_addr(s + PF - KS));
#endif
    }
    if (q == 3) {  // the group's two correction products
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const i32x8 whi = {(int)w8[nt][0].x, (int)w8[nt][0].y, (int)w8[nt][1].x, (int)w8[nt][1].y,
                           (int)w8[nt][2].x, (int)w8[nt][2].y, (int)w8[nt][3].x, (int)w8[nt][3].y};
        const WGroupF8<NT>& G = wg[g % NW];
        const i32x8 wlo = {(int)G.lo[nt][0].x, (int)G.lo[nt][0].y, (int)G.lo[nt][0].z, (int)G.lo[nt][0].w,
                           (int)G.lo[nt][1].x, (int)G.lo[nt][1].y, (int)G.lo[nt][1].z, (int)G.lo[nt][1].w};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          i32x8 ahi;
          if constexpr (A8HI > 0) ahi = ahi8[mt];
          else ahi = (i32x8){(int)a8[mt][0].x, (int)a8[mt][0].y, (int)a8[mt][1].x, (int)a8[mt][1].y,
                             (int)a8[mt][2].x, (int)a8[mt][2].y, (int)a8[mt][3].x, (int)a8[mt][3].y};
          // DROP (diagnostic builds only): bit 0 skips the activation-residual term, bit 1 the weight-residual term
          if (SWAP) {
            if (!(DROP & 1)) acc[mt][nt] = mfma_f8(whi, alo[mt], acc[mt][nt], kE8M0One, kE8M0One);
            if (!(DROP & 2)) acc[mt][nt] = mfma_f8(wlo, ahi, acc[mt][nt], G.sc[nt], kE8M0One);
          } else {
            if (!(DROP & 1)) acc[mt][nt] = mfma_f8(alo[mt], whi, acc[mt][nt], kE8M0One, kE8M0One);
            if (!(DROP & 2)) acc[mt][nt] = mfma_f8(ahi, wlo, acc[mt][nt], kE8M0One, G.sc[nt]);
          }
        }
      }
      if (NW < NG && g + NW < NG) f8_group_load<NT>(wg[g % NW], rec_lane + (size_t)(g + NW) * kF8Rec, nt_stride);
      else if (CONT && next_lane) f8_group_load<NT>(wg[g % NW], next_lane + (size_t)(g + NW - NG) * kF8Rec, nt_stride);
    }
    side(s);
    if (SIDE_VALU > 0) {
#pragma unroll
      for (int i = 0; i < MT * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, SIDE_VALU, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ring fill for the f8 stream: hi fragments of the first PF k-steps of a stage
template <int PF, int NT>
__device__ __forceinline__ void ring_fill_f8(WRing<1, PF, NT>& r, const uint4* __restrict__ rec_lane, size_t nt_stride) {
#ifdef EEC_ABLATE_W
  if (threadIdx.x > 100000)  // timing-only build: no weight loads at all
#endif
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) r.q[p][nt][0] = wload(rec_lane + nt * nt_stride + (size_t)(p >> 2) * kF8Rec + (size_t)(p & 3) * 64);
  __builtin_amdgcn_sched_barrier(0);
}

template <int I>
struct IntTag {
  static constexpr int value = I;
};
// f(IntTag<K0>{}), ..., f(IntTag<K1-1>{}): a loop whose index is a compile-time constant in the body (register arrays
// indexed by it never fall back to scratch memory, whatever the unroller decides)
template <int K0, int K1, typename F>
__device__ __forceinline__ void static_range(F&& f) {
  if constexpr (K0 < K1) {
    f(IntTag<K0>{});
    static_range<K0 + 1, K1>(f);
  }
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
}

// ---------------------------------------------------------------------------
// Prologues: fill the [rows][D] activation planes in LDS.
// ---------------------------------------------------------------------------
// N wave-wide sums as N independent, interleaved DPP chains (6 dependent cross-lane adds each).
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = EEC_DPP_ADD(v[i], 0xB1, 0xf);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = EEC_DPP_ADD(v[i], 0x4E, 0xf);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = EEC_DPP_ADD(v[i], 0x141, 0xf);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = EEC_DPP_ADD(v[i], 0x140, 0xf);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = EEC_DPP_ADD(v[i], 0x142, 0xa);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = EEC_DPP_ADD(v[i], 0x143, 0xc);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 63));
}

// A row of D columns is held by a wave as Q = D / 256 float4 pieces per lane: piece q = columns [256 q + 4 lane, + 4).
template <int Q>
struct RowV {
  float4 p[Q];
};
template <int D>
__device__ __forceinline__ RowV<Geo<D>::kQ> load_row(const float* __restrict__ row_ptr, int lane) {
  RowV<Geo<D>::kQ> r;
#pragma unroll
  for (int q = 0; q < Geo<D>::kQ; ++q) r.p[q] = ((const float4*)row_ptr)[q * 64 + lane];
  return r;
}
template <int D>
__device__ __forceinline__ void store_row(float* __restrict__ row_ptr, const RowV<Geo<D>::kQ>& r, int lane) {
#pragma unroll
  for (int q = 0; q < Geo<D>::kQ; ++q) ((float4*)row_ptr)[q * 64 + lane] = r.p[q];
}
template <int Q>
__device__ __forceinline__ RowV<Q> zero_row() {
  RowV<Q> r;
#pragma unroll
  for (int q = 0; q < Q; ++q) r.p[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  return r;
}

// LayerNorm of N rows (row = D columns over 64 lanes).  The reductions of the N rows run as N independent,
// interleaved DPP chains.
template <int D, int N>
__device__ __forceinline__ void layer_norm_rows(RowV<Geo<D>::kQ> (&v)[N], const RowV<Geo<D>::kQ>& g, const RowV<Geo<D>::kQ>& bt,
                                                float* mean_out = nullptr, float* rstd_out = nullptr) {  // optional: the N rows' statistics
  constexpr int Q = Geo<D>::kQ;
  float s[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    s[i] = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) s[i] += v[i].p[q].x + v[i].p[q].y + v[i].p[q].z + v[i].p[q].w;
  }
  wave_sum_n<N>(s);
  float sq[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float mean = s[i] * (1.0f / D);
    if (mean_out) mean_out[i] = mean;
    sq[i] = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float4& t = v[i].p[q];
      t.x -= mean, t.y -= mean, t.z -= mean, t.w -= mean;
      sq[i] += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
  }
  wave_sum_n<N>(sq);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float rs = rsqrtf(sq[i] * (1.0f / D) + kLnEps);
    if (rstd_out) rstd_out[i] = rs;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float4& t = v[i].p[q];
      t.x = t.x * rs * g.p[q].x + bt.p[q].x;
      t.y = t.y * rs * g.p[q].y + bt.p[q].y;
      t.z = t.z * rs * g.p[q].z + bt.p[q].z;
      t.w = t.w * rs * g.p[q].w + bt.p[q].w;
    }
  }
}

// N rows (local rows rl0 .. rl0+N-1 of the tile) -> activation planes:
// fp16 hi plane, plus the fp16 lo plane (NP == 3) or the permuted e5m2 lo byte plane (NP == 8).
// Rows at or beyond M are written as zeros when zero_tail is set (a LayerNorm turns a zero row into beta).
template <int D, int NP, int N>
__device__ __forceinline__ void rows_to_planes(char* lds_act, RowV<Geo<D>::kQ> (&v)[N], int rl0, int row0, int M, bool zero_tail,
                                               int lane = lane_id()) {
  using G = Geo<D>;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int rl = rl0 + i;
    if (zero_tail && row0 + rl >= M) v[i] = zero_row<G::kQ>();
#pragma unroll
    for (int q = 0; q < G::kQ; ++q) {
      const float4 t = v[i].p[q];
      const int col = q * 256 + lane * 4;
      const hl2_t s0 = split2<(NP == 1 ? 1 : 3)>(t.x, t.y), s1 = split2<(NP == 1 ? 1 : 3)>(t.z, t.w);
      h4 hi, lo;
      hi.xy = s0.hi, hi.zw = s1.hi, lo.xy = s0.lo, lo.zw = s1.lo;
      *(h4*)(lds_act + rl * G::kALd + col * 2) = hi;
      if (NP == 3) *(h4*)(lds_act + G::kAPlane + rl * G::kALd + col * 2) = lo;
      if (NP == 8) {  // e5m2 bytes of the (gain-compensated) lo halves, permuted to the MX slot order
        h4 lg;
        lg.xy = lo8_gain(s0.lo), lg.zw = lo8_gain(s1.lo);
        const uint2 lb = __builtin_bit_cast(uint2, lg);
        *(unsigned*)(lds_act + G::kAPlane + rl * G::kA8Ld + lo8_pos(col)) = __builtin_amdgcn_perm(lb.y, lb.x, 0x07050301u);
        if (EEC_X_HI8) {
          const uint2 hb = __builtin_bit_cast(uint2, hi);
          *(unsigned*)(lds_act + G::kAPlane + rl * G::kA8Ld + G::kA8Hi + lo8_pos(col)) = __builtin_amdgcn_perm(hb.y, hb.x, 0x07050301u);
        }
      }
    }
  }
}

struct NoPrefetch {
  __device__ __forceinline__ void operator()() const {}
};
// x fp32 [M][D] -> (optional LayerNorm) -> planes.  Each wave handles RPW rows; a row is Q coalesced 1 KiB float4 loads.
// `after_loads()` runs once, right after the first batch of row loads has been issued and before their first use: the
// place to start the weight streams.  (Loads return in order: a weight prefetch issued BEFORE the rows makes the
// LayerNorm wait behind ~100 KiB of weight traffic.)
template <int D, int NP, bool DO_LN, typename After = NoPrefetch>
__device__ __forceinline__ void rows_f32_to_planes(char* lds_act, const float* __restrict__ x, int row0, int M,
                                                   const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, After after_loads = After()) {
  using G = Geo<D>;
  constexpr int RPW = G::kRPW;
  const int lane = lane_id(), w = wave_id();
  RowV<G::kQ> g, bt;
  if (DO_LN) {
    g = load_row<D>(gamma, lane);
    bt = load_row<D>(beta, lane);
  }
  RowV<G::kQ> v[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = row0 + w * RPW + i;
    v[i] = zero_row<G::kQ>();
    if (row < M) v[i] = load_row<D>(x + (size_t)row * D, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  after_loads();
  if (DO_LN) layer_norm_rows<D, RPW>(v, g, bt);
  rows_to_planes<D, NP, RPW>(lds_act, v, w * RPW, row0, M, DO_LN);
}

// acc[MT][NT] (normal orientation, wave owns columns col0 + nt*32 ..) + bias -> fp32 tile in LDS (row stride e_ld).
template <int MT, int NT>
__device__ __forceinline__ void acc_to_etile(char* lds_e, int e_ld, const f32x16 (&acc)[MT][NT], int col0,
                                             const float* __restrict__ bias, int lane = lane_id()) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = col0 + nt * 32 + (lane & 31);
    const float b = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = mt * 32 + acc_row(i, lane);
        *(float*)(lds_e + r * e_ld + col * 4) = acc[mt][nt][i] + b;
      }
  }
}

}  // namespace eec

// Diagnostic phase timeline (tools/phase_timeline.py; -DEEC_TL builds only): s_memtime stamps of the first
// and last wave of workgroups 0..7, read back through eec_tl_read_<NAME>().
#ifdef EEC_TL
#define EEC_TL_DEFINE(NAME)                                                                       \
  __device__ unsigned long long g_tl_##NAME[8 * 2 * 16];                                          \
  extern "C" int eec_tl_read_##NAME(unsigned long long* out) {                                    \
    (void)hipDeviceSynchronize();                                                                 \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl_##NAME), sizeof(g_tl_##NAME));           \
  }
#define EEC_TL_STAMP(NAME, IDX)                                                                   \
  do {                                                                                            \
    const int w_ = (int)(threadIdx.x >> 6), wl_ = (int)(blockDim.x >> 6) - 1;                     \
    const int b_ = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));         \
    if (b_ < 8 && (w_ == 0 || w_ == wl_) && (threadIdx.x & 63) == 0)                              \
      g_tl_##NAME[(b_ * 2 + (w_ == wl_)) * 16 + (IDX)] = __builtin_amdgcn_s_memtime();            \
  } while (0)
#else
#define EEC_TL_DEFINE(NAME)
#define EEC_TL_STAMP(NAME, IDX)
#endif

