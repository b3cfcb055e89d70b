// Kernels of the training step (see eec_train.h): one general bf16-split MFMA GEMM and the row / column / pointwise
// kernels around it.  fp32 in HBM everywhere; nothing here is shared with the fused inference path.
#include "eec_drop.h"

namespace eect {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 32;   // k extent of one LDS tile
constexpr int kLdk = 40;  // bf16 elements per LDS tile row (80 B: the 16 lanes of a ds_read_b128 group cover all 64 banks once)

__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// wave-wide reductions on the DPP crossbar (no LDS traffic): quad swaps, half-row / row mirrors, then row broadcasts; lane 63
// holds the total, returned wave-uniform
#define EECT_DPP_ADD(v, ctrl, rmask) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, rmask, 0xf, false)))
#define EECT_DPP_MAX(v, ctrl, rmask) \
  fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), ctrl, rmask, 0xf, false)))
__device__ __forceinline__ float wave_sum(float v) {
  v = EECT_DPP_ADD(v, 0xB1, 0xf);
  v = EECT_DPP_ADD(v, 0x4E, 0xf);
  v = EECT_DPP_ADD(v, 0x141, 0xf);
  v = EECT_DPP_ADD(v, 0x140, 0xf);
  v = EECT_DPP_ADD(v, 0x142, 0xa);
  v = EECT_DPP_ADD(v, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = EECT_DPP_MAX(v, 0xB1, 0xf);
  v = EECT_DPP_MAX(v, 0x4E, 0xf);
  v = EECT_DPP_MAX(v, 0x141, 0xf);
  v = EECT_DPP_MAX(v, 0x140, 0xf);
  v = EECT_DPP_MAX(v, 0x142, 0xa);
  v = EECT_DPP_MAX(v, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// sigmoid on the hardware reciprocal (1 ulp) instead of an IEEE division (a ~10-instruction sequence): the epilogues and
// pointwise kernels evaluate it once per activation element
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------------------------------
// dropout
// ---------------------------------------------------------------------------------------------------------------------
// the generator (drop_key, drop_keep, DropState with mul / mul4) is in eec_drop.h: the fused feed-forward forward (ffn.hip) shares it

// ---------------------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------------------
// LDS image of one operand tile (R rows x 32 k), two layouts:
//   KC (the operand is k-contiguous in memory): [R][kLdk] bf16, fragments read with ds_read_b128;
//   transposed (row-contiguous in memory): [32 k][ldt(R)] bf16 -- stored as it arrives, 4 consecutive rows per thread with
//   one 8-byte store -- and read back through ds_read_b64_tr_b16, the hardware transpose: a 16-lane group reads a
//   4 (k) x 16 (rows) block and every lane receives the 4 k-values of its row.  ldt = R + 32 (R for R = 32) puts the four
//   k-rows of a block on disjoint bank quarters.
// row of a k-contiguous tile that thread group g (8 threads = the 32 k of one row) owns: neighbouring groups take rows 4 apart,
// so that the two rows a 16-lane ds_write_b64 group stores (80-byte row stride) fall on disjoint halves of the 32 write banks
__device__ __forceinline__ int kc_row(int g) { return (g & ~7) | ((g & 1) << 2) | ((g >> 1) & 3); }

template <int R>
struct TileGeo {
  static constexpr int kLdt = R == 32 ? 32 : R + 32;
  static constexpr int kElems = R * kLdk > 32 * kLdt ? R * kLdk : 32 * kLdt;
};

// Loader of the R x 32 tiles of one operand X(r, k) = X[s_r * r + s_k * k] (one of the strides is 1): each thread owns
// R / 32 groups of 4 consecutive elements along the unit-stride axis.  Pointers and row validity are set up once; a full
// interior tile is R / 32 unchecked 16-byte loads per thread, edges (last k-tile, ragged rows, unaligned bases) go element-wise.
template <int R, bool KC>
struct TileLoader {
  const float* p[R / 32];
  const float* x0;          // the operand's base (uniform)
  uint32_t off[R / 32];     // this thread's element offsets from x0 + k0 * kstep, rows clamped into the matrix (fast path)
  int nv[R / 32];  // KC: 4 if the row exists else 0; transposed: how many of the thread's 4 rows exist
  long kstep;      // elements between consecutive k
  bool vec;        // every 4-element group is 16-byte aligned
  bool nt = false; // the operand is a large tensor read once by this launch: loads carry the non-temporal hint (GemmArgs::stream_a / _b)
  bool clean;      // vec, and no 4-row group of a transposed operand straddles the last row
  __device__ __forceinline__ void init(const float* __restrict__ X, long s_r, long s_k, int r0, int Rmax, int tid) {
    kstep = KC ? 1 : s_k;
    x0 = X;
    vec = (((uintptr_t)X) & 15) == 0 && ((KC ? s_r : s_k) & 3) == 0;
    clean = vec && (KC || (Rmax & 3) == 0);
#pragma unroll
    for (int it = 0; it < R / 32; ++it) {
      const int idx = it * 256 + tid;
      if (KC) {
        const int r = r0 + kc_row(idx >> 3);
        nv[it] = r < Rmax ? 4 : 0;
        p[it] = X + (long)min(r, Rmax - 1) * s_r + (idx & 7) * 4;  // clamped: always a readable row
        off[it] = (uint32_t)(p[it] - X);
      } else {
        const int r = r0 + (idx % (R / 4)) * 4;
        nv[it] = min(max(Rmax - r, 0), 4);
        p[it] = X + (long)(idx / (R / 4)) * s_k + r;  // the element-wise path wants the true rows
        off[it] = (uint32_t)((long)(idx / (R / 4)) * s_k + min(r, max(Rmax - 4, 0)));
      }
    }
  }
  // interior tile of a clean operand: R / 32 unchecked 16-byte loads off one uniform base.  Rows past the end were clamped
  // to existing ones: they only feed accumulator rows / columns that the epilogue never stores.
  __device__ __forceinline__ void load_fast(float (&reg)[R / 32][4], int k0) const {
    const float* __restrict__ b = x0 + (long)k0 * kstep;
#pragma unroll
    for (int it = 0; it < R / 32; ++it) {
      const f32x4 v = nt ? __builtin_nontemporal_load((const f32x4*)(b + off[it])) : *(const f32x4*)(b + off[it]);
      reg[it][0] = v[0], reg[it][1] = v[1], reg[it][2] = v[2], reg[it][3] = v[3];
    }
  }
  __device__ __forceinline__ void load(float (&reg)[R / 32][4], int k0, int K, int tid) const {
    const bool full = k0 + kBK <= K;  // uniform
#pragma unroll
    for (int it = 0; it < R / 32; ++it) {
      const float* q = p[it] + (long)k0 * kstep;
      if (KC) {
        if (vec && full) {
          const f32x4 v = *(const f32x4*)q;
          reg[it][0] = v[0], reg[it][1] = v[1], reg[it][2] = v[2], reg[it][3] = v[3];
        } else {
          const int k = k0 + ((it * 256 + tid) & 7) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) reg[it][j] = (nv[it] != 0 && k + j < K) ? q[j] : 0.0f;
        }
      } else {
        const bool kok = k0 + (it * 256 + tid) / (R / 4) < K;
        if (vec && full && nv[it] == 4) {
          const f32x4 v = *(const f32x4*)q;
          reg[it][0] = v[0], reg[it][1] = v[1], reg[it][2] = v[2], reg[it][3] = v[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) reg[it][j] = (kok && j < nv[it]) ? q[j] : 0.0f;
        }
      }
    }
  }
};
template <int R, int NP, bool KC>
__device__ __forceinline__ void store_tile(bf16* __restrict__ hi, bf16* __restrict__ lo, const float (&reg)[R / 32][4], int tid) {
#pragma unroll
  for (int it = 0; it < R / 32; ++it) {
    const int idx = it * 256 + tid;
    const f32x4 x = {reg[it][0], reg[it][1], reg[it][2], reg[it][3]};
    const bf16x4 h = __builtin_convertvector(x, bf16x4);
    bf16x4 l;
    if (NP == 3) l = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), bf16x4);
    const int off = KC ? kc_row(idx >> 3) * kLdk + (idx & 7) * 4 : (idx / (R / 4)) * TileGeo<R>::kLdt + (idx % (R / 4)) * 4;
    *(bf16x4*)(hi + off) = h;
    if (NP == 3) *(bf16x4*)(lo + off) = l;
  }
}
// MFMA operand fragment (8 consecutive k of row rbase + lane % 32, k half lane / 32) of k-step ks from a tile image
template <int R, bool KC>
__device__ __forceinline__ bf16x8 read_frag(const bf16* __restrict__ t, int rbase, int ks, int lane) {
  if (KC) return *(const bf16x8*)(t + (rbase + (lane & 31)) * kLdk + ks * 16 + (lane >> 5) * 8);
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const int i = lane & 15, gi = lane >> 4;
  const bf16* a = t + (ks * 16 + 8 * (gi >> 1) + (i >> 2)) * TileGeo<R>::kLdt + rbase + 16 * (gi & 1) + 4 * (i & 3);
  const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * TileGeo<R>::kLdt));
  return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// The same tile images read as operands of v_mfma_f32_16x16x32_bf16 (EECT_MFMA16; csrc/eec_device.h has the inference path's form of
// this): the chip holds a higher clock on that shape under load (profiles/r04_micro_mfma_shape_clock.txt; timing-only swap in this
// kernel: training step 25.5 -> 24.1 ms).  Lane l = 16 g + c holds row rbase + 16 rb + c, k = 8 g .. 8 g + 7 of the 32-deep tile.
#ifndef EECT_MFMA16
#define EECT_MFMA16 1
#endif
template <int R, bool KC>
__device__ __forceinline__ bf16x8 read_frag16(const bf16* __restrict__ t, int rbase, int rb, int lane) {
  const int c = lane & 15, g = lane >> 4;
  if (KC) return *(const bf16x8*)(t + (rbase + 16 * rb + c) * kLdk + g * 8);
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16* a = t + (8 * g + (c >> 2)) * TileGeo<R>::kLdt + rbase + 16 * rb + 4 * (c & 3);
  const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * TileGeo<R>::kLdt));
  return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// quadrant (ra, cb) of a 32 x 32 tile (registers 4 (2 ra + cb) ..): m = 16 ra + 4 (lane >> 4) + i, n = 16 cb + (lane & 15)
__device__ __forceinline__ void quad_mac16(f32x16& acc, int ra, int cb, bf16x8 a, bf16x8 b) {
  const int q = 4 * (2 * ra + cb);
  f32x4 t = {acc[q], acc[q + 1], acc[q + 2], acc[q + 3]};
  t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, t, 0, 0, 0);
  acc[q] = t[0], acc[q + 1] = t[1], acc[q + 2] = t[2], acc[q + 3] = t[3];
}
// quadrant layout -> the 32x32x16 layout the epilogues expect: v_permlane16_swap + v_permlane32_swap on the register pairs
// (4 (2 ra) + i, 4 (2 ra + 1) + i), one asm block per tile (derivation and checks: csrc/eec_device.h, tools/mfma16_gemm_check.hip)
#define EECT_SWAP8(OP)                                                                                                       \
  "v_permlane" OP "_swap_b32 %0, %4\n\tv_permlane" OP "_swap_b32 %1, %5\n\tv_permlane" OP "_swap_b32 %2, %6\n\t"            \
  "v_permlane" OP "_swap_b32 %3, %7\n\tv_permlane" OP "_swap_b32 %8, %12\n\tv_permlane" OP "_swap_b32 %9, %13\n\t"          \
  "v_permlane" OP "_swap_b32 %10, %14\n\tv_permlane" OP "_swap_b32 %11, %15\n\t"
__device__ __forceinline__ void acc_q_to_std(f32x16& acc) {
  float r0 = acc[0], r1 = acc[1], r2 = acc[2], r3 = acc[3], r4 = acc[4], r5 = acc[5], r6 = acc[6], r7 = acc[7];
  float r8 = acc[8], r9 = acc[9], r10 = acc[10], r11 = acc[11], r12 = acc[12], r13 = acc[13], r14 = acc[14], r15 = acc[15];
  asm volatile("s_nop 1\n\t" EECT_SWAP8("16") EECT_SWAP8("32") "s_nop 1"
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(r8), "+v"(r9), "+v"(r10), "+v"(r11),
                 "+v"(r12), "+v"(r13), "+v"(r14), "+v"(r15));
  acc = (f32x16){r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15};
}

// f(IntTag<0>{}), ..., f(IntTag<N - 1>{}): a loop whose index is a compile-time constant in the body (register arrays indexed by
// it never fall back to scratch memory, whatever the unroller decides)
template <int I>
struct IntTag {
  static constexpr int value = I;
};
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntTag<I>{});
    static_for<I + 1, N>(f);
  }
}

template <bool V>
struct FastTag {
  static constexpr bool value = V;
};

// epilogue variants: C = v;  EPI_SILU: also C2 = drop(silu(v));  EPI_DSILU: C = v * dropmask * silu'(aux)
// EPI_RELU_DROP: C = v, C2 = drop(relu(v));  EPI_DRELU: C = v * dropmask * (aux > 0)   (the ReLU feed-forward of the AED decoder)
enum { EPI_NONE = 0, EPI_SILU = 1, EPI_DSILU = 2, EPI_RELU = 3, EPI_RESID = 4, EPI_RELU_DROP = 5, EPI_DRELU = 6 };

// STAGES: register prefetch depth of the k-loop.  2: the loads of tile kt + 2 are in flight across two MFMA phases (long
// contractions); 1: 64 registers fewer, so that three workgroups share a CU (short contractions, many tiles: +25-50 %)
#ifndef EECT_PIPE_SCHED
#define EECT_PIPE_SCHED 7  // STAGES == 3 only: VALU instructions placed behind every MFMA of the interval (0: the compiler's order)
#endif
#ifndef EECT_EPI_DIRECT
#define EECT_EPI_DIRECT 0  // 1: 4-byte stores straight from the accumulators (measured: epilogue 23.5 k cycles against 13.6 k through the slabs)
#endif
#ifdef EECT_TL
// Diagnostic build only (tools/train_gemm_timeline.py): s_memtime stamps of thread 0 of the first 64 workgroups of a launch.
__device__ unsigned long long eect_tl_buf[64 * 16];
extern "C" int eect_debug_tl(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eect_tl_buf), sizeof(eect_tl_buf));
}
#define EECT_STAMP(i)                                                                                        \
  do {                                                                                                       \
    const unsigned l_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                      \
    if (threadIdx.x == 0 && l_ < 512 && (l_ & 7) == 0) eect_tl_buf[(l_ >> 3) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define EECT_STAMP(i)
#endif

// EPI_T selects what the epilogue is compiled for: kEpiPlain (bias, alpha, accumulate, ReLU), kEpiAny (every per-element epilogue
// behind run-time switches: the fallback), or one of the EPI_* values alone.  With every variant in one body a 32-row slab's epilogue
// is 5.6 k instructions (134 KB of code per tile against a 64 KB instruction cache shared by two CUs): a plain K = 256 GEMM spent
// 30 k of its workgroup's 65 k cycles there (tools/train_gemm_timeline.py), 13.6 k in a body of its own.  The hot per-element
// epilogues (SiLU second output, SiLU' x mask, residual + dropout) get bodies of their own for the tile shapes and layouts the
// training plan uses them with.
constexpr int kEpiPlain = -2, kEpiAny = -1;
// LDS of one workgroup: the operand plane sets of the k-loop, reused as a [32][BN + 4] fp32 slab of the output tile in the epilogue
template <int TM, int TN, int WGM, int WGN, int NP, int STAGES>
constexpr int gemm_lds_bytes() {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int planes = (STAGES == 3 ? 2 : 1) * (NP == 3 ? 2 : 1) * (TileGeo<BM>::kElems + TileGeo<BN>::kElems) * 2;
  constexpr int slab = 32 * (BN + 4) * 4, red = 4 * 256 * 4;  // red: the row-sum reduction's [256 / (BM / 4)][BM] floats
  return planes > slab ? (planes > red ? planes : red) : (slab > red ? slab : red);
}
template <int TM, int TN, int WGM, int WGN, int NP, bool AKC, bool BKC, int STAGES, int EPI_T>
__global__ __launch_bounds__(256, STAGES == 1 ? 3 : 2) void gemm_kernel(GemmArgs g) {
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  // one LDS block: the operand planes during the k-loop, a [32][BN + 4] fp32 slab of the output tile in the epilogue
  constexpr int EA = TileGeo<BM>::kElems, EB = TileGeo<BN>::kElems, kSlabLd = BN + 4;
  constexpr int kPlaneBytes = (NP == 3 ? 2 : 1) * (EA + EB) * 2;
  // STAGES == 3: two plane sets, one barrier per k-tile (below); the block is dynamic LDS (gemm_lds_bytes: 80 KB at 128 x 128 x 3)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* a_hi = (bf16*)smem;
  bf16* b_hi = a_hi + EA;
  bf16* a_lo = NP == 3 ? b_hi + EB : a_hi;
  bf16* b_lo = NP == 3 ? a_lo + EA : b_hi;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w / WGN, wn = w % WGN;
  EECT_STAMP(0);
  // XCD-aware tile map.  Workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest, then y, z), each XCD with
  // its own L2: taken naively, the column blocks of one row tile -- which all read the same A rows -- land on eight different
  // L2s and every one of them fetches those rows (measured on the default model: 98 GB of L2-miss reads per training step).
  // Instead XCD x owns the CONTIGUOUS range [x, x + 1) * total / 8 of the tile order, so that tiles sharing operands share an L2.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned nbx = gridDim.x, nby = gridDim.y, total = nbx * nby * gridDim.z;
    if ((total & 7u) == 0) {
      const unsigned l = bx + nbx * (by + nby * bz), t = (l & 7u) * (total >> 3) + (l >> 3);
      bx = t % nbx, by = (t / nbx) % nby, bz = t / (nbx * nby);
    }
  }
  const int z0 = bz / g.zdiv, z1 = bz % g.zdiv;
  const float* __restrict__ A = g.A + z0 * g.a_z0 + z1 * g.a_z1;
  const float* __restrict__ B = g.B + z0 * g.b_z0 + z1 * g.b_z1;
  float* __restrict__ C = g.C + z0 * g.c_z0 + z1 * g.c_z1;
  const int m0 = by * BM, n0 = bx * BN;
  const long a_r = AKC ? g.a_m : 1, a_s = AKC ? 1 : g.a_k;  // (row stride, k stride) as load_tile wants them
  const long b_r = BKC ? g.b_n : 1, b_s = BKC ? 1 : g.b_k;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  // two register stages: the global loads of tile kt + 2 are issued as soon as stage (kt & 1) has been written to LDS, so
  // they have the MFMA phases of two k-tiles to land (one phase does not cover the L2 / HBM latency)
  float ra0[BM / 32][4], rb0[BN / 32][4], ra1[BM / 32][4], rb1[BN / 32][4];
  const int K = g.ktot > 0 ? min(g.K, g.ktot - z0 * g.K) : g.K;
  const int nk = (K + kBK - 1) / kBK;
  TileLoader<BM, AKC> la;
  TileLoader<BN, BKC> lb;
  la.init(A, a_r, a_s, m0, g.M, tid);
  lb.init(B, b_r, b_s, n0, g.N, tid);
  la.nt = g.stream_a != 0, lb.nt = g.stream_b != 0;
  // k-tiles are walked in a rotated order that differs between neighbouring workgroups: with power-of-two leading
  // dimensions the rows of a k-contiguous tile all fall on the same few L2 / HBM channels for a given k offset, and
  // workgroups in lockstep would all hit those at once (measured: 101 -> TFLOP/s class of the transposed layouts)
  const int rot = (int)((bx * 3 + by + bz) % (unsigned)nk);
  auto k_of = [&](int kt) __attribute__((always_inline)) { const int t = kt + rot; return (t >= nk ? t - nk : t) * kBK; };
  // fast: every tile of both operands is a full, aligned interior tile -- the loop then carries no bounds logic at all
  const bool fast = la.clean && lb.clean && (K % kBK) == 0;
  auto load_ab = [&](float (&ra)[BM / 32][4], float (&rb)[BN / 32][4], int k0, auto fast_tag) __attribute__((always_inline)) {
    if (decltype(fast_tag)::value) {
      la.load_fast(ra, k0);
      lb.load_fast(rb, k0);
    } else {
      la.load(ra, k0, K, tid);
      lb.load(rb, k0, K, tid);
    }
  };
  // optional row sums of A (first column block only): every A element passes through this thread's registers exactly once
  // (its BM / 32 groups of four all belong to the same four rows: 256 % (BM / 4) == 0)
  float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool do_rs = !AKC && g.rowsum != nullptr && bx == 0;
  constexpr int kBufElems = kPlaneBytes / 2;  // bf16 elements between the two plane sets
  // registers -> (split) -> plane set `buf`
  auto split_store = [&](float (&ra)[BM / 32][4], float (&rb)[BN / 32][4], int buf) __attribute__((always_inline)) {
    if constexpr (!AKC) {
      if (do_rs) {
#pragma unroll
        for (int it = 0; it < BM / 32; ++it)
#pragma unroll
          for (int j = 0; j < 4; ++j) rs[j] += ra[it][j];
      }
    }
    store_tile<BM, NP, AKC>(a_hi + buf * kBufElems, a_lo + buf * kBufElems, ra, tid);
    store_tile<BN, NP, BKC>(b_hi + buf * kBufElems, b_lo + buf * kBufElems, rb, tid);
  };
  // the 32-deep k-tile in plane set `buf`: fragments from LDS, 4 x TM x TN x (1 or 3) MFMAs
#ifndef EECT_MFMA16_KC_ONLY
#define EECT_MFMA16_KC_ONLY 0  // 1: the 16x16x32 form only where both operands are k-contiguous (no spills there)
#endif
  constexpr bool kM16 = EECT_MFMA16 && EECT_EPI_DIRECT != 2 && (!EECT_MFMA16_KC_ONLY || (AKC && BKC));
  auto mfma_tile = [&](int buf) __attribute__((always_inline)) {
#if EECT_MFMA16 && EECT_EPI_DIRECT != 2
    if constexpr (kM16)
    // 16x16x32 form: the 32-deep tile is ONE k-step.  Row block ra of A (fragments of every row tile, hi / lo) against row block cb of
    // B, in the order (0,0) (0,1) (1,1) (1,0) so that a B fragment set is read three times, not four; the same eight fragment
    // registers as a 16-deep step of the 32x32x16 form.
    {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
      auto load_a = [&](int ra) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
          ah[mt] = read_frag16<BM, AKC>(a_hi + buf * kBufElems, (wm * TM + mt) * 32, ra, lane);
          if (NP == 3) al[mt] = read_frag16<BM, AKC>(a_lo + buf * kBufElems, (wm * TM + mt) * 32, ra, lane);
        }
      };
      auto load_b = [&](int cb) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          bh[nt] = read_frag16<BN, BKC>(b_hi + buf * kBufElems, (wn * TN + nt) * 32, cb, lane);
          if (NP == 3) bl[nt] = read_frag16<BN, BKC>(b_lo + buf * kBufElems, (wn * TN + nt) * 32, cb, lane);
        }
      };
      auto macs = [&](int ra, int cb) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt) {
            if (NP == 3) {
              quad_mac16(acc[mt][nt], ra, cb, al[mt], bh[nt]);
              quad_mac16(acc[mt][nt], ra, cb, ah[mt], bl[nt]);
            }
            quad_mac16(acc[mt][nt], ra, cb, ah[mt], bh[nt]);
          }
      };
      // (fenced: left alone, hipcc hoists every fragment read of the tile in front of the first MFMA -- 20 fragments live at once)
      load_a(0);
      load_b(0);
      __builtin_amdgcn_sched_barrier(0);
      macs(0, 0);
      load_b(1);
      __builtin_amdgcn_sched_barrier(0);
      macs(0, 1);
      load_a(1);
      __builtin_amdgcn_sched_barrier(0);
      macs(1, 1);
      load_b(0);
      __builtin_amdgcn_sched_barrier(0);
      macs(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
        ah[mt] = read_frag<BM, AKC>(a_hi + buf * kBufElems, (wm * TM + mt) * 32, ks, lane);
        if (NP == 3) al[mt] = read_frag<BM, AKC>(a_lo + buf * kBufElems, (wm * TM + mt) * 32, ks, lane);
      }
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        bh[nt] = read_frag<BN, BKC>(b_hi + buf * kBufElems, (wn * TN + nt) * 32, ks, lane);
        if (NP == 3) bl[nt] = read_frag<BN, BKC>(b_lo + buf * kBufElems, (wn * TN + nt) * 32, ks, lane);
      }
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
#if EECT_EPI_DIRECT == 2
          if (NP == 3) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nt], al[mt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[nt], ah[mt], acc[mt][nt], 0, 0, 0);
          }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nt], ah[mt], acc[mt][nt], 0, 0, 0);
#else
          if (NP == 3) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#endif
        }
    }
  };
  auto k_tile = [&](int kt, float (&ra)[BM / 32][4], float (&rb)[BN / 32][4], auto fast_tag) __attribute__((always_inline)) {
    split_store(ra, rb, 0);
    __syncthreads();
#ifdef EECT_TL
    if (kt < 4) EECT_STAMP(4 + 2 * kt);
#endif
    if (kt + STAGES < nk) load_ab(ra, rb, k_of(kt + STAGES), fast_tag);
    mfma_tile(0);
#ifdef EECT_TL
    if (kt < 4) EECT_STAMP(5 + 2 * kt);
#endif
    __syncthreads();
  };
  // STAGES == 3, the pipelined form: two plane sets in LDS and two register sets.  While the MFMAs run on plane set kt % 2 the
  // same wave splits tile kt + 1 (already in registers) into the other set, and tile kt + 2's loads are in flight: ONE barrier per
  // k-tile, and the split's VALU work sits in the shadow of the MFMAs of the same wave instead of in a phase of its own.
  // steady_tag: both tiles ahead exist -- no conditions, so that the interval is ONE basic block (the scheduling hints below only
  // reorder inside one)
  auto pipe_step = [&](int kt, int buf, float (&r_next_a)[BM / 32][4], float (&r_next_b)[BN / 32][4], float (&r_load_a)[BM / 32][4],
                       float (&r_load_b)[BN / 32][4], auto fast_tag, auto steady_tag) __attribute__((always_inline)) {
    constexpr bool steady = decltype(steady_tag)::value;
    if (steady || kt + 2 < nk) load_ab(r_load_a, r_load_b, k_of(kt + 2), fast_tag);  // into the set whose tile went to LDS a step ago
    mfma_tile(buf);
    if (steady || kt + 1 < nk) split_store(r_next_a, r_next_b, buf ^ 1);
#if EECT_PIPE_SCHED
    // the interval as one interleaved stream: fragment reads of a k-step, then its MFMAs with the split's VALU work and LDS stores
    // in their shadows (hipcc otherwise emits the MFMAs and the split as two blocks)
    {
      constexpr int kMfma = 2 * TM * TN * (NP == 3 ? 3 : 1), kReads = (NP == 3 ? 2 : 1) * (TM + TN) * ((AKC ? 1 : 2) + (BKC ? 1 : 2)) / 2;
      constexpr int kValuPer = EECT_PIPE_SCHED;
      __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
#pragma unroll
      for (int i = 0; i < kMfma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kValuPer, 0);
        if (i & 1) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        if (i == kMfma / 4) __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
      }
    }
#endif
#ifdef EECT_TL
    if (kt < 4) EECT_STAMP(4 + 2 * kt), EECT_STAMP(5 + 2 * kt);
#endif
    __syncthreads();  // set buf ^ 1 complete; everybody done with set buf
  };
  auto k_loop = [&](auto fast_tag) __attribute__((always_inline)) {
    load_ab(ra0, rb0, k_of(0), fast_tag);
    if constexpr (STAGES == 1) {
      for (int kt = 0; kt < nk; ++kt) k_tile(kt, ra0, rb0, fast_tag);
    } else if constexpr (STAGES == 3) {
      if (nk > 1) load_ab(ra1, rb1, k_of(1), fast_tag);
      split_store(ra0, rb0, 0);
      __syncthreads();
      int kt = 0;
      for (; kt + 3 < nk; kt += 2) {
        pipe_step(kt, 0, ra1, rb1, ra0, rb0, fast_tag, FastTag<true>{});
        pipe_step(kt + 1, 1, ra0, rb0, ra1, rb1, fast_tag, FastTag<true>{});
      }
      for (; kt < nk; kt += 2) {
        pipe_step(kt, 0, ra1, rb1, ra0, rb0, fast_tag, FastTag<false>{});
        if (kt + 1 < nk) pipe_step(kt + 1, 1, ra0, rb0, ra1, rb1, fast_tag, FastTag<false>{});
      }
    } else {
      if (nk > 1) load_ab(ra1, rb1, k_of(1), fast_tag);
      for (int kt = 0; kt < nk; kt += 2) {
        k_tile(kt, ra0, rb0, fast_tag);
        if (kt + 1 < nk) k_tile(kt + 1, ra1, rb1, fast_tag);
      }
    }
  };
  EECT_STAMP(1);
  if (fast) k_loop(FastTag<true>{});
  else k_loop(FastTag<false>{});
#if EECT_MFMA16 && EECT_EPI_DIRECT != 2
  // the k-loop kept the accumulators in the quadrant layout: back to the layout of the epilogues, once.  (The swaps are inline asm:
  // the MFMA-result -> VALU-read hazard in front of them is padded by hand -- 19 wait states cover a 16-pass MFMA.)
  if constexpr (kM16) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 2" ::: "memory");
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc_q_to_std(acc[i][j]);
  }
#endif
  EECT_STAMP(2);
  if constexpr (!AKC) {
    if (do_rs) {  // uniform over the workgroup.  The k-loop ended with a barrier: LDS is free.
      constexpr int TPG = 256 / (BM / 4);  // threads that hold partial sums of the same four rows
      float* red = (float*)smem;            // [TPG][BM]
      const int rg = tid % (BM / 4), part = tid / (BM / 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) red[part * BM + 4 * rg + j] = rs[j];
      __syncthreads();
      if (tid < BM && m0 + tid < g.M) {
        float t = 0.0f;
#pragma unroll
        for (int p2 = 0; p2 < TPG; ++p2) t += red[p2 * BM + tid];
        g.rowsum[z0 * g.rowsum_z + m0 + tid] = t;
      }
      __syncthreads();
    }
  }
  // epilogue: the output tile leaves through LDS in slabs of 32 rows, so that every thread stores 16 contiguous bytes of a row
  // (an accumulator lane holds single columns: direct stores are 4-byte stores, four times as many instructions) and the
  // epilogue arithmetic runs on float4s.  The k-loop ended with a barrier: the planes are dead.
  const int epi = EPI_T >= 0 ? EPI_T : EPI_T == kEpiAny ? g.epi : (g.epi == EPI_RELU ? EPI_RELU : EPI_NONE);
  const DropState ds(EPI_T != kEpiPlain ? g.drop : Drop{0.0f, 0, 0});
  const float* __restrict__ aux = g.aux;
  float* __restrict__ C2 = g.C2;
  const bool cvec = ((((uintptr_t)C) | ((uintptr_t)C2) | ((uintptr_t)aux)) & 15) == 0 && (g.c_m & 3) == 0 && (g.N & 3) == 0;
  const bool nt = g.stream_out != 0;  // uniform
  auto store4 = [&](float* p, f32x4 v) __attribute__((always_inline)) {
    if (nt) __builtin_nontemporal_store(v, (f32x4*)p);
    else *(f32x4*)p = v;
  };
  auto load4 = [&](const float* p) __attribute__((always_inline)) {
    return nt ? __builtin_nontemporal_load((const f32x4*)p) : *(const f32x4*)p;
  };
  // one group of four consecutive columns of row m: v = alpha * acc + bias, the epilogue arithmetic, 16-byte stores
  auto finish4 = [&](int m, int n, f32x4 a4, f32x4 pre4) __attribute__((always_inline)) {
    const long ci = (long)m * g.c_m + n;
    const int nvalid = min(4, g.N - n);
    float v[4], v2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = g.alpha * a4[j] + ((g.bias && j < nvalid) ? g.bias[n + j] : 0.0f);
    if (cvec) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += g.accumulate ? pre4[j] : 0.0f;
      float dm[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      if (epi == EPI_DSILU || epi == EPI_DRELU || epi == EPI_RESID || epi == EPI_SILU || epi == EPI_RELU_DROP) ds.mul4((uint64_t)ci, dm);
      if (epi == EPI_DSILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = pre4[j], sg = sigmoidf_(x);
          v[j] *= dm[j] * sg * (1.0f + x * (1.0f - sg));
        }
      }
      if (epi == EPI_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
      }
      if (epi == EPI_DRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= pre4[j] > 0.0f ? dm[j] : 0.0f;
      }
      if (epi == EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = pre4[j] + g.res_scale * v[j] * dm[j];
      }
      store4(C + ci, (f32x4){v[0], v[1], v[2], v[3]});
      if (epi == EPI_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v2[j] = v[j] * sigmoidf_(v[j]) * dm[j];
        store4(C2 + ci, (f32x4){v2[0], v2[1], v2[2], v2[3]});
      }
      if (epi == EPI_RELU_DROP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v2[j] = fmaxf(v[j], 0.0f) * dm[j];
        store4(C2 + ci, (f32x4){v2[0], v2[1], v2[2], v2[3]});
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < nvalid) {
          float t = v[j];
          if (g.accumulate) t += C[ci + j];
          if (epi == EPI_DSILU) {
            const float x = aux[ci + j], sg = sigmoidf_(x);
            t *= ds.mul((uint64_t)(ci + j)) * sg * (1.0f + x * (1.0f - sg));
          }
          if (epi == EPI_RELU) t = fmaxf(t, 0.0f);
          if (epi == EPI_DRELU) t *= aux[ci + j] > 0.0f ? ds.mul((uint64_t)(ci + j)) : 0.0f;
          if (epi == EPI_RESID) t = aux[ci + j] + g.res_scale * t * ds.mul((uint64_t)(ci + j));
          C[ci + j] = t;
          if (epi == EPI_SILU) C2[ci + j] = t * sigmoidf_(t) * ds.mul((uint64_t)(ci + j));
          if (epi == EPI_RELU_DROP) C2[ci + j] = fmaxf(t, 0.0f) * ds.mul((uint64_t)(ci + j));
        }
      }
    }
  };
#if EECT_EPI_DIRECT == 2
  // operands swapped in the k-loop: the accumulators hold the TRANSPOSED 32 x 32 tiles, i.e. lane l has row m = l % 32 and, per
  // register quad q, the four consecutive columns 8 q + 4 (l / 32) ..: 16-byte stores straight from the registers, no LDS, no barrier
  {
    const bool want_pre = cvec && (g.accumulate || epi == EPI_DSILU || epi == EPI_RESID || epi == EPI_DRELU);
    const float* __restrict__ pre_src = g.accumulate ? (const float*)C : aux;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int m = m0 + (wm * TM + mt) * 32 + (lane & 31), nb = n0 + (wn * TN + nt) * 32 + 4 * (lane >> 5);
        f32x4 pre4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          pre4[q] = (want_pre && m < g.M && nb + 8 * q < g.N) ? *(const f32x4*)(pre_src + (long)m * g.c_m + nb + 8 * q) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q;
          if (m < g.M && n < g.N)
            finish4(m, n, (f32x4){acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]}, pre4[q]);
        }
      }
  }
#elif EECT_EPI_DIRECT
  // straight from the accumulators: for a fixed register index the 64 lanes of a wave hold two rows x 32 consecutive columns,
  // i.e. every store (and every load of old C / aux) instruction covers two full 128-byte segments -- no LDS round trip and no
  // barrier (the slab form below spent 13.6 k of a K = 256 workgroup's 47 k cycles on its 8 barriers)
  {
    const bool want_pre = g.accumulate || epi == EPI_DSILU || epi == EPI_RESID || epi == EPI_DRELU;
    const float* __restrict__ pre_src = g.accumulate ? (const float*)C : aux;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int n = n0 + (wn * TN + nt) * 32 + (lane & 31), mb = m0 + (wm * TM + mt) * 32;
        const bool nok = n < g.N;
        const float bv = (g.bias && nok) ? g.bias[n] : 0.0f;
        float pre[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = mb + acc_row(i, lane);
          pre[i] = (want_pre && nok && m < g.M) ? pre_src[(long)m * g.c_m + n] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = mb + acc_row(i, lane);
          if (nok && m < g.M) {
            const long ci = (long)m * g.c_m + n;
            float t = g.alpha * acc[mt][nt][i] + bv;
            if (g.accumulate) t += pre[i];
            if (epi == EPI_DSILU) {
              const float x = pre[i], sg = sigmoidf_(x);
              t *= ds.mul((uint64_t)ci) * sg * (1.0f + x * (1.0f - sg));
            }
            if (epi == EPI_RELU) t = fmaxf(t, 0.0f);
            if (epi == EPI_DRELU) t *= pre[i] > 0.0f ? ds.mul((uint64_t)ci) : 0.0f;
            if (epi == EPI_RESID) t = pre[i] + g.res_scale * t * ds.mul((uint64_t)ci);
            C[ci] = t;
            if (epi == EPI_SILU) C2[ci] = t * sigmoidf_(t) * ds.mul((uint64_t)ci);
            if (epi == EPI_RELU_DROP) C2[ci] = fmaxf(t, 0.0f) * ds.mul((uint64_t)ci);
          }
        }
      }
  }
#else
  float* slab = (float*)smem;
  // what a slab's elements need from memory -- old C (accumulate) or aux (EPI_DSILU), never both -- is requested one slab
  // ahead: the loads of slab s + 1 are in flight while slab s goes through LDS and out
  constexpr int NPS = 32 * (BN / 4) / 256;
  f32x4 pre[BM / 32][NPS];
  const bool want_pre = cvec && (g.accumulate || epi == EPI_DSILU || epi == EPI_RESID || epi == EPI_DRELU);
  const float* __restrict__ pre_src = g.accumulate ? (const float*)C : aux;
  auto request = [&](auto sl_tag) __attribute__((always_inline)) {
    constexpr int sl = decltype(sl_tag)::value;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int idx = ps * 256 + tid, row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
      const int m = m0 + sl * 32 + row, n = n0 + c4;
      pre[sl][ps] = (want_pre && m < g.M && n < g.N) ? load4(pre_src + (long)m * g.c_m + n) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  request(IntTag<0>{});
  static_for<0, BM / 32>([&](auto sl_tag) __attribute__((always_inline)) {
    constexpr int sl = decltype(sl_tag)::value;
    if constexpr (sl + 1 < BM / 32) request(IntTag<sl + 1>{});
    if (wm == sl / TM) {
#pragma unroll
      for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) slab[acc_row(i, lane) * kSlabLd + (wn * TN + nt) * 32 + (lane & 31)] = acc[sl % TM][nt][i];
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int idx = ps * 256 + tid, row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
      const int m = m0 + sl * 32 + row, n = n0 + c4;
      if (m < g.M && n < g.N) finish4(m, n, *(const f32x4*)(slab + row * kSlabLd + c4), pre[sl][ps]);
    }
    __syncthreads();
  });
#endif
  EECT_STAMP(3);
}

template <int TM, int TN, int WGM, int WGN>
static hipError_t launch_gemm_t(const GemmArgs& g, int np, hipStream_t st) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  const dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.nz);
  const bool akc = g.a_k == 1, bkc = g.b_k == 1;
#ifndef EECT_PIPE_MIN_K
#define EECT_PIPE_MIN_K 0  // > 0: also build the pipelined loop and use it for contractions at least this long (EEC_TRAIN_PIPE_MIN_K overrides)
#endif
#ifndef EECT_STAGES
#define EECT_STAGES 1  // measured on one box, default model: 1 -> 38.9 ms per step, 2 -> 40.6 ms, 2 for >= 32 k-tiles only -> 39.8 ms
#endif
  const bool fancy = g.epi != EPI_NONE && g.epi != EPI_RELU;
  // bodies of their own: (epilogue, tile, layout) as the encoder's training plan launches them
  constexpr bool big = TM == 2 && TN == 2 && WGM == 2 && WGN == 2, wide = TM == 2 && TN == 1 && WGM == 2 && WGN == 2;
  // the pipelined loop (STAGES 3: one barrier per k-tile, the split in the MFMAs' shadow, two workgroups per CU) is faster for long
  // contractions ALONE (same-box A/B, bf16x3: K = 2048 87.6 -> 74.4 us, one workgroup per CU 330 -> 220 us; K = 256 75 -> 81 us) but
  // buys nothing in the step, where the long GEMMs (weight gradients, side stream) share the chip with the dX chain: 25.2 vs 25.3 ms.
  // Built only with -DEECT_PIPE_MIN_K=<K>.
#if EECT_PIPE_MIN_K > 0
  static const int pipe_min_k = [] { const char* e = getenv("EEC_TRAIN_PIPE_MIN_K"); return e ? atoi(e) : EECT_PIPE_MIN_K; }();  // tuning knob
  const bool pipe = EECT_STAGES == 1 && g.K >= pipe_min_k;
#endif
#define EECT_GEMM_S(NP, AK, BK, E, ST)                                                                          \
  do {                                                                                                          \
    auto kfn = gemm_kernel<TM, TN, WGM, WGN, NP, AK, BK, ST, E>;                                                \
    constexpr int lds = gemm_lds_bytes<TM, TN, WGM, WGN, NP, ST>();                                             \
    if (lds > 65536) {                                                                                          \
      static bool raised = false; /* per instantiation */                                                       \
      if (!raised) {                                                                                            \
        if (hipError_t e_ = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); e_ != hipSuccess) return e_; \
        raised = true;                                                                                          \
      }                                                                                                         \
    }                                                                                                           \
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, g);                                                       \
  } while (0)
#if EECT_PIPE_MIN_K > 0
#define EECT_GEMM_E(NP, AK, BK, E)                     \
  do {                                                 \
    if (pipe) EECT_GEMM_S(NP, AK, BK, E, 3);           \
    else EECT_GEMM_S(NP, AK, BK, E, EECT_STAGES);      \
  } while (0)
#else
#define EECT_GEMM_E(NP, AK, BK, E) EECT_GEMM_S(NP, AK, BK, E, EECT_STAGES)
#endif
  if constexpr (big || wide) {
    if (akc && bkc && g.epi == EPI_RESID) {
      if (np == 1) EECT_GEMM_E(1, true, true, EPI_RESID);
      else EECT_GEMM_E(3, true, true, EPI_RESID);
      return hipGetLastError();
    }
  }
  if constexpr (big) {
    if (akc && bkc && g.epi == EPI_SILU) {
      if (np == 1) EECT_GEMM_E(1, true, true, EPI_SILU);
      else EECT_GEMM_E(3, true, true, EPI_SILU);
      return hipGetLastError();
    }
    if (akc && !bkc && g.epi == EPI_DSILU) {
      if (np == 1) EECT_GEMM_E(1, true, false, EPI_DSILU);
      else EECT_GEMM_E(3, true, false, EPI_DSILU);
      return hipGetLastError();
    }
  }
#define EECT_GEMM(NP, AK, BK)                          \
  do {                                                 \
    if (fancy) EECT_GEMM_E(NP, AK, BK, kEpiAny);       \
    else EECT_GEMM_E(NP, AK, BK, kEpiPlain);           \
  } while (0)
  if (np == 1) {
    if (akc && bkc) EECT_GEMM(1, true, true);
    else if (akc) EECT_GEMM(1, true, false);
    else if (bkc) EECT_GEMM(1, false, true);
    else EECT_GEMM(1, false, false);
  } else {
    if (akc && bkc) EECT_GEMM(3, true, true);
    else if (akc) EECT_GEMM(3, true, false);
    else if (bkc) EECT_GEMM(3, false, true);
    else EECT_GEMM(3, false, false);
  }
#undef EECT_GEMM
#undef EECT_GEMM_E
#undef EECT_GEMM_S
  return hipGetLastError();
}
hipError_t launch_gemm(const GemmArgs& g_in, int np, hipStream_t st) {
  GemmArgs g = g_in;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.nz <= 0) return hipSuccess;
  static const long nt_bytes = [] { const char* e = getenv("EEC_TRAIN_NT_MB"); return (e ? atol(e) : 32L) << 20; }();
  g.stream_out = nt_bytes > 0 && g.nz == 1 && (long)g.M * g.N * 4 >= nt_bytes;
  // an operand of >= EEC_TRAIN_NT_IN_MB (default 64; 0 = never) that the launch reads once -- the [M, F] operand of a feed-forward weight
  // gradient against the [M, D] one its 16 column tiles share -- is loaded with the non-temporal hint (same-box A/B of the step: -0.5 %)
  static const long nt_in = [] { const char* e = getenv("EEC_TRAIN_NT_IN_MB"); return (e ? atol(e) : 64L) << 20; }();
  if (nt_in > 0) {
    const long ktot = g.ktot > 0 ? g.ktot : g.K;
    g.stream_a = (long)g.M * ktot * 4 >= nt_in && (long)g.N * ktot * 4 < nt_in;
    g.stream_b = (long)g.N * ktot * 4 >= nt_in && (long)g.M * ktot * 4 < nt_in;
  }
  if ((g.a_m != 1 && g.a_k != 1) || (g.b_n != 1 && g.b_k != 1)) return hipErrorInvalidValue;
  if (g.epi != EPI_NONE && (g.nz != 1 || g.accumulate)) return hipErrorInvalidValue;  // the epilogues index C as one [M][N] matrix
  if (g.rowsum && (g.a_m != 1 || g.zdiv != 1)) return hipErrorInvalidValue;             // row sums: A row-contiguous, batch = splits
  auto wgs = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.nz; };
  // the largest tile that still gives every CU two workgroups
  if (g.N <= 32) return launch_gemm_t<1, 1, 4, 1>(g, np, st);  // 128 x 32
  if (g.N > 64 && wgs(128, 128) >= 512) return launch_gemm_t<2, 2, 2, 2>(g, np, st);
  if (wgs(128, 64) >= 512 || g.N <= 64) return launch_gemm_t<2, 1, 2, 2>(g, np, st);
  if (wgs(128, 128) >= 384) return launch_gemm_t<2, 2, 2, 2>(g, np, st);
  return launch_gemm_t<1, 1, 2, 2>(g, np, st);  // 64 x 64
}

#ifdef EECT_TL
// Diagnostic build only: one [M][N] = A . B^T launch with a per-element epilogue (tools/train_gemm_timeline.py)
extern "C" int eect_debug_gemm_epi(const float* A, const float* B, float* C, const float* aux, float* C2, int M, int N, int K, int epi, int b_transposed,
                                   float p, void* stream) {
  GemmArgs g = gemm_args(A, K, 1, B, b_transposed ? 1 : K, b_transposed ? N : 1, C, N, M, N, K);
  g.epi = epi, g.aux = aux, g.C2 = C2, g.drop = Drop{p, 1234, 7}, g.res_scale = 0.5f;
  return (int)launch_gemm(g, 3, (hipStream_t)stream);
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kLnMax = 16;  // elements per lane: D <= 1024

// NE = elements per lane (D <= 64 * NE): one wave per row
template <int NE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                     float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long)row * D;
  float v[NE], s = 0.0f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < D ? xr[c] : 0.0f;
    s += v[i];
  }
  const float mu = wave_sum(s) / D;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int c = lane + 64 * i;
    const float d = c < D ? v[i] - mu : 0.0f;
    q += d * d;
  }
  const float rs = rsqrtf(wave_sum(q) / D + 1e-5f);
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int c = lane + 64 * i;
    if (c < D) y[(long)row * D + c] = (v[i] - mu) * rs * g[c] + b[c];
  }
  if (lane == 0) mean[row] = mu, rstd[row] = rs;
}
hipError_t launch_ln_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, int M, int D, hipStream_t st) {
  if (D > 64 * kLnMax) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4);
  if (D <= 256) hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, dim3(256), 0, st, x, g, b, y, mean, rstd, M, D);
  else if (D <= 512) hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, dim3(256), 0, st, x, g, b, y, mean, rstd, M, D);
  else hipLaunchKernelGGL(ln_fwd_kernel<16>, grid, dim3(256), 0, st, x, g, b, y, mean, rstd, M, D);
  return hipGetLastError();
}

#ifndef EECT_LN_BLOCKS
#define EECT_LN_BLOCKS 1024
#endif
int ln_bwd_blocks(int M) { return max(1, min(EECT_LN_BLOCKS, (M + 7) / 8)); }
template <int NE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ part, int M, int D) {
  __shared__ float red[4][2][64 * NE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rpb = (M + gridDim.x - 1) / gridDim.x, r_begin = blockIdx.x * rpb, r_end = min(M, r_begin + rpb);
  float gam[NE], dg[NE], db[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int c = lane + 64 * i;
    gam[i] = c < D ? g[c] : 0.0f, dg[i] = 0.0f, db[i] = 0.0f;
  }
  // a wave walks its rows one after the other and every row ends in two wave reductions: the next row's operands are
  // requested before the current row is reduced, or each row costs a full trip to memory
  struct Row {
    float x[NE], dy[NE], r[NE], mu, rs;
  };
  auto fetch = [&](int row, Row& o) __attribute__((always_inline)) {
    o.mu = mean[row], o.rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < D;
      o.x[i] = ok ? x[(long)row * D + c] : 0.0f;
      o.dy[i] = ok ? dy[(long)row * D + c] : 0.0f;
      o.r[i] = (ok && dres) ? dres[(long)row * D + c] : 0.0f;
    }
  };
  Row cur{}, nxt{};
  int row = r_begin + w;
  if (row < r_end) fetch(row, cur);
  for (; row < r_end; row += 4) {
    if (row + 4 < r_end) fetch(row + 4, nxt);
    const float mu = cur.mu, rs = cur.rs;
    float xh[NE], dyv[NE], c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const bool ok = lane + 64 * i < D;
      xh[i] = ok ? (cur.x[i] - mu) * rs : 0.0f;
      dyv[i] = cur.dy[i];
      const float t = dyv[i] * gam[i];
      c1 += t, c2 += t * xh[i];
      dg[i] += dyv[i] * xh[i], db[i] += dyv[i];
    }
    c1 = wave_sum(c1) / D, c2 = wave_sum(c2) / D;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int c = lane + 64 * i;
      if (c < D) dx[(long)row * D + c] = rs * (dyv[i] * gam[i] - c1 - xh[i] * c2) + cur.r[i];
    }
    cur = nxt;
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) red[w][0][lane + 64 * i] = dg[i], red[w][1][lane + 64 * i] = db[i];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    part[((long)blockIdx.x * 2 + 0) * D + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
    part[((long)blockIdx.x * 2 + 1) * D + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
  }
}
// the same for D % 4 == 0 with 16-byte accesses: lane l owns columns 4 l + 256 i .. + 3
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_v4_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ part, int M, int D) {
  __shared__ float red[4][2][256 * NV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rpb = (M + gridDim.x - 1) / gridDim.x, r_begin = blockIdx.x * rpb, r_end = min(M, r_begin + rpb);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gam[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * lane + 256 * i;
    gam[i] = c < D ? *reinterpret_cast<const float4*>(g + c) : z4, dg[i] = z4, db[i] = z4;
  }
  struct Row {
    float4 x[NV], dy[NV], r[NV];
    float mu, rs;
  };
  auto fetch = [&](int row, Row& o) __attribute__((always_inline)) {
    o.mu = mean[row], o.rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * lane + 256 * i;
      const bool ok = c < D;
      o.x[i] = ok ? *reinterpret_cast<const float4*>(x + (long)row * D + c) : z4;
      o.dy[i] = ok ? *reinterpret_cast<const float4*>(dy + (long)row * D + c) : z4;
      o.r[i] = (ok && dres) ? *reinterpret_cast<const float4*>(dres + (long)row * D + c) : z4;
    }
  };
  Row cur{}, nxt{};
  int row = r_begin + w;
  if (row < r_end) fetch(row, cur);
  for (; row < r_end; row += 4) {
    if (row + 4 < r_end) fetch(row + 4, nxt);
    const float mu = cur.mu, rs = cur.rs;
    float4 xh[NV];
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool ok = 4 * lane + 256 * i < D;
      const float4 v = cur.x[i], d = cur.dy[i];
      xh[i] = ok ? make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs) : z4;
      const float tx = d.x * gam[i].x, ty = d.y * gam[i].y, tz = d.z * gam[i].z, tw = d.w * gam[i].w;
      c1 += (tx + ty) + (tz + tw);
      c2 += (tx * xh[i].x + ty * xh[i].y) + (tz * xh[i].z + tw * xh[i].w);
      dg[i].x += d.x * xh[i].x, dg[i].y += d.y * xh[i].y, dg[i].z += d.z * xh[i].z, dg[i].w += d.w * xh[i].w;
      db[i].x += d.x, db[i].y += d.y, db[i].z += d.z, db[i].w += d.w;
    }
    c1 = wave_sum(c1) / D, c2 = wave_sum(c2) / D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * lane + 256 * i;
      if (c < D) {
        const float4 d = cur.dy[i];
        float4 o;
        o.x = rs * (d.x * gam[i].x - c1 - xh[i].x * c2) + cur.r[i].x, o.y = rs * (d.y * gam[i].y - c1 - xh[i].y * c2) + cur.r[i].y;
        o.z = rs * (d.z * gam[i].z - c1 - xh[i].z * c2) + cur.r[i].z, o.w = rs * (d.w * gam[i].w - c1 - xh[i].w * c2) + cur.r[i].w;
        *reinterpret_cast<float4*>(dx + (long)row * D + c) = o;
      }
    }
    cur = nxt;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    *reinterpret_cast<float4*>(&red[w][0][4 * lane + 256 * i]) = dg[i];
    *reinterpret_cast<float4*>(&red[w][1][4 * lane + 256 * i]) = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    part[((long)blockIdx.x * 2 + 0) * D + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
    part[((long)blockIdx.x * 2 + 1) * D + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
  }
}
hipError_t launch_ln_bwd(const float* dy, const float* x, const float* g, const float* mean, const float* rstd, const float* dres,
                         float* dx, float* part, int M, int D, hipStream_t st) {
  if (D > 64 * kLnMax) return hipErrorInvalidValue;
  const dim3 grid(ln_bwd_blocks(M));
  if (D % 4 == 0) {
    if (D <= 256) hipLaunchKernelGGL(ln_bwd_v4_kernel<1>, grid, dim3(256), 0, st, dy, x, g, mean, rstd, dres, dx, part, M, D);
    else if (D <= 512) hipLaunchKernelGGL(ln_bwd_v4_kernel<2>, grid, dim3(256), 0, st, dy, x, g, mean, rstd, dres, dx, part, M, D);
    else hipLaunchKernelGGL(ln_bwd_v4_kernel<4>, grid, dim3(256), 0, st, dy, x, g, mean, rstd, dres, dx, part, M, D);
    return hipGetLastError();
  }
  if (D <= 256) hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, dim3(256), 0, st, dy, x, g, mean, rstd, dres, dx, part, M, D);
  else if (D <= 512) hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, dim3(256), 0, st, dy, x, g, mean, rstd, dres, dx, part, M, D);
  else hipLaunchKernelGGL(ln_bwd_kernel<16>, grid, dim3(256), 0, st, dy, x, g, mean, rstd, dres, dx, part, M, D);
  return hipGetLastError();
}

// out[j] = sum_s part[s * stride + j]: SL partial sums per column in flight (256 / SL columns per block), summed through LDS
template <int SL>
__global__ __launch_bounds__(256) void reduce_leading_kernel(const float* __restrict__ part, int S, long stride, long n, float* __restrict__ out,
                                                             float* __restrict__ out1, long n0) {
  constexpr int COLS = 256 / SL;
  __shared__ float red[SL][COLS];
  const int c = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const long j = (long)blockIdx.x * COLS + c;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (j < n) {
    int i = sl;
    for (; i + 3 * SL < S; i += 4 * SL) {
      s0 += part[(long)i * stride + j], s1 += part[(long)(i + SL) * stride + j];
      s2 += part[(long)(i + 2 * SL) * stride + j], s3 += part[(long)(i + 3 * SL) * stride + j];
    }
    for (; i < S; i += SL) s0 += part[(long)i * stride + j];
  }
  red[sl][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && j < n) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < SL; ++k) t += red[k][c];
    if (out1 && j >= n0) out1[j - n0] = t;  // two-output form: columns [n0, n) go to the second array
    else out[j] = t;
  }
}
hipError_t launch_reduce_leading(const float* part, int S, long stride, long n, float* out, hipStream_t st) {
  if (S <= 32) hipLaunchKernelGGL(reduce_leading_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, part, S, stride, n, out, (float*)nullptr, 0L);
  else hipLaunchKernelGGL(reduce_leading_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, part, S, stride, n, out, (float*)nullptr, 0L);
  return hipGetLastError();
}
hipError_t launch_reduce_leading_split(const float* part, int S, long stride, long n, long n0, float* out0, float* out1, hipStream_t st) {
  if (S <= 32) hipLaunchKernelGGL(reduce_leading_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, part, S, stride, n, out0, out1, n0);
  else hipLaunchKernelGGL(reduce_leading_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, part, S, stride, n, out0, out1, n0);
  return hipGetLastError();
}
hipError_t launch_reduce_leading2(const float* part, int S, long n, float* out0, float* out1, hipStream_t st) {
  if (S <= 32) hipLaunchKernelGGL(reduce_leading_kernel<4>, dim3((unsigned)((2 * n + 63) / 64)), dim3(256), 0, st, part, S, 2 * n, 2 * n, out0, out1, n);
  else hipLaunchKernelGGL(reduce_leading_kernel<16>, dim3((unsigned)((2 * n + 15) / 16)), dim3(256), 0, st, part, S, 2 * n, 2 * n, out0, out1, n);
  return hipGetLastError();
}

int colsum_blocks(int M) { return max(1, min(512, (M + 31) / 32)); }
// part[blk][which][n]: which = 0: sum of (x - shift), 1 (only if SQ): sum of (x - shift)^2
template <bool SQ>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, const float* __restrict__ shift, float* __restrict__ part) {
  const int rpb = (M + gridDim.x - 1) / gridDim.x, r_begin = blockIdx.x * rpb, r_end = min(M, r_begin + rpb);
  for (int c = threadIdx.x; c < N; c += 256) {
    const float sh = shift ? shift[c] : 0.0f;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int r = r_begin;
    for (; r + 3 < r_end; r += 4) {  // four independent loads in flight per thread
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v = X[(long)(r + u) * N + c] - sh;
        s[u] += v;
        if (SQ) q[u] += v * v;
      }
    }
    for (; r < r_end; ++r) {
      const float v = X[(long)r * N + c] - sh;
      s[0] += v;
      if (SQ) q[0] += v * v;
    }
    const float st = (s[0] + s[1]) + (s[2] + s[3]), qt = (q[0] + q[1]) + (q[2] + q[3]);
    if (SQ) part[((long)blockIdx.x * 2) * N + c] = st, part[((long)blockIdx.x * 2 + 1) * N + c] = qt;
    else part[(long)blockIdx.x * N + c] = st;
  }
}
hipError_t launch_colsum_partial(const float* X, int M, int N, float* part, hipStream_t st) {
  hipLaunchKernelGGL(colsum_kernel<false>, dim3(colsum_blocks(M)), dim3(256), 0, st, X, M, N, (const float*)nullptr, part);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// pointwise
// ---------------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void scale_drop_kernel(const float* __restrict__ dx, float scale, float* __restrict__ dh, long n, Drop d) {
  const DropState ds(d);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dh[i] = scale * dx[i] * ds.mul((uint64_t)i);
}
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}
static dim3 pw_grid(long n) { return dim3((unsigned)min((n + 255) / 256, (long)(256 * 16))); }
hipError_t launch_scale_drop(const float* dx, float scale, float* dh, long n, Drop d, hipStream_t st) {
  hipLaunchKernelGGL(scale_drop_kernel, pw_grid(n), dim3(256), 0, st, dx, scale, dh, n, d);
  return hipGetLastError();
}
hipError_t launch_axpy(float* y, const float* x, float a, long n, hipStream_t st) {
  hipLaunchKernelGGL(axpy_kernel, pw_grid(n), dim3(256), 0, st, y, x, a, n);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void glu_fwd_kernel(const float* __restrict__ u, float* __restrict__ g, long n, int D) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    g[i] = u[r * 2 * D + c] * sigmoidf_(u[r * 2 * D + D + c]);
  }
}
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ u, float* __restrict__ du, long n, int D) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    const float a = u[r * 2 * D + c], s = sigmoidf_(u[r * 2 * D + D + c]), d = dg[i];
    du[r * 2 * D + c] = d * s;
    du[r * 2 * D + D + c] = d * a * s * (1.0f - s);
  }
}
hipError_t launch_glu_fwd(const float* u, float* g, int M, int D, hipStream_t st) {
  hipLaunchKernelGGL(glu_fwd_kernel, pw_grid((long)M * D), dim3(256), 0, st, u, g, (long)M * D, D);
  return hipGetLastError();
}
hipError_t launch_glu_bwd(const float* dg, const float* u, float* du, int M, int D, hipStream_t st) {
  hipLaunchKernelGGL(glu_bwd_kernel, pw_grid((long)M * D), dim3(256), 0, st, dg, u, du, (long)M * D, D);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// depthwise conv over time
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kDwMaxK = 31;
#ifndef EECT_DW_ROWS
#define EECT_DW_ROWS 32
#endif
#ifndef EECT_DW_STEP
#define EECT_DW_STEP 32
#endif
constexpr int kDwRows = EECT_DW_ROWS;  // time steps per block
constexpr int kDwStep = EECT_DW_STEP;  // outputs per thread and window load
// y[b][t][d] = bias[d] + sum_j w[d][FLIP ? K-1-j : j] * x[b][t + j - pad][d]; a thread owns channel d and produces kDwStep
// consecutive outputs from one K + kDwStep - 1 long window of x (loads are coalesced over d).
// Two forms.  The FAST one (K = 31, the block of kDwStep rows is the first / an interior / the last one of its utterance) knows at
// compile time which window rows exist: no bounds logic at all, 62 (47 at the ends) independent loads and 992 multiply-adds per
// thread.  The GENERIC one (any K, ragged ends) clamps the row index and zeroes afterwards.  In both the row index is made per-lane
// on purpose (an opaque zero in a VGPR): left uniform, the compiler keeps one 64-bit row address -- and one mask -- per window row
// in SGPRs and spills hundreds of them (the first form of these kernels: 5 k instructions, 1.1 k of them SGPR spill traffic, for 1 k
// multiply-adds; 38 / 42 / 70 us per launch at the default model).
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}
constexpr int kDwPad = (kDwMaxK - 1) / 2, kDwWin = kDwMaxK + kDwStep - 1;
// EDGE: 0 interior, 1 the rows before the utterance are missing (t == 0), 2 the rows behind it are (t + kDwStep == T)
template <int EDGE>
__device__ __forceinline__ constexpr bool dw_have(int i) {
  return !(EDGE == 1 && i < kDwPad) && !(EDGE == 2 && i >= kDwStep + kDwPad);
}
template <int EDGE>
__device__ __forceinline__ void dw_window_fast(float (&win)[kDwWin], const float* __restrict__ utt, int d, int t, int D, int z) {
#pragma unroll
  for (int i = 0; i < kDwWin; ++i) {
    if (dw_have<EDGE>(i)) win[i] = utt[(unsigned)((t - kDwPad + i + z) * D + d)];
    else win[i] = 0.0f;
  }
}
// 0 / 1 / 2: the fast form applies with that EDGE; -1: generic
__device__ __forceinline__ int dw_edge(int t, int T, int K) {
  if (K != kDwMaxK) return -1;
  if (t - kDwPad >= 0 && t + kDwStep + kDwPad <= T) return 0;
  if (t == 0 && kDwStep + kDwPad <= T) return 1;
  if (t + kDwStep == T && t - kDwPad >= 0) return 2;
  return -1;
}
template <int W>
__device__ __forceinline__ void dw_window_generic(float (&win)[W], const float* __restrict__ utt, int d, int t_first, int T, int D, int wlen, int z) {
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int tt = t_first + j + z, tc = min(max(tt, 0), T - 1);
    const float v = utt[(unsigned)(tc * D + d)];
    win[j] = (tt == tc && j < wlen) ? v : 0.0f;
  }
}
template <bool FLIP>
__global__ __launch_bounds__(256) void dw_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                 float* __restrict__ y, int T, int D, int K) {
  const int b = blockIdx.y, t0 = blockIdx.x * kDwRows, pad = (K - 1) / 2;
  const float* __restrict__ utt = x + (long)b * T * D;
  float* __restrict__ yutt = y + (long)b * T * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    float wt[kDwMaxK];
#pragma unroll
    for (int j = 0; j < kDwMaxK; ++j) {
      const float v = w[(long)d * K + (j < K ? (FLIP ? K - 1 - j : j) : 0)];
      wt[j] = j < K ? v : 0.0f;
    }
    const float bv = bias ? bias[d] : 0.0f;
    const int t_end = min(T, t0 + kDwRows), z = opaque_zero();
    for (int t = t0; t < t_end; t += kDwStep) {
      float win[kDwWin];
      auto fast = [&](auto edge_tag) __attribute__((always_inline)) {
        constexpr int EDGE = decltype(edge_tag)::value;
        dw_window_fast<EDGE>(win, utt, d, t, D, z);
#pragma unroll
        for (int o = 0; o < kDwStep; ++o) {
          float s = bv;
#pragma unroll
          for (int j = 0; j < kDwMaxK; ++j)
            if (dw_have<EDGE>(o + j)) s += wt[j] * win[o + j];
          yutt[(unsigned)((t + o + z) * D + d)] = s;
        }
      };
      const int edge = dw_edge(t, T, K);  // uniform
      if (edge == 0) fast(IntTag<0>{});
      else if (edge == 1) fast(IntTag<1>{});
      else if (edge == 2) fast(IntTag<2>{});
      else {
        dw_window_generic(win, utt, d, t - pad, T, D, K + kDwStep - 1, z);
#pragma unroll
        for (int o = 0; o < kDwStep; ++o) {
          float s = bv;
#pragma unroll
          for (int j = 0; j < kDwMaxK; ++j) s += wt[j] * win[j + o];
          if (t + o < t_end) yutt[(unsigned)((t + o + z) * D + d)] = s;
        }
      }
    }
  }
}
hipError_t launch_dw_fwd(const float* x, const float* w, const float* b, float* y, int B, int T, int D, int K, hipStream_t st) {
  if (K > kDwMaxK || !(K & 1) || (long)T * D >= (1l << 30)) return hipErrorInvalidValue;  // 32-bit element offsets inside an utterance
  hipLaunchKernelGGL(dw_kernel<false>, dim3((T + kDwRows - 1) / kDwRows, B), dim3(256), 0, st, x, w, b, y, T, D, K);
  return hipGetLastError();
}
hipError_t launch_dw_bwd_data(const float* dy, const float* w, float* dx, int B, int T, int D, int K, hipStream_t st) {
  if (K > kDwMaxK || !(K & 1) || (long)T * D >= (1l << 30)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dw_kernel<true>, dim3((T + kDwRows - 1) / kDwRows, B), dim3(256), 0, st, dy, w, (const float*)nullptr, dx, T, D, K);
  return hipGetLastError();
}
#ifndef EECT_DWW_ROWS
#define EECT_DWW_ROWS 64
#endif
constexpr int kDwwRows = EECT_DWW_ROWS;
static_assert(kDwRows % kDwStep == 0 && kDwwRows % kDwStep == 0, "blocks are whole steps");
int dw_bwd_weight_blocks(int B, int T) { return B * ((T + kDwwRows - 1) / kDwwRows); }
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                            int T, int D, int K) {
  const int nchunk = (T + kDwwRows - 1) / kDwwRows, b = blockIdx.x / nchunk, t0 = (blockIdx.x % nchunk) * kDwwRows, pad = (K - 1) / 2;
  const float* __restrict__ xutt = x + (long)b * T * D;
  const float* __restrict__ gutt = dy + (long)b * T * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc[kDwMaxK + 1];
#pragma unroll
    for (int j = 0; j <= kDwMaxK; ++j) acc[j] = 0.0f;
    const int t_end = min(T, t0 + kDwwRows), z = opaque_zero();
    for (int t = t0; t < t_end; t += kDwStep) {
      float win[kDwWin], g[kDwStep];
      auto fast = [&](auto edge_tag) __attribute__((always_inline)) {
        constexpr int EDGE = decltype(edge_tag)::value;
        dw_window_fast<EDGE>(win, xutt, d, t, D, z);
#pragma unroll
        for (int o = 0; o < kDwStep; ++o) {
          g[o] = gutt[(unsigned)((t + o + z) * D + d)];
          acc[kDwMaxK] += g[o];
        }
#pragma unroll
        for (int j = 0; j < kDwMaxK; ++j)
#pragma unroll
          for (int o = 0; o < kDwStep; ++o)
            if (dw_have<EDGE>(o + j)) acc[j] += g[o] * win[o + j];
      };
      const int edge = dw_edge(t, T, K);  // uniform
      if (edge == 0) fast(IntTag<0>{});
      else if (edge == 1) fast(IntTag<1>{});
      else if (edge == 2) fast(IntTag<2>{});
      else {
        dw_window_generic(win, xutt, d, t - pad, T, D, K + kDwStep - 1, z);
#pragma unroll
        for (int o = 0; o < kDwStep; ++o) {
          const int tg = t + o + z;
          const float v = gutt[(unsigned)(min(tg, T - 1) * D + d)];
          g[o] = tg < t_end ? v : 0.0f;
          acc[kDwMaxK] += g[o];
        }
#pragma unroll
        for (int j = 0; j < kDwMaxK; ++j)
#pragma unroll
          for (int o = 0; o < kDwStep; ++o) acc[j] += g[o] * win[j + o];
      }
    }
#pragma unroll
    for (int j = 0; j < kDwMaxK; ++j)
      if (j < K) part[((long)blockIdx.x * (K + 1) + j) * D + d] = acc[j];
    part[((long)blockIdx.x * (K + 1) + K) * D + d] = acc[kDwMaxK];
  }
}
// dw[d][j] = tot[j][d], db[d] = tot[K][d]  (tot = the partials summed over the blocks, [K + 1][D])
__global__ __launch_bounds__(256) void dw_weight_finalize_kernel(const float* __restrict__ tot, int D, int K, float* __restrict__ dw, float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (K + 1) * D) return;
  const int j = i / D, d = i % D;
  if (j < K) dw[(long)d * K + j] = tot[i];
  else db[d] = tot[i];
}
hipError_t launch_dw_bwd_weight(const float* dy, const float* x, float* part, float* dw, float* db, int B, int T, int D, int K, hipStream_t st) {
  if (K > kDwMaxK || !(K & 1) || (long)T * D >= (1l << 30)) return hipErrorInvalidValue;
  const int S = dw_bwd_weight_blocks(B, T);
  const long n = (long)(K + 1) * D;
  float* tot = part + (size_t)S * n;  // [K + 1][D] behind the partials
  hipLaunchKernelGGL(dw_bwd_weight_kernel, dim3(S), dim3(256), 0, st, dy, x, part, T, D, K);
  (void)launch_reduce_leading(part, S, n, n, tot, st);
  hipLaunchKernelGGL(dw_weight_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)tot, D, K, dw, db);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm (batch statistics) + SiLU
// ---------------------------------------------------------------------------------------------------------------------
// which == 0: stats[0][c] = mean from sums[c]; which == 1: stats[1][c] = rstd, mv = (mean, biased var) from sums[2][c]
// (sums of (x - mean) and (x - mean)^2: the first is ~0 and only corrects the mean's rounding)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, int M, int D, int which, float* __restrict__ stats,
                                                          float* __restrict__ mv) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  if (which == 0) {
    stats[c] = sums[c] / M;
  } else {
    const float dm = sums[c] / M, var = fmaxf(sums[D + c] / M - dm * dm, 0.0f);
    stats[D + c] = rsqrtf(var + 1e-5f);
    if (mv) mv[c] = stats[c] + dm, mv[D + c] = var;
  }
}
hipError_t launch_bn_stats(const float* c, int M, int D, float* part, float* stats, float* mv, hipStream_t st) {
  const int S = colsum_blocks(M);
  float* sums = part + (size_t)S * 2 * D;  // [2][D] behind the partials
  hipLaunchKernelGGL(colsum_kernel<false>, dim3(S), dim3(256), 0, st, c, M, D, (const float*)nullptr, part);
  (void)launch_reduce_leading(part, S, D, D, sums, st);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((D + 255) / 256), dim3(256), 0, st, (const float*)sums, M, D, 0, stats, mv);
  hipLaunchKernelGGL(colsum_kernel<true>, dim3(S), dim3(256), 0, st, c, M, D, (const float*)stats, part);
  (void)launch_reduce_leading(part, S, (long)2 * D, (long)2 * D, sums, st);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((D + 255) / 256), dim3(256), 0, st, (const float*)sums, M, D, 1, stats, mv);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void bn_silu_fwd_kernel(const float* __restrict__ c, const float* __restrict__ stats, const float* __restrict__ g,
                                                          const float* __restrict__ b, float* __restrict__ s, long n, int D) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % D);
    const float y = (c[i] - stats[ch]) * stats[D + ch] * g[ch] + b[ch];
    s[i] = y * sigmoidf_(y);
  }
}
hipError_t launch_bn_silu_fwd(const float* c, const float* stats, const float* g, const float* b, float* s, int M, int D, hipStream_t st) {
  hipLaunchKernelGGL(bn_silu_fwd_kernel, pw_grid((long)M * D), dim3(256), 0, st, c, stats, g, b, s, (long)M * D, D);
  return hipGetLastError();
}
// part[blk][0][ch] = sum dy, part[blk][1][ch] = sum dy * xhat over the block's rows
__global__ __launch_bounds__(256) void bn_silu_bwd_sums_kernel(const float* __restrict__ ds, const float* __restrict__ c, const float* __restrict__ stats,
                                                               const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ part, int M, int D) {
  const int rpb = (M + gridDim.x - 1) / gridDim.x, r_begin = blockIdx.x * rpb, r_end = min(M, r_begin + rpb);
  for (int ch = threadIdx.x; ch < D; ch += 256) {
    const float mu = stats[ch], rs = stats[D + ch], gg = g[ch], bb = b[ch];
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 8  // eight rows' loads in flight (rolled, every row is its own round trip: 31 us per launch against 7 us of traffic)
    for (int r = r_begin; r < r_end; ++r) {
      const float xh = (c[(long)r * D + ch] - mu) * rs, y = xh * gg + bb, sg = sigmoidf_(y);
      const float dy = ds[(long)r * D + ch] * sg * (1.0f + y * (1.0f - sg));
      s0 += dy, s1 += dy * xh;
    }
    part[((long)blockIdx.x * 2) * D + ch] = s0, part[((long)blockIdx.x * 2 + 1) * D + ch] = s1;
  }
}
__global__ __launch_bounds__(256) void bn_silu_bwd_apply_kernel(const float* __restrict__ ds, const float* __restrict__ c, const float* __restrict__ stats,
                                                                const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ sums,
                                                                float* __restrict__ dc, long n, int D, float inv_m) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % D);
    const float rs = stats[D + ch], xh = (c[i] - stats[ch]) * rs, y = xh * g[ch] + b[ch], sg = sigmoidf_(y);
    const float dy = ds[i] * sg * (1.0f + y * (1.0f - sg));
    dc[i] = g[ch] * rs * (dy - sums[ch] * inv_m - xh * sums[D + ch] * inv_m);
  }
}
hipError_t launch_bn_silu_bwd(const float* ds, const float* c, const float* stats, const float* g, const float* b, float* part, float* sums,
                              float* dc, int M, int D, hipStream_t st) {
  const int S = colsum_blocks(M);
  hipLaunchKernelGGL(bn_silu_bwd_sums_kernel, dim3(S), dim3(256), 0, st, ds, c, stats, g, b, part, M, D);
  (void)launch_reduce_leading(part, S, (long)2 * D, (long)2 * D, sums, st);
  hipLaunchKernelGGL(bn_silu_bwd_apply_kernel, pw_grid((long)M * D), dim3(256), 0, st, ds, c, stats, g, b, (const float*)sums, dc, (long)M * D, D, 1.0f / M);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// softmax over keys (one wave per query row)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSmMax = 16;  // keys per lane held in registers: T <= 1024 (longer rows take the re-reading kernel)
// one wave per query row; REG: the row lives in registers between the three passes.  Pd (optional) = drop(P)
template <bool REG>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* __restrict__ S, float* __restrict__ Pd, const int32_t* __restrict__ key_len, long rows,
                                                          int H, int T, float scale, Drop d) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const DropState ds(d);
  const int len = key_len[row / ((long)H * T)];
  float* s = S + row * T;
  float v[kSmMax];
  float mx = -INFINITY;
  if (REG) {
#pragma unroll
    for (int i = 0; i < kSmMax; ++i) {
      const int k = lane + 64 * i;
      v[i] = (k < T && k < len) ? s[k] * scale : -INFINITY;
      mx = fmaxf(mx, v[i]);
    }
  } else {
    for (int k = lane; k < T; k += 64) mx = fmaxf(mx, k < len ? s[k] * scale : -INFINITY);
  }
  mx = wave_max(mx);
  float sum = 0.0f;
  if (REG) {
#pragma unroll
    for (int i = 0; i < kSmMax; ++i) {
      v[i] = __expf(v[i] - mx);  // exp(-inf) = 0 for masked keys; len == 0: exp(nan)
      sum += (lane + 64 * i < T) ? v[i] : 0.0f;
    }
  } else {
    for (int k = lane; k < T; k += 64) sum += k < len ? __expf(s[k] * scale - mx) : 0.0f;
  }
  const float inv = 1.0f / wave_sum(sum);  // len == 0: nan in every column, as torch
  if (REG) {
#pragma unroll
    for (int i = 0; i < kSmMax; ++i) {
      const int k = lane + 64 * i;
      if (k < T) {
        const float p = len > 0 ? v[i] * inv : NAN;
        s[k] = p;
        if (Pd) Pd[row * T + k] = p * ds.mul((uint64_t)(row * T + k));
      }
    }
  } else {
    for (int k = lane; k < T; k += 64) {
      const float p = k < len ? __expf(s[k] * scale - mx) * inv : (len > 0 ? 0.0f : NAN);
      s[k] = p;
      if (Pd) Pd[row * T + k] = p * ds.mul((uint64_t)(row * T + k));
    }
  }
}
hipError_t launch_softmax_fwd(float* S, float* Pd, const int32_t* key_len, int B, int H, int T, float scale, Drop d, hipStream_t st) {
  const long rows = (long)B * H * T;
  if (T <= 64 * kSmMax) hipLaunchKernelGGL(softmax_fwd_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, Pd, key_len, rows, H, T, scale, d);
  else hipLaunchKernelGGL(softmax_fwd_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, Pd, key_len, rows, H, T, scale, d);
  return hipGetLastError();
}
template <bool REG>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, long rows, int T, float scale, Drop d) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const DropState ds(d);
  const float* p = P + row * T;
  float* g = dP + row * T;
  float dot = 0.0f;
  if (REG) {
    float pv[kSmMax], gv[kSmMax];
#pragma unroll
    for (int i = 0; i < kSmMax; ++i) {
      const int k = lane + 64 * i;
      pv[i] = k < T ? p[k] : 0.0f;
      gv[i] = k < T ? g[k] * ds.mul((uint64_t)(row * T + k)) : 0.0f;
      dot += gv[i] * pv[i];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < kSmMax; ++i) {
      const int k = lane + 64 * i;
      if (k < T) g[k] = scale * pv[i] * (gv[i] - dot);
    }
  } else {
    for (int k = lane; k < T; k += 64) dot += g[k] * ds.mul((uint64_t)(row * T + k)) * p[k];
    dot = wave_sum(dot);
    for (int k = lane; k < T; k += 64) g[k] = scale * p[k] * (g[k] * ds.mul((uint64_t)(row * T + k)) - dot);
  }
}
// rectangular form: `rows` rows of Tk probabilities (the AED decoder's self- / cross-attention); dropout stream position = flat index
hipError_t launch_softmax_bwd_rows(const float* P, float* dP, long rows, int Tk, float scale, Drop d, hipStream_t st) {
  if (Tk <= 64 * kSmMax) hipLaunchKernelGGL(softmax_bwd_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, rows, Tk, scale, d);
  else hipLaunchKernelGGL(softmax_bwd_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, rows, Tk, scale, d);
  return hipGetLastError();
}
hipError_t launch_softmax_bwd(const float* P, float* dP, int B, int H, int T, float scale, Drop d, hipStream_t st) {
  const long rows = (long)B * H * T;
  if (T <= 64 * kSmMax) hipLaunchKernelGGL(softmax_bwd_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, rows, T, scale, d);
  else hipLaunchKernelGGL(softmax_bwd_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, rows, T, scale, d);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void logsoftmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int M, int V) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long)row * V;
  float mx = -INFINITY;
  for (int k = lane; k < V; k += 64) mx = fmaxf(mx, xr[k]);
  mx = wave_max(mx);
  float sum = 0.0f;
  for (int k = lane; k < V; k += 64) sum += expf(xr[k] - mx);
  const float lse = mx + logf(wave_sum(sum));
  for (int k = lane; k < V; k += 64) y[(long)row * V + k] = xr[k] - lse;
}
hipError_t launch_logsoftmax_fwd(const float* logits, float* logp, int M, int V, hipStream_t st) {
  hipLaunchKernelGGL(logsoftmax_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, st, logits, logp, M, V);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// AED decoder helpers
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_masked_kernel(float* __restrict__ S, long rows, int H, int Tq, int Tk, float scale, int causal,
                                                             const unsigned char* __restrict__ key_pad, float* __restrict__ Pd, Drop d) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const DropState dst(d);
  const int tq = (int)(row % Tq);
  const long b = row / ((long)H * Tq);
  const unsigned char* pad = key_pad ? key_pad + b * Tk : nullptr;
  float* s = S + row * Tk;
  auto live = [&](int k) { return !(causal && k > tq) && !(pad && pad[k]); };
  float mx = -INFINITY;
  for (int k = lane; k < Tk; k += 64) mx = fmaxf(mx, live(k) ? s[k] * scale : -INFINITY);
  mx = wave_max(mx);
  float sum = 0.0f;
  for (int k = lane; k < Tk; k += 64) sum += live(k) ? __expf(s[k] * scale - mx) : 0.0f;
  const float inv = 1.0f / wave_sum(sum);  // no live key: nan in every column, as torch
  for (int k = lane; k < Tk; k += 64) {
    const float p = live(k) ? __expf(s[k] * scale - mx) * inv : (mx == -INFINITY ? NAN : 0.0f);
    s[k] = p;
    if (Pd) Pd[row * Tk + k] = p * dst.mul((uint64_t)(row * Tk + k));
  }
}
hipError_t launch_softmax_masked(float* S, int B, int H, int Tq, int Tk, float scale, int causal, const unsigned char* key_pad, hipStream_t st) {
  return launch_softmax_masked_drop(S, nullptr, B, H, Tq, Tk, scale, causal, key_pad, Drop{0.0f, 0, 0}, st);
}
hipError_t launch_softmax_masked_drop(float* S, float* Pd, int B, int H, int Tq, int Tk, float scale, int causal, const unsigned char* key_pad,
                                      Drop d, hipStream_t st) {
  const long rows = (long)B * H * Tq;
  hipLaunchKernelGGL(softmax_masked_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, rows, H, Tq, Tk, scale, causal, key_pad, Pd, d);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void embed_pe_kernel(const long long* __restrict__ tok, const float* __restrict__ emb, const float* __restrict__ pe,
                                                       float* __restrict__ x, unsigned char* __restrict__ pad, long n_tok, int S, int D, int V, int pad_idx,
                                                       Drop d) {
  const long i = blockIdx.x;
  if (i >= n_tok) return;
  const DropState ds(d);
  const long long t = tok[i];
  const long long tc = t < 0 ? 0 : (t >= V ? V - 1 : t);  // nn.Embedding would raise; stay in bounds
  if (threadIdx.x == 0 && pad) pad[i] = t == pad_idx;
  for (int c = threadIdx.x; c < D; c += 256) x[i * D + c] = (emb[tc * D + c] + pe[(i % S) * D + c]) * ds.mul((uint64_t)(i * D + c));
}
hipError_t launch_embed_pe(const long long* tok, const float* emb, const float* pe, float* x, unsigned char* pad, long n_tok, int S, int D, int V,
                           int pad_idx, hipStream_t st) {
  return launch_embed_pe_drop(tok, emb, pe, x, pad, n_tok, S, D, V, pad_idx, Drop{0.0f, 0, 0}, st);
}
hipError_t launch_embed_pe_drop(const long long* tok, const float* emb, const float* pe, float* x, unsigned char* pad, long n_tok, int S, int D, int V,
                                int pad_idx, Drop d, hipStream_t st) {
  hipLaunchKernelGGL(embed_pe_kernel, dim3((unsigned)n_tok), dim3(256), 0, st, tok, emb, pe, x, pad, n_tok, S, D, V, pad_idx, d);
  return hipGetLastError();
}
// demb[v][:] = sum over the tokens i with tok[i] == v (clamped like the forward) of dx[i][:] * dropmask(i, :), in token order:
// one workgroup per vocabulary entry walks the token list, so the sum has a fixed order (no atomics)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ tok, const float* __restrict__ dx, float* __restrict__ demb,
                                                        long n_tok, int D, int V, Drop d) {
  const int v = blockIdx.x;
  const DropState ds(d);
  for (int c0 = 0; c0 < D; c0 += 256) {
    const int c = c0 + threadIdx.x;
    float acc = 0.0f;
    for (long i = 0; i < n_tok; ++i) {
      const long long t = tok[i];
      const long long tc = t < 0 ? 0 : (t >= V ? V - 1 : t);
      if (tc == v && c < D) acc += dx[i * D + c] * ds.mul((uint64_t)(i * D + c));
    }
    if (c < D) demb[(long)v * D + c] = acc;
  }
}
hipError_t launch_embed_bwd(const long long* tok, const float* dx, float* demb, long n_tok, int D, int V, Drop d, hipStream_t st) {
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)V), dim3(256), 0, st, tok, dx, demb, n_tok, D, V, d);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// stem helpers
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_mel_kernel(const float* __restrict__ mel, float* __restrict__ a, int C, int T, int T1, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int col = (int)(i % (3 * C));
    const long row = i / (3 * C);
    const int t1 = (int)(row % T1), b = (int)(row / T1), c = col / 3, j = col % 3;
    a[i] = mel[((long)b * C + c) * T + 2 * t1 + j];
  }
}
hipError_t launch_im2col_mel(const float* mel, float* a, int B, int C, int T, int T1, hipStream_t st) {
  const long n = (long)B * T1 * 3 * C;
  hipLaunchKernelGGL(im2col_mel_kernel, pw_grid(n), dim3(256), 0, st, mel, a, C, T, T1, n);
  return hipGetLastError();
}
// to_jc: wp[o][j][c] = w[o][c][j]; else: wp[o][c][j] = w[o][j][c]
__global__ __launch_bounds__(256) void permute_w3_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, long n, int to_jc) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long o = i / (3 * C);
    const int r = (int)(i % (3 * C));
    if (to_jc) { const int j = r / C, c = r % C; wp[i] = w[o * 3 * C + c * 3 + j]; }
    else { const int c = r / 3, j = r % 3; wp[i] = w[o * 3 * C + j * C + c]; }
  }
}
hipError_t launch_permute_w3(const float* w, float* wp, int O, int C, int to_jc, hipStream_t st) {
  const long n = (long)O * 3 * C;
  hipLaunchKernelGGL(permute_w3_kernel, pw_grid(n), dim3(256), 0, st, w, wp, C, n, to_jc);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void col2im_stride2_kernel(const float* __restrict__ G, float* __restrict__ dout1, int T1, int T2, int D, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % D);
    const long row = i / D;
    const int t1 = (int)(row % T1), b = (int)(row / T1);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int u = t1 - j;
      if (u >= 0 && !(u & 1) && (u >> 1) < T2) s += G[(((long)b * T2 + (u >> 1)) * 3 + j) * D + c];
    }
    dout1[i] = s;
  }
}
hipError_t launch_col2im_stride2(const float* G, float* dout1, int B, int T1, int T2, int D, hipStream_t st) {
  const long n = (long)B * T1 * D;
  hipLaunchKernelGGL(col2im_stride2_kernel, pw_grid(n), dim3(256), 0, st, G, dout1, T1, T2, D, n);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void add_pe_drop_kernel(float* __restrict__ x, const float* __restrict__ pe, int T, int D, long n, Drop d) {
  const DropState ds(d);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % D), t = (int)((i / D) % T);
    x[i] = (x[i] + pe[(long)t * D + c]) * ds.mul((uint64_t)i);
  }
}
hipError_t launch_add_pe_drop(float* x, const float* pe, int B, int T, int D, Drop d, hipStream_t st) {
  const long n = (long)B * T * D;
  hipLaunchKernelGGL(add_pe_drop_kernel, pw_grid(n), dim3(256), 0, st, x, pe, T, D, n, d);
  return hipGetLastError();
}

}  // namespace eect
