// Key-padding-masked multi-head self-attention core (torch
// nn.MultiheadAttention -> scaled_dot_product_attention with key_padding_mask):
//     O[b, t, h*dh + d] = sum_k softmax_k( q[t].k[k] / sqrt(dh) | k < len[b] ) * v[k][d]
// Only keys are masked; padded query rows are computed like any other (reference behaviour).
//
// Workgroup = one (utterance, head, 256-query block); wave = 32 queries.  Both products
// keep the query on the MFMA lane:
//     S^T[key][q] = K . Q^T            (A = K fragment from LDS, B = Q fragment in registers)
//     O^T[d][q]  += V^T . P^T          (A = V^T fragment from LDS, B = P straight from the
//                                       S^T accumulators, no LDS round trip)
// so every softmax statistic (running max, sum, rescale) is a per-lane scalar.  The k order
// of an accumulator-fed operand is permuted (rows 8a+4h+c <-> element 4a+c of half h); the
// QKV kernel stores V^T with that permutation already applied (vt_perm in linear.hip).
// Q arrives pre-multiplied by log2(e)/sqrt(dh): probabilities are exp2(s - m).
// fp16 operands, fp32 accumulate/statistics, online softmax over 32-key tiles so T' is unbounded.
#include "eec_kernels.h"

namespace eec {

constexpr int kKC = 256;  // keys staged in LDS per chunk
constexpr float kNegBig = -1.0e30f;

template <int DH>
struct AttnLds {
  static constexpr int KLD = (DH + 8) * 2;
  static constexpr int VLD = (kKC + 8) * 2;
  static constexpr int KBYTES = kKC * KLD;
  static constexpr int VBYTES = DH * VLD;
  static constexpr int TOTAL = KBYTES + VBYTES;
  static constexpr int TOTAL2 = KBYTES + 2 * VBYTES;  // with the residual plane of V^T
  static constexpr int TOTAL3 = 2 * KBYTES + 2 * VBYTES;  // ... and of K (exact mode)
};

EEC_TL_DEFINE(attn)
#ifdef EEC_TL
__device__ unsigned long long g_tl_attn_all[2048];  // [block][start, end] of wave 0 (first 1024 workgroups)
extern "C" int eec_tl_read_attn_all(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl_attn_all), sizeof(g_tl_attn_all));
}
#define EEC_TL_ALL(i)                                                                               \
  do {                                                                                              \
    const int b_ = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));           \
    if (threadIdx.x == 0 && b_ < 1024) g_tl_attn_all[b_ * 2 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define EEC_TL_ALL(i)
#endif
// Workgroup shape: kAttnWaves waves x 32 queries.  8 waves (256 queries) means K and V^T of an (utterance, head)
// pair are staged ONCE when T' <= 256; with KB = 2 the kernel needs < 128 VGPRs, so two such workgroups (16
// waves) share a CU and every workgroup of the BASELINE shape is resident in a single round.
#ifndef EEC_ATTN_KB
#define EEC_ATTN_KB 2
#define EEC_ATTN_WAVES 8
#define EEC_ATTN_OCC 4
#endif
constexpr int kAttnWaves = EEC_ATTN_WAVES, kAttnThreads = 64 * kAttnWaves;
// X3 (exact mode, precision f16x3): Q and K arrive with fp16 residual planes and the probabilities are split hi / lo in
// registers, so both products run as three fp16 MFMA products (hi.hi + lo.hi + hi.lo): the attention path then adds ~1e-6 to the
// log-prob error instead of ~1e-4 (single-fp16 Q, K, P were what kept f16x3 at 1.2e-4).  Needs vt_lo too.
template <int DH, int NP, bool X3 = false>
__global__ __launch_bounds__(kAttnThreads, ((DH == 64 || X3) ? 2 : EEC_ATTN_OCC)) void attn_kernel(AttnArgs a) {
  using L = AttnLds<DH>;
  constexpr int KSQ = DH / 16;  // k-steps of the score product
  constexpr int DT = DH / 32;   // 32-row tiles of O^T
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_k = smem;
  char* lds_v = smem + L::KBYTES;
  char* lds_vlo = lds_v + L::VBYTES;  // only allocated / used when a.vt_lo is given
  char* lds_klo = lds_vlo + L::VBYTES;  // X3 only
  const bool v2 = a.vt_lo != nullptr;  // uniform
  const int lane = lane_id(), w = wave_id();
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int bh = b * a.H + hd;
  const int q0 = (blockIdx.x * kAttnWaves + w) * 32;
  const bool active = q0 < a.Tq;  // wave-uniform
  const int len = min(a.enc_len[b], a.Tq);

  EEC_TL_STAMP(attn, 0);
  EEC_TL_ALL(0);
  h8 qf[KSQ];
  [[maybe_unused]] h8 qfl[KSQ];
  if (active) {
    const size_t qoff = ((size_t)bh * a.Tp + q0 + r) * DH + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) {
      qf[ks] = *(const h8*)(a.q + qoff + ks * 16);
      if constexpr (X3) qfl[ks] = *(const h8*)(a.q_lo + qoff + ks * 16);
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = kNegBig, l_run = 0.f;

  const half_t* kbase = a.k + (size_t)bh * a.Tp * DH;
  const half_t* vbase = a.vt + (size_t)bh * DH * a.Tp;
  const half_t* vlbase = v2 ? a.vt_lo + (size_t)bh * DH * a.Tp : vbase;
  [[maybe_unused]] const half_t* klbase = X3 ? a.k_lo + (size_t)bh * a.Tp * DH : kbase;
  for (int kc0 = 0; kc0 < len; kc0 += kKC) {
    if (kc0) __syncthreads();
    // stage K rows [kc0, kc0+KC) and V^T columns of the same keys; zero beyond Tp.  All global loads of
    // the chunk are issued before the first LDS write (one memory round trip, not one per piece).
    {
      constexpr int KP = DH / 8;  // 16-byte pieces per K row
      constexpr int VP = kKC / 8;
      constexpr int KIT = kKC * KP / kAttnThreads, VIT = DH * VP / kAttnThreads;
      static_assert(kKC * KP % kAttnThreads == 0 && DH * VP % kAttnThreads == 0, "staging loops assume whole passes");
      uint4 kv[KIT], vv[VIT], vl[VIT];
      [[maybe_unused]] uint4 kl[KIT];
#pragma unroll
      for (int it = 0; it < KIT; ++it) {
        const int p = it * kAttnThreads + threadIdx.x, row = p / KP, c = p % KP;
        kv[it] = make_uint4(0, 0, 0, 0);
        if constexpr (X3) kl[it] = make_uint4(0, 0, 0, 0);
        if (kc0 + row < a.Tp) {
          kv[it] = *(const uint4*)(kbase + (size_t)(kc0 + row) * DH + c * 8);
          if constexpr (X3) kl[it] = *(const uint4*)(klbase + (size_t)(kc0 + row) * DH + c * 8);
        }
      }
#pragma unroll
      for (int it = 0; it < VIT; ++it) {
        const int p = it * kAttnThreads + threadIdx.x, row = p / VP, c = p % VP;
        vv[it] = make_uint4(0, 0, 0, 0);
        vl[it] = make_uint4(0, 0, 0, 0);
        if (kc0 + c * 8 < a.Tp) {
          vv[it] = *(const uint4*)(vbase + (size_t)row * a.Tp + kc0 + c * 8);
          if (v2) vl[it] = *(const uint4*)(vlbase + (size_t)row * a.Tp + kc0 + c * 8);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < KIT; ++it) {
        const int p = it * kAttnThreads + threadIdx.x, row = p / KP, c = p % KP;
        *(uint4*)(lds_k + row * L::KLD + c * 16) = kv[it];
        if constexpr (X3) *(uint4*)(lds_klo + row * L::KLD + c * 16) = kl[it];
      }
#pragma unroll
      for (int it = 0; it < VIT; ++it) {
        const int p = it * kAttnThreads + threadIdx.x, row = p / VP, c = p % VP;
        *(uint4*)(lds_v + row * L::VLD + c * 16) = vv[it];
        if (v2) *(uint4*)(lds_vlo + row * L::VLD + c * 16) = vl[it];
      }
    }
    EEC_TL_STAMP(attn, 1);
    __syncthreads();
    EEC_TL_STAMP(attn, 2);
    if (!active) continue;
    const int nkt = (min(len - kc0, kKC) + 31) / 32;
    // online softmax over blocks of KB key tiles (32 KB keys): the per-block overhead (cross-half
    // max, rescale of O, running sums) is paid once per block; the key mask is applied only in a
    // block that actually contains keys >= len (wave-uniform test)
    constexpr int KB = EEC_ATTN_KB;
    for (int kt0 = 0; kt0 < nkt; kt0 += KB) {
      f32x16 s[KB];
#pragma unroll
      for (int j = 0; j < KB; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[j][i] = 0.f;
        if (kt0 + j < nkt) {  // wave-uniform
#pragma unroll
          for (int ks = 0; ks < KSQ; ++ks) {
            const int koff = ((kt0 + j) * 32 + r) * L::KLD + (ks * 16 + 8 * hh) * 2;
            const h8 kf = *(const h8*)(lds_k + koff);
            if constexpr (X3) {  // the two correction products first, the main product last (as the GEMMs of this mode)
              s[j] = mfma16(*(const h8*)(lds_klo + koff), qf[ks], s[j]);
              s[j] = mfma16(kf, qfl[ks], s[j]);
            }
            s[j] = mfma16(kf, qf[ks], s[j]);
          }
        }
      }
      const int key0 = kc0 + kt0 * 32;
      if (key0 + KB * 32 > len) {  // block touches the masked tail (or runs past the last tile)
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (key0 + j * 32 + acc_row(i, lane) >= len) s[j][i] = kNegBig;
      }
      float tmax = kNegBig;
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[j][i]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
      h8 pf[KB][2];
      [[maybe_unused]] h8 pfl[KB][2];
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(s[j][i] - m_new);
          psum += p;
          const half_t ph = (half_t)p;
          pf[j][i >> 3][i & 7] = ph;
          if constexpr (X3) pfl[j][i >> 3][i & 7] = (half_t)(p - (float)ph);
        }
      l_run = l_run * alpha + psum;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
#pragma unroll
        for (int j = 0; j < KB; ++j)
          if (kt0 + j < nkt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const int voff = (dt * 32 + r) * L::VLD + ((kt0 + j) * 32 + ks * 16 + 8 * hh) * 2;
              const h8 vf = *(const h8*)(lds_v + voff);
              if (v2) o[dt] = mfma16(*(const h8*)(lds_vlo + voff), pf[j][ks], o[dt]);
              if constexpr (X3) o[dt] = mfma16(vf, pfl[j][ks], o[dt]);
              o[dt] = mfma16(vf, pf[j][ks], o[dt]);
            }
          }
      }
    }
  }
  EEC_TL_STAMP(attn, 3);
  if (!active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  // an utterance with NO valid key (lengths < 4 mel frames -> length 0): every key is masked and the installed
  // torch (>= 2.5, the oracle's) returns zeros for such rows ("safe softmax"); nothing was accumulated, so O = 0
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  const int q = q0 + r;
  if (q < a.Tq) {
    const size_t rowoff = ((size_t)b * a.Tq + q) * (a.H * DH) + hd * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) EEC_SPLIT(o[dt][4 * g + j] * inv, hi, lo, j);
        const int d = dt * 32 + 8 * g + 4 * hh;
        *(h4*)(a.o_hi + rowoff + d) = hi;
        if (NP == 3) *(h4*)(a.o_lo + rowoff + d) = lo;
      }
  }
  EEC_TL_STAMP(attn, 4);
  EEC_TL_ALL(1);
}

template <int DH, int NP, bool X3 = false>
static hipError_t launch_attn_t(const AttnArgs& a, hipStream_t st) {
  auto k = attn_kernel<DH, NP, X3>;
  constexpr int lds_max = X3 ? AttnLds<DH>::TOTAL3 : AttnLds<DH>::TOTAL2;
  const int lds = X3 ? AttnLds<DH>::TOTAL3 : a.vt_lo ? AttnLds<DH>::TOTAL2 : AttnLds<DH>::TOTAL;
  if (hipError_t e = ensure_max_lds((const void*)k, lds_max); e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.Tq + 32 * kAttnWaves - 1) / (32 * kAttnWaves), a.H, a.B), dim3(kAttnThreads), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_attention(const AttnArgs& a, int np, hipStream_t st) {
  if (a.vt_frag) return hipErrorInvalidValue;  // fragment-major V^T planes are the fused launch's (linear.hip)
  if (np == 3 && a.q_lo && a.k_lo && a.vt_lo) {  // exact mode
    if (a.dh == 32) return launch_attn_t<32, 3, true>(a, st);
    if (a.dh == 64) return launch_attn_t<64, 3, true>(a, st);
    return hipErrorInvalidValue;
  }
  if (a.dh == 32) return np == 3 ? launch_attn_t<32, 3>(a, st) : launch_attn_t<32, 1>(a, st);
  if (a.dh == 64) return np == 3 ? launch_attn_t<64, 3>(a, st) : launch_attn_t<64, 1>(a, st);
  return hipErrorInvalidValue;
}

}  // namespace eec
