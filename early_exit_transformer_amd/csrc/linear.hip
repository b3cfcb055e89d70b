// K=256 projection kernels of a Conformer layer.  One 512-thread workgroup (8 waves) owns a
// 64-row tile; wave w owns the 32-wide output column tile(s) {w, 8+w, 16+w}.  All products run in
// the swapped orientation (frame on the lane, 4 consecutive output features per register quad), so
// every epilogue stores straight from the accumulators - no LDS round trip, no barrier after the
// prologue.  Weight fragments stream through the register ring of eec_device.h; the ring of a pass
// is filled during the previous pass (or the prologue).  Accumulators start at the bias.
//   qkv_kernel           LN -> in_proj (N=768) -> Q (pre-scaled), K, V^T in attention layouts   
//   proj_residual_kernel x += A . W^T + b   (attention out_proj a6; conv pointwise-2 a7)
//   proj_glu_kernel      out_proj + residual -> LN -> pointwise-1 (N=512) -> GLU -> fp16, one launch (a6, a7)
//   head_kernel          exit head: Linear(D,V) -> log_softmax -> fp32 log-probs                    (a9)
#include "eec_blocks.h"

namespace eec {

EEC_TL_DEFINE(qkv)
EEC_TL_DEFINE(glu)
// ---------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(kLinThreads, 2) void qkv_kernel(QkvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = wave_id();
  const int row0 = blockIdx.x * kTileRows;
  WRing<NP, kLPF> rq;
  EEC_TL_STAMP(qkv, 0);
  rows_f32_to_planes<NP, true, 8>(smem, a.x, row0, a.M, a.ln_g, a.ln_b,
                                  [&]() { ring_fill<NP, kLPF, 1>(rq, wfrag_lane(a.wp, w), 0, kD / 16); });
  EEC_TL_STAMP(qkv, 1);
  __syncthreads();
  EEC_TL_STAMP(qkv, 2);
  qkv_body<NP>(smem, a, row0, rq);
}

hipError_t launch_qkv(const QkvArgs& a, int np, hipStream_t st) {
  auto k = np == 3 ? qkv_kernel<3> : qkv_kernel<1>;
  hipError_t e = ensure_max_lds((const void*)k, kLinLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows), dim3(kLinThreads), kLinLds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fp16 planes in global ([M][256] hi, [M][256] lo) -> LDS planes, 512 threads.
template <int NP>
struct PlaneRegs {
  uint4 h[4], l[NP == 3 ? 4 : 1];
};
// issue: all global loads of the tile's pieces (64 rows x 32 sixteen-byte pieces, 4 per thread per plane);
// commit: the LDS writes.  Split so that a caller can queue further loads (weight ring, residual rows) behind
// them and so that no LDS write sits between two loads (one memory round trip for the tile, not four).
template <int NP>
__device__ __forceinline__ void planes_issue512(PlaneRegs<NP>& pr, const half_t* __restrict__ hi,
                                                const half_t* __restrict__ lo, int row0, int M) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = it * kLinThreads + threadIdx.x, rl = piece >> 5, c16 = piece & 31, row = row0 + rl;
    pr.h[it] = make_uint4(0, 0, 0, 0);
    if (NP == 3) pr.l[it] = make_uint4(0, 0, 0, 0);
    if (row < M) {
      pr.h[it] = *(const uint4*)(hi + (size_t)row * kD + c16 * 8);
      if (NP == 3) pr.l[it] = *(const uint4*)(lo + (size_t)row * kD + c16 * 8);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int NP>
__device__ __forceinline__ void planes_commit512(char* lds_act, const PlaneRegs<NP>& pr) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = it * kLinThreads + threadIdx.x, rl = piece >> 5, c16 = piece & 31;
    *(uint4*)(lds_act + rl * kALd + c16 * 16) = pr.h[it];
    if (NP == 3) *(uint4*)(lds_act + kAPlane + rl * kALd + c16 * 16) = pr.l[it];
  }
}

// MT = 32-row tiles per workgroup: MT = 1 halves the tile (512 workgroups, two per CU, 16 waves per
// CU): these K=256 kernels are latency-bound, so the extra occupancy pays for streaming the (small)
// weight matrix twice as often.
template <int NP, int MT>
__global__ __launch_bounds__(kLinThreads, 2) void proj_residual_kernel(ProjResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = 32 * MT, PLANE = ROWS * kALd;
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int row0 = blockIdx.x * ROWS;
  const char* a_lane = smem + (lane & 31) * kALd + hh * 16;
  WRing<NP, kLPF, 1> r;
  ring_fill<NP, kLPF, 1>(r, wfrag_lane(a.wp, w), 0, kD / 16);
  for (int piece = threadIdx.x; piece < ROWS * 32; piece += kLinThreads) {
    const int rl = piece >> 5, c16 = piece & 31, row = row0 + rl;
    uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
    if (row < a.M) {
      vh = *(const uint4*)(a.a_hi + (size_t)row * kD + c16 * 8);
      if (NP == 3) vl = *(const uint4*)(a.a_lo + (size_t)row * kD + c16 * 8);
    }
    *(uint4*)(smem + rl * kALd + c16 * 16) = vh;
    if (NP == 3) *(uint4*)(smem + PLANE + rl * kALd + c16 * 16) = vl;
  }
  __syncthreads();
  f32x16 acc[MT][1];
  acc_init_bias<MT>(acc, a.bias + 32 * w);
  gemm_ring<NP, kD / 16, 1, true, kLPF, NoSide, 0, MT>(acc, a_lane, kALd, PLANE, wfrag_lane(a.wp, w), 0, r);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = row0 + mt * 32 + (lane & 31);
    if (row < a.M) {
      float* xr = a.x + (size_t)row * kD + 32 * w + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = *(const float4*)(xr + 8 * g);
        v.x += acc[mt][0][4 * g + 0];
        v.y += acc[mt][0][4 * g + 1];
        v.z += acc[mt][0][4 * g + 2];
        v.w += acc[mt][0][4 * g + 3];
        *(float4*)(xr + 8 * g) = v;
      }
    }
  }
}

#ifndef EEC_PROJ_MT
#define EEC_PROJ_MT 1
#endif
hipError_t launch_proj_residual(const ProjResArgs& a, int np, hipStream_t st) {
  constexpr int MT = EEC_PROJ_MT, ROWS = 32 * MT, LDS = 2 * ROWS * kALd;
  auto k = np == 3 ? proj_residual_kernel<3, MT> : proj_residual_kernel<1, MT>;
  hipError_t e = ensure_max_lds((const void*)k, LDS);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + ROWS - 1) / ROWS), dim3(kLinThreads), LDS, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Fused attention tail + conv-module head (one launch instead of two):
//   x += O . Wo^T + bo ;  g = GLU( LN_conv(x) . Wpw1^T + bpw1 )
// The out-proj result crosses from "wave owns 32 columns" to "wave owns 8 rows" through an fp32
// tile in LDS; the updated rows are LayerNorm'ed while still in registers and become the planes of
// the GLU product.  The GLU weight stream starts before the tile exchange.
constexpr int kProjGluLds = 2 * kAPlane + kETile;  // 134144

template <int NP>
__global__ __launch_bounds__(kLinThreads, 2) void proj_glu_kernel(ProjResArgs a, GluArgs gl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_e = smem + 2 * kAPlane;
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int row0 = blockIdx.x * kTileRows;
  const char* a_lane = smem + (lane & 31) * kALd + hh * 16;
  WRing<NP, kLPF, 1> r, rv;
  EEC_TL_STAMP(glu, 0);
  PlaneRegs<NP> pr;
  planes_issue512<NP>(pr, a.a_hi, a.a_lo, row0, a.M);  // needed first: queued ahead of the weight ring
  ring_fill<NP, kLPF, 1>(r, wfrag_lane(a.wp, w), 0, kD / 16);
  // residual rows of this wave: requested now, consumed after the out-proj GEMM (their latency hides under it)
  float4 xres[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = row0 + w * 8 + i;
    xres[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < a.M) xres[i] = ((const float4*)(a.x + (size_t)row * kD))[lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  planes_commit512<NP>(smem, pr);
  EEC_TL_STAMP(glu, 1);
  __syncthreads();
  EEC_TL_STAMP(glu, 2);
  f32x16 acc[2][1];
  acc_init_bias<2>(acc, a.bias + 32 * w);
  gemm_ring<NP, kD / 16, 1, true, kLPF>(acc, a_lane, kALd, kAPlane, wfrag_lane(a.wp, w), 0, r);
  EEC_TL_STAMP(glu, 3);
  ring_fill<NP, kLPF, 1>(rv, wfrag_lane(gl.wp, w), 0, kD / 16);  // GLU value weights: in flight during the exchange
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    char* dst = lds_e + (mt * 32 + (lane & 31)) * kELd + (32 * w + 4 * hh) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(float4*)(dst + g * 32) = make_float4(acc[mt][0][4 * g], acc[mt][0][4 * g + 1], acc[mt][0][4 * g + 2], acc[mt][0][4 * g + 3]);
  }
  EEC_TL_STAMP(glu, 4);
  __syncthreads();  // tile complete; every wave is done reading the O planes
  EEC_TL_STAMP(glu, 5);
  {
    const float4 g = ((const float4*)gl.ln_g)[lane], bt = ((const float4*)gl.ln_b)[lane];
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = w * 8 + i, row = row0 + rl;
      v[i] = xres[i];
      const float4 e = *(const float4*)(lds_e + rl * kELd + lane * 16);
      v[i].x += e.x, v[i].y += e.y, v[i].z += e.z, v[i].w += e.w;
      if (row < a.M) ((float4*)(a.x + (size_t)row * kD))[lane] = v[i];
    }
    layer_norm_rows<8>(v, g, bt);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = w * 8 + i;
      if (row0 + rl >= a.M) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      const hl2_t s0 = split2<NP>(v[i].x, v[i].y), s1 = split2<NP>(v[i].z, v[i].w);
      h4 hi, lo;
      hi.xy = s0.hi, hi.zw = s1.hi, lo.xy = s0.lo, lo.zw = s1.lo;
      *(h4*)(smem + rl * kALd + lane * 8) = hi;
      if (NP == 3) *(h4*)(smem + kAPlane + rl * kALd + lane * 8) = lo;
    }
  }
  EEC_TL_STAMP(glu, 6);
  __syncthreads();
  EEC_TL_STAMP(glu, 7);
  WRing<NP, kLPF, 1> rg;
  ring_fill<NP, kLPF, 1>(rg, wfrag_lane(gl.wp, 8 + w), 0, kD / 16);
  f32x16 av[2][1], ag[2][1];
  acc_init_bias<2>(av, gl.bias + 32 * w);
  gemm_ring<NP, kD / 16, 1, true, kLPF>(av, a_lane, kALd, kAPlane, wfrag_lane(gl.wp, w), 0, rv);
  EEC_TL_STAMP(glu, 8);
  acc_init_bias<2>(ag, gl.bias + kD + 32 * w);
  gemm_ring<NP, kD / 16, 1, true, kLPF>(ag, a_lane, kALd, kAPlane, wfrag_lane(gl.wp, 8 + w), 0, rg);
  EEC_TL_STAMP(glu, 9);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = row0 + mt * 32 + (lane & 31);
    if (row < a.M) {
      half_t* dst = gl.g + (size_t)row * kD + 32 * w + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gate = ag[mt][0][4 * g + j];
          o[j] = to_half_sat(av[mt][0][4 * g + j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * gate)));
        }
        *(h4*)(dst + 8 * g) = o;
      }
    }
  }
  EEC_TL_STAMP(glu, 10);
}

hipError_t launch_proj_glu(const ProjResArgs& a, const GluArgs& g, int np, hipStream_t st) {
  auto k = np == 3 ? proj_glu_kernel<3> : proj_glu_kernel<1>;
  hipError_t e = ensure_max_lds((const void*)k, kProjGluLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows), dim3(kLinThreads), kProjGluLds, st, a, g);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Exit head.  V <= 256, V % 32 == 0.  Wave w owns vocabulary tile w of the GEMM (idle if 32w >= V).
constexpr int kHeadLds = kLinLds;

template <int NP>
__device__ __forceinline__ void head_body(char* smem, const HeadArgs& a) {
  const int lane = lane_id(), w = wave_id(), hh = lane >> 5;
  const int row0 = blockIdx.x * kTileRows;
  const char* a_lane = smem + (lane & 31) * kALd + hh * 16;
  const bool active = 32 * w < a.V;  // wave-uniform
  WRing<NP, kLPF> r;
  rows_f32_to_planes<NP, false, 8>(smem, a.x, row0, a.M, nullptr, nullptr,
                                   [&]() { if (active) ring_fill<NP, kLPF, 1>(r, wfrag_lane(a.wp, w), 0, kD / 16); });
  __syncthreads();
  f32x16 acc[2][1];
  if (active) {
    acc_init_bias<2>(acc, a.bias + 32 * w);
    gemm_ring<NP, kD / 16, 1, true, kLPF>(acc, a_lane, kALd, kAPlane, wfrag_lane(a.wp, w), 0, r);
  }
  // logits -> fp32 exchange tile (over the dead activation planes) -> each wave finishes 8 whole frames: the
  // log-sum-exp is a wave reduction and every output row leaves as one contiguous store (V * 4 bytes)
  __syncthreads();  // every wave is done reading the planes
  if (active) acc_swapped_to_etile(smem, acc);
  __syncthreads();
  const int c0 = lane * 4;  // this lane's 4 vocabulary entries
  const bool has = c0 < a.V;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int rl = w * 8 + i, row = row0 + rl;
    float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (has) v = *(const float4*)(smem + rl * kELd + c0 * 4);
    const float mx = wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    const float sm = wave_sum(has ? __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx) : 0.f);
    const float lse = mx + __logf(sm);
    if (has && row < a.M) *(float4*)(a.out + (size_t)row * a.V + c0) = make_float4(v.x - lse, v.y - lse, v.z - lse, v.w - lse);
  }
}

template <int NP>
__global__ __launch_bounds__(kLinThreads, 2) void head_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  head_body<NP>(smem, a);
}

// blockIdx.y = exit: the per-exit pointers are picked out of the argument block with a wave-uniform index
template <int NP>
__global__ __launch_bounds__(kLinThreads, 2) void head_batch_kernel(HeadBatchArgs b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int e = blockIdx.y;
  const HeadArgs a{b.x[e], b.M, b.V, b.wp[e], b.bias[e], b.out + (size_t)e * b.M * b.V};
  head_body<NP>(smem, a);
}

hipError_t launch_head(const HeadArgs& a, int np, hipStream_t st) {
  auto k = np == 3 ? head_kernel<3> : head_kernel<1>;
  hipError_t e = ensure_max_lds((const void*)k, kHeadLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows), dim3(kLinThreads), kHeadLds, st, a);
  return hipGetLastError();
}

hipError_t launch_head_batch(const HeadBatchArgs& a, int np, hipStream_t st) {
  if (a.E < 1 || a.E > kMaxHeadExits) return hipErrorInvalidValue;
  auto k = np == 3 ? head_batch_kernel<3> : head_batch_kernel<1>;
  hipError_t e = ensure_max_lds((const void*)k, kHeadLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((a.M + kTileRows - 1) / kTileRows, a.E), dim3(kLinThreads), kHeadLds, st, a);
  return hipGetLastError();
}

}  // namespace eec
