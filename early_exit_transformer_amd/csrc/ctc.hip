// Summed per-exit CTC loss, forward (reference train.py:53-65,259):
//   loss_e = mean_b( CTC(logp[e, b], targets[b, :len_b]) / max(len_b, 1) ),  blank = 0,
//   input length = T' for every utterance, zero_infinity=True;  train.py sums loss_e over exits.
// All E*B lattices run in ONE launch: one wave per lattice, the extended label sequence
// (2*len+1 states, <= 8 per lane) lives in registers, the time recursion
//   alpha_t[s] = logsumexp(alpha_{t-1}[s], alpha_{t-1}[s-1], [alpha_{t-1}[s-2]]) + logp[t][l'_s]
// takes its s-1 / s-2 neighbours from the previous lane with two shuffles per step, and the
// emission row of step t+1 is gathered while step t is computed.  fp32 log space (as torch).
#include "eec_kernels.h"

namespace eec {

constexpr int kCtcPerLane = 8;  // up to 512 states = target length <= 255
constexpr float kNegInf = -INFINITY;

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == kNegInf) return kNegInf;
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == kNegInf) return kNegInf;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

template <int P>
__global__ __launch_bounds__(64) void ctc_alpha_kernel(const float* __restrict__ logp, const long long* __restrict__ targets,
                                                       const long long* __restrict__ target_len, int B, int Tq, int V,
                                                       int S, int blank, float* __restrict__ nll) {
  const int lat = blockIdx.x, b = lat % B, lane = threadIdx.x;
  const float* lp = logp + (size_t)lat * Tq * V;
  const int len = (int)target_len[b];
  const int L = 2 * len + 1;
  int label[P];
  bool skip_ok[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int s = lane * P + i;
    label[i] = blank;
    skip_ok[i] = false;
    if (s < L && (s & 1)) {
      const int k = s >> 1;
      label[i] = (int)targets[(size_t)b * S + k];
      skip_ok[i] = k > 0 && label[i] != (int)targets[(size_t)b * S + k - 1];
    }
  }
  float alpha[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int s = lane * P + i;
    alpha[i] = (s < 2 && s < L) ? lp[label[i]] : kNegInf;
  }
  // emissions are gathered kCtcAhead time steps ahead of their use (each gather is an L2 round trip)
  constexpr int kCtcAhead = 8;
  float emit[kCtcAhead][P];
#pragma unroll
  for (int d = 0; d < kCtcAhead; ++d)
#pragma unroll
    for (int i = 0; i < P; ++i) emit[d][i] = (1 + d < Tq) ? lp[(size_t)(1 + d) * V + label[i]] : 0.f;
  for (int t0 = 1; t0 < Tq; t0 += kCtcAhead) {
#pragma unroll
    for (int d = 0; d < kCtcAhead; ++d) {
      const int t = t0 + d;
      if (t < Tq) {  // wave-uniform
        // neighbours from the previous lane: its last two states
        float up1 = __shfl_up(alpha[P - 1], 1, 64), up2 = __shfl_up(alpha[P - 2], 1, 64);
        if (lane == 0) up1 = up2 = kNegInf;
        float nxt[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
          // state s-1 / s-2: in this lane, or the previous lane's last (up1) / second to last (up2)
          const float p1 = i >= 1 ? alpha[i - 1] : up1;
          const float p2 = i >= 2 ? alpha[i - 2] : (i == 1 ? up1 : up2);
          nxt[i] = (skip_ok[i] ? lse3(alpha[i], p1, p2) : lse2(alpha[i], p1)) + emit[d][i];
        }
#pragma unroll
        for (int i = 0; i < P; ++i) alpha[i] = (lane * P + i < L) ? nxt[i] : kNegInf;
        const int tn = t + kCtcAhead;
        if (tn < Tq) {
#pragma unroll
          for (int i = 0; i < P; ++i) emit[d][i] = lp[(size_t)tn * V + label[i]];
        }
      }
    }
  }
  // -log( alpha[L-1] + alpha[L-2] )
  float last = kNegInf, prev = kNegInf;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int s = lane * P + i;
    if (s == L - 1) last = alpha[i];
    if (s == L - 2) prev = alpha[i];
  }
  last = wave_max(last);
  prev = wave_max(prev);
  if (lane == 0) nll[lat] = -lse2(last, prev);
}

// loss_e = mean_b( zero_inf(nll[e][b]) / max(len_b, 1) ): fixed summation order (bitwise reproducible)
__global__ void ctc_reduce_kernel(const float* nll, const long long* target_len, int B, float* out) {
  const int e = blockIdx.x;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      float v = nll[e * B + b];
      if (isinf(v) || isnan(v)) v = 0.f;  // zero_infinity=True
      const long long l = target_len[b] > 0 ? target_len[b] : 1;
      s += v / (float)l;
    }
    out[e] = s / (float)B;
  }
}

hipError_t launch_ctc_loss(const float* logp, const long long* targets, const long long* target_len, int E, int B, int Tq,
                           int V, int S, int blank, float* nll, float* out, hipStream_t st) {
  // state count 2*S+1 must fit 64 lanes x P
  if (2 * S + 1 <= 64 * 2)
    hipLaunchKernelGGL(ctc_alpha_kernel<2>, dim3(E * B), dim3(64), 0, st, logp, targets, target_len, B, Tq, V, S, blank, nll);
  else if (2 * S + 1 <= 64 * 4)
    hipLaunchKernelGGL(ctc_alpha_kernel<4>, dim3(E * B), dim3(64), 0, st, logp, targets, target_len, B, Tq, V, S, blank, nll);
  else if (2 * S + 1 <= 64 * kCtcPerLane)
    hipLaunchKernelGGL(ctc_alpha_kernel<kCtcPerLane>, dim3(E * B), dim3(64), 0, st, logp, targets, target_len, B, Tq, V, S, blank, nll);
  else
    return hipErrorInvalidValue;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ctc_reduce_kernel, dim3(E), dim3(64), 0, st, nll, target_len, B, out);
  return hipGetLastError();
}

}  // namespace eec
