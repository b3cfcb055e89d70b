// Summed per-exit CTC loss, forward (reference train.py:53-65,259):
//   loss_e = mean_b( CTC(logp[e, b], targets[b, :len_b]) / max(len_b, 1) ),  blank = 0,
//   input length = T' for every utterance, zero_infinity=True;  train.py sums loss_e over exits.
// All E*B lattices run in ONE launch: one wave per lattice, the extended label sequence
// (2*len+1 states, <= 8 per lane) lives in registers.  The time recursion is latency-bound (T' serial
// steps), so it runs in a BLOCK-FLOATING linear domain instead of log space:
//   a_t[s] = ( a_{t-1}[s] + a_{t-1}[s-1] + [a_{t-1}[s-2]] ) * p_t(l'_s),   p = exp(logp)
// Each lane keeps its states as fp32 mantissas times a lane-private power of two 2^e (renormalised
// every 2 steps from the lane's own maximum: exact, no log); the two states taken from the previous
// lane arrive with that lane's exponent and both sides are brought to the larger one.  Dynamic range
// ACROSS lanes is therefore unbounded -- necessary, because dead-end prefixes (e.g. "all blank so
// far") can be 1e60 times more probable than the states that will reach the end -- while the
// dependent chain per step is 3 DPP lane shifts, a few ldexp, 2 adds and 1 multiply.  The emission
// log-probs are gathered kCtcAhead steps ahead (their exp is independent of the alpha chain).
#include <limits.h>

#include "eec_kernels.h"

namespace eec {

constexpr int kCtcPerLane = 8;  // up to 512 states = target length <= 255
constexpr int kCtcEmpty = -(1 << 20);  // exponent of a lane that holds no probability mass yet

#define EEC_DPP_F(old, src, ctrl) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(src)), ctrl, 0xf, 0xf, false))
#define EEC_DPP_I(old, src, ctrl) __builtin_amdgcn_update_dpp((int)(old), (int)(src), ctrl, 0xf, 0xf, false)

template <int P>
__global__ __launch_bounds__(64) void ctc_alpha_kernel(const float* __restrict__ logp, const long long* __restrict__ targets,
                                                       const long long* __restrict__ target_len, int B, int Tq, int V,
                                                       int S, int blank, float* __restrict__ nll) {
  const int lat = blockIdx.x, b = lat % B, lane = threadIdx.x;
  const float* lp = logp + (size_t)lat * Tq * V;
  // inputs nn.CTCLoss validates on the host: a length outside [0, S] or a label outside [0, V) would index out of
  // bounds.  Such a lattice is not run on the caller's data: its nll becomes NaN (ctc_reduce_kernel propagates it).
  const long long len_raw = target_len[b];
  bool bad = len_raw < 0 || len_raw > (long long)S;
  const int len = bad ? 0 : (int)len_raw;
  const int L = 2 * len + 1;
  int label[P];
  bool skip_ok[P], live[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int s = lane * P + i;
    label[i] = blank;
    skip_ok[i] = false;
    live[i] = s < L;
    if (s < L && (s & 1)) {
      const int k = s >> 1;
      const long long lab = targets[(size_t)b * S + k];
      if (lab < 0 || lab >= (long long)V) bad = true;
      label[i] = (lab < 0 || lab >= (long long)V) ? blank : (int)lab;
      skip_ok[i] = k > 0 && lab != targets[(size_t)b * S + k - 1];
    }
  }
  bad = __any((int)bad) != 0;  // wave-uniform
  float livef[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    livef[i] = live[i] ? 1.f : 0.f;
    // keep every label in a VGPR: a label the compiler can prove wave-uniform (the blanks) would turn
    // its gather into s_load + s_waitcnt lgkmcnt(0), which serialises the prefetch ring
    asm volatile("" : "+v"(label[i]));
  }
  float alpha[P];
  int ex = kCtcEmpty;  // this lane's states are alpha[i] * 2^ex
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int s = lane * P + i;
    alpha[i] = (s < 2 && s < L) ? __expf(lp[label[i]]) : 0.f;
  }
  if (lane == 0) ex = 0;
  // emission log-probs are gathered kCtcAhead steps ahead of their use (exponentiated when used)
  constexpr int kCtcAhead = P <= 4 ? 16 : 8;
  float emit[kCtcAhead][P];
#pragma unroll
  for (int d = 0; d < kCtcAhead; ++d)
#pragma unroll
    for (int i = 0; i < P; ++i) emit[d][i] = lp[(size_t)min(1 + d, Tq - 1) * V + label[i]];
  // one time step, branch-free (selects only) so that the unrolled group below is a single basic
  // block and the emission loads keep their kCtcAhead-step lead (s_waitcnt vmcnt(N), not vmcnt(0))
  auto step = [&](const float (&em)[P], bool renorm) {
    // the previous lane's last two states and its exponent (lane 0 receives 0 / its own exponent)
    float up1 = EEC_DPP_F(0.f, alpha[P - 1], 0x138);  // wave_shr:1
    float up2 = EEC_DPP_F(0.f, alpha[P - 2], 0x138);
    const int ex_up = EEC_DPP_I(ex, ex, 0x138);
    const int ec = max(ex, ex_up);  // common scale of this step
    const int d_own = max(ex - ec, -200), d_up = max(ex_up - ec, -200);
    up1 = ldexpf(up1, d_up);
    up2 = ldexpf(up2, d_up);
    float cur[P];
#pragma unroll
    for (int i = 0; i < P; ++i) cur[i] = ldexpf(alpha[i], d_own);
    ex = ec;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const float p1 = i >= 1 ? cur[i - 1] : up1;
      const float p2 = i >= 2 ? cur[i - 2] : (i == 1 ? up1 : up2);
      // exp(em) * livef is independent of the alpha chain; states beyond 2*len+1 are zeroed by livef
      alpha[i] = (cur[i] + p1 + (skip_ok[i] ? p2 : 0.f)) * (__expf(em[i]) * livef[i]);
    }
    if (renorm) {  // compile-time: renormalise this lane every second step
      float m = alpha[0];
#pragma unroll
      for (int i = 1; i < P; ++i) m = fmaxf(m, alpha[i]);
      const bool any = m > 0.f;
      const int e = any ? (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127 : 0;
#pragma unroll
      for (int i = 0; i < P; ++i) alpha[i] = ldexpf(alpha[i], -e);
      ex = any ? ex + e : kCtcEmpty;
    }
  };
  int t0 = 1;
  for (; t0 + kCtcAhead <= Tq; t0 += kCtcAhead) {  // full groups
#pragma unroll
    for (int d = 0; d < kCtcAhead; ++d) {
      step(emit[d], (d & 1) != 0);
      const int tn = min(t0 + d + kCtcAhead, Tq - 1);  // clamped: a harmless re-read near the end
#pragma unroll
      for (int i = 0; i < P; ++i) emit[d][i] = lp[(size_t)tn * V + label[i]];  // raw: exp at use
    }
  }
#pragma unroll
  for (int d = 0; d < kCtcAhead; ++d)  // ragged tail (wave-uniform guard); its emissions are already in the ring
    if (t0 + d < Tq) step(emit[d], (d & 1) != 0);
  // p(target) = a[L-1] + a[L-2]: at most two lanes contribute, each with its own exponent
  float tail = 0.f;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int s = lane * P + i;
    if (s == L - 1 || s == L - 2) tail += alpha[i];
  }
  bad = bad || __any((int)(tail != tail)) != 0;  // NaN log-probs that reach the final states (fmaxf / > drop NaNs silently)
  const int e_lane = tail > 0.f ? ex : kCtcEmpty;
  const int e_max = (int)wave_max((float)e_lane);  // exponents are small integers: exact in fp32
  const float total = wave_sum(tail > 0.f ? ldexpf(tail, max(e_lane - e_max, -200)) : 0.f);
  // a NaN emission makes `total` NaN: (NaN > 0) is false, so test it explicitly instead of reporting "infeasible"
  if (lane == 0)
    nll[lat] = (bad || total != total) ? __builtin_nanf("") : (total > 0.f) ? -(logf(total) + (float)e_max * 0.6931471805599453f) : INFINITY;
}

// loss_e = mean_b( zero_inf(nll[e][b]) / max(len_b, 1) ): fixed summation order (bitwise reproducible)
__global__ void ctc_reduce_kernel(const float* nll, const long long* target_len, int B, float* out) {
  const int e = blockIdx.x;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      float v = nll[e * B + b];
      if (isinf(v)) v = 0.f;  // zero_infinity=True zeroes infinite losses only: NaN (bad input) propagates
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
