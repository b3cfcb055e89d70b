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

// STORE: every step's scaled alphas and lane exponents are also written to `astore` (layout ctc_store_index below):
// the backward pass multiplies them with the betas it recomputes (ctc_beta_kernel).
__device__ __forceinline__ size_t ctc_store_index(int lat, int Tq, int P, int t, int k, int lane) {
  return (((size_t)lat * Tq + t) * (P + 1) + k) * 64 + lane;  // k < P: state lane*P + k; k == P: the lane's exponent
}

template <int P, bool STORE>
__global__ __launch_bounds__(64) void ctc_alpha_kernel(const float* __restrict__ logp, const long long* __restrict__ targets,
                                                       const long long* __restrict__ target_len, int B, int Tq, int V,
                                                       int S, int blank, float* __restrict__ nll, float* __restrict__ astore) {
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
  auto store = [&](int t) {
    if constexpr (STORE) {
#pragma unroll
      for (int i = 0; i < P; ++i) astore[ctc_store_index(lat, Tq, P, t, i, lane)] = alpha[i];
      astore[ctc_store_index(lat, Tq, P, t, P, lane)] = __builtin_bit_cast(float, ex);
    }
  };
  store(0);
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
      store(t0 + d);
      const int tn = min(t0 + d + kCtcAhead, Tq - 1);  // clamped: a harmless re-read near the end
#pragma unroll
      for (int i = 0; i < P; ++i) emit[d][i] = lp[(size_t)tn * V + label[i]];  // raw: exp at use
    }
  }
#pragma unroll
  for (int d = 0; d < kCtcAhead; ++d)  // ragged tail (wave-uniform guard); its emissions are already in the ring
    if (t0 + d < Tq) {
      step(emit[d], (d & 1) != 0);
      store(t0 + d);
    }
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
  if constexpr (STORE) {  // p(target) = total * 2^e_max, kept exactly for the backward pass (nll alone rounds it to ~1e-4 relative)
    if (lane == 0) {
      float* pinfo = astore + ctc_store_index(gridDim.x, Tq, P, 0, 0, 0) + 2 * (size_t)lat;
      pinfo[0] = (bad || total != total) ? __builtin_nanf("") : total;
      pinfo[1] = __builtin_bit_cast(float, e_max);
    }
  }
}

// ---------------------------------------------------------------------------
// Backward (gradient of the summed per-exit loss with respect to the log-probs; reference: loss.backward() through
// nn.CTCLoss, train.py:60-68).  torch's CTC backward returns, for a lattice with upstream gradient g,
//     dlogp[t][c] = g * ( exp(logp[t][c]) - gamma_t(c) ),    gamma_t(c) = sum_{s: l'_s = c} alpha_t(s) beta'_t(s) / p(target)
// (the gradient with respect to the logits of a log-softmax: it sums to zero over c), with
//     beta'_{T-1}(s) = [s is one of the last two states],
//     beta'_{t-1}(s) = sum_{s' in {s, s+1, s+2 if allowed}} beta'_t(s') * p_t(l'_{s'}).
// ctc_beta_kernel walks t downwards with the same block-floating representation as the forward pass (one wave per
// lattice, neighbour states of the NEXT lane through a DPP wave shift) and overwrites the stored alphas with the state
// posteriors gamma_t(s); ctc_grad_kernel then turns them into the dense gradient, one wave per (lattice, frame).
template <int P>
__global__ __launch_bounds__(64) void ctc_beta_kernel(const float* __restrict__ logp, const long long* __restrict__ targets,
                                                      const long long* __restrict__ target_len, int B, int Tq, int V,
                                                      int S, int blank, const float* __restrict__ nll, float* __restrict__ astore) {
  const int lat = blockIdx.x, b = lat % B, lane = threadIdx.x;
  const float* lp = logp + (size_t)lat * Tq * V;
  const long long len_raw = target_len[b];
  const int len = (len_raw < 0 || len_raw > (long long)S) ? 0 : (int)len_raw;
  const int L = 2 * len + 1;
  int label[P];
  bool skip_ok[P];
  float livef[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int s = lane * P + i;
    label[i] = blank;
    skip_ok[i] = false;
    livef[i] = s < L ? 1.f : 0.f;
    if (s < L && (s & 1)) {
      const int k = s >> 1;
      const long long lab = targets[(size_t)b * S + k];
      label[i] = (lab < 0 || lab >= (long long)V) ? blank : (int)lab;
      skip_ok[i] = k > 0 && lab != targets[(size_t)b * S + k - 1];
    }
    asm volatile("" : "+v"(label[i]));
  }
  // transition s -> s + 2 is allowed when state s + 2 may be entered by a skip; states s + 1, s + 2 of the last slots of a
  // lane live in the next lane
  bool skip_dn[P];
  {
    const int n0 = EEC_DPP_I(0, (int)skip_ok[0], 0x130), n1 = EEC_DPP_I(0, (int)skip_ok[1], 0x130);  // wave_shl:1
#pragma unroll
    for (int i = 0; i < P; ++i) skip_dn[i] = i + 2 < P ? skip_ok[i + 2] : ((i + 2 - P == 0 ? n0 : n1) != 0);
  }
  // p(target) = total * 2^e_max exactly as the forward pass left it: gamma = alpha beta' 2^(ea + eb - e_max) / total
  const float* pinfo = astore + ctc_store_index(gridDim.x, Tq, P, 0, 0, 0) + 2 * (size_t)lat;
  const float ptot = pinfo[0];
  const bool usable = ptot > 0.f && ptot < INFINITY;  // false for an infeasible (0) or invalid (NaN) lattice
  const int ishift = usable ? -__builtin_bit_cast(int, pinfo[1]) : 0;
  const float fscale = usable ? 1.0f / ptot : 0.f;
  float beta[P];
  int eb = kCtcEmpty;
  bool any0 = false;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int s = lane * P + i;
    beta[i] = (s < L && (s == L - 1 || s == L - 2)) ? 1.f : 0.f;
    any0 = any0 || beta[i] != 0.f;
  }
  if (any0) eb = 0;
  constexpr int kAhead = P <= 4 ? 8 : 4;
  float emit[kAhead][P], al[kAhead][P + 1];
  auto fetch = [&](int slot, int t) {  // emissions and stored alphas of time t (clamped: harmless re-reads below t = 0)
    const int tc = max(t, 0);
#pragma unroll
    for (int i = 0; i < P; ++i) emit[slot][i] = lp[(size_t)tc * V + label[i]];
#pragma unroll
    for (int k = 0; k <= P; ++k) al[slot][k] = astore[ctc_store_index(lat, Tq, P, tc, k, lane)];
  };
#pragma unroll
  for (int d = 0; d < kAhead; ++d) fetch(d, Tq - 1 - d);
  auto step = [&](int slot, int t, bool renorm) {
    // posteriors of time t, in place of the stored alphas
    const int ea = __builtin_bit_cast(int, al[slot][P]);
    const int sh = max(min(ea + eb + ishift, 126), -300);
#pragma unroll
    for (int i = 0; i < P; ++i)
      astore[ctc_store_index(lat, Tq, P, t, i, lane)] = ldexpf(al[slot][i] * beta[i] * fscale, sh) * livef[i];
    // beta'_{t-1}
    float bw[P];
#pragma unroll
    for (int i = 0; i < P; ++i) bw[i] = beta[i] * (__expf(emit[slot][i]) * livef[i]);
    float dn1 = EEC_DPP_F(0.f, bw[0], 0x130);  // next lane's first two weighted betas and its exponent
    float dn2 = EEC_DPP_F(0.f, bw[1], 0x130);
    const int eb_dn = EEC_DPP_I(kCtcEmpty, eb, 0x130);
    const int ec = max(eb, eb_dn);
    const int d_own = max(eb - ec, -200), d_dn = max(eb_dn - ec, -200);
    dn1 = ldexpf(dn1, d_dn);
    dn2 = ldexpf(dn2, d_dn);
#pragma unroll
    for (int i = 0; i < P; ++i) bw[i] = ldexpf(bw[i], d_own);
    eb = ec;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const float n1 = i + 1 < P ? bw[i + 1 < P ? i + 1 : 0] : dn1;
      const float n2 = i + 2 < P ? bw[i + 2 < P ? i + 2 : 0] : (i + 2 - P == 0 ? dn1 : dn2);
      beta[i] = bw[i] + n1 + (skip_dn[i] ? n2 : 0.f);
    }
    if (renorm) {
      float m = beta[0];
#pragma unroll
      for (int i = 1; i < P; ++i) m = fmaxf(m, beta[i]);
      const bool any = m > 0.f;
      const int e = any ? (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127 : 0;
#pragma unroll
      for (int i = 0; i < P; ++i) beta[i] = ldexpf(beta[i], -e);
      eb = any ? eb + e : kCtcEmpty;
    }
  };
  int t = Tq - 1;
  for (; t - kAhead + 1 >= 0; t -= kAhead) {
#pragma unroll
    for (int d = 0; d < kAhead; ++d) {
      step(d, t - d, (d & 1) != 0);
      fetch(d, t - d - kAhead);
    }
  }
#pragma unroll
  for (int d = 0; d < kAhead; ++d)
    if (t - d >= 0) step(d, t - d, (d & 1) != 0);
}

// dlogp[lat][t][c] = gs * (exp(logp) - sum of the posteriors of the states labelled c), gs = grad_loss[e] / (B max(len, 1))
// (0 for an infeasible lattice: zero_infinity; NaN for a lattice whose loss is NaN).  One wave per (lattice, frame).
template <int P>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logp, const long long* __restrict__ targets,
                                                       const long long* __restrict__ target_len, int B, int Tq, int V, int S,
                                                       int blank, const float* __restrict__ nll, const float* __restrict__ astore,
                                                       const float* __restrict__ grad_loss, int n_rows, float* __restrict__ dlogp) {
  __shared__ float bins_all[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  float* bins = bins_all[w];
  const bool row_ok = row < n_rows;
  const int lat = row_ok ? row / Tq : 0, t = row_ok ? row - lat * Tq : 0, b = lat % B, e = lat / B;
  *(float4*)(bins + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const long long len_raw = target_len[b];
  const int len = (len_raw < 0 || len_raw > (long long)S) ? 0 : (int)len_raw;
  const int L = 2 * len + 1;
  if (row_ok) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int s = lane * P + i;
      if (s < L) {
        int lab = blank;
        if (s & 1) {
          const long long lr = targets[(size_t)b * S + (s >> 1)];
          lab = (lr < 0 || lr >= (long long)V) ? blank : (int)lr;
        }
        atomicAdd(&bins[lab], astore[ctc_store_index(lat, Tq, P, t, i, lane)]);
      }
    }
  }
  __syncthreads();
  if (!row_ok) return;
  const float nl = nll[lat];
  float gs = grad_loss[e] / ((float)B * (float)(len > 0 ? len : 1));
  if (nl != nl) gs = nl;            // NaN loss: NaN gradient
  else if (!(nl < INFINITY)) gs = 0.f;  // zero_infinity
  const int c0 = lane * 4;
  if (c0 < V) {
    const size_t off = ((size_t)lat * Tq + t) * V + c0;
    const float4 l = *(const float4*)(logp + off);
    const float4 g = *(const float4*)(bins + c0);
    float4 o;
    o.x = gs * (__expf(l.x) - g.x), o.y = gs * (__expf(l.y) - g.y), o.z = gs * (__expf(l.z) - g.z), o.w = gs * (__expf(l.w) - g.w);
    if (gs == 0.f) o = make_float4(0.f, 0.f, 0.f, 0.f);  // an infeasible lattice may hold inf / NaN posteriors
    *(float4*)(dlogp + off) = o;
  }
}

int ctc_states_per_lane(int S) {
  if (2 * S + 1 <= 64 * 2) return 2;
  if (2 * S + 1 <= 64 * 4) return 4;
  if (2 * S + 1 <= 64 * kCtcPerLane) return kCtcPerLane;
  return 0;
}

size_t ctc_store_floats(int E, int B, int Tq, int S) {
  const int P = ctc_states_per_lane(S);
  return P ? (size_t)E * B * Tq * (P + 1) * 64 + 2 * (size_t)E * B : 0;
}

hipError_t launch_ctc_backward(const float* logp, const long long* targets, const long long* target_len, int E, int B, int Tq,
                               int V, int S, int blank, const float* nll, float* astore, const float* grad_loss, float* dlogp,
                               hipStream_t st) {
  const int P = ctc_states_per_lane(S), n_rows = E * B * Tq;
  if (!P || V > 256 || V % 4) return hipErrorInvalidValue;
#define EEC_CTC_BWD(P_)                                                                                                      \
  hipLaunchKernelGGL(ctc_beta_kernel<P_>, dim3(E * B), dim3(64), 0, st, logp, targets, target_len, B, Tq, V, S, blank, nll,  \
                     astore);                                                                                               \
  hipLaunchKernelGGL(ctc_grad_kernel<P_>, dim3((n_rows + 3) / 4), dim3(256), 0, st, logp, targets, target_len, B, Tq, V, S,  \
                     blank, nll, astore, grad_loss, n_rows, dlogp);
  if (P == 2) {
    EEC_CTC_BWD(2)
  } else if (P == 4) {
    EEC_CTC_BWD(4)
  } else {
    EEC_CTC_BWD(kCtcPerLane)
  }
#undef EEC_CTC_BWD
  return hipGetLastError();
}

// loss_e = mean_b( zero_inf(nll[e][b]) / max(len_b, 1) ): fixed summation order (bitwise reproducible).  The 64 lanes fetch 64
// utterances' terms at once; lane order is then added up sequentially out of registers (one thread walking the batch with two
// dependent loads per utterance took 10.7 us per launch: 0.4 % of the headline step).
__global__ void ctc_reduce_kernel(const float* nll, const long long* target_len, int B, float* out) {
  const int e = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int b = b0 + lane;
    float term = 0.f;
    if (b < B) {
      float v = nll[e * B + b];
      if (isinf(v)) v = 0.f;  // zero_infinity=True zeroes infinite losses only: NaN (bad input) propagates
      const long long l = target_len[b] > 0 ? target_len[b] : 1;
      term = v / (float)l;
    }
    const int n = min(64, B - b0);
    for (int i = 0; i < n; ++i) s += __shfl(term, i, 64);
  }
  if (lane == 0) out[e] = s / (float)B;
}

hipError_t launch_ctc_loss(const float* logp, const long long* targets, const long long* target_len, int E, int B, int Tq,
                           int V, int S, int blank, float* nll, float* out, float* astore, hipStream_t st) {
  // state count 2*S+1 must fit 64 lanes x P
  const int P = ctc_states_per_lane(S);
  if (!P) return hipErrorInvalidValue;
#define EEC_CTC_FWD(P_)                                                                                                        \
  if (astore)                                                                                                                  \
    hipLaunchKernelGGL((ctc_alpha_kernel<P_, true>), dim3(E * B), dim3(64), 0, st, logp, targets, target_len, B, Tq, V, S,     \
                       blank, nll, astore);                                                                                    \
  else                                                                                                                         \
    hipLaunchKernelGGL((ctc_alpha_kernel<P_, false>), dim3(E * B), dim3(64), 0, st, logp, targets, target_len, B, Tq, V, S,    \
                       blank, nll, astore);
  if (P == 2) {
    EEC_CTC_FWD(2)
  } else if (P == 4) {
    EEC_CTC_FWD(4)
  } else {
    EEC_CTC_FWD(kCtcPerLane)
  }
#undef EEC_CTC_FWD
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ctc_reduce_kernel, dim3(E), dim3(64), 0, st, nll, target_len, B, out);
  return hipGetLastError();
}

// dlogits = g - exp(logp) * sum_c g  (backward of log_softmax over the last axis); one wave per row, V <= 256, V % 4 == 0
__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(const float* __restrict__ logp, const float* __restrict__ g, int M, int V,
                                                             float* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int c0 = lane * 4;
  float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), lv = gv;
  if (c0 < V) {
    gv = *(const float4*)(g + (size_t)row * V + c0);
    lv = *(const float4*)(logp + (size_t)row * V + c0);
  }
  const float sum = wave_sum(gv.x + gv.y + gv.z + gv.w);
  if (c0 < V)
    *(float4*)(dlogits + (size_t)row * V + c0) = make_float4(gv.x - __expf(lv.x) * sum, gv.y - __expf(lv.y) * sum,
                                                             gv.z - __expf(lv.z) * sum, gv.w - __expf(lv.w) * sum);
}

hipError_t launch_logsoftmax_backward(const float* logp, const float* g, int M, int V, float* dlogits, hipStream_t st) {
  if (V > 256 || V % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, st, logp, g, M, V, dlogits);
  return hipGetLastError();
}

}  // namespace eec
