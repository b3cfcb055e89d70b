// CTC prefix beam search on the device: what the reference's CTC inference calls through
// torchaudio's cuda_ctc_decoder(nbest=1, beam_size=10, blank_skip_threshold=0.95) (util/beam_infer.py:79-80,102-112,
// inference.py:65-77).  That decoder is third-party CUDA code outside the reference tree: the algorithm restated here is
// the published prefix beam search without a language model (oracle/ctc_beam_ref.py is its CPU statement, and the judge
// of this kernel) -- per frame every prefix stays or is extended by a label, extensions that spell an existing prefix
// merge into it, the `beam` most probable prefixes survive; frames whose blank probability exceeds the threshold are taken
// as blank frames without expansion.
//
// One 256-thread workgroup per sequence; thread c owns label c (V <= 256): its log-prob and its `beam` extension scores
// live in registers, the beam (<= 16 prefixes: two log-probabilities, last label, length, 64-bit prefix hash) in LDS.
// Survivors are picked by `beam` rounds of a block-wide arg-max (DPP wave maximum + ballot, 4 wave winners through LDS);
// ties go to the lower candidate id.  Prefixes are not copied: every frame records (parent beam, appended label) per
// survivor in a back-pointer table and the best prefix is read off backwards at the end.  Latency-bound integer / scalar
// work (T' serial frames): it is sized to keep all E*B sequences of a batch in flight at once, not for the roofline.
#include <limits.h>

#include "../../include/eec.h"
#include "eec_kernels.h"

namespace eec {

constexpr int kCbMaxBeam = 16;
constexpr int kCbThreads = 256;

struct CbBeam {
  float pb, pnb;
  int last, len;
  unsigned long long hash;
};

__device__ __forceinline__ float cb_lae(float a, float b) {  // log(exp a + exp b), -inf safe
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + log1pf(__expf(-fabsf(a - b)));
}
__device__ __forceinline__ unsigned long long cb_mix(unsigned long long h, int c) {
  unsigned long long z = h ^ ((unsigned long long)(c + 1) * 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(kCbThreads) void ctc_beam_kernel(const float* __restrict__ logp, int Tq, int V, int blank, int beam,
                                                              float log_thr, int skip_drops, int* __restrict__ backptr, int* __restrict__ tokens,
                                                              int* __restrict__ counts, float* __restrict__ scores) {
  __shared__ CbBeam bufs[2][kCbMaxBeam];
  __shared__ float tot[kCbMaxBeam], stay_pb[kCbMaxBeam], stay_pnb[kCbMaxBeam], row[256];
  __shared__ unsigned merge_mask[kCbMaxBeam];  // bit i of word j: extension of beam i by last[j] spells beam j
  __shared__ float red_v[4];
  __shared__ int red_id[4];
  __shared__ int nb_s;
  const int seq = blockIdx.x, c = threadIdx.x, lane = c & 63, w = c >> 6;
  const float* lp_seq = logp + (size_t)seq * Tq * V;
  int* bp = backptr + (size_t)seq * Tq * kCbMaxBeam;
  int cur = 0;
  if (c == 0) {
    bufs[0][0] = CbBeam{0.f, -INFINITY, -1, 0, 0x243F6A8885A308D3ull};
    nb_s = 1;
  }
  __syncthreads();
  for (int t = 0; t < Tq; ++t) {
    const CbBeam* B = bufs[cur];
    CbBeam* N = bufs[cur ^ 1];
    const int nb = nb_s;
    const float lpc = c < V ? lp_seq[(size_t)t * V + c] : -INFINITY;
    row[c] = lpc;
    __syncthreads();
    const float lpb = row[blank];
    if (lpb > log_thr) {  // blank-dominated frame, not expanded
      if (c < nb) {
        // skip_drops == 0: the frame counts as a blank frame (all mass ends in blank: a label repeated across it stays a repeat)
        // skip_drops != 0: the frame is DROPPED, as if the sequence were one frame shorter (the repeat collapses)
        N[c] = skip_drops ? B[c] : CbBeam{cb_lae(B[c].pb, B[c].pnb) + lpb, -INFINITY, B[c].last, B[c].len, B[c].hash};
        bp[t * kCbMaxBeam + c] = c << 16;
      }
      __syncthreads();
      cur ^= 1;
      continue;
    }
    if (c < nb) {
      const float tt = cb_lae(B[c].pb, B[c].pnb);
      tot[c] = tt;
      stay_pb[c] = tt + lpb;
      stay_pnb[c] = B[c].last >= 0 ? B[c].pnb + row[B[c].last] : -INFINITY;
      // which beams i, extended by this beam's last label, spell this beam?
      unsigned m = 0;
      if (B[c].last >= 0)
        for (int i = 0; i < nb; ++i)
          if (i != c && B[i].len + 1 == B[c].len && cb_mix(B[i].hash, B[c].last) == B[c].hash) m |= 1u << i;
      merge_mask[c] = m;
    }
    __syncthreads();
    if (c < nb && merge_mask[c]) {  // fold those extensions into this beam's "stay" candidate (ascending i: fixed order)
      float acc = stay_pnb[c];
      for (int i = 0; i < nb; ++i)
        if (merge_mask[c] >> i & 1) acc = cb_lae(acc, (B[i].last == B[c].last ? B[i].pb : tot[i]) + row[B[c].last]);
      stay_pnb[c] = acc;
    }
    __syncthreads();
    // this label's extension of every beam (killed where it merged into an existing prefix), and the stays
    float ext[kCbMaxBeam];
#pragma unroll
    for (int i = 0; i < kCbMaxBeam; ++i) {
      float v = -INFINITY;
      if (i < nb && c != blank && c < V) {
        v = (B[i].last == c ? B[i].pb : tot[i]) + lpc;
        for (int j = 0; j < nb; ++j)
          if (B[j].last == c && (merge_mask[j] >> i & 1)) v = -INFINITY;
      }
      ext[i] = v;
    }
    float stay = c < nb ? cb_lae(stay_pb[c], stay_pnb[c]) : -INFINITY;
    // `beam` rounds of block-wide arg-max; candidate id: stays = beam index (0 .. 15), extensions = 16 + 16 c + i
    int n_new = 0;
    for (int r = 0; r < beam; ++r) {
      float bv = stay;
      int bid = c < nb ? c : INT_MAX;
#pragma unroll
      for (int i = 0; i < kCbMaxBeam; ++i)
        if (ext[i] > bv || (ext[i] == bv && ext[i] > -INFINITY && 16 + 16 * c + i < bid)) {
          bv = ext[i];
          bid = 16 + 16 * c + i;
        }
      if (bv == -INFINITY) bid = INT_MAX;
      const float wm = wave_max(bv);
      int cand = (bv == wm && wm > -INFINITY) ? bid : INT_MAX;
      // lowest id among the lanes that hold the wave maximum
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
      if (lane == 0) {
        red_v[w] = wm;
        red_id[w] = cand;
      }
      __syncthreads();
      float gv = red_v[0];
      int gid = red_id[0];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (red_v[k] > gv || (red_v[k] == gv && red_id[k] < gid)) {
          gv = red_v[k];
          gid = red_id[k];
        }
      if (gv == -INFINITY || gid == INT_MAX) {
        __syncthreads();
        break;
      }
      if (gid < 16) {  // a stay: its owner is thread gid
        if (c == gid) {
          N[r] = CbBeam{stay_pb[c], stay_pnb[c], B[c].last, B[c].len, B[c].hash};
          bp[t * kCbMaxBeam + r] = c << 16;
          stay = -INFINITY;
        }
      } else {
        const int cc = (gid - 16) >> 4, ii = (gid - 16) & 15;
        if (c == cc) {
          N[r] = CbBeam{-INFINITY, gv, cc, B[ii].len + 1, cb_mix(B[ii].hash, cc)};
          bp[t * kCbMaxBeam + r] = (ii << 16) | (cc + 1);
#pragma unroll
          for (int i = 0; i < kCbMaxBeam; ++i)
            if (i == ii) ext[i] = -INFINITY;
        }
      }
      n_new = r + 1;
      __syncthreads();
    }
    if (c == 0) nb_s = n_new;
    __syncthreads();
    cur ^= 1;
  }
  // best prefix: highest total, ties to the lower beam index; read the labels off the back-pointers
  if (c == 0) {
    const CbBeam* B = bufs[cur];
    int best = 0;
    float bs = -INFINITY;
    for (int i = 0; i < nb_s; ++i) {
      const float s = cb_lae(B[i].pb, B[i].pnb);
      if (s > bs) {
        bs = s;
        best = i;
      }
    }
    int n = B[best].len, k = best;
    int* out = tokens + (size_t)seq * Tq;
    counts[seq] = n;
    scores[seq] = bs;
    for (int t = Tq - 1; t >= 0 && n > 0; --t) {
      const int e = bp[t * kCbMaxBeam + k];
      if (e & 0xffff) out[--n] = (e & 0xffff) - 1;
      k = e >> 16;
    }
  }
}

hipError_t launch_ctc_beam(const float* logp, int n_seq, int Tq, int V, int blank, int beam, float blank_skip_threshold, int skip_drops,
                           int* backptr, int* tokens, int* counts, float* scores, hipStream_t st) {
  if (V < 1 || V > 256 || beam < 1 || beam > kCbMaxBeam || blank < 0 || blank >= V) return hipErrorInvalidValue;
  const float log_thr = (blank_skip_threshold > 0.f && blank_skip_threshold < 1.f) ? logf(blank_skip_threshold) : INFINITY;
  hipLaunchKernelGGL(ctc_beam_kernel, dim3(n_seq), dim3(kCbThreads), 0, st, logp, Tq, V, blank, beam, log_thr, skip_drops, backptr, tokens,
                     counts, scores);
  return hipGetLastError();
}

}  // namespace eec

extern "C" {

size_t eec_ctc_beam_workspace_bytes(int n_seq, int Tq) {
  return n_seq > 0 && Tq > 0 ? (size_t)n_seq * Tq * eec::kCbMaxBeam * sizeof(int) : 0;
}

int eec_ctc_beam_decode(const float* logp, int n_seq, int Tq, int V, int blank, int beam_size, float blank_skip_threshold,
                        void* workspace, int32_t* tokens, int32_t* counts, float* scores, void* stream) {
  return eec_ctc_beam_decode_ex(logp, n_seq, Tq, V, blank, beam_size, blank_skip_threshold, 0, workspace, tokens, counts, scores, stream);
}

int eec_ctc_beam_decode_ex(const float* logp, int n_seq, int Tq, int V, int blank, int beam_size, float blank_skip_threshold,
                           int skip_drops_frame, void* workspace, int32_t* tokens, int32_t* counts, float* scores, void* stream) {
  if (!logp || !workspace || !tokens || !counts || !scores || n_seq <= 0 || Tq <= 0) return EEC_ERR_BAD_ARG;
  if (V < 1 || V > 256 || beam_size < 1 || beam_size > eec::kCbMaxBeam || blank < 0 || blank >= V) return EEC_ERR_UNSUPPORTED;
  return (int)eec::launch_ctc_beam(logp, n_seq, Tq, V, blank, beam_size, blank_skip_threshold, skip_drops_frame != 0, (int*)workspace, tokens,
                                   counts, scores, (hipStream_t)stream);
}

}  // extern "C"
