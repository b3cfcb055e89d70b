// Mel front end (reference util/data_loader.py:7-18): power spectrogram of 1024-point frames
// (hann window of 320 samples, hop 160, centred with reflect padding) -> 80 triangular mel filters, NO log.
//
//   P[t][k]   = | sum_j w[j] x[160 t - 160 + j] e^{-2 pi i k j / 1024} |^2     j = 0 .. 319, k = 0 .. 512
//   mel[m][t] = sum_k fb[k][m] P[t][k]
// (the window sits at samples 352 .. 671 of the 1024-point frame; the phase factor of that offset has unit modulus and
// drops out of the power).  The transform runs as an exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fma
// chain, the precision the reference's fp32 FFT works in) of the windowed frames [32 frames x 320] against the DFT basis
// [320 x (512 cos | 512 sin)] -- one 512-thread workgroup per 32 frames, wave w owns bins [64 w, 64 w + 64) as two cos and
// two sin 32x32 tiles, so re and im of a bin meet in the same lane and the power never leaves the registers; the Nyquist
// bin (sin = 0, cos = +-1) is an alternating sum on the vector ALU.  The power tile goes to LDS and the mel filters (two
// slopes per bin: ~1 k non-zeros) are applied from a compact per-filter table on the vector ALU.
// Bound: fp32 MFMA (21.6 GMAC per 64 x 10.3 s batch at ~150 TF/s dense fp32 peak); HBM traffic is the waveform (42 MB)
// and the mel output (21 MB).
#include <math.h>

#include <string>
#include <vector>

#include "../../include/eec.h"
#include "eec_kernels.h"

namespace eec {

constexpr int kFeThreads = 512;
constexpr int kFeFrames = 32;                 // frames per workgroup
constexpr int kFeQuads = 320 / 2 / 4;         // a k-step of the fp32 MFMA covers 2 samples; a lane's float4 covers 4 k-steps
constexpr int kFeALd = 320 + 4;               // floats per staged frame row: [160 even samples | 160 odd samples | pad]
constexpr int kFePLd = 516 + 1;               // floats per power row (513 bins)
constexpr int kFeLds = kFeFrames * (kFeALd + kFePLd) * 4;  // 107 648 B

struct FrontendArgs {
  const float* wave;       // [B][Lmax]
  const long long* length; // [B] valid samples (nullptr: Lmax for all)
  int B, Lmax, Tmax, n_mels, hop, win;
  const float* window;     // [320]
  const float4* basis;     // [16 bin tiles][2: cos, -sin][40 quads of k-steps][64 lanes] float4
  const int* fb_range;     // [n_mels][3]: first bin, last bin + 1, offset into fb_w
  const float* fb_w;       // filter weights, concatenated per mel bin over its support
  float* mel;              // [B][n_mels][Tmax]
};

typedef float f32x16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(kFeThreads, 2) void mel_frontend_kernel(FrontendArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_a = (float*)smem;                       // [32][324]
  float* lds_p = lds_a + kFeFrames * kFeALd;         // [32][517]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * kFeFrames;
  const int L = a.length ? (int)min((long long)a.Lmax, max(0ll, a.length[b])) : a.Lmax;
  const int T = L > 0 ? 1 + L / a.hop : 0;           // frames of this utterance (torch.stft, center=True)
  const float* x = a.wave + (size_t)b * a.Lmax;
  // ---- stage the windowed frames: sample j of frame f -> row f, slot (j & 1) * 160 + (j >> 1) ----
  for (int p = threadIdx.x; p < kFeFrames * 320; p += kFeThreads) {
    const int f = p / 320, j = p - f * 320;
    float v = 0.f;
    if (t0 + f < T) {
      int idx = (t0 + f) * a.hop - a.hop + j;        // = 160 t - 512 + 352 + j
      if (idx < 0) idx = -idx;                        // reflect padding (no edge repeat)
      if (idx >= L) idx = 2 * (L - 1) - idx;
      idx = min(max(idx, 0), L - 1);                  // utterances shorter than the pad: clamp (torch raises there)
      v = x[idx] * a.window[j];
    }
    lds_a[f * kFeALd + (j & 1) * 160 + (j >> 1)] = v;
  }
  __syncthreads();
  // ---- DFT: wave w owns bin tiles 2w, 2w+1 (cos and sin parts): 4 accumulators ----
  f32x16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][p][e] = 0.f;
  const float* a_lane = lds_a + r * kFeALd + hh * 160;
  const float4* b_lane = a.basis + (size_t)(2 * w) * 2 * kFeQuads * 64 + lane;
  float4 bq[2][2], bn[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 2; ++p) bq[i][p] = b_lane[((size_t)(i * 2 + p) * kFeQuads) * 64];
  for (int s4 = 0; s4 < kFeQuads; ++s4) {
    const float4 av = *(const float4*)(a_lane + 4 * s4);
    if (s4 + 1 < kFeQuads) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) bn[i][p] = b_lane[((size_t)(i * 2 + p) * kFeQuads + s4 + 1) * 64];
    }
    const float avq[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const float bv = q == 0 ? bq[i][p].x : q == 1 ? bq[i][p].y : q == 2 ? bq[i][p].z : bq[i][p].w;
          acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(avq[q], bv, acc[i][p], 0, 0, 0);
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) bq[i][p] = bn[i][p];
  }
  // power of this lane's bins: col = lane & 31 -> bin 32 (2w + i) + r; row (frame) = acc_row(e)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float re = acc[i][0][e], im = acc[i][1][e];
      lds_p[acc_row(e, lane) * kFePLd + 32 * (2 * w + i) + r] = re * re + im * im;
    }
  if (w == 0) {  // Nyquist bin 512: sum_j (-1)^j a_j; lane = (frame, parity)
    float s = 0.f;
    for (int q = 0; q < 160; q += 4) {
      const float4 v = *(const float4*)(a_lane + q);
      s += (v.x + v.y) + (v.z + v.w);
    }
    const float o = __shfl_xor(s, 32, 64);
    const float d = hh == 0 ? s - o : o - s;
    if (hh == 0) lds_p[r * kFePLd + 512] = d * d;
  }
  __syncthreads();
  // ---- mel filters: thread = (frame, group of mel bins) ----
  {
    const int f = threadIdx.x & 31, g = threadIdx.x >> 5;  // 16 groups
    const int t = t0 + f;
    for (int m = g; m < a.n_mels; m += 16) {
      const int k0 = a.fb_range[3 * m], k1 = a.fb_range[3 * m + 1];
      const float* wv = a.fb_w + a.fb_range[3 * m + 2];
      float s = 0.f;
      for (int k = k0; k < k1; ++k) s = fmaf(wv[k - k0], lds_p[f * kFePLd + k], s);
      if (t < a.Tmax) a.mel[((size_t)b * a.n_mels + m) * a.Tmax + t] = t < T ? s : 0.f;
    }
  }
}

}  // namespace eec

using namespace eec;

struct eec_frontend {
  int device = -1;
  int sample_rate, n_fft, win, hop, n_mels;
  float* window = nullptr;
  float4* basis = nullptr;
  int* fb_range = nullptr;
  float* fb_w = nullptr;
};

namespace {
thread_local std::string g_fe_err;
}

extern "C" {

const char* eec_frontend_last_error(void) { return g_fe_err.c_str(); }

int eec_frontend_create(int sample_rate, int n_fft, int win_length, int hop_length, int n_mels, eec_frontend** out) {
  if (!out) return EEC_ERR_BAD_ARG;
  if (n_fft != 1024 || win_length != 320 || hop_length != 160 || n_mels <= 0 || n_mels > 256 || sample_rate <= 0) {
    g_fe_err = "this build serves the reference's front end: 1024-point frames, window 320, hop 160, <= 256 mel bins";
    return EEC_ERR_UNSUPPORTED;
  }
  eec_frontend* fe = new eec_frontend();
  fe->sample_rate = sample_rate, fe->n_fft = n_fft, fe->win = win_length, fe->hop = hop_length, fe->n_mels = n_mels;
  if (hipGetDevice(&fe->device) != hipSuccess) {
    delete fe;
    g_fe_err = "hipGetDevice failed";
    return EEC_ERR_BAD_ARG;
  }
  const double kPi = 3.14159265358979323846;
  std::vector<float> window(win_length);
  for (int j = 0; j < win_length; ++j) window[j] = (float)(0.5 - 0.5 * cos(2.0 * kPi * j / win_length));  // periodic hann
  // basis[bt][p][s4][lane][q] = (p ? -sin : cos)(2 pi k j / 1024), k = 32 bt + (lane & 31), j = 2 (4 s4 + q) + (lane >> 5)
  std::vector<float> basis((size_t)16 * 2 * kFeQuads * 64 * 4);
  for (int bt = 0; bt < 16; ++bt)
    for (int p = 0; p < 2; ++p)
      for (int s4 = 0; s4 < kFeQuads; ++s4)
        for (int lane = 0; lane < 64; ++lane)
          for (int q = 0; q < 4; ++q) {
            const int k = 32 * bt + (lane & 31), j = 2 * (4 * s4 + q) + (lane >> 5);
            const double ang = 2.0 * kPi * (double)((k * j) % n_fft) / n_fft;
            basis[((((size_t)bt * 2 + p) * kFeQuads + s4) * 64 + lane) * 4 + q] = (float)(p ? -sin(ang) : cos(ang));
          }
  // htk mel filterbank (torchaudio.functional.melscale_fbanks, norm=None), evaluated in fp32 like the reference's table
  const int n_freqs = n_fft / 2 + 1;
  std::vector<float> all_freqs(n_freqs), f_pts(n_mels + 2);
  for (int k = 0; k < n_freqs; ++k) all_freqs[k] = (float)(sample_rate / 2) * (float)k / (float)(n_freqs - 1);
  const float m_min = 0.0f, m_max = 2595.0f * log10f(1.0f + (float)(sample_rate / 2) / 700.0f);
  for (int i = 0; i < n_mels + 2; ++i) {
    const float m = m_min + (m_max - m_min) * (float)i / (float)(n_mels + 1);
    f_pts[i] = 700.0f * (powf(10.0f, m / 2595.0f) - 1.0f);
  }
  std::vector<int> range(3 * n_mels);
  std::vector<float> weights;
  for (int m = 0; m < n_mels; ++m) {
    int k0 = -1, k1 = -1;
    std::vector<float> col(n_freqs);
    for (int k = 0; k < n_freqs; ++k) {
      const float down = -(f_pts[m] - all_freqs[k]) / (f_pts[m + 1] - f_pts[m]);
      const float up = (f_pts[m + 2] - all_freqs[k]) / (f_pts[m + 2] - f_pts[m + 1]);
      const float v = fmaxf(0.0f, fminf(down, up));
      col[k] = v;
      if (v > 0.f) {
        if (k0 < 0) k0 = k;
        k1 = k + 1;
      }
    }
    if (k0 < 0) k0 = k1 = 0;
    range[3 * m] = k0, range[3 * m + 1] = k1, range[3 * m + 2] = (int)weights.size();
    for (int k = k0; k < k1; ++k) weights.push_back(col[k]);
  }
  if (weights.empty()) weights.push_back(0.f);
  auto up = [&](void** dst, const void* src, size_t bytes) {
    return hipMalloc(dst, bytes) == hipSuccess && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (!up((void**)&fe->window, window.data(), window.size() * 4) || !up((void**)&fe->basis, basis.data(), basis.size() * 4) ||
      !up((void**)&fe->fb_range, range.data(), range.size() * 4) || !up((void**)&fe->fb_w, weights.data(), weights.size() * 4)) {
    g_fe_err = "device allocation / upload of the front-end tables failed";
    return EEC_ERR_WORKSPACE;
  }
  *out = fe;
  return 0;
}

void eec_frontend_destroy(eec_frontend* fe) {
  if (!fe) return;
  (void)hipFree(fe->window), (void)hipFree(fe->basis), (void)hipFree(fe->fb_range), (void)hipFree(fe->fb_w);
  delete fe;
}

int eec_frontend_frames(int n_samples, int hop_length) { return n_samples > 0 && hop_length > 0 ? 1 + n_samples / hop_length : 0; }

int eec_frontend_forward(eec_frontend* fe, const float* wave, const int64_t* lengths_opt, int B, int Lmax, float* mel, void* stream) {
  if (!fe || !wave || !mel || B <= 0 || Lmax <= 0) {
    g_fe_err = "bad argument";
    return EEC_ERR_BAD_ARG;
  }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != fe->device) {
    g_fe_err = "the front end was created on another device";
    return EEC_ERR_BAD_ARG;
  }
  FrontendArgs a{wave, (const long long*)lengths_opt, B, Lmax, 1 + Lmax / fe->hop, fe->n_mels, fe->hop, fe->win,
                 fe->window, fe->basis, fe->fb_range, fe->fb_w, mel};
  if (hipError_t e = ensure_max_lds((const void*)mel_frontend_kernel, kFeLds); e != hipSuccess) {
    g_fe_err = hipGetErrorString(e);
    return (int)e;
  }
  hipLaunchKernelGGL(mel_frontend_kernel, dim3((a.Tmax + kFeFrames - 1) / kFeFrames, B), dim3(kFeThreads), kFeLds, (hipStream_t)stream, a);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) {
    g_fe_err = hipGetErrorString(e);
    return (int)e;
  }
  return 0;
}

}  // extern "C"
