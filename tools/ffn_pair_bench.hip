// Stand-alone check + timing of the pair-split feed-forward stage (csrc/ffn_pair.hip), before it is wired into the plan:
//   * numerics: part_out[0] + part_out[1] against an fp64 evaluation of W2 . silu(W1 . LN(x') + b1) on sampled rows, with and
//     without the partial-sum prologue (x' = LN_fin(x + 0.5 (P0 + P1 + b2))), a ragged M, the stored rows;
//   * time: M = 16384 rows, F = 2048, rotating through several weight sets (as consecutive layers do), HIP events.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I early_exit_transformer_amd/csrc -I tools tools/ffn_pair_bench.hip -o tools/ffn_pair_bench [-DPAIR_STAMPS]
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "ffn_pair_kernel.hip"

namespace eec {
hipError_t ensure_max_lds(const void* kernel, int bytes) {
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace eec

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned long long rng_state = 0x9e3779b97f4a7c15ull;
static float urand() {  // U(-1, 1)
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((rng_state >> 40) & 0xffffff) / 8388608.0f - 1.0f;
}
template <typename T>
static T* dev(const std::vector<T>& h) {
  T* d;
  CK(hipMalloc(&d, h.size() * sizeof(T)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

// float index of (row, feature n) in plane `half` of a partial-sum buffer for M rows (the accumulator lane order of ffn_pair.hip)
static size_t pidx(int M, int half, int row, int n) {
  const size_t stride = eec::pair_part_stride(M);  // float4 per plane
  const int tile = row / 128, rt = (row % 128) / 32, r = row % 32, g = n / 8, h = (n % 8) / 4, j = n % 4;
  return (half * stride + ((size_t)(tile * 4 + rt) * 32 + g) * 64 + h * 32 + r) * 4 + j;
}

int main(int argc, char** argv) {
  const int F = 2048, D = 256;
  const int Mbig = argc > 1 ? atoi(argv[1]) : 16384;
  const float kL2e = 1.4426950408889634f;
  std::vector<float> w1((size_t)F * D), w2((size_t)D * F), b1(F), b2(D), lng(D), lnb(D), fing(D), finb(D);
  const float a1 = sqrtf(6.0f / (F + D));
  for (auto& v : w1) v = urand() * a1;
  for (auto& v : w2) v = urand() * a1;
  for (auto& v : b1) v = urand() * 0.06f;
  for (auto& v : b2) v = urand() * 0.02f;
  for (int i = 0; i < D; ++i) lng[i] = 1.0f + 0.1f * urand(), lnb[i] = 0.1f * urand(), fing[i] = 1.0f + 0.1f * urand(), finb[i] = 0.1f * urand();
  std::vector<float> b1s(F);
  for (int i = 0; i < F; ++i) b1s[i] = b1[i] * kL2e;
  float *dw1 = dev(w1), *dw2 = dev(w2), *db1 = dev(b1s), *db2 = dev(b2), *dlng = dev(lng), *dlnb = dev(lnb), *dfg = dev(fing), *dfb = dev(finb);
  const int NSET = 6;  // weight sets the timing loop rotates through (same values, different addresses)
  uint4* dwk[NSET];
  for (int i = 0; i < NSET; ++i) {
    CK(hipMalloc(&dwk[i], (size_t)(F / 32) * eec::kPairChunkU4 * 16));
    CK(eec::launch_pack_ffn_pair(dw1, dw2, F, dwk[i], kL2e, 1.0f / kL2e, 0));
  }
  CK(hipDeviceSynchronize());

  auto reference_row = [&](const std::vector<float>& x, const std::vector<float>* part, int M, int row, bool fin, std::vector<double>& xrow,
                           std::vector<double>& y) {
    std::vector<double> v(D);
    for (int k = 0; k < D; ++k) {
      v[k] = x[(size_t)row * D + k];
      if (part) v[k] += 0.5 * ((double)(*part)[pidx(M, 0, row, k)] + (double)(*part)[pidx(M, 1, row, k)]);
    }
    auto ln = [&](std::vector<double>& t, const std::vector<float>& g, const std::vector<float>& b) {
      double m = 0, q = 0;
      for (double e : t) m += e;
      m /= D;
      for (double e : t) q += (e - m) * (e - m);
      const double rs = 1.0 / sqrt(q / D + 1e-5);
      for (int k = 0; k < D; ++k) t[k] = (t[k] - m) * rs * g[k] + b[k];
    };
    if (fin) ln(v, fing, finb);
    xrow = v;
    ln(v, lng, lnb);
    std::vector<double> hbuf(F);
    for (int f = 0; f < F; ++f) {
      double s = b1[f];
      for (int k = 0; k < D; ++k) s += (double)w1[(size_t)f * D + k] * v[k];
      hbuf[f] = s / (1.0 + exp(-s));
    }
    y.assign(D, 0.0);
    for (int n = 0; n < D; ++n) {
      double s = 0;
      for (int f = 0; f < F; ++f) s += (double)w2[(size_t)n * F + f] * hbuf[f];
      y[n] = s + b2[n];
    }
  };

  int fails = 0;
  for (int variant = 0; variant < 3; ++variant) {
    const int M = variant == 2 ? 301 : 1024;  // ragged tail: 301 rows = 2 full tiles + 45 rows
    const bool part = variant >= 1, fin = variant >= 1;
    std::vector<float> x((size_t)M * D), pin(2 * eec::pair_part_stride(M) * 4);
    for (auto& v : x) v = urand() * 2.0f;
    for (auto& v : pin) v = urand();
    float *dx = dev(x), *dpin = dev(pin), *dxo, *dtap, *dpo;
    CK(hipMalloc(&dxo, x.size() * 4));
    CK(hipMalloc(&dtap, x.size() * 4));
    CK(hipMalloc(&dpo, pin.size() * 4));
    CK(hipMemset(dpo, 0xff, pin.size() * 4));
    eec::PairArgs a{dx, part ? dxo : nullptr, part ? dtap : nullptr, part ? dpin : nullptr, db2, 0.5f, fin ? dfg : nullptr, fin ? dfb : nullptr,
                    dlng, dlnb, dwk[0], db1, dpo, M, F};
    CK(eec::launch_ffn_pair(a, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> po(pin.size()), xo(x.size()), tp(x.size());
    CK(hipMemcpy(po.data(), dpo, po.size() * 4, hipMemcpyDeviceToHost));
    if (part) {
      CK(hipMemcpy(xo.data(), dxo, xo.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(tp.data(), dtap, tp.size() * 4, hipMemcpyDeviceToHost));
    }
    double worst = 0, worst_x = 0, scale = 0;
    const int rows[] = {0, 1, 31, 32, 63, 64, 127, 128, 129, 255, 256, 300, M - 1, M / 2};
    for (int row : rows) {
      if (row >= M) continue;
      std::vector<double> xr, y;
      reference_row(x, part ? &pin : nullptr, M, row, fin, xr, y);
      for (int n = 0; n < D; ++n) {
        const double got = (double)po[pidx(M, 0, row, n)] + (double)po[pidx(M, 1, row, n)];
        worst = fmax(worst, fabs(got - y[n]));
        scale = fmax(scale, fabs(y[n]));
        if (part) {
          worst_x = fmax(worst_x, fabs(xo[(size_t)row * D + n] - xr[n]));
          worst_x = fmax(worst_x, fabs(tp[(size_t)row * D + n] - xr[n]));
        }
      }
    }
    const bool okv = worst < 2e-5 * fmax(scale, 1.0) && worst_x < 2e-5;
    printf("variant %d (M=%d part=%d fin=%d): max |y - ref| = %.3e (scale %.3f), stored rows %.3e  %s\n", variant, M, part, fin, worst, scale,
           worst_x, okv ? "OK" : "FAIL");
    fails += !okv;
    CK(hipFree(dx)); CK(hipFree(dpin)); CK(hipFree(dxo)); CK(hipFree(dtap)); CK(hipFree(dpo));
  }

  // ---- timing ----
  {
    const int M = Mbig;
    std::vector<float> x((size_t)M * D);
    for (auto& v : x) v = urand() * 2.0f;
    float *dx = dev(x), *dxo, *dp[2];
    CK(hipMalloc(&dxo, x.size() * 4));
    for (int i = 0; i < 2; ++i) {
      CK(hipMalloc(&dp[i], 2 * eec::pair_part_stride(M) * 16));
      CK(hipMemset(dp[i], 0, 2 * eec::pair_part_stride(M) * 16));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {  // 0: plain stage; 1: with the partial-sum prologue, final LayerNorm and row stores
      auto run = [&](int i) {
        eec::PairArgs a{dx, mode ? dxo : nullptr, nullptr, mode ? dp[i & 1] : nullptr, db2, 0.5f, mode ? dfg : nullptr, mode ? dfb : nullptr,
                        dlng, dlnb, dwk[i % NSET], db1, dp[(i + 1) & 1], M, F};
        CK(eec::launch_ffn_pair(a, 0));
      };
      for (int i = 0; i < 6; ++i) run(i);
      CK(hipDeviceSynchronize());
      const int N = 48;
      float best = 1e9f, total = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < N; ++i) run(i);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / N);
        total += ms / N;
      }
      const double flop = 2.0 * M * 2.0 * D * F;
#ifdef PAIR_STAMPS
      {
        unsigned long long st[16 * 8];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(eec::g_pair_stamps), sizeof(st)));
        for (int b = 0; b < 16; b += 5) {
          const unsigned long long* q = st + b * 8;
          const double clk = (double)(q[6] - q[0]) / ((double)(q[7] - q[1]) * 10.0);  // cycles per ns (s_memrealtime ticks at 100 MHz)
          printf("  block %3d: prologue %6llu  main loop %6llu  epilogue %6llu cycles; in-kernel clock %.2f GHz; kernel %.1f us\n", b * 16, q[2] - q[0],
                 q[4] - q[2], q[6] - q[4], clk, (double)(q[7] - q[1]) * 0.01);
        }
      }
#endif
      printf("timing M=%d mode %d: %.2f us per launch (best of 4 x %d back-to-back; mean %.2f) = %.1f TFLOP/s algorithmic, %.1f executed (x3)\n", M,
             mode, best * 1e3, N, total / 4 * 1e3, flop / (best * 1e-3) / 1e12, 3 * flop / (best * 1e-3) / 1e12);
    }
  }
  return fails ? 1 : 0;
}
