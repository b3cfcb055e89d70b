// Stand-alone timing of the training step's fused feed-forward forward (csrc/ffn.hip, TR variants) with its ablation builds:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I early_exit_transformer_amd/csrc -DEEC_FFN_TRAIN [-DEEC_TR_ABLATE=n] [-DEEC_SIDE_VALU_NP3=n]
//         tools/ffn_train_bench.hip early_exit_transformer_amd/csrc/pack.hip -o tools/ffn_train_bench
// (-DEEC_FFN_TRAIN_BWD instead of -DEEC_FFN_TRAIN: the backward variant)
// M = 16384 rows, d_model 256, F = 2048, p = 0.1 and p = 0; six weight sets in rotation (as consecutive modules do), HIP events.
// Numerics are the training tests' business (tests/test_gpu_train.py); this prints one checksum so that ablations are visibly different.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "ffn.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned long long rng_state = 0x9e3779b97f4a7c15ull;
static float urand() {
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

int main(int argc, char** argv) {
  const int D = 256, F = 2048, M = argc > 1 ? atoi(argv[1]) : 16384, NSET = 6, np = argc > 2 ? atoi(argv[2]) : 3;
  std::vector<float> w1((size_t)F * D), w2((size_t)D * F), b1(F), b2(D), g(D), b(D), x((size_t)M * D);
  const float a1 = sqrtf(6.0f / (F + D));
  for (auto& v : w1) v = urand() * a1;
  for (auto& v : w2) v = urand() * a1;
  for (auto& v : b1) v = urand() * 0.06f;
  for (auto& v : b2) v = urand() * 0.02f;
  for (int i = 0; i < D; ++i) g[i] = 1.0f + 0.1f * urand(), b[i] = 0.1f * urand();
  for (auto& v : x) v = urand() * 2.0f;
  float *dw1 = dev(w1), *dw2 = dev(w2), *db1 = dev(b1), *db2 = dev(b2), *dg = dev(g), *dbt = dev(b), *dx = dev(x);
  uint4 *w1p[NSET], *w2p[NSET];
  for (int i = 0; i < NSET; ++i) {
    CK(hipMalloc(&w1p[i], (size_t)F * D * 4));
    CK(hipMalloc(&w2p[i], (size_t)F * D * 4));
#ifdef EEC_FFN_TRAIN_BWD  // backward: W2^T as [F][D], W1^T as [D][F], bf16 fragments
    CK(eec::launch_pack_frags_bf16(dw2, F, D, 1, F, w1p[i], 0));
    CK(eec::launch_pack_frags_bf16(dw1, D, F, 1, D, w2p[i], 0));
#else
    CK(eec::launch_pack_frags(dw1, F, D, w1p[i], 1.0f, 0));
    CK(eec::launch_pack_frags(dw2, D, F, w2p[i], 1.0f, 0));
#endif
  }
  float *y, *ln, *mean, *rstd, *pre[2], *act[2];
  CK(hipMalloc(&y, (size_t)M * D * 4));
  CK(hipMalloc(&ln, (size_t)M * D * 4));
  CK(hipMalloc(&mean, (size_t)M * 4));
  CK(hipMalloc(&rstd, (size_t)M * 4));
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&pre[i], (size_t)M * F * 4));
    CK(hipMalloc(&act[i], (size_t)M * F * 4));
    CK(hipMemset(pre[i], 0, (size_t)M * F * 4));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    const float p = mode == 0 ? 0.1f : 0.0f;
    auto run = [&](int i) {
      eec::ChainArgs a{};
      a.x = dx, a.M = M, a.F = F, a.nstage = 1, a.D = D;
      a.st[0] = eec::FfnStage{dg, dbt, w1p[i % NSET], db1, w2p[i % NSET], db2, nullptr, nullptr, nullptr, nullptr, 0.5f, nullptr};
      a.tr = eec::ChainTrain{y, ln, mean, rstd, pre[i & 1], act[i & 1], p, 1234ull, (unsigned)(2 * i + 1), (unsigned)(2 * i + 2)};
#ifdef EEC_FFN_TRAIN_BWD  // x = dh; pre[] read (whatever the forward mode left there: any finite values do for timing), act[] <- d(pre)
      a.st[0].ln_g = a.st[0].ln_b = a.st[0].b1 = a.st[0].b2 = nullptr, a.st[0].res_scale = 1.0f;
      CK(eec::launch_ffn_train_bwd(a, np, 0));
#else
      CK(eec::launch_ffn_train_fwd(a, np, 0));
#endif
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
    std::vector<float> yo(1024);
    CK(hipMemcpy(yo.data(), y + (size_t)(M / 2) * D, 4096, hipMemcpyDeviceToHost));
    double cs = 0;
    for (float v : yo) cs += v;
    printf("M=%d np=%d p=%.1f ablate=%d side_valu=%d: %.2f us per launch (best of 4 x %d; mean %.2f)  checksum %.6f\n", M, np, p, EEC_TR_ABLATE, EEC_SIDE_VALU_NP3,
           best * 1e3, N, total / 4 * 1e3, cs);
  }
  return 0;
}
