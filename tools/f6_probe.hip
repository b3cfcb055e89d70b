// Probe: v_cvt_scalef32_pk32_bf6_f16 + v_mfma_scale_f32_32x32x64_f8f6f4 with bf6 (e3m2) operands.
// Checks (1) what the conversion's scale operand means, (2) that operands produced by the conversion instruction
// pair element-wise in the MFMA (any consistent k <-> element map then works), (3) the E8M0 scale operands.
// hipcc --offload-arch=gfx950 -O2 tools/f6_probe.hip -o tools/f6_probe && ./tools/f6_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ int kmap(int p, int h) { return 16 * (p / 8) + 8 * h + p % 8; }

__global__ void probe(const float* A, const float* B, float* C, float cvt_scale, int e8a, int e8b) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  h32 a, b;
  for (int p = 0; p < 32; ++p) {
    a[p] = (_Float16)A[r * 64 + kmap(p, h)];
    b[p] = (_Float16)B[r * 64 + kmap(p, h)];
  }
  const i32x6 a6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(a, cvt_scale);
  const i32x6 b6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(b, 1.0f);
  const i32x8 a8 = {a6[0], a6[1], a6[2], a6[3], a6[4], a6[5], 0, 0}, b8 = {b6[0], b6[1], b6[2], b6[3], b6[4], b6[5], 0, 0};
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 3 /*bf6*/, 3 /*bf6*/, 0, e8a, 0, e8b);
  // C/D layout of 32x32: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * h
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

int main() {
  static float hA[32 * 64], hB[32 * 64], hC[32 * 32];
  const float vals[] = {0.f, 0.25f, 0.5f, 1.f, 1.5f, 2.f, 3.f, -1.f, -0.5f, -2.f, 4.f, -3.f};
  for (int i = 0; i < 32 * 64; ++i) hA[i] = vals[(i * 7 + i / 64) % 12], hB[i] = vals[(i * 5 + 3 * (i / 64)) % 12];
  float *A, *B, *C;
  hipMalloc(&A, sizeof(hA)); hipMalloc(&B, sizeof(hB)); hipMalloc(&C, sizeof(hC));
  hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
  struct { float cs; int ea, eb; const char* what; } cases[] = {
      {1.f, 127, 127, "scale 1, E8M0 127/127 (expect exact A.B^T)"},
      {2.f, 127, 127, "cvt scale 2 on A (ratio tells: 0.5 = divides, 2 = multiplies)"},
      {1.f, 128, 127, "E8M0 128 on A (expect 2x)"},
      {1.f, 127, 125, "E8M0 125 on B (expect 0.25x)"}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, C, cs.cs, cs.ea, cs.eb);
    hipMemcpy(hC, C, sizeof(hC), hipMemcpyDeviceToHost);
    double num = 0, den = 0, maxd = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < 64; ++k) ref += (double)hA[i * 64 + k] * hB[j * 64 + k];
        // A is the FIRST operand: rows of the result index A's rows?  report both orientations
        num += hC[i * 32 + j] * ref; den += ref * ref;
        maxd = fmax(maxd, fabs(hC[i * 32 + j] - ref));
      }
    double num_t = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < 64; ++k) ref += (double)hA[j * 64 + k] * hB[i * 64 + k];
        num_t += hC[i * 32 + j] * ref;
      }
    printf("%-70s  C ~ %.4f x (A.B^T)   [transposed fit %.4f]  max|C - A.B^T| %.3g\n", cs.what, num / den, num_t / den, maxd);
  }
  return 0;
}
