// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (bf8 = e5m2 operands) on gfx950: which k does byte i of lane l hold,
// and how do the per-lane E8M0 scales apply?  Exactly representable data, checked on the host.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k(const i8v* a, const i8v* b, const int* sa, const int* sb, float* out) {
  const int l = threadIdx.x;
  f16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 1 /*A bf8*/, 1 /*B bf8*/, 0, sa[l], 0, sb[l]);
  for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}

static float e5m2(unsigned char v) {  // decode
  int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
  float r = e == 0 ? ldexpf(m / 4.0f, -14) : ldexpf(1.0f + m / 4.0f, e - 15);
  return s ? -r : r;
}
static unsigned char enc(float x) {  // exact values only: sign, power of two times {1, 1.25, 1.5, 1.75}
  for (int v = 0; v < 256; ++v) if (e5m2((unsigned char)v) == x) return (unsigned char)v;
  printf("not representable %f\n", x); exit(1);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  // A[row][k], B[k][col] with small exactly representable values
  static float A[32][64], B[64][32];
  for (int r = 0; r < 32; ++r) for (int kk = 0; kk < 64; ++kk) A[r][kk] = ((r * 7 + kk * 3) % 5 - 2) * 0.5f;      // {-1,-.5,0,.5,1}
  for (int kk = 0; kk < 64; ++kk) for (int c = 0; c < 32; ++c) B[kk][c] = ((kk * 5 + c * 11) % 7 - 3) * 0.25f;  // multiples of .25
  // hypothesis H1: lane l holds row/col l%32, k = 32*(l/32) + byte index
  unsigned char ha[64][32], hb[64][32];
  int hsa[64], hsb[64];
  for (int l = 0; l < 64; ++l) {
    for (int i = 0; i < 32; ++i) { ha[l][i] = enc(A[l % 32][32 * (l / 32) + i]); hb[l][i] = enc(B[32 * (l / 32) + i][l % 32]); }
    hsa[l] = 127; hsb[l] = 127;
    if (mode >= 1) { hsa[l] = 127 + (l % 32) % 3; hsb[l] = 127 - ((l % 32) % 2); }
    if (mode >= 2 && l >= 32) { hsa[l] = 127 + 3; }
  }
  i8v *da, *db; int *dsa, *dsb; float* dout;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dout, 64 * 16 * 4));
  CK(hipMemcpy(da, ha, 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 64 * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
  static float out[64 * 16];
  CK(hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost));
  for (int hyp = 0; hyp < 3; ++hyp) {
    // hyp 0: scale block = lane half (k-block l/32), scale taken from the lane itself
    // hyp 1: scale block = byte index / 16 inside the lane's 32 bytes, scale of block b taken from lane (r + 32 b)
    // hyp 2: scale block = byte index / 16, but scale taken from the lane itself (lane half h supplies its own for both)
    int bad = 0; double maxerr = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
      const int c = l % 32, r = (i % 4) + 8 * (i / 4) + 4 * (l / 32);
      double ref = 0;
      for (int h = 0; h < 2; ++h) for (int bi = 0; bi < 32; ++bi) {
        const int kk = 32 * h + bi;  // how the data was loaded (lane half h, byte bi)
        int blk = hyp == 0 ? h : bi / 16;
        int src_half = hyp == 1 ? blk : h;
        double sA = 1.0, sB = 1.0;
        if (mode >= 1) { sA = ldexp(1.0, r % 3); sB = ldexp(1.0, -(c % 2)); }
        if (mode >= 2 && src_half == 1) sA = 8.0;
        ref += (double)A[r][kk] * B[kk][c] * sA * sB;
      }
      const double err = fabs(out[l * 16 + i] - ref);
      if (err > 1e-4) ++bad;
      if (err > maxerr) maxerr = err;
    }
    printf("mode %d scale hypothesis %d: %s (mismatches %d, max err %.3g)\n", mode, hyp, bad ? "rejected" : "CONFIRMED", bad, maxerr);
  }
  return 0;
}
