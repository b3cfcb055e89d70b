#include <stdio.h>
#include "eec_device.h"
using namespace eec;
__global__ void k(float* out) {
  const int lane = threadIdx.x, hh = lane >> 5, r32 = lane & 31;
  f32x16 a;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)(((i & 3) + 8 * (i >> 2) + 4 * hh) * 100 + r32);
  acc_std_to_q(a);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[i * 64 + lane] = a[i];
  acc_q_to_std(a);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[(16 + i) * 64 + lane] = a[i];
}
int main() {
  float* o; hipMallocManaged(&o, 32 * 64 * 4);
  k<<<1, 64>>>(o); hipDeviceSynchronize();
  int bad_q = 0, bad_rt = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int g = lane >> 4, c = lane & 15, hh = lane >> 5, r32 = lane & 31;
    for (int ra = 0; ra < 2; ++ra) for (int cb = 0; cb < 2; ++cb) for (int i = 0; i < 4; ++i) {
      const float want = (float)((16 * ra + 4 * g + i) * 100 + 16 * cb + c);
      const float got = o[(4 * (2 * ra + cb) + i) * 64 + lane];
      if (want != got) { if (bad_q < 8) printf("q: lane %d ra %d cb %d i %d want %.0f got %.0f\n", lane, ra, cb, i, want, got); ++bad_q; }
    }
    for (int i = 0; i < 16; ++i) {
      const float want = (float)(((i & 3) + 8 * (i >> 2) + 4 * hh) * 100 + r32);
      if (o[(16 + i) * 64 + lane] != want) ++bad_rt;
    }
  }
  printf("quadrant layout mismatches %d, round trip mismatches %d\n", bad_q, bad_rt);
  return 0;
}
