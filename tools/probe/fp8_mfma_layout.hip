// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) inputs on gfx950.
// One wave computes C[32][32] = A[32][64] * B[64][32]; the host tries the candidate lane/byte -> (row, k) maps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned char* a /*[64 lanes][32 B]*/, const unsigned char* b, float* c /*[64][16]*/) {
  const int l = threadIdx.x;
  v8i av, bv;
  memcpy(&av, a + l * 32, 32);
  memcpy(&bv, b + l * 32, 32);
  v16f acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 127, 0, 127);
  for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
}
// e4m3 (fn) decode
static float f8(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}
int main() {
  static unsigned char A[32][64], B[64][32];
  srand(1);
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 64; ++kk) A[i][kk] = (rand() % 0x50) | ((rand() & 1) << 7);   // |x| <= 2^3-ish, no NaN
  for (int kk = 0; kk < 64; ++kk) for (int j = 0; j < 32; ++j) B[kk][j] = (rand() % 0x50) | ((rand() & 1) << 7);
  static float ref[32][32];
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int kk = 0; kk < 64; ++kk) s += (double)f8(A[i][kk]) * f8(B[kk][j]); ref[i][j] = (float)s; }
  unsigned char *da, *db; float* dc;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 16 * 4);
  for (int la = 0; la < 2; ++la) for (int lb = 0; lb < 2; ++lb) {
    unsigned char ha[64][32], hb[64][32];
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
      const int k0 = 32 * (l / 32) + j;                                  // layout 0: 32 consecutive k per lane half
      const int k1 = 16 * (l / 32) + (j % 16) + 32 * (j / 16);           // layout 1: two groups of 16 consecutive k
      ha[l][j] = A[l % 32][la ? k1 : k0];
      hb[l][j] = B[lb ? k1 : k0][l % 32];
    }
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc);
    float hc[64][16]; hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    // accumulator layout of the 32x32 MFMAs: acc[i] of lane l = C[row = 8*(i/4) + 4*(l/32) + i%4][col = l%32]
    double worst = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
      const int row = 8 * (i / 4) + 4 * (l / 32) + i % 4, col = l % 32;
      worst = fmax(worst, fabs(hc[l][i] - ref[row][col]));
    }
    printf("A layout %d, B layout %d: max |C - ref| = %g %s\n", la, lb, worst, worst < 1e-3 ? "<== match" : "");
  }
  return 0;
}
