// Probe (second hypothesis: data in layout 1, lanes 0-31 scale block 0, lanes 32-63 block 1): scale-operand semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8).  Hypothesis: lane l supplies the E8M0
// scale of row l % 32 for the 32 k's of half l / 32 (operand layout 0 of fp8_mfma_layout.hip); opsel picks the byte.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int OPA, int OPB>
__global__ void k(const unsigned char* a, const unsigned char* b, const unsigned* sa, const unsigned* sb, float* c) {
  const int l = threadIdx.x;
  v8i av, bv;
  memcpy(&av, a + l * 32, 32);
  memcpy(&bv, b + l * 32, 32);
  v16f acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, OPA, sa[l], OPB, sb[l]);
  for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
}
static float f8(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}
int main() {
  static unsigned char A[32][64], B[64][32];
  srand(2);
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 64; ++kk) A[i][kk] = (rand() % 0x48) | ((rand() & 1) << 7);
  for (int kk = 0; kk < 64; ++kk) for (int j = 0; j < 32; ++j) B[kk][j] = (rand() % 0x48) | ((rand() & 1) << 7);
  unsigned hsa[64], hsb[64];
  for (int l = 0; l < 64; ++l) {          // four different bytes per lane, exponents 120..134
    hsa[l] = 0; hsb[l] = 0;
    for (int by = 0; by < 4; ++by) { hsa[l] |= (unsigned)(120 + rand() % 15) << (8 * by); hsb[l] |= (unsigned)(120 + rand() % 15) << (8 * by); }
  }
  unsigned char ha[64][32], hb[64][32];
  // layout 1: byte j of lane-half h holds k = 32 * (j / 16) + 16 * h + j % 16  (block j / 16 of the 64-deep step)
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { const int kk = 32 * (j / 16) + 16 * (l / 32) + j % 16; ha[l][j] = A[l % 32][kk]; hb[l][j] = B[kk][l % 32]; }
  unsigned char *da, *db; unsigned *dsa, *dsb; float* dc;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 4096);
  hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
  for (int op = 0; op < 4; ++op) {
    if (op == 0) k<0, 0><<<1, 64>>>(da, db, dsa, dsb, dc);
    if (op == 1) k<1, 1><<<1, 64>>>(da, db, dsa, dsb, dc);
    if (op == 2) k<2, 2><<<1, 64>>>(da, db, dsa, dsb, dc);
    if (op == 3) k<3, 3><<<1, 64>>>(da, db, dsa, dsb, dc);
    float hc[64][16]; hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    for (int hyp = 0; hyp < 4; ++hyp) {   // which byte does opsel = op use?
      double worst = 0;
      for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
        const int row = 8 * (i / 4) + 4 * (l / 32) + i % 4, col = l % 32;
        double s = 0, mag = 0;
        for (int h = 0; h < 2; ++h) {
          const int ea = (hsa[row + 32 * h] >> (8 * hyp)) & 0xff, eb = (hsb[col + 32 * h] >> (8 * hyp)) & 0xff;
          double part = 0;
          for (int kk = 0; kk < 32; ++kk) part += (double)f8(A[row][32 * h + kk]) * f8(B[32 * h + kk][col]);
          double pm = 0;
          for (int kk = 0; kk < 32; ++kk) pm += fabs((double)f8(A[row][32 * h + kk]) * f8(B[32 * h + kk][col]));
          s += part * ldexp(1.0, ea - 127) * ldexp(1.0, eb - 127);
          mag += pm * ldexp(1.0, ea - 127) * ldexp(1.0, eb - 127);
        }
        worst = fmax(worst, fabs(hc[l][i] - s) / mag);
      }
      printf("opsel %d, hypothesis byte %d: max rel err %g %s\n", op, hyp, worst, worst < 1e-5 ? "<== match" : "");
    }
  }
  return 0;
}
