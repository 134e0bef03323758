// Probe: operand and scale-operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3).
// Data hypotheses for lane l (row / column l & 15, lane group g = l >> 4), byte j of its 32:
//   D0: k = 32 g + j                          (one contiguous 32-block per lane group)
//   D1: k = 64 (j / 16) + 16 g + j % 16       (the 32x32x64 pattern widened to four lane groups)
// Scale hypothesis: lane (r, g) supplies, in byte `opsel` of its scale register, the E8M0 scale of row r for K-block b:
//   S0: b = g      S1: b = 2 (g & 1) + (g >> 1)
// C layout (known): lane l holds C[4 (l >> 4) + i][l & 15], i = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int OP>
__global__ void k(const unsigned char* a, const unsigned char* b, const unsigned* sa, const unsigned* sb, float* c) {
  const int l = threadIdx.x;
  v8i av, bv;
  memcpy(&av, a + l * 32, 32);
  memcpy(&bv, b + l * 32, 32);
  v4f acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, OP, sa[l], OP, sb[l]);
  for (int i = 0; i < 4; ++i) c[l * 4 + i] = acc[i];
}
static float f8(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}
static int kmap(int D, int g, int j) { return D == 0 ? 32 * g + j : 64 * (j / 16) + 16 * g + j % 16; }
static int bmap(int S, int g) { return S == 0 ? g : 2 * (g & 1) + (g >> 1); }
int main() {
  static unsigned char A[16][128], B[128][16];
  static unsigned char SA[16][4], SB[16][4];      // E8M0 scale per row / column and 32-block
  srand(3);
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) A[i][kk] = (rand() % 0x48) | ((rand() & 1) << 7);
  for (int kk = 0; kk < 128; ++kk) for (int j = 0; j < 16; ++j) B[kk][j] = (rand() % 0x48) | ((rand() & 1) << 7);
  for (int i = 0; i < 16; ++i) for (int b = 0; b < 4; ++b) { SA[i][b] = 120 + rand() % 15; SB[i][b] = 120 + rand() % 15; }
  double ref[16][16], mag[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0, m = 0;
    for (int b = 0; b < 4; ++b) {
      double part = 0, pm = 0;
      for (int kk = 0; kk < 32; ++kk) { const double t = (double)f8(A[i][32 * b + kk]) * f8(B[32 * b + kk][j]); part += t; pm += fabs(t); }
      const double sc = ldexp(1.0, SA[i][b] - 127) * ldexp(1.0, SB[j][b] - 127);
      s += part * sc; m += pm * sc;
    }
    ref[i][j] = s; mag[i][j] = m;
  }
  unsigned char *da, *db; unsigned *dsa, *dsb; float* dc;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 1024);
  for (int D = 0; D < 2; ++D) for (int S = 0; S < 2; ++S) for (int op = 0; op < 4; ++op) {
    unsigned char ha[64][32], hb[64][32]; unsigned hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) {
      const int r = l & 15, g = l >> 4;
      for (int j = 0; j < 32; ++j) { const int kk = kmap(D, g, j); ha[l][j] = A[r][kk]; hb[l][j] = B[kk][r]; }
      hsa[l] = 0x7b7c7d7eu; hsb[l] = 0x7e7d7c7bu;                 // decoys in the other bytes
      hsa[l] = (hsa[l] & ~(0xffu << (8 * op))) | ((unsigned)SA[r][bmap(S, g)] << (8 * op));
      hsb[l] = (hsb[l] & ~(0xffu << (8 * op))) | ((unsigned)SB[r][bmap(S, g)] << (8 * op));
    }
    hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    if (op == 0) k<0><<<1, 64>>>(da, db, dsa, dsb, dc);
    if (op == 1) k<1><<<1, 64>>>(da, db, dsa, dsb, dc);
    if (op == 2) k<2><<<1, 64>>>(da, db, dsa, dsb, dc);
    if (op == 3) k<3><<<1, 64>>>(da, db, dsa, dsb, dc);
    float hc[64][4]; hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      const int row = 4 * (l >> 4) + i, col = l & 15;
      worst = fmax(worst, fabs(hc[l][i] - ref[row][col]) / mag[row][col]);
    }
    printf("data D%d, scale S%d, opsel %d: max rel err %g %s\n", D, S, op, worst, worst < 1e-5 ? "<== match" : "");
  }
  return 0;
}
