// Probe: write a [M][N] bf16 matrix in GEMM-epilogue order: persistent workgroups, each writes 128x128 tiles
// (4 waves, 64x64 per wave, 8 rows x 128 B per store), tile order = grouped raster (group of GM tile rows).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
__global__ void k(unsigned short* out, int M, int N, int GM, int rowmajor) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int nbm = (M + 127) / 128, nbn = N / 128, ntiles = nbm * nbn;
  for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
    int m0, n0;
    if (rowmajor) { m0 = (id / nbn) * 128; n0 = (id % nbn) * 128; }
    else {
      const int per_group = GM * nbn, group = id / per_group, in_group = id - group * per_group;
      const int first_m = group * GM, rows_here = (nbm - first_m) < GM ? (nbm - first_m) : GM;
      m0 = (first_m + in_group % rows_here) * 128; n0 = (in_group / rows_here) * 128;
    }
    for (int ps = 0; ps < 8; ++ps) {
      const long row = m0 + wr * 64 + ps * 8 + (lane >> 3);
      const long col = n0 + wc * 64 + (lane & 7) * 8;
      if (row < M) { u4 v = {1u, 2u, 3u, (unsigned)id}; *(u4*)(out + row * N + col) = v; }
    }
  }
}
int main() {
  const int M = 35552, N = 5760;
  unsigned short* d; hipMalloc(&d, (size_t)M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rowmajor = 0; rowmajor < 2; ++rowmajor)
    for (int grid : {512, 2048, 12510}) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, M, N, 8, rowmajor);
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, M, N, 8, rowmajor);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
      printf("rowmajor=%d grid=%d: %.3f ms  %.2f TB/s\n", rowmajor, grid, ms, (double)M * N * 2 / ms / 1e9);
    }
  return 0;
}
