// Probe: HBM write bandwidth of a [M][N] bf16 matrix when each wave-store covers R rows x (1024/R) bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int R>
__global__ void k(unsigned short* out, int M, int N) {
  // tile of 128 rows x 64 cols per wave-iteration, like the GEMM epilogue; R rows per store instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long ntile_n = N / (64 * (8 / R) );
  const int lanes_per_row = 64 / R;                 // lanes covering one row segment (16 B each)
  const int seg_cols = lanes_per_row * 8;           // columns per row segment
  const long tiles_n = N / seg_cols;
  const long total = (long)(M / R) * tiles_n;       // (row group, col segment) items
  for (long it = (long)blockIdx.x * 4 + wave; it < total; it += (long)gridDim.x * 4) {
    const long rg = it / tiles_n, cs = it % tiles_n;
    const long row = rg * R + lane / lanes_per_row;
    const long col = cs * seg_cols + (lane % lanes_per_row) * 8;
    u4 v = {1u, 2u, 3u, (unsigned)it};
    *(u4*)(out + row * N + col) = v;
  }
}
int main() {
  const int M = 35552, N = 5760;
  unsigned short* d; hipMalloc(&d, (size_t)M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, M, N);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, M, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%s: %.3f ms  %.2f TB/s\n", name, ms, (double)M * N * 2 / ms / 1e9);
  };
  run(k<8>, "8 rows x 128 B per store");
  run(k<4>, "4 rows x 256 B per store");
  run(k<2>, "2 rows x 512 B per store");
  run(k<1>, "1 row x 1024 B per store");
  return 0;
}
