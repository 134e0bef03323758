// Speed-of-light probe for weight streaming at GEMV launch granularity: 24 separate buffers of S MB read back-to-back
// (one kernel each, same stream), per (workgroups, 16-byte loads in flight per lane).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ p, long n16, unsigned* sink) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  unsigned acc = 0;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  for (; i < n16; i += stride) { u32x4 v = p[i]; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (acc == 0x12345678u) *sink = acc;
}
static int g_nb = 24;   // buffers cycled over (24 x 90 MB never fits a cache; 1-2 x 45/90 MB fits the 256 MB Infinity Cache)
template <int U, bool NT>
float run(int wgs, char** bufs0, long bytes, unsigned* sink, hipStream_t st) {
  char* bufs[24];
  for (int i = 0; i < 24; ++i) bufs[i] = bufs0[i % g_nb];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) for (int i = 0; i < 24; ++i) rd<U, NT><<<wgs, 256, 0, st>>>((const u32x4*)bufs[i], bytes / 16, sink);
  hipEventRecord(e0, st);
  const int rep = 10;
  for (int r = 0; r < rep; ++r) for (int i = 0; i < 24; ++i) rd<U, NT><<<wgs, 256, 0, st>>>((const u32x4*)bufs[i], bytes / 16, sink);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / rep / 24 * 1e3f;
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  unsigned* sink; hipMalloc(&sink, 4);
  const long maxb = 90l << 20;
  char* bufs[24];
  for (int i = 0; i < 24; ++i) { hipMalloc(&bufs[i], maxb); hipMemset(bufs[i], i + 1, maxb); }
  const long sizes[] = {8l << 20, 25l << 20, 45l << 20, 90l << 20};
  const int wgs[] = {256, 512, 1024, 2048, 4096};
  for (long s : sizes) for (int g : wgs) {
    float a = run<1, true>(g, bufs, s, sink, st), b = run<2, true>(g, bufs, s, sink, st), c = run<4, true>(g, bufs, s, sink, st),
          d = run<8, true>(g, bufs, s, sink, st), e = run<4, false>(g, bufs, s, sink, st);
    printf("%3ld MiB wgs=%4d  us (TB/s): U1 %6.2f (%.2f)  U2 %6.2f (%.2f)  U4 %6.2f (%.2f)  U8 %6.2f (%.2f)  U4-temporal %6.2f (%.2f)\n", s >> 20, g,
           a, s / a / 1e6, b, s / b / 1e6, c, s / c / 1e6, d, s / d / 1e6, e, s / e / 1e6);
  }
  // cache residency: the same 1 / 2 / 4 buffers read over and over
  const int nbs[] = {1, 2, 4};
  for (long s : sizes) for (int nb : nbs) {
    g_nb = nb;
    float c = run<4, true>(512, bufs, s, sink, st), e = run<4, false>(512, bufs, s, sink, st), f = run<1, false>(2048, bufs, s, sink, st);
    printf("%3ld MiB cycling over %d buffer(s) (%ld MiB): U4-nt/512 %6.2f us (%.2f TB/s)  U4-temporal/512 %6.2f (%.2f)  U1-temporal/2048 %6.2f (%.2f)\n",
           s >> 20, nb, (s >> 20) * nb, c, s / c / 1e6, e, s / e / 1e6, f, s / f / 1e6);
  }
  return 0;
}
