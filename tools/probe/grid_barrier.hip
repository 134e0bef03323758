// Probe: cost and correctness of a software grid barrier (agent-scope atomics) on MI355X, for a persistent
// multi-phase kernel.  Every phase each workgroup writes one value, after the barrier it reads ALL workgroups'
// values (cross-XCD visibility check) -- the pattern of a fused GEMV chain (phase output = next phase input).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (MODE == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1l << 24)) { ok = false; break; }
      }
    } else if (MODE == 1) {          // release once, poll relaxed, acquire once
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1l << 24)) { ok = false; break; }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);      // system scope fence; agent would do
    } else {                         // lower bound: relaxed only (no cache maintenance) -- may read stale data
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1l << 24)) { ok = false; break; }
      }
    }
  }
  __syncthreads();
  return ok;
}
template <int MODE>
__global__ __launch_bounds__(256) void phases(unsigned* ctr, float* buf, float* result, int nphase, int* err) {
  const int G = gridDim.x;
  float acc = 0.f;
  for (int ph = 0; ph < nphase; ++ph) {
    float* cur = buf + (ph & 1) * G;
    if (threadIdx.x == 0) cur[blockIdx.x] = (float)(ph + 1) + acc * 1e-9f;
    if (!grid_barrier<MODE>(ctr, (unsigned)(ph + 1) * G)) { if (threadIdx.x == 0) *err = 1; return; }
    float s = 0.f;
    for (int i = threadIdx.x; i < G; i += 256) s += cur[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    acc = s;      // only a per-wave partial; enough to create the dependency
    if (threadIdx.x == 0 && cur[(blockIdx.x + G / 2) % G] < (float)(ph + 1)) *err = 2;     // stale read of a remote value
  }
  if (threadIdx.x == 0) result[blockIdx.x] = acc;
}
int main() {
  unsigned* ctr; float *buf, *res; int* err;
  hipMalloc(&ctr, 4); hipMalloc(&buf, 2 * 4096 * 4); hipMalloc(&res, 4096 * 4); hipMalloc(&err, 4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {256, 512, 1024};
  for (int mode = 0; mode < 3; ++mode) for (int G : grids) for (int nph : {1, 101}) {
    float best = 1e9f; int herr = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemsetAsync(ctr, 0, 4, st); hipMemsetAsync(err, 0, 4, st); hipMemsetAsync(buf, 0, 2 * 4096 * 4, st);
      hipEventRecord(e0, st);
      if (mode == 0) phases<0><<<G, 256, 0, st>>>(ctr, buf, res, nph, err);
      else if (mode == 1) phases<1><<<G, 256, 0, st>>>(ctr, buf, res, nph, err);
      else phases<2><<<G, 256, 0, st>>>(ctr, buf, res, nph, err);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
      int h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost); herr |= h;
    }
    printf("mode %d G=%4d phases=%3d: %8.2f us total%s\n", mode, G, nph, best * 1e3f, herr ? "  ERROR (timeout=1 / stale=2)" : "");
  }
  return 0;
}
