// Probe: cost and correctness of a software grid barrier on MI355X, for a persistent multi-phase kernel.  Every phase each
// workgroup writes one value, after the barrier it reads ALL workgroups' values (cross-XCD visibility check) -- the pattern of
// a fused GEMV chain (phase output = next phase input).
//   mode 0  release add / acquire poll (agent scope): the compiler's cache maintenance on every poll
//   mode 1  release add, relaxed poll, one acquire fence
//   mode 2  relaxed only, plain data accesses: lower bound, reads stale data
//   mode 3  relaxed counter; the DATA is written / read with agent-scope relaxed atomics (sc1: write-through, L2-coherent read)
//           and the stores are drained (vmcnt 0) before the arrival -- no cache maintenance at all
//   mode 4  as 3 with 8 arrival counters on separate cache lines (workgroup b arrives on counter b & 7), one lane polls each
//   mode 5  as 4, poll loop without s_sleep
//   mode 6  as 4 with 32 counters
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int CSTRIDE = 32;      // counters 128 bytes apart
template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned phase1, int G) {
  bool ok = true;
  if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned target = phase1 * (unsigned)G;
  if (MODE <= 3) {
    if (threadIdx.x == 0) {
      if (MODE == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1l << 24)) { ok = false; break; }
        }
      } else if (MODE == 1) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1l << 24)) { ok = false; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
      } else {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1l << 24)) { ok = false; break; }
        }
      }
    }
  } else {
    constexpr int NC = MODE == 6 ? 32 : 8;
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      if (lane == 0) __hip_atomic_fetch_add(ctr + (blockIdx.x & (NC - 1)) * CSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // counter c collects the workgroups b with b % NC == c
      const unsigned mine = lane < NC ? phase1 * (unsigned)((G - lane + NC - 1) / NC) : 0u;
      long spins = 0;
      while (true) {
        unsigned v = lane < NC ? __hip_atomic_load(ctr + lane * CSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (__builtin_amdgcn_ballot_w64(v < mine) == 0) break;
        if (MODE != 5) __builtin_amdgcn_s_sleep(1);
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
    const float mine = (float)(ph + 1) + acc * 1e-9f;
    if (threadIdx.x == 0) {
      if (MODE >= 3) __hip_atomic_store(cur + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else cur[blockIdx.x] = mine;
    }
    if (!grid_barrier<MODE>(ctr, (unsigned)(ph + 1), G)) { if (threadIdx.x == 0) *err = 1; return; }
    float s = 0.f;
    for (int i = threadIdx.x; i < G; i += 256) {
      const float v = MODE >= 3 ? __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cur[i];
      if (v < (float)(ph + 1)) *err = 2;                      // stale read of a remote value
      s += v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    acc = s;      // only a per-wave partial; enough to create the dependency
  }
  if (threadIdx.x == 0) result[blockIdx.x] = acc;
}
template <int MODE>
void run(int G, int nph, unsigned* ctr, float* buf, float* res, int* err, hipStream_t st) {
  phases<MODE><<<G, 256, 0, st>>>(ctr, buf, res, nph, err);
}
int main() {
  unsigned* ctr; float *buf, *res; int* err;
  hipMalloc(&ctr, 64 * CSTRIDE * 4); hipMalloc(&buf, 2 * 4096 * 4); hipMalloc(&res, 4096 * 4); hipMalloc(&err, 4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {256, 512, 768, 1024};
  for (int mode = 0; mode < 7; ++mode) for (int G : grids) for (int nph : {1, 201}) {
    float best = 1e9f; int herr = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemsetAsync(ctr, 0, 64 * CSTRIDE * 4, st); hipMemsetAsync(err, 0, 4, st); hipMemsetAsync(buf, 0, 2 * 4096 * 4, st);
      hipEventRecord(e0, st);
      switch (mode) {
        case 0: run<0>(G, nph, ctr, buf, res, err, st); break;
        case 1: run<1>(G, nph, ctr, buf, res, err, st); break;
        case 2: run<2>(G, nph, ctr, buf, res, err, st); break;
        case 3: run<3>(G, nph, ctr, buf, res, err, st); break;
        case 4: run<4>(G, nph, ctr, buf, res, err, st); break;
        case 5: run<5>(G, nph, ctr, buf, res, err, st); break;
        default: run<6>(G, nph, ctr, buf, res, err, st); break;
      }
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
      int h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost); herr |= h;
    }
    printf("mode %d G=%4d phases=%3d: %8.2f us total%s\n", mode, G, nph, best * 1e3f, herr ? "  ERROR (timeout=1 / stale=2)" : "");
  }
  return 0;
}
