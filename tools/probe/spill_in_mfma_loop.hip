// Positive control for tools/audit_spills.py: a kernel that MUST spill inside a loop that issues MFMAs (more live values than the
// 128 registers a 1024-thread workgroup leaves a lane).  tests/test_cabi_and_host.py builds it and expects the audit to flag it.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern "C" __global__ __launch_bounds__(1024) void ld_probe_spill_in_mfma_loop(const float* in, float* out, int n) {
  float v[176];
#pragma unroll
  for (int j = 0; j < 176; ++j) v[j] = in[threadIdx.x * 176 + j];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[e + threadIdx.x]; b[e] = (__bf16)in[8 + e + threadIdx.x]; }
  for (int i = 0; i < n; ++i) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 176; ++j) v[j] = v[j] * 1.0001f + acc[j & 3];
  }
  float s = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
  for (int j = 0; j < 176; ++j) s += v[j];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
