// Unmasked DiT attention (joint text+video, head_dim 64) with a 64-query-row wave tile.  Same reference op, LDS images,
// LDS-DMA ring and max-free fast pass / safe fallback as ld_attn_p16.hip (sat attention_fn_default ->
// F.scaled_dot_product_attention, landiff/diffusion/dit_video_concat.py:636-664); what changes is how much work every byte
// brought into the CU and every fragment read from LDS feeds.
//
// Why: the ablation of ld_attn_p16.hip (profiles/r02_attn_ablation.txt) shows that the loop does not wait on anything; what
// its non-MFMA parts cost is their VOLUME -- every other K / V^T fragment read and every other LDS-DMA piece left out is worth
// +15 % on the launch.  A wave that owns 64 query rows (four 16-row blocks qb) uses each K / V^T fragment for four MFMAs
// instead of two, and a workgroup of four such waves (256 query rows) uses each DMA'd tile for twice the rows: half the
// ds_read_b128 and half the L2 -> LDS bytes per MFMA, the same 32 v_exp_f32 + 16 v_cvt_pk_bf16_f32 per 36 MFMAs.
//
// Register budget (two waves per SIMD = 256 registers): O^T 64 + Q^T fragments 32 leave room for the score tile only at
// 32-key granularity, so the software pipeline runs on HALF tiles (key group kg of tile t = half 2t + kg):
//   S^T[b][qb] (16 keys x 16 q, b = 0..1)  = K[2kg + b] (A) . Q^T[qb] (B)      16 MFMAs, 4 fragment reads
//   O^T[db][qb] (16 d x 16 q)             += V^T[db][kg] (A) . P[qb] (B)        16 MFMAs, 4 fragment reads
//   l[qb]                                 += ones . P[qb]                        4 MFMAs
// with QK^T of half h+1 issued over the exp2 / packing of half h, and PV of half h over the first exp2 of half h+1, exactly as
// ld_attn_p16.hip does per tile.  The K-row permutation (row rho = 4h + r of block 2kg + b holds key kg*32 + 8h + 4b + r)
// makes the S^T accumulators of a half the PV B fragment as they are.
//
// LDS ring: four K and four V^T slots of 8 KB, one period = two tiles = four halves; the K fragments run one tile ahead of the
// V^T fragments (half h+2), so a period that starts at tile t needs K_{t+1}, K_{t+2}, V_t, V_{t+1} resident and refills the
// other slots with K_{t+3}, K_{t+4} (K waves) and V_{t+2}, V_{t+3} (V^T waves); one workgroup barrier per period.
#include "ld_attn.h"
#include <mutex>

namespace {

#include "ld_attn_q64_body.h"

__global__ __launch_bounds__(256, 2) void ld_attn_q64_kernel(AttnParams p, int force_safe) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_q64_body<Q64_FAST>(p, force_safe, smem, xcd_remap(blockIdx.x, gridDim.x));
}

// ---- dynamic form: the XCDs of one chip do not run at one speed ----
// tools/attn_q64_trace.py (profiles/r04_attn_q64_wg_trace.txt): under the package power cap the eight XCDs hold different clocks
// (1741 ... 1876 MHz in one launch), the hardware deals workgroups to them round-robin -- 525 each -- and the launch ends when
// the slowest XCD does (3724 us against 3482 us for the fastest).  Here one workgroup per slot (2 x CUs) PULLS query blocks: every
// XCD owns the contiguous range of (remapped) block indices it would have been dealt, so its K / V stay in its L2, and takes
// them in order through an agent-scope counter; an XCD that runs dry takes blocks from the others' ranges.  Same per-block
// arithmetic: bit-identical output.
//
// The counters are the ONE piece of state this library keeps: Q64_QSETS sets of eight words in static device memory (one copy
// per device, the library still allocates nothing).  What makes a set safe to use:
//   * a set belongs to one (device, stream) pair for the life of the process (q64_set_for): two launches can only share a set
//     if they are on the same stream, where they run one after the other;
//   * the launcher zeroes the set ON THE LAUNCH STREAM right before the kernel (hipMemsetAsync of 64 bytes), so whatever an
//     earlier launch left there -- an aborted kernel, a poisoned set (ld_attn_queue_poke) -- cannot reach this one; the kernel
//     itself never needs to clean up;
//   * a launch that cannot get a set of its own -- more than Q64_QSETS streams of one device have launched attention since the
//     last ld_reset(), or the stream is being captured into a graph (a replayed graph may run on any stream, next to anything)
//     -- takes the static kernel: the same bits, the hardware's own dispatch.
constexpr int Q64_QSETS = 64;
constexpr int Q64_MAXDEV = 16;
__device__ unsigned g_q64_queue[Q64_QSETS][16];          // [set][0..7]: next block of XCD x's range (64 bytes per set)

__global__ __launch_bounds__(256, 2) void ld_attn_q64_dyn_kernel(AttnParams p, int force_safe, int total, int set) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* bcast = (int*)(smem + 8 * KTILE_BYTES + 32);
  int* flags = (int*)(smem + 8 * KTILE_BYTES);             // the body's per-wave "a row left the window" words
  unsigned* Q = g_q64_queue[set];
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const int per = total / 8, rem = total % 8;            // xcd_remap: XCD x owns [base(x), base(x) + cnt(x))
  auto base = [&](int x) { return x < rem ? x * (per + 1) : rem * (per + 1) + (x - rem) * per; };
  auto cnt = [&](int x) { return per + (x < rem ? 1 : 0); };
#ifndef LD_Q64_NO_FBCOUNT
  if (threadIdx.x == 0) flags[0] = flags[1] = flags[2] = flags[3] = 0;
#endif
  for (;;) {
    if (threadIdx.x == 0) {
#ifndef LD_Q64_NO_FBCOUNT
      // Word 8 of the set: blocks of this launch that left the fast pass's window and were recomputed (ld_attn_last_fallbacks).
      // The block just finished left its per-wave flags in LDS (the barrier that ends an iteration orders them); counted and
      // cleared HERE, inside the one thread-0 section of the iteration: a second `if (threadIdx.x == 0)` at the end of the loop
      // body gets threaded into this one by hipcc, which puts the barriers below inside a divergent loop (the launch hangs).
      if ((flags[0] | flags[1] | flags[2] | flags[3]) != 0) __hip_atomic_fetch_add(&Q[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flags[0] = flags[1] = flags[2] = flags[3] = 0;      // (a block past Nq returns before it writes them)
#endif
      int bid = -1;
      for (int kx = 0; kx < 8 && bid < 0; ++kx) {        // own range first, then the neighbours'
        const int x = (int)((xcc + kx) & 7u);
        if (__hip_atomic_load(&Q[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)cnt(x)) continue;
        const unsigned i = __hip_atomic_fetch_add(&Q[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i < (unsigned)cnt(x)) bid = base(x) + (int)i;
      }
      bcast[0] = bid;
    }
    __syncthreads();
    const int bid = bcast[0];
    __syncthreads();
    if (bid < 0) break;
    attn_q64_body<Q64_FAST>(p, force_safe, smem, bid);
    __syncthreads();
  }
}

__global__ void ld_q64_queue_fill_kernel(int set, unsigned value) {
  const int s = blockIdx.x;
  if ((set < 0 || s == set) && threadIdx.x < 16) g_q64_queue[s][threadIdx.x] = value;
}

// ---- host side of the counter sets ----
struct Q64Owner { int dev; hipStream_t st; };
std::mutex g_q64_mu;
Q64Owner g_q64_owner[Q64_MAXDEV][Q64_QSETS];
int g_q64_owners[Q64_MAXDEV];                            // sets handed out per device
unsigned* g_q64_base[Q64_MAXDEV];                        // device address of g_q64_queue, per device
int g_q64_slots[Q64_MAXDEV];                             // 2 x CUs, per device

// the set of (dev, st), handing out a new one on first sight; -1 when the device's sets are all taken
int q64_set_for(int dev, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_q64_mu);
  const int n = g_q64_owners[dev];
  for (int i = 0; i < n; ++i) if (g_q64_owner[dev][i].st == st) return i;
  if (n >= Q64_QSETS) return -1;
  g_q64_owner[dev][n] = Q64Owner{dev, st};
  g_q64_owners[dev] = n + 1;
  return n;
}

int q64_device_info(int* dev_out, unsigned** base_out, int* slots_out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return ld_set_error(LD_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= Q64_MAXDEV) { *dev_out = -1; return LD_OK; }         // no dynamic form on such a device
  std::lock_guard<std::mutex> lk(g_q64_mu);
  if (!g_q64_base[dev]) {
    void* sym = nullptr;
    e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_q64_queue));
    if (e != hipSuccess) return ld_set_error(LD_ERR_LAUNCH, "hipGetSymbolAddress(g_q64_queue) on device %d: %s", dev, hipGetErrorString(e));
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    g_q64_slots[dev] = 2 * cus;
    g_q64_base[dev] = (unsigned*)sym;
  }
  *dev_out = dev; *base_out = g_q64_base[dev]; *slots_out = g_q64_slots[dev];
  return LD_OK;
}

}  // namespace

void ld_attn_set_last_kernel(const char* name);   // ld_attn.hip
void ld_attn_set_fallback_source(const unsigned* src, int kind);   // ld_attn.hip

// LD_ATTN_SAFE=1 forces the running-max pass (testing).  LD_ATTN_DYN=0: the hardware's round-robin dispatch of one workgroup per
// query block instead of the dynamic form (the default for grids of at least four rounds of the chip's slots).
int ld_attn_q64_launch(const AttnParams& p, hipStream_t st) {
  constexpr int SMEM = 8 * KTILE_BYTES + 64;
  static int k_safe = LD_KNOB_UNSET, k_dyn = LD_KNOB_UNSET;
  const int safe = ld_knob("LD_ATTN_SAFE", 0, &k_safe);
  static thread_local LdSmemCache cache{};
  if (int rc = ld_ensure_dyn_smem((const void*)ld_attn_q64_kernel, SMEM, &cache)) return rc;
  const int total = (int)((long)p.B * p.H * ((p.Npad + Q64_ROWS - 1) / Q64_ROWS));
  if (ld_knob("LD_ATTN_DYN", 1, &k_dyn) > 0 && !safe) {
    int dev = -1, slots = 0; unsigned* base = nullptr;
    if (int rc = q64_device_info(&dev, &base, &slots)) return rc;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (dev >= 0 && total >= 4 * slots && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
      const int set = q64_set_for(dev, st);
      if (set >= 0) {
        static thread_local LdSmemCache cache_d{};
        if (int rc = ld_ensure_dyn_smem((const void*)ld_attn_q64_dyn_kernel, SMEM, &cache_d)) return rc;
        hipError_t e = hipMemsetAsync(base + set * 16, 0, 64, st);
        if (e != hipSuccess) return ld_set_error(LD_ERR_LAUNCH, "ld_attn_fwd_bf16(q64 dynamic): zeroing counter set %d: %s", set, hipGetErrorString(e));
        ld_attn_set_last_kernel("ld_attn_q64_dyn_kernel");
        ld_attn_set_fallback_source(base + set * 16 + 8, 1);
        hipLaunchKernelGGL(ld_attn_q64_dyn_kernel, dim3((unsigned)slots), dim3(256), SMEM, st, p, safe, total, set);
        return ld_check_launch("ld_attn_fwd_bf16(q64 dynamic)");
      }
    }
  }
  ld_attn_set_last_kernel(safe ? "ld_attn_q64_kernel[safe pass forced]" : "ld_attn_q64_kernel");
  ld_attn_set_fallback_source(nullptr, safe ? 0 : 2);
  hipLaunchKernelGGL(ld_attn_q64_kernel, dim3((unsigned)total), dim3(256), SMEM, st, p, safe);
  return ld_check_launch("ld_attn_fwd_bf16(q64)");
}

// Forget which stream owns which counter set on the current device and zero the sets (on `stream`).  The caller guarantees that no
// ld_attn_fwd_bf16 launch of this device is in flight or enqueued on another stream (synchronise first): afterwards the next 64
// streams to launch attention get sets again.  Needed only by a process that keeps creating streams, or to put the library back
// into its initial state after a device error; an ordinary caller never has to call it.
LD_API int ld_reset(void* stream) {
  int dev = -1, slots = 0; unsigned* base = nullptr;
  if (int rc = q64_device_info(&dev, &base, &slots)) return rc;
  if (dev < 0) return LD_OK;
  { std::lock_guard<std::mutex> lk(g_q64_mu); g_q64_owners[dev] = 0; }
  hipError_t e = hipMemsetAsync(base, 0, sizeof(unsigned) * Q64_QSETS * 16, (hipStream_t)stream);
  if (e != hipSuccess) return ld_set_error(LD_ERR_LAUNCH, "ld_reset: %s", hipGetErrorString(e));
  return LD_OK;
}

// Test hook: fill counter set `set` (every set when set < 0) of the current device with `value` -- what an aborted launch would
// leave behind.  The next launches must not care (tests/test_gpu_attn.py).
LD_API int ld_attn_queue_poke(int32_t set, uint32_t value, void* stream) {
  LD_REQUIRE(set < Q64_QSETS, "ld_attn_queue_poke: set %d of %d", (int)set, Q64_QSETS);
  hipLaunchKernelGGL(ld_q64_queue_fill_kernel, dim3(Q64_QSETS), dim3(64), 0, (hipStream_t)stream, (int)set, (unsigned)value);
  return ld_check_launch("ld_attn_queue_poke");
}
