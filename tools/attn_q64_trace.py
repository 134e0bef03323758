"""Where the DiT attention launch spends its wall time, workgroup by workgroup (timing build of ld_attn_q64.hip, -DLD_Q64_TRACE).

  tools/attn_q64_trace.py build     cross-compiles landiff_amd/variants/libq64_trace.so (the shipped objects + the traced kernel)
  tools/attn_q64_trace.py           (GPU box) one launch at B 2 / H 30 / N 17 776; every workgroup records s_memrealtime at its
                                    start and end, its shader cycles and HW_ID / XCC_ID.
Prints: the launch window; per XCD the first start / last end and the workgroups it ran; per CU the busy fraction of its two
slots and the gaps between one workgroup's end and the next one's start; the distribution of workgroup durations and clocks.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "landiff_amd", "variants", "libq64_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    csrc = os.path.join(ROOT, "landiff_amd", "csrc")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result", "-fno-slp-vectorize",
                    "-DLD_Q64_TRACE", "-c", "ld_attn_q64.hip", "-o", "/tmp/q64_trace.o"], check=True, cwd=csrc)
    objs = [os.path.join(csrc, "obj", f) for f in os.listdir(os.path.join(csrc, "obj")) if f.endswith(".o") and f != "ld_attn_q64.o"]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, "/tmp/q64_trace.o"] + objs, check=True)
    print("built", LIB)
    sys.exit(0)

os.environ["LANDIFF_HIP_LIB"] = LIB
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from landiff_amd import ops  # noqa: E402

B, H, N = 2, 30, 17776
Npad = (N + 127) // 128 * 128
q = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, 64, Npad, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
nwg = B * H * ((Npad + 255) // 256)
dbg = torch.zeros(nwg * 4, device="cuda", dtype=torch.int64)
for _ in range(8):
    ops.attn_fwd(q, k, vt, out, N, N, 0.125, kt_min=dbg.view(torch.int32))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.attn_fwd(q, k, vt, out, N, N, 0.125, kt_min=dbg.view(torch.int32))
e1.record(); torch.cuda.synchronize()
d = dbg.view(-1, 4).cpu().numpy().astype(np.uint64)
ran = d[:, 1] > 0
d = d[ran]
t0, t1, cyc = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64), d[:, 2].astype(np.float64)
hw = (d[:, 3] & np.uint64(0xffffffff)).astype(np.int64); xcc = ((d[:, 3] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0              # microseconds (100 MHz counter)
dur = t1 - t0
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"launch by HIP events {e0.elapsed_time(e1) * 1000:.0f} us; first workgroup start -> last end {t1.max():.0f} us; {len(d)} workgroups ran")
print(f"workgroup duration: median {np.median(dur):.1f} us, p5 {np.percentile(dur, 5):.1f}, p95 {np.percentile(dur, 95):.1f}, max {dur.max():.1f}; "
      f"clock median {np.median(cyc / dur):.0f} MHz (p5 {np.percentile(cyc / dur, 5):.0f}, p95 {np.percentile(cyc / dur, 95):.0f})")
print(f"sum of durations / 512 slots = {dur.sum() / 512:.0f} us  ({100 * dur.sum() / 512 / t1.max():.1f} % of the window)")
print("start times of the first generation: p50 %.1f us, p95 %.1f, max of the first 512 starts %.1f" % (np.percentile(np.sort(t0)[:512], 50), np.percentile(np.sort(t0)[:512], 95), np.sort(t0)[:512].max()))
print("per XCD: workgroups, median duration, last end")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"   xcc {x}: {m.sum():4d} wgs, {len(set(key[m].tolist())):3d} CUs, median {np.median(dur[m]):.1f} us, clock {np.median(cyc[m] / dur[m]):.0f} MHz, last end {t1[m].max():.0f} us, first idle slot at {np.sort(t1[m])[-min(64, m.sum()):].min():.0f} us")
gaps, busy = [], []
for c in sorted(set(key.tolist())):
    m = key == c
    ev = sorted(zip(t0[m], t1[m]))
    # two slots per CU: replay with a 2-slot greedy to find, for each start, the end it follows
    ends = []
    for s, e in ev:
        prev = [x for x in ends if x <= s + 0.5]
        if prev and len(ends) >= 2:
            p = max(prev); gaps.append(s - p); ends.remove(p)
        ends.append(e)
    busy.append(sum(e - s for s, e in ev) / (2 * t1.max()))
gaps = np.array(gaps)
print(f"{len(set(key.tolist()))} CUs seen; slot busy fraction per CU: median {np.median(busy):.3f}, min {min(busy):.3f}, max {max(busy):.3f}")
print(f"gap between a workgroup's end and the next start on the same CU: median {np.median(gaps):.1f} us, p95 {np.percentile(gaps, 95):.1f}, sum per slot {gaps.sum() / 512:.0f} us")
nper = np.array([int((key == c).sum()) for c in sorted(set(key.tolist()))])
print("workgroups per CU: min %d, median %d, max %d" % (nper.min(), np.median(nper), nper.max()))
late = np.sort(t1)[::-1][:8]
print("last eight ends (us):", " ".join(f"{x:.0f}" for x in late), "| time at which half the slots were idle: %.0f us" % np.sort(t1)[-256])
