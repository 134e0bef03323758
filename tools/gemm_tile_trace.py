"""Tile-by-tile timeline of the persistent 8-phase GEMM on the four DiT shapes (timing build, -DLD_GEMM_TRACE).

  tools/gemm_tile_trace.py build    cross-compiles landiff_amd/variants/libgemm_trace.so
  tools/gemm_tile_trace.py          (GPU box) per shape: the launch window of the 8-phase kernel, main loop vs epilogue time per
                                    tile, tiles and last end per XCD, how long CUs sit idle at the end
(the 128 x 128 tail launch of a shape is not traced: it shows up as the gap to the HIP-event time).
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "landiff_amd", "variants", "libgemm_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    csrc = os.path.join(ROOT, "landiff_amd", "csrc")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result",
                    "-DLD_GEMM_TRACE", "-c", "ld_gemm.hip", "-o", "/tmp/gemm_trace.o"], check=True, cwd=csrc)
    objs = [os.path.join(csrc, "obj", f) for f in os.listdir(os.path.join(csrc, "obj")) if f.endswith(".o") and f != "ld_gemm.o"]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, "/tmp/gemm_trace.o"] + objs, check=True)
    print("built", LIB)
    sys.exit(0)

os.environ["LANDIFF_HIP_LIB"] = LIB
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from landiff_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.ld_gemm_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_int64]
lib.ld_gemm_trace_set.restype = ctypes.c_int
M, D = 35552, 1920
dev = "cuda"
torch.manual_seed(0)


def rnd(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)


x = rnd(M, D); x4 = rnd(M, 4 * D); gate = rnd(2, 12 * D); resid = rnd(M, D)
B, Ntok, H, Npad = 2, M // 2, 30, (M // 2 + 127) // 128 * 128
q = torch.zeros(B, H, Npad, 64, device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros(B, H, 64, Npad, device=dev, dtype=torch.bfloat16)
ln = tuple(rnd(64) for _ in range(4))
wq, bq = rnd(3 * D, D, sc=0.02), rnd(3 * D)
w1, b1 = rnd(D, D, sc=0.02), rnd(D); o1 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
w2, b2 = rnd(4 * D, D, sc=0.02), rnd(4 * D); o2 = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
w3, b3 = rnd(D, 4 * D, sc=0.02), rnd(D)
cases = {
    "qkv": lambda: ops.gemm_qkv_heads(x, wq, bq, q, k, vt, B, Ntok, H, Npad, ln),
    "proj": lambda: ops.gemm(x, w1, out=o1, bias=b1, resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=2 * D, gate_off_txt=8 * D, rows_per_batch=M // 2, text_len=226),
    "ff1": lambda: ops.gemm(x, w2, out=o2, bias=b2, act="gelu_tanh"),
    "ff2": lambda: ops.gemm(x4, w3, out=o1, bias=b3, resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=5 * D, gate_off_txt=11 * D, rows_per_batch=M // 2, text_len=226),
}
CAP = 8192
buf = torch.zeros(1 + 4 * CAP, device=dev, dtype=torch.int64)
for name, fn in cases.items():
    assert lib.ld_gemm_trace_set(None, 0) == 0
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    assert lib.ld_gemm_trace_set(ctypes.c_void_p(buf.data_ptr()), CAP) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    assert lib.ld_gemm_trace_set(None, 0) == 0
    raw = buf.cpu().numpy().astype(np.uint64)
    n = int(raw[0]); d = raw[1:1 + 4 * n].reshape(n, 4)
    t0, t1, t2 = (d[:, i].astype(np.float64) for i in range(3))
    base = t0.min(); t0, t1, t2 = (t0 - base) / 100.0, (t1 - base) / 100.0, (t2 - base) / 100.0
    xcc = ((d[:, 3] >> np.uint64(48)) & np.uint64(0xf)).astype(np.int64)
    hw = ((d[:, 3] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
    cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)
    print(f"== {name}: launch by HIP events {e0.elapsed_time(e1) * 1000:.0f} us (incl. the 128 x 128 tail launch, if any); 8-phase kernel window {t2.max():.0f} us, {n} tiles on {len(set(cu.tolist()))} CUs")
    print(f"   per tile: main loop median {np.median(t1 - t0):.1f} us (p5 {np.percentile(t1 - t0, 5):.1f}, p95 {np.percentile(t1 - t0, 95):.1f}); epilogue median {np.median(t2 - t1):.1f} us (p95 {np.percentile(t2 - t1, 95):.1f})"
          f" = {100 * (t2 - t1).sum() / (t2 - t0).sum():.1f} % of the tile time")
    ends = []
    for xx in sorted(set(xcc.tolist())):
        m = xcc == xx
        ends.append(t2[m].max())
        print(f"   xcc {xx}: {m.sum():4d} tiles, tile median {np.median((t2 - t0)[m]):.1f} us, last end {t2[m].max():.0f} us")
    last_per_cu = np.array([t2[cu == c].max() for c in sorted(set(cu.tolist()))])
    busy = np.array([(t2 - t0)[cu == c].sum() for c in sorted(set(cu.tolist()))])
    print(f"   CUs: last end min {last_per_cu.min():.0f} / median {np.median(last_per_cu):.0f} / max {last_per_cu.max():.0f} us; busy {100 * busy.sum() / (len(busy) * t2.max()):.1f} % of the window;"
          f" XCD end spread {max(ends) - min(ends):.0f} us")
