"""Samples socket power / clocks (rocm-smi, a child process) while one kernel family runs in a loop for a few seconds.
usage: python tools/power_probe.py [attn|gemm|gemv|idle] [seconds]"""
import json, subprocess, sys, threading, time, torch
sys.path.insert(0, ".")
from landiff_amd import ops
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
samples, stop = [], False
def poll():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)["card0"]
            samples.append({k: v for k, v in d.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "junction", "hotspot"))})
        except Exception as e:  # noqa
            samples.append({"err": str(e)[:80]})
        time.sleep(0.25)
dev = "cuda"
if what == "attn":
    B, H, N = 2, 30, 17776; Np = (N + 127) // 128 * 128
    q = torch.randn(B, H, Np, 64, device=dev).to(torch.bfloat16); k = torch.randn_like(q); vt = torch.randn(B, H, 64, Np, device=dev).to(torch.bfloat16)
    out = torch.empty(B, N, H * 64, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.attn_fwd(q, k, vt, out, N, N, 0.125); flop = 4.0 * B * H * N * N * 64
elif what == "gemm":
    M, K, Nn = 35552, 1920, 7680
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(Nn, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.gemm(a, w, out=out); flop = 2.0 * M * K * Nn
elif what == "gemv":
    K, Nn = 2048, 11008
    x = torch.randn(2, K, device=dev).to(torch.bfloat16); ws = [(torch.randn(Nn, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(24)]
    out = torch.empty(2, Nn, device=dev, dtype=torch.bfloat16)
    def fn():
        for w in ws: ops.gemv(x, w, out)
    flop = 24 * 2.0 * 2 * K * Nn
else:
    fn = lambda: time.sleep(0.01); flop = 0.0
for _ in range(3): fn()
torch.cuda.synchronize()
th = threading.Thread(target=poll); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
stop = True; th.join()
print(f"{what}: {n} launches in {dt:.2f} s = {dt / n * 1e3:.3f} ms each, {flop * n / dt / 1e12:.0f} TFLOP/s")
for s in samples[2:]: print("  ", s)
