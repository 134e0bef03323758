"""Within-process A/B of two GEMM main loops on the four DiT shapes (real epilogues): interleaved rounds, median and min per
variant.  The library reads its knobs once per process, so each variant runs in a child process that loops over the shapes;
this driver interleaves child runs.  usage: python tools/gemm_ab.py "LD_GEMM_8P=0" "LD_GEMM_8P=1" [rounds]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from landiff_amd import ops
M, D = 35552, 1920
dev = "cuda"
torch.manual_seed(0)
import os
ZERO = os.environ.get("AB_ZERO") == "1"
def rnd(*s, sc=1.0): return torch.zeros(*s, device=dev, dtype=torch.bfloat16) if ZERO else (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
x = rnd(M, D); x4 = rnd(M, 4 * D); gate = rnd(2, 12 * D); resid = rnd(M, D)
B, Ntok, H, Npad = 2, M // 2, 30, (M // 2 + 127) // 128 * 128
q = torch.zeros(B, H, Npad, 64, device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros(B, H, 64, Npad, device=dev, dtype=torch.bfloat16)
ln = tuple(rnd(64) for _ in range(4))
wq, bq = rnd(3 * D, D, sc=0.02), rnd(3 * D)
cases = {
  "qkv": (2.0 * M * 3 * D * D, lambda: ops.gemm_qkv_heads(x, wq, bq, q, k, vt, B, Ntok, H, Npad, ln)),
}
w1, b1 = rnd(D, D, sc=0.02), rnd(D)
o1 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
cases["proj"] = (2.0 * M * D * D, lambda: ops.gemm(x, w1, out=o1, bias=b1, resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=2 * D, gate_off_txt=8 * D, rows_per_batch=M // 2, text_len=226))
w2, b2 = rnd(4 * D, D, sc=0.02), rnd(4 * D)
o2 = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
cases["ff1"] = (2.0 * M * 4 * D * D, lambda: ops.gemm(x, w2, out=o2, bias=b2, act="gelu_tanh"))
w3, b3 = rnd(D, 4 * D, sc=0.02), rnd(D)
cases["ff2"] = (2.0 * M * 4 * D * D, lambda: ops.gemm(x4, w3, out=o1, bias=b3, resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=5 * D, gate_off_txt=11 * D, rows_per_batch=M // 2, text_len=226))
res = {}
for name, (fl, fn) in cases.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    res[name] = (ms, fl / ms / 1e9)
print(json.dumps(res))
''' % ROOT
variants = [a for a in sys.argv[1:] if "=" in a or a == "-"]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
acc = {v: {} for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        for kv in v.split(","):
            if "=" in kv:
                k, val = kv.split("=", 1); env[k] = val
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(v, "FAILED", out.stderr[-2000:]); continue
        res = json.loads(out.stdout.strip().splitlines()[-1])
        for name, (ms, tf) in res.items():
            acc[v].setdefault(name, []).append((ms, tf))
for v in variants:
    line = [v.ljust(34)]
    tot = 0.0
    for name, xs in acc[v].items():
        ms = sorted(m for m, _ in xs)
        med = ms[len(ms) // 2]; tot += med
        tf = sorted(t for _, t in xs)[len(xs) // 2]
        line.append(f"{name} {med:.3f} ms (min {ms[0]:.3f}) {tf:.0f} TF")
    line.append(f"| sum {tot:.3f} ms")
    print("  ".join(line), flush=True)
