"""Driver of tools/probe/pk_f32_coresidency.hip: runs every variant of the packed-fp32 sequence (i) on a quiet GPU, (ii) while DiT-sized
attention launches (LOAD=attn, the 64-row kernel; LOAD=gemm: the persistent GEMM) run on another stream, and prints mismatches (packed
result != scalar result, bitwise) in total and per quarter of the wave's lanes."""
import ctypes, os, subprocess, sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")
so = "/tmp/libpkprobe.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-fno-slp-vectorize", "-shared", "-fPIC", "-o", so, os.path.join(here, "pk_f32_coresidency.hip")], check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
nblocks, iters = int(os.environ.get("NBLOCKS", "8192")), int(os.environ.get("ITERS", "2000"))
n = nblocks * 64
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(n, device=dev, generator=g).to(torch.bfloat16).float(); B = torch.randn(n, device=dev, generator=g).to(torch.bfloat16).float()
ang = torch.rand(n, device=dev, generator=g) * 6.28
C, S = torch.cos(ang), torch.sin(ang)
bad = torch.zeros(64, device=dev, dtype=torch.int32); first = torch.zeros(n * 4, device=dev, dtype=torch.int32)
BF = torch.bfloat16
Bq, H, N = 2, 30, 17776; Npad = (N + 127) // 128 * 128
q = torch.randn(Bq, H, Npad, 64, device=dev).to(BF); k = torch.randn(Bq, H, Npad, 64, device=dev).to(BF); vt = torch.randn(Bq, H, 64, Npad, device=dev).to(BF)
ao = torch.zeros(Bq, N, H * 64, device=dev, dtype=BF)
M, D = 35552, 1920
x, w, bb = torch.randn(M, D, device=dev).to(BF), (torch.randn(4 * D, D, device=dev) * 0.02).to(BF), torch.randn(4 * D, device=dev).to(BF)
o = torch.empty(M, 4 * D, device=dev, dtype=BF)
load_stream, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
P = ctypes.c_void_p
names = {0: "distance 1 (no filler)", 1: "distance 2", 2: "distance 3 (as compiled)", 3: "distance 4", 4: "distance 3, no operand swizzle", 5: "distance 3 + s_nop 1", 6: "distance 3 + s_nop 7"}
def run(variant, load, reps=int(os.environ.get("REPS", "6"))):
    bad.zero_(); first.zero_()
    torch.cuda.synchronize()
    if load:
        with torch.cuda.stream(load_stream):
            for _ in range(reps * 3):
                if load == "attn": ops.attn_fwd(q, k, vt, ao, N, N, 0.125)
                else: ops.gemm(x, w, out=o, bias=bb, act="gelu_tanh")
    with torch.cuda.stream(side):
        for _ in range(reps):
            rc = lib.pk_probe_run(variant, P(A.data_ptr()), P(B.data_ptr()), P(C.data_ptr()), P(S.data_ptr()), P(bad.data_ptr()), P(first.data_ptr()),
                                  nblocks, iters, P(side.cuda_stream))
            assert rc == 0, rc
    torch.cuda.synchronize()
    b = bad.cpu().long()
    tot = int(b.sum())
    return tot, [int(b[i * 16:(i + 1) * 16].sum()) for i in range(4)]
total_ops = nblocks * 64 * iters * int(os.environ.get("REPS", "6"))
print(f"# {total_ops:.3g} evaluations of the sequence per cell; mismatches = packed result != scalar v_mul / v_mul / v_sub result (bitwise)")
print(f"# {'variant':36s} {'quiet':>10s} {'under attention (64-row kernel)':>34s} {'under the persistent GEMM':>28s}   per lane quarter 0-15 / 16-31 / 32-47 / 48-63 (under attention)")
for v in range(7):
    tq, _ = run(v, None); ta, qa = run(v, "attn"); tg, _ = run(v, "gemm")
    print(f"  {names[v]:36s} {tq:10d} {ta:34d} {tg:28d}   {qa}", flush=True)
v = 2
run(v, "attn")
fb = first.view(-1, 4).cpu()
rows = torch.nonzero(fb[:, 0] != fb[:, 1]).flatten()[:6]
for r in rows.tolist():
    as_f = lambda u: torch.tensor([u], dtype=torch.int32).view(torch.float32).item()
    print(f"  example (thread {r}, lane {r % 64}): packed {as_f(int(fb[r,0])):.9g} scalar {as_f(int(fb[r,1])):.9g}; a {as_f(int(fb[r,2])):.6g} c {as_f(int(fb[r,3])):.6g}  -> a (the value the register held BEFORE the in-place product) - s*b would be {as_f(int(fb[r,2])) - (as_f(int(fb[r,2]))*as_f(int(fb[r,3])) - as_f(int(fb[r,1]))):.9g}")
