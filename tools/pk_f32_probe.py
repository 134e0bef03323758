"""Driver of tools/probe/pk_f32_coresidency.hip: runs every variant of the packed-fp32 sequence (i) on a quiet GPU, (ii) while DiT-sized
attention launches (LOAD=attn, the 64-row kernel; LOAD=gemm: the persistent GEMM) run on another stream, and prints mismatches (packed
result != scalar result, bitwise) in total and per quarter of the wave's lanes."""
import ctypes, os, subprocess, sys, torch
os.environ["LD_TUNING"] = "1"          # the library re-reads its knobs per call: the attention kernel is switched between loads
sys.path.insert(0, ".")
from landiff_amd import ops
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")
so = "/tmp/libpkprobe.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-fno-slp-vectorize", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, os.path.join(here, "pk_f32_coresidency.hip")], check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
nblocks, iters = int(os.environ.get("NBLOCKS", "8192")), int(os.environ.get("ITERS", "2000"))
n = nblocks * 64
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(n, device=dev, generator=g).to(torch.bfloat16).float(); B = torch.randn(n, device=dev, generator=g).to(torch.bfloat16).float()
ang = torch.rand(n, device=dev, generator=g) * 6.28
C, S = torch.cos(ang), torch.sin(ang)
bad = torch.zeros(64, device=dev, dtype=torch.int32); first = torch.zeros(n * 8, device=dev, dtype=torch.int32)
BF = torch.bfloat16
Bq, H, N = 2, 30, 17776; Npad = (N + 127) // 128 * 128
q = torch.randn(Bq, H, Npad, 64, device=dev).to(BF); k = torch.randn(Bq, H, Npad, 64, device=dev).to(BF); vt = torch.randn(Bq, H, 64, Npad, device=dev).to(BF)
ao = torch.zeros(Bq, N, H * 64, device=dev, dtype=BF)
M, D = 35552, 1920
x, w, bb = torch.randn(M, D, device=dev).to(BF), (torch.randn(4 * D, D, device=dev) * 0.02).to(BF), torch.randn(4 * D, device=dev).to(BF)
o = torch.empty(M, 4 * D, device=dev, dtype=BF)
load_stream, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
cal_ops = torch.randn(1 << 19, device=dev).to(BF); cal_sink = torch.zeros(4, device=dev, dtype=torch.float32)
P = ctypes.c_void_p
names = {0: "distance 1 (no filler)", 1: "distance 2", 2: "distance 3 (as compiled)", 3: "distance 4", 4: "distance 3, no operand swizzle", 5: "distance 3 + s_nop 1", 6: "distance 3 + s_nop 7",
         7: "swizzled v_pk_add alone (scalar products)", 8: "producers without op_sel_hi broadcast", 9: "v_pk_mov_b32 op_sel:[1,0] alone"}
def run(variant, load, reps=int(os.environ.get("REPS", "6"))):
    bad.zero_(); first.zero_()
    torch.cuda.synchronize()
    if load:
        with torch.cuda.stream(load_stream):
            for _ in range(reps * 3):
                if load.startswith("attn"):
                    for kk in ("LD_ATTN_VARIANT", "LD_ATTN_Q64"): os.environ.pop(kk, None)
                    if load == "attn_plain": os.environ["LD_ATTN_VARIANT"] = "9"
                    if load == "attn_p16": os.environ["LD_ATTN_Q64"] = "0"
                    ops.attn_fwd(q, k, vt, ao, N, N, 0.125)
                elif load == "mfma":
                    from landiff_amd import _lib
                    fl = ctypes.c_double(0.0)
                    _lib.check(_lib.load().ld_calib_mfma_bf16(P(cal_ops.data_ptr()), cal_ops.numel() * 2, P(cal_sink.data_ptr()), 2048, 400, ctypes.byref(fl), P(load_stream.cuda_stream)), "calib")
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
for v in range(10):
    tq, _ = run(v, None); ta, qa = run(v, "attn"); tg, _ = run(v, "gemm")
    print(f"  {names[v]:36s} {tq:10d} {ta:34d} {tg:28d}   {qa}", flush=True)
print("# which co-resident kernel does it take?  (variant 'distance 3 (as compiled)' / 'swizzled v_pk_add alone' / 'no operand swizzle')")
for load, what in (("attn", "64-row attention (16x16x32 MFMAs, 2 waves per SIMD)"), ("attn_p16", "32-row attention (16x16x32 MFMAs)"), ("attn_plain", "plain attention kernel (32x32x16 MFMAs, 3 waves per SIMD)"),
                   ("mfma", "MFMA-only loop (16x16x32, one wave per SIMD, nothing else)"), ("gemm", "persistent 8-phase GEMM (160 KB LDS: nothing else fits on its CU)")):
    print(f"  {what:78s} {run(2, load)[0]:10d} {run(7, load)[0]:10d} {run(4, load)[0]:10d}", flush=True)
print("# synthetic co-resident load: 256-thread workgroups looping over ONE kind of instruction (same three victim variants)")
agg_sink = torch.zeros(4, device=dev, dtype=torch.float32)
kinds = ["v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "ds_read_b128", "v_permlane32_swap_b32", "v_pk_fma_f32 (no modifiers)", "v_pk_mul_f32 op_sel_hi:[0,1]",
         "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "ds_write_b128", "v_mov_b32 dpp row_shr:1"]
def run_synth(variant, kind, reps=6):
    bad.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(load_stream):
        for _ in range(3):
            assert lib.pk_aggressor_run(kind, P(agg_sink.data_ptr()), 4096, int(os.environ.get("TRIPS", "6000")), P(load_stream.cuda_stream)) == 0
    with torch.cuda.stream(side):
        for _ in range(reps):
            assert lib.pk_probe_run(variant, P(A.data_ptr()), P(B.data_ptr()), P(C.data_ptr()), P(S.data_ptr()), P(bad.data_ptr()), P(first.data_ptr()), nblocks, iters, P(side.cuda_stream)) == 0
    torch.cuda.synchronize()
    return int(bad.cpu().long().sum())
for kind, name in enumerate(kinds):
    print(f"  {name:34s} {run_synth(2, kind):10d} {run_synth(7, kind):10d} {run_synth(4, kind):10d}", flush=True)
import struct
f32 = lambda u: struct.unpack("<f", struct.pack("<I", u & 0xffffffff))[0]
import numpy as np
rn = lambda x: float(np.float32(x))
for v in (2, 7):
    run(v, "attn")
    fb = first.view(-1, 8).cpu()
    rows = torch.nonzero(fb[:, 0] != fb[:, 1]).flatten()
    print(f"# variant '{names[v]}': {rows.numel()} threads saw a mismatch; lanes of the first 24: {[int(r) % 64 for r in rows[:24].tolist()]}")
    for r in rows[:8].tolist():
        got, ref, a, b, c, s_, hi = (f32(int(fb[r, i])) for i in range(7)); it = int(fb[r, 7])
        ca, cb, sa, sb = rn(np.float32(c) * np.float32(a)), rn(np.float32(c) * np.float32(b)), rn(np.float32(s_) * np.float32(a)), rn(np.float32(s_) * np.float32(b))
        cands = {"c*a - s*b (right)": rn(np.float32(ca) - np.float32(sb)), "a - s*b (stale AB.lo)": rn(np.float32(a) - np.float32(sb)), "c*a - s*a (no swizzle)": rn(np.float32(ca) - np.float32(sa)),
                 "c*a + s*b (neg lost)": rn(np.float32(ca) + np.float32(sb)), "c*b - s*a (hi lane's value)": rn(np.float32(cb) - np.float32(sa)), "c*b - s*b": rn(np.float32(cb) - np.float32(sb)),
                 "c*a - b (stale P.hi = AB.hi)": rn(np.float32(ca) - np.float32(b))}
        match = [k for k, x in cands.items() if np.float32(x) == np.float32(got)]
        print(f"  lane {r % 64:2d} iteration {it}: packed {got:.9g}  scalar {ref:.9g}  (a {a:.6g} b {b:.6g} c {c:.6g} s {s_:.6g}; hi lane {hi:.9g})  matches: {match or 'none of the candidates'}")
