"""Interleaved A/B timing of the DiT attention launch (B=2, H=30, N=17776) in ONE process: the 64-row wave tile (ld_attn_q64)
against the 128-row one-wave-per-SIMD tile (ld_attn_q128, three exp2 splits).  LD_ATTN_Q128 / LD_ATTN_NPRE are read on every
call, so the arms alternate round by round on the same box, clock state and inputs (cdna_hip_programming.md 5.4 rule 24).
  python tools/attn_ab.py [rounds] [launches per arm and round]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from landiff_amd import _lib, ops  # noqa: E402

B, H, N = 2, 30, int(os.environ.get("N", "17776"))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
it = int(sys.argv[2]) if len(sys.argv) > 2 else 10
Npad = (N + 127) // 128 * 128
q = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
k = torch.randn(B, H, Npad, 64, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, H, 64, Npad, device="cuda").to(torch.bfloat16)
out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
arms = [("q64", {"LD_ATTN_Q128": "0", "LD_ATTN_DYN": "0"}), ("q64 dyn", {"LD_ATTN_Q128": "0", "LD_ATTN_DYN": "1"}), ("q128 npre44", {"LD_ATTN_Q128": "1", "LD_ATTN_NPRE": "44"}),
        ("q128 npre36", {"LD_ATTN_Q128": "1", "LD_ATTN_NPRE": "36"}), ("q128 npre52", {"LD_ATTN_Q128": "1", "LD_ATTN_NPRE": "52"}),
        ("q128 s1 n36", {"LD_ATTN_Q128": "1", "LD_ATTN_NPRE": "1036"}), ("q128 s1 n40", {"LD_ATTN_Q128": "1", "LD_ATTN_NPRE": "1040"})]
if os.environ.get("ARMS"):
    arms = [a for a in arms if a[0] in os.environ["ARMS"].split(",")]
ref = None
times = {a: [] for a, _ in arms}
names = {}
for r in range(rounds + 1):
    for name, env in arms:
        os.environ.update(env)
        ops.attn_fwd(q, k, vt, out, N, N, 0.125)          # (also the warm-up round)
        names[name] = (_lib.load().ld_attn_last_kernel() or b"").decode()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            ops.attn_fwd(q, k, vt, out, N, N, 0.125)
        e1.record(); torch.cuda.synchronize()
        if r:
            times[name].append(e0.elapsed_time(e1) / it)
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref), name                # bit-identical arms
fl = 4.0 * B * H * N * N * 64
for name, _ in arms:
    t = sorted(times[name])
    med, best = t[len(t) // 2], t[0]
    print(f"{name:12s} [{names[name]}]: median {med:.3f} ms = {fl / med / 1e9:.0f} TFLOP/s ({fl / med / 1e9 / 2500:.3f} of peak), best {best:.3f} ms", flush=True)
