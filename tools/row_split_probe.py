"""Would cutting the MLP half of a DiT layer-call into two ROW ranges pay?  (GPU probe, no product code.)

dense -> LayerNorm -> 4h -> 4h->h are row-wise: rows [0, 32768) are exactly 128 tile rows -- 4, 15 and 4 whole rounds of the
persistent GEMM, no 128 x 128 tail launch -- and rows [32768, 35552) (7.8 % of the work) can run as a second chain on another
stream, in the gaps the first leaves (tools/gemm_tile_trace.py: the tail launches cost 5-21 % of a GEMM for 2-8 % of its tiles).
Times LAYERS x (attention stand-in excluded) [dense -> LN -> 4h -> 4h->h] in two forms, alternated: one chain over all rows (the
product today) / two chains with a fork and a join per layer.  Checks bit-identity.
usage: python tools/row_split_probe.py [layers] [rounds] [split_row]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from landiff_amd import ops  # noqa: E402

LAYERS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
M0 = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
dev = "cuda"
torch.manual_seed(0)
B, N, D = 2, 17776, 1920
M = B * N
BF = torch.bfloat16


def rnd(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).to(BF)


x = rnd(M, D)
att = rnd(M, D)
gate = rnd(B, 12 * D, sc=0.1)
lnw, lnb = rnd(D), rnd(D, sc=0.1)
w1, b1 = rnd(D, D, sc=0.02), rnd(D, sc=0.02)
w2, b2 = rnd(4 * D, D, sc=0.02), rnd(4 * D, sc=0.02)
w3, b3 = rnd(D, 4 * D, sc=0.02), rnd(D, sc=0.02)
h = torch.empty(M, D, device=dev, dtype=BF)
ln = torch.empty(M, D, device=dev, dtype=BF)
mlp = torch.empty(M, 4 * D, device=dev, dtype=BF)
outA, outB = torch.empty(M, D, device=dev, dtype=BF), torch.empty(M, D, device=dev, dtype=BF)


def mlp_half(src, dst, r0, r1):
    """dense (gated residual) -> LN + modulate -> 4h (GELU) -> 4h->h (gated residual) on rows [r0, r1) of the [M, .] buffers."""
    r = slice(r0, r1)
    if r0 == 0:
        g = dict(gate=gate, gate_bstride=12 * D, rows_per_batch=N, text_len=226)
        md = dict(mod=gate, mod_bstride=12 * D, rows_per_batch=N, text_len=226)
    else:                                     # the second range lies inside batch 1's image rows
        assert r0 >= N + 226
        g = dict(gate=gate[1:], gate_bstride=12 * D, rows_per_batch=1 << 30, text_len=0)
        md = dict(mod=gate[1:], mod_bstride=12 * D, rows_per_batch=1 << 30, text_len=0)
    ops.gemm(att[r], w1, out=h[r], bias=b1, resid=src[r], gate_off_img=2 * D, gate_off_txt=8 * D, **g)
    ops.layernorm(h[r], lnw, lnb, ln[r], 1e-5, shift_img=3 * D, scale_img=4 * D, shift_txt=9 * D, scale_txt=10 * D, **md)
    ops.gemm(ln[r], w2, out=mlp[r], bias=b2, act="gelu_tanh")
    ops.gemm(mlp[r], w3, out=dst[r], bias=b3, resid=h[r], gate_off_img=5 * D, gate_off_txt=11 * D, **g)


s2 = torch.cuda.Stream()
ev_fork = [torch.cuda.Event() for _ in range(LAYERS)]
ev_join = [torch.cuda.Event() for _ in range(LAYERS)]


def run_one():
    src = x
    for i in range(LAYERS):
        dst = outA if i % 2 == 0 else outB
        mlp_half(src, dst, 0, M)
        src = dst
    return src


def run_split():
    cur = torch.cuda.current_stream()
    src = x
    for i in range(LAYERS):
        dst = outA if i % 2 == 0 else outB
        ev_fork[i].record(cur)                # (in the product: after the attention launch)
        s2.wait_event(ev_fork[i])
        mlp_half(src, dst, 0, M0)
        with torch.cuda.stream(s2):
            mlp_half(src, dst, M0, M)
            ev_join[i].record(s2)
        cur.wait_event(ev_join[i])            # (in the product: before the next layer's LayerNorm)
        src = dst
    return src


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


ref = run_one().clone()
torch.cuda.synchronize()
got = run_split()
torch.cuda.synchronize()
print(f"layers {LAYERS}, split at row {M0}: bit-identical to the one-chain form: {bool(torch.equal(got, ref))}")
forms = {"one chain": run_one, "row split": run_split}
acc = {k: [] for k in forms}
for _ in range(2):
    for fn in forms.values():
        fn()
for _ in range(ROUNDS):
    for name, fn in forms.items():
        acc[name].append(timed(fn) / LAYERS)
for name, v in acc.items():
    print(f"{name:10s} {min(v) * 1000:.0f} us per MLP half (min of {ROUNDS}), median {sorted(v)[len(v) // 2] * 1000:.0f}")
