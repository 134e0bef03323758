"""Would running the two CFG halves of the denoiser as two independent chains pay?  (GPU probe, no product code.)

A DiT layer-call at B = 2 is qkv GEMM -> attention -> dense -> 4h -> 4h->h on M = 35 552 rows.  The two batch rows never meet
inside the layer stack, so they could run as two chains of B = 1 kernels on two streams, each filling the partial last rounds
of the other's launches (4200 attention workgroups on 512 slots, the GEMMs' 128 x 128 tail launches).  This script times
LAYERS chained layer-calls in three forms, interleaved, ROUNDS times:
  b2        one stream, B = 2 kernels (what the product does for the layers that have no control partner)
  split     two streams, B = 1 kernels each on its own half of every tensor
  split_1s  the B = 1 kernels on ONE stream (what the smaller launches cost by themselves)
and checks that the halves' results equal the B = 2 results bit for bit.
usage: python tools/cfg_split_probe.py [layers] [rounds]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from landiff_amd import ops  # noqa: E402

LAYERS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda"
torch.manual_seed(0)
B, N, D, H = 2, 17776, 1920, 30
M, Npad = B * N, (N + 127) // 128 * 128
BF = torch.bfloat16


def rnd(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).to(BF)


x = rnd(M, D)
gate = rnd(B, 12 * D, sc=0.1)
ln = tuple(rnd(64) for _ in range(4))
wq, bq = rnd(3 * D, D, sc=0.02), rnd(3 * D, sc=0.02)
w1, b1 = rnd(D, D, sc=0.02), rnd(D, sc=0.02)
w2, b2 = rnd(4 * D, D, sc=0.02), rnd(4 * D, sc=0.02)
w3, b3 = rnd(D, 4 * D, sc=0.02), rnd(D, sc=0.02)


class Work:
    def __init__(self):
        self.q = torch.zeros(B, H, Npad, 64, device=dev, dtype=BF)
        self.k = torch.zeros_like(self.q)
        self.vt = torch.zeros(B, H, 64, Npad, device=dev, dtype=BF)
        self.att = torch.empty(B, N, D, device=dev, dtype=BF)
        self.h = torch.empty(M, D, device=dev, dtype=BF)
        self.h2 = torch.empty(M, D, device=dev, dtype=BF)
        self.mlp = torch.empty(M, 4 * D, device=dev, dtype=BF)


def layer(ws: Work, src, dst, b0: int, nb: int):
    """One layer-call on batch rows [b0, b0 + nb): src -> dst ([M, D] buffers)."""
    r = slice(b0 * N, (b0 + nb) * N)
    bs = slice(b0, b0 + nb)
    g = gate[bs]
    ops.gemm_qkv_heads(src[r], wq, bq, ws.q[bs], ws.k[bs], ws.vt[bs], nb, N, H, Npad, ln)
    ops.attn_fwd(ws.q[bs], ws.k[bs], ws.vt[bs], ws.att[bs], N, N, 0.125)
    ops.gemm(ws.att[bs].reshape(nb * N, D), w1, out=ws.h[r], bias=b1, resid=src[r], gate=g, gate_bstride=12 * D, gate_off_img=2 * D,
             gate_off_txt=8 * D, rows_per_batch=N, text_len=226)
    ops.gemm(ws.h[r], w2, out=ws.mlp[r], bias=b2, act="gelu_tanh")
    ops.gemm(ws.mlp[r], w3, out=dst[r], bias=b3, resid=ws.h[r], gate=g, gate_bstride=12 * D, gate_off_img=5 * D, gate_off_txt=11 * D,
             rows_per_batch=N, text_len=226)


def chain(ws, b0, nb, out_a, out_b):
    src = x
    for i in range(LAYERS):
        dst = out_a if i % 2 == 0 else out_b
        layer(ws, src, dst, b0, nb)
        src = dst
    return src


ws = Work()
oa, ob = torch.empty(M, D, device=dev, dtype=BF), torch.empty(M, D, device=dev, dtype=BF)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run_b2():
    return chain(ws, 0, 2, oa, ob)


def run_split():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        chain(ws, 0, 1, oa, ob)
    with torch.cuda.stream(s2):
        r = chain(ws, 1, 1, oa, ob)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return r


def run_split_1s():
    chain(ws, 0, 1, oa, ob)
    return chain(ws, 1, 1, oa, ob)


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


ref = run_b2().clone()
torch.cuda.synchronize()
same = bool(torch.equal(run_split(), ref))
torch.cuda.synchronize()
print(f"layers {LAYERS}: split halves bit-identical to the B=2 chain: {same}")
forms = {"b2": run_b2, "split": run_split, "split_1s": run_split_1s}
acc = {k: [] for k in forms}
for _ in range(2):
    for fn in forms.values():
        fn()
for _ in range(ROUNDS):
    for name, fn in forms.items():
        acc[name].append(timed(fn) / LAYERS)
for name, v in acc.items():
    print(f"{name:9s} {min(v):.3f} ms per layer-call (min of {ROUNDS}), median {sorted(v)[len(v) // 2]:.3f}")
