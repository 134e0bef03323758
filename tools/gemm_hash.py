"""sha256 of the outputs of the four DiT GEMMs (real epilogues, seeded operands) + a ragged-edge case: run under two settings of
the library's knobs and diff the lines -- bit-identity check of a new main loop / epilogue against the shipped one."""
import hashlib, sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
M, D = 35552, 1920
dev = "cuda"
torch.manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
def h(t): return hashlib.sha256(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]
x = rnd(M, D); x4 = rnd(M, 4 * D); gate = rnd(2, 12 * D); resid = rnd(M, D); add2 = rnd(M, D)
B, Ntok, H, Npad = 2, M // 2, 30, (M // 2 + 127) // 128 * 128
q = torch.zeros(B, H, Npad, 64, device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros(B, H, 64, Npad, device=dev, dtype=torch.bfloat16)
ln = tuple(rnd(64) for _ in range(4))
ops.gemm_qkv_heads(x, rnd(3 * D, D, sc=0.02), rnd(3 * D), q, k, vt, B, Ntok, H, Npad, ln)
print("qkv ", h(q), h(k), h(vt))
o = ops.gemm(x, rnd(D, D, sc=0.02), bias=rnd(D), resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=2 * D, gate_off_txt=8 * D, rows_per_batch=M // 2, text_len=226)
print("proj", h(o))
o = ops.gemm(x, rnd(4 * D, D, sc=0.02), bias=rnd(4 * D), act="gelu_tanh")
print("ff1 ", h(o))
o = ops.gemm(x4, rnd(D, 4 * D, sc=0.02), bias=rnd(D), resid=resid, gate=gate, gate_bstride=12 * D, gate_off_img=5 * D, gate_off_txt=11 * D, rows_per_batch=M // 2, text_len=226, add2=add2)
print("ff2 ", h(o))
o = ops.gemm(x, rnd(D, D, sc=0.02))
print("zero", h(o))
# ragged: M, N not multiples of the tile, N % 8 == 0
Mr, Nr, Kr = 70003, 1000, 1024
o = ops.gemm(rnd(Mr, Kr), rnd(Nr, Kr, sc=0.03), bias=rnd(Nr), act="gelu_tanh")
print("rag ", h(o))
