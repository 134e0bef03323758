"""Race screen for the round-2 kernels: repeated launches on random shapes, every output compared with a reference
(attention: torch SDPA in fp32; fused qkv GEMM: the two-launch path), plus run-to-run bit-identity."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
torch.manual_seed(1)
dev = "cuda"
g = torch.Generator().manual_seed(7)
bad = 0
for it in range(40):
    B = int(torch.randint(1, 3, (1,), generator=g)); H = int(torch.randint(1, 4, (1,), generator=g))
    N = int(torch.randint(384, 3000, (1,), generator=g))
    Npad = (N + 127) // 128 * 128
    q = torch.zeros(B, H, Npad, 64, device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, Npad, device=dev, dtype=torch.bfloat16)
    q[:, :, :N] = torch.randn(B, H, N, 64, device=dev); k[:, :, :N] = torch.randn(B, H, N, 64, device=dev)
    vt[:, :, :, :N] = torch.randn(B, H, 64, N, device=dev)
    outs = []
    for rep in range(3):
        out = torch.empty(B, N, H * 64, device=dev, dtype=torch.bfloat16)
        ops.attn_fwd(q, k, vt, out, N, N, 0.125)
        outs.append(out)
    ref = torch.nn.functional.scaled_dot_product_attention(q[:, :, :N].float(), k[:, :, :N].float(), vt[:, :, :, :N].float().transpose(2, 3))
    ref = ref.permute(0, 2, 1, 3).reshape(B, N, H * 64)
    err = (outs[0].float() - ref).abs().max().item()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    if err > 2e-2 or not same:
        bad += 1; print(f"ATTN FAIL B={B} H={H} N={N}: err {err:.4f} identical {same}")
print("attention: 40 random shapes x 3 runs,", "all ok" if bad == 0 else f"{bad} failures")
bad = 0
for it in range(30):
    B = int(torch.randint(1, 3, (1,), generator=g)); H = int(torch.randint(1, 7, (1,), generator=g))
    N = int(torch.randint(32, 600, (1,), generator=g)) * 8; K = int(torch.randint(1, 9, (1,), generator=g)) * 64
    Npad = (N + 127) // 128 * 128
    a = torch.randn(B * N, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(3 * H * 64, K, device=dev) * 0.1).to(torch.bfloat16); bias = torch.randn(3 * H * 64, device=dev).to(torch.bfloat16)
    ln = tuple(torch.randn(64, device=dev).to(torch.bfloat16) for _ in range(4))
    mk = lambda: (torch.zeros(B, H, Npad, 64, device=dev, dtype=torch.bfloat16), torch.zeros(B, H, Npad, 64, device=dev, dtype=torch.bfloat16),
                  torch.zeros(B, H, 64, Npad, device=dev, dtype=torch.bfloat16))
    q2, k2, v2 = mk()
    ops.qkv_split(ops.gemm(a, w, bias=bias), q2, k2, v2, B, N, H, Npad, ln=ln, eps=1e-6)
    for rep in range(3):
        q1, k1, v1 = mk()
        ops.gemm_qkv_heads(a, w, bias, q1, k1, v1, B, N, H, Npad, ln, eps=1e-6)
        ok = torch.equal(v1, v2) and all(((x1.float() - x2.float()).abs() <= 2.0 ** -7 * x2.float().abs() + 1e-6).all().item() for x1, x2 in ((q1, q2), (k1, k2)))
        if not ok:
            bad += 1; print(f"QKV FAIL B={B} H={H} N={N} K={K} rep {rep}")
print("fused qkv GEMM: 30 random shapes x 3 runs,", "all ok" if bad == 0 else f"{bad} failures")
