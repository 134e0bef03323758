import hashlib, sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
for (B, H, N) in [(2, 30, 17776), (1, 3, 3001), (1, 2, 1122), (2, 1, 1400), (1, 1, 2175), (1, 1, 17792), (1, 2, 17700)]:
    g = torch.Generator(device="cuda").manual_seed(N)
    Npad = (N + 127) // 128 * 128
    q = torch.zeros(B, H, Npad, 64, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros(B, H, 64, Npad, device="cuda", dtype=torch.bfloat16)
    q[:, :, :N] = torch.randn(B, H, N, 64, device="cuda", generator=g).to(torch.bfloat16); k[:, :, :N] = torch.randn(B, H, N, 64, device="cuda", generator=g).to(torch.bfloat16)
    vt[:, :, :, :N] = torch.randn(B, H, 64, N, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.zeros(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
    ops.attn_fwd(q, k, vt, out, N, N, 0.125)
    print(B, H, N, hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16])
