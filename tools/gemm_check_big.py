"""Full-size correctness probe of the DiT GEMM shapes against torch (fp32 accumulate of the same bf16 operands), run twice (race screen)."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
torch.manual_seed(0)
M = 35552
for N, K in ((5760, 1920), (1920, 1920), (7680, 1920), (1920, 7680)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    ref = torch.addmm(bias.float(), a.float(), w.float().t())
    outs = []
    for rep in range(3):
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, out=out, bias=bias)
        outs.append(out)
    err = (outs[0].float() - ref).abs()
    rel = err.max().item() / ref.abs().max().item()
    bad = (err > 0.02 * ref.abs().max()).sum().item()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"N={N} K={K}: max err {err.max().item():.4f} (rel {rel:.2e}), elements off by > 2% of max: {bad}, run-to-run identical: {same}")
