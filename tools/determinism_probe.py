"""Run-to-run determinism of the VAE building blocks (same inputs twice, bitwise compare)."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev = "cuda"
torch.manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
def same(a, b): return "bitwise equal" if torch.equal(a, b) else f"DIFF max {(a.float()-b.float()).abs().max().item():.3e} frac {(a!=b).float().mean().item():.2e}"
# conv
T, H, W, Cin, Cout = 3, 16, 24, 128, 128
xp = torch.zeros(T + 2, H + 2, W + 2, Cin, device=dev, dtype=torch.bfloat16); xp[:, 1:-1, 1:-1] = rnd(T + 2, H, W, Cin)
w = rnd(Cout, 3, 3, 3, Cin, sc=0.05); b = rnd(Cout); res = rnd(T * H * W, Cout)
for kw in (dict(bias=b), dict(bias=b, resid=res)):
    o1 = ops.conv_cl(xp, w, T, H, W, **kw); o2 = ops.conv_cl(xp, w, T, H, W, **kw)
    print("conv_cl", sorted(kw), same(o1, o2))
# gemm
a = rnd(5000, 256); w2 = rnd(384, 256, sc=0.05)
print("gemm", same(ops.gemm(a, w2, bias=rnd(384)), ops.gemm(a, w2, bias=rnd(384)) * 0 + ops.gemm(a, w2, bias=torch.zeros(384, device=dev, dtype=torch.bfloat16))) if False else "", end="")
g1 = ops.gemm(a, w2); g2 = ops.gemm(a, w2); print("gemm", same(g1, g2))
# groupnorm
C, G = 128, 32
x = rnd(T * H * W, C)
s1 = torch.empty(1, G, 2, device=dev, dtype=torch.float64); s2 = torch.empty_like(s1)
ops.groupnorm_stats(x, s1, 1, T * H * W, C, G); ops.groupnorm_stats(x, s2, 1, T * H * W, C, G)
print("gn_stats", "bitwise equal" if torch.equal(s1, s2) else f"DIFF rel {((s1-s2).abs()/s1.abs().clamp_min(1e-30)).max().item():.3e}")
gam, bet = rnd(C), rnd(C)
o1 = torch.zeros(T + 2, H + 2, W + 2, C, device=dev, dtype=torch.bfloat16); o2 = torch.zeros_like(o1)
ops.groupnorm_apply(x, o1, s1, gam, bet, 1, T, H, W, C, G, tpad=2, hpad=1, wpad=1)
ops.groupnorm_apply(x, o2, s1, gam, bet, 1, T, H, W, C, G, tpad=2, hpad=1, wpad=1)
print("gn_apply(same stats)", same(o1, o2))
ops.groupnorm_apply(x, o2, s2, gam, bet, 1, T, H, W, C, G, tpad=2, hpad=1, wpad=1)
print("gn_apply(other stats)", same(o1, o2))
