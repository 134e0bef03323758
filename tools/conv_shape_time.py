"""Times one channels-last 3x3x3 convolution shape through ld_conv_cl_bf16 (bias epilogue).
usage: python tools/conv_shape_time.py T H W Cin Cout [reps] [kT]   (LD_GEMM_TILE=1 forces the 128 x 128 two-stage route: the knob is read
once per process, so an A/B is two runs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from landiff_amd import _lib, ops
T, H, W, Cin, Cout = (int(a) for a in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
kT = int(sys.argv[7]) if len(sys.argv) > 7 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
xp = torch.zeros(T + kT - 1, H + 2, W + 2, Cin, device=dev, dtype=torch.bfloat16)
xp[:, 1:-1, 1:-1] = torch.randn(T + kT - 1, H, W, Cin, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(Cout, kT, 3, 3, Cin, device=dev, generator=g) * 0.03).to(torch.bfloat16)
b = torch.randn(Cout, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(T * H * W, Cout, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv_cl(xp, w, T, H, W, out=out, bias=b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv_cl(xp, w, T, H, W, out=out, bias=b)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * T * H * W * Cout * kT * 9 * Cin
print(f"T{T} {H}x{W} Cin{Cin} Cout{Cout}: route {_lib.load().ld_conv_route(T, H, W, Cin, Cout, kT, 3, 3)} k{kT}33 LD_GEMM_TILE={os.environ.get('LD_GEMM_TILE', '-')}: {ms:.3f} ms, {fl / ms / 1e9:.0f} TFLOP/s, checksum {out.float().sum().item():.1f}")
