"""Why do the first chunk's (9 frames) 480x720 convolutions run 2.4-16x slower per row than the 8-frame chunks'
(profiles/r03_vae_conv_shapes.txt)?  Times ld_conv_cl_bf16 in isolation, warm, per frame count."""
import sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
dev = "cuda"
def run(T, H, W, Cin, Cout, kT, reps=5):
    xp = torch.zeros(T + kT - 1, H + 2, W + 2, Cin, device=dev, dtype=torch.bfloat16)
    xp[:, 1:-1, 1:-1] = torch.randn(T + kT - 1, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, kT, 3, 3, Cin, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(T * H * W, Cout, device=dev, dtype=torch.bfloat16)
    for _ in range(2): ops.conv_cl(xp, w, T, H, W, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv_cl(xp, w, T, H, W, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    fl = 2.0 * T * H * W * Cout * kT * 9 * Cin
    ms = sorted(ts)[len(ts) // 2]
    print(f"T{T} {H}x{W} Cin{Cin} Cout{Cout} kT{kT}: {ms:8.3f} ms  {fl / ms / 1e9:6.0f} TFLOP/s  ({ms / T:.3f} ms per frame)  in {xp.numel() * 2 / 2**30:.2f} GiB", flush=True)
    del xp, out
    torch.cuda.empty_cache()
for T in (7, 8, 9, 10):
    run(T, 480, 720, 128, 128, 3)
for T in (8, 9):
    run(T, 480, 720, 256, 256, 1)
    run(T, 480, 720, 256, 128, 3)
    run(T, 240, 360, 256, 256, 3)
