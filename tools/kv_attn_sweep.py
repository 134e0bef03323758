"""Decode-step attention launch (ld_llm_kv_attn: fused RoPE + append + key-split attention + in-launch merge) at B 2, H 16, D 128
over context lengths, for a forced number of splits: 24 launches on 24 distinct KV caches (no cache reuse between launches, like
the 24 blocks of a decode step) replayed from a HIP graph.  One child process per split rule (LD_KV_SPLIT_T is read once).
  python tools/kv_attn_sweep.py            -> table us per launch: rows = L, columns = splits in use
"""
import os, subprocess, sys
sys.path.insert(0, ".")
LENS = [70, 100, 128, 160, 200, 256, 320, 384, 512, 640, 768, 900, 1024, 1150, 1313]
RULES = {"1": "100000,0,0", "2": "0,100000,0", "4": "0,0,100000", "8": "0,0,0"}

def child(nsplit_max):
    import torch
    from landiff_amd import ops
    from oracle.llm import rope_table
    dev = "cuda"
    B, H, D, Lmax, NL = 2, 16, 128, 1344, 24
    cos, sin = rope_table(D, Lmax, 10000.0)
    cos, sin = cos.to(dev).contiguous(), sin.to(dev).contiguous()
    kc = [torch.randn(B, Lmax, H, D, device=dev).to(torch.bfloat16) for _ in range(NL)]
    vc = [torch.randn(B, Lmax, H, D, device=dev).to(torch.bfloat16) for _ in range(NL)]
    qkv = torch.randn(B, 3 * H * D, device=dev).to(torch.bfloat16)
    out = torch.empty(B, H * D, device=dev, dtype=torch.bfloat16)
    ws = torch.zeros(B * H * (nsplit_max * 130 + 1), device=dev, dtype=torch.float32)
    pos = torch.zeros(1, device=dev, dtype=torch.int32)
    def run():
        for i in range(NL):
            ops.llm_kv_attn(None, kc[i], vc[i], pos, out, B, 1, H, Lmax, workspace=ws, nsplit=nsplit_max, qkv_fused=qkv, cos_t=cos, sin_t=sin)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            run()
    res = []
    for L in LENS:
        pos.fill_(L - 1)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 / NL * 1e3)
    print("RES " + " ".join(f"{r:.2f}" for r in res), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(int(sys.argv[2]))
else:
    cols = {}
    for nmax in (8, 16):
        for name, rule in RULES.items():
            if nmax == 16 and name != "8":
                continue
            e = dict(os.environ, LD_KV_SPLIT_T=rule)
            r = subprocess.run([sys.executable, __file__, "child", str(nmax)], env=e, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
            if not line:
                print(r.stdout[-500:], r.stderr[-1500:]); continue
            cols[f"{name if nmax == 8 else 16}"] = [float(v) for v in line[0].split()[1:]]
    names = list(cols)
    print("# ld_llm_kv_attn, B 2, H 16, D 128: us per launch (24 launches on 24 caches, graph replay; a forced split count that would put more")
    print("# than 256 keys into a split is raised by the rule itself: those cells repeat the next column)")
    print("L     " + "".join(f"{'ns=' + n:>9s}" for n in names))
    for i, L in enumerate(LENS):
        print(f"{L:5d} " + "".join(f"{cols[n][i]:9.2f}" for n in names))
