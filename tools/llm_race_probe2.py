"""Which operation of the AR prefill goes wrong while DiT-sized attention launches share the GPU?  The prefill's launches are
issued one by one on a side stream (as LLMRunner._prefill does), every output is snapshotted, and the snapshots of a run under
attention load are compared with those of a quiet run."""
import os, sys, torch
sys.path.insert(0, ".")
from landiff_amd import ops
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.weights import init_state, llm_spec
dev = torch.device("cuda:0")
cfg = LLMConfig()
run = LLMRunner(init_state(llm_spec(cfg), 1, dtype=torch.bfloat16, device=dev), cfg, dev)
text = torch.randn(64, cfg.text_dim, device=dev)
BF = torch.bfloat16
B, H, N = 2, 30, 17776; Npad = (N + 127) // 128 * 128
q = torch.zeros(B, H, Npad, 64, device=dev, dtype=BF); k = torch.zeros_like(q); vt = torch.zeros(B, H, 64, Npad, device=dev, dtype=BF)
q[:, :, :N] = torch.randn(B, H, N, 64, device=dev).to(BF); k[:, :, :N] = torch.randn(B, H, N, 64, device=dev).to(BF)
vt[:, :, :, :N] = torch.randn(B, H, 64, N, device=dev).to(BF)
ao = torch.zeros(B, N, H * 64, device=dev, dtype=BF)
load_stream, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
feats = run.prefix_features(text, 13.0, 0.1)
torch.cuda.synchronize()

def prefill(loaded, nl=int(os.environ.get("LAYERS", "6"))):
    snaps = []
    c = cfg
    Bm, m, d = feats.shape
    M = Bm * m
    if loaded:
        with torch.cuda.stream(load_stream):
            for _ in range(int(os.environ.get("NLOAD", "40"))):
                ops.attn_fwd(q, k, vt, ao, N, N, 0.125)
    with torch.cuda.stream(side):
        x = feats.reshape(M, d).contiguous().clone()
        xn = torch.empty_like(x); qkv = torch.empty(M, 3 * d, device=dev, dtype=BF); qr = torch.empty(M, d, device=dev, dtype=BF)
        att = torch.empty(M, d, device=dev, dtype=BF); h3 = torch.empty(M, c.mlp, device=dev, dtype=BF); gate = torch.empty(M, c.mlp, device=dev, dtype=BF)
        run.pos0.zero_()
        snap = lambda name, t: snaps.append((name, t.clone()))
        for i, w in enumerate(run.blocks[:nl]):
            ops.rmsnorm(x, w["n0"], xn, c.rms_eps); snap(f"L{i} rmsnorm0", xn)
            ops.gemm(xn, w["wqkv"], out=qkv); snap(f"L{i} gemm qkv", qkv)
            ops.llm_rope_append(qkv, run.cos, run.sin, run.pos0, qr, run.kc[i], run.vc[i], Bm, m, c.heads, run.Lmax); snap(f"L{i} rope q", qr)
            snap(f"L{i} kcache", run.kc[i][:, :m])
            ops.llm_kv_attn(qr, run.kc[i], run.vc[i], run.pos0, att, Bm, m, c.heads, run.Lmax); snap(f"L{i} kv_attn", att)
            ops.gemm(att, w["wo"], out=x, resid=x); snap(f"L{i} gemm wo+res", x)
            ops.rmsnorm(x, w["n1"], xn, c.rms_eps); snap(f"L{i} rmsnorm1", xn)
            ops.gemm(xn, w["w3"], out=h3); snap(f"L{i} gemm w3", h3)
            ops.gemm(xn, w["w1"], out=gate, act="gelu_tanh", mul=h3); snap(f"L{i} gemm w1*w3", gate)
            ops.gemm(gate, w["w2"], out=x, resid=x); snap(f"L{i} gemm w2+res", x)
    torch.cuda.synchronize()
    return snaps

ref = prefill(False)
again = prefill(False)
print("quiet vs quiet:", "equal" if all(torch.equal(a[1], b[1]) for a, b in zip(ref, again)) else "DIFFER", flush=True)
for rep in range(3):
    got = prefill(True)
    bad = [(n, (a.float() - b.float()).abs().max().item(), (a != b).float().mean().item(), (a != b)) for (n, a), (_, b) in zip(got, ref) if not torch.equal(a, b)]
    print(f"loaded run {rep}: {len(bad)} of {len(got)} snapshots differ; first: " + ("; ".join(f"{n} (max {mx:.3g}, {fr:.2e} of elements)" for n, mx, fr, _ in bad[:4]) if bad else "none"), flush=True)
    if bad:
        n, mx, fr, mask = bad[0]
        idx = torch.nonzero(mask)
        rows = idx[:, 0].unique()
        print(f"   {n}: {idx.shape[0]} elements in {rows.numel()} rows; rows {rows[:12].tolist()} ...; columns of the first row {idx[idx[:, 0] == rows[0]][:16, -1].tolist()}", flush=True)
