"""Times the TiTok decoder's frame-masked attention (18 768 tokens, 12 heads) and a small unmasked problem: both run the plain kernel of ld_attn.hip."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from landiff_amd import ops, _lib
from landiff_amd.config import TokenizerConfig
from landiff_amd.detokenizer import decoder_frame_ids
cuda = torch.device("cuda")
cfg = TokenizerConfig(); fid = decoder_frame_ids(cfg); N, H = cfg.seq_len, cfg.heads
Npad = (N + 127) // 128 * 128
fq = np.zeros(Npad, np.int32); fq[:N] = fid
fk = np.full(Npad, np.iinfo(np.int32).max, np.int32); fk[:N] = fid
kt = fk.reshape(-1, 64)
q = torch.randn(1, H, Npad, 64, device=cuda).to(torch.bfloat16); k = torch.randn_like(q); vt = torch.randn(1, H, 64, Npad, device=cuda).to(torch.bfloat16)
out = torch.empty(1, N, H * 64, device=cuda, dtype=torch.bfloat16)
a = dict(fid_q=torch.from_numpy(fq).to(cuda), fid_k=torch.from_numpy(fk).to(cuda), kt_min=torch.from_numpy(kt.min(1).copy()).to(cuda), kt_max=torch.from_numpy(kt.max(1).copy()).to(cuda))
for _ in range(3): ops.attn_fwd(q, k, vt, out, N, N, 0.125, **a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.attn_fwd(q, k, vt, out, N, N, 0.125, **a)
e1.record(); torch.cuda.synchronize()
import hashlib
print((_lib.load().ld_attn_last_kernel() or b"").decode(), f"{e0.elapsed_time(e1)/20:.4f} ms", hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12])
# small unmasked shape (plain kernel as well)
B, Hh, n = 2, 8, 600
Np = (n + 127) // 128 * 128
q = torch.randn(B, Hh, Np, 64, device=cuda).to(torch.bfloat16); k = torch.randn_like(q); vt = torch.randn(B, Hh, 64, Np, device=cuda).to(torch.bfloat16)
out = torch.empty(B, n, Hh * 64, device=cuda, dtype=torch.bfloat16)
for _ in range(3): ops.attn_fwd(q, k, vt, out, n, n, 0.125)
torch.cuda.synchronize(); e0.record()
for _ in range(50): ops.attn_fwd(q, k, vt, out, n, n, 0.125)
e1.record(); torch.cuda.synchronize()
print("small:", (_lib.load().ld_attn_last_kernel() or b"").decode(), f"{e0.elapsed_time(e1)/50*1e3:.1f} us")
