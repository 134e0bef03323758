"""The tuning-knob kernel variants (selected by environment variables that the library reads once per process) stay
correct: each is run in a fresh process against a torch fp32 reference."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GEMM_SNIPPET = r'''
import sys, torch
sys.path.insert(0, %r)
from landiff_amd import ops
torch.manual_seed(0)
dev = "cuda"
worst = 0.0
for (M, N, K) in [(256, 256, 128), (300, 200, 256), (1000, 1920, 384), (5000, 512, 1920), (70000, 256, 128)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    a[:, 0] += torch.arange(M, device=dev).to(torch.bfloat16) * 0.01          # transposition-detecting
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    ref = a.float() @ w.float().t() + bias.float()
    for kw, r in ((dict(bias=bias), ref), (dict(bias=bias, act="gelu_tanh"), torch.nn.functional.gelu(ref.to(torch.bfloat16).float(), approximate="tanh")),
                  (dict(bias=bias, resid=res), res.float() + ref.to(torch.bfloat16).float())):
        out = ops.gemm(a, w, **kw)
        worst = max(worst, ((out.float() - r).abs().max() / (r.abs().max() + 1e-6)).item())
print("WORST", worst)
''' % ROOT

ATTN_SNIPPET = r'''
import sys, torch
sys.path.insert(0, %r)
from landiff_amd import ops
torch.manual_seed(0)
worst = 0.0
for (B, H, N) in [(1, 2, 1122), (2, 1, 1400), (1, 1, 2175)]:
    q = torch.randn(B, H, N, 64).cuda().bfloat16(); k = torch.randn(B, H, N, 64).cuda().bfloat16(); v = torch.randn(B, H, N, 64).cuda().bfloat16()
    Npad = (N + 127) // 128 * 128
    def pack(x):
        o = torch.zeros(B, H, Npad, 64, device="cuda", dtype=x.dtype); o[:, :, :N] = x; return o
    out = torch.zeros(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
    ops.attn_fwd(pack(q), pack(k), pack(v).transpose(2, 3).contiguous(), out, N, N, 0.125)
    s = (q.float() @ k.float().transpose(-1, -2)) * 0.125
    ref = (torch.softmax(s, -1) @ v.float()).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    worst = max(worst, (out.float() - ref).abs().max().item())
print("WORST", worst)
''' % ROOT


def _run(snippet, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", snippet], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("WORST")][-1]
    return float(line.split()[1])


@pytest.mark.parametrize("env", [{"LD_GEMM_TILE": "11"}, {"LD_GEMM_TILE": "3", "LD_GEMM_MSPLIT": "0"},
                                 {"LD_GEMM_TILE": "1"}, {"LD_GEMM_TILE": "3", "LD_GEMM_M16": "0"}, {"LD_GEMM_TILE": "1", "LD_GEMM_M16": "0"}])
def test_gemm_main_loop_variants(cuda, env):
    assert _run(GEMM_SNIPPET, env) < 1e-2


@pytest.mark.parametrize("env", [{"LD_ATTN_SAFE": "1"}, {"LD_ATTN_NW": "8"}, {"LD_ATTN_NW": "8", "LD_ATTN_SAFE": "1"}, {"LD_ATTN_VARIANT": "9"}, {"LD_ATTN_VARIANT": "1"}, {"LD_ATTN_VARIANT": "4"},
                                 {"LD_ATTN_VARIANT": "8"}, {"LD_ATTN_VARIANT": "8", "LD_ATTN_SAFE": "1"}, {"LD_ATTN_VARIANT": "8", "LD_ATTN_NW": "8"},
                                 {"LD_ATTN_MSUM": "0"}, {"LD_ATTN_MSUM": "0", "LD_ATTN_SAFE": "1"}])
def test_attention_variants(cuda, env):
    assert _run(ATTN_SNIPPET, env) < 2e-2


def test_gemv_streaming_loop_variant(cuda):
    """LD_GEMV_MODE=1 (the wave-per-row streaming kernel for every shape) against the same references as the default."""
    e = dict(os.environ); e["LD_GEMV_MODE"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_gemv.py"), "-x", "-q", "-m", "gpu"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_rccl_path_single_rank(cuda):
    """bench.py launched the way the driver launches the N-GPU runs (torch.distributed.run, one rank per GPU), with one
    rank and LD_BENCH_FORCE_DIST=1: RCCL init, barrier, all_gather of the uint8 frames, all_reduce(MAX) of the time."""
    import json
    e = dict(os.environ); e["LD_BENCH_FORCE_DIST"] = "1"; e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--tiny",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["unit"] == "frames/s"
