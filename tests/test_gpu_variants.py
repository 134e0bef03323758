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
# (the last three: more tiles than CUs -> persistent workgroups walk several tiles, with 1, 2 and 3 (odd) K-tiles per tile)
for (M, N, K) in [(256, 256, 128), (300, 200, 256), (1000, 1920, 384), (5000, 512, 1920), (70000, 256, 128), (70000, 512, 64), (66000, 264, 192),
                  (70000, 520, 256), (3000, 1000, 512)]:       # 4 and 8 K-tiles: the shapes the software-pipelined loop takes (even, >= 4)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    a[:, 0] += torch.arange(M, device=dev).to(torch.bfloat16) * 0.01          # transposition-detecting
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    ref = a.float() @ w.float().t() + bias.float()
    for kw, r in ((dict(bias=bias), ref), (dict(bias=bias, act="gelu_tanh"), torch.nn.functional.gelu(ref.to(torch.bfloat16).float(), approximate="tanh")),
                  (dict(bias=bias, resid=res), res.float() + ref.to(torch.bfloat16).float())):
        out = ops.gemm(a, w, **kw)
        worst = max(worst, (torch.nan_to_num((out.float() - r).abs(), nan=1e9).max() / (r.abs().max() + 1e-6)).item())
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
    worst = max(worst, torch.nan_to_num((out.float() - ref).abs(), nan=1e9).max().item())      # (max(x, nan) would drop a NaN)
print("WORST", worst)
''' % ROOT

LN_SNIPPET = r"""
import sys, torch
sys.path.insert(0, %r)
from landiff_amd import ops
torch.manual_seed(0)
dev, BF = "cuda", torch.bfloat16
worst = 0.0
def ref(x, w, b, ada, d, rpb, tl, eps=1e-5):          # torch restatement: LayerNorm in fp32 -> bf16, modulate() in bf16 ops
    y = torch.nn.functional.layer_norm(x.float(), (d,), w.float(), b.float(), eps).to(BF)
    r = torch.arange(x.shape[0], device=dev)
    bb = r // rpb; txt = (r - bb * rpb) < tl
    sh = torch.where(txt[:, None], ada[bb, 6 * d:7 * d], ada[bb, 0:d]); sc = torch.where(txt[:, None], ada[bb, 7 * d:8 * d], ada[bb, d:2 * d])
    return y * (1 + sc) + sh
# (rows, D, rows_per_batch, text_len): even / odd row counts, region and batch boundaries inside a row pair, every chunk count
for rows, d, rpb, tl in [(666, 1920, 333, 7), (667, 1920, 334, 6), (5, 1920, 1 << 30, 0), (64, 128, 32, 3), (33, 768, 17, 4), (40, 1536, 20, 20)]:
    x = (torch.randn(rows, d, device=dev) * 2 + 0.5).to(BF)
    w = (1 + 0.1 * torch.randn(d, device=dev)).to(BF); b = (0.1 * torch.randn(d, device=dev)).to(BF)
    ada = (0.3 * torch.randn((rows + rpb - 1) // rpb, 12 * d, device=dev)).to(BF)
    o = torch.empty_like(x)
    ops.layernorm(x, w, b, o, 1e-5, mod=ada, mod_bstride=12 * d, rows_per_batch=rpb, text_len=tl, shift_img=0, scale_img=d, shift_txt=6 * d, scale_txt=7 * d)
    r = ref(x, w, b, ada, d, rpb, tl).float()
    worst = max(worst, torch.nan_to_num((o.float() - r).abs() / (r.abs() + 1.0), nan=1e9).max().item())
print("WORST", worst)
""" % ROOT

LLM_SNIPPET = r"""
import sys, hashlib, torch
sys.path.insert(0, %r)
from landiff_amd.config import PipelineConfig
from landiff_amd.weights import init_pipeline_state
from landiff_amd.llm import LLMRunner
cfg = PipelineConfig.tiny(num_steps=3).check()
c = cfg.llm
st = init_pipeline_state(cfg, seed=1234, parts=["llm"])
dev = torch.device("cuda:0")
run = LLMRunner(st["llm"], c, dev, max_text=32, max_frames=c.segment_length)
text = torch.randn(6, c.text_dim, generator=torch.Generator().manual_seed(4))
h = hashlib.sha256()
for seed, graph in ((21, False), (22, True)):
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    log = []
    codes = run.sample(text, num_frames=c.segment_length, guidance_scale=7.5, generator=gen, logits_log=None if graph else log, use_graph=graph)
    h.update(codes.cpu().numpy().tobytes())
    if log: h.update(torch.cat(log, 0).cpu().numpy().tobytes())
print("HASH", h.hexdigest())
""" % ROOT

GN_SNIPPET = r"""
import sys, hashlib, torch
sys.path.insert(0, %r)
from landiff_amd import ops
torch.manual_seed(0)
dev, BF = "cuda", torch.bfloat16
h = hashlib.sha256()
worst = 0.0
# (F, T, H, W, C, G, zq shape or None, pads, swish): odd T with zq (first-frame rule), even T, no zq, narrow groups, C = 768 (flat kernel)
for F, T, H, W, C, G, zs, pads, swish in [(1, 3, 10, 12, 128, 32, (2, 5, 6), (2, 1, 1), True), (2, 2, 6, 9, 512, 32, (1, 3, 3), (0, 1, 1), True),
                                           (3, 1, 7, 5, 64, 32, None, (0, 1, 1), True), (1, 5, 8, 8, 256, 32, (3, 2, 2), (2, 1, 1), False),
                                           (1, 1, 4, 6, 768, 32, None, (0, 0, 0), True)]:
    x = (torch.randn(F * T, H, W, C, device=dev) * 1.5 + 0.3).to(BF)
    gamma = (1 + 0.2 * torch.randn(C, device=dev)).to(BF); beta = (0.2 * torch.randn(C, device=dev)).to(BF)
    zy = zb = None
    if zs:
        zy = torch.randn(*zs, C, device=dev).to(BF); zb = torch.randn(*zs, C, device=dev).to(BF)
    stats = torch.zeros(F, G, 2, device=dev, dtype=torch.float64)
    ops.groupnorm_stats(x, stats, F, T * H * W, C, G)
    tp, hp, wp = pads
    out = torch.zeros(F, T + tp, H + 2 * hp, W + 2 * wp, C, device=dev, dtype=BF)
    ops.groupnorm_apply(x, out, stats, gamma, beta, F, T, H, W, C, G, zy=zy, zb=zb, zshape=zs or (1, 1, 1), tpad=tp, hpad=hp, wpad=wp, swish=swish)
    h.update(out.cpu().view(torch.int16).numpy().tobytes())
    # torch restatement (GroupNorm in fp32 -> bf16, SpatialNorm terms and swish in bf16 ops)
    xf = x.float().view(F, T, H, W, G, C // G)
    mu = xf.mean(dim=(1, 2, 3, 5), keepdim=True); var = xf.var(dim=(1, 2, 3, 5), unbiased=False, keepdim=True)
    y = (((xf - mu) * torch.rsqrt(var + 1e-6)).view(F, T, H, W, C) * gamma.float() + beta.float()).to(BF)
    if zs:
        ti = torch.tensor([0 if (T > 1 and (T & 1) and t == 0) else (1 + ((t - 1) * (zs[0] - 1)) // (T - 1) if (T > 1 and (T & 1)) else (t * zs[0]) // T) for t in range(T)], device=dev)
        hi = (torch.arange(H, device=dev) * zs[1]) // H; wi = (torch.arange(W, device=dev) * zs[2]) // W
        y = y * zy[ti][:, hi][:, :, wi] + zb[ti][:, hi][:, :, wi]
    if swish:
        y = y * torch.sigmoid(y)
    got = out[:, tp:, hp:hp + H, wp:wp + W].float()
    worst = max(worst, torch.nan_to_num((got - y.float()).abs() / (y.float().abs() + 1.0), nan=1e9).max().item())
print("HASH", h.hexdigest())
print("WORST", worst)
""" % ROOT

ATTN_HASH_SNIPPET = r"""
import sys, hashlib, torch
sys.path.insert(0, %r)
from landiff_amd import ops
torch.manual_seed(1)
h = hashlib.sha256()
# ragged last tile / exact tiles / a key count whose last HALF tile is empty / 6 tiles (the minimum), odd head counts
for (B, H, N) in [(1, 3, 1122), (2, 1, 1408), (1, 2, 2065), (1, 1, 384), (1, 2, 777)]:
    q = torch.randn(B, H, N, 64).cuda().bfloat16(); k = torch.randn(B, H, N, 64).cuda().bfloat16(); v = torch.randn(B, H, N, 64).cuda().bfloat16()
    Npad = (N + 127) // 128 * 128
    def pack(x):
        o = torch.zeros(B, H, Npad, 64, device="cuda", dtype=x.dtype); o[:, :, :N] = x; return o
    out = torch.zeros(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
    ops.attn_fwd(pack(q), pack(k), pack(v).transpose(2, 3).contiguous(), out, N, N, 0.125)
    h.update(out.cpu().view(torch.int16).numpy().tobytes())
print("HASH", h.hexdigest())
""" % ROOT


def variants_lib() -> str:
    """Path of the variants build of the library (the measured alternatives live only there); __graft_entry__.build() makes it, and
    a tree that lacks it gets it built here (hipcc is on the GPU box too; a few minutes, once)."""
    path = os.path.join(ROOT, "landiff_amd", "variants", "liblandiff_hip_variants.so")
    if not os.path.exists(path):
        e = dict(os.environ, LD_BUILD_VARIANTS="1")
        subprocess.run(["bash", os.path.join(ROOT, "landiff_amd", "csrc", "build.sh")], env=e, check=True, timeout=3000)
    return path


def _needs_variants(env) -> bool:
    return (env.get("LD_GEMM_TILE") == "11" or env.get("LD_GEMM_SP") == "1" or env.get("LD_ATTN_VARIANT") == "8"
            or env.get("LD_ATTN_Q128", "0") != "0")


def _env(env):
    e = dict(os.environ); e.update(env)
    if _needs_variants(env):
        e["LANDIFF_HIP_LIB"] = variants_lib()
    return e


def _run(snippet, env):
    e = _env(env)
    r = subprocess.run([sys.executable, "-c", snippet], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("WORST")][-1]
    return float(line.split()[1])


@pytest.mark.parametrize("env", [{"LD_GEMM_TILE": "11"}, {"LD_GEMM_TILE": "3", "LD_GEMM_8P": "0", "LD_GEMM_MSPLIT": "0"},
                                 {"LD_GEMM_TILE": "1"}, {"LD_GEMM_TILE": "3", "LD_GEMM_8P": "0", "LD_GEMM_M16": "0"},
                                 {"LD_GEMM_TILE": "1", "LD_GEMM_M16": "0"},
                                 # the 8-phase loop (round-3 default for large problems) forced onto every size, with and
                                 # without the M-split tail launch, and with one workgroup per tile instead of persistent tiles
                                 {"LD_GEMM_TILE": "8"}, {"LD_GEMM_TILE": "8", "LD_GEMM_MSPLIT": "0"}, {"LD_GEMM_TILE": "8", "LD_GEMM_PERSIST": "0"},
                                 # the software-pipelined loop (one barrier per K-tile, fragment reads under the wave's own MFMAs)
                                 {"LD_GEMM_TILE": "8", "LD_GEMM_SP": "1"}, {"LD_GEMM_TILE": "8", "LD_GEMM_SP": "1", "LD_GEMM_PERSIST": "0"}])
def test_gemm_main_loop_variants(cuda, env):
    assert _run(GEMM_SNIPPET, env) < 1e-2


@pytest.mark.parametrize("env", [{"LD_ATTN_SAFE": "1"}, {"LD_ATTN_Q64": "0"}, {"LD_ATTN_Q64": "0", "LD_ATTN_SAFE": "1"}, {"LD_ATTN_Q64": "0", "LD_ATTN_NW": "8"}, {"LD_ATTN_Q64": "0", "LD_ATTN_NW": "8", "LD_ATTN_SAFE": "1"}, {"LD_ATTN_VARIANT": "9"}, {"LD_ATTN_VARIANT": "1"}, {"LD_ATTN_VARIANT": "4"},
                                 {"LD_ATTN_VARIANT": "8"}, {"LD_ATTN_VARIANT": "8", "LD_ATTN_SAFE": "1"}, {"LD_ATTN_VARIANT": "8", "LD_ATTN_NW": "8"},
                                 {"LD_ATTN_Q64": "0", "LD_ATTN_MSUM": "0"}, {"LD_ATTN_Q64": "0", "LD_ATTN_MSUM": "0", "LD_ATTN_SAFE": "1"},
                                 # the 128-query-row / one-wave-per-SIMD tile forced onto the small test problems (LD_ATTN_Q128=2), its
                                 # running-max fallback, and the other two exp2 splits
                                 {"LD_ATTN_Q128": "2"}, {"LD_ATTN_Q128": "2", "LD_ATTN_SAFE": "1"},
                                 {"LD_ATTN_Q128": "2", "LD_ATTN_NPRE": "36"}, {"LD_ATTN_Q128": "2", "LD_ATTN_NPRE": "52"}])
def test_attention_variants(cuda, env):
    assert _run(ATTN_SNIPPET, env) < 2e-2


@pytest.mark.parametrize("env", [{}, {"LD_LN_FAST": "0"}])
def test_layernorm_modulate_forms(cuda, env):
    """The DiT's LayerNorm + modulate: the two-rows-per-wave pair kernel (default) and the general kernel (LD_LN_FAST=0)
    against a torch restatement -- at most one bf16 step apart (the kernels round where the reference's bf16 ops round)."""
    assert _run(LN_SNIPPET, env) < 2 ** -7


def test_attention_wave_tiles_bit_identical(cuda):
    """The 64-query-row wave tile (ld_attn_q64), the 32-row one (LD_ATTN_Q64=0) and the 128-row one-wave-per-SIMD tile
    (LD_ATTN_Q128=2: ld_attn_q128.hip, every MFMA / exp2 / pack an asm statement over fixed accumulator registers, three exp2
    splits) do the same per-lane arithmetic in the same order -- outputs equal bit for bit, whatever the tail of the key axis
    looks like."""
    outs = []
    for env in ({"LD_ATTN_Q128": "0"}, {"LD_ATTN_Q64": "0", "LD_ATTN_Q128": "0"}, {"LD_ATTN_Q128": "2"},
                {"LD_ATTN_Q128": "2", "LD_ATTN_NPRE": "36"}, {"LD_ATTN_Q128": "2", "LD_ATTN_NPRE": "52"}):
        e = _env(env)                       # (the first two run on the shipped library, the 128-row tile on the variants build)
        r = subprocess.run([sys.executable, "-c", ATTN_HASH_SNIPPET], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][-1])
    assert all(o == outs[0] for o in outs[1:]), outs


def test_variants_build_suite(cuda):
    """tests/variants/variant_cases.py in a child pytest process on the variants build of the library: the 128-query-row attention
    tile (15 shapes, overflow fallback, bit-identity with the 64-row tile) and the chained / persistent forms of the decode step
    (bit-identical ids, logits and KV cache, tiny and full size)."""
    e = dict(os.environ, LANDIFF_HIP_LIB=variants_lib())
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "variants", "variant_cases.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=e, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]


def test_groupnorm_apply_forms(cuda):
    """GroupNorm / SpatialNorm / swish: the row-per-workgroup kernel (default where C / 8 divides 256) and the flat kernel
    (LD_GN_ROWS=0) are the same arithmetic -- equal bit for bit -- and within two bf16 steps of a torch restatement."""
    outs = []
    for env in ({}, {"LD_GN_ROWS": "0"}):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", GN_SNIPPET], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][-1])
        assert float([l for l in r.stdout.splitlines() if l.startswith("WORST")][-1].split()[1]) < 2 ** -6
    assert outs[0] == outs[1]


def test_llm_sampling_tail_forms(cuda):
    """AR decode with the step's tail in one launch (default) or as torch.multinomial + ld_llm_decode_advance + ld_llm_embed
    (LD_LLM_FUSED_TAIL=0): the same tokens and CFG logits from the same seeds, eager and graph-replayed."""
    outs = []
    for env in ({}, {"LD_LLM_FUSED_TAIL": "0"}):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", LLM_SNIPPET], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][-1])
    assert outs[0] == outs[1]


def test_gemv_streaming_loop_variant(cuda):
    """LD_GEMV_MODE=1 (the wave-per-row streaming kernel for every shape) against the same references as the default."""
    e = dict(os.environ); e["LD_GEMV_MODE"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_gemv.py"), "-x", "-q", "-m", "gpu"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_rccl_path_single_rank(cuda, launcher):
    """bench.py launched the way the driver launches the N-GPU runs (torch.distributed.run, one rank per GPU) and the plain way
    (`python bench.py --gpus N`, which starts its own ranks as child processes), with one rank and LD_BENCH_FORCE_DIST=1: RCCL
    init, barrier, all_gather of the uint8 frames, all_reduce(MAX) of the time."""
    import json
    e = dict(os.environ); e["LD_BENCH_FORCE_DIST"] = "1"; e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LD_BENCH_DRY_SPAWN"):
        e.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--tiny", "--no-cpu-baseline"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29517"] + tail
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if launcher == "self":
        assert "starting 1 rank(s)" in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["unit"] == "frames/s"
    # the line's contract: the driver's keys, the roofline object of the dominant kernel, the calibration of the box, per-rank reports
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_stages", "calibration", "per_rank", "n_ranks_seen"):
        assert k in res, k
    assert res["n_ranks_seen"] == 1                      # RCCL's own count of the job
    assert res["scaling"] == "weak" and res["higher_is_better"] is True and res["vs_baseline"] is None and res["data"] == "synthetic"
    assert "workload" in res["config"] and "model" not in res["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms"):
        assert k in res["roofline"], k
    assert res["roofline"]["achieved"] > 0 and res["roofline"]["launches"] > 0      # (tiny config: no launch runs alone -> all launches, labelled)
    assert res["roofline"]["attn_policy"] == {"mode": "auto", "layers_on_exact_form": 0}
    assert res["roofline"]["bound"] == "mfma" and res["roofline"]["frac"] == pytest.approx(res["roofline"]["achieved"] / res["roofline"]["peak"], abs=1e-3)
    assert res["calibration"]["mfma_tflops"] > 100 and res["calibration"]["hbm_gbs"] > 500
    assert len(res["per_rank"]["frames_per_s"]) == 1
    # every rank calibrates its own GPU (round 6): a future scaling curve can be read net of the pool's GPU-to-GPU spread
    pr = res["per_rank"]
    assert len(pr["calibration"]) == 1 and pr["calibration"][0]["mfma_tflops"] == res["calibration"]["mfma_tflops"]
    assert pr["calibration"][0]["hbm_gbs"] > 500 and "timed_region_sclk_mhz" in pr["calibration"][0]
    assert pr["frames_per_s_normalised"] == pytest.approx(pr["frames_per_s"], rel=1e-3)      # one rank: the mean is its own rate
