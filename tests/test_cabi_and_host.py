"""CPU-side checks: the C-ABI library loads and exports every declared symbol (no compute calls without a GPU),
and the product's host logic (schedule, forced-token schedule, mask tables, key maps) matches the goldens."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(os.path.dirname(__file__), "golden")


def test_library_exports_every_declared_symbol():
    """The shipped library exports exactly what include/landiff_hip.h declares outside its LD_VARIANTS section (= _lib.SIGNATURES);
    the variants build exports those plus the section's entry points (= _lib.VARIANT_SIGNATURES)."""
    import subprocess
    from landiff_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "landiff_hip.h")).read()
    var_sec = re.search(r"#ifdef LD_VARIANTS\n(.*?)#endif /\* LD_VARIANTS \*/", hdr, re.S).group(1)
    names = lambda text: set(re.findall(r"\b(ld_[a-z0-9_]+)\s*\(", text))
    variant_only = names(var_sec)
    declared = names(hdr) - variant_only
    assert len(declared) >= 24 and variant_only == set(_lib.VARIANT_SIGNATURES), variant_only ^ set(_lib.VARIANT_SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/landiff_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    exported = lambda path: set(re.findall(r" T (ld_[a-z0-9_]+)", subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout))
    assert exported(_lib.LIB_PATH) == declared, exported(_lib.LIB_PATH) ^ declared          # nothing undeclared, no variant entry point
    if os.path.exists(_lib.VARIANTS_LIB_PATH):
        assert exported(_lib.VARIANTS_LIB_PATH) == declared | variant_only
    assert lib.ld_version() == _lib.ABI_VERSION == int(re.search(r"#define LD_ABI_VERSION (\d+)", hdr).group(1))
    assert ctypes.sizeof(_lib.Epilogue) == 120          # layout of ld_epilogue_t


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch (negative code + message), so this is safe on CPU."""
    from landiff_amd import _lib
    lib = _lib.load()
    rc = lib.ld_gemm_bf16(None, 0, None, None, 0, 1, 1, 64, None, None)
    assert rc < 0 and b"null" in lib.ld_last_error()
    rc = lib.ld_attn_fwd_bf16(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16),
                              1, 1, 100, 100, 100, 0, 64, 0.125, None, None, None, None, None)
    assert rc < 0 and b"Npad" in lib.ld_last_error()
    # the convolution that also leaves GroupNorm partial sums: the workspace size rule, and refusals before any launch
    assert lib.ld_conv_gn_partials_size(130, 128) == 3 * 32 * 2 and lib.ld_conv_gn_partials_size(64, 512) == 128 * 2
    p16 = ctypes.c_void_p(16)
    rc = lib.ld_conv_cl_bf16_gn(p16, p16, p16, 128, 1, 8, 8, 64, 128, 3, 3, 3, None, None, None)
    assert rc < 0 and b"gn_partials" in lib.ld_last_error()
    rc = lib.ld_conv_cl_bf16_gn(p16, p16, p16, 12, 1, 8, 8, 64, 12, 3, 3, 3, None, p16, None)          # Cout % 8 != 0
    assert rc < 0 and b"multiples of 8" in lib.ld_last_error()
    rc = lib.ld_groupnorm_stats_from_conv(p16, p16, p16, 64, 96, 32, None)                              # C / 4 = 24: not a power of two
    assert rc < 0 and b"unsupported" in lib.ld_last_error()
    rc = lib.ld_groupnorm_stats_from_conv(p16, p16, p16, 64, 64, 32, None)                              # half a quad per group
    assert rc < 0 and b"unsupported" in lib.ld_last_error()


def test_conv_route_keeps_large_inputs_off_the_8_phase_kernel():
    """The 8-phase kernel walks a convolution's padded input through one raw buffer descriptor (2^31 - 1 records, 32-bit byte
    offsets): inputs of 2 GiB or more must take the two-stage kernel (32-bit ELEMENT offsets, 8 GiB), and nothing beyond that may
    launch at all.  ld_conv_route is the launcher's own decision, run dry (no HIP call)."""
    from landiff_amd import _lib
    lib = _lib.load()
    gib = lambda T: (T + 2) * 482 * 722 * 256 * 2 / 2 ** 30
    # the largest shipped VAE convolution on this route: 9 frames at 480 x 720, Cin = Cout = 256 -- 1.83 GiB, 8-phase
    assert gib(9) < 2 and lib.ld_conv_route(9, 480, 720, 256, 256, 3, 3, 3) == 2
    assert gib(10) < 2 and lib.ld_conv_route(10, 480, 720, 256, 256, 3, 3, 3) == 2
    # one more frame per chunk crosses 2 GiB: same tile, two-stage loop
    assert gib(11) > 2 and lib.ld_conv_route(11, 480, 720, 256, 256, 3, 3, 3) == 1
    assert lib.ld_conv_route(45, 480, 720, 256, 256, 3, 3, 3) == 1          # 7.8 GiB: still addressable
    assert lib.ld_conv_route(47, 480, 720, 256, 256, 3, 3, 3) < 0 and b"8 GiB" in lib.ld_last_error()
    # narrow / small problems stay on 128 x 128 tiles whatever their size
    assert lib.ld_conv_route(8, 480, 720, 128, 128, 3, 3, 3) == 0            # (the 512 x 128 tile of the variants build: no gain in the VAE)
    assert lib.ld_conv_route(2, 60, 90, 512, 512, 3, 3, 3) == 0
    assert lib.ld_conv_route(2, 60, 90, 100, 512, 3, 3, 3) < 0               # Cin must be a multiple of 64
    # round 5: a 256-wide tile must be at least 3/4 used (Cout 128 is not: half of its waves would idle), K >= 2048 is long enough
    assert lib.ld_conv_route(8, 480, 720, 256, 128, 3, 3, 3) == 0            # (K = 6912: the 512 x 128 tile measured slower there)
    assert lib.ld_conv_route(8, 480, 720, 256, 256, 1, 3, 3) == 2 and lib.ld_conv_route(8, 240, 360, 256, 256, 1, 3, 3) == 2
    assert lib.ld_conv_route(8, 480, 720, 128, 256, 1, 3, 3) == 0            # K = 1152: short


def test_decode_step_forms_validate_without_gpu():
    """The alternative forms of a decode step's blocks (one persistent launch / dependent launches on two streams) reject null
    pointers and shapes outside their register forms before any HIP call: LD_ERR_INVALID (-1) / LD_ERR_UNSUPPORTED (-3).
    They live in the variants build of the library only (a second handle here: the shipped library stays the one _lib binds)."""
    import subprocess
    from landiff_amd import _lib
    if not os.path.exists(_lib.VARIANTS_LIB_PATH):
        subprocess.run(["bash", os.path.join(ROOT, "landiff_amd", "csrc", "build.sh")], check=True, env=dict(os.environ, LD_BUILD_VARIANTS="1"))
    assert not hasattr(_lib.load(), "ld_llm_decode_blocks_fused")         # the shipped library does not carry them
    lib = ctypes.CDLL(_lib.VARIANTS_LIB_PATH)
    for name, argtypes in _lib.VARIANT_SIGNATURES.items():
        getattr(lib, name).argtypes = argtypes
        getattr(lib, name).restype = ctypes.c_int
    lib.ld_last_error.restype = ctypes.c_char_p
    P = ctypes.c_void_p
    fake = [P(4096)] * 9                                            # pos, x, qkv, att, gate, attn_ws, cos, sin (+1): never dereferenced
    rc = lib.ld_llm_decode_blocks_fused(None, 24, *fake[:8], 2, 2048, 16, 11008, 1762, 8, 1e-5, P(4096), None)
    assert rc == -1 and b"null" in lib.ld_last_error()
    rc = lib.ld_llm_decode_blocks_fused(P(4096), 24, *fake[:8], 3, 2048, 16, 11008, 1762, 8, 1e-5, P(4096), None)        # B = 3
    assert rc == -3 and b"fused form" in lib.ld_last_error()
    rc = lib.ld_llm_decode_blocks_fused(P(4096), 24, *fake[:8], 2, 4096, 32, 11008, 1762, 8, 1e-5, P(4096), None)        # hidden 4096
    assert rc == -3
    table = (_lib.LlmLayer * 1)()
    args = (ctypes.addressof(table), 1, 5, *fake[:7], 2, 2048, 16, 11008, 1762, 8, 1e-5, P(4096), 0)
    rc = lib.ld_llm_decode_blocks_chained(*args, P(8), P(8))                                                           # one stream twice
    assert rc == -1 and b"two different streams" in lib.ld_last_error()
    rc = lib.ld_llm_decode_blocks_chained(ctypes.addressof(table), 1, 5000, *fake[:7], 2, 2048, 16, 11008, 1762, 8, 1e-5, P(4096), 0, P(8), P(16))
    assert rc == -1 and b"position" in lib.ld_last_error()
    rc = lib.ld_llm_decode_blocks_chained(ctypes.addressof(table), 1, 5, *fake[:7], 2, 2048, 16, 20000, 1762, 8, 1e-5, P(4096), 0, P(8), P(16))
    assert rc == -3 and b"chained form" in lib.ld_last_error()                                                         # mlp 20000
    rc = lib.ld_llm_decode_blocks_chained(ctypes.addressof(table), 1, 5, *fake[:7], 2, 2048, 16, 11008, 1762, 8, 1e-5, P(4096), 0, P(8), P(16))
    assert rc == -1 and b"layer 0 has a null pointer" in lib.ld_last_error()                                           # empty layer table entry


def test_ops_refuse_cpu_tensors():
    from landiff_amd import _lib, ops
    a = torch.zeros(128, 64, dtype=torch.bfloat16)
    with pytest.raises(_lib.LandiffHipError):
        ops.gemm(a, a)


def test_pipeline_fails_loudly_without_gpu():
    from landiff_amd import _lib
    from landiff_amd.config import PipelineConfig
    from landiff_amd.pipeline import LanDiffPipeline
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.LandiffHipError):
        LanDiffPipeline(PipelineConfig.tiny(), {}, "cuda:0")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "landiff_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


def test_schedule_plan_matches_reference_tables():
    from landiff_amd.config import SamplerConfig
    from landiff_amd.schedule import build_plan
    g = np.load(os.path.join(G, "schedule.npz"))
    plan = build_plan(SamplerConfig())
    assert len(plan) == 50
    ts = g["timesteps"].tolist()
    assert [p.timestep for p in plan] == ts[::-1][:50]
    assert np.array_equal(np.array([p.cfg_scale for p in plan]), g["cfg_scales"])
    assert np.array_equal(np.array([p.c_skip for p in plan], dtype=np.float32), g["quantized"])
    assert plan[-1].last and not plan[0].has_prev and plan[1].has_prev
    # multipliers against an independent evaluation from the golden alpha table
    a = torch.from_numpy(g["alpha_cumprod_sqrt"])
    i = 7
    cur, nxt = a[i], a[i + 1]
    lam = lambda v: ((v ** 2 / (1 - v ** 2)) ** 0.5).log()
    h = lam(nxt) - lam(cur)
    assert plan[i].m1 == float(((1 - nxt ** 2) / (1 - cur ** 2)) ** 0.5 * (-h).exp())
    assert plan[i].m2 == float((-2 * h).expm1() * nxt)


def test_forced_token_schedule_full_size():
    from landiff_amd.config import LLMConfig
    from landiff_amd.llm import forced_token_schedule
    cfg = LLMConfig()
    S = 67
    full_len, forced, restricted, n_visual = forced_token_schedule(cfg, S, 13)
    assert full_len - (S + 1) == 1244 and n_visual == 1218 and len(forced) == 26
    assert forced[S + 331] == cfg.END_I and forced[S + 332] == cfg.START_P and forced[full_len - 1] == cfg.EOS
    assert (cfg.EOS, cfg.BOS, cfg.START_I, cfg.END_I, cfg.START_P, cfg.END_P, cfg.PAD) == tuple(range(2048, 2055))


def test_decoder_mask_tables():
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.detokenizer import decoder_frame_ids, rope3d_tables
    g = np.load(os.path.join(G, "decoder_mask.npz"))
    Tn, tpf, nI, nP = g["tiny_cfg"].tolist()
    cfg = TokenizerConfig(grid_h=1, grid_w=tpf, temporal=Tn, pframe_tokens=nP, num_latent_tokens=nI + (Tn - 1) * nP)
    fid = decoder_frame_ids(cfg)
    assert np.array_equal(fid[None, :] <= fid[:, None], g["tiny_dense"])
    r = np.load(os.path.join(G, "rope.npz"))
    Tn, H, W, nI, nP = r["grid"].tolist()
    cfg = TokenizerConfig(width=128, heads=2, grid_h=H, grid_w=W, temporal=Tn, pframe_tokens=nP, num_latent_tokens=nI + (Tn - 1) * nP)
    c3, s3 = rope3d_tables(cfg)
    assert np.array_equal(c3.numpy(), r["f3_real"]) and np.array_equal(s3.numpy(), r["f3_imag"])


def test_full_size_key_map_shapes():
    from landiff_amd.config import PipelineConfig
    from landiff_amd.weights import dit_spec, llm_spec, tokenizer_spec, upsampler_spec, vae_spec
    cfg = PipelineConfig.full().check()
    n = lambda spec: sum(int(np.prod(s)) for _, s, _ in spec)
    assert abs(n([x for x in llm_spec(cfg.llm) if x[0].startswith("transformer.")]) / 1e6 - 2030.2) < 1.0   # Appendix A
    dec = [x for x in tokenizer_spec(cfg.tok) if x[0].startswith("decoder.")]
    assert abs(n(dec) / 1e6 - 88.0) < 0.5
    assert abs(n([x for x in upsampler_spec(cfg.ups) if x[0].startswith("upsample_model.")]) / 1e6 - 39.2) < 0.3
    assert abs(n(vae_spec(cfg.vae)) / 1e6 - 123.4) < 0.5
    shapes = {k: s for k, s, _ in dit_spec(cfg.dit, True)}
    assert shapes["mixins.pos_embed.pos_embedding"] == (1, 17776, 1920)
    assert shapes["mixins.adaln_layer.adaLN_modulations.0.1.weight"] == (23040, 512)
    assert shapes["mixins.adaln_layer.zero_linears.14.weight"] == (1920, 1920)


def test_prompt_sharding():
    from landiff_amd.pipeline import shard_prompts
    assert shard_prompts(8, 3, 8) == [3] and shard_prompts(10, 1, 4) == [1, 5, 9]
    allp = sorted(sum((shard_prompts(13, r, 4) for r in range(4)), []))
    assert allp == list(range(13))


def test_tokenizer_encoder_checkpoint_loader(tmp_path):
    """load_tokenizer_encoder_state picks exactly the encoder-half keys out of a VideoVQ safetensors file."""
    import torch
    from safetensors.torch import save_file
    from landiff_amd.config import TokenizerConfig
    from landiff_amd.weights import init_state, load_tokenizer_encoder_state, tokenizer_encoder_spec, tokenizer_spec
    c = TokenizerConfig.tiny()
    sd = init_state(tokenizer_encoder_spec(c), 1)
    sd.update(init_state(tokenizer_spec(c), 2))                       # decoder half + project_out share the file
    f = str(tmp_path / "model.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, f)
    out = load_tokenizer_encoder_state(f)
    assert set(out) == {n for n, _, _ in tokenizer_encoder_spec(c)}
    assert all(torch.equal(out[k].float(), sd[k].reshape(out[k].shape).float()) for k in out)
    # no statistics in the file: identity normalisation
    sd.pop("mean"); sd.pop("std")
    save_file({k: v.contiguous() for k, v in sd.items()}, f)
    out = load_tokenizer_encoder_state(f)
    assert float(out["mean"].abs().max()) == 0.0 and float((out["std"] - 1).abs().max()) == 0.0


def test_sat_checkpoint_layout_round_trip(tmp_path):
    """load_diffusion_states on a miniature of the reference's checkpoint tree (ckpts/README.md:27-45): sat `.pt` files with
    ['module'] + a `latest` pointer, the lightning VAE `.pt` with ['state_dict'], each a FULL pickle carrying non-tensor
    metadata (argparse namespaces) -- torch >= 2.6 refuses those under its weights_only default."""
    import argparse
    import torch
    from landiff_amd.weights import load_diffusion_states
    root = tmp_path / "LanDiff"
    t = lambda *s: torch.randn(*s)
    base = {"model.diffusion_model.transformer.layers.0.mlp.dense_h_to_4h.weight": t(8, 2),
            "model.diffusion_model.mixins.pos_embed.pos_embedding": t(1, 4, 2)}
    ctl = "model.control_model.diffusion_model."
    mod = {"model.main_model.diffusion_model.mixins.final_layer.linear.weight": t(4, 2),
           ctl + "transformer.layers.0.mlp.dense_h_to_4h.weight": t(8, 2),                    # overrides the base weight
           ctl + "mixins.adaln_layer.zero_linears.0.weight": t(2, 2),
           ctl + "semantic_conditioner.semantic_model.model.decoder.mask_token": t(1, 1, 2),
           ctl + "semantic_conditioner.semantic_model.model.mean": t(2),
           ctl + "semantic_conditioner.upsample_model.conv_in.weight": t(2, 2, 3, 3),
           ctl + "semantic_conditioner.conv_out.weight": t(2, 2, 3, 3)}
    vae = {"decoder.conv_in.conv.weight": t(2, 2, 3, 3, 3), "loss.discriminator.w": t(2), "encoder.conv_in.conv.weight": t(2)}
    for d, sd in ((root / "diffusion", mod), (root / "CogVideoX-2b-sat" / "transformer", base)):
        (d / "1").mkdir(parents=True)
        (d / "latest").write_text("1\n")
        torch.save({"module": sd, "args": argparse.Namespace(mode="inference", bf16=True)}, d / "1" / "mp_rank_00_model_states.pt")
    (root / "CogVideoX-2b-sat" / "vae").mkdir(parents=True)
    torch.save({"state_dict": vae, "hyper_parameters": argparse.Namespace(lr=1e-4)}, root / "CogVideoX-2b-sat" / "vae" / "3d-vae.pt")
    out = load_diffusion_states(str(root / "diffusion"), str(root))
    assert set(out) == {"dit_main", "dit_control", "tok", "ups", "vae"}
    k = "transformer.layers.0.mlp.dense_h_to_4h.weight"
    assert torch.equal(out["dit_main"][k], base["model.diffusion_model." + k])                # base weights, final layer added
    assert "mixins.final_layer.linear.weight" in out["dit_main"]
    assert torch.equal(out["dit_control"][k], mod[ctl + k])                                   # control ckpt overrides the base
    assert "mixins.adaln_layer.zero_linears.0.weight" in out["dit_control"]
    assert not any(x.startswith("semantic_conditioner.") for x in out["dit_control"])
    assert set(out["tok"]) == {"decoder.mask_token", "mean"}
    assert set(out["ups"]) == {"upsample_model.conv_in.weight", "conv_out.weight"}
    assert set(out["vae"]) == {"decoder.conv_in.conv.weight"}


def test_bench_starts_its_own_ranks_or_refuses_a_mismatched_launcher():
    """`python bench.py --gpus N` without a launcher starts N ranks as child processes through torch.distributed.run (before any
    GPU call; here only the command is shown: LD_BENCH_DRY_SPAWN=1).  Under a launcher, --gpus must equal WORLD_SIZE: a mismatch
    would report a wrong n_gpus, so it stops before touching any device."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LD_BENCH_FORCE_DIST")}
    env["LD_BENCH_DRY_SPAWN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    msg = r.stderr
    assert "starting 8 rank(s)" in msg and "torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1" in msg
    assert msg.rstrip().endswith("bench.py --gpus 8 --steps 2 --warmup 1")
    env.update(WORLD_SIZE="8", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=8" in (r.stderr + r.stdout)


def test_rank_core_plan_prefers_the_gpus_numa_node():
    """rank_core_plan: ranks get cores of their GPU's NUMA node (ranks sharing a node split it), inside the cpuset; any missing
    piece of information gives the plain contiguous cut for every rank.  gpu_numa_nodes reads a sysfs tree without HIP."""
    from landiff_amd.pipeline import gpu_numa_nodes, rank_core_plan, rank_core_slice
    cores = list(range(64))
    node_cpus = {0: list(range(0, 32)), 1: list(range(32, 64))}
    plan = rank_core_plan(4, cores, [1, 1, 0, 0], node_cpus)
    assert plan == [list(range(32, 48)), list(range(48, 64)), list(range(0, 16)), list(range(16, 32))]
    plain = [rank_core_slice(r, 4, cores) for r in range(4)]
    assert rank_core_plan(4, cores, [1, None, 0, 0], node_cpus) == plain          # one GPU without a node
    assert rank_core_plan(4, cores, [0, 1], node_cpus) == plain                   # fewer GPUs known than ranks
    assert rank_core_plan(4, cores, None, None) == plain
    assert rank_core_plan(4, list(range(33)), [1, 1, 1, 0], node_cpus) == [rank_core_slice(r, 4, list(range(33))) for r in range(4)]  # node 1: 1 allowed core, 3 ranks
    flat = [c for sl in plan for c in sl]
    assert len(flat) == len(set(flat)) == 64


def test_gpu_numa_nodes_reads_sysfs_without_hip(tmp_path, monkeypatch):
    from landiff_amd.pipeline import gpu_numa_nodes
    # two compute GPUs, a BMC display chip of another vendor, and an AMD display-only part without a shader-clock hwmon (skipped)
    for i, (pci, vendor, node, compute) in enumerate([("0000:05:00.0", "0x1002", 0, True), ("0000:85:00.0", "0x1002", 1, True),
                                                      ("0000:01:00.0", "0x1a03", -1, False), ("0000:03:00.0", "0x1002", 0, False),
                                                      ("0000:c5:00.0", "0x1002", 1, True)]):
        real = tmp_path / "devices" / "pci" / pci
        real.mkdir(parents=True)
        (real / "vendor").write_text(vendor + "\n"); (real / "numa_node").write_text(f"{node}\n")
        if compute:
            (real / "hwmon" / "hwmon0").mkdir(parents=True); (real / "hwmon" / "hwmon0" / "freq1_input").write_text("2400000000\n")
        card = tmp_path / "class" / "drm" / f"card{i}"
        card.mkdir(parents=True)
        os.symlink(real, card / "device")
    for n, cl in ((0, "0-3,8-9"), (1, "4-7")):
        d = tmp_path / "devices" / "system" / "node" / f"node{n}"
        d.mkdir(parents=True); (d / "cpulist").write_text(cl + "\n")
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    nodes, cpus = gpu_numa_nodes(str(tmp_path))
    assert nodes == [0, 1, 1] and cpus == {0: [0, 1, 2, 3, 8, 9], 1: [4, 5, 6, 7]}
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert gpu_numa_nodes(str(tmp_path))[0] == [1]
    # the filters compose: ROCR hides devices from HIP, HIP indexes what is left; CUDA_VISIBLE_DEVICES only when HIP's is unset
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,2"); monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")
    assert gpu_numa_nodes(str(tmp_path))[0] == [1, 0]
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "0")
    assert gpu_numa_nodes(str(tmp_path))[0] == [1, 0]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert gpu_numa_nodes(str(tmp_path))[0] == [0]
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-1234abcd")          # a UUID list: no claim -> the plain core cut
    assert gpu_numa_nodes(str(tmp_path))[0] == []


def test_save_video_tensor_fallback_writes_playable_avi(tmp_path):
    """Without imageio (as in this image) save_video_tensor writes a Motion-JPEG AVI + the exact frames: the RIFF structure
    parses, every frame decodes to the right size and close to the input, the .npy holds trunc(video * 255)."""
    import io, struct, warnings
    import torch
    from PIL import Image
    from landiff.utils import cthw_to_numpy_images, save_video_tensor
    try:
        import imageio  # noqa: F401
        pytest.skip("imageio present: the reference path is taken")
    except ImportError:
        pass
    T, H, W = 5, 48, 80
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    video = torch.stack([torch.stack([yy * (t + 1) / T, xx, 1 - yy]) for t in range(T)], dim=1)      # [3, T, H, W] in [0, 1]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        save_video_tensor(video, str(tmp_path / "out.mp4"), fps=8)
    assert any("Motion-JPEG" in str(x.message) for x in w)
    frames = np.load(tmp_path / "out.frames.npy")
    assert np.array_equal(frames, cthw_to_numpy_images(video)) and frames.shape == (T, H, W, 3)
    raw = (tmp_path / "out.avi").read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8
    movi = raw.index(b"movi")
    pos, n = movi + 4, 0
    while raw[pos:pos + 4] == b"00dc":
        size = struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        img = np.asarray(Image.open(io.BytesIO(raw[pos + 8:pos + 8 + size])).convert("RGB"))
        assert img.shape == (H, W, 3) and np.abs(img.astype(int) - frames[n].astype(int)).mean() < 4.0
        pos += 8 + size + (size & 1); n += 1
    assert n == T and raw[pos:pos + 4] == b"idx1"


def test_no_packed_f32_instruction_reads_a_high_register_into_its_low_lane():
    """tools/audit_pk_f32.py on the built libraries: the operand form that round 5 found to misread beside co-resident MFMA waves
    (v_pk_add / mul / fma_f32 with op_sel selecting the high register for the low lane) must not be in any kernel we ship -- nor in the
    variants build.  A toolchain bump or a new kernel that brings it back fails here, on CPU, before it costs anyone a token."""
    import importlib.util
    from landiff_amd import _lib
    spec = importlib.util.spec_from_file_location("audit_pk_f32", os.path.join(ROOT, "tools", "audit_pk_f32.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    for lib in (_lib.LIB_PATH, _lib.VARIANTS_LIB_PATH):
        if not os.path.exists(lib):
            continue
        n_obj, n_k, found = mod.audit(lib)
        assert n_obj >= 10 and n_k >= 150, (lib, n_obj, n_k)          # the audit really saw the kernels
        assert not found, (lib, found[:5])


def test_no_spill_traffic_inside_an_mfma_loop():
    """tools/audit_spills.py on the shipped library: no scratch_load / scratch_store inside an innermost loop that issues MFMAs.  The
    pipelined kernels wait for their LDS-DMA with counted vmcnt; a spilled register's reload is a vector-memory load that hipcc waits
    for with vmcnt(0), so ONE reload in a K loop drains the prefetch queue every iteration (round 6: three MXFP8 GEMM instantiations
    and pass A of the exact attention form ran that way until this audit existed)."""
    import importlib.util
    from landiff_amd import _lib
    spec = importlib.util.spec_from_file_location("audit_spills", os.path.join(ROOT, "tools", "audit_spills.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    n_obj, n_k, found = mod.audit(_lib.LIB_PATH)
    assert n_obj >= 10 and n_k >= 150, (n_obj, n_k)
    # One kernel is let through: the plain two-stage attention kernel (TiTok frame mask, short problems; 13 ms of a 15.9 s video)
    # waits for every tile with vmcnt(0) by construction, so a reload drains nothing -- and its 3-waves-per-SIMD register budget
    # WITH the spills measured faster than 2 waves without (1.108 vs 1.211 ms on the TiTok shape, tools/titok_attn_time.py).
    found = [f for f in found if "ld_attn_kernel" not in f[0]]
    assert not found, found[:5]
    # positive control: the audit must SEE a spill when there is one (tools/probe/spill_in_mfma_loop.hip cannot avoid it)
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        lib = os.path.join(tmp, "libspill.so")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                        os.path.join(ROOT, "tools", "probe", "spill_in_mfma_loop.hip"), "-o", lib], check=True, capture_output=True)
        n_obj, n_k, found = mod.audit(lib)
        assert n_obj == 1 and n_k == 1 and len(found) == 1 and found[0][0] == "ld_probe_spill_in_mfma_loop" and found[0][3] > 0, found


def test_attention_q128_isa_audit():
    """ld_attn_q128.hip owns a[0:223] by name: the build must not spill, and the compiler must not emit a single accumulator-
    register access of its own in that kernel; the hot loop must hold the instruction mix it was written for, the exp2 / pack
    work interleaved with the MFMAs (tools/audit_attn_q128.py rebuilds the file with -save-temps and reads the ISA)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("audit_attn_q128", os.path.join(ROOT, "tools", "audit_attn_q128.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        report = mod.audit(mod.build(tmp))
    assert len(report) == 5 and all("no compiler a[] traffic" in r for r in report)


def test_verify_real_checkpoint_tool(tmp_path, capsys):
    """tools/verify_real_checkpoint.py (the executable checklist for the seams that rest on absent third-party code): runs clean on
    a synthetic checkpoint tree in the reference layout, fails on a shape mismatch, and its qkv-layout statistic tells the
    [q|k|v]-thirds layout from a permuted one on weights whose per-head q / k projections are coupled (as trained ones are)."""
    import importlib.util
    import subprocess
    import sys
    from facade_helpers import build_config0_workdir
    from landiff_amd.config import DiTConfig
    work = str(tmp_path)
    build_config0_workdir(work)
    tool = os.path.join(ROOT, "tools", "verify_real_checkpoint.py")
    ck = os.path.join(work, "ckpts", "LanDiff")
    r = subprocess.run([sys.executable, tool, "--ckpt", ck, "--config", "config0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "keys/shapes OK" in r.stdout and "undecidable" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, tool, "--ckpt", ck, "--config", "tiny"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "SHAPE" in r.stdout                        # the config0 tree audited against another configuration
    spec = importlib.util.spec_from_file_location("verify_real_checkpoint", tool)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    c = DiTConfig.tiny()
    d, H, hd = c.hidden, c.heads, c.head_dim
    g = torch.Generator().manual_seed(0)
    q = torch.randn(H, hd, d, generator=g)
    k = q + 0.5 * torch.randn(H, hd, d, generator=g)                        # a head's k projection shares its q projection's subspace
    v = torch.randn(H, hd, d, generator=g)
    name = "transformer.layers.0.attention.query_key_value.weight"
    assert mod.qkv_layout({name: torch.cat([q.reshape(d, d), k.reshape(d, d), v.reshape(d, d)])}, c, 0) is True
    assert mod.qkv_layout({name: torch.cat([v.reshape(d, d), q.reshape(d, d), k.reshape(d, d)])}, c, 0) is False
    assert mod.qkv_layout({name: torch.stack([q, k, v], 1).reshape(3 * d, d)}, c, 0) is False      # per-head interleaved
    capsys.readouterr()


def test_vae_causal_conv_windows_keep_the_halo_in_place(monkeypatch):
    """VAEDecoder's per-conv input windows (round 5): a chunk's producer writes behind the previous chunk's last two padded frames,
    the conv's window starts two frames earlier, and the two frames move to the front only when the window would run past the end
    of the conv's buffer -- the reference's per-conv cache (ContextParallelCausalConv3d.forward(x, clear_cache),
    cp_enc_dec.py:436-466) without its copy out and copy in.  Host logic on CPU tensors, the conv launch stubbed: what the conv
    would read is [the previous chunk's last two frames | this chunk's frames] (first chunk: its first frame three times), the
    spatial border stays zero, the last chunk clears the state, a later decode starts fresh, a kept state continues."""
    import torch
    from landiff_amd import ops, vae as vae_mod
    from landiff_amd.config import PipelineConfig
    seen = []
    monkeypatch.setattr(ops, "conv_cl", lambda xp, w, T, H, W, **kw: seen.append(xp.clone()) or None)
    dec = vae_mod.VAEDecoder({}, PipelineConfig.tiny().vae, torch.device("cpu"))
    dec.w = {"c.conv.weight": None, "c.conv.bias": None}
    H, W, C = 3, 2, 8
    frame = [0]

    def chunk(T, clear):
        win = dec._conv_window("c", T, H, W, C)
        assert tuple(win.shape) == (T + 2, H + 2, W + 2, C) and win.is_contiguous()
        ids = []
        for t in range(T):
            frame[0] += 1
            win[2 + t, 1:-1, 1:-1] = float(frame[0])
            ids.append(frame[0])
        dec._causal_conv(win, "c", T, H, W, clear)
        return ids

    def check(xp, halo, ids):
        want = list(halo) + list(ids)
        assert [int(xp[t, 1, 1, 0]) for t in range(xp.shape[0])] == want, (want, [int(xp[t, 1, 1, 0]) for t in range(xp.shape[0])])
        assert (xp[:, 1:-1, 1:-1].amin(dim=(1, 2, 3)) == xp[:, 1:-1, 1:-1].amax(dim=(1, 2, 3))).all()      # whole frames moved
        assert float(xp[:, 0].abs().sum() + xp[:, -1].abs().sum() + xp[:, :, 0].abs().sum() + xp[:, :, -1].abs().sum()) == 0.0

    for rep in range(2):                                    # two decodes on the same decoder: the second starts fresh
        dec.cache = {}
        seen.clear()
        schedule = [9, 8, 8, 8, 8, 8]
        prev = None
        for i, T in enumerate(schedule):
            ids = chunk(T, clear=(i == len(schedule) - 1))
            check(seen[-1], [ids[0], ids[0]] if prev is None else prev[-2:], ids)
            prev = ids
            assert ("c" in dec.cache) == (i < len(schedule) - 1)
        assert dec._win["c"].shape[0] == 2 * 9 + 2            # one buffer, two chunks + 2 frames long, reused throughout
    # a kept state (streaming): decode(stream_keep) then continued chunks of other lengths, single frames included -- the halo is
    # always the last two frames of the conv's padded input so far
    dec.cache = {}
    hist = chunk(5, clear=False)
    for T in (2, 2, 1, 1, 3, 2, 1, 9, 1, 14, 2, 14):          # 14: longer than the buffer -- a new one, the halo moves along
        halo = hist[-2:]
        ids = chunk(T, clear=False)
        check(seen[-1], halo, ids)
        hist += ids
    assert dec.cache
