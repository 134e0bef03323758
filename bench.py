"""bench.py -- end-to-end frames/sec of the LanDiff hot path on MI355X (BASELINE.json metric).

One "step" = one full 49-frame 480x720 video per GPU: AR semantic-token decode (1244 steps, 2.03 B params) ->
detokenize (TiTok decoder + conv upsampler) -> 50 sampler steps over the control+main DiT (B=2 CFG pair,
17 776 tokens) -> chunked 3D-VAE decode -> uint8 frames resident in HBM.  Inputs (prompt embeddings) and the
seeded random-init weights are resident in HBM before the timed region.  N > 1: one process per GPU
(torchrun), prompts sharded data-parallel, one RCCL all_gather of the finished uint8 frames per step.

  python bench.py --gpus 1 --steps 1 --warmup 1
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

METRIC = "end-to-end frames/sec, 49f 480x720 @50 steps"
# HBM bytes per attention launch at the headline shape from separate rocprofv3 --pmc passes
# (profiles/r02_attn_q64_pmc_hbm.txt): (2 x FETCH_SIZE [gfx950 correction, MI355X_MICROARCH.md HBM] + WRITE_SIZE) x 1024
# = (2 * 489400 + 141700) KiB.  Reported only when the run is that kernel at that shape.
ATTN_TRAFFIC = {"kernel": "ld_attn_q64_kernel", "bytes": (2 * 489400 + 141700) * 1024}


def cpu_baseline(cfg, budget_s: float = 25.0):
    """The oracle (CPU restatement, "port") timed on the host cores on a bounded sample of the same workload:
    one DiT layer-call at the full shape (B=2, N=17776), one LLM decode step, one TiTok decoder layer and one VAE
    3x3x3 conv at a reduced extent; extrapolated with the multipliers of BASELINE.md section 3."""
    import dataclasses
    from landiff_amd.weights import dit_spec, init_state, llm_spec, tokenizer_spec
    from oracle.dit import DiTOracle
    from oracle.llm import LLMOracle
    from oracle.tokenizer import DetokenizerOracle, rope3d_table, frame_ids
    cores = min(os.cpu_count() or 1, 64)      # torch CPU ops stop scaling (and small ops regress) far below 256 threads
    torch.set_num_threads(cores)
    out = {}
    with torch.no_grad():
        d1 = dataclasses.replace(cfg.dit, layers_main=1, layers_control=1)
        orc = DiTOracle(init_state(dit_spec(d1, False), 1), d1, False, torch.float32)
        h = torch.randn(2, d1.seq_len, d1.hidden)
        emb = torch.randn(2, d1.time_embed_dim)
        t0 = time.perf_counter(); orc.layer(0, h, emb); out["dit_layer_s"] = time.perf_counter() - t0
        l1 = dataclasses.replace(cfg.llm, num_layers=2)
        lo = LLMOracle(init_state(llm_spec(l1), 2), l1, torch.float32)
        cache = [(torch.randn(2, 1200, l1.heads, l1.head_dim), torch.randn(2, 1200, l1.heads, l1.head_dim)) for _ in range(2)]
        cos, sin = torch.ones(1, 1, l1.head_dim // 2), torch.zeros(1, 1, l1.head_dim // 2)
        x = torch.randn(2, 1, l1.hidden)
        def llm_layers():
            c2 = list(cache)
            y = x
            for i in range(2):
                y = lo.block(i, y, c2, cos, sin)
        llm_layers()                                     # warm-up (thread pool start-up, first-touch)
        t0 = time.perf_counter(); llm_layers(); out["llm_layer_s"] = (time.perf_counter() - t0) / 2
        t1 = dataclasses.replace(cfg.tok, layers=1)
        to = DetokenizerOracle(init_state(tokenizer_spec(t1), 3), {}, t1, cfg.ups, torch.float32)
        xt = torch.randn(1, t1.seq_len, t1.width)
        c3, s3 = rope3d_table(t1)
        fid = torch.from_numpy(frame_ids(t1))
        mask = (fid[None, :] <= fid[:, None])[None, None]
        t0 = time.perf_counter(); to.titok_block(0, xt, c3[None], s3[None], mask); out["titok_layer_s"] = time.perf_counter() - t0
        xc = torch.randn(1, 128, 4, 120, 180)
        wc = torch.randn(128, 128, 3, 3, 3)
        torch.nn.functional.conv3d(xc, wc, padding=(0, 1, 1))      # warm-up
        t0 = time.perf_counter(); torch.nn.functional.conv3d(xc, wc, padding=(0, 1, 1)); dt = time.perf_counter() - t0
        out["conv_tflops"] = 2 * 128 * 128 * 27 * 2 * 120 * 180 / dt / 1e12
    d, l = cfg.dit, cfg.llm
    total = (out["dit_layer_s"] * (d.layers_main + d.layers_control) * cfg.sampler.num_steps
             + out["llm_layer_s"] * l.num_layers * 1244 + out["titok_layer_s"] * cfg.tok.layers + 315.0 / out["conv_tflops"])
    frames = 4 * d.latent_frames - 3
    return {"value": frames / total, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": ("oracle fp32 on host cores: 1 DiT layer-call (B=2,N=17776) %.1fs, 1 LLM decode layer %.3fs, 1 TiTok layer %.1fs, "
                       "conv3d %.2f TFLOP/s; extrapolated x(45x50), x(24x1244), x12, 315 TFLOP -> %.0f s/video"
                       % (out["dit_layer_s"], out["llm_layer_s"], out["titok_layer_s"], out["conv_tflops"], total))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tiny", action="store_true", help="tiny random-init config (plumbing check, not the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prompts-per-step", type=int, default=1,
                    help="serving mode, NOT the BASELINE configuration (which is one prompt per GPU and step): P prompts per GPU "
                         "and step through LanDiffPipeline.generate_many -- the AR decode of prompt i+1 runs on a second stream "
                         "while prompt i is in the DiT loop; per-prompt results are identical to the one-prompt path")
    ap.add_argument("--stream-chunks", type=int, default=0,
                    help="BASELINE config 3 instead of the headline config: N-chunk streaming long video (prefix 7 latent "
                         "frames pinned per later chunk, LLM KV / latents / VAE conv caches reused in HBM)")
    ap.add_argument("--fp8-gemm", nargs="?", const="mx", default=None, choices=["mx", "row"],
                    help="BASELINE configs[4] only (with --stream-chunks): e4m3 operands for the DiT's qkv / dense / 4h / 4h->h "
                         "linears -- mx (default): MXFP8 block scales, quantisation fused into LayerNorm / GELU epilogue; row: "
                         "per-row scales with quantise passes.  Reduced precision: not the headline metric, own workload name and dtype")
    args = ap.parse_args()
    if args.prompts_per_step > 1 and (args.stream_chunks or args.fp8_gemm or args.tiny):
        raise SystemExit("--prompts-per-step applies to the headline workload only")
    if args.fp8_gemm and not args.stream_chunks:
        raise SystemExit("--fp8-gemm belongs to the streaming long-video configuration (BASELINE configs[4]): add --stream-chunks N")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N-GPU runs with "
                         "`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...`")
    import torch.distributed as dist
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    # LD_BENCH_FORCE_DIST=1 under torchrun --nproc-per-node 1 takes the RCCL code path (init, barrier, all_gather,
    # all_reduce) with a single rank: the only way to exercise it on a 1-GPU box (tests/test_gpu_variants.py)
    use_dist = world > 1 or (os.environ.get("LD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm

    from landiff_amd.config import PipelineConfig
    from landiff_amd.pipeline import LanDiffPipeline, gather_frames, synthetic_inputs
    from landiff_amd.weights import init_pipeline_state

    cfg = (PipelineConfig.tiny(3) if args.tiny else PipelineConfig.full()).check()
    states = init_pipeline_state(cfg, seed=1234, dtype=torch.bfloat16, device=dev)
    # fp32 where the reference keeps fp32 parameters on the LLM path (norm gains, final LN, head, embedding table)
    stream = args.stream_chunks
    prefix = 7 if not args.tiny else 1
    if stream:
        T_, new_, n_seg = (cfg.dit.latent_frames, cfg.dit.latent_frames - prefix,
                           -(-(cfg.dit.latent_frames + (stream - 1) * (cfg.dit.latent_frames - prefix)) // cfg.llm.segment_length))
        pipe = LanDiffPipeline(cfg, states, dev, max_llm_frames=n_seg * cfg.llm.segment_length, fp8_gemm=args.fp8_gemm)
    else:
        pipe = LanDiffPipeline(cfg, states, dev)
    del states
    torch.cuda.empty_cache()
    # rank r works on prompt r (weak scaling: one prompt per GPU per step); same seed convention as a single-GPU run
    inp = synthetic_inputs(cfg, dev, n_text=64 if not args.tiny else 6, seed=42 + rank)

    P = args.prompts_per_step
    import dataclasses
    inps = [dataclasses.replace(inp, seed=inp.seed + 1000 * i) for i in range(P)]

    def one_step():
        if P > 1:
            frames = torch.cat(pipe.generate_many(inps), dim=0)          # [P * 49, H, W, 3]
        else:
            frames = pipe.generate_stream(inp, stream, prefix_frames=prefix) if stream else pipe(inp)
        gathered = gather_frames(frames[None], world, force=use_dist)
        return gathered

    for _ in range(args.warmup):
        one_step()
    pipe.timings = {}
    pipe.dit.attn_events = []
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    n_frames = out[0].shape[1]
    if rank == 0:
        d = cfg.dit
        ev = pipe.dit.attn_events
        attn_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        flops = 4.0 * 2 * d.heads * d.seq_len * d.seq_len * d.head_dim      # algorithmic FLOPs of one launch
        achieved = flops / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
        peak = 2500.0
        from landiff_amd import _lib
        kname = (_lib.load().ld_attn_last_kernel() or b"").decode()      # what the launcher actually ran (shape + LD_ATTN_* knobs)
        # the PMC traffic figure was collected on the default kernel at the headline shape only (profiles/, see ATTN_TRAFFIC)
        headline = (not args.tiny and not stream and not args.fp8_gemm and kname == ATTN_TRAFFIC["kernel"]
                    and not any(os.environ.get(k) for k in ("LD_ATTN_NW", "LD_ATTN_SAFE", "LD_ATTN_VARIANT", "LD_ATTN_MSUM", "LD_ATTN_Q64")))
        res = {
            "metric": METRIC, "value": world * n_frames * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.fp8_gemm else f"bf16 with fp8-e4m3 ({args.fp8_gemm}) DiT linears (configs[4], not the headline precision)",
            "data": "synthetic",
            "config": {"workload": ("tiny random-init plumbing config" if args.tiny else
                                    "LanDiff 5B full pipeline, single prompt per GPU, 49f 480x720, 50 sampler steps "
                                    "(VPSDE DPM-Solver++(2M), DynamicCFG), bf16, random-init weights at true shapes")
                                   + (f"; streaming long video: {stream} chunks, {prefix} prefix latent frames pinned per later "
                                      f"chunk, {n_frames} frames" if stream else "")
                                   + (f"; fp8 e4m3 MFMA ({args.fp8_gemm} scaling) for the DiT qkv/dense/4h/4h->h linears" if args.fp8_gemm else "")
                                   + (f"; SERVING MODE, not the BASELINE configuration: {P} prompts per GPU and step, AR decode of prompt "
                                      f"i+1 overlapped with the DiT loop of prompt i (generate_many)" if P > 1 else ""),
                       "frames": n_frames, "height": 8 * d.latent_h, "width": 8 * d.latent_w,
                       "sampler_steps": cfg.sampler.num_steps, "llm_steps": (1244 if not stream else None) if not args.tiny else None,
                       "prompts_per_step": P,
                       "parallelism": f"dp{world} over prompts, RCCL all_gather of uint8 frames only"},
            "stage_seconds_rank0": {k: round(v / args.steps, 3) for k, v in pipe.timings.items()},
            "roofline": {"kernel": "%s (DiT joint text+video attention, B=2,H=%d,N=%d,D=64)" % (kname, d.heads, d.seq_len),
                         "bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "traffic": ATTN_TRAFFIC["bytes"] if headline else None, "launches": len(ev),
                         "avg_launch_ms": round(attn_ms, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(PipelineConfig.full())
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
