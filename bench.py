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
# HBM bytes per attention launch at the headline shape: read from the newest profiles/r*_attn_pmc_hbm.txt, the summary of separate
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over tools/attn_one.py (tools/pmc_attn.sh writes it, with the gfx950
# FETCH_SIZE correction of MI355X_MICROARCH.md applied).  Reported only when the run is that kernel at that shape.
def _attn_traffic():
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_attn_pmc_hbm.txt")))
    for f in reversed(files):
        m = re.search(r"^TRAFFIC_BYTES (\d+) KERNEL (\S+)", open(f).read(), re.M)
        if m:
            return {"kernel": m.group(2), "bytes": int(m.group(1)), "source": os.path.relpath(f, ROOT)}
    return {"kernel": None, "bytes": None, "source": None}


ATTN_TRAFFIC = _attn_traffic()
MFMA_BF16_PEAK = 2500.0      # TFLOP/s, dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK = 8000.0            # GB/s
VAE_TFLOP = 315.0            # per 49-frame video (58.8 + 5 x 51.3)


class ClockPowerSampler:
    """sclk / socket power of the GPU this rank runs on, sampled from sysfs hwmon files (freq1_input, power1_input) by a helper
    thread that never touches HIP.  The card is found by the PCI address of the torch device; None when the files are not there."""

    def __init__(self, dev, period=0.2):
        import glob
        self.period, self.samples, self._run, self._thread = period, [], False, None
        self.freq = self.power = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            want = None
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
            if not hw or not os.path.exists(os.path.join(hw[0], "freq1_input")):
                continue
            if want is None or os.path.basename(os.path.realpath(card)) == want:
                self.freq, self.power = os.path.join(hw[0], "freq1_input"), os.path.join(hw[0], "power1_input")
                self.card = os.path.basename(os.path.realpath(card))
                break

    def _loop(self):
        while self._run:
            try:
                f = int(open(self.freq).read()) / 1e6
                w = int(open(self.power).read()) / 1e6 if os.path.exists(self.power) else float("nan")
                self.samples.append((f, w))
            except Exception:
                pass
            time.sleep(self.period)

    def start(self):
        if self.freq is None:
            return
        import threading
        self.samples, self._run = [], True
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._run = False
        self._thread.join()
        if not self.samples:
            return None
        f = [a for a, _ in self.samples]; w = [b for _, b in self.samples if b == b]
        return {"mean_sclk_mhz": round(sum(f) / len(f), 0), "min_sclk_mhz": round(min(f), 0), "max_sclk_mhz": round(max(f), 0),
                "mean_power_w": round(sum(w) / len(w), 0) if w else None, "samples": len(f), "pci": self.card}


def calibrate(dev, sampler):
    """What THIS box sustains, measured right before the timed region (~1 s each): an MFMA-only loop on random bf16 operands
    (ld_calib_mfma_bf16: no memory traffic, the matrix pipe under the chip's power governor) and a 16-byte-per-lane streaming read
    of 1 GiB (ld_calib_stream_read).  The roofline fractions against the datasheet peaks stay the headline; `frac_of_box_ceiling`
    divides by these instead, which takes the +-4-5 % box-to-box spread of the pool out of a round-over-round comparison."""
    import ctypes
    from landiff_amd import _lib
    from landiff_amd._lib import check
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ops_ = torch.randn(1 << 19, device=dev).to(torch.bfloat16)                    # 1 MiB of random operands
    sink = torch.zeros(4, device=dev, dtype=torch.float32)
    isink = torch.zeros(4, device=dev, dtype=torch.int32)
    buf = torch.empty(1 << 28, device=dev, dtype=torch.int32).random_()          # 1 GiB: four times the Infinity Cache
    flops = ctypes.c_double(0.0)
    res = {}

    def timed(launch, seconds):
        launch(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, n = time.perf_counter(), 0
        e0.record()
        while time.perf_counter() - t0 < seconds:
            for _ in range(4):
                launch()
            n += 4
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n

    sampler.start()
    dt = timed(lambda: check(lib.ld_calib_mfma_bf16(ctypes.c_void_p(ops_.data_ptr()), ops_.numel() * 2, ctypes.c_void_p(sink.data_ptr()),
                                                    2048, 2000, ctypes.byref(flops), st), "ld_calib_mfma_bf16"), 1.0)
    clk = sampler.stop()
    res["mfma_tflops"] = round(flops.value / dt / 1e12, 1)
    res["mfma_loop_sclk_mhz"] = clk["mean_sclk_mhz"] if clk else None
    dt = timed(lambda: check(lib.ld_calib_stream_read(ctypes.c_void_p(buf.data_ptr()), buf.numel() * 4, ctypes.c_void_p(isink.data_ptr()), st),
                             "ld_calib_stream_read"), 1.0)
    res["hbm_gbs"] = round(buf.numel() * 4 / dt / 1e9, 1)
    res["what"] = ("1 s each before the timed region: MFMA-only loop (v_mfma_f32_16x16x32_bf16, random operands, one wave per SIMD) "
                   "and a 16-B/lane read of 1 GiB; sclk / power: sysfs hwmon of this GPU sampled every 0.2 s")
    del buf
    torch.cuda.empty_cache()
    return res


def detok_tflop(cfg) -> float:
    """Algorithmic FLOPs of the detokenize stage (TiTok decoder: linears + frame-masked attention at its 0.5385 mask density;
    conv upsampler + conv_out), in TFLOP."""
    from landiff_amd.weights import upsampler_levels
    t, u = cfg.tok, cfg.ups
    N, w = t.seq_len, t.width
    lin = t.layers * 2.0 * N * (3 * w * w + w * w + 8 * w * w)
    fid_density = 0.5385 if t.temporal == 13 else 0.6
    attn = t.layers * 4.0 * t.heads * N * N * t.head_dim * fid_density
    head = 2.0 * t.n_visual * (w * 2 * w + 2 * w * t.out_channels) if hasattr(t, "n_visual") else 0.0
    F, H, W = t.temporal, t.grid_h, t.grid_w
    conv = lambda cin, cout, h, ww: 2.0 * F * h * ww * cin * cout * 9
    top = u.ch * u.ch_mult[-1]
    fl = conv(u.z_channels, top, H, W) + 4 * conv(top, top, H, W)
    ch = top
    for lvl, blocks, up in upsampler_levels(u):
        for cin, cout in blocks:
            fl += conv(cin, cout, H, W) + conv(cout, cout, H, W) + (2.0 * F * H * W * cin * cout if cin != cout else 0.0)
            ch = cout
        if up:                      # PixelShuffle(2), then Conv2d(C / 4 -> C) at twice the resolution
            H, W = 2 * H, 2 * W
            fl += conv(ch // 4, ch, H, W)
    fl += conv(ch, u.out_ch, H, W) + conv(u.out_ch, u.target_dim, H, W)
    return (lin + attn + head + fl) / 1e12


def _median(fn, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def cpu_baseline(cfg):
    """The oracle (CPU restatement, "port") timed on the host cores, as BASELINE.md section 3 lays out:
      (i)  BASELINE configs[0] end to end (random-init tiny DiT, 8 latent frames, 64x64 latent, 2 DDIM steps: AR decode ->
           detokenize -> sampler -> chunked VAE decode -> uint8 frames), and
      (ii) at full shapes one unit of each hot kernel -- one DiT layer-call (B=2, N=17 776), one LLM decode layer (median of 5),
           one TiTok decoder layer, one level-0 VAE resblock at 480x720 (2 frames) -- extrapolated with the explicit multipliers
           45 x 50, 24 x 1244, 12 and 315 TFLOP / (resblock TFLOP/s) to seconds per 49-frame video (`value`)."""
    import dataclasses
    from landiff_amd.config import PipelineConfig, VAEConfig
    from landiff_amd.weights import _res3d, dit_spec, init_pipeline_state, init_state, llm_spec, tokenizer_spec
    from oracle.dit import DiTOracle
    from oracle.llm import LLMOracle
    from oracle.pipeline import PipelineOracle
    from oracle.tokenizer import DetokenizerOracle, rope3d_table, frame_ids
    from oracle.vae import VAEDecoderOracle
    cores = min(os.cpu_count() or 1, 64)      # torch CPU ops stop scaling (and small ops regress) far below 256 threads
    torch.set_num_threads(cores)
    out = {}
    with torch.no_grad():
        # (i) configs[0] end to end
        c0 = PipelineConfig.config0().check()
        st0 = init_pipeline_state(c0, seed=1234)
        orc = PipelineOracle(c0, st0, torch.float32)
        g = torch.Generator().manual_seed(42)
        text = torch.randn(6, c0.llm.text_dim, generator=g)
        ctx = torch.randn(1, c0.dit.text_len, c0.dit.text_dim, generator=g)
        d0 = c0.dit
        noise = torch.randn(1, d0.latent_frames, d0.in_channels, d0.latent_h, d0.latent_w, generator=g)
        t0 = time.perf_counter()
        tok = orc.tokens(text)
        z = orc.latent(tok, ctx, noise=noise)
        _, fr = orc.frames(z)
        out["config0_s"] = time.perf_counter() - t0
        out["config0_frames"] = int(fr.shape[0])
        del orc, st0
        # (ii) full-shape units
        d1 = dataclasses.replace(cfg.dit, layers_main=1, layers_control=1)
        orc = DiTOracle(init_state(dit_spec(d1, False), 1), d1, False, torch.float32)
        h = torch.randn(2, d1.seq_len, d1.hidden)
        emb = torch.randn(2, d1.time_embed_dim)
        out["dit_layer_s"] = _median(lambda: orc.layer(0, h, emb), 3)
        del orc, h
        l1 = dataclasses.replace(cfg.llm, num_layers=1)
        lo = LLMOracle(init_state(llm_spec(l1), 2), l1, torch.float32)
        kv = (torch.randn(2, 1200, l1.heads, l1.head_dim), torch.randn(2, 1200, l1.heads, l1.head_dim))
        cos, sin = torch.ones(1, 1, l1.head_dim // 2), torch.zeros(1, 1, l1.head_dim // 2)
        x = torch.randn(2, 1, l1.hidden)
        llm_layer = lambda: lo.block(0, x, [kv], cos, sin)
        llm_layer()                                      # warm-up (thread pool start-up, first touch)
        out["llm_layer_s"] = _median(llm_layer, 5)
        del lo
        t1 = dataclasses.replace(cfg.tok, layers=1)
        to = DetokenizerOracle(init_state(tokenizer_spec(t1), 3), {}, t1, cfg.ups, torch.float32)
        xt = torch.randn(1, t1.seq_len, t1.width)
        c3, s3 = rope3d_table(t1)
        fid = torch.from_numpy(frame_ids(t1))
        mask = (fid[None, :] <= fid[:, None])[None, None]
        out["titok_layer_s"] = _median(lambda: to.titok_block(0, xt, c3[None], s3[None], mask), 3)
        del to, mask
        vc = VAEConfig()
        C, T, H, W = vc.ch, 2, 480, 720
        p = "decoder.up.0.block.1."
        vo = VAEDecoderOracle(init_state(_res3d(p, C, C, vc.z_channels), 4), vc, torch.float32)
        xv = torch.randn(1, C, T, H, W)
        zq = torch.randn(1, vc.z_channels, 1, 60, 90)
        dt = _median(lambda: vo.resblock(xv, zq, p, C, C, True), 3)
        out["vae_resblock_s"] = dt
        out["vae_tflops"] = 2 * (2.0 * T * H * W * C * C * 27) / dt / 1e12
    d, l = cfg.dit, cfg.llm
    total = (out["dit_layer_s"] * (d.layers_main + d.layers_control) * cfg.sampler.num_steps
             + out["llm_layer_s"] * l.num_layers * 1244 + out["titok_layer_s"] * cfg.tok.layers + VAE_TFLOP / out["vae_tflops"])
    frames = 4 * d.latent_frames - 3
    return {"value": frames / total, "unit": "frames/s", "cores": cores, "kind": "port", **host_cpu(),
            "threads_note": "torch intra-op threads = min(os.cpu_count(), 64): the oracle's CPU kernels stop scaling well below the host's core count",
            "extrapolated_seconds_per_video": round(total, 1),
            "config0_end_to_end": {"seconds": round(out["config0_s"], 2), "frames": out["config0_frames"],
                                   "frames_per_s": round(out["config0_frames"] / out["config0_s"], 3),
                                   "workload": "BASELINE configs[0]: random-init tiny DiT, 8 latent frames, 64x64 latent, 2 DDIM steps, whole pipeline"},
            "sample": ("oracle fp32 on host cores: configs[0] end to end %.1fs (one run); full shapes: 1 DiT layer-call (B=2,N=17776) %.1fs, 1 LLM decode "
                       "layer %.4fs (median of 5), 1 TiTok layer %.1fs, 1 level-0 VAE resblock (2 frames 480x720) %.1fs = %.2f TFLOP/s (medians of 3); "
                       "extrapolated x(45x50), x(24x1244), x12, 315 TFLOP -> %.0f s/video"
                       % (out["config0_s"], out["dit_layer_s"], out["llm_layer_s"], out["titok_layer_s"], out["vae_resblock_s"],
                          out["vae_tflops"], total))}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one (which has not touched the
    GPU and never will -- an exec of a GPU-initialised process is what the pool forbids), one per GPU, through
    torch.distributed.run on the loopback address and a free port.  The children inherit stdout / stderr, so rank 0's JSON line is
    this command's JSON line; the return code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))       # torchrun would force 1: starves the CPU-side packing
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"bench.py: no launcher in the environment, starting {n} rank(s): {' '.join(cmd)}", file=sys.stderr, flush=True)
    if os.environ.get("LD_BENCH_DRY_SPAWN") == "1":          # (tests on a box without GPUs: show the command, start nothing)
        return 0
    return subprocess.run(cmd, env=env).returncode


def host_cpu() -> dict:
    """Model string and counts of the host CPU the baseline ran on (BASELINE.md section 3 asks for them next to the number)."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "cores_allowed": aff}


def llm_step_bytes(c) -> float:
    """Weight bytes one AR decode step must stream (bf16 blocks + the fp32 head), SURVEY 8d: 4.06 GB at the shipped config."""
    per_layer = (3 * c.hidden * c.hidden + c.hidden * c.hidden + 3 * c.mlp * c.hidden) * 2
    return c.num_layers * per_layer + c.vocab * c.hidden * 4


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
    ap.add_argument("--with-t5", action="store_true",
                    help="SECONDARY line (SURVEY 8d: 'end-to-end incl. both T5 encoders'): every step also runs the two text encoders "
                         "on the MI355X kernels -- FLAN-T5-XXL over 64 prompt tokens (LLM condition) and T5-v1.1-XXL over 226 padded "
                         "tokens (DiT context), random-init weights at true shapes -- and feeds their states to the pipeline; the "
                         "stage appears as `t5` in stage_seconds_rank0.  Not the BASELINE metric (which starts at the encoder outputs)")
    args = ap.parse_args()
    if args.with_t5 and (args.tiny or args.stream_chunks or args.prompts_per_step > 1):
        raise SystemExit("--with-t5 applies to the headline workload only")
    if args.prompts_per_step > 1 and (args.stream_chunks or args.fp8_gemm or args.tiny):
        raise SystemExit("--prompts-per-step applies to the headline workload only")
    if args.fp8_gemm and not args.stream_chunks:
        raise SystemExit("--fp8-gemm belongs to the streaming long-video configuration (BASELINE configs[4]): add --stream-chunks N")

    if "RANK" not in os.environ and (args.gpus > 1 or os.environ.get("LD_BENCH_FORCE_DIST") == "1"):
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but this process is rank {rank} of WORLD_SIZE={world}: the launcher's --nproc-per-node "
                         "and --gpus must agree (plain `python bench.py --gpus N` starts its own N ranks)")
    import torch.distributed as dist
    from landiff_amd.pipeline import gather_rank_reports, pin_rank_cores
    my_cores = pin_rank_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else None
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

    t5 = None
    if args.with_t5:
        from landiff_amd.t5 import T5Config, T5EncoderRunner, random_state
        tcfg = T5Config()
        t5 = {"flan": T5EncoderRunner(random_state(tcfg, 77, dev), tcfg, dev), "v11": T5EncoderRunner(random_state(tcfg, 78, dev), tcfg, dev)}
        gi = torch.Generator().manual_seed(4242 + rank)
        ids_llm = torch.randint(0, tcfg.vocab, (64,), generator=gi).to(dev)
        ids_dit = torch.randint(0, tcfg.vocab, (cfg.dit.text_len,), generator=gi).to(dev)      # 226, padded, unmasked (FrozenT5Embedder)

    def one_step():
        nonlocal inp
        if t5 is not None:
            t_0 = time.perf_counter()
            inp = dataclasses.replace(inp, llm_text_emb=t5["flan"].encode(ids_llm).float(), dit_context=t5["v11"].encode(ids_dit)[None].float())
            torch.cuda.synchronize()
            pipe.timings["t5"] = pipe.timings.get("t5", 0.0) + time.perf_counter() - t_0
        if P > 1:
            frames = torch.cat(pipe.generate_many(inps), dim=0)          # [P * 49, H, W, 3]
        else:
            frames = pipe.generate_stream(inp, stream, prefix_frames=prefix) if stream else pipe(inp)
        gathered = gather_frames(frames[None], world, force=use_dist)
        return gathered

    for _ in range(args.warmup):
        one_step()
    sampler = ClockPowerSampler(dev)
    calib = calibrate(dev, sampler)           # EVERY rank: the pool's GPUs differ by +-5 % under the power cap (per_rank.calibration)
    pipe.timings = {}
    pipe.dit.attn_events = []                 # HIP events around every attention launch: the roofline object's `achieved`
    pipe.dit.gemm_events = None               # (the GEMM events are collected in a separate, untimed step below)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    clocks = sampler.stop()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    n_frames = out[0].shape[1]
    stage_s = {k: v / args.steps for k, v in pipe.timings.items()}
    gemm_step_s = None
    if rank == 0 and P == 1 and not args.tiny and (not stream or args.fp8_gemm):      # (bf16 streaming runs: the headline line carries the figure)
        # one more step OUTSIDE the timed region with HIP events around every large DiT Linear (~20 k event records per video
        # would otherwise sit in the headline number)
        keep_t, keep_a, keep_o = pipe.timings, pipe.dit.attn_events, pipe.dit.overlap
        pipe.timings, pipe.dit.attn_events, pipe.dit.gemm_events = {}, None, []
        pipe.dit.overlap = False                  # serial step: every GEMM launch has the GPU to itself while it is timed
        pipe(inp)                                 # rank 0 only: NOT one_step(), whose frame gather is a collective every rank must enter
        torch.cuda.synchronize()
        gemm_step_s = dict(pipe.timings)
        pipe.timings, pipe.dit.attn_events, pipe.dit.overlap = keep_t, keep_a, keep_o
    reports = gather_rank_reports({"rank": rank, "frames_per_s": round(n_frames * args.steps / local_elapsed, 4),
                                   "stage_seconds": {k: round(v, 3) for k, v in stage_s.items()},
                                   "cores": len(my_cores) if my_cores else None,
                                   "calibration": {"mfma_tflops": calib["mfma_tflops"], "hbm_gbs": calib["hbm_gbs"],
                                                   "mfma_loop_sclk_mhz": calib["mfma_loop_sclk_mhz"],
                                                   "timed_region_sclk_mhz": clocks["mean_sclk_mhz"] if clocks else None,
                                                   "timed_region_power_w": clocks["mean_power_w"] if clocks else None}},
                                  world if use_dist else 1)
    if rank == 0:
        d = cfg.dit
        ev_all = pipe.dit.attn_events
        # With the control and the main chain overlapped on two streams (the default), a launch that has a partner shares the GPU
        # with it and its event-to-event time is not the kernel's own: the kernel's duration is taken from the launches that run
        # alone -- the main layers behind the last control state, 14 of a step's 45 -- all of them inside the timed region.
        ev = [(a, b) for a, b, solo in ev_all if solo]
        solo_rule = pipe.dit.overlap
        if not ev:                    # (tiny configs: no main layer lies behind the last control state) every launch, and say so
            ev, solo_rule = [(a, b) for a, b, _ in ev_all], False
        attn_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        attn_all_ms = sum(a.elapsed_time(b) for a, b, _ in ev_all) / max(len(ev_all), 1)
        flops = 4.0 * 2 * d.heads * d.seq_len * d.seq_len * d.head_dim      # algorithmic FLOPs of one launch
        achieved = flops / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
        peak = MFMA_BF16_PEAK
        from landiff_amd import _lib
        kname = (_lib.load().ld_attn_last_kernel() or b"").decode()      # what the launcher actually ran (shape + LD_ATTN_* knobs)
        # the PMC traffic figure was collected on the default kernel at the headline shape only (profiles/, see ATTN_TRAFFIC)
        headline = (not args.tiny and not stream and not args.fp8_gemm and kname == ATTN_TRAFFIC["kernel"]
                    and not any(os.environ.get(k) for k in ("LD_ATTN_NW", "LD_ATTN_SAFE", "LD_ATTN_VARIANT", "LD_ATTN_MSUM", "LD_ATTN_Q64")))
        attn_roof = {"kernel": "%s (DiT joint text+video attention, B=2,H=%d,N=%d,D=64)" % (kname, d.heads, d.seq_len),
                     "bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4),
                     "traffic": ATTN_TRAFFIC["bytes"] if headline else None, "traffic_source": ATTN_TRAFFIC["source"] if headline else None,
                     "launches": len(ev), "launches_total": len(ev_all),
                     "avg_launch_ms": round(attn_ms, 4), "avg_launch_ms_all_incl_overlapped": round(attn_all_ms, 4),
                     "overlap": ("control chain on a second stream (LD_DIT_OVERLAP=1): `achieved` is over the launches that run alone"
                                 if solo_rule else
                                 "control chain on a second stream and NO launch runs alone in this configuration: `achieved` is over all launches, "
                                 "whose event intervals include time shared with the other chain" if pipe.dit.overlap else
                                 "serial step: every launch runs alone"),
                     # the runner's per-layer choice between the default launch and ld_attn_fwd_bf16_exact (landiff_amd/dit.py):
                     # with these weights no layer leaves the fast pass's window, so every launch above is the default kernel
                     "attn_policy": {"mode": "auto" if pipe.dit.attn_auto else ("exact" if pipe.dit.attn_exact else "default"),
                                     "layers_on_exact_form": len(pipe.dit.exact_layers)}}
        # ---- per-stage achieved vs peak (rank 0) ----------------------------------------------------
        stages = {"dit_attention": {k: attn_roof[k] for k in ("bound", "achieved", "peak", "unit", "frac")}}
        stages["dit_attention"]["seconds_per_step"] = round(attn_ms * 1e-3 * len(ev_all) / args.steps, 3)
        stages["dit_attention"]["what"] = ("the kernel's own duration (launches that run alone) x all launches of a step: GPU time, not wall time -- "
                                           "overlapped launches take longer on the wall (avg_launch_ms_all_incl_overlapped)")
        gev = pipe.dit.gemm_events
        if gev:
            # events of the e4m3 linears carry (flops, "mx"), the bf16 ones (all of them in the headline configuration; the control
            # zero-linears under --fp8-gemm) plain flops
            split = {"bf16": [0.0, 0.0, 0], "mx": [0.0, 0.0, 0]}
            for a, b, f in gev:
                kind = "bf16"
                if isinstance(f, tuple):
                    f, kind = f
                acc = split[kind]
                acc[0] += a.elapsed_time(b) * 1e-3; acc[1] += f; acc[2] += 1
            what = ("HIP events around every qkv / dense / 4h / 4h->h Linear and control zero-linear of the DiT loop "
                    "(2 M N K each; 3.145 TFLOP per layer-call + 0.262 per control layer), tail launches included; "
                    "collected in ONE extra, serial step after the timed region (the headline loop carries no GEMM events)")
            if split["bf16"][2]:
                g_s, g_fl, n_ = split["bf16"]
                g_ach = g_fl / g_s / 1e12
                stages["dit_gemm"] = {"bound": "mfma", "achieved": round(g_ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(g_ach / peak, 4),
                                      "seconds_per_step": round(g_s, 3), "launches": n_,
                                      "what": what if not args.fp8_gemm else "the bf16 linears left under --fp8-gemm (control zero-linears); " + what}
            if split["mx"][2]:
                g_s, g_fl, n_ = split["mx"]
                g_ach = g_fl / g_s / 1e12
                stages["dit_gemm_mxfp8"] = {"bound": "mfma", "achieved": round(g_ach, 1), "peak": 2 * peak, "unit": "TFLOP/s", "frac": round(g_ach / (2 * peak), 4),
                                            "seconds_per_step": round(g_s, 3), "launches": n_,
                                            "what": "the e4m3 (MXFP8) qkv / dense / 4h / 4h->h linears against the dense MX-fp8 MFMA peak (~5 PFLOP/s); " + what}
        if "detokenize" in stage_s and not stream and P == 1 and not args.tiny:
            dtf = detok_tflop(cfg)
            stages["detokenize"] = {"bound": "mfma", "achieved": round(dtf / stage_s["detokenize"], 1), "peak": peak, "unit": "TFLOP/s",
                                    "frac": round(dtf / stage_s["detokenize"] / peak, 4), "seconds_per_step": round(stage_s["detokenize"], 4),
                                    "what": "%.2f TFLOP (TiTok decoder linears + frame-masked attention at its mask density, conv upsampler, conv_out) "
                                            "/ the detokenize stage's wall time" % dtf}
        if "llm" in stage_s and not stream and P == 1 and not args.tiny:
            steps_llm = 1244
            gbs = llm_step_bytes(cfg.llm) * steps_llm / stage_s["llm"] / 1e9
            stages["llm_decode"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK, "unit": "GB/s", "frac": round(gbs / HBM_PEAK, 4),
                                    "seconds_per_step": round(stage_s["llm"], 3),
                                    "what": "%d decode steps x %.3f GB of weights (bf16 blocks + fp32 head) / the llm stage's wall time (prefill included)"
                                            % (steps_llm, llm_step_bytes(cfg.llm) / 1e9)}
        if "vae" in stage_s and not stream and P == 1 and not args.tiny:
            tf = VAE_TFLOP / stage_s["vae"]
            stages["vae_decode"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                                    "seconds_per_step": round(stage_s["vae"], 3),
                                    "what": "315 TFLOP of causal 3D / 2D convolutions per 49-frame video / the vae stage's wall time (norm, upsample and uint8 passes included)"}
        serving = P > 1
        res = {
            "metric": (METRIC + " [SECONDARY: incl. both T5-XXL text encoders on the GPU, SURVEY 8d]") if args.with_t5 else METRIC if not serving else METRIC + " [SERVING MODE: %d prompts per GPU and step, AR decode overlapped -- not the BASELINE configuration]" % P,
            "value": world * n_frames * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.fp8_gemm else f"bf16 with fp8-e4m3 ({args.fp8_gemm}) DiT linears (configs[4], not the headline precision)",
            "data": "synthetic",
            "serving_mode": serving,
            "config": {"workload": ("tiny random-init plumbing config" if args.tiny else
                                    "LanDiff 5B full pipeline, single prompt per GPU, 49f 480x720, 50 sampler steps "
                                    "(VPSDE DPM-Solver++(2M), DynamicCFG), bf16, random-init weights at true shapes")
                                   + (f"; streaming long video: {stream} chunks, {prefix} prefix latent frames pinned per later "
                                      f"chunk, {n_frames} frames" if stream else "")
                                   + (f"; fp8 e4m3 MFMA ({args.fp8_gemm} scaling) for the DiT qkv/dense/4h/4h->h linears" if args.fp8_gemm else "")
                                   + ("; PLUS both text encoders per step (FLAN-T5-XXL over 64 tokens, T5-v1.1-XXL over 226), random-init at true shapes" if args.with_t5 else "")
                                   + (f"; SERVING MODE, not the BASELINE configuration: {P} prompts per GPU and step, AR decode of prompt "
                                      f"i+1 overlapped with the DiT loop of prompt i (generate_many)" if serving else ""),
                       "frames": n_frames, "height": 8 * d.latent_h, "width": 8 * d.latent_w,
                       "sampler_steps": cfg.sampler.num_steps, "llm_steps": (1244 if not stream else None) if not args.tiny else None,
                       "prompts_per_step": P,
                       "parallelism": f"dp{world} over prompts, RCCL all_gather of uint8 frames only"},
            "roofline": attn_roof,
            "roofline_stages": stages,
            "stage_figures_note": "llm_decode / vae_decode / detokenize divide the stage's algorithmic bytes or FLOPs by its WALL time "
                                  "(prefill, norm, upsample, uint8 passes and launch gaps included); dit_attention / dit_gemm are HIP-event "
                                  "kernel times",
            "calibration": dict(calib or {}, **({"timed_region": clocks} if clocks else {})),
        }
        if calib:                # the same fractions against what THIS box sustained a few seconds earlier
            boxc = {"dit_attention": round(achieved / calib["mfma_tflops"], 4)}
            if "dit_gemm" in stages:
                boxc["dit_gemm"] = round(stages["dit_gemm"]["achieved"] / calib["mfma_tflops"], 4)
            if "vae_decode" in stages:
                boxc["vae_decode"] = round(stages["vae_decode"]["achieved"] / calib["mfma_tflops"], 4)
            if "llm_decode" in stages:
                boxc["llm_decode"] = round(stages["llm_decode"]["achieved"] / calib["hbm_gbs"], 4)
            res["frac_of_box_ceiling"] = boxc
        if not serving:            # generate_many records no per-stage wall times (its stages overlap)
            res["stage_seconds_rank0"] = {k: round(v, 3) for k, v in stage_s.items()}
        if world > 1 or use_dist:
            fps = [r["frames_per_s"] for r in reports]
            res["n_ranks_seen"] = dist.get_world_size()          # what RCCL itself says the job size was
            # What a rank's GPU sustained right before its timed region, so that a scaling curve can be read net of the pool's
            # GPU-to-GPU spread (+-5 % under the power cap): frames_per_s_normalised = frames_per_s x (mean MFMA-only rate of the
            # job's GPUs / this GPU's) -- a rank on a slow GPU is not a scaling loss.  Each rank's own wall time, before the MAX.
            cal = [r["calibration"] for r in reports]
            mean_mfma = sum(c["mfma_tflops"] for c in cal) / len(cal)
            res["per_rank"] = {"frames_per_s_min": min(fps), "frames_per_s_max": max(fps), "frames_per_s": fps,
                               "frames_per_s_normalised": [round(f * mean_mfma / c["mfma_tflops"], 4) for f, c in zip(fps, cal)],
                               "calibration": cal, "calibration_mean_mfma_tflops": round(mean_mfma, 1),
                               "stage_seconds": [r["stage_seconds"] for r in reports], "cores_per_rank": [r["cores"] for r in reports]}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(PipelineConfig.full())
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
