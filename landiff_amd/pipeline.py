"""Per-prompt pipeline on one MI355X and prompt-batch data parallelism over the GPUs of a node.

Stage order of landiff/infer_video.py:105-114: AR token decode (ArModelInferWrapper, landiff/llm/llm_infer.py:74-105)
-> CogWrapper.forward (landiff/diffusion/dif_infer.py:152-243): semantic condition once per video, 50 sampler steps over
the control+main DiT, chunked VAE decode, post-process.  Unlike the reference nothing leaves the device between the
stages (no .npy round trip of the tokens, no model .cpu()/.cuda() shuffling, VAE caches in HBM).
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

import torch

from . import _lib
from .config import PipelineConfig
from .detokenizer import Detokenizer
from .dit import ControlDiTRunner
from .llm import LLMRunner
from .sampler import DiffusionSampler
from .vae import VAEDecoder


@dataclass
class PromptInputs:
    """What the boundary hands to the hot path for one prompt (T5 encoders are outside the path)."""
    llm_text_emb: torch.Tensor        # [n, text_dim]   FLAN-T5-XXL states of the prompt (LLM condition)
    dit_context: torch.Tensor         # [1, text_len, text_dim]  T5-v1.1-XXL states, padded to text_len
    seed: int = 42
    cfg: float = 7.5
    motion_score: float = 0.1


class LanDiffPipeline:
    def __init__(self, cfg: PipelineConfig, states: dict, device="cuda:0", max_llm_frames: int | None = None,
                 fp8_gemm: bool = False):
        if not torch.cuda.is_available():
            raise _lib.LandiffHipError("LanDiffPipeline needs an MI355X GPU: there is no CPU fallback")
        _lib.load()
        self.cfg = cfg.check()
        self.dev = torch.device(device)
        torch.cuda.set_device(self.dev)
        # max_llm_frames > segment_length sizes the KV cache / position tables for multi-segment (streaming) decodes
        self.llm = LLMRunner(states["llm"], cfg.llm, self.dev,
                             max_frames=max_llm_frames or cfg.llm.segment_length) if "llm" in states else None
        self.detok = Detokenizer(states["tok"], states["ups"], cfg.tok, cfg.ups, self.dev)
        # fp8_gemm: BASELINE configs[4] (e4m3 operands for the DiT's four large linears); never the headline configuration
        self.dit = ControlDiTRunner(states["dit_main"], states["dit_control"], cfg.dit, self.dev, fp8_gemm=fp8_gemm)
        self.sampler = DiffusionSampler(cfg.sampler)
        self.vae = VAEDecoder(states["vae"], cfg.vae, self.dev)
        self.timings = {}

    def _t(self, name, t0):
        torch.cuda.current_stream(self.dev).synchronize()      # (the stage's own stream: an overlapped AR decode keeps running)
        self.timings[name] = self.timings.get(name, 0.0) + time.perf_counter() - t0

    @torch.no_grad()
    def generate_tokens(self, inp: PromptInputs) -> torch.Tensor:
        """ArModelInferWrapper.forward: set_seed_for_single_process(seed) then Semantic1DLM.sample(seed=seed)."""
        torch.manual_seed(inp.seed)
        torch.cuda.manual_seed(inp.seed)
        return self.llm.sample(inp.llm_text_emb, motion_score=inp.motion_score, num_frames=self.cfg.llm.segment_length,
                               guidance_scale=inp.cfg, temperature=1.0, seed=inp.seed)

    @torch.no_grad()
    def generate_latent(self, tokens: torch.Tensor, inp: PromptInputs, noise: torch.Tensor | None = None) -> torch.Tensor:
        """CogWrapper.forward up to model.sample: seeds, semantic condition, sampler.  Returns [1,T,C,h,w] fp32."""
        d = self.cfg.dit
        torch.manual_seed(inp.seed)
        torch.cuda.manual_seed(inp.seed)
        t0 = time.perf_counter()
        sem = self.detok.semantic_condition(tokens.to(self.dev))
        self.dit.set_condition(inp.dit_context, sem)
        self._t("detokenize", t0)
        t0 = time.perf_counter()
        if noise is None:
            noise = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=self.dev, dtype=torch.float32)
        z = self.sampler.run(self.dit.step, noise)
        self._t("dit", t0)
        return z

    @torch.no_grad()
    def decode(self, latent: torch.Tensor, want_float: bool = False):
        t0 = time.perf_counter()
        lat = latent.to(torch.bfloat16).float()        # samples.to(self.dtype) (diffusion_video.py:314)
        out = self.vae.decode(lat, want_float=want_float)
        self._t("vae", t0)
        return out

    @torch.no_grad()
    def __call__(self, inp: PromptInputs, want_float: bool = False):
        """prompt embeddings -> uint8 frames [4T-3, H, W, 3] in device memory (BASELINE metric region)."""
        t0 = time.perf_counter()
        tokens = self.generate_tokens(inp)
        self._t("llm", t0)
        z = self.generate_latent(tokens, inp)
        return self.decode(z, want_float=want_float)


    # ---- several prompts on one GPU: prompt-level software pipeline ---------------------------------------
    @torch.no_grad()
    def generate_many(self, inputs: list, want_float: bool = False) -> list:
        """Frames for a list of prompts, each identical to what `self(inp)` returns for it, with the AR token decode of
        prompt i+1 issued from a helper thread on a second, high-priority HIP stream while prompt i is in detokenize / DiT /
        VAE on the current stream.

        Why it pays on MI355X: the DiT loop runs at the package power cap (1390 W) and is bound by energy, the AR decode
        is a chain of short HBM-bound launches at 980 W; next to each other the decode costs only its incremental energy
        (measured: 16.9 s for the pair of stages instead of 1.67 + 16.1 s, tools/overlap_prompts_probe.py).  No kernel
        changes: the hardware interleaves the decode's workgroups as the MFMA kernels' workgroups retire.

        RNG: the decode draws from its own generator seeded with the prompt's seed (as Semantic1DLM.sample(seed=...) does,
        lm_model.py:398-402) and never touches the default generator, which stays with the sampler of the prompt in flight --
        so per-prompt results do not depend on the overlap.  Stage timings are not recorded here (they would need
        device-wide synchronisation between the stages)."""
        from concurrent.futures import ThreadPoolExecutor
        if not inputs:
            return []
        for inp in inputs:
            assert inp.seed, "generate_many needs a non-zero seed per prompt (a zero seed would draw from the shared default generator)"
        side = torch.cuda.Stream(device=self.dev, priority=-1)
        main = torch.cuda.current_stream(self.dev)

        def decode_tokens(inp):
            torch.cuda.set_device(self.dev)
            with torch.cuda.stream(side):
                tok = self.llm.sample(inp.llm_text_emb, motion_score=inp.motion_score, num_frames=self.cfg.llm.segment_length,
                                      guidance_scale=inp.cfg, temperature=1.0, seed=inp.seed,
                                      mode="chain").clone()   # (the runner reuses its token buffer; mode="chain": the decode runs under the DiT loop)
                tok.record_stream(main)               # allocated on the side stream, read on the main one: keep the block until that read is done
                side.synchronize()                    # the tokens are complete before any other stream reads them
            return tok

        def submit(pool, inp):
            # The decode reads the prompt embedding (and the LLM's buffers) on the side stream: order it after everything the
            # caller has queued on the current stream so far -- e.g. a text encoder that has just produced inp.llm_text_emb.
            side.wait_stream(main)
            return pool.submit(decode_tokens, inp)

        out = []
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = submit(pool, inputs[0])
            for i, inp in enumerate(inputs):
                tokens = fut.result()
                if i + 1 < len(inputs):
                    fut = submit(pool, inputs[i + 1])
                d = self.cfg.dit
                torch.manual_seed(inp.seed)
                torch.cuda.manual_seed(inp.seed)
                sem = self.detok.semantic_condition(tokens.to(self.dev))
                self.dit.set_condition(inp.dit_context, sem)
                noise = torch.randn(1, d.latent_frames, d.in_channels, d.latent_h, d.latent_w, device=self.dev, dtype=torch.float32)
                z = self.sampler.run(self.dit.step, noise)
                out.append(self.vae.decode(z.to(torch.bfloat16).float(), want_float=want_float))
        return out

    @torch.no_grad()
    def generate_batch(self, inputs: list, rank: int = 0, world: int = 1) -> list:
        """A batch of prompts data-parallel over the GPUs of a node (BASELINE configs[3]; the reference's only hook is
        LOCAL_RANK -> set_device, infer_video.py:109-110): this rank runs prompts rank, rank + world, ... through
        generate_many (per-prompt seeds, so a prompt's frames do not depend on the number of GPUs) and one RCCL all_gather
        returns every prompt's uint8 frames, in prompt order, on every rank."""
        mine = shard_prompts(len(inputs), rank, world)
        local = self.generate_many([inputs[i] for i in mine])
        return gather_prompt_frames(local, len(inputs), rank, world)

    # ---- streaming long video (BASELINE config 3) --------------------------------
    def stream_plan(self, n_chunks: int, prefix_frames: int):
        """(latent frames per chunk, new latent frames per later chunk, LLM segments needed)."""
        T, seg = self.cfg.dit.latent_frames, self.cfg.llm.segment_length
        new = T - prefix_frames
        assert 0 < prefix_frames < T and new % 2 == 0, "later chunks decode their new latent frames in pairs"
        total = T + (n_chunks - 1) * new
        return T, new, -(-total // seg)

    @torch.no_grad()
    def generate_stream(self, inp: PromptInputs, n_chunks: int, prefix_frames: int = 7, want_float: bool = False,
                        tokens: torch.Tensor | None = None, noises=None, randn_like=torch.randn_like, overlap_decode: bool = True,
                        latents_out: list | None = None):
        """Chunked long-video generation out of the reference's streaming primitives -- the reference ships the pieces
        (yaml :213,231 "fixed_frames: 7 # 49 frames, 13 latent, prefix_length=7"), not the loop:
          * ONE multi-segment AR decode (Semantic1DLM.sample with num_frames = n_seg * segment_length, lm_model.py:278-291,
            361-396): the KV cache carries over the segments, tokens never leave the device;
          * per segment detokenize + upsample -> semantic features for all latent frames, resident in HBM;
          * chunk c > 0: CogWrapper.forward(vae_feature_prefix = last `prefix_frames` latents of chunk c-1)
            (dif_infer.py:159,232; diffusion_video.py:287-288) with the sampler pinning those frames
            (sampling.py:800-835), conditioned on the semantic-feature window of its 13 latent frames;
          * the new latent frames are decoded against the VAE's causal-conv caches of the previous chunk
            (cp_enc_dec.py:436-466), which stay in HBM.
        Chunk c is seeded with inp.seed + c.  Returns uint8 frames [4T-3 + (n_chunks-1)*4*new, H, W, 3] (+ fp32 video).
        latents_out: a list that receives every chunk's sampled latent [1, T, C, h, w] (fp32 values of the bf16 result), for tests."""
        d, lc = self.cfg.dit, self.cfg.llm
        T, new, n_seg = self.stream_plan(n_chunks, prefix_frames)
        per_seg = self.cfg.tok.num_latent_tokens
        t0 = time.perf_counter()
        seg_tokens = None          # segment s -> its token ids, int64 [per_seg] on the device
        decode_thread = None
        if tokens is None and overlap_decode and inp.seed:
            # The multi-segment AR decode runs on a second, high-priority stream issued by a helper thread; chunk c only needs
            # the segments its 13 latent frames fall into, so the DiT loop of the first chunk starts as soon as segment 0 is
            # decoded and the later segments are decoded underneath it (980 W of HBM-bound launches next to the power-capped
            # MFMA kernels: see generate_many).  The decode draws from its own generator (seed given), so the tokens equal
            # those of the serial path.
            import threading
            side = torch.cuda.Stream(device=self.dev, priority=-1)
            queued = [threading.Event() for _ in range(n_seg)]
            done = [torch.cuda.Event() for _ in range(n_seg)]
            def on_segment(sidx):
                done[sidx].record(side)
                queued[sidx].set()
            decode_state = {"t_start": time.perf_counter()}
            def run_decode():
                try:
                    torch.cuda.set_device(self.dev)
                    with torch.cuda.stream(side):
                        self.llm.sample(inp.llm_text_emb, motion_score=inp.motion_score, num_frames=n_seg * lc.segment_length,
                                        guidance_scale=inp.cfg, temperature=1.0, seed=inp.seed, on_segment=on_segment, segment_tokens=per_seg,
                                        mode="chain")
                        side.synchronize()
                    decode_state["seconds"] = time.perf_counter() - decode_state["t_start"]
                except BaseException as e:              # never leave the consumer waiting on a segment that will not come
                    decode_state["error"] = e
                    for ev in queued:
                        ev.set()
            side.wait_stream(torch.cuda.current_stream(self.dev))     # the decode reads inp.llm_text_emb: after whatever produced it
            decode_thread = threading.Thread(target=run_decode)
            decode_thread.start()
            def seg_tokens(sidx):
                queued[sidx].wait()
                if "error" in decode_state:
                    raise decode_state["error"]
                torch.cuda.current_stream(self.dev).wait_event(done[sidx])
                return self.llm.out_tokens[sidx * per_seg:(sidx + 1) * per_seg].clamp(0, lc.visual_vocab - 1)
        else:
            if tokens is None:
                torch.manual_seed(inp.seed); torch.cuda.manual_seed(inp.seed)
                tokens = self.llm.sample(inp.llm_text_emb, motion_score=inp.motion_score, num_frames=n_seg * lc.segment_length,
                                         guidance_scale=inp.cfg, temperature=1.0, seed=inp.seed)
            self._t("llm", t0)
            tokens = tokens.to(self.dev).reshape(n_seg, per_seg)
            seg_tokens = lambda sidx: tokens[sidx]
        sem_seg = {}               # segment -> semantic features [T, C, H, W], computed when a chunk first needs them
        def sem_window(f0, f1):    # latent frames [f0, f1) of the concatenated per-segment features
            parts = []
            for sidx in range(f0 // T, (f1 - 1) // T + 1):
                if sidx not in sem_seg:
                    tok = seg_tokens(sidx)                 # (overlapped decode: blocks until the segment's last token is queued)
                    t1 = time.perf_counter()
                    sem_seg[sidx] = self.detok.semantic_condition(tok)
                    self._t("detokenize", t1)
                lo, hi = max(f0, sidx * T) - sidx * T, min(f1, (sidx + 1) * T) - sidx * T
                parts.append(sem_seg[sidx][lo:hi])
            return torch.cat(parts, dim=0).contiguous()
        outs, vids, prev = [], [], None
        for c in range(n_chunks):
            t_w = time.perf_counter()
            d0 = self.timings.get("detokenize", 0.0)
            sem = sem_window(c * new, c * new + T)             # (may wait on the overlapped decode; detokenize is timed inside)
            wait = time.perf_counter() - t_w - (self.timings.get("detokenize", 0.0) - d0)
            if decode_thread is not None:
                self.timings["segment_wait"] = self.timings.get("segment_wait", 0.0) + wait
            t0 = time.perf_counter()
            torch.manual_seed(inp.seed + c); torch.cuda.manual_seed(inp.seed + c)
            self.dit.set_condition(inp.dit_context, sem)
            noise = noises[c].to(self.dev) if noises is not None else torch.randn(
                1, T, d.in_channels, d.latent_h, d.latent_w, device=self.dev, dtype=torch.float32)
            if c == 0:
                z = self.sampler.run(self.dit.step, noise, randn_like=randn_like)
            else:
                z = self.sampler.run(self.dit.step, noise, randn_like=randn_like, prefix=prev[:, T - prefix_frames:],
                                     fixed_frames=prefix_frames)
            prev = z.to(torch.bfloat16).float()              # samples.to(self.dtype) (diffusion_video.py:314)
            if latents_out is not None:
                latents_out.append(prev.clone())
            self._t("dit", t0)
            t0 = time.perf_counter()
            lat = prev if c == 0 else prev[:, prefix_frames:]
            r = self.vae.decode(lat, want_float=want_float, stream_continue=c > 0, stream_keep=c < n_chunks - 1)
            self._t("vae", t0)
            if want_float:
                outs.append(r[0]); vids.append(r[1])
            else:
                outs.append(r)
        if decode_thread is not None:
            decode_thread.join()
            if "error" in decode_state:
                raise decode_state["error"]
            self.timings["llm_overlapped"] = self.timings.get("llm_overlapped", 0.0) + decode_state["seconds"]   # wall time of the decode thread
        frames = torch.cat(outs, dim=0)
        return (frames, torch.cat(vids, dim=1)) if want_float else frames


def synthetic_inputs(cfg: PipelineConfig, device, n_text: int = 64, seed: int = 42) -> PromptInputs:
    """Synthetic prompt embeddings: N(0,1), seeds 42 / 43."""
    g = torch.Generator().manual_seed(seed)
    llm_text = torch.randn(n_text, cfg.llm.text_dim, generator=g)
    g2 = torch.Generator().manual_seed(seed + 1)
    ctx = torch.randn(1, cfg.dit.text_len, cfg.dit.text_dim, generator=g2)
    return PromptInputs(llm_text.to(device), ctx.to(device), seed=seed)


# ------------------------------------------------------------------------------------------------
# prompt-batch data parallelism (one process per GPU; the only collective is the final gather)
# ------------------------------------------------------------------------------------------------
def shard_prompts(n_prompts: int, rank: int, world: int) -> list[int]:
    """Rank r processes prompts r, r+world, ...; each keeps the user's seed, so a prompt's result
    does not depend on the number of GPUs."""
    return list(range(rank, n_prompts, world))


def gather_frames(frames: torch.Tensor, world: int, force: bool = False):
    """all_gather of uint8 frames [n_local, T, H, W, 3] over RCCL/xGMI (gloo on CPU in tests).
    force: issue the collective even for a single rank (exercises the RCCL path on a 1-GPU box)."""
    import torch.distributed as dist
    if not dist.is_initialized() or (world == 1 and not force):
        return [frames]
    out = [torch.empty_like(frames) for _ in range(world)]
    dist.all_gather(out, frames.contiguous())
    return out


def gather_prompt_frames(local_frames: list, n_prompts: int, rank: int, world: int, force: bool = False) -> list:
    """The final gather of a prompt batch (BASELINE configs[3]): rank r holds the uint8 frames [T, H, W, 3] of prompts
    shard_prompts(n_prompts, r, world), in that order; every rank gets back the frames of ALL prompts in prompt order.
    Shards may be uneven (n_prompts not a multiple of world): the short ranks pad with one zero video, which is dropped again.
    One all_gather per batch -- the only collective of the data-parallel path."""
    mine = shard_prompts(n_prompts, rank, world)
    assert len(local_frames) == len(mine), (len(local_frames), len(mine))
    per_rank = -(-n_prompts // world)
    if not local_frames and per_rank == 0:
        return []
    ref = local_frames[0] if local_frames else None
    import torch.distributed as dist
    live = dist.is_initialized() and world > 1
    if ref is not None:
        dev = ref.device
    else:                  # a rank with no prompt at all (world > n_prompts) learns the frame shape from rank 0
        dev = torch.device("cuda", torch.cuda.current_device()) if live and dist.get_backend() == "nccl" else torch.device("cpu")
    shape = torch.tensor(ref.shape if ref is not None else (0, 0, 0, 0), dtype=torch.int64, device=dev)
    if live:
        dist.broadcast(shape, src=0)
    T, H, W, C = (int(v) for v in shape.tolist())
    stack = torch.zeros(per_rank, T, H, W, C, dtype=torch.uint8, device=dev)
    for i, f in enumerate(local_frames):
        stack[i] = f
    gathered = gather_frames(stack, world, force=force)
    out = [None] * n_prompts
    for r, g in enumerate(gathered):
        for i, pid in enumerate(shard_prompts(n_prompts, r if len(gathered) > 1 else rank, world)):
            out[pid] = g[i]
    return out


def rank_core_slice(local_rank: int, local_world: int, cores: list | None = None) -> list:
    """The host cores of one rank: the cores this process may run on (its cpuset), cut into `local_world` contiguous, disjoint
    slices.  One process per GPU each runs a latency-sensitive enqueue loop (the AR decode queues ~125 launches per 1.2 ms
    step from one thread, 0.54 ms of host time per step): eight of them left to the scheduler migrate across cores and share
    caches; a fixed slice per rank keeps every rank's enqueue thread, its helper thread (generate_many / generate_stream) and
    its RCCL proxy thread on cores of their own."""
    if cores is None:
        cores = sorted(os.sched_getaffinity(0))
    n = len(cores)
    if local_world <= 1 or n < local_world:
        return list(cores)
    per = n // local_world
    return list(cores[local_rank * per:(local_rank + 1) * per])


def _parse_cpulist(text: str) -> list:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes(sysfs: str = "/sys") -> tuple:
    """(numa node of every visible AMD compute GPU in device order, {node: [cpus]}) read from sysfs WITHOUT touching HIP: the
    cards whose PCI vendor is 0x1002 AND that expose a compute hwmon (hwmon*/freq1_input: display-only parts and iGPUs without an
    sclk file are skipped), sorted by PCI address (the order the runtime enumerates them in on one node), then filtered the way the
    runtimes compose their filters: ROCR_VISIBLE_DEVICES first (it hides devices from HIP), then HIP_VISIBLE_DEVICES or, if that
    is unset, CUDA_VISIBLE_DEVICES (indices into what ROCR left).  Plain integer lists only: a UUID list, an index out of range
    or an unreadable file yields an EMPTY list -- no claim about the order, and rank_core_plan falls back to the plain core cut
    (as it does whenever the visible-card count is smaller than the local world).  A node of -1 gives None for that GPU."""
    import glob
    cards = []
    for dev in glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*/device")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
            if not glob.glob(os.path.join(dev, "hwmon", "hwmon*", "freq1_input")):
                continue                        # not a compute part (no shader clock): HIP does not enumerate it
            pci = os.path.basename(os.path.realpath(dev))
            try:
                node = int(open(os.path.join(dev, "numa_node")).read())
            except (OSError, ValueError):
                node = -1
            cards.append((pci, node if node >= 0 else None))
        except OSError:
            continue
    cards = [n for _, n in sorted(set(cards))]

    def pick(cur, var):
        v = os.environ.get(var)
        if not v:
            return cur, False
        try:
            return [cur[int(i)] for i in v.split(",")], True
        except (ValueError, IndexError):
            return [], True                     # UUID lists etc.: no claim about the order
    cards, _ = pick(cards, "ROCR_VISIBLE_DEVICES")
    cards, used_hip = pick(cards, "HIP_VISIBLE_DEVICES")
    if not used_hip:
        cards, _ = pick(cards, "CUDA_VISIBLE_DEVICES")
    cpus = {}
    for n in {c for c in cards if c is not None}:
        try:
            cpus[n] = _parse_cpulist(open(os.path.join(sysfs, f"devices/system/node/node{n}/cpulist")).read())
        except OSError:
            cpus[n] = []
    return cards, cpus


def rank_core_plan(local_world: int, cores: list, gpu_nodes: list | None = None, node_cpus: dict | None = None) -> list:
    """Core slice of EVERY local rank (a pure function of its arguments, so all ranks compute the same plan).  When the NUMA node
    of every rank's GPU is known and each node has at least as many allowed cores as ranks on it, a rank gets a contiguous share
    of the allowed cores of ITS GPU's node (ranks that share a node split it in rank order): launches, the RCCL proxy and the
    pinned prompt buffers stay on the socket the GPU hangs off.  Otherwise the plain cut of the cpuset into `local_world`
    contiguous slices (rank_core_slice)."""
    plain = [rank_core_slice(r, local_world, cores) for r in range(local_world)]
    if not gpu_nodes or len(gpu_nodes) < local_world or any(n is None for n in gpu_nodes[:local_world]) or not node_cpus:
        return plain
    allowed = set(cores)
    by_node = {}
    for r in range(local_world):
        by_node.setdefault(gpu_nodes[r], []).append(r)
    plan = [None] * local_world
    for node, ranks in by_node.items():
        mine = [c for c in sorted(node_cpus.get(node, [])) if c in allowed]
        if len(mine) < len(ranks):
            return plain
        per = len(mine) // len(ranks)
        for i, r in enumerate(ranks):
            plan[r] = mine[i * per:(i + 1) * per]
    return plan


def pin_rank_cores(local_rank: int, local_world: int) -> list:
    """Applies this rank's slice of rank_core_plan (the cores of its GPU's NUMA node when sysfs exposes it, else the plain cut) to
    this process -- call it BEFORE the first HIP call: all threads created afterwards inherit it -- and sizes torch's intra-op pool
    to it.  Returns the slice.  LD_NO_PIN=1 leaves the affinity alone."""
    if os.environ.get("LD_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    cores = sorted(os.sched_getaffinity(0))
    try:
        nodes, cpus = gpu_numa_nodes()
    except Exception:                            # sysfs layout surprises must never stop a run
        nodes, cpus = None, None
    mine = rank_core_plan(local_world, cores, nodes, cpus)[local_rank] if local_world > 1 else cores
    if local_world > 1 and os.environ.get("LD_PIN_VERBOSE", "1") != "0":
        numa = (nodes[local_rank] if nodes and len(nodes) >= local_world else None)
        print(f"[landiff_amd] local rank {local_rank}/{local_world}: {len(mine)} host cores "
              f"({mine[0]}..{mine[-1]})" + (f" on NUMA node {numa} of its GPU" if numa is not None else " (plain cut of the cpuset: GPU NUMA nodes unknown)"),
              file=sys.stderr, flush=True) if mine else None
    if mine:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), torch.get_num_threads())))
        if local_world > 1 and len(mine) < 2:
            import warnings
            warnings.warn(f"rank {local_rank}: {len(mine)} host core(s) per rank -- the AR decode's enqueue loop (~0.5 ms of host time per "
                          "1.2 ms step) shares its core with the process's other threads; expect a lower decode rate")
    return mine


def gather_rank_reports(report: dict, world: int) -> list:
    """Every rank's small report dict (stage seconds, frames/s, ...) on every rank, in rank order: how an N > 1 bench line
    carries per-rank numbers next to the max-over-ranks time.  One all_gather_object outside the timed region."""
    import torch.distributed as dist
    if not dist.is_initialized() or world == 1:
        return [report]
    out = [None] * world
    dist.all_gather_object(out, report)
    return out
