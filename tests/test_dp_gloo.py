"""The N>1 path on CPU: world_size-2 gloo processes shard prompts and gather uint8 frames (RCCL on the GPU node)."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from landiff_amd.pipeline import gather_frames, shard_prompts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_prompts(4, rank, world)
    # stand-in frames: a deterministic function of the prompt id (the pipeline itself needs a GPU)
    frames = torch.stack([torch.full((3, 4, 6, 3), 10 * p + 1, dtype=torch.uint8) for p in mine])
    gathered = gather_frames(frames, world)
    ok = len(gathered) == world
    for r, g in enumerate(gathered):
        ids = shard_prompts(4, r, world)
        ok &= all(bool((g[i] == 10 * p + 1).all()) for i, p in enumerate(ids))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, mine))


def _worker_uneven(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from landiff_amd.pipeline import gather_prompt_frames, shard_prompts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_prompts = 3                                                  # rank 0: prompts 0, 2; rank 1: prompt 1
    mine = shard_prompts(n_prompts, rank, world)
    local = [torch.full((3, 4, 6, 3), 10 * p + 1, dtype=torch.uint8) for p in mine]
    out = gather_prompt_frames(local, n_prompts, rank, world)
    ok = len(out) == n_prompts and all(bool((out[p] == 10 * p + 1).all()) and tuple(out[p].shape) == (3, 4, 6, 3) for p in range(n_prompts))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, mine))


def test_two_rank_uneven_prompt_batch():
    """gather_prompt_frames: 3 prompts over 2 ranks -- every rank ends up with all videos in prompt order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 200
    procs = [ctx.Process(target=_worker_uneven, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert sorted(sum((m for _, _, m in res), [])) == [0, 1, 2]


def test_two_rank_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert sorted(sum((m for _, _, m in res), [])) == [0, 1, 2, 3]


def _worker_eight(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from landiff_amd.pipeline import gather_prompt_frames, gather_rank_reports, rank_core_slice, shard_prompts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for n_prompts in (5, 11, 8, 0):       # ranks 5..7 idle; uneven 2/1 split; one each; nothing at all
        mine = shard_prompts(n_prompts, rank, world)
        local = [torch.full((2, 4, 6, 3), 7 * p + 3, dtype=torch.uint8) for p in mine]
        out = gather_prompt_frames(local, n_prompts, rank, world)
        ok &= len(out) == n_prompts
        ok &= all(tuple(out[p].shape) == (2, 4, 6, 3) and bool((out[p] == 7 * p + 3).all()) for p in range(n_prompts))
    reports = gather_rank_reports({"rank": rank, "stage_seconds": {"dit": 10.0 + rank}, "frames_per_s": 3.0 - 0.01 * rank}, world)
    ok &= [r["rank"] for r in reports] == list(range(world)) and reports[rank]["stage_seconds"]["dit"] == 10.0 + rank
    cores = list(range(64))
    slices = [rank_core_slice(r, world, cores) for r in range(world)]
    ok &= sorted(sum(slices, [])) == cores and all(len(s) == 8 for s in slices) and slices[rank] == cores[8 * rank: 8 * rank + 8]
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, None))


def test_eight_rank_uneven_batches_and_rank_reports():
    """BASELINE configs[3] geometry on CPU: 8 ranks (gloo), prompt batches that do not divide by 8 incl. ranks with no prompt and
    an empty batch, the per-rank report gather of the N > 1 bench line, and the per-rank core slices."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 200
    procs = [ctx.Process(target=_worker_eight, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert len(res) == 8 and all(ok for _, ok, _ in res)


def test_rank_core_slices_are_disjoint_and_cover():
    from landiff_amd.pipeline import rank_core_slice
    cores = [0, 1, 2, 3, 8, 9, 10, 11, 16, 17]                         # a cpuset with holes, not a multiple of the world
    s = [rank_core_slice(r, 4, cores) for r in range(4)]
    assert s == [[0, 1], [2, 3], [8, 9], [10, 11]]
    assert rank_core_slice(0, 1, cores) == cores and rank_core_slice(3, 16, cores) == cores    # fewer cores than ranks: no pinning


def _pin_worker(rank, world, q, go):
    sys.path.insert(0, ROOT)
    from landiff_amd.pipeline import pin_rank_cores
    before = sorted(os.sched_getaffinity(0))
    mine = pin_rank_cores(rank, world)
    go.wait(timeout=60)                                            # hold the pin until every rank has applied its own
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0))))     # threads created later inherit the slice
    t.start(); t.join()
    q.put((rank, before, list(mine), sorted(os.sched_getaffinity(0)), seen[0], torch.get_num_threads()))


def test_eight_ranks_pin_disjoint_core_slices_concurrently():
    """pin_rank_cores as eight LOCAL ranks of one node run it (bench.py / landiff.infer_video under torchrun): eight live
    processes at the same time, each reading the cpuset it was started in and cutting its own slice out of it -- the affinity
    masks the kernel reports back must be pairwise disjoint, lie inside the launch cpuset, be inherited by later threads,
    and bound torch's intra-op pool.  (With fewer cores than ranks nothing is pinned: every rank keeps the whole set.)"""
    world = 8
    cpuset = sorted(os.sched_getaffinity(0))
    ctx = mp.get_context("spawn")
    q, go = ctx.Queue(), ctx.Barrier(world)
    procs = [ctx.Process(target=_pin_worker, args=(r, world, q, go)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r for r, *_ in res] == list(range(world))
    for rank, before, mine, after, in_thread, nthreads in res:
        assert before == cpuset                                    # every rank starts from the launch cpuset
        assert after == in_thread == sorted(mine)                  # what the kernel reports is the slice, in new threads too
        assert set(after) <= set(cpuset) and 1 <= nthreads <= max(1, len(after))
    if len(cpuset) >= world:
        per = len(cpuset) // world
        masks = [set(after) for _, _, _, after, _, _ in res]
        assert all(len(m) == per for m in masks)
        assert all(masks[i].isdisjoint(masks[j]) for i in range(world) for j in range(i + 1, world))
        assert set().union(*masks) == set(cpuset[: per * world])
    else:
        assert all(after == cpuset for _, _, _, after, _, _ in res)
