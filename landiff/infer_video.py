"""python -m landiff.infer_video --prompt ... : the reference's CLI (landiff/infer_video.py:12-118) on the MI355X path."""
from pathlib import Path

import numpy as np
import torch

from landiff.diffusion.dif_infer import CogModelInferWrapper, VideoTask
from landiff.llm.llm_cfg import build_llm
from landiff.llm.llm_infer import ArModelInferWrapper, ARSampleCfg, CodeTask
from landiff.utils import save_video_tensor


def parse_args(argv=None):
    import argparse

    parser = argparse.ArgumentParser(description="Landiff Video Inference")
    parser.add_argument("--prompt", type=str, help="Prompt for the video generation.")
    parser.add_argument("--llm_ckpt", type=str, default="ckpts/LanDiff/llm/model.safetensors", help="Path to the LLM checkpoint.")
    parser.add_argument("--diffusion_ckpt", type=str, default="ckpts/LanDiff/diffusion", help="Path to the diffusion checkpoint.")
    parser.add_argument("--save_file_name", type=str, default="results/video", help="Path to save the generated video.")
    parser.add_argument("--cfg", type=float, default=7.5, help="CFG scale for the video generation.")
    parser.add_argument("--motion_score", type=float, default=0.1, help="Motion score for the video generation.")
    parser.add_argument("--seed", type=int, default=42, help="Random seed for video generation.")
    return parser.parse_args(argv)


def llm_infer(args):
    llm_model_cfg = build_llm()
    llm = ArModelInferWrapper(args.llm_ckpt, llm_model_cfg)
    # one segment of semantic frames = one 49-frame clip: 13 for the shipped configuration (ARSampleCfg's default, which the
    # reference's llm_infer relies on); a configuration with another segment length (BASELINE configs[0]: 8) decodes its own
    task = CodeTask(save_file_name=f"{args.save_file_name}.npy", prompt=args.prompt, seed=args.seed,
                    sample_cfg=ARSampleCfg(temperature=1.0, cfg=args.cfg, motion_score=args.motion_score,
                                           num_frames=llm_model_cfg.segment_length))
    task = llm(task)
    tokens = task.result.reshape(-1)
    path = Path(task.save_file_name)
    path.parent.mkdir(parents=True, exist_ok=True)
    np.save(path, tokens.cpu().numpy())
    del llm
    torch.cuda.empty_cache()
    return tokens.cuda()


def infer_diffusion(args, semantic_token):
    model = CogModelInferWrapper(ckpt_path=args.diffusion_ckpt)
    task = VideoTask(save_file_name=f"{args.save_file_name}.mp4", prompt=args.prompt, seed=args.seed, fps=8,
                     semantic_token=semantic_token)
    task = model(task)
    save_video_tensor(task.result, task.save_file_name, fps=task.fps)
    print(f"save video to {task.save_file_name}")


def main():
    import os

    args = parse_args()
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > 1:      # one process per GPU (prompt-level data parallelism): own core slice per rank
        from landiff_amd.pipeline import pin_rank_cores
        pin_rank_cores(local_rank, int(os.environ["LOCAL_WORLD_SIZE"]))
    torch.cuda.set_device(local_rank)
    infer_diffusion(args, llm_infer(args))


if __name__ == "__main__":
    main()
