"""Helpers of landiff/utils.py that callers of the entry point use (seed, frame conversion, mp4 write)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch


def set_seed_for_single_process(seed: int):
    """landiff/utils.py:409-414."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)


def cthw_to_numpy_images(video: torch.Tensor) -> np.ndarray:
    """landiff/utils.py:327-331: [C,T,H,W] in [0,1] -> uint8 [T,H,W,C] by truncation."""
    assert video.dim() == 4, "video must be 4D tensor"
    images = video.permute(1, 2, 3, 0) * 255
    return images.clip(0, 255).cpu().numpy().astype(np.uint8)


def save_video_tensor(video: torch.Tensor, video_path: str, fps: int = 8):
    """landiff/utils.py:334-342.  Needs imageio + imageio-ffmpeg like the reference; without them the frames are
    written next to the requested path as <name>.frames.npy and an ImportError explains what is missing."""
    images = cthw_to_numpy_images(video) if video.dtype != torch.uint8 else video.cpu().numpy()
    path = Path(video_path)
    path.parent.mkdir(parents=True, exist_ok=True)
    try:
        import imageio
    except ImportError as e:
        np.save(path.with_suffix(".frames.npy"), images)
        raise ImportError(f"imageio is not installed: wrote uint8 frames to {path.with_suffix('.frames.npy')} instead of mp4") from e
    with open(path, "wb") as f:
        with imageio.get_writer(f, format="mp4", fps=fps) as writer:
            for image in images:
                writer.append_data(image)
