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


def stable_hash(key: str) -> int:
    """landiff/utils.py:317-324: first 20 hex digits of sha256(key) (Python's hash() is salted per process)."""
    import hashlib
    return int(hashlib.sha256(key.encode()).hexdigest()[:20], 16)


def cthw_to_numpy_images(video: torch.Tensor) -> np.ndarray:
    """landiff/utils.py:327-331: [C,T,H,W] in [0,1] -> uint8 [T,H,W,C] by truncation."""
    assert video.dim() == 4, "video must be 4D tensor"
    images = video.permute(1, 2, 3, 0) * 255
    return images.clip(0, 255).cpu().numpy().astype(np.uint8)


def write_mjpeg_avi(images: np.ndarray, path, fps: int = 8, quality: int = 95):
    """Motion-JPEG AVI (RIFF) writer with nothing but PIL for the JPEG frames: the playable fallback when imageio-ffmpeg (the
    reference's H.264 mp4 writer, landiff/utils.py:334-342) is not installed.  images: uint8 [T, H, W, 3]."""
    import io
    import struct
    from PIL import Image
    T, H, W, _ = images.shape
    frames = []
    for img in images:
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(img), "RGB").save(buf, format="JPEG", quality=quality)
        b = buf.getvalue()
        frames.append(b + b"\0" * (len(b) & 1))                      # chunks are word aligned
    chunk = lambda tag, data: tag + struct.pack("<I", len(data)) + data + b"\0" * (len(data) & 1)
    lst = lambda tag, data: b"LIST" + struct.pack("<I", len(data) + 4) + tag + data
    maxb = max(len(f) for f in frames)
    avih = struct.pack("<14I", 1000000 // fps, maxb * fps, 0, 0x10, T, 0, 1, maxb, W, H, 0, 0, 0, 0)       # 0x10 = AVIF_HASINDEX
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, 1, fps, 0, T, maxb, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"MJPG", W * H * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi_body, idx, off = b"", b"", 4
    for f in frames:
        movi_body += b"00dc" + struct.pack("<I", len(f)) + f
        idx += b"00dc" + struct.pack("<III", 0x10, off, len(f))   # AVIIF_KEYFRAME, offset from the 'movi' tag
        off += 8 + len(f)
    body = b"AVI " + hdrl + lst(b"movi", movi_body) + chunk(b"idx1", idx)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def save_video_tensor(video: torch.Tensor, video_path: str, fps: int = 8):
    """landiff/utils.py:334-342: H.264 mp4 through imageio + imageio-ffmpeg, like the reference.  Without them (this image
    has neither, and no ffmpeg binary) the video is written next to the requested path as a Motion-JPEG <name>.avi plus the
    exact uint8 frames <name>.frames.npy, with a warning naming what is missing."""
    images = cthw_to_numpy_images(video) if video.dtype != torch.uint8 else video.cpu().numpy()
    path = Path(video_path)
    path.parent.mkdir(parents=True, exist_ok=True)
    try:
        import imageio
    except ImportError:
        import warnings
        np.save(path.with_suffix(".frames.npy"), images)
        write_mjpeg_avi(images, path.with_suffix(".avi"), fps=fps)
        warnings.warn(f"imageio / imageio-ffmpeg are not installed: wrote {path.with_suffix('.avi')} (Motion-JPEG) and "
                      f"{path.with_suffix('.frames.npy')} (exact uint8 frames) instead of {path}")
        return
    with open(path, "wb") as f:
        with imageio.get_writer(f, format="mp4", fps=fps) as writer:
            for image in images:
                writer.append_data(image)
