"""Helpers of landiff/utils.py that callers of the entry point use (seed, frame conversion, mp4 write)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch


_LANDIFF_MODEL_PATH = None

# md5 of the 15 files of the released checkpoint tree (huggingface.co/yinaoxiong/LanDiff) -- the content of the reference's
# ckpts/CHECKSUM.md5, kept here as data so that a tree can be verified without that file; a ckpts/CHECKSUM.md5 in the working tree
# (same "<md5>  ./<path>" format) takes precedence.
RELEASED_MD5 = {
    "CogVideoX-2b-sat/t5-v1_1-xxl/added_tokens.json": "abe4fe2e108e8ed9432b843545238e53",
    "CogVideoX-2b-sat/t5-v1_1-xxl/config.json": "874a573dae8c4f440916bab515dc0606",
    "CogVideoX-2b-sat/t5-v1_1-xxl/model-00001-of-00002.safetensors": "221271c19cda66748bba20871b37d0e9",
    "CogVideoX-2b-sat/t5-v1_1-xxl/model-00002-of-00002.safetensors": "64fc13b5063d462a73c000b54f1be785",
    "CogVideoX-2b-sat/t5-v1_1-xxl/model.safetensors.index.json": "ebfc181cf184f4dc9cfffebaeae24d06",
    "CogVideoX-2b-sat/t5-v1_1-xxl/special_tokens_map.json": "9583322ae544dfbdb0001350ab4c3fd9",
    "CogVideoX-2b-sat/t5-v1_1-xxl/spiece.model": "9d15ef55d09d5a425ceb63fa31f7cae3",
    "CogVideoX-2b-sat/t5-v1_1-xxl/tokenizer_config.json": "ae93f8b5fb04998524bc84b3c517b05d",
    "CogVideoX-2b-sat/transformer/1000/mp_rank_00_model_states.pt": "0c5f18a7b46ab52bae4ffec8a0195e59",
    "CogVideoX-2b-sat/transformer/latest": "ad865d2f63b9feb2552c220385fbb7e3",
    "CogVideoX-2b-sat/vae/3d-vae.pt": "e8be4352ababa00ebab19501d2e932bc",
    "diffusion/1/mp_rank_00_model_states.pt": "f91f8e8eec665115061866529d1ea735",
    "diffusion/latest": "c4ca4238a0b923820dcc509a6f75849b",
    "llm/model.safetensors": "db87f2939b0e0a0f4bc74ac8a04ce920",
    "tokenizer/model.safetensors": "a8b4ed9d4061759ea4314780898767e0",
}


def verify_md5_checksum(root_dir, checksum_file=None) -> bool:
    """landiff/utils.py:23-90: every file of the checksum list ("<md5>  ./<relative path>" per line: <repo>/ckpts/CHECKSUM.md5 if
    present, else the built-in RELEASED_MD5 table of the 15 released files) exists under root_dir and has that md5.  Stops at the
    first missing / different file, printing which one; raises FileNotFoundError when a checksum file is named but missing."""
    import hashlib
    root_dir = Path(root_dir)
    default_file = Path(__file__).resolve().parents[1] / "ckpts" / "CHECKSUM.md5"
    if checksum_file is None and not default_file.exists():
        wanted = dict(RELEASED_MD5)
    else:
        checksum_file = Path(checksum_file) if checksum_file else default_file
        if not checksum_file.exists():
            raise FileNotFoundError(f"Checksum file does not exist: {checksum_file}")
        wanted = {}
        for line in checksum_file.read_text().splitlines():
            line = line.strip()
            if line:
                md5, rel = line.split("  ", 1)
                wanted[rel[2:] if rel.startswith("./") else rel] = md5
    for rel, md5 in wanted.items():
        f = root_dir / rel
        if not f.exists():
            print(f"Error: File does not exist: {f}")
            return False
        h = hashlib.md5()
        with open(f, "rb") as fh:
            for chunk in iter(lambda: fh.read(1 << 20), b""):
                h.update(chunk)
        if h.hexdigest() != md5:
            print(f"Error: File verification failed: {f}\n  Expected MD5: {md5}\n  Actual MD5: {h.hexdigest()}")
            return False
    return True


def _link_workspace(workspace_path: Path, model_path: Path):
    """ckpts/LanDiff in the working tree becomes a symlink to where the model really is (landiff/utils.py:150-169), so the
    relative checkpoint paths of the CLI defaults and the YAML files resolve.  An existing real directory is never touched."""
    if model_path == workspace_path:
        return
    if workspace_path.exists() and not workspace_path.is_symlink():
        raise FileExistsError(f"Workspace path '{workspace_path}' already exists and is not a symbolic link. Please remove or "
                              f"rename it manually to create a symbolic link to the model path '{model_path}'.")
    if workspace_path.is_symlink():
        workspace_path.unlink()
    workspace_path.parent.mkdir(parents=True, exist_ok=True)
    workspace_path.symlink_to(model_path, target_is_directory=True)
    print(f"Created symbolic link from {workspace_path} to {model_path}")


def initialize_landiff_model_path(skip_hash_verification: bool = False) -> Path:
    """landiff/utils.py:93-217: locate the checkpoint tree -- $LANDIFF_HOME, then <repo>/ckpts/LanDiff -- verify it against
    ckpts/CHECKSUM.md5 (unless skipped), link it into the working tree; if none is valid, download yinaoxiong/LanDiff from the
    Hugging Face hub, verify and link that.  The result is cached for the process."""
    global _LANDIFF_MODEL_PATH
    if _LANDIFF_MODEL_PATH is not None:
        return _LANDIFF_MODEL_PATH
    import os
    root_dir = Path(__file__).resolve().parents[1]
    workspace_path = root_dir / "ckpts" / "LanDiff"
    candidates = ([Path(os.environ["LANDIFF_HOME"])] if os.environ.get("LANDIFF_HOME") else []) + [workspace_path]
    for model_path in candidates:
        if model_path.exists() and model_path.is_dir() and (skip_hash_verification or verify_md5_checksum(model_path)):
            _LANDIFF_MODEL_PATH = model_path
            _link_workspace(workspace_path, model_path)
            return model_path
    print("No valid model path found. Will automatically download LanDiff model from Hugging Face...")
    from huggingface_hub import snapshot_download
    download_path = Path(snapshot_download(repo_id="yinaoxiong/LanDiff"))
    print(f"Model downloaded to {download_path}, performing hash verification...")
    if not (skip_hash_verification or verify_md5_checksum(download_path)):
        raise ValueError("Hash verification of the downloaded model failed. Please ensure a stable network connection, "
                         "or manually download the model and set the LANDIFF_HOME environment variable.")
    _LANDIFF_MODEL_PATH = download_path
    _link_workspace(workspace_path, download_path)
    return download_path


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
