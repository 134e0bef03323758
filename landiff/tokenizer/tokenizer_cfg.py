"""landiff/tokenizer/tokenizer_cfg.py:29-112: the shipped video-tokenizer configuration (TiTok encoder/decoder, 13 frames on
a 30x45 grid, 1218 latent tokens = 330 I + 12 x 74 P, codebook 2048 x 16) as a plain dataclass instead of a fiddle graph.  The
model YAML names this function in `config_str`; landiff_amd.config.load_diffusion_config imports it the way VQWarp.__init__
does (vq_warp.py:29-37)."""
from landiff_amd.config import TokenizerConfig


def build_tokenizer() -> TokenizerConfig:
    return TokenizerConfig()
