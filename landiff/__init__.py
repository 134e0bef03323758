"""Drop-in façade: the reference's import paths (landiff.infer_video, landiff.llm.llm_infer, landiff.diffusion.dif_infer,
landiff.utils) backed by the MI355X path in ``landiff_amd``.  Checkpoint discovery follows the reference
(landiff/__init__.py:14-50 / landiff/utils.py:129-179: $LANDIFF_HOME, then <repo>/ckpts/LanDiff); hash verification
and the HF download need network access and are left to the user (LANDIFF_SKIP_INIT / LANDIFF_SKIP_HASH_CHECK are
accepted and have nothing left to skip)."""
__version__ = "0.1.0"
