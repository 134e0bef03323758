"""Drop-in façade: the reference's import paths (landiff.infer_video, landiff.llm.llm_infer, landiff.diffusion.dif_infer,
landiff.utils) backed by the MI355X path in ``landiff_amd``.  As in the reference (landiff/__init__.py:14-50) importing the
package locates the checkpoint tree -- $LANDIFF_HOME, then <repo>/ckpts/LanDiff, else a Hugging Face download --, verifies it
against the md5 list of the released files and links it to ckpts/LanDiff; LANDIFF_SKIP_INIT / LANDIFF_SKIP_HASH_CHECK switch that off, and a
failure is a warning, not an import error."""
import os

__version__ = "0.1.0"

_TRUE = ("1", "true", "yes", "y", "on")
skip_init = os.environ.get("LANDIFF_SKIP_INIT", "").lower() in _TRUE
skip_hash = os.environ.get("LANDIFF_SKIP_HASH_CHECK", "").lower() in _TRUE

if not skip_init:
    try:
        from .utils import initialize_landiff_model_path
        model_path = initialize_landiff_model_path(skip_hash_verification=skip_hash)
        print(f"LanDiff model initialized at: {model_path}")
        if skip_hash:
            print("Hash verification was skipped due to LANDIFF_SKIP_HASH_CHECK environment variable.")
    except Exception as e:      # the reference does not raise here either: other parts of the package stay usable
        print(f"Warning: Failed to initialize LanDiff model path: {e}")
        print("Some LanDiff functionality may not work until the model is properly set up.")
else:
    print("LanDiff model initialization skipped due to LANDIFF_SKIP_INIT environment variable.")
