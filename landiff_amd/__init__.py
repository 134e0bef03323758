"""landiff_amd -- MI355X (gfx950) native inference path for LanDiff text-to-video.

Python host (PyTorch-ROCm for memory, streams, RNG, RCCL) over hand-written HIP kernels in
``liblandiff_hip.so`` (C ABI declared in ``include/landiff_hip.h``).  There is no CPU or
PyTorch fallback for the compute path: if the library or a GPU is missing the ops raise.
"""
__version__ = "0.1.0"
