"""pisces_amd: the MI355X (gfx950) pileup-and-likelihood engine behind Pisces' caller interfaces (see README.md)."""
import os as _os

# Two settings of the ROCm runtime that the library's launches are measured with; they only count when they are in the environment
# before the HIP runtime starts in this process (INTEGRATION.md asks a C# host for the same):
#   GPU_MAX_HW_QUEUES=16     every HIP stream of pisces_hip_call_tiles_batched gets its own hardware queue
#   HIP_FORCE_DEV_KERNARG=1  kernel arguments live in device memory: a wave's first loads are not PCIe round trips (-1.6 us per launch
#                            of the hot kernel)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
