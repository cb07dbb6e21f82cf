"""adder_amd -- Python plumbing over the MI355X framed->ADDER C-ABI (libadder_hip.so).

The compute path is the HIP library; this package only moves pointers around
(ctypes, torch device memory, torch.distributed) so that tests and bench.py can
drive it.  The C++ mirror of the reference's Source/Video/Encoder surface lives in
../host/.
"""
from ._native import (  # noqa: F401
    AdderHipError, AdderHipParams, AdderFramerParams, AdderCompressedParams, EVENT_DTYPE, SPARSE_STEP_DTYPE, LIB_PATH, load,
    TIME_DELTA_T, TIME_ABSOLUTE_T, TIME_MIXED, MULTI_NORMAL, MULTI_COLLAPSE,
    CONTENT_STATIC, CONTENT_NOISE, CONTENT_SCENE, D_EMPTY, D_ZERO_INTEGRATION, D_MAX, C_NONE,
    KERNEL_LEAN, KERNEL_GENERIC, KERNEL_CONTINUOUS, KERNEL_BOUNDED, KERNEL_CONSTANT_RUNS, KERNEL_RUN_RECORDS, KERNEL_LEAN_RUNS, KERNEL_LEAN_RUNS_PACKED,
    KERNEL_NAMES,
    OK, E_BAD_PARAMS, E_HIP, E_NO_DEVICE, E_OUT_CAPACITY, E_ARENA_DEPTH, E_TIMEOUT, E_POISONED,
)
from .video import CRF, crf_feature_radius, HipVideo, raw_header, raw_events, raw_eof, synth_clip_device  # noqa: F401
from .framer import HipFramer, contiguous_run_segments, FRAMED_U8, DVS, FRAME_U8, FRAME_U16, FRAME_U32  # noqa: F401
from .compressed import CompressedEncoder, compressed_decode  # noqa: F401
from .quality import calculate_quality_metrics, calculate_mse, calculate_psnr  # noqa: F401
