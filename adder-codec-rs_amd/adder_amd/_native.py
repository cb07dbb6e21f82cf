"""ctypes loader of libadder_hip.so -- the C-ABI declared in include/adder_hip.h.

The library is built in-tree by `make -C adder-codec-rs_amd` (or __graft_entry__.build()).
There is no fallback of any kind: if the shared object is missing, or no gfx950 device
is visible when a context is created, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# ADDER_HIP_LIB: A/B builds of the same library (tools/build_variants.sh); the default is the in-tree build
LIB_PATH = os.environ.get("ADDER_HIP_LIB") or os.path.join(os.path.dirname(_PKG), "libadder_hip.so")

ABI_VERSION = 1
TIME_DELTA_T, TIME_ABSOLUTE_T, TIME_MIXED = 0, 1, 2
MULTI_NORMAL, MULTI_COLLAPSE = 0, 1
CONTENT_STATIC, CONTENT_NOISE, CONTENT_SCENE = 0, 1, 2
D_MAX, D_ZERO_INTEGRATION, D_EMPTY, C_NONE = 127, 128, 255, 0xFF

# adder_hip_last_batch_kernel (include/adder_hip.h)
(KERNEL_LEAN, KERNEL_GENERIC, KERNEL_CONTINUOUS, KERNEL_BOUNDED, KERNEL_CONSTANT_RUNS, KERNEL_RUN_RECORDS,
 KERNEL_LEAN_RUNS, KERNEL_LEAN_RUNS_PACKED) = range(8)
KERNEL_NAMES = ("adder_lean_kernel", "adder_frame_kernel", "adder_cont_kernel", "adder_cb_kernel", "adder_cr_kernel", "adder_rr_kernel",
                "adder_lr_kernel", "adder_lp_kernel")

OK = 0
E_BAD_PARAMS, E_HIP, E_NO_DEVICE, E_OUT_CAPACITY, E_ARENA_DEPTH, E_TIMEOUT, E_POISONED = -1, -2, -3, -4, -5, -6, -7

EVENT_DTYPE = np.dtype(
    [("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("d", "u1"), ("pad", "<u2"), ("t", "<u4")]
)
assert EVENT_DTYPE.itemsize == 12


class AdderHipParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("width", C.c_uint16),
        ("height", C.c_uint16),
        ("channels", C.c_uint8),
        ("time_mode", C.c_uint8),
        ("multi_mode", C.c_uint8),
        ("pixel_mode", C.c_uint8),
        ("row_begin", C.c_uint32),
        ("row_end", C.c_uint32),
        ("ref_time", C.c_uint32),
        ("delta_t_max", C.c_uint32),
        ("c_thresh_max", C.c_uint8),
        ("c_increase_velocity", C.c_uint8),
        ("c_thresh_start", C.c_uint8),
        ("c_counter_start", C.c_uint8),
        ("chunk_rows", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("device_id", C.c_int32),
    ]


SPARSE_STEP_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("frame_val", "u1"), ("pad", "<u2"),
                              ("intensity", "<f4"), ("time", "<f4")])  # AdderSparseStep


class AdderFramerParams(C.Structure):
    """include/adder_framer.h::AdderFramerParams"""
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("width", C.c_uint16),
        ("height", C.c_uint16),
        ("channels", C.c_uint8),
        ("codec_version", C.c_uint8),
        ("time_mode", C.c_uint8),
        ("reserved0", C.c_uint8),
        ("row_begin", C.c_uint32),
        ("row_end", C.c_uint32),
        ("tps", C.c_uint32),
        ("ref_interval", C.c_uint32),
        ("delta_t_max", C.c_uint32),
        ("output_fps", C.c_float),
        ("source_camera", C.c_uint32),
        ("ring_frames", C.c_uint32),
        ("device_id", C.c_int32),
        ("view_mode", C.c_uint8),
        ("source_type", C.c_uint8),
        ("value_type", C.c_uint8),
        ("reserved1", C.c_uint8),
        ("practical_d_max", C.c_float),
    ]


class AdderBandRecords(C.Structure):
    """include/adder_hip.h::AdderBandRecords (records over the wire)"""
    _fields_ = [
        ("num_frames", C.c_uint32),
        ("num_segments", C.c_uint32),
        ("record_bytes", C.c_uint32),
        ("row_begin", C.c_uint32),
        ("rows", C.c_uint32),
        ("d_counts", C.c_void_p),
        ("d_prefix", C.c_void_p),
        ("d_runs", C.c_void_p),
        ("d_records", C.c_void_p),
        ("d_frame_offsets", C.c_void_p),
        ("d_frame_table", C.c_void_p),
    ]


class AdderCompressedParams(C.Structure):
    """include/adder_compressed.h::AdderCompressedParams"""
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("width", C.c_uint16),
        ("height", C.c_uint16),
        ("channels", C.c_uint8),
        ("codec_version", C.c_uint8),
        ("time_mode", C.c_uint8),
        ("write_header", C.c_uint8),
        ("tps", C.c_uint32),
        ("ref_interval", C.c_uint32),
        ("delta_t_max", C.c_uint32),
        ("adu_interval", C.c_uint32),
        ("source_camera", C.c_uint32),
        ("c_thresh_max", C.c_uint8),
        ("reserved", C.c_uint8 * 3),
        ("threads", C.c_uint32),
    ]


# every symbol include/*.h declares: name -> (restype, argtypes)
_vp, _u8, _u16, _u32, _u64, _f32, _i32, _sz = (
    C.c_void_p, C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_float, C.c_int, C.c_size_t)
SYMBOLS = {
    "adder_hip_default_params": (None, [C.POINTER(AdderHipParams), _u16, _u16, _u8]),
    "adder_hip_create": (_i32, [C.POINTER(AdderHipParams), C.POINTER(_vp)]),
    "adder_hip_destroy": (None, [_vp]),
    "adder_hip_last_error": (C.c_char_p, [_vp]),
    "adder_hip_set_crf_parameters": (_i32, [_vp, _u8, _u8]),
    "adder_hip_reset_c_thresh": (_i32, [_vp, _u8]),
    "adder_hip_update_detect_features": (_i32, [_vp, _i32, _i32]),
    "adder_hip_set_feature_parameters": (_i32, [_vp, _u8, _u16]),
    "adder_hip_update_roi": (_i32, [_vp, _i32, _u16, _u16, _u16, _u16]),
    "adder_hip_feature_set": (_i32, [_vp, _vp]),
    "adder_hip_c_thresh_plane": (_i32, [_vp, _vp]),
    "adder_hip_last_new_features": (_u32, [_vp]),
    "adder_hip_frames_configure": (_i32, [_vp, _u32, _sz]),
    "adder_hip_frames_set_format": (_i32, [_vp, _i32]),
    "adder_hip_frame_collect_wire": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_vp)]),
    "adder_hip_frame_submit": (_i32, [_vp, _vp, _sz, _f32]),
    "adder_hip_frame_collect": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp)]),
    "adder_hip_frames_in_flight": (_u32, [_vp]),
    "adder_hip_set_delta_t_max": (_i32, [_vp, _u32]),
    "adder_hip_set_time_mode": (_i32, [_vp, _u8]),
    "adder_hip_alloc_pinned": (_vp, [_sz]),
    "adder_hip_free_pinned": (None, [_vp]),
    "adder_hip_num_chunks": (_u32, [_vp]),
    "adder_hip_max_events_per_frame": (_sz, [_vp]),
    "adder_hip_integrate": (_i32, [_vp, _vp, _sz, _f32, _vp, _sz, C.POINTER(_sz), _vp]),
    "adder_hip_integrate_batch": (_i32, [_vp, _vp, _u32, _sz, _sz, _f32, _vp, _sz, C.POINTER(_sz), _vp]),
    "adder_hip_integrate_device": (_i32, [_vp, _vp, _u32, _f32, _vp, _sz, _vp, _vp]),
    "adder_hip_integrate_wire_device": (_i32, [_vp, _vp, _u32, _f32, _vp, _sz, _vp, _vp]),
    "adder_hip_finish": (_i32, [_vp, C.POINTER(_sz)]),
    "adder_hip_chunk_offsets_device": (_i32, [_vp, _vp, _sz, _vp, _vp]),
    "adder_hip_running_intensities": (_i32, [_vp, _vp]),
    "adder_hip_enable_running_intensities": (_i32, [_vp, _i32]),
    "adder_hip_last_batch_ms": (_f32, [_vp]),
    "adder_hip_launch_plan_settled": (_i32, [_vp]),
    "adder_hip_last_batch_kernel": (_u32, [_vp]),
    "adder_hip_integrate_sparse": (_i32, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "adder_hip_integrate_sparse_device": (_i32, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz), _vp]),
    "adder_hip_debug_timeline": (_i32, [_vp, _vp]),
    "adder_hip_set_launch_timing": (_i32, [_vp, _i32]),
    "adder_hip_last_launch_avg_us": (_f32, [_vp]),
    "adder_hip_last_launch_frames": (_f32, [_vp]),
    "adder_hip_last_post_avg_us": (_f32, [_vp]),
    "adder_hip_last_post_chunks": (_u32, [_vp]),
    "adder_hip_chunk_frames": (_u32, [_vp]),
    "adder_hip_band_segments": (_u32, [_vp]),
    "adder_hip_integrate_records_device": (C.c_int, [_vp, _vp, _u32, C.c_float, _vp, _vp, _vp]),
    "adder_hip_expand_records_device": (C.c_int, [_vp, _vp, _u32, _vp, C.c_size_t, _u64, _vp, _vp]),
    "adder_hip_expand_records_wire_device": (C.c_int, [_vp, _vp, _u32, _vp, C.c_size_t, _u64, _vp, _vp]),
    "adder_hip_segment_units": (_u32, []),
    "adder_hip_wire_record_bytes": (_u32, [_vp]),
    "adder_hip_expand_status": (C.c_int, [_vp, _vp]),
    "adder_hip_records_wire_bytes": (_sz, [_u32, _u32, _u32, _u64]),
    "adder_hip_records_wire_sections": (None, [_u32, _u32, _u32, _vp]),
    "adder_hip_records_to_wire": (C.c_int, [_vp, _vp, _u64, _vp, _sz, _vp]),
    "adder_hip_last_batch_stream": (_vp, [_vp]),
    "adder_hip_sync_last_batch_stream": (C.c_int, [_vp]),
    "adder_hip_last_batch_records": (_u64, [_vp]),
    "adder_hip_set_frames_per_launch": (_i32, [_vp, _u32]),
    "adder_hip_reset": (_i32, [_vp]),
    "adder_hip_wire_events_device": (_i32, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz), _vp]),
    "adder_hip_sink_layout_device": (_i32, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "adder_hip_wire_scatter_device": (_i32, [_vp, _vp, _vp, _u32, _vp, _vp, C.c_uint64, C.c_uint64, _vp]),
    "adder_hip_integrate_batch_raw": (_i32, [_vp, _vp, _u32, _sz, _sz, _f32, _vp, _sz, C.POINTER(_sz), C.POINTER(_sz), _vp]),
    "adder_hip_stream_submit": (_i32, [_vp, _vp, _u32, _sz, _sz, _f32, _sz]),
    "adder_hip_stream_collect": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_vp)]),
    "adder_hip_selftest_division": (_i32, [_vp]),
    "adder_hip_synth_clip_device": (_i32, [_vp, _i32, _u64, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp]),
    "adder_hip_merge_work_bytes": (_sz, [_u32, _u32]),
    "adder_hip_merge_streams_device": (_i32, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _sz, _vp, _vp]),
    "adder_hip_merge_streams_device_at": (_i32, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _sz, _vp, _u64, _vp]),
    "adder_hip_check_status": (_i32, [_vp, _vp]),
    "adder_hip_feature_halo_bytes": (_sz, [_vp]),
    "adder_hip_feature_halo_export": (_i32, [_vp, _vp, _vp, _vp]),
    "adder_hip_feature_halo_import": (_i32, [_vp, _vp, _vp, _vp]),
    "adder_hip_feature_detect": (_i32, [_vp, _vp, _u32, C.POINTER(_u32), _vp]),
    "adder_hip_feature_apply": (_i32, [_vp, _vp, _u32, _vp]),
    "adder_raw_header": (_sz, [_vp, _u8, _u16, _u16, _u8, _u32, _u32, _u32, _u32, _u32, _u32]),
    "adder_raw_events": (_sz, [_vp, _vp, _sz, _u8]),
    "adder_raw_eof": (_sz, [_vp]),
    # include/adder_compressed.h
    "adder_compressed_default_params": (None, [C.POINTER(AdderCompressedParams), _u16, _u16, _u8]),
    "adder_compressed_encoder_create": (_i32, [C.POINTER(AdderCompressedParams), C.POINTER(_vp)]),
    "adder_compressed_encoder_destroy": (None, [_vp]),
    "adder_compressed_last_error": (C.c_char_p, [_vp]),
    "adder_compressed_encoder_ingest": (_i32, [_vp, _vp, _sz]),
    "adder_compressed_encoder_close": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    "adder_compressed_encoder_progress": (_i32, [_vp, C.POINTER(_u32), C.POINTER(_sz)]),
    "adder_compressed_decode": (_i32, [_vp, _sz, _i32, C.POINTER(AdderCompressedParams), _vp, _sz, C.POINTER(_sz)]),
    # include/adder_framer.h
    "adder_framer_default_params": (None, [C.POINTER(AdderFramerParams), _u16, _u16, _u8]),
    "adder_framer_create": (_i32, [C.POINTER(AdderFramerParams), C.POINTER(_vp)]),
    "adder_framer_destroy": (None, [_vp]),
    "adder_framer_last_error": (C.c_char_p, [_vp]),
    "adder_framer_tpf": (_u32, [_vp]),
    "adder_framer_frames_written": (C.c_int64, [_vp]),
    "adder_framer_ingest_device": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "adder_framer_ingest_frames_device": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "adder_framer_ingest_frames_device_offsets": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "adder_framer_ingest": (_i32, [_vp, _vp, _vp, _u32]),
    "adder_framer_frames_ready": (_i32, [_vp, C.POINTER(_u32)]),
    "adder_framer_pop_device": (_i32, [_vp, _vp, _u32, C.POINTER(_u32), _vp]),
    "adder_framer_pop": (_i32, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "adder_framer_write_frame": (_i32, [_vp, _vp]),
    "adder_framer_flush": (_i32, [_vp, C.POINTER(_i32)]),
}

_lib = None


class AdderHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"adder_hip error {code}: {msg}")
        self.code = code


def load():
    """Loads libadder_hip.so and binds every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64 (same SONAME
    # as /opt/rocm's).  If torch is going to be used in this process it must be loaded
    # FIRST so that the dynamic loader binds libadder_hip.so to that same copy -- two
    # copies cannot both open the device (the second one reports "no devices").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: build it with `make -C adder-codec-rs_amd` "
            "(hipcc, gfx950).  This path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check_framer(fr, rc):
    if rc != OK:
        msg = load().adder_framer_last_error(fr)
        raise AdderHipError(rc, msg.decode() if msg else "")


def check(ctx, rc):
    if rc != OK:
        msg = load().adder_hip_last_error(ctx)
        raise AdderHipError(rc, msg.decode() if msg else "")
