"""ctypes binding of the CPU oracle (oracle/adder_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libadder_oracle.so")

FRAME_PERFECT, CONTINUOUS = 0, 1
NORMAL, COLLAPSE = 0, 1
DELTA_T, ABSOLUTE_T, MIXED = 0, 1, 2
CONTENT_STATIC, CONTENT_NOISE, CONTENT_SCENE = 0, 1, 2
SEED = 0xADDE5EED

# same 12-byte host-order record as include/adder_hip.h::AdderEvent
EVENT_DTYPE = np.dtype(
    [("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("d", "u1"), ("pad", "<u2"), ("t", "<u4")]
)
assert EVENT_DTYPE.itemsize == 12


def build(force=False):
    srcs = [os.path.join(_HERE, n) for n in ("adder_oracle.c", "adder_framer_oracle.c")]
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libadder_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, u8, u16, u32, f32, i32, sz = (
        C.c_void_p, C.c_uint8, C.c_uint16, C.c_uint32, C.c_float, C.c_int, C.c_size_t,
    )
    L.oracle_px_new.restype = vp
    L.oracle_px_new.argtypes = [f32, u16, u16, u8]
    L.oracle_px_free.argtypes = [vp]
    L.oracle_px_time_mode.argtypes = [vp, i32]
    L.oracle_px_integrate.argtypes = [vp, f32, f32, i32, u32, u32, u8, u8, i32]
    L.oracle_px_pop_best_events.restype = sz
    L.oracle_px_pop_best_events.argtypes = [vp, i32, i32, u32, f32]
    L.oracle_px_pop_top_event.argtypes = [vp, f32, i32, u32]
    L.oracle_px_set_d_for_continuous.restype = i32
    L.oracle_px_set_d_for_continuous.argtypes = [vp, f32, u32]
    L.oracle_px_step.restype = i32
    L.oracle_px_step.argtypes = [vp, u8, f32, f32, i32, i32, u32, u32, u8, u8]
    L.oracle_px_num_events.restype = sz
    L.oracle_px_num_events.argtypes = [vp]
    L.oracle_px_events.restype = vp
    L.oracle_px_events.argtypes = [vp]
    L.oracle_px_clear_events.argtypes = [vp]
    L.oracle_px_length.restype = sz
    L.oracle_px_length.argtypes = [vp]
    L.oracle_px_need_to_pop_top.restype = i32
    L.oracle_px_need_to_pop_top.argtypes = [vp]
    L.oracle_px_popped_dtm.restype = i32
    L.oracle_px_popped_dtm.argtypes = [vp]
    L.oracle_px_c_thresh.restype = u8
    L.oracle_px_c_thresh.argtypes = [vp]
    L.oracle_px_set_c_thresh.argtypes = [vp, u8, u8]
    L.oracle_px_node.argtypes = [vp, sz, C.POINTER(f32)]

    L.oracle_video_new.restype = vp
    L.oracle_video_new.argtypes = [u16, u16, u8, u32, i32, i32, u32, u32, u32, i32]
    L.oracle_video_free.argtypes = [vp]
    L.oracle_video_set_crf_parameters.argtypes = [vp, u8, u8]
    L.oracle_video_reset_c_thresh.argtypes = [vp, u8]
    L.oracle_video_set_delta_t_max.argtypes = [vp, u32]
    L.oracle_video_set_time_mode.argtypes = [vp, i32]
    L.oracle_video_set_threads.argtypes = [vp, i32]
    L.oracle_video_set_pixel_mode.argtypes = [vp, i32]
    L.oracle_video_update_detect_features.argtypes = [vp, i32, i32, u8, u16]
    L.oracle_video_set_roi.argtypes = [vp, i32, u16, u16, u16, u16, u8]
    L.oracle_video_new_features.restype = sz
    L.oracle_video_new_features.argtypes = [vp, vp, sz]
    L.oracle_video_feature_set.restype = vp
    L.oracle_video_feature_set.argtypes = [vp]
    L.oracle_video_c_thresh_plane.argtypes = [vp, vp]
    L.oracle_fast_is_feature.restype = i32
    L.oracle_fast_is_feature.argtypes = [vp, u32, u32, u32, u32, u32]
    L.oracle_video_running_intensities.restype = vp
    L.oracle_video_running_intensities.argtypes = [vp]
    L.oracle_video_integrate_matrix.restype = sz
    L.oracle_video_integrate_matrix.argtypes = [vp, vp, sz, f32, vp, sz, C.POINTER(sz), vp]

    L.oracle_video_integrate_matrix_chunks.restype = sz
    L.oracle_video_integrate_matrix_chunks.argtypes = [vp, vp, sz, f32]
    L.oracle_video_integrate_clip.restype = sz
    L.oracle_video_integrate_clip.argtypes = [vp, vp, sz, sz, sz, f32, vp]
    L.oracle_video_chunks_copy_out.restype = sz
    L.oracle_video_chunks_copy_out.argtypes = [vp, vp]
    L.oracle_video_chunks_raw_events.restype = sz
    L.oracle_video_chunks_raw_events.argtypes = [vp, vp]

    L.oracle_raw_header.restype = sz
    L.oracle_raw_header.argtypes = [vp, u8, u16, u16, u8, u32, u32, u32, u32, u32, u32]
    L.oracle_raw_events.restype = sz
    L.oracle_raw_events.argtypes = [vp, vp, sz, u8]
    L.oracle_raw_eof.restype = sz
    L.oracle_raw_eof.argtypes = [vp]

    L.oracle_framer_new.restype = vp
    L.oracle_framer_new.argtypes = [u16, u16, u8, u32, u32, u32, u32, f32, u8, i32, u32]
    L.oracle_framer_free.argtypes = [vp]
    L.oracle_framer_buffer_limit.argtypes = [vp, i32, u32]
    L.oracle_framer_tpf.restype = u32
    L.oracle_framer_tpf.argtypes = [vp]
    L.oracle_framer_frames_written.restype = C.c_int64
    L.oracle_framer_frames_written.argtypes = [vp]
    L.oracle_framer_num_chunks.restype = sz
    L.oracle_framer_num_chunks.argtypes = [vp]
    L.oracle_framer_is_frame_0_filled.restype = i32
    L.oracle_framer_is_frame_0_filled.argtypes = [vp]
    L.oracle_framer_ingest_event.restype = i32
    L.oracle_framer_ingest_event.argtypes = [vp, vp]
    L.oracle_framer_ingest_events_events.restype = i32
    L.oracle_framer_ingest_events_events.argtypes = [vp, vp, vp]
    L.oracle_framer_flush_frame_buffer.restype = i32
    L.oracle_framer_flush_frame_buffer.argtypes = [vp]
    L.oracle_framer_is_frame_filled.restype = i32
    L.oracle_framer_is_frame_filled.argtypes = [vp, sz]
    L.oracle_framer_write_frame_bytes.restype = sz
    L.oracle_framer_write_frame_bytes.argtypes = [vp, vp]
    L.oracle_framer_write_multi_frame_bytes.restype = i32
    L.oracle_framer_write_multi_frame_bytes.argtypes = [vp, vp, sz, C.POINTER(sz)]

    L.oracle_synth_clip.argtypes = [vp, i32, C.c_uint64, u32, u32, u32, u32, u32, u32, u32]
    L.oracle_max_threads.restype = i32
    _lib = L
    return L


class Pixel:
    """One PixelArena (event_pixel_tree.rs:53-87) for the unit-test known answers."""

    def __init__(self, start_intensity, x=0, y=0, c=0xFF):
        self.L = lib()
        self.h = self.L.oracle_px_new(start_intensity, x, y, c)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_px_free(self.h)
            self.h = None

    def time_mode(self, tm):
        self.L.oracle_px_time_mode(self.h, tm)

    def integrate(self, intensity, time, mode, dtm, ref_time, c_max, velocity, multi_mode):
        self.L.oracle_px_integrate(self.h, intensity, time, mode, dtm, ref_time, c_max, velocity, multi_mode)

    def pop_best_events(self, mode, multi_mode, ref_time, intensity):
        n0 = self.L.oracle_px_num_events(self.h)
        n = self.L.oracle_px_pop_best_events(self.h, mode, multi_mode, ref_time, intensity)
        return self.events()[n0 : n0 + n]

    def pop_top_event(self, next_intensity, mode, ref_time):
        self.L.oracle_px_pop_top_event(self.h, next_intensity, mode, ref_time)
        return self.events()[-1]

    def set_d_for_continuous(self, next_intensity, ref_time):
        if self.L.oracle_px_set_d_for_continuous(self.h, next_intensity, ref_time):
            return self.events()[-1]
        return None

    def step(self, frame_val, time_spanned, mode, multi_mode, dtm, ref_time, c_max, velocity):
        return self.L.oracle_px_step(
            self.h, frame_val, float(frame_val), time_spanned, mode, multi_mode, dtm, ref_time, c_max, velocity
        )

    def set_c_thresh(self, c, counter):
        self.L.oracle_px_set_c_thresh(self.h, c, counter)

    def events(self):
        n = self.L.oracle_px_num_events(self.h)
        if n == 0:
            return np.zeros(0, EVENT_DTYPE)
        p = self.L.oracle_px_events(self.h)
        buf = (C.c_uint8 * (n * 12)).from_address(p)
        return np.frombuffer(buf, dtype=EVENT_DTYPE).copy()

    @property
    def length(self):
        return self.L.oracle_px_length(self.h)

    @property
    def need_to_pop_top(self):
        return bool(self.L.oracle_px_need_to_pop_top(self.h))

    def node(self, idx):
        out = (C.c_float * 7)()
        self.L.oracle_px_node(self.h, idx, out)
        return dict(
            d=int(out[0]), integration=np.float32(out[1]), delta_t=np.float32(out[2]),
            has_best=bool(out[3]), best_d=int(out[4]), best_delta_t=np.float32(out[5]), alt=bool(out[6]),
        )


SPARSE_STEP_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("frame_val", "u1"), ("pad", "<u2"),
                              ("intensity", "<f4"), ("time", "<f4")])


class Video:
    """Video<W> driver state + integrate_matrix (video.rs:350-438, 651-778)."""

    def __init__(self, width, height, channels=1, *, row_begin=0, time_mode=ABSOLUTE_T,
                 multi_mode=COLLAPSE, ref_time=255, delta_t_max=7650, chunk_rows=1, threads=1):
        self.L = lib()
        self.width, self.height, self.channels = width, height, channels
        self.chunk_rows = chunk_rows
        self.num_chunks = (height + chunk_rows - 1) // chunk_rows
        self.h = self.L.oracle_video_new(
            width, height, channels, row_begin, time_mode, multi_mode, ref_time, delta_t_max, chunk_rows, threads
        )
        if not self.h:
            raise ValueError("bad oracle video parameters")
        self._cap = max(1024, width * height * channels * 2)
        self._out = np.zeros(self._cap, EVENT_DTYPE)
        self._chunks = np.zeros(self.num_chunks + 1, np.uint32)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_video_free(self.h)
            self.h = None

    def set_crf_parameters(self, c_thresh_max, c_increase_velocity):
        self.L.oracle_video_set_crf_parameters(self.h, c_thresh_max, c_increase_velocity)

    def reset_c_thresh(self, baseline):
        self.L.oracle_video_reset_c_thresh(self.h, baseline)

    def set_delta_t_max(self, dtm):
        self.L.oracle_video_set_delta_t_max(self.h, dtm)

    def set_time_mode(self, tm):
        self.L.oracle_video_set_time_mode(self.h, tm)

    def set_threads(self, n):
        self.L.oracle_video_set_threads(self.h, n)

    def update_detect_features(self, detect, rate_adjustment, c_thresh_baseline, feature_c_radius):
        """Video::update_detect_features + the CrfParameters the feedback uses (video.rs:825-840, 1085-1105)."""
        self.L.oracle_video_update_detect_features(self.h, int(detect), int(rate_adjustment), c_thresh_baseline,
                                                   feature_c_radius)

    def set_roi(self, roi, c_thresh_baseline):
        if roi is None:
            self.L.oracle_video_set_roi(self.h, 0, 0, 0, 0, 0, c_thresh_baseline)
        else:
            self.L.oracle_video_set_roi(self.h, 1, roi[0], roi[1], roi[2], roi[3], c_thresh_baseline)

    def new_features(self):
        buf = np.zeros(self.width * self.height, np.uint32)
        n = self.L.oracle_video_new_features(self.h, buf.ctypes.data, len(buf))
        return buf[:n].copy()

    def feature_set(self):
        p = self.L.oracle_video_feature_set(self.h)
        n = self.width * self.height
        return np.frombuffer((C.c_uint8 * n).from_address(p), dtype=np.uint8).reshape(self.height, self.width).copy()

    def c_thresh_plane(self):
        out = np.zeros(self.width * self.height * self.channels, np.uint8)
        self.L.oracle_video_c_thresh_plane(self.h, out.ctypes.data)
        return out.reshape(self.height, self.width, self.channels)

    def integrate_sparse(self, steps):
        """integrate_for_px(px, &mut 0, frame_val, intensity, time) per step, in order (prophesee.rs:170-258):
        steps = array of SPARSE_STEP_DTYPE; returns the events of all steps in one buffer."""
        steps = np.ascontiguousarray(steps, SPARSE_STEP_DTYPE)
        self.L.oracle_video_integrate_sparse.restype = C.c_int
        self.L.oracle_video_integrate_sparse.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                         C.POINTER(C.c_size_t)]
        cap = max(1024, len(steps) * 24)
        out = np.zeros(cap, EVENT_DTYPE)
        n = C.c_size_t(0)
        rc = self.L.oracle_video_integrate_sparse(self.h, steps.ctypes.data, len(steps), out.ctypes.data, cap, C.byref(n))
        if rc != 0:
            raise ValueError(f"oracle_video_integrate_sparse failed: {rc}")
        return out[: n.value].copy()

    def fill_running_intensities(self, value):
        self.L.oracle_video_fill_running_intensities.argtypes = [C.c_void_p, C.c_uint8]
        self.L.oracle_video_fill_running_intensities(self.h, value)

    def set_pixel_mode(self, mode):
        """0 = Mode::FramePerfect (default), 1 = Mode::Continuous (lib.rs:196-205)."""
        self.L.oracle_video_set_pixel_mode(self.h, mode)

    def running_intensities(self):
        p = self.L.oracle_video_running_intensities(self.h)
        n = self.width * self.height * self.channels
        buf = (C.c_uint8 * n).from_address(p)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, self.channels).copy()

    def integrate_matrix(self, frame, time_spanned=None, ref_time=255, want_chunks=False):
        frame = np.ascontiguousarray(frame, dtype=np.uint8).reshape(self.height, self.width * self.channels)
        if time_spanned is None:
            time_spanned = float(ref_time)
        n = C.c_size_t(0)
        while True:
            r = self.L.oracle_video_integrate_matrix(
                self.h, frame.ctypes.data, frame.strides[0], time_spanned,
                self._out.ctypes.data, self._cap, C.byref(n), self._chunks.ctypes.data,
            )
            if r != C.c_size_t(-1).value:
                break
            raise RuntimeError("oracle event buffer too small (state already advanced)")
        ev = self._out[: n.value].view(np.uint32).copy().view(EVENT_DTYPE)
        if want_chunks:
            return ev, self._chunks.copy()
        return ev

    # ---- CPU-baseline timing path (bench.py): events stay in the per-chunk buffers, like the
    # reference's Vec<Vec<Event>> ----
    def integrate_matrix_chunks(self, frame_ptr, row_stride, time_spanned):
        return self.L.oracle_video_integrate_matrix_chunks(self.h, frame_ptr, row_stride, time_spanned)

    def integrate_clip(self, frames_ptr, num_frames, frame_stride, row_stride, time_spanned, sink_ptr=None):
        """CPU-baseline timing: the clip through one persistent OpenMP team (oracle_video_integrate_clip)."""
        return self.L.oracle_video_integrate_clip(self.h, frames_ptr, num_frames, frame_stride, row_stride, time_spanned,
                                                  sink_ptr)

    def chunks_copy_out(self, n):
        if n > self._cap:
            self._cap = n
            self._out = np.zeros(n, EVENT_DTYPE)
        got = self.L.oracle_video_chunks_copy_out(self.h, self._out.ctypes.data)
        return self._out[:got]

    def chunks_raw_events(self, sink_ptr):
        return self.L.oracle_video_chunks_raw_events(self.h, sink_ptr)

    def ensure_capacity(self, events_per_unit):
        cap = int(self.width * self.height * self.channels * events_per_unit) + 1024
        if cap > self._cap:
            self._cap = cap
            self._out = np.zeros(cap, EVENT_DTYPE)


FRAMED_U8, DVS = 0, 6  # SourceCamera (adder-codec-core/src/lib.rs:35-47)


class Framer:
    """FrameSequence<u8>, INSTANTANEOUS / Intensity (framer/driver.rs:261-981)."""

    def __init__(self, width, height, channels=1, *, chunk_rows=64, tps, ref_interval, delta_t_max,
                 output_fps=None, codec_version=1, time_mode=DELTA_T, source_camera=FRAMED_U8):
        self.L = lib()
        self.width, self.height, self.channels = width, height, channels
        self.frame_bytes = width * height * channels
        self.h = self.L.oracle_framer_new(width, height, channels, chunk_rows, tps, ref_interval, delta_t_max,
                                          -1.0 if output_fps is None else float(output_fps), codec_version,
                                          time_mode, source_camera)
        if not self.h:
            raise ValueError("bad oracle framer parameters")
        self.chunk_rows = chunk_rows
        self.num_chunks = self.L.oracle_framer_num_chunks(self.h)

    def set_value_type(self, value_type):
        """The frame element type T of FrameSequence<T>: 0 u8, 1 u16, 2 u32 (big-endian in the written bytes)."""
        self.L.oracle_framer_set_value_type.argtypes = [C.c_void_p, C.c_int]
        if self.L.oracle_framer_set_value_type(self.h, value_type) != 0:
            raise ValueError("value_type must be 0, 1 or 2")
        self.frame_bytes = (self.width * self.height * self.channels) << value_type

    def set_view(self, view_mode, source_type=0, practical_d_max=0.0):
        """FramerBuilder::view_mode / ::source: 0 Intensity, 1 D, 2 DeltaT, 3 SAE; source 0 U8 .. 3 U64."""
        self.L.oracle_framer_set_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
        self.L.oracle_framer_set_view(self.h, view_mode, source_type, practical_d_max)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_framer_free(self.h)
            self.h = None

    @property
    def tpf(self):
        return self.L.oracle_framer_tpf(self.h)

    @property
    def frames_written(self):
        return self.L.oracle_framer_frames_written(self.h)

    def ingest_event(self, x, y, c, d, t):
        ev = np.zeros(1, EVENT_DTYPE)
        ev[0] = (x, y, 0xFF if c is None else c, d, 0, t)
        return bool(self.L.oracle_framer_ingest_event(self.h, ev.ctypes.data))

    def ingest_events(self, events):
        """Event by event (Framer::ingest_event); returns the frames that became ready, in order,
        popped exactly where the reference's read loop pops them (tests/integration_tests.rs:86-101)."""
        events = np.ascontiguousarray(events, dtype=EVENT_DTYPE).copy()
        out = []
        p = events.ctypes.data
        for i in range(len(events)):
            if self.L.oracle_framer_ingest_event(self.h, p + 12 * i):
                out.append(self.write_multi_frame_bytes())
        return b"".join(out)

    def ingest_events_events(self, events, chunk_offsets):
        events = np.ascontiguousarray(events, dtype=EVENT_DTYPE).copy()
        offs = np.ascontiguousarray(chunk_offsets, dtype=np.uint64)
        assert len(offs) == self.num_chunks + 1
        return bool(self.L.oracle_framer_ingest_events_events(self.h, events.ctypes.data, offs.ctypes.data))

    def flush_frame_buffer(self):
        return bool(self.L.oracle_framer_flush_frame_buffer(self.h))

    def is_frame_filled(self, idx):
        r = self.L.oracle_framer_is_frame_filled(self.h, idx)
        if r < 0:
            raise IndexError("InvalidIndex" if r == -1 else "BadFillCount")
        return bool(r)

    def write_frame_bytes(self):
        buf = np.zeros(self.frame_bytes, np.uint8)
        n = self.L.oracle_framer_write_frame_bytes(self.h, buf.ctypes.data)
        return buf[:n].tobytes()

    def write_multi_frame_bytes(self, max_frames=4096):
        buf = np.zeros(self.frame_bytes * max_frames, np.uint8)
        n = C.c_size_t(0)
        frames = self.L.oracle_framer_write_multi_frame_bytes(self.h, buf.ctypes.data, buf.nbytes, C.byref(n))
        if frames < 0:
            raise RuntimeError("write_multi_frame_bytes failed")
        return buf[: n.value].tobytes()


def raw_header(codec_version, width, height, channels, tps, ref_interval, delta_t_max,
               source_camera=0, time_mode=ABSOLUTE_T, adu_interval=0):
    buf = np.zeros(64, np.uint8)
    n = lib().oracle_raw_header(buf.ctypes.data, codec_version, width, height, channels, tps,
                                ref_interval, delta_t_max, source_camera, time_mode, adu_interval)
    return buf[:n].tobytes()


def raw_events(events, channels):
    events = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
    buf = np.zeros(len(events) * 11 + 16, np.uint8)
    n = lib().oracle_raw_events(buf.ctypes.data, events.ctypes.data, len(events), channels)
    return buf[:n].tobytes()


def raw_eof():
    buf = np.zeros(16, np.uint8)
    n = lib().oracle_raw_eof(buf.ctypes.data)
    return buf[:n].tobytes()


def synth_clip(content, W, H, C_, frames, *, y0=0, rows=None, k0=0, seed=SEED):
    rows = H - y0 if rows is None else rows
    out = np.zeros((frames, rows, W, C_), np.uint8)
    lib().oracle_synth_clip(out.ctypes.data, content, seed, W, H, C_, y0, rows, k0, frames)
    return out


def max_threads():
    return lib().oracle_max_threads()


def sizeof_pixel_arena():
    L = lib()
    L.oracle_sizeof_pixel_arena.restype = C.c_size_t
    return int(L.oracle_sizeof_pixel_arena())


def stream_triad_GBs(n_floats, threads, reps=3):
    L = lib()
    L.oracle_stream_triad.restype = C.c_double
    L.oracle_stream_triad.argtypes = [C.c_size_t, C.c_int, C.c_int]
    return float(L.oracle_stream_triad(int(n_floats), int(threads), int(reps)))
