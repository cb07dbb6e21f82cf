"""HipFramer -- ctypes plumbing over include/adder_framer.h (events -> u8 / u16 / u32 frames on the GPU).

Mirrors FramerBuilder / FrameSequence<T> (adder-codec-rs/src/framer/driver.rs:55-138, 261-981) for
FramerMode::INSTANTANEOUS; T picked by value_type (scale_intensity.rs:16-209).
"""
import ctypes as C

import numpy as np

from . import _native as N

FRAMED_U8, DVS = 0, 6  # SourceCamera (adder-codec-core/src/lib.rs:35-47)
FRAME_U8, FRAME_U16, FRAME_U32 = 0, 1, 2  # the frame element type T of FrameSequence<T>


class HipFramer:
    def __init__(self, width, height, channels=1, *, tps, ref_interval, delta_t_max, output_fps=None,
                 codec_version=1, time_mode=N.TIME_DELTA_T, source_camera=FRAMED_U8, row_begin=0, row_end=None,
                 ring_frames=0, device_id=0, view_mode=0, source_type=0, practical_d_max=0.0,
                 value_type=FRAME_U8):
        self.L = N.load()
        p = N.AdderFramerParams()
        self.L.adder_framer_default_params(C.byref(p), width, height, channels)
        p.codec_version, p.time_mode = codec_version, time_mode
        p.row_begin, p.row_end = row_begin, height if row_end is None else row_end
        p.tps, p.ref_interval, p.delta_t_max = tps, ref_interval, delta_t_max
        p.output_fps = 0.0 if output_fps is None else float(output_fps)
        p.source_camera, p.ring_frames, p.device_id = source_camera, ring_frames, device_id
        # FramedViewMode 0 Intensity / 1 D / 2 DeltaT / 3 SAE; SourceType 0 U8 .. 3 U64; practical_d_max for the D view
        p.view_mode, p.source_type, p.practical_d_max = view_mode, source_type, float(practical_d_max)
        p.value_type = value_type  # popped frames hold big-endian elements of 1 << value_type bytes (bincode, driver.rs:279)
        h = C.c_void_p()
        rc = self.L.adder_framer_create(C.byref(p), C.byref(h))
        if rc != N.OK:
            msg = self.L.adder_framer_last_error(None)
            raise N.AdderHipError(rc, msg.decode() if msg else "")
        self.h = h
        self.width, self.height, self.channels = width, height, channels
        self.rows = p.row_end - p.row_begin
        self.frame_bytes = (self.rows * width * channels) << value_type

    def close(self):
        if getattr(self, "h", None):
            self.L.adder_framer_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def tpf(self):
        return self.L.adder_framer_tpf(self.h)

    @property
    def frames_written(self):
        return self.L.adder_framer_frames_written(self.h)

    def ingest(self, events, seg_offsets=None):
        """events: host array (EVENT_DTYPE); seg_offsets: boundaries of segments inside which every
        pixel-channel's events are contiguous (default: computed here -- a new segment wherever a
        pixel-channel shows up again, so an arbitrary stream is safe to pass)."""
        events = np.ascontiguousarray(events, dtype=N.EVENT_DTYPE)
        offs = contiguous_run_segments(events) if seg_offsets is None else \
            np.ascontiguousarray(seg_offsets, dtype=np.uint64)
        N.check_framer(self.h, self.L.adder_framer_ingest(self.h, events.ctypes.data, offs.ctypes.data, len(offs) - 1))

    def ingest_device(self, d_events, seg_offsets, stream=None):
        """d_events: CUDA tensor holding 12-byte events; seg_offsets: host array of event indices."""
        offs = np.ascontiguousarray(seg_offsets, dtype=np.uint64)
        N.check_framer(self.h, self.L.adder_framer_ingest_device(
            self.h, d_events.data_ptr(), offs.ctypes.data, len(offs) - 1, C.c_void_p(stream) if stream else None))

    def ingest_frames_device(self, d_events, frame_offsets, stream=None):
        """The transcoder's output as it is: per-frame raster-ordered segments, one launch per batch."""
        offs = np.ascontiguousarray(frame_offsets, dtype=np.uint64)
        N.check_framer(self.h, self.L.adder_framer_ingest_frames_device(
            self.h, d_events.data_ptr(), offs.ctypes.data, len(offs) - 1, C.c_void_p(stream) if stream else None))

    def ingest_frames_device_offsets(self, d_events, d_frame_offsets, num_frames, stream=None):
        """The same with the frame offsets where integrate_device left them (int64 / uint64 CUDA tensor of
        num_frames + 1 entries): nothing crosses the bus, the call only queues."""
        N.check_framer(self.h, self.L.adder_framer_ingest_frames_device_offsets(
            self.h, d_events.data_ptr(), d_frame_offsets.data_ptr(), num_frames, C.c_void_p(stream) if stream else None))

    def frames_ready(self):
        n = C.c_uint32(0)
        N.check_framer(self.h, self.L.adder_framer_frames_ready(self.h, C.byref(n)))
        return n.value

    def pop(self, max_frames=1 << 16):
        """write_multi_frame_bytes: all complete frames -> bytes ([n][rows][W][C] u8)."""
        out = []
        while max_frames > 0:
            step = min(max_frames, 256)
            buf = np.empty(step * self.frame_bytes, np.uint8)
            n = C.c_uint32(0)
            N.check_framer(self.h, self.L.adder_framer_pop(self.h, buf.ctypes.data, step, C.byref(n)))
            if not n.value:
                break
            out.append(buf[: n.value * self.frame_bytes].tobytes())
            max_frames -= n.value
            if n.value < step:
                break
        return b"".join(out)

    def pop_device(self, d_out, max_frames, stream=None):
        n = C.c_uint32(0)
        N.check_framer(self.h, self.L.adder_framer_pop_device(self.h, d_out.data_ptr(), max_frames, C.byref(n),
                                                              C.c_void_p(stream) if stream else None))
        return n.value

    def write_frame_bytes(self):
        buf = np.empty(self.frame_bytes, np.uint8)
        N.check_framer(self.h, self.L.adder_framer_write_frame(self.h, buf.ctypes.data))
        return buf.tobytes()

    def flush_frame_buffer(self):
        r = C.c_int(0)
        N.check_framer(self.h, self.L.adder_framer_flush(self.h, C.byref(r)))
        return bool(r.value)


def contiguous_run_segments(events):
    """Greedy split of an arbitrary event stream into segments inside which every pixel-channel's
    events are contiguous (what HipFramer.ingest needs); per-pixel order is untouched.  Streams
    produced frame by frame by the transcoder do not need this: their frame offsets already are
    such segments."""
    ev = np.ascontiguousarray(events, dtype=N.EVENT_DTYPE)
    c = np.where(ev["c"] == 0xFF, 0, ev["c"]).astype(np.int64)
    key = (ev["y"].astype(np.int64) << 24) | (ev["x"].astype(np.int64) << 8) | c
    offs = [0]
    seen = {}
    for i, k in enumerate(key.tolist()):
        j = seen.get(k)
        if j is not None and j >= offs[-1] and j != i - 1:
            offs.append(i)  # the pixel already occurred in this segment, and not just before
        seen[k] = i
    offs.append(len(ev))
    return np.array(offs, np.uint64)
