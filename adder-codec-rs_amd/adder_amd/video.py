"""HipVideo: the `Video<W>` surface of the reference for the framed->ADDER path,
bound to the C-ABI (include/adder_hip.h).

Mirrors adder-codec-rs/src/transcoder/source/video.rs: construction (Video::new
:350-438), time_parameters (:493-537), write_out's mode arguments (:546-636),
chunk_rows (:471-479), update_crf / update_quality_manual (:1241-1287) and
integrate_matrix (:651-778).  Every call goes through libadder_hip.so; nothing here
computes events.
"""
import ctypes as C

import numpy as np

from . import _native as N

# adder-codec-core/src/codec/rate_controller.rs:5-18 : baseline C, max C, C increase velocity
CRF = [
    (0, 0, 10), (0, 1, 9), (1, 3, 8), (2, 7, 7), (5, 9, 6),
    (6, 10, 5), (7, 13, 4), (8, 16, 3), (10, 20, 2), (15, 25, 1),
]
# feature radius column of the same table: X * min resolution, in pixels, evaluated in f32 (:5-18, :66-67)
CRF_FEATURE_RADIUS = [np.float32(1e-9)] + [np.float32(1.0) / np.float32(k) for k in (12, 14, 15, 18, 20, 25, 30, 30, 30)]
DEFAULT_CRF_QUALITY = 3


def crf_feature_radius(crf, width, height):
    """(CRF[crf][3] * plane.min_resolution() as f32) as u16 (rate_controller.rs:66-67)."""
    return int(np.float32(CRF_FEATURE_RADIUS[crf]) * np.float32(min(width, height)))


class HipVideo:
    def __init__(self, width, height, channels=1, *, row_begin=0, row_end=None,
                 time_mode=N.TIME_ABSOLUTE_T, multi_mode=N.MULTI_COLLAPSE, ref_time=255,
                 delta_t_max=7650, chunk_rows=1, max_depth=16, device_id=-1,
                 c_thresh_start=None, c_counter_start=None, pixel_mode=0):
        self.L = N.load()
        p = N.AdderHipParams()
        self.L.adder_hip_default_params(C.byref(p), width, height, channels)
        p.row_begin = row_begin
        p.row_end = height if row_end is None else row_end
        p.time_mode, p.multi_mode = time_mode, multi_mode
        p.ref_time, p.delta_t_max = ref_time, delta_t_max
        p.chunk_rows, p.max_depth, p.device_id = chunk_rows, max_depth, device_id
        p.pixel_mode = pixel_mode  # 0 = Mode::FramePerfect, 1 = Mode::Continuous
        if c_thresh_start is not None:
            p.c_thresh_start = c_thresh_start
        if c_counter_start is not None:
            p.c_counter_start = c_counter_start
        self.params = p
        self.width, self.height, self.channels = width, height, channels
        self.row_begin, self.row_end = int(p.row_begin), int(p.row_end)
        self.rows = p.row_end - p.row_begin
        self.n_units = self.rows * width * channels
        self.ref_time = ref_time
        h = C.c_void_p()
        rc = self.L.adder_hip_create(C.byref(p), C.byref(h))
        if rc != N.OK:
            msg = self.L.adder_hip_last_error(None)
            raise N.AdderHipError(rc, msg.decode() if msg else "")
        self.h = h
        self.num_chunks = self.L.adder_hip_num_chunks(self.h)
        self.max_events_per_frame = self.L.adder_hip_max_events_per_frame(self.h)
        self._out = None
        self._pinned = None

    def close(self):
        if getattr(self, "h", None):
            self.L.adder_hip_destroy(self.h)  # waits for whatever is still in flight
            self.h = None
        self._free_pinned(frames_too=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- quality controls -------------------------------------------------------------
    def set_crf_parameters(self, c_thresh_max, c_increase_velocity):
        N.check(self.h, self.L.adder_hip_set_crf_parameters(self.h, c_thresh_max, c_increase_velocity))

    def reset_c_thresh(self, baseline):
        N.check(self.h, self.L.adder_hip_reset_c_thresh(self.h, baseline))

    def update_crf(self, crf):
        """Video::update_crf (video.rs:1241-1251)."""
        base, cmax, vel = CRF[crf]
        self.set_crf_parameters(cmax, vel)
        self.set_feature_parameters(base, crf_feature_radius(crf, self.width, self.height))
        self.reset_c_thresh(base)

    def update_quality_manual(self, c_thresh_baseline, c_thresh_max, delta_t_max_multiplier, c_increase_velocity,
                              feature_c_radius=None):
        """Video::update_quality_manual (video.rs:1264-1287); feature_c_radius: the absolute pixel count
        (None leaves the current one, for callers that do not use feature-driven rate control)."""
        self.set_crf_parameters(c_thresh_max, c_increase_velocity)
        if feature_c_radius is not None:
            self.set_feature_parameters(c_thresh_baseline, int(feature_c_radius))
        else:
            self._baseline_only(c_thresh_baseline)
        N.check(self.h, self.L.adder_hip_set_delta_t_max(self.h, delta_t_max_multiplier * self.ref_time))
        self.reset_c_thresh(c_thresh_baseline)

    # ---- feature-driven rate control, ROI ------------------------------------
    def update_detect_features(self, detect_features, feature_rate_adjustment=False):
        """Video::update_detect_features (video.rs:825-837)."""
        N.check(self.h, self.L.adder_hip_update_detect_features(self.h, int(detect_features),
                                                               int(feature_rate_adjustment)))

    def set_feature_parameters(self, c_thresh_baseline, feature_c_radius):
        self._feature_radius = int(feature_c_radius)
        N.check(self.h, self.L.adder_hip_set_feature_parameters(self.h, c_thresh_baseline, int(feature_c_radius)))

    def _baseline_only(self, c_thresh_baseline):
        r = getattr(self, "_feature_radius", crf_feature_radius(DEFAULT_CRF_QUALITY, self.width, self.height))
        self.set_feature_parameters(c_thresh_baseline, r)

    def update_roi(self, roi):
        """Video::update_roi (video.rs:1291-1293); roi = (start_x, start_y, end_x, end_y) inclusive, or None."""
        if roi is None:
            N.check(self.h, self.L.adder_hip_update_roi(self.h, 0, 0, 0, 0, 0))
        else:
            N.check(self.h, self.L.adder_hip_update_roi(self.h, 1, *[int(v) for v in roi]))

    def feature_set(self):
        out = np.zeros((self.rows, self.width), np.uint8)
        N.check(self.h, self.L.adder_hip_feature_set(self.h, out.ctypes.data))
        return out

    def c_thresh_plane(self):
        out = np.zeros(self.n_units, np.uint8)
        N.check(self.h, self.L.adder_hip_c_thresh_plane(self.h, out.ctypes.data))
        return out.reshape(self.rows, self.width, self.channels)

    def last_new_features(self):
        return int(self.L.adder_hip_last_new_features(self.h))

    def set_delta_t_max(self, dtm):
        N.check(self.h, self.L.adder_hip_set_delta_t_max(self.h, dtm))

    def set_time_mode(self, tm):
        N.check(self.h, self.L.adder_hip_set_time_mode(self.h, tm))

    def enable_running_intensities(self, on=True):
        N.check(self.h, self.L.adder_hip_enable_running_intensities(self.h, int(on)))

    def running_intensities(self):
        out = np.zeros(self.n_units, np.uint8)
        N.check(self.h, self.L.adder_hip_running_intensities(self.h, out.ctypes.data))
        return out.reshape(self.rows, self.width, self.channels)

    # ---- host-buffer entry points --------------------------------------------------------
    def _host_out(self, cap):
        """Event buffer in page-locked host memory (PCIe copies at link speed)."""
        if self._out is None or len(self._out) < cap:
            self._free_pinned()
            ptr = self.L.adder_hip_alloc_pinned(cap * 12)
            if not ptr:
                raise MemoryError("adder_hip_alloc_pinned failed")
            self._pinned = ptr
            buf = (C.c_uint8 * (cap * 12)).from_address(ptr)
            self._out = np.frombuffer(buf, dtype=N.EVENT_DTYPE)
        return self._out

    def _free_pinned(self, frames_too=False):
        if frames_too:
            for ptr in getattr(self, "_pinned_frames", []):
                self.L.adder_hip_free_pinned(ptr)
            self._pinned_frames = []
        if getattr(self, "_pinned", None):
            self._out = None
            self.L.adder_hip_free_pinned(self._pinned)
            self._pinned = None

    def integrate_matrix(self, frame, time_spanned=None, want_chunks=False, out_cap=None):
        """One frame in, events out in the reference's order (video.rs:651-740)."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8).reshape(self.rows, self.width * self.channels)
        ts = float(self.ref_time) if time_spanned is None else float(time_spanned)
        cap = self.max_events_per_frame if out_cap is None else out_cap
        out = self._host_out(cap)
        n = C.c_size_t(0)
        chunks = np.zeros(self.num_chunks + 1, np.uint32)
        rc = self.L.adder_hip_integrate(self.h, frame.ctypes.data, frame.strides[0], ts, out.ctypes.data, cap,
                                        C.byref(n), chunks.ctypes.data)
        self.last_required = n.value
        N.check(self.h, rc)
        ev = _copy_events(out, n.value)
        return (ev, chunks) if want_chunks else ev

    def integrate_sparse(self, steps, out_cap=None):
        """Event-camera sources: one integrate_for_px call per step (N.SPARSE_STEP_DTYPE), in order -> events."""
        steps = np.ascontiguousarray(steps, dtype=N.SPARSE_STEP_DTYPE)
        cap = max(len(steps), 1) * (self.params.max_depth + 3) if out_cap is None else out_cap
        out = np.zeros(cap, N.EVENT_DTYPE)
        n = C.c_size_t(0)
        rc = self.L.adder_hip_integrate_sparse(self.h, steps.ctypes.data, len(steps), out.ctypes.data, cap, C.byref(n))
        self.last_required = n.value
        N.check(self.h, rc)
        return out[: n.value].copy()

    # ---- per-frame ring: submit returns once the work is queued, collect hands back the oldest frame --------
    def frames_configure(self, slots=0, events_per_slot=0):
        N.check(self.h, self.L.adder_hip_frames_configure(self.h, slots, events_per_slot))

    def pinned_frame(self):
        """A page-locked [rows][width*channels] u8 array to put input frames in (uploads then run asynchronously)."""
        ptr = self.L.adder_hip_alloc_pinned(self.n_units)
        if not ptr:
            raise MemoryError("adder_hip_alloc_pinned failed")
        self._pinned_frames = getattr(self, "_pinned_frames", []) + [ptr]
        buf = (C.c_uint8 * self.n_units).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.rows, self.width * self.channels)

    def frame_submit(self, frame, time_spanned=None):
        frame = np.ascontiguousarray(frame, dtype=np.uint8).reshape(self.rows, self.width * self.channels)
        ts = float(self.ref_time) if time_spanned is None else float(time_spanned)
        self._inflight = getattr(self, "_inflight", []) + [frame]  # keep the source alive until it is collected
        N.check(self.h, self.L.adder_hip_frame_submit(self.h, frame.ctypes.data, frame.strides[0], ts))

    def frame_collect(self, want_chunks=False, copy=True):
        ev, n, ch = C.c_void_p(), C.c_size_t(0), C.c_void_p()
        rc = self.L.adder_hip_frame_collect(self.h, C.byref(ev), C.byref(n), C.byref(ch))
        if rc == N.E_BAD_PARAMS:  # nothing in flight, or the slot holds the other format: nothing was consumed
            N.check(self.h, rc)
        if getattr(self, "_inflight", None):
            self._inflight.pop(0)
        self.last_required = n.value
        N.check(self.h, rc)
        events = np.frombuffer((C.c_uint8 * (n.value * 12)).from_address(ev.value), dtype=N.EVENT_DTYPE) if n.value \
            else np.zeros(0, N.EVENT_DTYPE)
        chunks = np.frombuffer((C.c_uint32 * (self.num_chunks + 1)).from_address(ch.value), dtype=np.uint32)
        if copy:
            events, chunks = events.copy(), chunks.copy()
        return (events, chunks) if want_chunks else events

    def frames_set_format(self, wire_records):
        """The ring hands out 9 / 11-byte wire records (what RawOutput writes) instead of AdderEvents."""
        N.check(self.h, self.L.adder_hip_frames_set_format(self.h, 1 if wire_records else 0))

    def frame_collect_wire(self, want_chunks=False, copy=True):
        """-> (wire bytes of the oldest frame in flight, its number of events[, chunk offsets in events])."""
        by, nb, n, ch = C.c_void_p(), C.c_size_t(0), C.c_size_t(0), C.c_void_p()
        rc = self.L.adder_hip_frame_collect_wire(self.h, C.byref(by), C.byref(nb), C.byref(n), C.byref(ch))
        if rc == N.E_BAD_PARAMS:
            N.check(self.h, rc)
        if getattr(self, "_inflight", None):
            self._inflight.pop(0)
        self.last_required = n.value
        N.check(self.h, rc)
        data = np.frombuffer((C.c_uint8 * nb.value).from_address(by.value), dtype=np.uint8) if nb.value else np.zeros(0, np.uint8)
        chunks = np.frombuffer((C.c_uint32 * (self.num_chunks + 1)).from_address(ch.value), dtype=np.uint32)
        if copy:
            data, chunks = data.copy(), chunks.copy()
        return (data, n.value, chunks) if want_chunks else (data, n.value)

    def frames_in_flight(self):
        return int(self.L.adder_hip_frames_in_flight(self.h))

    def integrate_batch(self, frames, time_spanned=None, out_cap=None):
        """T frames (host array [T, rows, W, C]) -> (events, frame_offsets[T+1])."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(len(frames), self.n_units)
        T = frames.shape[0]
        ts = float(self.ref_time) if time_spanned is None else float(time_spanned)
        cap = min(self.max_events_per_frame, 4 * self.n_units) * T if out_cap is None else out_cap
        out = self._host_out(cap)
        n = C.c_size_t(0)
        offs = np.zeros(T + 1, np.uint64)
        rc = self.L.adder_hip_integrate_batch(self.h, frames.ctypes.data, T, self.n_units,
                                              self.width * self.channels, ts, out.ctypes.data, cap,
                                              C.byref(n), offs.ctypes.data)
        self.last_required = n.value
        N.check(self.h, rc)
        return _copy_events(out, n.value), offs

    def integrate_batch_raw(self, frames, time_spanned=None, out_cap_bytes=None):
        """T frames (host array) -> (wire bytes of the events as the raw `.adder` sink writes them,
        number of events, frame_offsets[T+1] in events).  Serialisation happens on the device."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(len(frames), self.n_units)
        T = frames.shape[0]
        ts = float(self.ref_time) if time_spanned is None else float(time_spanned)
        rec = 9 if self.channels == 1 else 11
        cap = (min(self.max_events_per_frame, 4 * self.n_units) * T * rec) if out_cap_bytes is None else out_cap_bytes
        out = self._host_out((cap + 11) // 12).view(np.uint8)
        nb, ne = C.c_size_t(0), C.c_size_t(0)
        offs = np.zeros(T + 1, np.uint64)
        rc = self.L.adder_hip_integrate_batch_raw(self.h, frames.ctypes.data, T, self.n_units,
                                                  self.width * self.channels, ts, out.ctypes.data, cap,
                                                  C.byref(nb), C.byref(ne), offs.ctypes.data)
        self.last_required = ne.value
        N.check(self.h, rc)
        return out[: nb.value].tobytes(), ne.value, offs

    def stream_submit(self, frames, time_spanned=None, out_cap_events=None):
        """Pipelined raw transcode: upload + integrate this batch, queue its serialisation + download."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(len(frames), self.n_units)
        T = frames.shape[0]
        ts = float(self.ref_time) if time_spanned is None else float(time_spanned)
        cap = min(self.max_events_per_frame, 4 * self.n_units) * T if out_cap_events is None else out_cap_events
        N.check(self.h, self.L.adder_hip_stream_submit(self.h, frames.ctypes.data, T, self.n_units,
                                                       self.width * self.channels, ts, cap))

    def stream_collect(self, copy=True):
        """Oldest batch in flight -> (wire bytes, number of events, frame_offsets[T+1])."""
        p, po = C.c_void_p(), C.c_void_p()
        nb, ne = C.c_size_t(0), C.c_size_t(0)
        rc = self.L.adder_hip_stream_collect(self.h, C.byref(p), C.byref(nb), C.byref(ne), C.byref(po))
        N.check(self.h, rc)
        buf = (C.c_uint8 * nb.value).from_address(p.value) if nb.value else b""
        return (bytes(buf) if copy else buf), ne.value, po.value

    def wire_events_device(self, d_events, n_events, d_out, stream=None):
        """n_events events in HBM -> wire bytes in HBM (uint8 CUDA tensor); returns the byte count."""
        nb = C.c_size_t(0)
        N.check(self.h, self.L.adder_hip_wire_events_device(
            self.h, d_events.data_ptr(), n_events, d_out.data_ptr(), d_out.numel() * d_out.element_size(),
            C.byref(nb), C.c_void_p(stream) if stream else None))
        return nb.value

    # ---- device-resident entry points (torch tensors provide the HBM buffers) ------------------
    def integrate_device(self, d_frames, d_events, d_offsets, time_spanned=None, stream=None):
        """Queues T frames resident in HBM.  d_frames: uint8 CUDA tensor [T, n_units];
        d_events: uint8 CUDA tensor of 12*cap bytes; d_offsets: int64 CUDA tensor [T+1]."""
        T = d_frames.shape[0]
        assert d_frames.is_contiguous() and d_frames.numel() == T * self.n_units
        assert d_offsets.numel() >= T + 1 and d_offsets.element_size() == 8
        cap = d_events.numel() * d_events.element_size() // 12
        ts = float(self.ref_time) if time_spanned is None else float(time_spanned)
        N.check(self.h, self.L.adder_hip_integrate_device(
            self.h, d_frames.data_ptr(), T, ts, d_events.data_ptr(), cap, d_offsets.data_ptr(),
            C.c_void_p(stream) if stream else None))

    def integrate_wire_device(self, d_frames, d_wire, d_offsets, time_spanned=None, stream=None):
        """As integrate_device, but d_wire (uint8 CUDA tensor) receives the raw sink's 9 / 11-byte records back to back
        (adder_hip_integrate_wire_device); d_offsets still counts events."""
        T = d_frames.shape[0]
        assert d_frames.is_contiguous() and d_frames.numel() == T * self.n_units
        assert d_offsets.numel() >= T + 1 and d_offsets.element_size() == 8
        ts = float(self.ref_time) if time_spanned is None else float(time_spanned)
        N.check(self.h, self.L.adder_hip_integrate_wire_device(
            self.h, d_frames.data_ptr(), T, ts, d_wire.data_ptr(), d_wire.numel() * d_wire.element_size(), d_offsets.data_ptr(),
            C.c_void_p(stream) if stream else None))

    # ---- records over the wire (include/adder_hip.h): a band ships its parked records, root expands them ----
    def band_segments(self):
        return int(self.L.adder_hip_band_segments(self.h))

    def integrate_records_device(self, d_frames, d_offsets, time_spanned=None, stream=None):
        """Queues T <= chunk_frames() frames WITHOUT expanding them; returns the AdderBandRecords description (device
        pointers into this context's scratch, valid until its next batch).  finish() then completes the batch;
        last_batch_records() is the number of records behind d_records."""
        T = d_frames.shape[0]
        assert d_frames.is_contiguous() and d_frames.numel() == T * self.n_units
        assert d_offsets.numel() >= T + 1 and d_offsets.element_size() == 8
        ts = float(self.ref_time) if time_spanned is None else float(time_spanned)
        rec = N.AdderBandRecords()
        N.check(self.h, self.L.adder_hip_integrate_records_device(
            self.h, d_frames.data_ptr(), T, ts, d_offsets.data_ptr(), C.c_void_p(stream) if stream else None, C.byref(rec)))
        return rec

    def expand_records_device(self, bands, d_merged, merged_base, d_merged_offsets, stream=None):
        """On root, after its own integrate_records_device + finish of the same frames: bands = AdderBandRecords in raster
        order (pointers in this device's memory).  Appends to d_merged at event index merged_base; d_merged_offsets
        (int64 / uint64 CUDA tensor, >= T + 1 entries) receives the absolute frame offsets."""
        arr = (N.AdderBandRecords * len(bands))(*bands)
        cap = d_merged.numel() * d_merged.element_size() // 12
        N.check(self.h, self.L.adder_hip_expand_records_device(
            self.h, C.byref(arr), len(bands), d_merged.data_ptr(), cap, int(merged_base), d_merged_offsets.data_ptr(),
            C.c_void_p(stream) if stream else None))

    def expand_records_wire_device(self, bands, d_wire, merged_base, d_merged_offsets, stream=None):
        """As expand_records_device with the raw sink's 9 / 11-byte records as the output (d_wire: uint8 CUDA tensor; event
        k of the merged stream at byte k * record size; merged_base and the offsets count events)."""
        arr = (N.AdderBandRecords * len(bands))(*bands)
        N.check(self.h, self.L.adder_hip_expand_records_wire_device(
            self.h, C.byref(arr), len(bands), d_wire.data_ptr(), d_wire.numel() * d_wire.element_size(), int(merged_base),
            d_merged_offsets.data_ptr(), C.c_void_p(stream) if stream else None))

    def records_to_wire(self, rec, n_records, d_dst, stream=None):
        """One contiguous image of the batch `rec` describes (adder_hip_records_to_wire) into the uint8 CUDA tensor d_dst."""
        N.check(self.h, self.L.adder_hip_records_to_wire(self.h, C.byref(rec), int(n_records), d_dst.data_ptr(),
                                                         d_dst.numel() * d_dst.element_size(),
                                                         C.c_void_p(stream) if stream else None))

    def sync_last_batch_stream(self):
        N.check(self.h, self.L.adder_hip_sync_last_batch_stream(self.h))

    def expand_status(self, stream=None):
        N.check(self.h, self.L.adder_hip_expand_status(self.h, C.c_void_p(stream) if stream else None))

    def finish(self):
        n = C.c_size_t(0)
        rc = self.L.adder_hip_finish(self.h, C.byref(n))
        self.last_required = n.value
        N.check(self.h, rc)
        return n.value

    def last_batch_ms(self):
        return float(self.L.adder_hip_last_batch_ms(self.h))

    def set_launch_timing(self, on=True):
        N.check(self.h, self.L.adder_hip_set_launch_timing(self.h, int(on)))

    def last_batch_kernel(self):
        """ADDER_KERNEL_* of the batch queued last (include/adder_hip.h)."""
        return int(self.L.adder_hip_last_batch_kernel(self.h))

    def launch_plan_settled(self):
        return bool(self.L.adder_hip_launch_plan_settled(self.h))

    def last_launch_avg_us(self):
        return float(self.L.adder_hip_last_launch_avg_us(self.h))

    def last_post_avg_us(self):
        return float(self.L.adder_hip_last_post_avg_us(self.h))

    def last_post_chunks(self):
        return int(self.L.adder_hip_last_post_chunks(self.h))

    def chunk_frames(self):
        return int(self.L.adder_hip_chunk_frames(self.h))

    def last_batch_records(self):
        return int(self.L.adder_hip_last_batch_records(self.h))

    def last_launch_frames(self):
        return float(self.L.adder_hip_last_launch_frames(self.h))

    def set_frames_per_launch(self, frames):
        N.check(self.h, self.L.adder_hip_set_frames_per_launch(self.h, frames))

    def reset(self):
        """Back to the freshly constructed state (Video::new); parameters are kept."""
        N.check(self.h, self.L.adder_hip_reset(self.h))

    def merge_streams_device(self, d_stage, d_rank_offsets, world, T, d_out, d_merged_offsets=None, stream=None,
                             merged_base=0):
        """Multi-GPU merge (include/adder_hip.h): `world` frame-major streams laid back to back in d_stage
        (rank offsets: int64 CUDA tensor [world, T+1]) -> one frame-major stream in d_out, rank order inside
        every frame.  Asynchronous; check_status() reports a capacity overflow."""
        import torch
        need = self.L.adder_hip_merge_work_bytes(world, T)
        if getattr(self, "_merge_work", None) is None or self._merge_work.numel() < need:
            self._merge_work = torch.empty(max(need, 8), dtype=torch.uint8, device=d_out.device)
        cap = d_out.numel() * d_out.element_size() // 12
        # (merged_base: d_out / d_merged_offsets are where a CHUNK of a longer merged stream begins)
        N.check(self.h, self.L.adder_hip_merge_streams_device_at(
            self.h, d_stage.data_ptr(), d_rank_offsets.data_ptr(), world, T, self._merge_work.data_ptr(),
            d_out.data_ptr(), cap, None if d_merged_offsets is None else d_merged_offsets.data_ptr(), merged_base,
            C.c_void_p(stream) if stream else None))

    def check_status(self, stream=None):
        N.check(self.h, self.L.adder_hip_check_status(self.h, C.c_void_p(stream) if stream else None))

    def chunk_offsets_device(self, d_events_ptr, n_events, d_chunk_offsets, stream=None):
        N.check(self.h, self.L.adder_hip_chunk_offsets_device(
            self.h, d_events_ptr, n_events, d_chunk_offsets.data_ptr(), C.c_void_p(stream) if stream else None))


def _copy_events(buf, n):
    """Plain memcpy of the first n 12-byte records (a structured-dtype copy goes field by field)."""
    return buf[:n].view(np.uint32).copy().view(N.EVENT_DTYPE)


def synth_clip_device(d_dst, content, width, height, channels, *, row_begin=0, rows=None, frame_begin=0,
                      num_frames=1, seed=0xADDE5EED, stream=None):
    """Fills a uint8 CUDA tensor [num_frames, rows, width, channels] with SURVEY 8(d) content."""
    rows = height - row_begin if rows is None else rows
    assert d_dst.numel() == num_frames * rows * width * channels
    rc = N.load().adder_hip_synth_clip_device(d_dst.data_ptr(), content, seed, width, height, channels,
                                              row_begin, rows, frame_begin, num_frames,
                                              C.c_void_p(stream) if stream else None)
    N.check(None, rc)


# ---- raw `.adder` sink through the C-ABI ------------------------------------------------------
def raw_header(codec_version, width, height, channels, tps, ref_interval, delta_t_max,
               source_camera=0, time_mode=N.TIME_ABSOLUTE_T, adu_interval=0):
    buf = np.zeros(64, np.uint8)
    n = N.load().adder_raw_header(buf.ctypes.data, codec_version, width, height, channels, tps,
                                  ref_interval, delta_t_max, source_camera, time_mode, adu_interval)
    return buf[:n].tobytes()


def raw_events(events, channels):
    events = np.ascontiguousarray(events, dtype=N.EVENT_DTYPE)
    buf = np.zeros(len(events) * 11 + 16, np.uint8)
    n = N.load().adder_raw_events(buf.ctypes.data, events.ctypes.data, len(events), channels)
    return buf[:n].tobytes()


def raw_eof():
    buf = np.zeros(16, np.uint8)
    n = N.load().adder_raw_eof(buf.ctypes.data)
    return buf[:n].tobytes()
