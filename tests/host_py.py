"""ctypes access to adder-codec-rs_amd/host/libadder_host.so (the C++ mirror's test facade)."""
import ctypes as C
import os

import numpy as np

import adder_amd

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(_ROOT, "adder-codec-rs_amd", "host", "libadder_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        adder_amd.load()  # libadder_hip.so (and torch's HIP runtime) first
        L = C.CDLL(LIB)
        L.adder_host_last_error.restype = C.c_char_p
        L.adder_host_transcode_raw.restype = C.c_longlong
        L.adder_host_transcode_raw.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                               C.c_float, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                               C.c_uint32, C.c_int, C.c_char_p, C.POINTER(C.c_uint32)]
        L.adder_host_transcode.restype = C.c_longlong
        L.adder_host_transcode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                           C.c_float, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                           C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_char_p, C.POINTER(C.c_uint32)]
        L.adder_host_transcode_features.restype = C.c_longlong
        L.adder_host_transcode_features.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                    C.c_float, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                    C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_void_p]
        L.adder_host_prophesee.restype = C.c_longlong
        L.adder_host_prophesee.argtypes = [C.c_void_p, C.c_size_t, C.c_uint16, C.c_uint16, C.c_uint32, C.c_void_p,
                                           C.c_size_t, C.POINTER(C.c_uint32)]
        L.adder_host_frame_events.restype = C.c_longlong
        L.adder_host_frame_events.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint8,
                                              C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_size_t]
        L.adder_host_decode_raw.restype = C.c_longlong
        L.adder_host_decode_raw.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        L.adder_host_crf_parameters.restype = C.c_int
        L.adder_host_crf_parameters.argtypes = [C.c_int, C.c_uint16, C.c_uint16, C.c_void_p]
        L.adder_host_encode_raw.restype = C.c_longlong
        L.adder_host_encode_raw.argtypes = [C.c_uint8, C.c_uint16, C.c_uint16, C.c_uint8, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_int,
                                            C.c_void_p, C.c_size_t]
        L.adder_host_simulproc.restype = C.c_longlong
        L.adder_host_simulproc.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_uint32,
                                           C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_int32, C.c_uint8, C.c_int,
                                           C.c_char_p, C.c_char_p]
        L.adder_host_encode_events.restype = C.c_longlong
        L.adder_host_encode_events.argtypes = [C.c_void_p, C.c_size_t, C.c_uint16, C.c_uint16, C.c_uint8, C.c_uint32,
                                               C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                               C.c_size_t, C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def encode_events(events, w, h, c, dtm, *, interleaved=False, manual=None, clock=None):
    """Encoder (raw) with EventOrder::Interleaved / EventDrop::Manual(target_rate, alpha) -> (file bytes, queued)."""
    events = np.ascontiguousarray(events, adder_amd.EVENT_DTYPE)
    dst = np.zeros(64 + 11 * len(events) + 16, np.uint8)
    clk = None if clock is None else np.ascontiguousarray(clock, np.float64)
    q = C.c_size_t(0)
    n = lib().adder_host_encode_events(events.ctypes.data, len(events), w, h, c, dtm, int(interleaved),
                                       int(manual is not None), manual[0] if manual else 0.0,
                                       manual[1] if manual else 0.0, None if clk is None else clk.ctypes.data,
                                       dst.ctypes.data, len(dst), C.byref(q))
    assert n >= 0, err()
    return dst[:n].tobytes(), q.value


def err():
    return lib().adder_host_last_error().decode()


def decode_raw(raw):
    buf = np.frombuffer(raw, np.uint8)
    meta = np.zeros(10, np.uint32)
    n = lib().adder_host_decode_raw(buf.ctypes.data, len(buf), meta.ctypes.data, None, 0)
    assert n >= 0, err()
    ev = np.zeros(n, adder_amd.EVENT_DTYPE)
    lib().adder_host_decode_raw(buf.ctypes.data, len(buf), meta.ctypes.data, ev.ctypes.data, n)
    return meta, ev


def encode_raw(version, w, h, c, tps, ref, dtm, cam, tm, events, close=True):
    events = np.ascontiguousarray(events, adder_amd.EVENT_DTYPE)
    dst = np.zeros(64 + 11 * len(events) + 16, np.uint8)
    n = lib().adder_host_encode_raw(version, w, h, c, tps, ref, dtm, cam, tm, events.ctypes.data, len(events),
                                    int(close), dst.ctypes.data, len(dst))
    assert n >= 0, err()
    return dst[:n].tobytes()


def crf_parameters(crf, w, h):
    out = np.zeros(4, np.uint32)
    assert lib().adder_host_crf_parameters(crf, w, h, out.ctypes.data) == 0, err()
    return tuple(int(x) for x in out)


def transcode_raw(frames, *, color_input=False, fps=30.0, crf=-1, ref_time=255, delta_t_max=7650, time_mode=1,
                  multi_mode=1, chunk_rows=1, encoder_crf=-1, out_path=None):
    frames = np.ascontiguousarray(frames, np.uint8)
    T, H, W, Cin = frames.shape
    chunks = C.c_uint32(0)
    n = lib().adder_host_transcode_raw(frames.ctypes.data, T, W, H, Cin, int(color_input), fps, crf, ref_time,
                                       delta_t_max, time_mode, multi_mode, chunk_rows, encoder_crf, out_path.encode(),
                                       C.byref(chunks))
    if n < 0:
        raise RuntimeError(err())
    return n, chunks.value


def transcode_features(frames, *, color_input=False, fps=30.0, crf=3, ref_time=255, delta_t_max=7650, time_mode=1,
                       multi_mode=1, chunk_rows=1, detect=True, rate_adjustment=True, roi=None, out_path=None):
    """Framed + update_detect_features / update_roi on its Video -> raw file; returns (events, feature set)."""
    frames = np.ascontiguousarray(frames, np.uint8)
    T, H, W, Cin = frames.shape
    fs = np.zeros((H, W), np.uint8)
    r = None if roi is None else (C.c_uint16 * 4)(*roi)
    n = lib().adder_host_transcode_features(frames.ctypes.data, T, W, H, Cin, int(color_input), fps, crf,
                                            ref_time, delta_t_max, time_mode, multi_mode, chunk_rows, int(detect),
                                            int(rate_adjustment), r, out_path.encode(), fs.ctypes.data)
    if n < 0:
        raise RuntimeError(err())
    return n, fs


DVS_DTYPE = np.dtype([("t", "<u4"), ("x", "<u2"), ("y", "<u2"), ("p", "u1"), ("pad", "u1")])


def prophesee(dvs, width, height, ref_time):
    """Prophesee source of the C++ mirror over in-memory DVS events -> (all ADDER events in order, consume() calls)."""
    dvs = np.ascontiguousarray(dvs, DVS_DTYPE)
    cap = max(4096, (len(dvs) * 2 + 3 * width * height) * 8)
    out = np.zeros(cap, adder_amd.EVENT_DTYPE)
    calls = C.c_uint32(0)
    n = lib().adder_host_prophesee(dvs.ctypes.data, len(dvs), width, height, ref_time, out.ctypes.data, cap, C.byref(calls))
    if n < 0:
        raise RuntimeError(err())
    assert n <= cap
    return out[:n].copy(), calls.value


DAVIS_DVS_DTYPE = np.dtype([("t", "<i8"), ("x", "<u2"), ("y", "<u2"), ("on", "u1"), ("pad", "u1", (3,))])
DAVIS_META_DTYPE = np.dtype([("c", "<f8"), ("start", "<i8"), ("end", "<i8"), ("n_before", "<u4"), ("n_after", "<u4")])


def davis(packets, width, height, mode, *, tps, ref_time, delta_t_max, time_mode=1, crf=-1):
    """Davis source of the C++ mirror over in-memory EDI output.  packets: list of dicts {frame [h, w] f64, c, start, end,
    before, after (arrays of DAVIS_DVS_DTYPE)} -> (every event the source ingested, in order; events consume() returned)."""
    P = len(packets)
    crf = -1 if crf is None else crf
    frames = np.ascontiguousarray(np.stack([p["frame"] for p in packets]).astype(np.float64))
    meta = np.zeros(P, DAVIS_META_DTYPE)
    dvs = []
    for k, p in enumerate(packets):
        meta[k] = (p["c"], p["start"], p["end"], len(p["before"]), len(p["after"]))
        dvs += [np.ascontiguousarray(p["before"], DAVIS_DVS_DTYPE), np.ascontiguousarray(p["after"], DAVIS_DVS_DTYPE)]
    dvs = np.concatenate(dvs) if dvs else np.zeros(0, DAVIS_DVS_DTYPE)
    cap = max(4096, (len(dvs) * 4 + (3 * P + 2) * width * height) * 6)
    out = np.zeros(cap, adder_amd.EVENT_DTYPE)
    ret = C.c_ulonglong(0)
    fn = lib().adder_host_davis
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint16, C.c_uint16, C.c_int, C.c_uint32, C.c_uint32,
                   C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_ulonglong)]
    n = fn(frames.ctypes.data, meta.ctypes.data, dvs.ctypes.data, P, width, height, mode, tps, ref_time, delta_t_max,
           time_mode, crf, out.ctypes.data, cap, C.byref(ret))
    if n < 0:
        raise RuntimeError(err())
    assert n <= cap
    return out[:n].copy(), ret.value


def frame_events(events, chunk_offsets, width, height, channels, *, tps, ref_interval, delta_t_max, codec_version,
                 time_mode, chunk_rows, output_fps=0.0, framer_mode=0, view_mode=0, source_type=0, practical_d_max=0.0,
                 flushes=0, cap=1 << 26, value_type=0):
    """FramerBuilder ... finish() -> ingest_events_events -> write_multi_frame_bytes (+ flushes) of the C++ mirror."""
    ev = np.ascontiguousarray(events, adder_amd.EVENT_DTYPE)
    offs = np.ascontiguousarray(chunk_offsets, np.uint64)
    params = np.array([tps, ref_interval, delta_t_max, codec_version, time_mode, framer_mode, view_mode, source_type,
                       chunk_rows, value_type], np.uint32)
    out = np.zeros(cap, np.uint8)
    n = lib().adder_host_frame_events(ev.ctypes.data, offs.ctypes.data, len(offs) - 1, width, height, channels,
                                      params.ctypes.data, output_fps, practical_d_max, flushes, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError(err())
    return out[:n].tobytes()


def transcode_compressed(frames, *, color_input=False, fps=30.0, crf=-1, ref_time=255, delta_t_max=7650, time_mode=1,
                         multi_mode=1, chunk_rows=1, encoder_crf=-1, adu_interval=30, out_path=None):
    """Framed -> Video (GPU) -> Encoder::new_compressed -> file; returns the number of events ingested."""
    frames = np.ascontiguousarray(frames, np.uint8)
    T, H, W, Cin = frames.shape
    chunks = C.c_uint32(0)
    n = lib().adder_host_transcode(frames.ctypes.data, T, W, H, Cin, int(color_input), fps, crf, ref_time,
                                   delta_t_max, time_mode, multi_mode, chunk_rows, encoder_crf, 0, adu_interval,
                                   out_path.encode(), C.byref(chunks))
    if n < 0:
        raise RuntimeError(err())
    return n


def simulproc(frames, *, fps, crf, ref_time, delta_t_max, time_mode, multi_mode, chunk_rows=1, frame_count_max=0,
              framer_codec_version=1, framer_time_mode=1, out_events, out_frames):
    """SimulProcessor::new::<u8> + run over gray frames [T][H][W] (the reference's `dark` test)."""
    frames = np.ascontiguousarray(frames, np.uint8)
    T, H, W = frames.shape[:3]
    n = lib().adder_host_simulproc(frames.ctypes.data, T, W, H, fps, crf, ref_time, delta_t_max, time_mode,
                                   multi_mode, chunk_rows, frame_count_max, framer_codec_version, framer_time_mode,
                                   out_events.encode(), out_frames.encode())
    assert n >= 0, err()
    return n
