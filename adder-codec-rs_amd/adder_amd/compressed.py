"""CompressedEncoder / compressed_decode: ctypes binding of the CPU compressed sink (include/adder_compressed.h;
Encoder::new_compressed + CompressedOutput / CompressedInput of the reference).  Nothing is computed here."""
import ctypes as C

import numpy as np

from . import _native as N


def _check(h, rc):
    if rc != N.OK:
        msg = N.load().adder_compressed_last_error(h)
        raise N.AdderHipError(rc, msg.decode() if msg else "")


class CompressedEncoder:
    def __init__(self, width, height, channels=1, *, tps, ref_interval, delta_t_max, adu_interval, codec_version=3,
                 source_camera=0, time_mode=N.TIME_ABSOLUTE_T, c_thresh_max=7, write_header=True, threads=0):
        self.L = N.load()
        p = N.AdderCompressedParams()
        self.L.adder_compressed_default_params(C.byref(p), width, height, channels)
        p.codec_version, p.time_mode, p.write_header = codec_version, time_mode, int(write_header)
        p.tps, p.ref_interval, p.delta_t_max, p.adu_interval = tps, ref_interval, delta_t_max, adu_interval
        p.source_camera, p.c_thresh_max, p.threads = source_camera, c_thresh_max, threads
        self.params = p
        h = C.c_void_p()
        _check(None, self.L.adder_compressed_encoder_create(C.byref(p), C.byref(h)))
        self.h = h

    def ingest(self, events):
        events = np.ascontiguousarray(events, dtype=N.EVENT_DTYPE)
        _check(self.h, self.L.adder_compressed_encoder_ingest(self.h, events.ctypes.data, len(events)))

    def progress(self):
        a, n = C.c_uint32(0), C.c_size_t(0)
        _check(self.h, self.L.adder_compressed_encoder_progress(self.h, C.byref(a), C.byref(n)))
        return a.value, n.value

    def close(self):
        """Encoder::close_writer: compresses the partial last ADU and returns the whole stream."""
        p, n = C.c_void_p(), C.c_size_t(0)
        _check(self.h, self.L.adder_compressed_encoder_close(self.h, C.byref(p), C.byref(n)))
        return bytes((C.c_uint8 * n.value).from_address(p.value)) if n.value else b""

    def destroy(self):
        if getattr(self, "h", None):
            self.L.adder_compressed_encoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def compressed_decode(data, *, has_header=True, width=0, height=0, channels=1, ref_interval=0, adu_interval=0):
    """-> (events, params).  A stream without header (a bare CompressedOutput) needs the plane and intervals."""
    L = N.load()
    p = N.AdderCompressedParams()
    L.adder_compressed_default_params(C.byref(p), width or 1, height or 1, channels)
    p.ref_interval, p.adu_interval = ref_interval or 1, adu_interval or 1
    buf = np.frombuffer(data, np.uint8)
    n = C.c_size_t(0)
    _check(None, L.adder_compressed_decode(buf.ctypes.data, len(buf), int(has_header), C.byref(p), None, 0, C.byref(n)))
    out = np.zeros(n.value, N.EVENT_DTYPE)
    _check(None, L.adder_compressed_decode(buf.ctypes.data, len(buf), int(has_header), C.byref(p), out.ctypes.data,
                                           len(out), C.byref(n)))
    return out, p
