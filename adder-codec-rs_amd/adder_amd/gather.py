"""HipGather: ctypes binding of libadder_rccl.so (include/adder_gather.h) -- the multi-GPU ordered
gather of the row bands' event streams over RCCL, as a Rust host would call it (no torch inside).

The communicator is created from an RCCL unique id that rank 0 makes and the caller ships to the other
ranks (bench.py / the tests broadcast it with torch.distributed; a Rust host would use its own means).
"""
import ctypes as C
import os

import numpy as np

from . import _native as N

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libadder_rccl.so")
UNIQUE_ID_BYTES = 128

_vp, _i32, _u32, _sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
SYMBOLS = {
    "adder_gather_unique_id": (_i32, [_vp]),
    "adder_gather_create": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "adder_gather_create_from_id": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "adder_gather_destroy": (None, [_vp]),
    "adder_gather_last_error": (C.c_char_p, [_vp]),
    "adder_gather_world": (_i32, [_vp]),
    "adder_gather_events": (_i32, [_vp, _vp, _vp, _u32, _i32, _vp, _sz, _vp, C.POINTER(_sz), _vp]),
    "adder_gather_events_at": (_i32, [_vp, _vp, _vp, _u32, _i32, _vp, _sz, C.c_uint64, _vp, C.POINTER(_sz), _vp]),
    "adder_gather_layout": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "adder_gather_records_at": (_i32, [_vp, _vp, C.c_uint64, C.c_uint64, _i32, _vp, _sz, C.c_uint64, _vp, C.POINTER(_sz), _vp]),
    "adder_gather_create_with_transport": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "adder_gather_local_group_create": (_i32, [_i32, C.POINTER(_vp)]),
    "adder_gather_local_group_destroy": (None, [_vp]),
    "adder_gather_create_local": (_i32, [_vp, _vp, _i32, C.POINTER(_vp)]),
    "adder_gather_records_begin": (_i32, [_vp, _i32, _vp, _sz, C.c_uint64, _vp, _vp]),
    "adder_gather_records_begin_wire": (_i32, [_vp, _i32, _vp, _sz, C.c_uint64, _vp, _vp]),
    "adder_gather_records_push": (_i32, [_vp, _vp, C.c_uint64, C.c_uint64]),
    "adder_gather_records_end": (_i32, [_vp, C.POINTER(_sz), C.POINTER(C.c_uint64)]),
    "adder_gather_records_host_us": (C.c_double, [_vp]),
    "adder_gather_host_sink_open": (_i32, [_vp, _vp, C.c_uint64, C.c_uint64, _vp]),
    "adder_gather_host_sink_chunk": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "adder_gather_host_sink_close": (_i32, [_vp, C.POINTER(C.c_uint64), _vp]),
    "adder_host_image_open": (_i32, [C.c_char_p, C.c_uint64, _i32, C.POINTER(_vp)]),
    "adder_host_image_host_ptr": (_vp, [_vp]),
    "adder_host_image_device_ptr": (_vp, [_vp]),
    "adder_host_image_close": (_i32, [_vp, C.c_int64, _i32]),
}
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    N.load()  # libadder_hip.so first (and torch before it, see _native.load): one HIP / RCCL copy per process
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not found: build it with `make -C adder-codec-rs_amd`")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def unique_id():
    """ncclGetUniqueId -> 128 bytes (rank 0 calls this and ships the bytes to the other ranks)."""
    buf = np.zeros(UNIQUE_ID_BYTES, np.uint8)
    rc = load().adder_gather_unique_id(buf.ctypes.data)
    if rc != N.OK:
        raise N.AdderHipError(rc, (load().adder_gather_last_error(None) or b"").decode())
    return buf.tobytes()


class LocalGroup:
    """adder_gather_local_group_*: the rendezvous of an in-process transport whose ranks are threads (one HipVideo each;
    where a box has one GPU and RCCL refuses two ranks on it).  HipGather(video, None, rank, world, local=group)."""

    def __init__(self, world):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.adder_gather_local_group_create(world, C.byref(h))
        if rc != N.OK:
            raise N.AdderHipError(rc, "local group")
        self.h, self.world = h, world

    def close(self):
        if getattr(self, "h", None):
            self.L.adder_gather_local_group_destroy(self.h)
            self.h = None


class HostImage:
    """adder_host_image_*: the .adder image the ranks' sinks store into -- a POSIX shared-memory file mapped into the
    process and registered with HIP."""

    def __init__(self, name, nbytes, create):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.adder_host_image_open(name.encode(), int(nbytes), 1 if create else 0, C.byref(h))
        if rc != N.OK:
            raise N.AdderHipError(rc, (self.L.adder_gather_last_error(None) or b"").decode())
        self.h, self.name, self.nbytes = h, name, int(nbytes)
        self.host_ptr = self.L.adder_host_image_host_ptr(h)
        self.device_ptr = self.L.adder_host_image_device_ptr(h)

    def host_array(self):
        return np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.host_ptr))

    def close(self, final_bytes=-1, unlink=False):
        if getattr(self, "h", None):
            rc = self.L.adder_host_image_close(self.h, int(final_bytes), 1 if unlink else 0)
            self.h = None
            if rc != N.OK:
                raise N.AdderHipError(rc, "closing the host image")


class HipGather:
    def __init__(self, video, uid, rank, world, local=None):
        self.L = load()
        self.video = video  # keeps the context alive
        h = C.c_void_p()
        if local is not None:
            self._group = local
            rc = self.L.adder_gather_create_local(video.h, local.h, rank, C.byref(h))
        else:
            idbuf = np.frombuffer(uid, np.uint8).copy()
            rc = self.L.adder_gather_create_from_id(video.h, idbuf.ctypes.data, rank, world, C.byref(h))
        if rc != N.OK:
            raise N.AdderHipError(rc, (self.L.adder_gather_last_error(None) or b"").decode())
        self.h = h
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "h", None):
            self.L.adder_gather_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != N.OK:
            raise N.AdderHipError(rc, (self.L.adder_gather_last_error(self.h) or b"").decode())

    def gather_events(self, d_events, d_offsets, T, root=0, d_merged=None, d_merged_offsets=None, stream=None):
        """d_events / d_offsets: this rank's stream (CUDA tensors); on root d_merged (uint8/int32 CUDA tensor
        of 12*cap bytes) and d_merged_offsets (int64 [T+1]) receive the merged stream.  Returns its length."""
        n = C.c_size_t(0)
        cap = 0 if d_merged is None else d_merged.numel() * d_merged.element_size() // 12
        rc = self.L.adder_gather_events(
            self.h, d_events.data_ptr(), d_offsets.data_ptr(), T, root,
            None if d_merged is None else d_merged.data_ptr(), cap,
            None if d_merged_offsets is None else d_merged_offsets.data_ptr(), C.byref(n),
            C.c_void_p(stream) if stream else None)
        self.last_required = n.value
        self._check(rc)
        return n.value

    def gather_events_at(self, d_events, d_offsets, frame_begin, T, root, d_merged, merged_base, d_merged_offsets, stream=None):
        """One chunk (frames [frame_begin, frame_begin + T) of this rank's stream) appended to the merged stream behind
        merged_base events; returns the chunk's merged length.  d_offsets: the rank's int64 offsets of the whole clip."""
        n = C.c_size_t(0)
        cap = 0 if d_merged is None else d_merged.numel() * d_merged.element_size() // 12
        rc = self.L.adder_gather_events_at(
            self.h, d_events.data_ptr(), d_offsets.data_ptr() + 8 * frame_begin, T, root,
            None if d_merged is None else d_merged.data_ptr(), cap, merged_base,
            None if d_merged_offsets is None else d_merged_offsets.data_ptr() + 8 * frame_begin, C.byref(n),
            C.c_void_p(stream) if stream else None)
        self.last_required = n.value
        self._check(rc)
        return n.value

    def gather_records_at(self, rec, n_records, n_events, root, d_merged, merged_base, d_merged_offsets, stream=None):
        """Records over the wire (adder_gather_records_at): rec = HipVideo.integrate_records_device's description of the
        chunk just finished; the chunk's merged events are appended behind merged_base, d_merged_offsets is the (already
        advanced) view of the merged offsets at the chunk's first frame.  Returns the chunk's merged length on root."""
        n = C.c_size_t(0)
        cap = 0 if d_merged is None else d_merged.numel() * d_merged.element_size() // 12
        rc = self.L.adder_gather_records_at(
            self.h, C.byref(rec), int(n_records), int(n_events), root,
            None if d_merged is None else d_merged.data_ptr(), cap, int(merged_base),
            None if d_merged_offsets is None else d_merged_offsets.data_ptr(), C.byref(n),
            C.c_void_p(stream) if stream else None)
        self.last_required = n.value
        self._check(rc)
        return n.value

    # ---- streamed records gather: no host wait per chunk (adder_gather_records_begin / _push / _end) ----
    def records_begin(self, root, d_merged, merged_base, d_merged_offsets, stream=None):
        cap = 0 if d_merged is None else d_merged.numel() * d_merged.element_size() // 12
        self._check(self.L.adder_gather_records_begin(
            self.h, root, None if d_merged is None else d_merged.data_ptr(), cap, int(merged_base),
            None if d_merged_offsets is None else d_merged_offsets.data_ptr(), C.c_void_p(stream) if stream else None))

    def records_begin_wire(self, root, d_wire, merged_base, d_merged_offsets, stream=None):
        """As records_begin with the raw sink's 9 / 11-byte records as root's output (d_wire: uint8 CUDA tensor on root)."""
        cap = 0 if d_wire is None else d_wire.numel() * d_wire.element_size()
        self._check(self.L.adder_gather_records_begin_wire(
            self.h, root, None if d_wire is None else d_wire.data_ptr(), cap, int(merged_base),
            None if d_merged_offsets is None else d_merged_offsets.data_ptr(), C.c_void_p(stream) if stream else None))

    def records_push(self, rec, n_records, n_events):
        self._check(self.L.adder_gather_records_push(self.h, C.byref(rec), int(n_records), int(n_events)))

    def records_end(self):
        """-> (merged events of the clip on root, else 0; bytes this rank sent)."""
        n, sent = C.c_size_t(0), C.c_uint64(0)
        rc = self.L.adder_gather_records_end(self.h, C.byref(n), C.byref(sent))
        self.last_required = n.value
        self._check(rc)
        return n.value, sent.value

    def world(self):
        """The communicator's size as the library sees it (adder_gather_world)."""
        return int(self.L.adder_gather_world(self.h))

    def records_host_us(self):
        return float(self.L.adder_gather_records_host_us(self.h))

    # ---- sink per rank (adder_gather_host_sink_*) ----
    def host_sink_open(self, image, header_bytes, stream=None):
        self._check(self.L.adder_gather_host_sink_open(self.h, image.device_ptr, image.nbytes, int(header_bytes),
                                                       C.c_void_p(stream) if stream else None))

    def host_sink_chunk(self, d_events, d_offsets, num_frames, stream=None):
        self._check(self.L.adder_gather_host_sink_chunk(self.h, d_events.data_ptr(), d_offsets.data_ptr(), int(num_frames),
                                                        C.c_void_p(stream) if stream else None))

    def host_sink_close(self, stream=None):
        tot = C.c_uint64(0)
        self._check(self.L.adder_gather_host_sink_close(self.h, C.byref(tot), C.c_void_p(stream) if stream else None))
        return tot.value

    def layout(self, d_offsets, T, stream=None):
        """-> (merged frame offsets [T+1], this rank's base per frame [T]) as numpy uint64."""
        merged = np.zeros(T + 1, np.uint64)
        base = np.zeros(max(T, 1), np.uint64)
        self._check(self.L.adder_gather_layout(self.h, d_offsets.data_ptr(), T, merged.ctypes.data, base.ctypes.data,
                                               C.c_void_p(stream) if stream else None))
        return merged, base[:T]
