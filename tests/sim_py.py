"""ctypes wrapper of tests/cpu_sim (g++ build of the device header) -- test helper."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SRC = os.path.join(_HERE, "cpu_sim", "sim.cpp")
_HDR = os.path.join(_ROOT, "adder-codec-rs_amd", "csrc", "adder_pixel.hpp")
_LIB = os.path.join(_HERE, "cpu_sim", "libadder_sim.so")

EVENT_DTYPE = np.dtype(
    [("x", "<u2"), ("y", "<u2"), ("c", "u1"), ("d", "u1"), ("pad", "<u2"), ("t", "<u4")]
)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    newest = max(os.path.getmtime(_SRC), os.path.getmtime(_HDR),
                 os.path.getmtime(os.path.join(os.path.dirname(_HDR), "adder_framer.hpp")))
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < newest:
        subprocess.check_call([
            "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wextra",
            "-I", os.path.dirname(_HDR), _SRC, "-o", _LIB])
    L = C.CDLL(_LIB)
    vp, u8, u32, f32, i32, sz = C.c_void_p, C.c_uint8, C.c_uint32, C.c_float, C.c_int, C.c_size_t
    L.sim_new.restype = vp
    L.sim_new.argtypes = [u32, u32, u32, u32, i32, i32, u32, u32, u32]
    L.sim_free.argtypes = [vp]
    L.sim_set_crf_parameters.argtypes = [vp, u8, u8]
    L.sim_set_continuous.argtypes = [vp]
    L.sim_fast9_plane.argtypes = [vp, u32, u32, u32, vp]
    L.sim_update_detect_features.argtypes = [vp, i32, i32, u32, u32, u32]
    L.sim_update_roi.argtypes = [vp, i32, u32, u32, u32, u32, u32]
    L.sim_feature_set.argtypes = [vp, vp]
    L.sim_c_thresh_plane.argtypes = [vp, vp]
    L.sim_new_features.restype = u32
    L.sim_new_features.argtypes = [vp]
    L.sim_reset_c_thresh.argtypes = [vp, u8]
    L.sim_set_delta_t_max.argtypes = [vp, u32]
    L.sim_plan_mismatches.restype = C.c_uint64
    L.sim_plan_mismatches.argtypes = [vp]
    L.sim_max_m.restype = u32
    L.sim_max_m.argtypes = [vp]
    L.sim_set_use_fast.argtypes = [vp, i32]
    L.sim_fast_steps.restype = C.c_uint64
    L.sim_fast_steps.argtypes = [vp]
    L.sim_generic_steps.restype = C.c_uint64
    L.sim_generic_steps.argtypes = [vp]
    L.sim_lean_steps.restype = C.c_uint64
    L.sim_lean_steps.argtypes = [vp]
    L.sim_cb_steps.restype = C.c_uint64
    L.sim_cb_steps.argtypes = [vp]
    L.sim_cb_quiet_steps.restype = C.c_uint64
    L.sim_cb_quiet_steps.argtypes = [vp]
    L.sim_set_cb_quiet_path.argtypes = [vp, C.c_int]
    L.sim_set_quiet_group_path.argtypes = [vp, C.c_int]
    for f in (L.sim_quiet_groups, L.sim_quiet_group_fires, L.sim_quiet_group_slow):
        f.restype = C.c_uint64
        f.argtypes = [vp]
    L.sim_lean_quiet_steps.restype = C.c_uint64
    L.sim_lean_quiet_steps.argtypes = [vp]
    L.sim_set_use_cb.argtypes = [vp, i32]
    L.sim_integrate_cb_block.restype = i32
    L.sim_integrate_cb_block.argtypes = [vp, vp, u32, f32, vp, sz, C.POINTER(sz)]
    L.sim_integrate_lr_block.restype = i32
    L.sim_integrate_lr_block.argtypes = [vp, vp, u32, f32, vp, sz, C.POINTER(sz)]
    L.sim_lean_group_check.restype = C.c_uint64
    L.sim_lean_group_check.argtypes = [f32, u32, u32, C.POINTER(C.c_uint64)]
    L.sim_integrate_lp_block.restype = i32
    L.sim_integrate_lp_block.argtypes = [vp, vp, u32, f32, vp, sz, C.POINTER(sz)]
    L.sim_integrate_rr_block.restype = i32
    L.sim_integrate_rr_block.argtypes = [vp, vp, u32, f32, vp, sz, C.POINTER(sz)]
    L.sim_integrate_cr_block.restype = i32
    L.sim_integrate_cr_block.argtypes = [vp, vp, u32, f32, vp, sz, C.POINTER(sz)]
    L.sim_integrate.restype = i32
    L.sim_integrate.argtypes = [vp, vp, f32, vp, sz, C.POINTER(sz)]
    L.sim_framer_run.restype = C.c_int64
    L.sim_framer_run.argtypes = [vp, sz, u32, u32, u32, u32, u32, u32, u32, vp, sz]
    _lib = L
    return L


def lean_group_check(T, groups=20000, seed=1):
    """(groups whose closed form differs from the stepped form, groups the closed form took) at time step T."""
    applied = C.c_uint64(0)
    bad = lib().sim_lean_group_check(T, groups, seed, C.byref(applied))
    return int(bad), int(applied.value)


def fast9_plane(img):
    """fast9_is_feature (adder_pixel.hpp) at every pixel of img = [h][w] or [h][w][c] u8."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.zeros((h, w), np.uint8)
    lib().sim_fast9_plane(img.ctypes.data, w, h, ch, out.ctypes.data)
    return out


def framer_run(events, width, height, channels, *, tpf, ref_interval, abs_t, round_up, max_frames=4096, view_mode=0,
               source_type=0, practical_d_max=0.0, delta_t_max=0, value_type=0):
    """All events through the device header's framer_step on the host -> bytes of the complete frames."""
    ev = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
    lib().sim_framer_set_view.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_uint32]
    lib().sim_framer_set_view(view_mode, source_type, practical_d_max, delta_t_max)
    lib().sim_framer_set_value_type.argtypes = [C.c_uint32]
    lib().sim_framer_set_value_type(value_type)
    out = np.zeros((max_frames * width * height * channels) << value_type, np.uint8)
    n = lib().sim_framer_run(ev.ctypes.data, len(ev), width, height, channels, tpf, ref_interval, int(abs_t),
                             int(round_up), out.ctypes.data, max_frames)
    if n < 0:
        raise RuntimeError(f"sim_framer_run failed: {n}")
    return out[: (n * width * height * channels) << value_type].tobytes()


class Sim:
    def __init__(self, width, height, channels=1, *, row_begin=0, time_mode=1, multi_mode=1,
                 ref_time=255, delta_t_max=7650, max_depth=16):
        self.L = lib()
        self.n = width * height * channels
        self.h = self.L.sim_new(width, height, channels, row_begin, time_mode, multi_mode, ref_time,
                                delta_t_max, max_depth)
        self._cap = self.n * (max_depth + 2)
        self._out = np.zeros(self._cap, EVENT_DTYPE)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sim_free(self.h)
            self.h = None

    def set_crf_parameters(self, c_max, velocity):
        self.L.sim_set_crf_parameters(self.h, c_max, velocity)

    def set_continuous(self):
        self.L.sim_set_continuous(self.h)

    def reset_c_thresh(self, baseline):
        self.L.sim_reset_c_thresh(self.h, baseline)

    def set_delta_t_max(self, dtm):
        self.L.sim_set_delta_t_max(self.h, dtm)

    def integrate_sparse(self, steps):
        """steps: array with the oracle's SPARSE_STEP_DTYPE layout; returns (rc, events)."""
        steps = np.ascontiguousarray(steps)
        cap = max(1024, len(steps) * 24)
        out = np.zeros(cap, EVENT_DTYPE)
        n = C.c_size_t(0)
        self.L.sim_integrate_sparse.restype = C.c_int
        self.L.sim_integrate_sparse.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                C.POINTER(C.c_size_t)]
        rc = self.L.sim_integrate_sparse(self.h, steps.ctypes.data, len(steps), out.ctypes.data, cap, C.byref(n))
        return rc, out[: n.value].copy()

    def update_detect_features(self, detect, adjust, baseline, radius, chunk_rows=1):
        self.L.sim_update_detect_features(self.h, int(detect), int(adjust), baseline, radius, chunk_rows)

    def update_roi(self, roi, baseline):
        if roi is None:
            self.L.sim_update_roi(self.h, 0, 0, 0, 0, 0, baseline)
        else:
            self.L.sim_update_roi(self.h, 1, *[int(v) for v in roi], baseline)

    def feature_set(self, width, height):
        out = np.zeros((height, width), np.uint8)
        self.L.sim_feature_set(self.h, out.ctypes.data)
        return out

    def c_thresh_plane(self):
        out = np.zeros(self.n, np.uint8)
        self.L.sim_c_thresh_plane(self.h, out.ctypes.data)
        return out

    @property
    def new_features(self):
        return self.L.sim_new_features(self.h)

    def set_use_fast(self, on):
        self.L.sim_set_use_fast(self.h, int(on))

    @property
    def fast_steps(self):
        return self.L.sim_fast_steps(self.h)

    @property
    def generic_steps(self):
        return self.L.sim_generic_steps(self.h)

    @property
    def lean_steps(self):
        return self.L.sim_lean_steps(self.h)

    def set_use_cb(self, on):
        self.L.sim_set_use_cb(self.h, int(on))

    @property
    def cb_steps(self):
        return self.L.sim_cb_steps(self.h)

    @property
    def cb_quiet_steps(self):
        return self.L.sim_cb_quiet_steps(self.h)

    @property
    def lean_quiet_steps(self):
        return self.L.sim_lean_quiet_steps(self.h)

    def set_cb_quiet_path(self, on):
        self.L.sim_set_cb_quiet_path(self.h, int(on))

    def set_quiet_group_path(self, on):
        self.L.sim_set_quiet_group_path(self.h, int(on))

    @property
    def quiet_groups(self):
        """(groups applied in closed form, of those with a firing inside, groups a unit stepped itself)"""
        return (self.L.sim_quiet_groups(self.h), self.L.sim_quiet_group_fires(self.h), self.L.sim_quiet_group_slow(self.h))

    def integrate_cb_block(self, frames, time_spanned):
        """nb frames as ONE temporally blocked launch of the bounded Collapse step; (rc, events frame-major)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(len(frames), -1)
        assert frames.shape[1] == self.n
        cap = self._cap * len(frames)
        out = np.zeros(cap, EVENT_DTYPE)
        n = C.c_size_t(0)
        rc = self.L.sim_integrate_cb_block(self.h, frames.ctypes.data, len(frames), time_spanned, out.ctypes.data, cap,
                                           C.byref(n))
        return rc, out[: n.value].copy()

    def integrate_lr_block(self, frames, time_spanned):
        """nb frames as ONE launch of the lean-runs step (lean regime, DeltaT, c_thresh 0 throughout)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(len(frames), -1)
        assert frames.shape[1] == self.n
        cap = self._cap * len(frames)
        out = np.zeros(cap, EVENT_DTYPE)
        n = C.c_size_t(0)
        rc = self.L.sim_integrate_lr_block(self.h, frames.ctypes.data, len(frames), time_spanned, out.ctypes.data, cap,
                                           C.byref(n))
        return rc, out[: n.value].copy()

    def integrate_lp_block(self, frames, time_spanned):
        """nb frames as ONE launch of the PACKED lean-runs step (four units per word; DeltaT, c_thresh 0 throughout)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(len(frames), -1)
        assert frames.shape[1] == self.n
        cap = self._cap * len(frames)
        out = np.zeros(cap, EVENT_DTYPE)
        n = C.c_size_t(0)
        rc = self.L.sim_integrate_lp_block(self.h, frames.ctypes.data, len(frames), time_spanned, out.ctypes.data, cap,
                                           C.byref(n))
        return rc, out[: n.value].copy()

    def integrate_cr_block(self, frames, time_spanned):
        """nb frames as ONE launch of the constant-run step (c_thresh 0 throughout); (rc, events frame-major)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(len(frames), -1)
        assert frames.shape[1] == self.n
        cap = self._cap * len(frames)
        out = np.zeros(cap, EVENT_DTYPE)
        n = C.c_size_t(0)
        rc = self.L.sim_integrate_cr_block(self.h, frames.ctypes.data, len(frames), time_spanned, out.ctypes.data, cap,
                                           C.byref(n))
        return rc, out[: n.value].copy()

    def integrate_rr_block(self, frames, time_spanned):
        """nb frames as ONE launch of the run-records step (integer state; events worked out from the records)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(len(frames), -1)
        assert frames.shape[1] == self.n
        cap = self._cap * len(frames)
        out = np.zeros(cap, EVENT_DTYPE)
        n = C.c_size_t(0)
        rc = self.L.sim_integrate_rr_block(self.h, frames.ctypes.data, len(frames), time_spanned, out.ctypes.data, cap,
                                           C.byref(n))
        return rc, out[: n.value].copy()

    @property
    def plan_mismatches(self):
        return self.L.sim_plan_mismatches(self.h)

    @property
    def max_m(self):
        return self.L.sim_max_m(self.h)

    def integrate(self, frame, time_spanned):
        frame = np.ascontiguousarray(frame, dtype=np.uint8).reshape(-1)
        assert frame.size == self.n
        n = C.c_size_t(0)
        rc = self.L.sim_integrate(self.h, frame.ctypes.data, time_spanned, self._out.ctypes.data,
                                  self._cap, C.byref(n))
        return rc, self._out[: n.value].copy()
