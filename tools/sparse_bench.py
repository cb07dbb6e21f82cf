"""Timing helper (not a test): sparse steps of an event camera on the device (adder_hip_integrate_sparse_device).
1280x720 plane (a Prophesee Gen4 sensor), N steps per call (default 1M = what ~0.5M camera events make), steps and
events resident in HBM; prints steps/s of the call (sort + per-pixel runs + scan + emit, and the host's wait)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
from adder_amd import _native as N
E = os.environ
W, H, n = int(E.get("W", 1280)), int(E.get("H", 720)), int(E.get("N", 1 << 20))
rng = np.random.default_rng(1)
hv = A.HipVideo(W, H, 1, time_mode=A.TIME_ABSOLUTE_T, multi_mode=A.MULTI_COLLAPSE, ref_time=20, delta_t_max=40,
                pixel_mode=1, max_depth=24)
start = np.full((H, W, 1), 128, np.uint8)
for _ in range(2):
    hv.integrate_matrix(start, time_spanned=20.0)
best = 1e9
events = 0
d_out = torch.empty((n * 4, 3), dtype=torch.int32, device="cuda")
for it in range(6):
    st = np.zeros(n, N.SPARSE_STEP_DTYPE)
    hot = rng.integers(0, W * H, W * H // 20)
    pix = np.where(rng.random(n) < 0.3, hot[rng.integers(0, len(hot), n)], rng.integers(0, W * H, n))
    st["x"], st["y"], st["c"] = pix % W, pix // W, 0xFF
    val = rng.integers(0, 256, n)
    span = rng.choice(np.array([1, 1, 1, 2, 5, 40, 700]), n)
    st["frame_val"], st["intensity"], st["time"] = val, (val * span).astype(np.float32), (span * 20).astype(np.float32)
    d_st = torch.from_numpy(st.view(np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    got = C.c_size_t(0)
    t0 = time.perf_counter()
    rc = hv.L.adder_hip_integrate_sparse_device(hv.h, d_st.data_ptr(), n, d_out.data_ptr(), d_out.shape[0], C.byref(got),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    dt = time.perf_counter() - t0
    N.check(hv.h, rc)
    if it:
        best, events = min(best, dt), got.value
print(json.dumps({"plane": [W, H], "steps_per_call": n, "events_last_call": events, "ms_per_call": round(best * 1e3, 3),
                  "Msteps_per_s": round(n / best / 1e6, 1)}))
