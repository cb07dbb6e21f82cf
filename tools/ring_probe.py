"""Per-frame ring probe (adder_hip_frame_submit / _collect): every submit's and collect's host time, the sustained rate.
DQ=1: the reference's default mode at its default quality with wire records out (bench.py's default_quality_raw leg); T frames."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
W, H, T = 1920, 1080, int(os.environ.get("T", 100))
DQ = os.environ.get("DQ") == "1"
SLOTS = int(os.environ.get("SLOTS", 3))
d = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
A.synth_clip_device(d, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
L = A.load()
fp = L.adder_hip_alloc_pinned(T * W * H)
frames = np.frombuffer((C.c_uint8 * (T * W * H)).from_address(fp), dtype=np.uint8).reshape(T, H, W)
frames[...] = d.cpu().numpy().reshape(T, H, W)
if DQ:
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_ABSOLUTE_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=7650, c_thresh_start=2, c_counter_start=0)
    hv.set_crf_parameters(7, 7)
    hv.frames_set_format(True)
else:
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
if SLOTS != 3:
    assert L.adder_hip_frames_configure(hv.h, SLOTS, 0) == 0
ev_p, n_p, ch_p, nb_p = C.c_void_p(), C.c_size_t(0), C.c_void_p(), C.c_size_t(0)
def collect():
    if DQ:
        rc = L.adder_hip_frame_collect_wire(hv.h, C.byref(ev_p), C.byref(nb_p), C.byref(n_p), C.byref(ch_p))
    else:
        rc = L.adder_hip_frame_collect(hv.h, C.byref(ev_p), C.byref(n_p), C.byref(ch_p))
    assert rc == 0, rc
def submit(k):
    rc = L.adder_hip_frame_submit(hv.h, frames[k].ctypes.data, W, 255.0); assert rc == 0, rc
for rnd in range(3):
    hv.reset()
    sub, col = [], []
    t0 = time.perf_counter()
    for k in range(T):
        if L.adder_hip_frames_in_flight(hv.h) == SLOTS:
            t1 = time.perf_counter(); collect(); col.append(time.perf_counter() - t1)
        t1 = time.perf_counter(); submit(k); sub.append(time.perf_counter() - t1)
    while L.adder_hip_frames_in_flight(hv.h):
        collect()
    el = time.perf_counter() - t0
    sub = np.array(sub[T // 2:]) * 1e6; col = np.array(col[T // 2:]) * 1e6   # (the second half: past the set-up and the ring's own tuning)
    print(json.dumps({"round": rnd, "us_per_frame": round(el / T * 1e6, 1), "submit_median": round(float(np.median(sub)), 1),
                      "submit_max": round(float(sub.max()), 1), "submit_argmax": int(sub.argmax()),
                      "submit_over_200us": [int(i) for i in np.nonzero(sub > 200)[0][:10]],
                      "collect_median": round(float(np.median(col)), 1), "collect_max": round(float(col.max()), 1), "events_last": n_p.value}))
if DQ:
    sys.exit(0)
# blocking call for comparison
cap = hv.max_events_per_frame
out = hv._host_out(cap); n = C.c_size_t(0); offs = np.zeros(hv.num_chunks + 1, np.uint32)
hv.reset()
ts = []
for k in range(T):
    t0 = time.perf_counter()
    assert L.adder_hip_integrate(hv.h, frames[k].ctypes.data, W, 255.0, out.ctypes.data, cap, C.byref(n), offs.ctypes.data) == 0
    ts.append(time.perf_counter() - t0)
ts = np.array(ts[5:]) * 1e6
print(json.dumps({"blocking_call_median_us": round(float(np.median(ts)), 1), "min": round(float(ts.min()), 1), "max": round(float(ts.max()), 1)}))
