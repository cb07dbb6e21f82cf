"""Latency of the per-frame drop-in call adder_hip_integrate (host frame in, host events out)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
for (W, H) in [(640, 480), (1920, 1080)]:
    T = 60
    d = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
    A.synth_clip_device(d, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    L = A.load()
    fp = L.adder_hip_alloc_pinned(T * W * H)
    frames = np.frombuffer((C.c_uint8 * (T * W * H)).from_address(fp), dtype=np.uint8).reshape(T, H, W)
    frames[...] = d.cpu().numpy().reshape(T, H, W)
    pageable = frames.copy()
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    cap = W * H * 2
    out = hv._host_out(cap)
    n = C.c_size_t(0)
    offs = np.zeros(hv.num_chunks + 1, np.uint32)
    for name, src in (("pinned", frames), ("pageable", pageable)):
        hv.reset()
        ts = []
        for k in range(T):
            t0 = time.perf_counter()
            rc = L.adder_hip_integrate(hv.h, src[k].ctypes.data, W, 255.0, out.ctypes.data, cap, C.byref(n), offs.ctypes.data)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        ts = np.array(ts[5:]) * 1e6
        print(json.dumps({"plane": [W, H], "frames": name, "us_per_call_median": round(float(np.median(ts)), 1),
                          "us_min": round(float(ts.min()), 1), "Mpx_per_s": round(W * H / np.median(ts), 1), "events_last": n.value}))
