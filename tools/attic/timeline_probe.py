"""Kernel timelines (in-kernel timestamps, no profiler) of a fast and a slow graph instance of the headline batch."""
import os, sys, time
os.environ["ADDER_HIP_TIMELINE"] = "1"
os.environ.setdefault("ADDER_HIP_GRAPH_CANDIDATES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
W, H, T = 1920, 1080, 300
st = torch.cuda.current_stream().cuda_stream
d_frames = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)
d_ev = torch.empty((int(W * H * T * 0.5), 3), dtype=torch.int32, device="cuda")
d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
hv.set_crf_parameters(0, 10)
res = []
for n in range(300, 288, -1):
    g = []
    for k in range(7):
        hv.reset()
        hv.integrate_device(d_frames[:n], d_ev, d_off[: n + 1], stream=st); hv.finish()
        g.append(hv.last_batch_ms())
    tl = np.zeros(4 * 64 * 2, np.uint64)
    assert hv.L.adder_hip_debug_timeline(hv.h, tl.ctypes.data) == 0
    res.append((n, float(np.median(g[2:])), tl.reshape(4, 64, 2).astype(np.int64)))
res.sort(key=lambda r: r[1])
for label, (n, ms, tl) in (("FASTEST", res[0]), ("SLOWEST", res[-1])):
    print(f"{label}: {n} frames, {ms:.3f} ms  (all: {[round(r[1], 3) for r in res]})")
    t0 = tl[0, 0, 0]
    nchunks = (n + 31) // 32
    for c in range(nchunks):
        row = []
        for kind, name in enumerate(("frame", "scan", "offs", "expand")):
            s_, e_ = (tl[kind, c] - t0) / 100.0
            row.append(f"{name} {s_:7.1f}-{e_:7.1f}")
        print(f"  chunk {c}: " + " | ".join(row))
