"""Variance probe: the headline step timed several times in ONE process with fresh contexts/buffers each time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
W, H, T = 1920, 1080, 300
st = torch.cuda.current_stream().cuda_stream
res = []
for rep in range(int(os.environ.get("REPS", 6))):
    d_frames = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
    A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)
    d_ev = torch.empty((int(W * H * T * 0.5), 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    ts = []
    skip = 2 * int(os.environ.get("ADDER_HIP_GRAPH_CANDIDATES", 6)) + 3
    for k in range(skip + 16):
        hv.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hv.integrate_device(d_frames, d_ev, d_off, stream=st); hv.finish()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[skip:]) * 1e3
    res.append((round(float(np.median(ts)), 3), round(float(ts.min()), 3), hex(d_ev.data_ptr()), hex(d_frames.data_ptr())))
    hv.close(); del d_frames, d_ev, d_off
print(os.environ.get("ADDER_HIP_PARK_PAD", "0"), res)
