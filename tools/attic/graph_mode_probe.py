"""Is the fast / slow mode decided per graph instantiation?  One context, graphs for different batch lengths."""
import os, sys, time
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
out = []
for n in list(range(300, 288, -1)) + [300, 299]:
    ts = []
    for k in range(12):
        hv.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hv.integrate_device(d_frames[:n], d_ev, d_off[: n + 1], stream=st); hv.finish()
        ts.append(time.perf_counter() - t0)
    out.append((n, round(float(np.median(ts[3:])) * 1e3 * 300 / n, 3)))
print(out)
