"""One process = one sample: GPU ms of the settled headline batch.  STREAM=default|new|ctx picks the launch stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
W, H, T = 1920, 1080, 300
mode = os.environ.get("STREAM", "default")
ts = torch.cuda.Stream() if mode == "new" else None
st = ts.cuda_stream if ts else (torch.cuda.current_stream().cuda_stream if mode == "default" else None)
d_frames = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
d_ev = torch.empty((int(W * H * T * 0.5), 3), dtype=torch.int32, device="cuda")
d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
hv.set_crf_parameters(0, 10)
g, w = [], []
for k in range(34):
    hv.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hv.integrate_device(d_frames, d_ev, d_off, stream=st); hv.finish()
    w.append(time.perf_counter() - t0); g.append(hv.last_batch_ms())
print(mode, "gpu_ms %.3f wall_ms %.3f" % (float(np.median(g[16:])), float(np.median(w[16:])) * 1e3))
