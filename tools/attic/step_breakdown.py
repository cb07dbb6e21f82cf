"""Host-side wall time of the pieces of one bench step (reset / enqueue / finish) vs the GPU time of the batch."""
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
rows = []
for k in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); hv.reset()
    t1 = time.perf_counter(); hv.integrate_device(d_frames, d_ev, d_off, stream=st)
    t2 = time.perf_counter(); hv.finish()
    t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0, hv.last_batch_ms() * 1e-3))
r = np.array(rows[16:]) * 1e6
print("median us: reset %.1f enqueue %.1f finish(wait) %.1f total %.1f | gpu batch %.1f | total - gpu %.1f" % (*np.median(r, axis=0), np.median(r[:, 3] - r[:, 4])))
