"""Which buffer's placement decides the fast / slow mode?  K contexts x K event buffers x K clips held at once."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
W, H, T = 1920, 1080, 300
st = torch.cuda.current_stream().cuda_stream
K = 4
frames, evs, ctxs = [], [], []
for k in range(K):
    f = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
    A.synth_clip_device(f, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)
    frames.append(f)
    evs.append(torch.empty((int(W * H * T * 0.5), 3), dtype=torch.int32, device="cuda"))
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    ctxs.append(hv)
d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
def run(hv, f, e):
    ts = []
    for k in range(14):
        hv.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hv.integrate_device(f, e, d_off, stream=st); hv.finish()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts[3:])) * 1e3, 3)
print("vary ctx   :", [run(ctxs[k], frames[0], evs[0]) for k in range(K)])
print("vary events:", [run(ctxs[0], frames[0], evs[k]) for k in range(K)])
print("vary frames:", [run(ctxs[0], frames[k], evs[0]) for k in range(K)])
print("ev ptrs", [hex(e.data_ptr()) for e in evs])
