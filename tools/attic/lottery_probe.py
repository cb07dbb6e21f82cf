"""Which allocation decides a context's speed (the 'context lottery': scan + offsets + expansion take ~149 or ~165 us per
chunk for a context's life)?  Round A: one context, the OUTPUT buffer reallocated each time.  Round B: one output buffer,
the context (scratch ring, state) recreated each time.  Round C: both fixed (repeatability)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
W, H, T = 1920, 1080, 300
st = torch.cuda.current_stream().cuda_stream
d_frames = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)
d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")

def mk_ctx():
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    return hv

def mk_out():
    return torch.empty((int(W * H * T * 0.5), 3), dtype=torch.int32, device="cuda")

def measure(hv, d_ev, n=22):
    ts = []
    for k in range(n):
        hv.reset(); hv.integrate_device(d_frames, d_ev, d_off, stream=st); hv.finish()
        ts.append(hv.last_batch_ms())
    return round(float(np.median(ts[-6:])), 3)

keep = []
hv = mk_ctx()
print("A (one context, new output buffers):", end=" ")
for r in range(6):
    d_ev = mk_out(); keep.append(d_ev)
    print(measure(hv, d_ev), hex(d_ev.data_ptr()), end=" | ")
print()
d_ev = keep[0]
print("B (one output buffer, new contexts):", end=" ")
ctxs = [hv]
for r in range(6):
    h2 = mk_ctx(); ctxs.append(h2)
    print(measure(h2, d_ev), end=" | ")
print()
print("C (both fixed, repeated):", [measure(ctxs[1], d_ev, 8) for _ in range(4)], [measure(ctxs[2], d_ev, 8) for _ in range(4)])
