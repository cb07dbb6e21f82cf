"""Context-to-context spread: K contexts alive at once, the same clip and event buffer, eager on one stream;
prints each context's median step time next to the addresses of its scratch (ADDER_HIP_DEBUG_ADDRS=1)."""
import os, sys, time
os.environ.setdefault("ADDER_HIP_NO_GRAPH", "1")
os.environ["ADDER_HIP_DEBUG_ADDRS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
W, H, T = 1920, 1080, 300
st = torch.cuda.current_stream().cuda_stream
K = int(os.environ.get("K", 8))
f = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
A.synth_clip_device(f, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)
e = torch.empty((int(W * H * T * 0.5), 3), dtype=torch.int32, device="cuda")
d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
pad = None
if float(os.environ.get("PAD_GB", 0)) > 0:  # something else between the caller's buffers and the contexts' scratch
    pad = torch.empty(int(float(os.environ["PAD_GB"]) * (1 << 30)), dtype=torch.uint8, device="cuda")
    if os.environ.get("PAD_FREE"):  # ... given back before the contexts allocate (torch keeps the block cached)
        del pad
        if os.environ.get("PAD_FREE") == "2":
            torch.cuda.empty_cache()
ctxs = []
for k in range(K):
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    ctxs.append(hv)
def run(hv, n=12):
    ts = []
    for k in range(n):
        hv.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hv.integrate_device(f, e, d_off, stream=st); hv.finish()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts[3:])) * 1e3, 3)
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    print("round", rnd, [run(hv) for hv in ctxs], flush=True)
