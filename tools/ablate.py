"""Timing helper (not a test): frame-loop time per frame for a mode, best of 4 batches.
env: CONTENT (0 static,1 noise,2 scene), MULTI (0 normal,1 collapse), TMODE (0 delta,1 abs), DTM, W, H, C, T,
CRF ("baseline,max,velocity", default 0,0,10), ITERS (batches, default 4: graph contexts need ~16 to settle their launch plan)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import torch
import adder_amd as A

E = os.environ
W, H, Cn, T = int(E.get("W", 1920)), int(E.get("H", 1080)), int(E.get("C", 1)), int(E.get("T", 96))
content, multi, tmode, dtm = int(E.get("CONTENT", 2)), int(E.get("MULTI", 1)), int(E.get("TMODE", 0)), int(E.get("DTM", 255))
n_units = W * H * Cn
d_frames = torch.empty((T, n_units), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
A.synth_clip_device(d_frames, content, W, H, Cn, num_frames=T, stream=st)
d_ev = torch.empty((int(n_units * T * 1.3) + 1024, 3), dtype=torch.int32, device="cuda")
d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
crf = [int(x) for x in E.get("CRF", "0,0,10").split(",")]
hv = A.HipVideo(W, H, Cn, time_mode=tmode, multi_mode=multi, delta_t_max=dtm, c_thresh_start=crf[0], c_counter_start=0, max_depth=20)
hv.set_crf_parameters(crf[1], crf[2])
best, n = 1e9, -1
for it in range(int(E.get("ITERS", 4))):
    hv.reset()
    hv.integrate_device(d_frames, d_ev, d_off, stream=st)
    n = hv.finish()
    best = min(best, hv.last_batch_ms() / T * 1000)
print(json.dumps({"plane": [W, H, Cn], "content": content, "multi": multi, "tmode": tmode, "dtm": dtm,
                  "us_per_frame": round(best, 2), "Mpx_per_s": round(W * H / best, 1), "events_per_unit_frame": round(n / (n_units * T), 4)}))
