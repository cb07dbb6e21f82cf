"""Where the bounded Collapse kernel's waves spend their cycles (a library built with -DADDER_CB_PROFILE=1; sets
ADDER_HIP_TIMELINE=1): general-path frames against quiet sections, the busiest waves, prologue / epilogue -- summed over the
waves of the launches of ONE batch (s_memtime ticks).  env like tools/ablate.py; SKIP frames run first (steady state)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
os.environ["ADDER_HIP_TIMELINE"] = "1"
os.environ["ADDER_HIP_NO_GRAPH"] = "1"
import numpy as np
import torch
import adder_amd as A

E = os.environ
W, H, Cn, T = int(E.get("W", 1920)), int(E.get("H", 1080)), int(E.get("C", 1)), int(E.get("T", 120))
content, tmode, dtm = int(E.get("CONTENT", 2)), int(E.get("TMODE", 1)), int(E.get("DTM", 7650))
n_units = W * H * Cn
d_frames = torch.empty((T, n_units), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
A.synth_clip_device(d_frames, content, W, H, Cn, num_frames=T, stream=st)
d_ev = torch.empty((int(n_units * T * 1.3) + 1024, 3), dtype=torch.int32, device="cuda")
d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
crf = [int(x) for x in E.get("CRF", "2,7,7").split(",")]
hv = A.HipVideo(W, H, Cn, time_mode=tmode, multi_mode=1, delta_t_max=dtm, c_thresh_start=crf[0], c_counter_start=0, max_depth=20)
hv.set_crf_parameters(crf[1], crf[2])
skip = int(E.get("SKIP", 60))
if skip:
    hv.integrate_device(d_frames[:skip], d_ev, d_off[: skip + 1], stream=st)
    hv.finish()
hv.integrate_device(d_frames[skip:], d_ev, d_off[: T - skip + 1], stream=st)
hv.finish()
buf = np.zeros(4 * 64 * 2, dtype=np.uint64)
assert hv.L.adder_hip_debug_timeline(hv.h, buf.ctypes.data_as(ctypes.c_void_p)) == 0
v = [int(buf[(3 * 64 + 32 + n) * 2 + 1]) for n in range(14)]
names = ["general-path ticks", "general frames", "quiet-section ticks", "quiet frames", "wave ticks", "waves", "slowest wave ticks",
         "most general-path ticks in a wave", "most general frames in a wave", "prologue ticks", "epilogue ticks",
         "busy waves' ticks (>= 48 general frames)", "busy waves", "busy waves' general-path ticks"]
for n, x in zip(names, v):
    print(f"{n:44s} {x}")
print("per general frame", v[0] / max(v[1], 1), "| per quiet frame", v[2] / max(v[3], 1), "| per wave", v[4] / max(v[5], 1),
      "| busy wave", v[11] / max(v[12], 1), "of it general path", v[13] / max(v[12], 1), "| kernel", A.KERNEL_NAMES[hv.last_batch_kernel()])
