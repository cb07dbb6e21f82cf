"""Does the default-quality per-frame ring leg depend on what the process ran before it?  The leg after N other contexts (the HW-queue mapping of DESIGN 5b)."""
import json, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import torch
import adder_amd as A
import bench_legs as B
W, H = 1920, 1080
print("fresh      ", B.end_to_end_default_quality(torch, A, W, H)["us_per_frame_sustained"])
print("again      ", B.end_to_end_default_quality(torch, A, W, H)["us_per_frame_sustained"])
class Args: secondary_ms = 30.0
legs = B.secondary_legs(Args, torch, A)
print("after legs ", B.end_to_end_default_quality(torch, A, W, H)["us_per_frame_sustained"])
# headline-like context kept alive
st = torch.cuda.current_stream().cuda_stream
d_frames = torch.empty((300, W * H), dtype=torch.uint8, device="cuda")
A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=300, stream=st)
d_ev = torch.empty((int(W * H * 300 * 0.75), 3), dtype=torch.int32, device="cuda")
d_off = torch.zeros(301, dtype=torch.int64, device="cuda")
hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
hv.set_crf_parameters(0, 10)
for _ in range(16):
    hv.reset(); hv.integrate_device(d_frames, d_ev, d_off, stream=st); hv.finish()
print("with a live headline context + 8 GB of buffers", B.end_to_end_default_quality(torch, A, W, H)["us_per_frame_sustained"])
r = B.end_to_end(hv, d_frames, 300, W * H, W, H, 1)
print("after end_to_end legs", B.end_to_end_default_quality(torch, A, W, H)["us_per_frame_sustained"])
hv.close()
print("... and with that context closed", B.end_to_end_default_quality(torch, A, W, H)["us_per_frame_sustained"])
print("again", B.end_to_end_default_quality(torch, A, W, H)["us_per_frame_sustained"])
