"""The per-frame ring legs of bench.py's end_to_end on their own (for sweeps: ADDER_HIP_WIRE_BLOCKS, ADDER_HIP_OUT_BLOCKS):
the headline content through the ring in AdderEvents and in wire records, and the default-quality leg."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import torch
import adder_amd as A
import bench
W, H, T = 1920, 1080, 80
st = torch.cuda.current_stream().cuda_stream
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "ring"):
    d_frames = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
    A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    r = bench.end_to_end(hv, d_frames, T, W * H, W, H, 1)
    for k in ("per_frame_ring", "per_frame_ring_wire_records"):
        print(k, json.dumps({q: r[k].get(q) for q in ("value", "us_per_frame_sustained", "mpixels_per_s", "error")}))
    hv.close()
if which in ("all", "dq"):
    print("default_quality_raw", json.dumps(bench.end_to_end_default_quality(torch, A, W, H)))
