"""Host time of the streamed records gather per chunk (adder_gather_records_push) with a real one-rank RCCL communicator:
what a rank's host thread spends between two chunks besides waiting for its own batch (adder_hip_finish).
1080p headline clip, 64-frame chunks; also the per-chunk time of the blocking form (adder_gather_records_at)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
from adder_amd.gather import HipGather, unique_id
W, H, T = 1920, 1080, 300
st = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
d_frames = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)
hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
hv.set_crf_parameters(0, 10)
g = HipGather(hv, unique_id(), 0, 1)
d_m = torch.empty((int(W * H * T * 0.5), 3), dtype=torch.int32, device="cuda")
d_mo = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
d_bo = torch.zeros(65, dtype=torch.int64, device="cuda")
for mode in ("streamed", "blocking"):
    res = []
    for rep in range(6):
        hv.reset()
        torch.cuda.synchronize()
        t_fin = t_push = 0.0
        t0 = time.perf_counter()
        if mode == "streamed":
            g.records_begin(0, d_m, 0, d_mo, stream=side.cuda_stream)
        pos = 0
        for f0 in range(0, T, 64):
            nf = min(64, T - f0)
            rec = hv.integrate_records_device(d_frames[f0:f0 + nf], d_bo, stream=st)
            t1 = time.perf_counter()
            n_k = hv.finish()
            t2 = time.perf_counter()
            if mode == "streamed":
                g.records_push(rec, hv.last_batch_records(), n_k)
            else:
                side.wait_stream(torch.cuda.current_stream())
                pos += g.gather_records_at(rec, hv.last_batch_records(), n_k, 0, d_m, pos, d_mo[f0:], stream=side.cuda_stream)
            t3 = time.perf_counter()
            t_fin += t2 - t1
            t_push += t3 - t2
        if mode == "streamed":
            n, _ = g.records_end()
        else:
            side.synchronize(); n = pos
        el = time.perf_counter() - t0
        res.append((el * 1e3, t_fin * 1e3 / 5, t_push * 1e6 / 5, g.records_host_us() / 5 if mode == "streamed" else 0.0))
    r = np.array(res[2:])
    print(f"{mode}: step {r[:,0].mean():.3f} ms; per chunk: finish wait {r[:,1].mean()*1e3:.0f} us, gather call {r[:,2].mean():.0f} us "
          f"(inside the library {r[:,3].mean():.0f} us); events {n}")
