"""Timing helper (not a test): transcode -> reframe on the device, 1080p.  Prints, per stream kind, the time of the
framer's ingest (adder_framer_ingest_frames_device: slices + tiles kernels, HIP events around the call on a warm
context) and of the whole ingest + frames_ready + pop round trip."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
E = os.environ
W, H, T = int(E.get("W", 1920)), int(E.get("H", 1080)), int(E.get("T", 64))
multi, tmode, dtm = int(E.get("MULTI", 1)), int(E.get("TMODE", 1)), int(E.get("DTM", 255))
content = {"scene": A.CONTENT_SCENE, "noise": A.CONTENT_NOISE}[E.get("CONTENT", "scene")]
n_units = W * H
st = torch.cuda.current_stream().cuda_stream
d_frames = torch.empty((T, n_units), dtype=torch.uint8, device="cuda")
A.synth_clip_device(d_frames, content, W, H, 1, num_frames=T, stream=st)
d_ev = torch.empty((int(n_units * T * 1.5) + 1024, 3), dtype=torch.int32, device="cuda")
d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
hv = A.HipVideo(W, H, 1, time_mode=tmode, multi_mode=multi, delta_t_max=dtm, c_thresh_start=0, c_counter_start=0, max_depth=20)
hv.set_crf_parameters(0, 10)
d_out = torch.empty((T + 8, n_units), dtype=torch.uint8, device="cuda")
best = best_ingest = 1e9
hv.integrate_device(d_frames, d_ev, d_off, stream=st)
n = hv.finish()
offs = d_off.cpu().numpy().astype(np.uint64)
for it in range(4):
    fr = A.HipFramer(W, H, 1, tps=255 * 30, ref_interval=255, delta_t_max=dtm, output_fps=30.0, codec_version=3,
                     time_mode=tmode, ring_frames=T + dtm // 255 + 16)
    ingest = fr.ingest_frames_device if E.get("BATCH", "1") == "1" else fr.ingest_device
    if E.get("OFFS") == "device":  # frame offsets stay in HBM (adder_framer_ingest_frames_device_offsets)
        ingest = lambda ev, o, stream: fr.ingest_frames_device_offsets(ev, d_off[int(o[0] != offs[0]):], len(o) - 1, stream=stream)
    ingest(d_ev, offs[:2], stream=st)  # first call: the source's first frame
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0.record()
    ingest(d_ev, offs[1:], stream=st)
    e1.record()
    ready = fr.frames_ready()  # (also surfaces a device-side status flag as an error)
    m = fr.pop_device(d_out, T + 8, stream=st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    best, best_ingest = min(best, dt), min(best_ingest, e0.elapsed_time(e1) * 1e-3)
    fr.close()
print(json.dumps({"plane": [W, H], "frames_in": T - 1, "events": n, "events_per_px_frame": round(n / (n_units * T), 3),
                  "frames_ready": ready, "frames_out": m,
                  "ingest_us_per_source_frame": round(best_ingest / (T - 1) * 1e6, 2),
                  "round_trip_us_per_source_frame": round(best / (T - 1) * 1e6, 2),
                  "multi": multi, "tmode": tmode, "dtm": dtm, "content": E.get("CONTENT", "scene")}))
