"""Root's side of the records gather on ONE GPU: 1080p in N bands (contexts of this process), per 64-frame chunk the time of
the bands' integrate_records_device (lean kernel + scan + pack: what a peer does), of the wire images, and of root's
expansion of all bands (adder_hip_expand_records_device) -- against the whole-plane pipeline's chunk."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np, torch
import adder_amd as A
from adder_amd import sharding
from adder_amd.records import wire_bytes

W, H, T, NB = 1920, 1080, 64, int(os.environ.get("BANDS", 8))
_S = torch.cuda.Stream()  # (a real stream: the handle of torch's default stream is 0, which the C-ABI reads as "the context's own")
torch.cuda.set_stream(_S)
st = _S.cuda_stream
kw = dict(time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
bands = sharding.row_bands(H, NB)
ctxs, frames, offs = [], [], []
for (y0, y1) in bands:
    hv = A.HipVideo(W, H, 1, row_begin=y0, row_end=y1, **kw); hv.set_crf_parameters(0, 10); ctxs.append(hv)
    d = torch.empty((T, (y1 - y0) * W), dtype=torch.uint8, device="cuda")
    A.synth_clip_device(d, A.CONTENT_SCENE, W, H, 1, row_begin=y0, rows=y1 - y0, num_frames=T, stream=st)
    frames.append(d); offs.append(torch.zeros(T + 1, dtype=torch.int64, device="cuda"))
d_merged = torch.empty((int(W * H * T * 0.5), 3), dtype=torch.int32, device="cuda")
d_moff = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
res = []
for it in range(6):
    for hv in ctxs: hv.reset()
    recs, nrec, nev = [], [], []
    ev[0].record()
    for r, hv in enumerate(ctxs):
        recs.append(hv.integrate_records_device(frames[r], offs[r], stream=st)); nev.append(hv.finish()); nrec.append(hv.last_batch_records())
    ev[1].record()
    imgs = []
    for r, hv in enumerate(ctxs):
        img = torch.empty(wire_bytes(T, recs[r].num_segments, recs[r].record_bytes, nrec[r]), dtype=torch.uint8, device="cuda")
        hv.records_to_wire(recs[r], nrec[r], img, stream=st); imgs.append(img)
    ev[2].record()
    ctxs[0].expand_records_device(recs, d_merged, 0, d_moff, stream=st)
    ev[3].record()
    torch.cuda.synchronize()
    ctxs[0].expand_status(st)
    res.append((ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3, ev[2].elapsed_time(ev[3]) * 1e3))
r = np.median(np.array(res[2:]), axis=0)
print(f"{NB} bands, one 64-frame chunk of 1080p: bands' integrate_records (all {NB}, one after the other) {r[0]:.0f} us, "
      f"wire images {r[1]:.0f} us ({sum(i.numel() for i in imgs) / 1e6:.1f} MB vs {12 * sum(nev) / 1e6:.1f} MB of events), "
      f"root's expansion of all bands {r[2]:.0f} us")
