"""Worker of test_two_ranks_hip_video_gather_shared_device (launched by torch.distributed.run, 2 ranks)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "adder-codec-rs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import adder_amd as A  # noqa: E402
from adder_amd import sharding  # noqa: E402
import clips  # noqa: E402


def run(hv, clip_band, T):
    st = torch.cuda.current_stream().cuda_stream
    d_frames = torch.from_numpy(np.ascontiguousarray(clip_band).reshape(T, -1)).cuda()
    d_ev = torch.empty((d_frames.numel() * 3 + 16, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=st)
    n = hv.finish()
    return d_ev[:n], d_off


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, H, W = 45, 90, 130
    clip = clips.make_clip("runs", T, H, W, 1, seed=33)
    kw = dict(time_mode=A.TIME_ABSOLUTE_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=255, c_thresh_start=0,
              c_counter_start=0)
    y0, y1 = sharding.row_bands(H, world)[rank]
    hv = A.HipVideo(W, H, 1, row_begin=y0, row_end=y1, **kw)
    hv.set_crf_parameters(0, 10)
    ev, off = run(hv, clip[:, y0:y1], T)
    out = sharding.gather_event_stream(ev, off, dst=0, video=hv)
    lay = sharding.exchange_stream_layout(off.cpu())
    # the same clip again with RECORDS on the wire (adder_amd.records): chunks of 16 frames, the last one short
    from adder_amd.records import RecordsPipelinedGather
    hv.reset()
    st = torch.cuda.current_stream().cuda_stream
    d_band = torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda()
    d_boff = torch.zeros(17, dtype=torch.int64, device="cuda")
    rg = RecordsPipelinedGather(T, hv, merged_cap_events=H * W * T * 3 if rank == 0 else 0, dst=0)
    sent = 0
    for f0 in range(0, T, 16):
        nf = min(16, T - f0)
        rec = hv.integrate_records_device(d_band[f0:f0 + nf], d_boff, stream=st)
        n = hv.finish()
        sent += rg.push(rec, hv.last_batch_records(), n)
    rout = rg.result()
    if rank == 0:
        hv.check_status()
        whole = A.HipVideo(W, H, 1, **kw)
        whole.set_crf_parameters(0, 10)
        wev, woff = run(whole, clip, T)
        assert torch.equal(out[1].cpu(), woff.cpu()) and torch.equal(out[0].cpu(), wev.cpu())
        assert torch.equal(lay[0], woff.cpu())
        assert torch.equal(rout[1].cpu(), woff.cpu()) and torch.equal(rout[0].cpu(), wev.cpu())  # records == events
        print("rank0 ok", flush=True)
    else:
        assert 0 < sent < 12 * int(ev.shape[0])  # fewer bytes than the band's events would have been
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
