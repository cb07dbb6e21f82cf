"""Records over the wire (include/adder_hip.h: adder_hip_integrate_records_device / adder_hip_expand_records_device): every
band hands out its parked records instead of expanding them, root expands all of them into the merged frame-major
stream.  Here the bands are contexts of one process on one GPU (the transport -- RCCL / torch.distributed -- only moves the
described buffers); the merged stream must equal the whole-plane context's and the oracle's, byte for byte."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _hip():
    import adder_amd as A
    return A


def _bands(H, n):
    from adder_amd import sharding
    return sharding.row_bands(H, n)


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("W,H,Cn,n_bands", [(64, 48, 1, 2), (333, 41, 1, 3), (50, 24, 3, 4), (1920, 1080, 1, 8)])
def test_records_gather_equals_the_whole_plane_stream(time_mode, W, H, Cn, n_bands):
    import torch
    A = _hip()
    T = 150 if W < 1000 else 70
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, Cn, T)
    if W < 1000:  # some quiet stretches, a cut, black rows
        clip[40:90] = clip[40]
        clip[:, : H // 6] = 0
        clip[120:] = 255 - clip[120:]
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(time_mode=time_mode, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    # expected: one context over the whole plane
    whole = A.HipVideo(W, H, Cn, **kw)
    whole.set_crf_parameters(0, 10)
    d_all = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_ev = torch.empty((int(d_all.numel() * 1.3) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    whole.integrate_device(d_all, d_ev, d_off, stream=st)
    n_want = whole.finish()
    want = d_ev[:n_want].cpu().numpy().tobytes()
    want_off = d_off.cpu().numpy()
    # the bands
    bands = _bands(H, n_bands)
    ctxs, d_band_frames = [], []
    for (y0, y1) in bands:
        hv = A.HipVideo(W, H, Cn, row_begin=y0, row_end=y1, **kw)
        hv.set_crf_parameters(0, 10)
        ctxs.append(hv)
        d_band_frames.append(torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda())
    chunk = min(hv.chunk_frames() for hv in ctxs) if False else 64
    d_merged = torch.empty((n_want + 16, 3), dtype=torch.int32, device="cuda")
    d_moff = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    d_boffs = [torch.zeros(chunk + 1, dtype=torch.int64, device="cuda") for _ in bands]
    merged_base, total_records, total_events = 0, 0, 0
    for f0 in range(0, T, 37 if W < 1000 else 64):  # odd chunk lengths: the tables' rows, the logs' cursors, short tails
        nf = min(37 if W < 1000 else 64, T - f0)
        recs = []
        for r, hv in enumerate(ctxs):
            rec = hv.integrate_records_device(d_band_frames[r][f0:f0 + nf], d_boffs[r], stream=st)
            total_events += hv.finish()
            total_records += hv.last_batch_records()
            assert rec.num_frames == nf and rec.num_segments == hv.band_segments() and rec.rows == bands[r][1] - bands[r][0]
            recs.append(rec)
        root = ctxs[0]
        root.expand_records_device(recs, d_merged, merged_base, d_moff[f0:], stream=st)
        root.expand_status(stream=st)
        merged_base = int(d_moff[f0 + nf].item())
    assert merged_base == n_want == total_events and 0 < total_records < n_want
    assert np.array_equal(d_moff.cpu().numpy(), want_off)
    assert d_merged[:n_want].cpu().numpy().tobytes() == want
    if W < 1000:  # and the oracle agrees with both
        ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
        ov.ensure_capacity(8)
        ov.set_crf_parameters(0, 10)
        ov.reset_c_thresh(0)
        ora = np.concatenate([ov.integrate_matrix(f) for f in clip])
        assert ora.tobytes() == want


def test_records_are_refused_outside_the_lean_regime_and_capacity_is_reported():
    import torch
    A = _hip()
    W, H, T = 64, 32, 20
    clip = O.synth_clip(O.CONTENT_NOISE, W, H, 1, T)
    d = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=7650)  # the default mode: not lean
    d_ev = torch.empty((4 * W * H * T, 3), dtype=torch.int32, device="cuda")
    hv.integrate_device(d[:7], d_ev, d_off[:8])
    n7 = hv.finish()
    with pytest.raises(A.AdderHipError) as ei:
        hv.integrate_records_device(d, d_off)
    assert ei.value.code == A.E_BAD_PARAMS
    #  the refusal is the caller's cue to gather events instead: the context is NOT poisoned and its pixel
    # state is untouched -- the next frames continue the stream exactly as if the refused call had not happened
    hv.integrate_device(d[7:], d_ev[n7:], d_off[7:])
    n_rest = hv.finish()
    ov = O.Video(W, H, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=7650)
    want = np.concatenate([ov.integrate_matrix(f) for f in clip])
    assert n7 + n_rest == len(want) and d_ev[:n7 + n_rest].cpu().numpy().tobytes() == want.tobytes()
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    with pytest.raises(A.AdderHipError):  # more frames than a chunk
        big = torch.zeros((hv.chunk_frames() + 1, W * H), dtype=torch.uint8, device="cuda")
        hv.integrate_records_device(big, torch.zeros(hv.chunk_frames() + 2, dtype=torch.int64, device="cuda"))
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    rec = hv.integrate_records_device(d, d_off)
    n = hv.finish()
    small = torch.empty((n // 2, 3), dtype=torch.int32, device="cuda")
    moff = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.expand_records_device([rec], small, 0, moff)
    with pytest.raises(A.AdderHipError) as ei:
        hv.expand_status()
    assert ei.value.code == A.E_OUT_CAPACITY


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_records_pipelined_gather_single_rank_through_the_wire_image(time_mode):
    """RecordsPipelinedGather with one rank: every chunk goes through adder_hip_records_to_wire and back
    (records_from_wire) before root expands it on the side stream -- the transport's code path without a transport."""
    import torch
    A = _hip()
    from adder_amd.records import RecordsPipelinedGather
    W, H, T = 320, 48, 200
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, 1, T)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(time_mode=time_mode, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    d_all = torch.from_numpy(clip.reshape(T, -1)).cuda()
    whole = A.HipVideo(W, H, 1, **kw)
    whole.set_crf_parameters(0, 10)
    d_ev = torch.empty((int(d_all.numel() * 1.3) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    whole.integrate_device(d_all, d_ev, d_off, stream=st)
    n_want = whole.finish()
    hv = A.HipVideo(W, H, 1, **kw)
    hv.set_crf_parameters(0, 10)
    rg = RecordsPipelinedGather(T, hv, merged_cap_events=n_want + 8)
    d_boff = torch.zeros(65, dtype=torch.int64, device="cuda")
    for rep in range(2):  # a second clip through the same objects (reset)
        hv.reset()
        rg.reset()
        for f0 in range(0, T, 64):
            nf = min(64, T - f0)
            rec = hv.integrate_records_device(d_all[f0:f0 + nf], d_boff, stream=st)
            n = hv.finish()
            rg.push(rec, hv.last_batch_records(), n)
        merged, moff = rg.result()
        assert merged.shape[0] == n_want and torch.equal(moff, d_off)
        assert merged.cpu().numpy().tobytes() == d_ev[:n_want].cpu().numpy().tobytes()
