"""GPU parity proper: the HIP path (through the C-ABI, libadder_hip.so) against the CPU
oracle on the same seeded inputs -- bit-exact on every field of every event, in order.
"""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
import clips

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CRFS = {0: (0, 0, 10), 3: (2, 7, 7), 6: (7, 13, 4), 9: (15, 25, 1)}


def _hip():
    import adder_amd
    return adder_amd


def run_pair(clip, *, time_mode, multi_mode, dtm, ref_time=255, crf=None, default_pixels=False,
             max_depth=20, batch=False, row_band=None, chunk_rows=1):
    A = _hip()
    frames, H, W, Cn = clip.shape
    y0, y1 = (0, H) if row_band is None else row_band
    sub = clip[:, y0:y1]
    ov = O.Video(W, y1 - y0, Cn, row_begin=y0, time_mode=time_mode, multi_mode=multi_mode, ref_time=ref_time,
                 delta_t_max=dtm, chunk_rows=chunk_rows)
    hv = A.HipVideo(W, H, Cn, row_begin=y0, row_end=y1, time_mode=time_mode, multi_mode=multi_mode,
                    ref_time=ref_time, delta_t_max=dtm, max_depth=max_depth, chunk_rows=chunk_rows)
    ov.ensure_capacity(max_depth + 2)
    if crf is not None:
        base, cmax, vel = crf
        ov.set_crf_parameters(cmax, vel)
        hv.set_crf_parameters(cmax, vel)
        if not default_pixels:
            ov.reset_c_thresh(base)
            hv.reset_c_thresh(base)
    total = 0
    if batch:
        want = [ov.integrate_matrix(sub[k], time_spanned=float(ref_time)) for k in range(frames)]
        got, offs = hv.integrate_batch(sub)
        assert [int(offs[k + 1] - offs[k]) for k in range(frames)] == [len(w) for w in want]
        want = np.concatenate(want)
        assert np.array_equal(got, want)
        total = len(want)
    else:
        for k in range(frames):
            a, ca = ov.integrate_matrix(sub[k], time_spanned=float(ref_time), want_chunks=True)
            b, cb = hv.integrate_matrix(sub[k], want_chunks=True)
            assert len(a) == len(b), (k, len(a), len(b))
            assert np.array_equal(a, b), k
            assert np.array_equal(ca, cb), k
            total += len(a)
    hv.close()
    return total


@pytest.mark.parametrize("kind", ["noise", "static", "dark", "jitter", "runs", "steps"])
@pytest.mark.parametrize("multi_mode", [O.NORMAL, O.COLLAPSE])
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_modes_crf0(kind, multi_mode, time_mode):
    clip = clips.make_clip(kind, 40, 37, 53, 1, seed=17 + multi_mode * 2 + time_mode)
    for dtm in (255, 7650):
        run_pair(clip, time_mode=time_mode, multi_mode=multi_mode, dtm=dtm, crf=CRFS[0])


@pytest.mark.parametrize("crf", [3, 9])
@pytest.mark.parametrize("multi_mode", [O.NORMAL, O.COLLAPSE])
def test_lossy_crf(crf, multi_mode):
    for kind in ("jitter", "runs", "dark"):
        clip = clips.make_clip(kind, 60, 33, 41, 1, seed=crf * 100 + multi_mode)
        run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=multi_mode, dtm=7650, crf=CRFS[crf], batch=True)
        run_pair(clip, time_mode=O.DELTA_T, multi_mode=multi_mode, dtm=255, crf=CRFS[crf], batch=True)


def test_construction_default_pixels():
    clip = clips.make_clip("jitter", 50, 20, 30, 1, seed=5)
    run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, dtm=7650, crf=(2, 7, 7), default_pixels=True)
    run_pair(clip, time_mode=O.DELTA_T, multi_mode=O.NORMAL, dtm=510, crf=None)


def test_rgb_interleaved():
    clip = clips.make_clip("runs", 40, 19, 23, 3, seed=11)
    for mm in (O.NORMAL, O.COLLAPSE):
        for tm in (O.DELTA_T, O.ABSOLUTE_T):
            run_pair(clip, time_mode=tm, multi_mode=mm, dtm=1020, crf=CRFS[0], batch=True)


@pytest.mark.parametrize("ref_time,dtm", [(5000, 240000), (1000, 2000), (20, 10000)])
def test_other_tick_rates(ref_time, dtm):
    clip = clips.make_clip("runs", 50, 16, 24, 1, seed=ref_time)
    for mm in (O.NORMAL, O.COLLAPSE):
        run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=mm, dtm=dtm, ref_time=ref_time, crf=CRFS[0], batch=True)
        run_pair(clip, time_mode=O.DELTA_T, multi_mode=mm, dtm=dtm, ref_time=ref_time, crf=CRFS[3], batch=True)


def test_ragged_tiny_and_odd_planes():
    # 1x1, single row, widths that are not multiples of the lane/tile size
    for (H, W, Cn) in [(1, 1, 1), (1, 1, 3), (1, 7, 1), (3, 5, 3), (2, 1025, 1), (5, 341, 3), (1, 4097, 1)]:
        clip = clips.make_clip("runs", 24, H, W, Cn, seed=H * 1000 + W)
        run_pair(clip, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=255, crf=CRFS[0])
        run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=O.NORMAL, dtm=510, crf=CRFS[0], batch=True)


def test_many_segments_full_plane_noise():
    # a full 1080p plane of noise: every one of the 16 200 wave segments parks a record per unit (the
    # expansion's more-than-32-records path), and more workgroups than the chip holds at once
    clip = clips.make_clip("noise", 6, 1080, 1920, 1, seed=3)
    run_pair(clip, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=255, crf=CRFS[0], batch=True)
    clip = clips.make_clip("runs", 12, 540, 960, 3, seed=4)
    run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=O.NORMAL, dtm=1020, crf=CRFS[0], batch=True)


def test_row_band_and_chunks():
    clip = clips.make_clip("runs", 30, 64, 48, 1, seed=8)
    run_pair(clip, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=255, crf=CRFS[0], row_band=(16, 40), chunk_rows=5)
    run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=O.NORMAL, dtm=510, crf=CRFS[0], row_band=(40, 64), chunk_rows=64)


def test_long_static_deep_arena():
    clip = clips.make_clip("static", 400, 8, 16, 1, seed=2)
    clip[-1] = 255 - clip[-1]
    run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=O.NORMAL, dtm=255, crf=CRFS[0], batch=True)
    run_pair(clip, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=7650, crf=CRFS[0], batch=True)


def test_long_runs_large_t_lean_records():
    """Collapse / delta_t_max = ref_time (the lean kernel): pixels that stay put for > 514 frames and then
    change emit t >= 2^17 (round 1's compact records needed an overflow list there; the 16-byte records
    carry the full best_delta_t).  Several such pixels per segment, different run lengths, and D_EMPTY
    events (their t is the frame's running_t, not stored) in the same frames."""
    rng = np.random.default_rng(5)
    T, H, W = 760, 6, 70
    clip = np.broadcast_to(rng.integers(1, 256, (1, H, W, 1), dtype=np.uint8), (T, H, W, 1)).copy()
    for (y, x, k) in [(0, 0, 600), (0, 1, 650), (0, 2, 700), (3, 40, 700), (3, 41, 701), (5, 69, 759), (2, 10, 530)]:
        clip[k:, y, x, 0] = 255 - clip[0, y, x, 0]
    clip[300:, 4, :, 0] = rng.integers(0, 256, (T - 300, W), dtype=np.uint8)  # a busy row next to them
    # six neighbours (same lane pairs and different lanes of one segment) that all overflow in one frame
    clip[:, 0, 1:7, 0] = clip[:650, 0, 1:2, 0][:1]
    clip[650:, 0, 1:7, 0] = 255 - clip[0, 0, 1, 0]
    ov = O.Video(W, H, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, delta_t_max=255)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    wide = [int(((e["d"] != 255) & (e["t"] >= 0x1FFFF)).sum()) for e in (ov.integrate_matrix(f) for f in clip)]
    assert max(wide) >= 6 and sum(w > 0 for w in wide) >= 3  # the case is really exercised
    run_pair(clip, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=255, crf=CRFS[0], batch=True)
    run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, dtm=255, crf=CRFS[0], batch=True)
    run_pair(clip[:, :, :, :], time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, dtm=255, crf=CRFS[3], batch=True)


def test_lake_golden_bytes(golden_dir):
    """Reference golden: frames -> HIP path -> raw sink == lake_scaled_hd_out.adder byte for byte."""
    A = _hip()
    raw = gzip.open(os.path.join(golden_dir, "lake_scaled_hd_out.adder.gz")).read()
    frames = np.load(os.path.join(golden_dir, "lake_scaled_hd_frames_reconstructed.npz"))["frames"]
    hv = A.HipVideo(200, 50, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_NORMAL, ref_time=255, delta_t_max=6120)
    hv.update_crf(0)
    ev, offs = hv.integrate_batch(frames[:, :, :, None])
    assert len(ev) == 201_620 and offs[1] == 0
    blob = A.raw_header(3, 200, 50, 1, 6113, 255, 6120, 0, A.TIME_DELTA_T, 0) + A.raw_events(ev, 1) + A.raw_eof()
    assert hashlib.sha256(blob).hexdigest() == "b3ceb84fbef8c6f0f054c521b3d66fdda397219befc8201d6e3e1dd64cb80967"
    assert blob == raw


def test_second_opinion_vectors(golden_dir):
    A = _hip()
    d = json.load(open(os.path.join(golden_dir, "model_second_opinion_vectors.json")))
    for case in d["cases"]:
        W, H, Cn = case["width"], case["height"], case["channels"]
        hv = A.HipVideo(W, H, Cn, time_mode=case["time_mode"], multi_mode=case["multi_mode"],
                        ref_time=case["ref_time"], delta_t_max=case["delta_t_max"])
        hv.set_crf_parameters(case["c_thresh_max"], case["c_increase_velocity"])
        if case["c_start"] is not None:
            hv.reset_c_thresh(case["c_start"])
        frames = np.array(case["input_hwc_u8"], dtype=np.uint8).reshape(case["frames"], H, W, Cn)
        got, offs = hv.integrate_batch(frames)
        assert [int(offs[k + 1] - offs[k]) for k in range(case["frames"])] == case["events_per_frame"]
        want = np.array(case["events"], dtype=np.int64).reshape(-1, 5)
        gc = got["c"].astype(np.int64)
        gc[gc == 0xFF] = -1
        have = np.stack([got["x"].astype(np.int64), got["y"].astype(np.int64), gc,
                         got["d"].astype(np.int64), got["t"].astype(np.int64)], axis=1)
        assert np.array_equal(have, want), case["name"]


def test_running_intensities_side_plane():
    A = _hip()
    clip = clips.make_clip("runs", 30, 12, 20, 1, seed=21)
    ov = O.Video(20, 12, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=7650)
    hv = A.HipVideo(20, 12, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=7650)
    hv.enable_running_intensities(True)
    for v in (ov, hv):
        v.set_crf_parameters(0, 10)
        v.reset_c_thresh(0)
    for k in range(30):
        ov.integrate_matrix(clip[k])
        hv.integrate_matrix(clip[k])
        assert np.array_equal(ov.running_intensities(), hv.running_intensities()), k


def test_errors_are_loud():
    A = _hip()
    with pytest.raises(A.AdderHipError):
        A.HipVideo(16, 16, 2)  # channels must be 1 or 3
    with pytest.raises(A.AdderHipError):
        A.HipVideo(16, 16, 1, delta_t_max=100)  # dtm < ref_time
    # capacity overflow is reported with the required size and is RECOVERABLE: the pixel state is rolled back,
    # the caller retries with a larger buffer (test_capacity_overflow_rolls_back_and_retry_matches)
    clip = clips.make_clip("noise", 3, 16, 16, 1, seed=1)
    hv = A.HipVideo(16, 16, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255)
    hv.update_crf(0)
    hv.integrate_matrix(clip[0])
    with pytest.raises(A.AdderHipError) as ei:
        hv.integrate_matrix(clip[1], out_cap=10)
    assert ei.value.code == -4 and hv.last_required > 10
    hv.integrate_matrix(clip[1])
    # arena depth overflow
    clip = clips.make_clip("static", 100, 4, 4, 1, seed=1)
    hv = A.HipVideo(4, 4, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_NORMAL, delta_t_max=255, max_depth=3)
    hv.update_crf(0)
    with pytest.raises(A.AdderHipError) as ei:
        hv.integrate_batch(clip)
    assert ei.value.code == -5


@pytest.mark.parametrize("multi_mode,dtm", [(O.COLLAPSE, 255), (O.COLLAPSE, 7650), (O.NORMAL, 1020)])
def test_capacity_overflow_rolls_back_and_retry_matches(multi_mode, dtm):
    """An event buffer that is too small does not kill the stream: the call fails with ADDER_E_OUT_CAPACITY and
    the size needed, the pixel state (all levels, c_thresh, running_t) is what it was before the call, and the
    retry produces exactly the oracle's events -- per-frame calls and batches, lean and generic kernels."""
    A = _hip()
    clip = clips.make_clip("runs", 40, 20, 33, 1, seed=7)
    clip[25] = 255 - clip[24]  # a scene cut: every pixel flushes its whole arena (> 2 events per pixel)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov = O.Video(33, 20, 1, time_mode=tm, multi_mode=multi_mode, ref_time=255, delta_t_max=dtm)
        hv = A.HipVideo(33, 20, 1, time_mode=tm, multi_mode=multi_mode, ref_time=255, delta_t_max=dtm)
        ov.ensure_capacity(22)
        for v in (ov, hv):
            v.set_crf_parameters(3, 4)
            v.reset_c_thresh(1)
        want = [ov.integrate_matrix(f) for f in clip]
        got, offs = hv.integrate_batch(clip[:20])
        assert np.array_equal(got, np.concatenate(want[:20]))
        for k in range(20, 30):  # per-frame calls with a hopeless buffer first
            with pytest.raises(A.AdderHipError) as ei:
                hv.integrate_matrix(clip[k], out_cap=max(len(want[k]) // 2, 1) if len(want[k]) > 1 else 0)
            assert ei.value.code == A.E_OUT_CAPACITY and hv.last_required == len(want[k]), k
            assert np.array_equal(hv.integrate_matrix(clip[k], out_cap=hv.last_required), want[k]), k
        need = sum(len(w) for w in want[30:])
        with pytest.raises(A.AdderHipError) as ei:  # a batch
            hv.integrate_batch(clip[30:], out_cap=need - 1)
        assert ei.value.code == A.E_OUT_CAPACITY and hv.last_required == need
        got, offs = hv.integrate_batch(clip[30:], out_cap=need)
        assert np.array_equal(got, np.concatenate(want[30:]))


def test_synth_clip_matches_oracle_generator():
    import torch
    A = _hip()
    for content in (0, 1, 2):
        for (W, H, Cn) in [(64, 48, 1), (31, 17, 3)]:
            want = O.synth_clip(content, W, H, Cn, 5, y0=3, rows=H - 5, k0=2)
            d = torch.empty(want.shape, dtype=torch.uint8, device="cuda")
            A.synth_clip_device(d, content, W, H, Cn, row_begin=3, rows=H - 5, frame_begin=2, num_frames=5,
                                stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(d.cpu().numpy(), want), (content, W, H, Cn)


def _events_device(A, clip_np, *, time_mode, multi_mode, dtm, frames_per_launch=None, env=None, row_band=None,
                   H_total=None, misalign=0):
    """Runs a clip resident in HBM through the device-pointer entry point; returns (events, offsets)."""
    import torch
    T, H, W, Cn = clip_np.shape
    y0, y1 = (0, H) if row_band is None else row_band
    hv = A.HipVideo(W, H if H_total is None else H_total, Cn, row_begin=y0, row_end=y1, time_mode=time_mode,
                    multi_mode=multi_mode, delta_t_max=dtm, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    if frames_per_launch is not None:
        hv.set_frames_per_launch(frames_per_launch)
    h_frames = torch.from_numpy(np.ascontiguousarray(clip_np[:, y0:y1]).reshape(T, -1))
    if misalign:  # the frames start `misalign` bytes into an allocation (a caller's pointer need not be aligned)
        flat = torch.empty(h_frames.numel() + 64, dtype=torch.uint8, device="cuda")
        d_frames = flat[misalign:misalign + h_frames.numel()].view(T, -1)
        d_frames.copy_(h_frames)
        assert d_frames.data_ptr() % 16 == misalign % 16
    else:
        d_frames = h_frames.cuda()
    d_ev = torch.empty((int(d_frames.numel() * 1.3) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=torch.cuda.current_stream().cuda_stream)
    n = hv.finish()
    ev = np.frombuffer(d_ev[:n].cpu().numpy().tobytes(), dtype=A.EVENT_DTYPE).copy()
    offs = d_off.cpu().numpy().copy()
    hv.close()
    return ev, offs


def test_full_size_properties_1080p():
    """BASELINE config 2 size (1920x1080): size-independent properties of the full pipeline --
    the result does not depend on how many frames one launch steps (temporal blocking), on
    graph vs eager submission, or on row-band sharding; a prefix is checked against the oracle."""
    import subprocess, sys
    A = _hip()
    T = 24
    clip = O.synth_clip(O.CONTENT_SCENE, 1920, 1080, 1, T)
    base, offs = _events_device(A, clip, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, dtm=255)
    assert offs[0] == 0 and offs[-1] == len(base) and np.all(np.diff(offs.astype(np.int64)) >= 0)
    for fpl in (1, 3, 5):
        ev, offs2 = _events_device(A, clip, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, dtm=255,
                                   frames_per_launch=fpl)
        assert np.array_equal(offs, offs2) and np.array_equal(base, ev), fpl
    # raster order inside every frame
    for k in range(T):
        seg = base[int(offs[k]):int(offs[k + 1])]
        key = seg["y"].astype(np.int64) * 1920 + seg["x"]
        assert np.all(np.diff(key) >= 0)
    # row bands [0,400) + [400,1080) concatenated per frame == whole plane
    top, ot = _events_device(A, clip, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, dtm=255,
                             row_band=(0, 400))
    bot, ob = _events_device(A, clip, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, dtm=255,
                             row_band=(400, 1080))
    merged = np.concatenate([np.concatenate([top[int(ot[k]):int(ot[k + 1])], bot[int(ob[k]):int(ob[k + 1])]])
                             for k in range(T)])
    assert np.array_equal(merged, base)
    # oracle on the first frames
    ov = O.Video(1920, 1080, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255, threads=8)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    want = np.concatenate([ov.integrate_matrix(clip[k]) for k in range(6)])
    assert np.array_equal(base[: int(offs[6])], want)


def test_rgb_and_secondary_modes_at_size():
    """1920x1080 RGB (config 3 shape) and the delta_t_max = 7650 / Normal / AbsoluteT points on a few frames."""
    A = _hip()
    clip = O.synth_clip(O.CONTENT_SCENE, 1920, 1080, 3, 5)
    for (tm, mm, dtm) in [(O.DELTA_T, O.COLLAPSE, 255), (O.ABSOLUTE_T, O.COLLAPSE, 7650), (O.ABSOLUTE_T, O.NORMAL, 255)]:
        got, offs = _events_device(A, clip, time_mode=tm, multi_mode=mm, dtm=dtm)
        ov = O.Video(1920, 1080, 3, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm, threads=8)
        ov.set_crf_parameters(0, 10)
        ov.reset_c_thresh(0)
        want = np.concatenate([ov.integrate_matrix(f) for f in clip])
        assert len(got) == len(want) and np.array_equal(got, want), (tm, mm, dtm)


def test_eager_and_graph_submission_agree():
    import subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/adder-codec-rs_amd"); sys.path.insert(0, "%s/tests")
import adder_amd as A
from oracle import oracle as O
from test_gpu_parity import _events_device
clip = O.synth_clip(O.CONTENT_NOISE, 640, 480, 1, 40)
ev, offs = _events_device(A, clip, time_mode=A.TIME_ABSOLUTE_T, multi_mode=A.MULTI_COLLAPSE, dtm=255)
import hashlib; print(hashlib.sha256(ev.tobytes() + offs.tobytes()).hexdigest())
''' % (ROOT, ROOT, ROOT)
    outs = []
    for env in ({}, {"ADDER_HIP_NO_GRAPH": "1"}, {"ADDER_HIP_CHUNK": "3", "ADDER_HIP_FRAMES_PER_LAUNCH": "2"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2]


def _oracle_events(clip, *, time_mode, multi_mode, dtm, crf=(0, 0, 10), threads=None):
    T, H, W, Cn = clip.shape
    threads = min(O.max_threads(), 64) if threads is None else threads
    ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=255, delta_t_max=dtm, threads=threads)
    ov.ensure_capacity(8)
    ov.set_crf_parameters(crf[1], crf[2])
    ov.reset_c_thresh(crf[0])
    per = [ov.integrate_matrix(f) for f in clip]
    return np.concatenate(per), np.concatenate([[0], np.cumsum([len(p) for p in per])])


def test_baseline_config_1_plumbing_640x480(tmp_path):
    """BASELINE.json configs[0]: 640x480 gray, 30 frames, raw .adder out -- through the C++ host
    mirror (Framed -> Video -> Encoder), file compared with the oracle's serialisation."""
    import host_py as Hst
    clip = O.synth_clip(O.CONTENT_SCENE, 640, 480, 1, 30)
    out = str(tmp_path / "c1.adder")
    n, chunks = Hst.transcode_raw(clip, fps=30.0, crf=0, ref_time=255, delta_t_max=255, time_mode=0, multi_mode=1,
                                  encoder_crf=0, out_path=out)
    want, _ = _oracle_events(clip, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=255)
    assert n == len(want) and chunks == 480
    blob = O.raw_header(3, 640, 480, 1, 7650, 255, 255, 0, O.DELTA_T, 0) + O.raw_events(want, 1) + O.raw_eof()
    assert open(out, "rb").read() == blob


def test_baseline_config_4_shape_row_bands_3840x2160():
    """configs[3] shape: 3840x2160 gray split into 8 row bands of 270 rows (one context per band, as
    one rank per GPU would hold), merged frame-major in band order == the single-context stream;
    a prefix is checked against the oracle."""
    import torch
    A = _hip()
    from adder_amd import sharding
    T = 6
    clip = O.synth_clip(O.CONTENT_SCENE, 3840, 2160, 1, T)
    whole, offs = _events_device(A, clip, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, dtm=255)
    bands = sharding.row_bands(2160, 8)
    assert bands == [(i * 270, (i + 1) * 270) for i in range(8)]
    segs = []
    for (y0, y1) in bands:
        ev, o = _events_device(A, clip, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, dtm=255,
                               row_band=(y0, y1))
        segs.append((torch.from_numpy(np.frombuffer(ev.tobytes(), dtype=np.int32).reshape(-1, 3).copy()),
                     torch.from_numpy(o.astype(np.int64))))
    merged, moffs = sharding.merge_frame_major(segs)
    got = np.frombuffer(merged.numpy().tobytes(), dtype=A.EVENT_DTYPE)
    assert np.array_equal(moffs.numpy(), offs.astype(np.int64)) and np.array_equal(got, whole)
    want, woffs = _oracle_events(clip[:2], time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=255)
    assert np.array_equal(whole[: int(offs[2])], want)


def test_baseline_config_5_shape_4k_rgb_lossy():
    """configs[4] shape: 3840x2160 RGB, crf-3 numbers (baseline 2, max 7, velocity 7), AbsoluteT,
    Collapse, delta_t_max = 7650 (the generic kernel variants) -- 44 frames against the oracle: past the
    delta_t_max pop of frame 30 (pop_top, popped_dtm, D_EMPTY fillers) and the whole c_thresh ramp."""
    import torch
    A = _hip()
    T = 44
    clip = O.synth_clip(O.CONTENT_SCENE, 3840, 2160, 3, T)
    hv = A.HipVideo(3840, 2160, 3, time_mode=A.TIME_ABSOLUTE_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=7650)
    hv.update_crf(3)
    d_frames = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_ev = torch.empty((int(d_frames.numel() * 0.2) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=torch.cuda.current_stream().cuda_stream)
    n = hv.finish()
    got = np.frombuffer(d_ev[:n].cpu().numpy().tobytes(), dtype=A.EVENT_DTYPE)
    want, woffs = _oracle_events(clip, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, dtm=7650, crf=(2, 7, 7))
    assert np.array_equal(d_off.cpu().numpy(), woffs.astype(np.int64))
    assert n == len(want) and np.array_equal(got, want)
    assert int((want["d"] == 255).sum()) > 0  # D_EMPTY fillers: collapsed flushes after the frame-30 pop happened


@pytest.mark.parametrize("case", ["C3", "C4band", "C5"])
def test_full_size_configs_across_chunk_boundaries(case):
    """BASELINE configs 3, 4 (one band) and 5 at full size for long enough to cross the pipeline's chunk boundaries (64
    frames; 48 for the 4K RGB plane, whose scratch is budget-limited) -- the scratch ring wraps, the state goes out and
    comes back, in config 5 the arenas pop at frame 30 and the quiet-wave loop takes over -- every event and every frame
    offset against the oracle."""
    import torch
    A = _hip()
    if case == "C3":
        W, H, Cn, T, band, tm, dtm, crf = 1920, 1080, 3, 70, None, O.DELTA_T, 255, (0, 0, 10)
    elif case == "C4band":
        W, H, Cn, T, band, tm, dtm, crf = 3840, 2160, 1, 135, (270, 540), O.DELTA_T, 255, (0, 0, 10)
    else:
        W, H, Cn, T, band, tm, dtm, crf = 3840, 2160, 3, 100, None, O.ABSOLUTE_T, 7650, (2, 7, 7)
    y0, y1 = band if band else (0, H)
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, Cn, T, y0=y0, rows=y1 - y0)
    hv = A.HipVideo(W, H, Cn, row_begin=y0, row_end=y1, time_mode=tm, multi_mode=A.MULTI_COLLAPSE, delta_t_max=dtm,
                    c_thresh_start=crf[0], c_counter_start=0)
    hv.set_crf_parameters(crf[1], crf[2])
    d_frames = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_ev = torch.empty((int(d_frames.numel() * (0.1 if case == "C5" else 0.5)) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=torch.cuda.current_stream().cuda_stream)
    n = hv.finish()
    assert T > hv.chunk_frames()  # the clip really crosses a chunk boundary
    got = np.frombuffer(d_ev[:n].cpu().numpy().tobytes(), dtype=A.EVENT_DTYPE)
    del d_frames, d_ev
    ov = O.Video(W, y1 - y0, Cn, row_begin=y0, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm,
                 threads=min(O.max_threads(), 64))  # (the oracle's plane IS the band)
    ov.ensure_capacity(8)
    ov.set_crf_parameters(crf[1], crf[2])
    ov.reset_c_thresh(crf[0])
    pos, offs = 0, d_off.cpu().numpy()
    for k in range(T):  # frame by frame: no second copy of the whole stream
        w = ov.integrate_matrix(clip[k])
        assert int(offs[k]) == pos and int(offs[k + 1]) == pos + len(w), (case, k)
        assert np.array_equal(got[pos:pos + len(w)], w), (case, k)
        pos += len(w)
    assert pos == n and n > 0


@pytest.mark.parametrize("case", ["lean_abs", "default_abs", "default_delta", "normal_abs", "normal_lean"])
def test_full_size_1080p_integer_state_kernels_across_a_chunk_boundary(case):
    """The round-4 kernels at full size, 70 frames of the 1080p scene (a 64-frame chunk + 6: the record slots wrap, the
    state goes out and comes back): lean runs in AbsoluteT, run records in the reference's default mode in both time modes,
    Mode Normal with delta_t_max 7650 and with delta_t_max = time_spanned (flush + pop in one record) -- every event and
    every frame offset against the oracle."""
    import torch
    A = _hip()
    W, H, T = 1920, 1080, 70
    tm, mm, dtm = {"lean_abs": (O.ABSOLUTE_T, O.COLLAPSE, 255), "default_abs": (O.ABSOLUTE_T, O.COLLAPSE, 7650),
                   "default_delta": (O.DELTA_T, O.COLLAPSE, 7650), "normal_abs": (O.ABSOLUTE_T, O.NORMAL, 7650),
                   "normal_lean": (O.DELTA_T, O.NORMAL, 255)}[case]
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, 1, T)
    hv = A.HipVideo(W, H, 1, time_mode=tm, multi_mode=mm, delta_t_max=dtm, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    d_frames = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_ev = torch.empty((int(d_frames.numel() * 0.6) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=torch.cuda.current_stream().cuda_stream)
    n = hv.finish()
    assert T > hv.chunk_frames()
    got = np.frombuffer(d_ev[:n].cpu().numpy().tobytes(), dtype=A.EVENT_DTYPE)
    del d_frames, d_ev
    ov = O.Video(W, H, 1, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm, threads=min(O.max_threads(), 64))
    ov.ensure_capacity(24)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    pos, offs = 0, d_off.cpu().numpy()
    for k in range(T):
        w = ov.integrate_matrix(clip[k])
        assert int(offs[k]) == pos and int(offs[k + 1]) == pos + len(w), (case, k)
        assert np.array_equal(got[pos:pos + len(w)], w), (case, k)
        pos += len(w)
    assert pos == n and n > 0
    hv.close()


def test_full_plane_long_run_1080p_540_frames():
    """1920x1080, 540 frames, the lean kernel: a static plane (every pixel's run exceeds 514 frames, so the
    t of its flush is >= 2^17) with a noisy band and scattered late changes, bit-exact against the oracle
    over the whole clip."""
    import torch
    A = _hip()
    T, H, W = 540, 1080, 1920
    rng = np.random.default_rng(11)
    base = rng.integers(1, 256, (H, W), dtype=np.uint8)
    d_frames = torch.from_numpy(base).cuda().reshape(1, -1).repeat(T, 1)
    view = d_frames.view(T, H, W)
    view[200:, 500:520] = torch.from_numpy(rng.integers(0, 256, (T - 200, 20, W), dtype=np.uint8)).cuda()
    late = torch.from_numpy(rng.random((H, W)) < 0.3).cuda()
    for k in (520, 530, 539):  # 30 % of the plane flushes a > 514-frame run in each of these frames
        view[k:] = torch.where(late, (255 - view[k - 1]).to(torch.uint8), view[k - 1]).unsqueeze(0)
        late = ~late if k == 530 else late
    clip = d_frames.cpu().numpy().reshape(T, H, W, 1)
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=255,
                    c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    want, woffs = _oracle_events(clip, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=255)
    d_ev = torch.empty((len(want) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=torch.cuda.current_stream().cuda_stream)
    n = hv.finish()
    got = np.frombuffer(d_ev[:n].cpu().numpy().tobytes(), dtype=A.EVENT_DTYPE)
    assert int(((want["d"] != 255) & (want["t"] >= 0x1FFFF)).sum()) > 100_000
    assert np.array_equal(d_off.cpu().numpy(), woffs.astype(np.int64))
    assert n == len(want) and np.array_equal(got, want)


def test_quality_change_mid_stream_keeps_parity():
    """update_quality_manual mid-stream (video.rs:1264-1287): generic batches at delta_t_max 7650 leave
    pixels with several fired levels; lowering delta_t_max to ref_time afterwards must NOT switch to the
    lean kernel (which only understands m <= 1) -- the variant choice is sticky until reset."""
    A = _hip()
    clip = clips.make_clip("runs", 150, 24, 40, 1, seed=99)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov = O.Video(40, 24, 1, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=7650)
        hv = A.HipVideo(40, 24, 1, time_mode=tm, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=7650)
        ov.ensure_capacity(22)
        for v in (ov, hv):
            v.set_crf_parameters(0, 10)
            v.reset_c_thresh(0)
        want = [ov.integrate_matrix(f) for f in clip[:60]]
        got, offs = hv.integrate_batch(clip[:60])
        assert np.array_equal(got, np.concatenate(want))
        ov.set_crf_parameters(3, 4)   # update_quality_manual(baseline 1, max 3, multiplier 1, velocity 4)
        ov.set_delta_t_max(255)
        ov.reset_c_thresh(1)
        hv.update_quality_manual(1, 3, 1, 4)
        want = [ov.integrate_matrix(f) for f in clip[60:]]
        got, offs = hv.integrate_batch(clip[60:])
        assert np.array_equal(got, np.concatenate(want)), tm
        # after a reset the lean kernel is allowed again and still matches
        hv.reset()
        ov2 = O.Video(40, 24, 1, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
        ov2.set_crf_parameters(3, 4)
        ov2.reset_c_thresh(1)
        hv.reset_c_thresh(1)
        want = [ov2.integrate_matrix(f) for f in clip[:40]]
        got, offs = hv.integrate_batch(clip[:40])
        assert np.array_equal(got, np.concatenate(want)), tm


def test_merge_kernel_equals_single_context_stream():
    """adder_hip_merge_streams_device: three row bands' streams laid back to back -> one frame-major stream
    == the whole-plane context's (the multi-GPU merge, on one GPU)."""
    import torch
    A = _hip()
    from adder_amd import sharding
    T, H, W = 37, 120, 200
    clip = clips.make_clip("runs", T, H, W, 1, seed=21)
    whole, offs = _events_device(A, clip, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, dtm=255)
    bands = sharding.row_bands(H, 3)
    evs, oss = [], []
    for (y0, y1) in bands:
        ev, o = _events_device(A, clip, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, dtm=255, row_band=(y0, y1))
        evs.append(np.frombuffer(ev.tobytes(), dtype=np.int32).reshape(-1, 3))
        oss.append(o.astype(np.int64))
    stage = torch.from_numpy(np.concatenate(evs)).cuda()
    all_offs = torch.from_numpy(np.stack(oss)).cuda()
    hv = A.HipVideo(W, H, 1)
    out = torch.full((len(whole) + 5, 3), -1, dtype=torch.int32, device="cuda")
    moffs = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.merge_streams_device(stage, all_offs, 3, T, out, moffs)
    hv.check_status()
    got = np.frombuffer(out[: len(whole)].cpu().numpy().tobytes(), dtype=A.EVENT_DTYPE)
    assert np.array_equal(moffs.cpu().numpy(), offs.astype(np.int64)) and np.array_equal(got, whole)
    assert int((out[len(whole):] != -1).sum()) == 0
    # a merged buffer that is too small is reported, not overrun
    small = torch.full((len(whole) - 7, 3), -1, dtype=torch.int32, device="cuda")
    hv.merge_streams_device(stage, all_offs, 3, T, small, None)
    with pytest.raises(A.AdderHipError) as ei:
        hv.check_status()
    assert ei.value.code == A.E_OUT_CAPACITY


def test_gather_cabi_single_rank_world():
    """libadder_rccl.so (include/adder_gather.h) end to end with a real RCCL communicator of one rank: the
    unique id, ncclCommInitRank, the offsets all-gather, the merge -- the calls a Rust host makes."""
    import torch
    A = _hip()
    from adder_amd.gather import HipGather, unique_id
    T, H, W = 20, 64, 96
    clip = clips.make_clip("runs", T, H, W, 1, seed=5)
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=255,
                    c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    st = torch.cuda.current_stream().cuda_stream
    d_frames = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_ev = torch.empty((W * H * T * 3, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=st)
    n = hv.finish()
    g = HipGather(hv, unique_id(), 0, 1)
    d_m = torch.full((n + 3, 3), -1, dtype=torch.int32, device="cuda")
    d_mo = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    assert g.gather_events(d_ev, d_off, T, 0, d_m, d_mo, stream=st) == n
    assert torch.equal(d_m[:n], d_ev[:n]) and torch.equal(d_mo, d_off) and int((d_m[n:] != -1).sum()) == 0
    merged, base = g.layout(d_off, T, stream=st)
    assert np.array_equal(merged.astype(np.int64), d_off.cpu().numpy()) and np.array_equal(base, merged[:-1])
    # the chunk-range entry point (adder_gather_events_at): three chunks appended one after the other, the rank's
    # offsets passed as they are (they do not start at 0 for the later chunks), on a side stream
    d_m2 = torch.full((n + 3, 3), -1, dtype=torch.int32, device="cuda")
    d_mo2 = torch.full((T + 1,), -7, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    pos = 0
    for f0, nf in ((0, 8), (8, 8), (16, 4)):
        pos += g.gather_events_at(d_ev, d_off, f0, nf, 0, d_m2, pos, d_mo2, stream=side.cuda_stream)
    side.synchronize()
    assert pos == n and torch.equal(d_m2[:n], d_ev[:n]) and torch.equal(d_mo2, d_off) and int((d_m2[n:] != -1).sum()) == 0
    with pytest.raises(A.AdderHipError, match="too small"):
        g.gather_events_at(d_ev, d_off, 0, T, 0, d_m2[: n - 1], 0, d_mo2, stream=side.cuda_stream)
    # records over the wire through the same communicator (adder_gather_records_at): three chunks, the image round trip,
    # root's expansion on the side stream while the context integrates the next chunk
    hv.reset()
    d_m3 = torch.full((n + 3, 3), -1, dtype=torch.int32, device="cuda")
    d_mo3 = torch.full((T + 1,), -7, dtype=torch.int64, device="cuda")
    d_boff = torch.zeros(9, dtype=torch.int64, device="cuda")
    pos = 0
    for f0, nf in ((0, 8), (8, 8), (16, 4)):
        rec = hv.integrate_records_device(d_frames[f0:f0 + nf], d_boff, stream=st)
        n_k = hv.finish()
        side.wait_stream(torch.cuda.current_stream())
        pos += g.gather_records_at(rec, hv.last_batch_records(), n_k, 0, d_m3, pos, d_mo3[f0:], stream=side.cuda_stream)
    side.synchronize()
    hv.expand_status(side.cuda_stream)
    assert pos == n and torch.equal(d_m3[:n], d_ev[:n]) and torch.equal(d_mo3, d_off) and int((d_m3[n:] != -1).sum()) == 0
    with pytest.raises(A.AdderHipError, match="too small"):
        hv.reset()
        rec = hv.integrate_records_device(d_frames[:8], d_boff, stream=st)
        n_k = hv.finish()
        g.gather_records_at(rec, hv.last_batch_records(), n_k, 0, d_m3[: n_k - 1], 0, d_mo3, stream=st)
    # the streamed form with the raw sink's records as root's output (adder_gather_records_begin_wire / _push / _end: what
    # bench.py --gpus N runs): lean-runs records through the real communicator, root's wire bytes == the oracle's raw sink
    ov = O.Video(W, H, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    want = O.raw_events(np.concatenate([ov.integrate_matrix(f) for f in clip]), 1)
    hv.reset()
    d_w = torch.full((n * 9 + 32,), 0xAB, dtype=torch.uint8, device="cuda")
    d_mo4 = torch.full((T + 1,), -7, dtype=torch.int64, device="cuda")
    g.records_begin_wire(0, d_w, 0, d_mo4, stream=side.cuda_stream)
    for f0, nf in ((0, 8), (8, 8), (16, 4)):
        rec = hv.integrate_records_device(d_frames[f0:f0 + nf], d_boff, stream=st)
        n_k = hv.finish()
        assert rec.record_bytes == (8 | 0x100)
        g.records_push(rec, hv.last_batch_records(), n_k)
    n_m, _ = g.records_end()
    got = d_w.cpu().numpy()
    assert n_m == n and got[:n * 9].tobytes() == want and (got[n * 9:] == 0xAB).all() and torch.equal(d_mo4, d_off)
    g.close()


def test_two_ranks_hip_video_gather_shared_device():
    """Two processes, one HipVideo per rank on its row band (both on cuda:0 over gloo: a 1-GPU box cannot
    host two RCCL ranks), gather_event_stream with the HIP merge kernel == the single-context stream."""
    import subprocess, sys
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731",
                        os.path.join(ROOT, "tests", "mp_hip_gather.py")],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "rank0 ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
def test_fast_division_is_exact_on_its_domain():
    """step_fast's 4-instruction division == IEEE f32 division for every integer numerator in
    [1, 2^24] and every u8 intensity in [1, 255] (adder_pixel.hpp fdiv_small)."""
    import ctypes
    import adder_amd
    lib = adder_amd.load()
    bad = ctypes.c_uint64(123)
    assert lib.adder_hip_selftest_division(ctypes.byref(bad)) == 0
    assert bad.value == 0


@pytest.mark.gpu
def test_lake_golden_bytes_device_sink(golden_dir):
    """Same golden, but the events are serialised by the device-side raw sink (K3)."""
    A = _hip()
    raw = gzip.open(os.path.join(golden_dir, "lake_scaled_hd_out.adder.gz")).read()
    frames = np.load(os.path.join(golden_dir, "lake_scaled_hd_frames_reconstructed.npz"))["frames"]
    T, H, W = frames.shape[:3]
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_NORMAL, ref_time=255, delta_t_max=6120)
    hv.update_crf(0)
    body = b""
    n_total = 0
    for k0 in range(0, T, 32):  # several batches: state carries over, bytes concatenate
        b, n, offs = hv.integrate_batch_raw(frames[k0:k0 + 32])
        assert len(b) == 9 * n and int(offs[-1]) == n
        body += b
        n_total += n
    assert n_total == 201620
    assert raw[37:-11] == body


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3])
def test_device_sink_equals_host_sink(channels):
    """adder_hip_wire_events_device == adder_raw_events == oracle raw sink, incl. ragged sizes
    (partial last workgroup, partial last dword)."""
    import torch
    A = _hip()
    clip = clips.make_clip("noise", 5, 37, 53, channels, seed=11)
    hv = A.HipVideo(53, 37, channels, time_mode=A.TIME_ABSOLUTE_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=255,
                    c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    ev, _ = hv.integrate_batch(clip)
    assert len(ev) > 3000
    rec = 9 if channels == 1 else 11
    for n in (0, 1, 2, 3, 5, 1023, 1024, 1025, 2049, len(ev)):
        sub = np.ascontiguousarray(ev[:n])
        want = O.raw_events(sub, channels)
        assert A.raw_events(sub, channels) == want
        d_ev = torch.from_numpy(sub.view(np.uint8).copy()).cuda() if n else torch.zeros(12, dtype=torch.uint8, device="cuda")
        d_out = torch.full((n * rec + 64,), 0xAB, dtype=torch.uint8, device="cuda")
        nb = hv.wire_events_device(d_ev, n, d_out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert nb == n * rec
        got = d_out.cpu().numpy()
        assert got[:nb].tobytes() == want
        assert (got[nb:] == 0xAB).all()  # nothing written past the stream's end


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3])
def test_expansion_writes_wire_records_directly(channels):
    """adder_hip_integrate_wire_device: the expansion itself serialises (no AdderEvents in between).  Every expansion
    format (lean runs, lean AbsoluteT / DeltaT, run records, constant runs, bounded Collapse, generic Normal), ragged
    planes, batches that start at every stream index mod 4, a sentinel past the stream's end: the bytes equal the
    oracle's raw sink byte for byte and the offsets still count events."""
    import torch
    A = _hip()
    W, H = 157, 61
    rec = 9 if channels == 1 else 11
    st = torch.cuda.current_stream().cuda_stream
    clip = clips.make_clip("runs", 72, H, W, channels, seed=31 + channels)
    cases = [(O.DELTA_T, O.COLLAPSE, 255, 0), (O.ABSOLUTE_T, O.COLLAPSE, 255, 0), (O.DELTA_T, O.COLLAPSE, 255, 3),
             (O.ABSOLUTE_T, O.COLLAPSE, 7650, 0), (O.DELTA_T, O.COLLAPSE, 7650, 0), (O.ABSOLUTE_T, O.COLLAPSE, 7650, 3),
             (O.DELTA_T, O.NORMAL, 255, 0), (O.ABSOLUTE_T, O.NORMAL, 7650, 3)]
    for tm, mm, dtm, crf in cases:
        ov = O.Video(W, H, channels, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm)
        hv = A.HipVideo(W, H, channels, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm)
        ov.ensure_capacity(24)
        for v in (ov, hv):
            v.set_crf_parameters(CRFS[crf][1], CRFS[crf][2])
            v.reset_c_thresh(CRFS[crf][0])
        k = 0
        for nb in (1, 2, 3, 30, 34):
            want = [ov.integrate_matrix(f) for f in clip[k:k + nb]]
            n = sum(len(w) for w in want)
            d_frames = torch.from_numpy(np.ascontiguousarray(clip[k:k + nb]).reshape(nb, -1)).cuda()
            d_wire = torch.full((n * rec + 64,), 0xAB, dtype=torch.uint8, device="cuda")
            d_offs = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
            hv.integrate_wire_device(d_frames, d_wire, d_offs, stream=st)
            hv.finish()
            offs = d_offs.cpu().numpy()
            assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == [len(w) for w in want], (tm, mm, dtm, crf, k)
            got = d_wire.cpu().numpy()
            assert got[:n * rec].tobytes() == O.raw_events(np.concatenate(want), channels), (tm, mm, dtm, crf, k, nb)
            assert (got[n * rec:] == 0xAB).all()
            k += nb
        # too small a buffer: reported with the size needed, rolled back, and the retry continues the stream
        want = np.concatenate([ov.integrate_matrix(f) for f in clip[k:k + 2]])
        d_frames = torch.from_numpy(np.ascontiguousarray(clip[k:k + 2]).reshape(2, -1)).cuda()
        d_small = torch.zeros(max(rec, (len(want) // 2) * rec), dtype=torch.uint8, device="cuda")
        d_offs = torch.zeros(3, dtype=torch.int64, device="cuda")
        hv.integrate_wire_device(d_frames, d_small, d_offs, stream=st)
        with pytest.raises(A.AdderHipError):
            hv.finish()
        d_wire = torch.zeros(len(want) * rec, dtype=torch.uint8, device="cuda")
        hv.integrate_wire_device(d_frames, d_wire, d_offs, stream=st)
        hv.finish()
        assert d_wire.cpu().numpy().tobytes() == O.raw_events(want, channels), (tm, mm, dtm, crf)
        hv.close()


def test_lake_golden_bytes_pipelined_stream(golden_dir):
    """Same golden through the pipelined submit/collect form (two batches in flight)."""
    A = _hip()
    raw = gzip.open(os.path.join(golden_dir, "lake_scaled_hd_out.adder.gz")).read()
    frames = np.load(os.path.join(golden_dir, "lake_scaled_hd_frames_reconstructed.npz"))["frames"]
    hv = A.HipVideo(200, 50, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_NORMAL, ref_time=255, delta_t_max=6120)
    hv.update_crf(0)
    batches = [frames[k:k + 24] for k in range(0, len(frames), 24)]
    body, n_total = b"", 0
    hv.stream_submit(batches[0])
    for k in range(len(batches)):
        if k + 1 < len(batches):
            hv.stream_submit(batches[k + 1])
        b, n, _ = hv.stream_collect()
        assert len(b) == 9 * n
        body += b
        n_total += n
    assert n_total == 201620 and raw[37:-11] == body
    with pytest.raises(Exception):
        hv.stream_collect()  # nothing in flight


@pytest.mark.parametrize("tmode", ["ABSOLUTE_T", "DELTA_T"])
def test_batch_lengths_around_chunk_and_lag_boundaries(tmode):
    """The scratch ring holds three chunks of 64 frames and a launch steps up to 64 frames: batch lengths on
    both sides of every boundary (launch depth, chunk, ring wrap-around at 192), several submission forms, consecutive batches on one context -- always
    the oracle's stream and frame offsets.  (AbsoluteT: the lean-runs kernel and the offsets kernel; DeltaT: the packed
    kernels, whose scan chains the frame offsets itself -- several frame-kernel launches per chunk included.)"""
    import subprocess, sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/adder-codec-rs_amd"); sys.path.insert(0, "%s/tests")
import torch
import adder_amd as A
from oracle import oracle as O
import clips
W, H = 70, 23
lens = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 191, 192, 193, 257, 5]
clip = clips.make_clip("runs", sum(lens), H, W, 1, seed=4)
ov = O.Video(W, H, 1, time_mode=O.TMODE, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
ov.set_crf_parameters(0, 10); ov.reset_c_thresh(0)
hv = A.HipVideo(W, H, 1, time_mode=A.TIME_TMODE, multi_mode=A.MULTI_COLLAPSE, delta_t_max=255,
                c_thresh_start=0, c_counter_start=0)
hv.set_crf_parameters(0, 10)
st = torch.cuda.current_stream().cuda_stream
k0 = 0
for T in lens:
    sub = clip[k0:k0 + T]; k0 += T
    per = [ov.integrate_matrix(f) for f in sub]
    want = np.concatenate(per)
    d_frames = torch.from_numpy(sub.reshape(T, -1)).cuda()
    d_ev = torch.full((W * H * T * 4, 3), -1, dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=st)
    n = hv.finish()
    got = np.frombuffer(d_ev[:n].cpu().numpy().tobytes(), dtype=A.EVENT_DTYPE)
    assert n == len(want) and np.array_equal(got, want), T
    assert d_off.cpu().tolist() == np.concatenate([[0], np.cumsum([len(p) for p in per])]).tolist(), T
    assert int((d_ev[n:] != -1).sum()) == 0, T  # nothing written past the stream
    if "TMODE" == "DELTA_T" and T > 1 and not os.environ.get("ADDER_HIP_FRAMES_PER_LAUNCH") == "1":
        assert hv.last_batch_kernel() == A.KERNEL_LEAN_RUNS_PACKED, (T, hv.last_batch_kernel())
print("ok")
''' % (ROOT, ROOT, ROOT)
    code = code.replace("TMODE", tmode)
    for env in ({}, {"ADDER_HIP_NO_GRAPH": "1"}, {"ADDER_HIP_NO_GRAPH": "2"},
                {"ADDER_HIP_FRAMES_PER_LAUNCH": "1"}, {"ADDER_HIP_FRAMES_PER_LAUNCH": "5"},
                {"ADDER_HIP_CHUNK": "4", "ADDER_HIP_FRAMES_PER_LAUNCH": "4"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (env, r.stdout[-500:], r.stderr[-2000:])


def test_baseline_config_5_end_to_end_compressed_sink_4k_rgb():
    """configs[4] end to end on one GPU at full plane size: 3840x2160 RGB, crf-3 numbers, AbsoluteT, Collapse,
    delta_t_max 7650 -> events (HIP) -> adu_interval 30 compressed sink (CPU, include/adder_compressed.h) ->
    decode: every ADU decodes, per pixel the d sequences survive and t stays within the lossy tolerance.
    (Bit-exactness of the events is test_baseline_config_5_shape_4k_rgb_lossy; of the sink's bytes,
    tests/test_compressed_product.py.)"""
    import torch
    A = _hip()
    T = 36
    clip = O.synth_clip(O.CONTENT_SCENE, 3840, 2160, 3, T)
    hv = A.HipVideo(3840, 2160, 3, time_mode=A.TIME_ABSOLUTE_T, multi_mode=A.MULTI_COLLAPSE, delta_t_max=7650)
    hv.update_crf(3)
    d_frames = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_ev = torch.empty((int(d_frames.numel() * 0.2) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=torch.cuda.current_stream().cuda_stream)
    n = hv.finish()
    ev = np.frombuffer(d_ev[:n].cpu().numpy().tobytes(), dtype=A.EVENT_DTYPE)
    enc = A.CompressedEncoder(3840, 2160, 3, tps=7650, ref_interval=255, delta_t_max=7650, adu_interval=30,
                              time_mode=A.TIME_ABSOLUTE_T, c_thresh_max=7, threads=16)
    offs = d_off.cpu().numpy()
    for k in range(T):  # per frame, as Video::integrate_matrix hands its events to the encoder
        enc.ingest(ev[int(offs[k]):int(offs[k + 1])])
    blob = enc.close()
    enc.destroy()
    assert blob[:5] == b"addec" and len(blob) < n * 11 * 0.8
    dec, p = A.compressed_decode(blob)
    assert (p.width, p.height, p.channels) == (3840, 2160, 3) and 0.95 * n < len(dec) <= n
    # a sample of pixels: same d sequence, t within one reference interval
    key_in = (ev["y"].astype(np.int64) * 3840 + ev["x"]) * 3 + ev["c"]
    key_out = (dec["y"].astype(np.int64) * 3840 + dec["x"]) * 3 + dec["c"]
    oi, oo = np.argsort(key_in, kind="stable"), np.argsort(key_out, kind="stable")
    ki, ko = key_in[oi], key_out[oo]
    rng = np.random.default_rng(3)
    checked = 0
    for px in rng.choice(np.unique(ki), 3000, replace=False):
        a = ev[oi[np.searchsorted(ki, px, "left"):np.searchsorted(ki, px, "right")]]
        b = dec[oo[np.searchsorted(ko, px, "left"):np.searchsorted(ko, px, "right")]]
        if len(a) != len(b):
            continue  # the cube's drop rule removed an event of this pixel
        assert np.array_equal(a["d"], b["d"])
        assert np.all(np.abs(a["t"].astype(np.int64) - b["t"].astype(np.int64)) <= 255 * 8)
        checked += 1
    assert checked > 2500


@pytest.mark.parametrize("multi_mode", [O.NORMAL, O.COLLAPSE])
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_continuous_mode_kernel(multi_mode, time_mode):
    """Mode::Continuous through the C-ABI (adder_cont_kernel + the staged expansion) against the
    oracle, whose Continuous paths are pinned by the reference's PixelArena unit tests: remainders handed to the
    child, zero events, set_d_for_continuous, pop_top on a root without a best event -- per-frame calls (with chunk
    offsets), batches, RGB, ragged planes, a row band, the running-intensities side plane."""
    A = _hip()
    cases = [("dark", 21, 13, 1, 255, 0), ("runs", 33, 20, 1, 7650, 3), ("jitter", 9, 130, 3, 1020, 9),
             ("noise", 40, 40, 1, 255, 0), ("steps", 16, 17, 3, 7650, 0)]
    for kind, H, W, Cn, dtm, crf in cases:
        clip = clips.make_clip(kind, 70, H, W, Cn, seed=H * 7 + dtm)
        y0, y1 = (0, H) if kind != "runs" else (5, 17)
        sub = clip[:, y0:y1]
        ov = O.Video(W, y1 - y0, Cn, row_begin=y0, time_mode=time_mode, multi_mode=multi_mode, ref_time=255,
                     delta_t_max=dtm, chunk_rows=3)
        ov.set_pixel_mode(1)
        hv = A.HipVideo(W, H, Cn, row_begin=y0, row_end=y1, time_mode=time_mode, multi_mode=multi_mode, ref_time=255,
                        delta_t_max=dtm, chunk_rows=3, max_depth=24, pixel_mode=1)
        ov.ensure_capacity(28)
        base, cmax, vel = CRFS[crf]
        for v in (ov, hv):
            v.set_crf_parameters(cmax, vel)
            v.reset_c_thresh(base)
        if kind == "dark":
            hv.enable_running_intensities(True)
        total = 0
        for k in range(30):
            a, ca = ov.integrate_matrix(sub[k], want_chunks=True)
            b, cb = hv.integrate_matrix(sub[k], want_chunks=True)
            assert np.array_equal(a, b) and np.array_equal(ca, cb), (kind, k)
            total += len(a)
        if kind == "dark":
            assert np.array_equal(hv.running_intensities(), ov.running_intensities())
        want = [ov.integrate_matrix(f) for f in sub[30:]]
        got, offs = hv.integrate_batch(sub[30:])
        assert np.array_equal(got, np.concatenate(want)), kind
        assert [int(offs[i + 1] - offs[i]) for i in range(40)] == [len(w) for w in want]
        assert total + len(got) > 0


def _feature_pair(A, W, H, Cn, *, multi_mode, dtm, time_mode=O.ABSOLUTE_T, chunk_rows=1, crf=(13, 4), baseline=6,
                  detect=True, adjust=True, radius=3, roi=None, max_depth=30):
    ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=255, delta_t_max=dtm,
                 chunk_rows=chunk_rows)
    hv = A.HipVideo(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=255, delta_t_max=dtm,
                    chunk_rows=chunk_rows, max_depth=max_depth)
    ov.ensure_capacity(max_depth + 2)
    for v in (ov, hv):
        v.set_crf_parameters(*crf)
        v.reset_c_thresh(baseline)
    ov.update_detect_features(detect, adjust, baseline, radius)
    ov.set_roi(roi, baseline)
    hv.update_detect_features(detect, adjust)
    hv.set_feature_parameters(baseline, radius)
    hv.update_roi(roi)
    return ov, hv


def _same_feature_state(ov, hv, detect=True):
    assert np.array_equal(hv.c_thresh_plane(), ov.c_thresh_plane())
    if detect:
        assert np.array_equal(hv.feature_set(), ov.feature_set())
        assert np.array_equal(hv.running_intensities(), ov.running_intensities())


@pytest.mark.gpu
@pytest.mark.parametrize("multi_mode,dtm", [(O.COLLAPSE, 255), (O.COLLAPSE, 7650), (O.NORMAL, 2550)])
@pytest.mark.parametrize("channels", [1, 3])
def test_feature_driven_rate_control(multi_mode, dtm, channels):
    """FAST 9_16 on the running intensities at the pixels that fired, VideoState::features, and the
    c_thresh reset around new features (adder_feature_kernel + the per-pixel c_thresh planes of the generic K1)
    against the oracle's handle_features -- per-frame calls, then a batch (frame f+1 must see frame f's resets
    inside ONE C-ABI call), ragged planes, row chunks of several sizes."""
    A = _hip()
    for (H, W, chunk_rows, tm) in [(40, 56, 1, O.ABSOLUTE_T), (67, 131, 5, O.DELTA_T)]:
        clip = clips.make_clip("corners", 40, H, W, channels, seed=H + channels)
        ov, hv = _feature_pair(A, W, H, channels, multi_mode=multi_mode, dtm=dtm, time_mode=tm, chunk_rows=chunk_rows)
        new = 0
        for k in range(14):
            a, ca = ov.integrate_matrix(clip[k], want_chunks=True)
            b, cb = hv.integrate_matrix(clip[k], want_chunks=True)
            assert np.array_equal(a, b) and np.array_equal(ca, cb), k
            assert hv.last_new_features() == len(ov.new_features()), k
            new += len(ov.new_features())
            if k % 4 == 0:
                _same_feature_state(ov, hv)
        assert new > 20
        want = []
        batch_new = 0
        for f in clip[14:]:
            want.append(ov.integrate_matrix(f))
            batch_new += len(ov.new_features())
        got, offs = hv.integrate_batch(clip[14:])
        assert np.array_equal(got, np.concatenate(want))
        assert [int(offs[i + 1] - offs[i]) for i in range(len(want))] == [len(w) for w in want]
        assert hv.last_new_features() == batch_new
        _same_feature_state(ov, hv)


@pytest.mark.gpu
def test_feature_detection_without_adjustment_keeps_the_lean_kernel_and_the_stream():
    A = _hip()
    clip = clips.make_clip("corners", 24, 48, 64, 1, seed=3)
    ov, hv = _feature_pair(A, 64, 48, 1, multi_mode=O.COLLAPSE, dtm=255, adjust=False)
    plain = A.HipVideo(64, 48, 1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255,
                       max_depth=30)
    plain.set_crf_parameters(13, 4)
    plain.reset_c_thresh(6)
    want = [ov.integrate_matrix(f) for f in clip]
    got, _ = hv.integrate_batch(clip)
    ref, _ = plain.integrate_batch(clip)
    assert np.array_equal(got, np.concatenate(want)) and np.array_equal(got, ref)
    _same_feature_state(ov, hv)
    assert ov.feature_set().sum() > 0 and len(np.unique(hv.c_thresh_plane())) == 1


@pytest.mark.gpu
def test_roi_and_mode_switches_mid_stream():
    """update_roi / update_detect_features / update_crf between calls: uniform thresholds -> per-pixel (ROI) ->
    features on top -> update_crf makes them uniform again (video.rs:1241-1251), all against the oracle."""
    A = _hip()
    W, H = 72, 50
    clip = clips.make_clip("corners", 50, H, W, 1, seed=21)
    ov, hv = _feature_pair(A, W, H, 1, multi_mode=O.COLLAPSE, dtm=255, detect=False, adjust=False, radius=0,
                           baseline=9, crf=(20, 2))
    def step(lo, hi, detect):
        want = [ov.integrate_matrix(f) for f in clip[lo:hi]]
        got, _ = hv.integrate_batch(clip[lo:hi])
        assert np.array_equal(got, np.concatenate(want)), (lo, hi)
        _same_feature_state(ov, hv, detect)
    step(0, 10, False)  # lean kernel, uniform
    roi = (10, 8, 40, 30)
    ov.set_roi(roi, 9)
    hv.update_roi(roi)
    step(10, 20, False)
    assert (hv.c_thresh_plane()[8:31, 10:41] == 2).all()
    ov.update_detect_features(True, True, 9, 5)
    hv.update_detect_features(True, True)
    hv.set_feature_parameters(9, 5)
    step(20, 32, True)
    ov.set_roi(None, 9)
    hv.update_roi(None)
    step(32, 40, True)
    # update_crf(6): baseline 7, max 13, velocity 4, every pixel back to the baseline; features stay on
    ov.set_crf_parameters(13, 4)
    ov.reset_c_thresh(7)
    ov.update_detect_features(True, True, 7, A.video.crf_feature_radius(6, W, H))
    hv.update_crf(6)
    step(40, 50, True)


@pytest.mark.gpu
def test_features_at_1080p_default_radius():
    """Full plane, the default feature radius of quality 3 (1080 / 15 = 72 px): thousands of new features on the
    first frames, each resetting a 145 x 145 neighbourhood."""
    A = _hip()
    W, H = 1920, 1080
    clip = clips.make_clip("corners", 5, H, W, 1, seed=1)
    ov, hv = _feature_pair(A, W, H, 1, multi_mode=O.COLLAPSE, dtm=255, chunk_rows=64, crf=(7, 7), baseline=2,
                           radius=A.video.crf_feature_radius(3, W, H))
    ov.set_threads(8)
    want = []
    new = 0
    for f in clip:
        want.append(ov.integrate_matrix(f))
        new += len(ov.new_features())
    got, offs = hv.integrate_batch(clip)
    assert np.array_equal(got, np.concatenate(want))
    assert hv.last_new_features() == new and new > 1000
    _same_feature_state(ov, hv)


@pytest.mark.gpu
def test_feature_path_errors_and_rollback():
    A = _hip()
    # a row band takes the mode, but one frame per call and with the per-frame halo protocol
    band = A.HipVideo(64, 48, 1, row_begin=8, row_end=40)
    band.update_detect_features(True, True)
    with pytest.raises(A.AdderHipError, match="one frame per call"):
        band.integrate_batch(np.zeros((3, 32, 64), np.uint8))
    band.integrate_matrix(np.zeros((32, 64), np.uint8))
    with pytest.raises(A.AdderHipError, match="feature step is missing"):
        band.integrate_matrix(np.zeros((32, 64), np.uint8))
    band.close()
    cont = A.HipVideo(64, 48, 1, pixel_mode=1)
    with pytest.raises(A.AdderHipError):
        cont.update_detect_features(True, False)
    # an event buffer that is too small: the per-pixel thresholds, the feature set and the side plane roll back too
    clip = clips.make_clip("corners", 20, 40, 56, 1, seed=13)
    ov, hv = _feature_pair(A, 56, 40, 1, multi_mode=O.COLLAPSE, dtm=255)
    want = [ov.integrate_matrix(f) for f in clip]
    got, _ = hv.integrate_batch(clip[:8])
    assert np.array_equal(got, np.concatenate(want[:8]))
    need = sum(len(w) for w in want[8:])
    with pytest.raises(A.AdderHipError) as ei:
        hv.integrate_batch(clip[8:], out_cap=need // 2)
    assert ei.value.code == A.E_OUT_CAPACITY
    got, _ = hv.integrate_batch(clip[8:], out_cap=need)
    assert np.array_equal(got, np.concatenate(want[8:]))
    _same_feature_state(ov, hv)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["lean", "generic", "normal_rgb", "continuous", "features"])
def test_frame_ring_submit_collect(case):
    """The per-frame `consume` contract without a blocking round trip (adder_hip_frame_submit / _collect): up to
    three frames in flight, each handed over by adder_frame_out_kernel into page-locked host memory together with
    its row-chunk offsets -- events and chunks equal the oracle's per frame, in every kernel family."""
    A = _hip()
    W, H, Cn, mm, dtm, tm, pm = 131, 67, 1, O.COLLAPSE, 255, O.ABSOLUTE_T, 0
    if case == "generic":
        dtm, tm = 7650, O.DELTA_T
    elif case == "normal_rgb":
        Cn, mm, dtm = 3, O.NORMAL, 2550
    elif case == "continuous":
        pm, dtm = 1, 1020
    clip = clips.make_clip("corners" if case == "features" else "runs", 45, H, W, Cn, seed=17)
    ov = O.Video(W, H, Cn, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm, chunk_rows=7)
    hv = A.HipVideo(W, H, Cn, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm, chunk_rows=7, max_depth=30,
                    pixel_mode=pm)
    ov.ensure_capacity(34)
    if pm:
        ov.set_pixel_mode(1)
    for v in (ov, hv):
        v.set_crf_parameters(7, 7)
        v.reset_c_thresh(2)
    if case == "features":
        ov.update_detect_features(True, True, 2, 3)
        hv.update_detect_features(True, True)
        hv.set_feature_parameters(2, 3)
    want = [ov.integrate_matrix(f, want_chunks=True) for f in clip]
    pinned = [hv.pinned_frame() for _ in range(3)]
    got = []
    for k, f in enumerate(clip):
        if hv.frames_in_flight() == 3:
            got.append(hv.frame_collect(want_chunks=True))
        pinned[k % 3][...] = f.reshape(H, W * Cn)
        hv.frame_submit(pinned[k % 3])
    while hv.frames_in_flight():
        got.append(hv.frame_collect(want_chunks=True))
    assert len(got) == len(want)
    for k, ((a, ca), (b, cb)) in enumerate(zip(want, got)):
        assert np.array_equal(a, b) and np.array_equal(ca, cb), (case, k)
    # the blocking call afterwards continues the same stream (pageable frame, pageable and pinned `out`)
    extra = clips.make_clip("runs", 4, H, W, Cn, seed=18)
    for f in extra:
        a, ca = ov.integrate_matrix(f, want_chunks=True)
        b, cb = hv.integrate_matrix(f, want_chunks=True)
        assert np.array_equal(a, b) and np.array_equal(ca, cb)


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3])
def test_frame_ring_hands_out_wire_records(channels):
    """adder_hip_frames_set_format(ctx, 1): the hand-over kernel serialises on the device and the slots receive the
    9 / 11-byte records RawOutput::ingest_event writes (raw/stream.rs:101-120) -- per frame they equal the oracle's
    raw_events of that frame, the chunk offsets stay in events, and switching back mid-stream continues the stream."""
    A = _hip()
    W, H = 157, 61
    clip = clips.make_clip("runs", 40, H, W, channels, seed=23)
    ov = O.Video(W, H, channels, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255, chunk_rows=5)
    hv = A.HipVideo(W, H, channels, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255, chunk_rows=5)
    ov.ensure_capacity(4)
    for v in (ov, hv):
        v.set_crf_parameters(0, 10)
        v.reset_c_thresh(0)
    hv.frames_set_format(True)
    rec = 9 if channels == 1 else 11
    want = [ov.integrate_matrix(f, want_chunks=True) for f in clip[:30]]
    pinned = [hv.pinned_frame() for _ in range(3)]
    got = []
    for k, f in enumerate(clip[:30]):
        if hv.frames_in_flight() == 3:
            got.append(hv.frame_collect_wire(want_chunks=True))
        pinned[k % 3][...] = f.reshape(H, W * channels)
        hv.frame_submit(pinned[k % 3])
    with pytest.raises(A.AdderHipError):  # the format belongs to the frames in flight
        hv.frames_set_format(False)
    while hv.frames_in_flight():
        got.append(hv.frame_collect_wire(want_chunks=True))
    assert len(got) == 30 and sum(n for _, n, _ in got) > 1000
    for k, ((ev, ch), (data, n, gch)) in enumerate(zip(want, got)):
        assert n == len(ev) and len(data) == n * rec, k
        assert data.tobytes() == O.raw_events(ev, channels), k
        assert np.array_equal(ch, gch), k
    hv.frames_set_format(False)
    with pytest.raises(A.AdderHipError):
        hv.frame_collect_wire()
    for f in clip[30:]:  # back to AdderEvents, same stream: the ring, then the blocking call
        pinned[0][...] = f.reshape(H, W * channels)
        hv.frame_submit(pinned[0])
        assert np.array_equal(hv.frame_collect(), ov.integrate_matrix(f))
    extra = clips.make_clip("runs", 2, H, W, channels, seed=24)
    for f in extra:
        assert np.array_equal(hv.integrate_matrix(f), ov.integrate_matrix(f))


@pytest.mark.gpu
def test_frame_ring_rules_and_overflow():
    A = _hip()
    clip = clips.make_clip("noise", 6, 20, 33, 1, seed=1)
    hv = A.HipVideo(33, 20, 1, delta_t_max=255)
    with pytest.raises(A.AdderHipError):
        hv.frame_collect()
    hv.frames_configure(2, 0)
    hv.frame_submit(clip[0])
    hv.frame_submit(clip[1])
    with pytest.raises(A.AdderHipError, match="in flight"):
        hv.frame_submit(clip[2])
    with pytest.raises(A.AdderHipError, match="in flight"):
        hv.integrate_batch(clip[2:4])
    with pytest.raises(A.AdderHipError, match="in flight"):
        hv.frames_configure(3, 0)
    a = hv.frame_collect()
    b = hv.frame_collect()
    ov = O.Video(33, 20, 1, delta_t_max=255)
    assert np.array_equal(a, ov.integrate_matrix(clip[0])) and np.array_equal(b, ov.integrate_matrix(clip[1]))
    got, _ = hv.integrate_batch(clip[2:4])  # the ring is empty again: other entry points work
    assert np.array_equal(got, np.concatenate([ov.integrate_matrix(f) for f in clip[2:4]]))
    # a slot that is too small: the frame fails at collect with the size it needed, and the context is poisoned
    hv.frames_configure(2, 50)
    hv.frame_submit(clip[4])
    with pytest.raises(A.AdderHipError) as ei:
        hv.frame_collect()
    assert ei.value.code == A.E_OUT_CAPACITY and hv.last_required == len(ov.integrate_matrix(clip[4])) > 50
    with pytest.raises(A.AdderHipError) as ei:
        hv.frame_submit(clip[5])
    assert ei.value.code == A.E_POISONED


def test_frame_ring_wire_format_overflow_and_format_guards():
    """The ring in its wire format (9 / 11-byte records in the slots): a frame that overflows a small slot is reported at
    collect with the size it needed -- the hand-over reads only what the expansion kept of it (round 4's kernel walked the
    uncapped count past the slot's event buffer) -- and a slot is collected in the format it was submitted in."""
    A = _hip()
    for Cn in (1, 3):
        clip = clips.make_clip("noise", 6, 40, 66, Cn, seed=2 + Cn)
        ov = O.Video(66, 40, Cn, delta_t_max=255)
        hv = A.HipVideo(66, 40, Cn, delta_t_max=255)
        hv.frames_set_format(True)
        hv.frames_configure(2, 0)
        hv.frame_submit(clip[0])
        hv.frame_submit(clip[1])
        with pytest.raises(A.AdderHipError, match="wire records"):   # the slot's own format decides, nothing is consumed
            hv.frame_collect()
        assert hv.frames_in_flight() == 2
        for k in range(2):
            data, n = hv.frame_collect_wire()
            want = ov.integrate_matrix(clip[k])
            assert n == len(want) and data.tobytes() == O.raw_events(want, Cn)
        hv.frames_configure(2, 64)   # 64 events per slot: every noise frame past the first overflows it
        hv.frame_submit(clip[2])
        need = len(ov.integrate_matrix(clip[2]))
        assert need > 1000
        with pytest.raises(A.AdderHipError) as ei:
            hv.frame_collect_wire()
        assert ei.value.code == A.E_OUT_CAPACITY and hv.last_required == need
        with pytest.raises(A.AdderHipError) as ei:
            hv.frame_submit(clip[3])
        assert ei.value.code == A.E_POISONED
        hv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["events", "wire", "per_event_records"])
def test_graph_instances_are_interchangeable_and_the_plan_settles(form):
    """A batch length of several chunks tries up to six instances of its captured graph on the first batches and keeps
    the fastest (include/adder_hip.h, adder_hip_launch_plan_settled): every one of those batches must produce the same
    stream, the choice must be made after fourteen of them (six instances x two batches, then the first -- the one-stream
    instance, measured on a cold chip -- twice more) AND NOT BEFORE TWELVE, whichever form the output takes (round 6: the
    tuner's report read the variant at a stale bit position and settled wire batches on their first candidate); batches of
    per-event records have one candidate and settle at once; reset / finish without host copies must keep working."""
    import torch
    A = _hip()
    W, H, T = 640, 360, 200  # 4 chunks (3 of 64 frames + one of 8)
    st = torch.cuda.current_stream().cuda_stream
    d_frames = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
    A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)
    torch.cuda.synchronize()
    d_ev = torch.empty((W * H * T, 3), dtype=torch.int32, device="cuda")
    d_wire = torch.empty(W * H * T * 9, dtype=torch.uint8, device="cuda") if form == "wire" else None
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    if form == "per_event_records":  # the bounded Collapse kernel (crf-3 numbers, delta_t_max 7650)
        hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=7650, c_thresh_start=2, c_counter_start=0)
        hv.set_crf_parameters(7, 7)
    else:
        hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
        hv.set_crf_parameters(0, 10)
    ref = None
    settled_at = None
    for k in range(17):
        hv.reset()
        if form == "wire":
            hv.integrate_wire_device(d_frames, d_wire, d_off, stream=st)
            n = hv.finish()
            digest = hashlib.sha256(d_wire[:n * 9].cpu().numpy().tobytes() + d_off.cpu().numpy().tobytes()).hexdigest()
        else:
            hv.integrate_device(d_frames, d_ev, d_off, stream=st)
            n = hv.finish()
            digest = hashlib.sha256(d_ev[:n].cpu().numpy().tobytes() + d_off.cpu().numpy().tobytes()).hexdigest()
        ref = ref or (n, digest)
        assert (n, digest) == ref, k
        if settled_at is None and hv.launch_plan_settled():
            settled_at = k
    if form == "per_event_records":
        assert settled_at is not None and settled_at <= 2, settled_at
        hv.close()
        return
    assert settled_at is not None and 11 <= settled_at <= 14, settled_at
    if form == "wire":
        hv.close()
        return
    clip = d_frames.cpu().numpy().reshape(T, H, W, 1)
    want, _ = _oracle_events(clip[:8], time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, dtm=255)
    offs = d_off.cpu().numpy()
    got = d_ev[: int(offs[8])].cpu().numpy().view(A.EVENT_DTYPE).reshape(-1)
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("multi_mode", [O.NORMAL, O.COLLAPSE])
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_sparse_steps_of_event_camera_sources(multi_mode, time_mode):
    """SURVEY 8(f)3, the sparse half (adder_hip_integrate_sparse: stable sort by pixel, a thread per pixel run with
    cont_step and per-pixel c_thresh / running_t, scan of the counts in step order, scatter): the flow of
    Prophesee::consume -- two dense start-up frames, then steps in the camera's order, hot pixels repeating inside a
    call -- against the oracle, plus the side plane, the refusal of dense frames afterwards, and reset."""
    from test_device_logic_cpu import _sparse_steps
    A = _hip()
    rng = np.random.default_rng(17 * multi_mode + time_mode)
    for (W, H, Cn, crf, nsteps) in ((37, 23, 1, (7, 7), 3000), (130, 9, 3, (0, 10), 20000)):
        ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=20, delta_t_max=40)
        ov.set_pixel_mode(1)
        hv = A.HipVideo(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=20, delta_t_max=40, max_depth=24,
                        pixel_mode=1)
        ov.ensure_capacity(30)
        for v in (ov, hv):
            v.set_crf_parameters(*crf)
        hv.enable_running_intensities(True)
        start = np.full((H, W, Cn), 128, np.uint8)
        for _ in range(2):
            assert np.array_equal(ov.integrate_matrix(start, time_spanned=20.0), hv.integrate_matrix(start, time_spanned=20.0))
        total = 0
        for k in range(4):
            st = _sparse_steps(rng, W, H, Cn, nsteps)
            a, b = ov.integrate_sparse(st), hv.integrate_sparse(st)
            assert len(a) == len(b) and np.array_equal(a, b), k
            total += len(a)
        assert total > nsteps
        assert np.array_equal(hv.running_intensities(), ov.running_intensities())
        with pytest.raises(A.AdderHipError, match="dense frames after sparse"):
            hv.integrate_matrix(start, time_spanned=20.0)
        bad = _sparse_steps(rng, W, H, Cn, 10)
        bad["x"][3] = W
        hv2 = A.HipVideo(W, H, Cn, pixel_mode=1, ref_time=20, delta_t_max=40)
        with pytest.raises(A.AdderHipError):
            hv2.integrate_sparse(bad)
        hv.reset()  # a fresh transcoder again: dense frames work, and so do sparse steps after them
        ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=20, delta_t_max=40)
        ov.set_pixel_mode(1)
        ov.ensure_capacity(30)
        ov.set_crf_parameters(*crf)
        assert np.array_equal(ov.integrate_matrix(start, time_spanned=20.0), hv.integrate_matrix(start, time_spanned=20.0))
        st = _sparse_steps(rng, W, H, Cn, 500)
        assert np.array_equal(ov.integrate_sparse(st), hv.integrate_sparse(st))
    fp = A.HipVideo(16, 16, 1)
    with pytest.raises(A.AdderHipError, match="Continuous"):
        fp.integrate_sparse(_sparse_steps(rng, 16, 16, 1, 5))


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_one_frame_per_launch_wide_kernel_ragged_planes(time_mode):
    """adder_lean1w_kernel (4 units per lane, a wave covers a pair of segments, frame-major parking): frame by
    frame through the per-frame call, on planes whose unit count is not a multiple of 4 / 128 / 256 (the input dword
    of the lane that straddles the band's end, padding units, a band of exactly 4 units; 3 units falls back to the
    2-unit kernel), gray and RGB, both time modes, with the oracle's chunk offsets."""
    for (W, H, Cn), kind in (((37, 23, 1), "noise"), ((61, 17, 3), "jitter"), ((130, 9, 1), "steps"),
                             ((259, 3, 1), "noise"), ((2, 2, 1), "noise"), ((1, 3, 1), "noise"),
                             ((640, 7, 1), "runs")):
        clip = clips.make_clip(kind, 14, H, W, Cn, seed=W * 131 + H)
        n = run_pair(clip, time_mode=time_mode, multi_mode=O.COLLAPSE, dtm=255, crf=CRFS[0])
        assert n > 0, (W, H, Cn)
    # ... and as one device batch stepped one frame per launch (graph replay + frame-major ring), a band of a plane
    A = _hip()
    clip = clips.make_clip("noise", 40, 45, 77, 1, seed=5)
    tm = A.TIME_DELTA_T if time_mode == O.DELTA_T else A.TIME_ABSOLUTE_T
    base, offs = _events_device(A, clip, time_mode=tm, multi_mode=A.MULTI_COLLAPSE, dtm=255, row_band=(3, 41))
    one, offs1 = _events_device(A, clip, time_mode=tm, multi_mode=A.MULTI_COLLAPSE, dtm=255, row_band=(3, 41),
                                frames_per_launch=1)
    assert np.array_equal(offs, offs1) and np.array_equal(base, one)
    ov = O.Video(77, 38, 1, row_begin=3, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    want = np.concatenate([ov.integrate_matrix(clip[k, 3:41]) for k in range(40)])
    assert np.array_equal(one, want)


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_blocked_kernel_input_paths_direct_to_lds_and_register_fallback(time_mode):
    """adder_lean_kernel parks the launch's input bytes in LDS either with global_load_lds_dwordx4 (whole, 16-byte
    aligned segments) or through registers eight frames at a time (anything else).  Both against the oracle, 70 frames
    (a launch of 64 and one of 6): an aligned plane of whole segments (direct), the same bytes 1 and 4 bytes into an
    allocation (fallback for every segment), a plane whose frames are 5250 bytes apart (misaligned from frame 1 on,
    last segment not whole), and a band of a plane (row_begin != 0)."""
    A = _hip()
    tm = A.TIME_DELTA_T if time_mode == O.DELTA_T else A.TIME_ABSOLUTE_T
    T = 70
    for (W, H), band, kind in (((256, 20), None, "noise"), ((250, 21), None, "jitter"), ((256, 40), (8, 24), "steps")):
        clip = clips.make_clip(kind, T, H, W, 1, seed=W + H)
        y0, y1 = (0, H) if band is None else band
        ov = O.Video(W, y1 - y0, 1, row_begin=y0, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255,
                     delta_t_max=255)
        ov.set_crf_parameters(0, 10)
        ov.reset_c_thresh(0)
        per_frame = [ov.integrate_matrix(clip[k, y0:y1]) for k in range(T)]
        want = np.concatenate(per_frame)
        want_offs = np.concatenate([[0], np.cumsum([len(e) for e in per_frame])])
        for mis in (0, 1, 4):
            got, offs = _events_device(A, clip, time_mode=tm, multi_mode=A.MULTI_COLLAPSE, dtm=255, row_band=band,
                                       misalign=mis)
            assert np.array_equal(offs.astype(np.int64), want_offs), (W, H, mis)
            assert np.array_equal(got, want), (W, H, mis)


# ---- the bounded Collapse kernel (adder_cb_kernel: Collapse with delta_t_max > time_spanned, the reference's defaults) ----
def _cb_pair(W, H, Cn, tm, dtm, *, ref_time=255, crf=CRFS[0], max_depth=20):
    A = _hip()
    ov = O.Video(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm)
    hv = A.HipVideo(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm, max_depth=max_depth)
    ov.ensure_capacity(max_depth + 2)
    base, cmax, vel = crf
    for v in (ov, hv):
        v.set_crf_parameters(cmax, vel)
        v.reset_c_thresh(base)
    return ov, hv


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("crf", [0, 3])
def test_cb_kernel_batches_of_every_shape(time_mode, crf):
    """The default mode in batches whose lengths meet the frame-30 pop, the flushes and the chunk edges differently:
    blocked launches of 64 frames, short tails, single frames (the per-frame contract) -- the levels stay in LDS in
    prefix coordinates inside a launch and return to the deep planes between launches."""
    rng = np.random.default_rng(50 + crf + time_mode)
    W, H, frames = 333, 41, 330  # ragged: 13 653 units, the last segment partly padding
    for kind in ("scene", "runs", "jitter"):
        clip = (O.synth_clip(O.CONTENT_SCENE, W, H, 1, frames) if kind == "scene"
                else clips.make_clip(kind, frames, H, W, 1, seed=crf * 7 + len(kind)))
        ov, hv = _cb_pair(W, H, 1, time_mode, 7650, crf=CRFS[crf])
        k, total = 0, 0
        while k < frames:
            nb = min(int(rng.choice([1, 2, 29, 31, 64, 65, 130])), frames - k)
            want = [ov.integrate_matrix(clip[k + i]) for i in range(nb)]
            got, offs = hv.integrate_batch(clip[k:k + nb])
            assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == [len(w) for w in want], (kind, k, nb)
            assert np.array_equal(got, np.concatenate(want)), (kind, k, nb)
            total += len(got)
            k += nb
        assert total > 0
        hv.close()


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("depth", [5, 16, 64])
def test_cb_kernel_quiet_waves_hand_over_to_the_general_loop_and_back(time_mode, depth):
    """The quiet-frame loop of adder_cb_kernel (cb_quiet / cb_step_quiet): whole segments popped down to their roots
    for hundreds of frames, black rows (d = 128 roots that never pop), a flush inside ONE segment in the middle of a
    launch (that wave leaves the loop, the others stay), jitter inside the contrast band, launches of 5 / 16 / 64 frames so
    that the hand-over back at an input-group boundary happens at every phase -- against the oracle, state included (the
    last frames flush everything the quiet frames accumulated)."""
    W, H, frames = 256, 12, 260      # 3072 units = 24 whole segments: no padding keeps a wave out of the loop
    rng = np.random.default_rng(9 + depth + time_mode)
    base = rng.integers(0, 256, (1, H, W, 1))
    base[0, :2] = 0                                        # four black segments
    clip = np.repeat(base, frames, axis=0)
    clip[:, 4:] = np.clip(clip[:, 4:] + rng.integers(-1, 2, (frames, H - 4, W, 1)), 0, 255)
    clip[77:, 6, 10:40] = 255 - clip[77:, 6, 10:40]        # one segment flushes at frame 77 ...
    clip[141:, 1, 200:210] = 9                             # ... a black one wakes up at 141 ...
    clip[250:] = rng.integers(0, 256, (10, H, W, 1))       # ... and everything flushes at the end
    clip = clip.astype(np.uint8)
    for crf in (3, 0):
        ov, hv = _cb_pair(W, H, 1, time_mode, 7650, crf=CRFS[crf])
        hv.set_frames_per_launch(depth)
        k = 0
        while k < frames:
            nb = min(int(rng.choice([1, 37, 64, 100])), frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            got, _ = hv.integrate_batch(clip[k:k + nb])
            assert len(got) == len(want) and np.array_equal(got, want), (crf, k, nb)
            k += nb
        hv.close()


def test_cb_kernel_deep_levels_spill_to_the_deep_planes_and_depth_is_reported():
    """delta_t_max of 500 frames: static pixels reach seven levels before the pop, past the four LDS slots, so levels 5+
    are stepped in the deep planes; flushes drain them.  With max_depth 3 the same clip must fail with
    ADDER_E_ARENA_DEPTH instead of overrunning."""
    A = _hip()
    frames = 560
    clip = clips.make_clip("static", frames, 9, 70, 1, seed=4)
    clip[300:] = 255 - clip[300:]
    clip[520:] = clip[0]
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, hv = _cb_pair(70, 9, 1, tm, 255 * 500)
        for k in range(0, frames, 70):
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(70)])
            got, _ = hv.integrate_batch(clip[k:k + 70])
            assert np.array_equal(got, want), (tm, k)
        hv.close()
    _, hv = _cb_pair(70, 9, 1, O.DELTA_T, 255 * 500, max_depth=3)
    with pytest.raises(A.AdderHipError, match="max_depth"):
        hv.integrate_batch(clip[:40])
    hv.close()


def test_cb_kernel_rgb_bands_other_rates_and_fractional_time_fallback():
    A = _hip()
    clip = clips.make_clip("runs", 96, 24, 50, 3, seed=12)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        for ref_time, dtm in ((255, 7650), (5000, 240000), (20, 10000)):
            ov, hv = _cb_pair(50, 24, 3, tm, dtm, ref_time=ref_time, crf=CRFS[3])
            want = np.concatenate([ov.integrate_matrix(f, time_spanned=float(ref_time)) for f in clip])
            got, _ = hv.integrate_batch(clip, time_spanned=float(ref_time))
            assert np.array_equal(got, want), (tm, ref_time)
            hv.close()
    # a fractional time_spanned has no exact prefix sums: those batches (and every later one) take the generic kernel
    ov, hv = _cb_pair(50, 24, 3, O.ABSOLUTE_T, 7650, crf=CRFS[3])
    spans = [255.0] * 40 + [254.5] * 16 + [255.0] * 40
    k = 0
    for span, n in ((255.0, 40), (254.5, 16), (255.0, 40)):
        want = np.concatenate([ov.integrate_matrix(f, time_spanned=span) for f in clip[k:k + n]])
        got, _ = hv.integrate_batch(clip[k:k + n], time_spanned=span)
        assert np.array_equal(got, want), (span, k)
        k += n
    hv.close()


# ---- the default mode at crf 0: three frame kernels for the same regime (run records, constant runs, bounded Collapse) ----
_STEP_ENV = {"rr": {}, "cr": {"ADDER_HIP_NO_RR": "1"}, "cb": {"ADDER_HIP_NO_RR": "1", "ADDER_HIP_NO_CR": "1"}}


def _use_step(monkeypatch, step):
    for k in ("ADDER_HIP_NO_RR", "ADDER_HIP_NO_CR"):
        monkeypatch.delenv(k, raising=False)
    for k, v in _STEP_ENV[step].items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize("step", ["rr", "cr", "cb"])
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_default_mode_crf0_every_step_kernel(monkeypatch, step, time_mode):
    """adder_rr_kernel (integer state, one record per flush / pop, the events worked out by the expansion), adder_cr_kernel
    and adder_cb_kernel on the same ragged clips in batches of every shape: every event equals the oracle's."""
    _use_step(monkeypatch, step)
    rng = np.random.default_rng(70 + time_mode)
    W, H, frames = 333, 41, 330
    for kind in ("scene", "runs", "jitter", "dark", "noise"):
        clip = (O.synth_clip(O.CONTENT_SCENE, W, H, 1, frames) if kind == "scene"
                else clips.make_clip(kind, frames, H, W, 1, seed=11 + len(kind)))
        ov, hv = _cb_pair(W, H, 1, time_mode, 7650, crf=CRFS[0])
        k, total = 0, 0
        while k < frames:
            nb = min(int(rng.choice([1, 2, 29, 31, 64, 65, 130])), frames - k)
            want = [ov.integrate_matrix(clip[k + i]) for i in range(nb)]
            got, offs = hv.integrate_batch(clip[k:k + nb])
            assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == [len(w) for w in want], (kind, k, nb)
            assert np.array_equal(got, np.concatenate(want)), (kind, k, nb)
            total += len(got)
            k += nb
        assert total > 0
        hv.close()


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_default_mode_crf0_step_kernels_interleave_in_one_stream(monkeypatch, time_mode):
    """All three keep the planes in the same resident form: any of them may take the next batch of a stream (rgb, a band
    of the plane, 1 .. 100 frames per batch)."""
    rng = np.random.default_rng(5 + time_mode)
    clip = clips.make_clip("runs", 300, 24, 50, 3, seed=13)
    A = _hip()
    ov = O.Video(50, 24, 3, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=7650)
    hv = A.HipVideo(50, 24, 3, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=7650)
    ov.ensure_capacity(22)
    for v in (ov, hv):
        v.set_crf_parameters(CRFS[0][1], CRFS[0][2])
        v.reset_c_thresh(CRFS[0][0])
    k, used = 0, set()
    while k < len(clip):
        step = ("rr", "cr", "cb")[int(rng.integers(0, 3))]
        _use_step(monkeypatch, step)
        used.add(step)
        nb = min(int(rng.choice([1, 3, 17, 64, 100])), len(clip) - k)
        want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
        got, _ = hv.integrate_batch(clip[k:k + nb])
        assert np.array_equal(got, want), (k, nb, step)
        k += nb
    assert used == {"rr", "cr", "cb"}
    hv.close()


@pytest.mark.parametrize("step", ["rr", "generic"])
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_mode_normal_crf0_run_records_and_generic_kernel(monkeypatch, step, time_mode):
    """Mode Normal at crf 0 with delta_t_max > time_spanned runs adder_rr_kernel too (a popped arena is the arena of a
    shorter run; a flush after the pop emits the whole chain); ADDER_HIP_NO_RR=1 keeps adder_frame_kernel.  Ragged plane,
    rgb, batches of every shape, and the two kernels taking turns on one stream."""
    rng = np.random.default_rng(90 + time_mode)
    A = _hip()
    W, H, Cn, frames = 131, 23, 3, 200
    for kind, dtm in (("runs", 7650), ("scene", 255 * 4), ("jitter", 7650), ("runs", 255), ("scene", 255)):
        clip = (O.synth_clip(O.CONTENT_SCENE, W, H, Cn, frames) if kind == "scene"
                else clips.make_clip(kind, frames, H, W, Cn, seed=17 + len(kind)))
        ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=O.NORMAL, ref_time=255, delta_t_max=dtm)
        hv = A.HipVideo(W, H, Cn, time_mode=time_mode, multi_mode=O.NORMAL, ref_time=255, delta_t_max=dtm, max_depth=24)
        ov.ensure_capacity(26)
        for v in (ov, hv):
            v.set_crf_parameters(0, 10)
            v.reset_c_thresh(0)
        k = 0
        while k < frames:
            if step == "rr" and rng.integers(0, 4) != 0:
                monkeypatch.delenv("ADDER_HIP_NO_RR", raising=False)
            else:
                monkeypatch.setenv("ADDER_HIP_NO_RR", "1")
            nb = min(int(rng.choice([1, 2, 29, 31, 64, 65])), frames - k)
            want = [ov.integrate_matrix(clip[k + i]) for i in range(nb)]
            got, offs = hv.integrate_batch(clip[k:k + nb])
            assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == [len(w) for w in want], (kind, k, nb)
            assert np.array_equal(got, np.concatenate(want)), (kind, k, nb)
            k += nb
        hv.close()


def test_run_records_long_runs_deep_chains_other_rates_and_depth():
    """delta_t_max of 500 frames: roots that have not fired for more than the table's 32 rows (the chain length is worked
    out), records of eight and more events, rounds of the expansion that outgrow its staging buffer; other tick rates (in
    AbsoluteT the step needs time_spanned == ref_time >= 255, else the constant-run kernel takes the batch); a max_depth
    below the chain is reported."""
    A = _hip()
    frames = 560
    clip = clips.make_clip("static", frames, 9, 70, 1, seed=4)
    clip[300:] = 255 - clip[300:]
    clip[520:] = clip[0]
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, hv = _cb_pair(70, 9, 1, tm, 255 * 500)
        for k in range(0, frames, 70):
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(70)])
            got, _ = hv.integrate_batch(clip[k:k + 70])
            assert np.array_equal(got, want), (tm, k)
        hv.close()
    _, hv = _cb_pair(70, 9, 1, O.DELTA_T, 255 * 500, max_depth=3)
    with pytest.raises(A.AdderHipError, match="max_depth"):
        hv.integrate_batch(clip[:40])
    hv.close()
    clip = clips.make_clip("runs", 96, 24, 50, 3, seed=12)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        for ref_time, dtm in ((255, 7650), (5000, 240000), (20, 10000), (255, 510)):
            ov, hv = _cb_pair(50, 24, 3, tm, dtm, ref_time=ref_time)
            want = np.concatenate([ov.integrate_matrix(f, time_spanned=float(ref_time)) for f in clip])
            got, _ = hv.integrate_batch(clip, time_spanned=float(ref_time))
            assert np.array_equal(got, want), (tm, ref_time)
            hv.close()


def test_frame_ring_warm_submits_return_quickly():
    """framed.rs:127-157 calls integrate_matrix once per decoded frame: adder_hip_frame_submit only QUEUES a frame
    (upload, kernels, hand-over) and must come back at once.  The first submits of a process pay for the slots'
    buffers and for the HIP runtime's own pools (one call of several ms some 90 submits in, tools/ring_probe.py); after
    that no submit may take as long as 1 ms and three frames in flight must sustain well under the blocking call."""
    import ctypes as Ct
    import time
    A = _hip()
    W, H = 1920, 1080
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, 1, 8)
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255,
                    c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    pinned = [hv.pinned_frame() for _ in range(8)]
    for k in range(8):
        pinned[k][...] = clip[k].reshape(H, W)
    L = hv.L
    ev_p, n_p, ch_p = Ct.c_void_p(), Ct.c_size_t(0), Ct.c_void_p()

    def cycle(n):
        sub = []
        t0 = time.perf_counter()
        for k in range(n):
            if L.adder_hip_frames_in_flight(hv.h) == 3:
                assert L.adder_hip_frame_collect(hv.h, Ct.byref(ev_p), Ct.byref(n_p), Ct.byref(ch_p)) == 0
            t1 = time.perf_counter()
            assert L.adder_hip_frame_submit(hv.h, pinned[k % 8].ctypes.data, W, 255.0) == 0
            sub.append(time.perf_counter() - t1)
        while L.adder_hip_frames_in_flight(hv.h):
            assert L.adder_hip_frame_collect(hv.h, Ct.byref(ev_p), Ct.byref(n_p), Ct.byref(ch_p)) == 0
        return np.array(sub) * 1e6, (time.perf_counter() - t0) / n * 1e6

    cycle(160)  # warm: slot buffers, runtime pools
    hv.reset()
    sub, per_frame = cycle(96)
    assert sub.max() < 1000.0, sub.max()
    assert np.median(sub) < 200.0
    assert per_frame < 1000.0  # (217 us measured: the PCIe transfer of 7.5 MB of events; generous for a shared box)
    hv.close()


def test_model_fixtures_on_the_gpu(golden_dir):
    """The HIP path against the 216 known answers of the independent second restatement
    (tests/golden/make_model_fixtures.py): the lean, bounded Collapse and generic kernels in the modes no reference
    artefact pins, per-frame calls for half of the cases and one batch for the others."""
    import model_fixtures
    A = _hip()
    for k, cs in enumerate(model_fixtures.load(golden_dir)):
        T, H, W, Cn = cs["frames"].shape
        hv = A.HipVideo(W, H, Cn, time_mode=A.TIME_ABSOLUTE_T if cs["abs_t"] else A.TIME_DELTA_T,
                        multi_mode=A.MULTI_COLLAPSE if cs["collapse"] else A.MULTI_NORMAL, ref_time=cs["ref"],
                        delta_t_max=cs["dtm"], max_depth=24)
        hv.set_crf_parameters(cs["c_max"], cs["vel"])
        if (cs["c_start"], cs["ctr_start"]) != (10, 1):
            hv.reset_c_thresh(cs["c_start"])
        if k % 2:
            got, offs = hv.integrate_batch(cs["frames"], time_spanned=float(cs["ref"]))
            counts = [int(offs[i + 1] - offs[i]) for i in range(T)]
        else:
            per = [hv.integrate_matrix(f, time_spanned=float(cs["ref"])) for f in cs["frames"]]
            counts, got = [len(p) for p in per], np.concatenate(per)
        assert counts == list(cs["counts"]), k
        assert np.array_equal(got, cs["events"]), k
        hv.close()


@pytest.mark.parametrize("bands", [2, 3])
@pytest.mark.parametrize("channels", [1, 3])
def test_feature_mode_over_row_bands_matches_the_whole_plane(bands, channels):
    """SURVEY 8(f)4, the multi-GPU half: band contexts exchange a 3-row halo of the running intensities and the list of
    each frame's new features (whose reset squares reach into the neighbours' rows) -- sharding.FeatureBands drives
    adder_hip_feature_halo_export / _import / _detect / _apply.  Two and three bands on one device must produce the
    whole-plane context's events, feature set, per-pixel thresholds and new-feature counts, which equal the
    oracle's; a radius larger than a band and an ROI that straddles the band edges are part of it."""
    A = _hip()
    from adder_amd import sharding
    H, W = 61, 83
    clip = clips.make_clip("corners", 30, H, W, channels, seed=31 + bands)
    for radius, roi in ((3, None), (25, (5, 17, 70, 44))):
        ov, whole = _feature_pair(A, W, H, channels, multi_mode=O.COLLAPSE, dtm=7650, radius=radius, roi=roi)
        vids = []
        for y0, y1 in sharding.row_bands(H, bands):
            v = A.HipVideo(W, H, channels, row_begin=y0, row_end=y1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE,
                           ref_time=255, delta_t_max=7650, max_depth=30)
            v.set_crf_parameters(13, 4)
            v.reset_c_thresh(6)
            v.update_detect_features(True, True)
            v.set_feature_parameters(6, radius)
            v.update_roi(roi)
            vids.append(v)
        fb = sharding.FeatureBands(vids)
        total_new = 0
        for k in range(len(clip)):
            want = ov.integrate_matrix(clip[k])
            ref = whole.integrate_matrix(clip[k])
            got = fb.integrate_matrix(clip[k])
            assert np.array_equal(ref, want), k
            assert len(got) == len(want) and np.array_equal(got, want), (bands, radius, k)
            assert fb.new_features == len(ov.new_features()) == whole.last_new_features(), k
            total_new += fb.new_features
            if k % 5 == 4:
                cth = np.concatenate([v.c_thresh_plane().reshape(-1) for v in vids])
                assert np.array_equal(cth, ov.c_thresh_plane().reshape(-1)), k
                fset = np.concatenate([v.feature_set().reshape(-1) for v in vids])
                assert np.array_equal(fset, ov.feature_set().reshape(-1)), k
        assert total_new > 10
        for v in vids + [whole]:
            v.close()


def test_band_in_feature_mode_survives_a_too_small_event_buffer():
    """a row band in feature / ROI mode whose event buffer is too small is rolled back (ADDER_E_OUT_CAPACITY)
    and the documented retry -- the same frame with a larger buffer -- must be accepted: the pending-feature-step flag
    goes with the rollback.  The retried stream equals the whole-plane context's and the oracle's."""
    import torch
    A = _hip()
    from adder_amd import sharding
    from adder_amd import _native as N
    import ctypes as C
    H, W, channels, radius = 61, 83, 1, 3
    clip = clips.make_clip("corners", 12, H, W, channels, seed=77)
    ov, whole = _feature_pair(A, W, H, channels, multi_mode=O.COLLAPSE, dtm=7650, radius=radius, roi=None)
    vids = []
    for y0, y1 in sharding.row_bands(H, 2):
        v = A.HipVideo(W, H, channels, row_begin=y0, row_end=y1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE,
                       ref_time=255, delta_t_max=7650, max_depth=30)
        v.set_crf_parameters(13, 4)
        v.reset_c_thresh(6)
        v.update_detect_features(True, True)
        v.set_feature_parameters(6, radius)
        vids.append(v)
    fb = sharding.FeatureBands(vids)
    st = torch.cuda.current_stream().cuda_stream
    retried = 0
    for k in range(len(clip)):
        want = ov.integrate_matrix(clip[k])
        n_band0 = int(np.count_nonzero(want["y"] < vids[0].row_end))
        if n_band0 > 2 and retried < 2 and k >= 2:  # band 0: a device call with room for 2 events, then the retry
            v = vids[0]
            d_fr = torch.from_numpy(np.ascontiguousarray(clip[k, v.row_begin:v.row_end]).reshape(1, -1)).cuda()
            d_off = torch.zeros(2, dtype=torch.int64, device="cuda")
            tiny = torch.empty((2, 3), dtype=torch.int32, device="cuda")
            v.integrate_device(d_fr, tiny, d_off, stream=st)
            with pytest.raises(A.AdderHipError) as ei:
                v.finish()
            assert ei.value.code == A.E_OUT_CAPACITY
            retried += 1
        got = fb.integrate_matrix(clip[k])  # (band 0's frame k again: must not be refused by band_precheck)
        assert len(got) == len(want) and np.array_equal(got, want), k
        assert fb.new_features == len(ov.new_features()), k
    assert retried == 2
    for v in vids + [whole]:
        v.close()


def _wire_pair(W, H, Cn, tm, dtm=255):
    A = _hip()
    ov = O.Video(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
    hv = A.HipVideo(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
    for v in (ov, hv):
        v.set_crf_parameters(0, 10)
        v.reset_c_thresh(0)
    return ov, hv


@pytest.mark.parametrize("channels,frames", [(1, 70), (3, 20)])
def test_full_size_1080p_wire_records_of_the_headline_entry_point(channels, frames):
    """adder_hip_integrate_wire_device at BASELINE config 2 / 3's size (1920x1080 gray x 70 frames across the chunk boundary,
    1920x1080 RGB x 20 frames), scene content: the expansion's 9 / 11-byte records == the oracle's raw sink over the oracle's
    events (raw/stream.rs:79-120), the offsets == the oracle's counts.  Frames start at every stream index mod 4 (the
    flush's heads and tails at the waves' edges, the kilobyte-per-instruction body, both record sizes); a second pass cuts
    the clip into batches whose sizes leave every residue at their ends."""
    import torch
    W, H = 1920, 1080
    rec = 9 if channels == 1 else 11
    st = torch.cuda.current_stream().cuda_stream
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, channels, frames)
    for tm, cuts in ((O.DELTA_T, [frames]), (O.ABSOLUTE_T, [1, 2, 3, frames - 6])):
        ov, hv = _wire_pair(W, H, channels, tm)
        k, residues = 0, set()
        for nb in cuts:
            want = [ov.integrate_matrix(f) for f in clip[k:k + nb]]
            counts = [len(w) for w in want]
            n = sum(counts)
            starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
            residues |= {int(s) % 4 for s, c in zip(starts, counts) if c}
            d_frames = torch.from_numpy(np.ascontiguousarray(clip[k:k + nb]).reshape(nb, -1)).cuda()
            d_wire = torch.full((n * rec + 64,), 0xAB, dtype=torch.uint8, device="cuda")
            d_offs = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
            hv.integrate_wire_device(d_frames, d_wire, d_offs, stream=st)
            assert hv.finish() == n
            offs = d_offs.cpu().numpy()
            assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == counts, (tm, k)
            got = d_wire.cpu().numpy()
            assert got[:n * rec].tobytes() == O.raw_events(np.concatenate(want), channels), (tm, k, nb)
            assert (got[n * rec:] == 0xAB).all()
            k += nb
        if len(cuts) == 1:
            assert residues == {0, 1, 2, 3}, residues   # frames of the one batch start at every residue
        hv.close()


def test_config_2_exactly_as_benchmarked_300_frames_wire_sha256():
    """BASELINE config 2 at its STATED length, through the entry point bench.py times: 1920x1080 gray, 300 frames of the scene
    clip in ONE adder_hip_integrate_wire_device batch from a fresh transcoder (Collapse, DeltaT, delta_t_max 255, crf-0
    numbers).  Every frame's count and the sha256 of the whole stream's 9-byte records == the oracle's raw sink over the
    oracle's 300 frames (video.rs:651-778, raw/stream.rs:101-120) -- frames 65-300 included, which bench.py's own check only
    compares GPU against GPU."""
    import torch
    A = _hip()
    W, H, T = 1920, 1080, 300
    st = torch.cuda.current_stream().cuda_stream
    ov, hv = _wire_pair(W, H, 1, O.DELTA_T)
    d_frames = torch.empty((T, W * H), dtype=torch.uint8, device="cuda")
    A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, num_frames=T, stream=st)   # (the clip bench.py times)
    torch.cuda.synchronize()   # (the generator runs on the stream given, the batch on the context's own when that is the null stream)
    d_wire = torch.empty(int(W * H * T * 0.4) * 9, dtype=torch.uint8, device="cuda")
    d_offs = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    hv.integrate_wire_device(d_frames, d_wire, d_offs, stream=st)
    n = hv.finish()
    assert hv.last_batch_kernel() == A.KERNEL_LEAN_RUNS_PACKED
    offs = d_offs.cpu().numpy()
    h_want, h_got, total = hashlib.sha256(), hashlib.sha256(), 0
    for k in range(0, T, 20):                    # (20 frames of events at a time: the whole stream is 1.7 GB)
        clip = O.synth_clip(O.CONTENT_SCENE, W, H, 1, 20, k0=k)
        assert k != 0 or np.array_equal(clip[0].reshape(-1), d_frames[0].cpu().numpy())
        want = [ov.integrate_matrix(f) for f in clip]
        assert [int(offs[k + i + 1] - offs[k + i]) for i in range(20)] == [len(w) for w in want], k
        h_want.update(O.raw_events(np.concatenate(want), 1))
        h_got.update(d_wire[int(offs[k]) * 9:int(offs[k + 20]) * 9].cpu().numpy().tobytes())
        total += sum(len(w) for w in want)
    assert n == total and int(offs[T]) == total
    assert h_got.hexdigest() == h_want.hexdigest()
    hv.close()


def test_config_4_one_band_at_its_stated_1200_frames():
    """One of BASELINE config 4's eight row bands -- rows 270..539 of the 3840x2160 plane -- over the config's 1 200 frames
    (Collapse, DeltaT, delta_t_max 255, crf 0), in the four 300-frame batches a rank of the bench integrates, wire records:
    counts and bytes == the oracle's band (row_begin = 270: the events carry plane coordinates)."""
    import torch
    A = _hip()
    W, H, T, y0, y1 = 3840, 2160, 1200, 270, 540
    st = torch.cuda.current_stream().cuda_stream
    ov = O.Video(W, y1 - y0, 1, row_begin=y0, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
    ov.ensure_capacity(8)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    hv = A.HipVideo(W, H, 1, row_begin=y0, row_end=y1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, ref_time=255,
                    delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    units = W * (y1 - y0)
    nb = 300
    d_frames = torch.empty((nb, units), dtype=torch.uint8, device="cuda")
    d_wire = torch.empty(int(units * nb * 0.45) * 9, dtype=torch.uint8, device="cuda")
    d_offs = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
    for k0 in range(0, T, nb):
        A.synth_clip_device(d_frames, A.CONTENT_SCENE, W, H, 1, row_begin=y0, rows=y1 - y0, frame_begin=k0, num_frames=nb, stream=st)
        torch.cuda.synchronize()
        hv.integrate_wire_device(d_frames, d_wire, d_offs, stream=st)
        n = hv.finish()
        offs = d_offs.cpu().numpy()
        h_want, h_got, total = hashlib.sha256(), hashlib.sha256(), 0
        for k in range(0, nb, 30):
            clip = O.synth_clip(O.CONTENT_SCENE, W, H, 1, 30, y0=y0, rows=y1 - y0, k0=k0 + k)
            want = [ov.integrate_matrix(f) for f in clip]
            assert [int(offs[k + i + 1] - offs[k + i]) for i in range(30)] == [len(w) for w in want], (k0, k)
            h_want.update(O.raw_events(np.concatenate(want), 1))
            h_got.update(d_wire[int(offs[k]) * 9:int(offs[k + 30]) * 9].cpu().numpy().tobytes())
            total += sum(len(w) for w in want)
        assert n == total and h_got.hexdigest() == h_want.hexdigest(), k0
    hv.close()


def test_config_1_adder_file_through_the_wire_path():
    """BASELINE config 1 (640x480 gray, 30 frames, raw .adder out) as a whole FILE: header + the expansion's wire records +
    EOF == the oracle's header, raw sink and EOF over the oracle's events (encoder.rs:170-229, raw/stream.rs:79-120)."""
    import torch
    A = _hip()
    W, H, T = 640, 480, 30
    st = torch.cuda.current_stream().cuda_stream
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, 1, T)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, hv = _wire_pair(W, H, 1, tm)
        want = np.concatenate([ov.integrate_matrix(f) for f in clip])
        oracle_file = O.raw_header(3, W, H, 1, 255 * 30, 255, 255, 0, tm, 0) + O.raw_events(want, 1) + O.raw_eof()
        d_frames = torch.from_numpy(clip.reshape(T, -1)).cuda()
        d_wire = torch.zeros(len(want) * 9 + 16, dtype=torch.uint8, device="cuda")
        d_offs = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
        hv.integrate_wire_device(d_frames, d_wire, d_offs, stream=st)
        n = hv.finish()
        ours = A.raw_header(3, W, H, 1, 255 * 30, 255, 255, 0, tm, 0) + d_wire[:n * 9].cpu().numpy().tobytes() + A.raw_eof()
        assert n == len(want) and ours == oracle_file
        assert len(ours) == 37 + 9 * n + 11 and ours[:5] == b"adder"
        hv.close()


def test_integer_state_kernels_hand_over_past_65k_frames_since_the_reset():
    """The lean-runs / run-records kernels keep rho * 255 and rho * time_spanned exact in binary32: the host stops choosing
    them once (frames since the reset) * 255 would reach 2^24 -- 65 793 frames, 36 minutes of 30 fps video -- and the float
    kernels (adder_lean_kernel / adder_cr_kernel) take over ON THE SAME PLANES for the rest of the stream (slower, not
    different: DESIGN section 4).  A stream that crosses the limit, with runs tens of thousands of frames long across it,
    against the oracle -- the switch-over and the steady state behind it."""
    A = _hip()
    W, H, T = 128, 2, 66400
    rng = np.random.default_rng(8)
    clip = np.zeros((T, H, W, 1), np.uint8)
    cur = rng.integers(0, 256, (H, W, 1)).astype(np.uint8)
    cur[0, :8] = 0
    change_at = set(rng.integers(1, T, 400).tolist()) | {65000, 65790, 65793, 65794, 65800, 66000}
    for k in range(T):
        if k in change_at:
            m = rng.random((H, W, 1)) < 0.3
            cur = np.where(m, rng.integers(0, 256, (H, W, 1)).astype(np.uint8), cur)
        clip[k] = cur
    for tm, dtm in ((O.DELTA_T, 255), (O.ABSOLUTE_T, 7650)):
        ov = O.Video(W, H, 1, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
        hv = A.HipVideo(W, H, 1, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm, max_depth=24)
        ov.ensure_capacity(26)
        for v in (ov, hv):
            v.set_crf_parameters(0, 10)
            v.reset_c_thresh(0)
        k, total = 0, 0
        while k < T:
            nb = min(3700, T - k)
            want = np.concatenate([ov.integrate_matrix(f) for f in clip[k:k + nb]])
            got, offs = hv.integrate_batch(clip[k:k + nb])
            assert len(got) == len(want) and np.array_equal(got, want), (tm, k)
            total += len(got)
            k += nb
        assert total > 10000
        hv.close()


def test_integer_state_kernels_stay_past_65k_frames_while_the_runs_are_short():
    """The bound of the integer-state kernels is the longest RUN, not the stream's length: the kernels report the longest run
    their units hold (BatchResult::max_run) and the host's bound follows it -- a DeltaT stream whose pixels keep changing
    (every unit at least every 4 000 frames here) is still on adder_lr_kernel / adder_rr_kernel 72 000 frames after the
    reset; one static pixel in the plane sends it to the float kernels at 65 793 frames as before, and so does AbsoluteT
    (last_fired_t / T is an integer of the stream's length there).  All of it against the oracle."""
    A = _hip()
    W, H, T = 128, 2, 72000
    rng = np.random.default_rng(18)
    period = rng.integers(1500, 4000, (H, W, 1))
    phase = rng.integers(0, 4000, (H, W, 1))
    vals = rng.integers(0, 256, (64, H, W, 1)).astype(np.uint8)
    clip = np.empty((T, H, W, 1), np.uint8)
    idx = np.arange(H * W).reshape(H, W, 1)
    for k in range(T):
        step = (k + phase) // period            # every unit takes a new value every `period` frames of its own
        clip[k] = np.take_along_axis(vals, (step % 64)[None], axis=0)[0]
    cases = ((O.DELTA_T, 255, False, A.KERNEL_LEAN_RUNS_PACKED), (O.DELTA_T, 7650, False, A.KERNEL_RUN_RECORDS),
             (O.DELTA_T, 255, True, A.KERNEL_LEAN), (O.ABSOLUTE_T, 255, False, A.KERNEL_LEAN))
    for tm, dtm, one_static, last_kernel in cases:
        c = clip
        if one_static:
            c = clip.copy()
            c[:, 1, 77] = 131                    # one pixel that never changes: its run is the stream
        ov = O.Video(W, H, 1, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
        hv = A.HipVideo(W, H, 1, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm, max_depth=24)
        ov.ensure_capacity(26)
        for v in (ov, hv):
            v.set_crf_parameters(0, 10)
            v.reset_c_thresh(0)
        k, kernels = 0, []
        while k < T:
            nb = min(4096, T - k)
            want = np.concatenate([ov.integrate_matrix(f) for f in c[k:k + nb]])
            got, offs = hv.integrate_batch(c[k:k + nb])
            assert len(got) == len(want) and np.array_equal(got, want), (tm, dtm, one_static, k)
            kernels.append(hv.last_batch_kernel())
            k += nb
        integer = (A.KERNEL_LEAN_RUNS_PACKED if tm == O.DELTA_T else A.KERNEL_LEAN_RUNS) if dtm == 255 else A.KERNEL_RUN_RECORDS
        assert kernels[0] == integer and kernels[-1] == last_kernel, (tm, dtm, one_static, kernels)
        if last_kernel != integer:               # the switch happens where frames * 255 reaches 2^24, not before
            assert kernels[65793 // 4096 - 1] == integer, kernels
        hv.close()


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_kernel_switch_points_random_walk_with_the_kernel_asserted(monkeypatch, time_mode):
    """The step kernels are chosen per batch by host-side properties of the stream ("constant runs since the reset", the
    window, the ramp): every switch point is a correctness surface.  A random walk over all the kernels a stream can meet,
    batch by batch, with the kernel that ran ASSERTED (adder_hip_last_batch_kernel) -- the lean regime (lean runs, the lean
    step, the one-frame kernels), the default mode (run records, constant runs, bounded Collapse; then a mid-stream
    update_crf, after which only the bounded step may run), and a window that grows and shrinks mid-stream (lean -> bounded
    Collapse -> the generic step, for good) -- against the oracle, with a rollback at some of the switch points."""
    A = _hip()
    rng = np.random.default_rng(401 + time_mode)
    W, H = 131, 21

    def pair(dtm, crf):
        ov = O.Video(W, H, 1, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
        hv = A.HipVideo(W, H, 1, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm, max_depth=24)
        ov.ensure_capacity(26)
        for v in (ov, hv):
            v.set_crf_parameters(crf[1], crf[2])
            v.reset_c_thresh(crf[0])
        return ov, hv

    def batch(ov, hv, clip, k, nb, expect):
        want = [ov.integrate_matrix(clip[k + i]) for i in range(nb)]
        need = sum(len(w) for w in want)
        if need > 4 and rng.random() < 0.25:   # a buffer that is too small first: the rollback lands on the switch point
            with pytest.raises(A.AdderHipError) as ei:
                hv.integrate_batch(clip[k:k + nb], out_cap=need - 1)
            assert ei.value.code == A.E_OUT_CAPACITY and hv.last_required == need
        got, offs = hv.integrate_batch(clip[k:k + nb], out_cap=max(need, 1))
        assert hv.last_batch_kernel() in expect, (k, nb, hv.last_batch_kernel(), expect)
        assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == [len(w) for w in want], (k, nb)
        assert np.array_equal(got, np.concatenate(want)), (k, nb)

    # ---- the lean regime at crf 0: lean runs <-> the lean step <-> the one-frame kernels ----
    clip = clips.make_clip("runs", 360, H, W, 1, seed=23)
    clip[200:260] = clips.make_clip("dark", 60, H, W, 1, seed=24)
    ov, hv = pair(255, CRFS[0])
    k, seen = 0, set()
    while k < len(clip):
        choice = ("lp", "lr", "lean", "one")[int(rng.integers(0, 4))]   # (lp: lean runs in packed bytes -- DeltaT only)
        nb = 1 if choice == "one" else min(int(rng.choice([2, 5, 16, 60, 64, 70])), len(clip) - k)
        monkeypatch.setenv("ADDER_HIP_NO_LR", "1" if choice == "lean" else "0")
        monkeypatch.setenv("ADDER_HIP_NO_LP", "1" if choice == "lr" else "0")
        packed = choice == "lp" and time_mode == O.DELTA_T
        expect = {A.KERNEL_LEAN_RUNS_PACKED if packed else A.KERNEL_LEAN_RUNS} if (choice in ("lp", "lr") and nb > 1) else {A.KERNEL_LEAN}
        batch(ov, hv, clip, k, nb, expect)
        seen.add((choice, nb > 1))
        k += nb
    assert {("lp", True), ("lr", True), ("lean", True), ("one", False)} <= seen
    monkeypatch.delenv("ADDER_HIP_NO_LR")
    monkeypatch.delenv("ADDER_HIP_NO_LP")
    hv.close()

    # ---- the default mode at crf 0: run records <-> constant runs <-> bounded Collapse; then update_crf mid-stream ----
    ov, hv = pair(7650, CRFS[0])
    names = {"rr": A.KERNEL_RUN_RECORDS, "cr": A.KERNEL_CONSTANT_RUNS, "cb": A.KERNEL_BOUNDED}
    k, seen = 0, set()
    while k < 240:
        step = ("rr", "cr", "cb")[int(rng.integers(0, 3))]
        _use_step(monkeypatch, step)
        nb = min(int(rng.choice([1, 3, 17, 64, 100])), 240 - k)
        batch(ov, hv, clip, k, nb, {names[step]})
        seen.add(step)
        k += nb
    assert seen == {"rr", "cr", "cb"}
    for v in (ov, hv):   # update_crf: the ramp is alive from here on, the constant-run property is gone for good
        v.set_crf_parameters(CRFS[3][1], CRFS[3][2])
        v.reset_c_thresh(CRFS[3][0])
    while k < len(clip):
        _use_step(monkeypatch, ("rr", "cr", "cb")[int(rng.integers(0, 3))])
        nb = min(int(rng.choice([1, 9, 64])), len(clip) - k)
        batch(ov, hv, clip, k, nb, {A.KERNEL_BOUNDED})
        k += nb
    _use_step(monkeypatch, "rr")
    hv.close()

    # ---- the window grows and shrinks mid-stream (update_quality_manual): lean runs -> bounded Collapse -> generic ----
    ov, hv = pair(255, CRFS[0])
    k = 0
    for dtm, expect, until in ((255, {A.KERNEL_LEAN_RUNS_PACKED, A.KERNEL_LEAN_RUNS, A.KERNEL_LEAN}, 90), (7650, {A.KERNEL_BOUNDED, A.KERNEL_CONSTANT_RUNS, A.KERNEL_RUN_RECORDS}, 230),
                               (255, {A.KERNEL_GENERIC}, len(clip))):
        for v in (ov, hv):
            v.set_delta_t_max(dtm)
        while k < until:
            nb = min(int(rng.choice([1, 4, 33, 64])), until - k)
            batch(ov, hv, clip, k, nb, expect)
            k += nb
    hv.close()
