"""Pins the oracle to the reference's golden event file.

tests/golden/lake_scaled_hd_out.adder.gz is the file the reference checks in as
adder-codec-rs/tests/samples/lake_scaled_hd_out.adder (the event stream its
`dark` transcode test writes: adder_simulproc.rs:170-268 -> 200x50 gray, crf 0,
ref 255, dtm 6120, DeltaT, PixelMultiMode::Normal, raw).  The 110 input frames
were inverted from that file; replaying them must give
the same 201 620 events in the same order and the same bytes.
"""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
import adder_stream_np as S

GOLDEN_SHA = "b3ceb84fbef8c6f0f054c521b3d66fdda397219befc8201d6e3e1dd64cb80967"


def _golden(golden_dir):
    raw = gzip.open(os.path.join(golden_dir, "lake_scaled_hd_out.adder.gz")).read()
    assert hashlib.sha256(raw).hexdigest() == GOLDEN_SHA
    frames = np.load(os.path.join(golden_dir, "lake_scaled_hd_frames_reconstructed.npz"))["frames"]
    assert frames.shape == (110, 50, 200) and frames.dtype == np.uint8
    return raw, frames


def lake_events(frames, threads=1, chunk_rows=1):
    v = O.Video(200, 50, 1, time_mode=O.DELTA_T, multi_mode=O.NORMAL, ref_time=255, delta_t_max=6120,
                chunk_rows=chunk_rows, threads=threads)
    v.set_crf_parameters(0, 10)  # CRF[0] = (0, 0, 10): rate_controller.rs:9
    v.reset_c_thresh(0)          # .crf(0): video.rs:1241-1251
    out = []
    for f in frames:
        out.append(v.integrate_matrix(f))
    return out


def test_lake_golden_events_and_bytes(golden_dir):
    raw, frames = _golden(golden_dir)
    meta, gold_ev, closed = S.read_adder(raw)
    assert closed and len(gold_ev) == 201_620
    assert (meta["width"], meta["height"], meta["channels"]) == (200, 50, 1)
    assert (meta["tps"], meta["ref_interval"], meta["delta_t_max"]) == (6113, 255, 6120)
    assert (meta["version"], meta["time_mode"], meta["adu_interval"], meta["source_camera"]) == (3, 0, 0, 0)

    per_frame = lake_events(frames)
    assert len(per_frame[0]) == 0  # first consume emits nothing (SURVEY 8(a) note 1)
    ev = np.concatenate(per_frame)
    assert len(ev) == len(gold_ev)
    for f in ("x", "y", "c", "d", "t"):
        assert np.array_equal(ev[f], gold_ev[f]), f
    # bytes: oracle C serializer and the independent numpy writer both reproduce the file
    hdr = O.raw_header(3, 200, 50, 1, 6113, 255, 6120, 0, O.DELTA_T, 0)
    assert len(hdr) == 37
    blob = hdr + O.raw_events(ev, 1) + O.raw_eof()
    assert hashlib.sha256(blob).hexdigest() == GOLDEN_SHA
    assert S.write_adder(meta, ev) == raw


def test_lake_thread_and_chunk_invariance(golden_dir):
    _, frames = _golden(golden_dir)
    a = np.concatenate(lake_events(frames[:40], threads=1, chunk_rows=1))
    b = np.concatenate(lake_events(frames[:40], threads=4, chunk_rows=7))
    assert np.array_equal(a, b)


def test_second_opinion_vectors(golden_dir):
    """Model-derived (not reference-derived) vectors for the unpinned mode combinations."""
    d = json.load(open(os.path.join(golden_dir, "model_second_opinion_vectors.json")))
    for case in d["cases"]:
        W, H, Cn = case["width"], case["height"], case["channels"]
        v = O.Video(W, H, Cn, time_mode=case["time_mode"], multi_mode=case["multi_mode"],
                    ref_time=case["ref_time"], delta_t_max=case["delta_t_max"])
        v.set_crf_parameters(case["c_thresh_max"], case["c_increase_velocity"])
        if case["c_start"] is not None:
            v.reset_c_thresh(case["c_start"])
            assert case["c_counter_start"] == 0
        v.ensure_capacity(4)
        frames = np.array(case["input_hwc_u8"], dtype=np.uint8).reshape(case["frames"], H, W, Cn)
        got, counts = [], []
        for f in frames:
            e = v.integrate_matrix(f, ref_time=case["ref_time"])
            counts.append(len(e))
            got.append(e)
        got = np.concatenate(got)
        want = np.array(case["events"], dtype=np.int64).reshape(-1, 5)
        assert counts == case["events_per_frame"], case["name"]
        gc = got["c"].astype(np.int64)
        gc[gc == 0xFF] = -1
        have = np.stack([got["x"].astype(np.int64), got["y"].astype(np.int64), gc,
                         got["d"].astype(np.int64), got["t"].astype(np.int64)], axis=1)
        assert np.array_equal(have, want), case["name"]


def test_model_fixtures_pin_the_unpinned_modes(golden_dir):
    """216 randomised known answers from the independent second restatement (tests/golden/make_model_fixtures.py,
    written from SURVEY Appendix A alone): Collapse with delta_t_max = ref_time, Collapse + AbsoluteT at 30 frames,
    crf 3 / 6 / 9 numbers, RGB, construction-default pixels, other tick rates.  The C oracle must agree event for
    event, frame for frame."""
    import model_fixtures
    cases = model_fixtures.load(golden_dir)
    assert len(cases) >= 200
    seen = set()
    for k, cs in enumerate(cases):
        T, H, W, Cn = cs["frames"].shape
        v = O.Video(W, H, Cn, time_mode=O.ABSOLUTE_T if cs["abs_t"] else O.DELTA_T,
                    multi_mode=O.COLLAPSE if cs["collapse"] else O.NORMAL, ref_time=cs["ref"], delta_t_max=cs["dtm"])
        v.set_crf_parameters(cs["c_max"], cs["vel"])
        if (cs["c_start"], cs["ctr_start"]) != (10, 1):  # (10, 1) = PixelArena::new's own values
            assert cs["ctr_start"] == 0
            v.reset_c_thresh(cs["c_start"])
        v.ensure_capacity(24)
        got = [v.integrate_matrix(f, time_spanned=float(cs["ref"])) for f in cs["frames"]]
        assert [len(g) for g in got] == list(cs["counts"]), k
        assert np.array_equal(np.concatenate(got), cs["events"]), k
        seen.add((cs["collapse"], cs["abs_t"], cs["dtm"] // cs["ref"], cs["c_max"], Cn))
    assert len(seen) >= 12
