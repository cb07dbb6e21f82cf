"""Pins the framer oracle (oracle/adder_framer_oracle.c) to what the reference's own tests hold:
  * get_frame_bytes_u8 / test_get_empty_frame   (adder-codec-rs/tests/integration_tests.rs:555-611, 782-820)
  * sample_3_{ordered,unordered}.adder -> sample_3.gray, 405 frames (:822-975)
  * the `dark` test: lake_scaled_hd_out.adder -> lake_scaled_out   (src/bin/adder_simulproc.rs:170-268)
CPU only."""
import gzip
import os

import numpy as np
import pytest

from oracle import oracle as O
import adder_stream_np as S


@pytest.fixture(scope="module")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_get_frame_bytes_u8():
    """integration_tests.rs:555-611: 25 events (d=5, t=5100) on a 5x5 plane, tpf = 50000/50 = 1000."""
    fr = O.Framer(5, 5, 1, chunk_rows=64, tps=50000, ref_interval=1000, delta_t_max=1000, output_fps=50.0,
                  codec_version=1, time_mode=O.DELTA_T, source_camera=O.FRAMED_U8)
    assert fr.tpf == 1000
    for i in range(5):
        for j in range(5):
            filled = fr.ingest_event(i, j, None, 5, 5100)
            assert filled == (i == 4 and j == 4)
        assert fr.is_frame_filled(0) == (i == 4)
    out = fr.write_multi_frame_bytes()
    assert len(out) == 150  # Ok(6) frames of 25 bytes
    # 2^5 / 5100 * 1000 = 6.27 -> 6 in every pixel of every frame
    assert set(out) == {6}


def test_get_empty_frame():
    """integration_tests.rs:782-820: an unfilled frame is still written (None -> 0)."""
    fr = O.Framer(5, 5, 1, chunk_rows=64, tps=50000, ref_interval=1000, delta_t_max=1000, output_fps=50.0,
                  codec_version=1, time_mode=O.DELTA_T, source_camera=O.FRAMED_U8)
    assert fr.write_frame_bytes() == bytes(25)
    assert fr.ingest_event(0, 0, None, 5, 500) is False


@pytest.mark.parametrize("name", ["sample_3_ordered.adder", "sample_3_unordered.adder"])
def test_sample_3(golden_dir, name):
    """integration_tests.rs:822-975: 405 frames, byte-identical to sample_3.gray."""
    meta, events, _ = S.read_adder(open(os.path.join(golden_dir, name), "rb").read())
    want = open(os.path.join(golden_dir, "sample_3.gray"), "rb").read()
    assert meta["tps"] // meta["ref_interval"] == 60
    fr = O.Framer(meta["width"], meta["height"], meta["channels"], chunk_rows=64, tps=meta["tps"],
                  ref_interval=meta["ref_interval"], delta_t_max=meta["delta_t_max"], output_fps=60.0,
                  codec_version=meta["version"], time_mode=O.DELTA_T, source_camera=meta["source_camera"])
    got = fr.ingest_events(events)
    assert len(got) // (meta["width"] * meta["height"]) == 405
    assert got == want


def test_dark_lake(golden_dir):
    """adder_simulproc.rs:170-268: the golden event file through SimulProcessor's framer
    (codec_version 1, TimeMode::default(), reconstructed rate = source fps, chunk_rows 1 ->
    ingest_events_events per input frame) equals lake_scaled_out."""
    raw = gzip.open(os.path.join(golden_dir, "lake_scaled_hd_out.adder.gz")).read()
    want = gzip.open(os.path.join(golden_dir, "lake_scaled_out.gz")).read()
    meta, events, _ = S.read_adder(raw)
    W, H = meta["width"], meta["height"]
    # source fps: tps = (255 * fps) as u32 = 6113 (video.rs / framed.rs:101) -> 24000/1001
    fps = np.float32(24000.0 / 1001.0)
    assert int(np.float32(255.0) * fps) == meta["tps"]
    fr = O.Framer(W, H, 1, chunk_rows=1, tps=meta["tps"], ref_interval=255, delta_t_max=meta["delta_t_max"],
                  output_fps=float(fps), codec_version=1, time_mode=O.ABSOLUTE_T, source_camera=O.FRAMED_U8)
    # per-input-frame Vec<Vec<Event>>: the frame boundaries of the golden stream are where the raster
    # order restarts; within a frame one inner Vec per row (chunk_rows = 1)
    key = events["y"].astype(np.int64) * W + events["x"]
    starts = np.concatenate([[0], np.nonzero(np.diff(key) < 0)[0] + 1, [len(events)]])
    got = b""
    for a, b in zip(starts[:-1], starts[1:]):
        seg = events[a:b]
        offs = np.searchsorted(seg["y"], np.arange(H + 1), side="left").astype(np.uint64)
        if fr.ingest_events_events(seg, offs):
            got += fr.write_multi_frame_bytes()
    assert len(want) % (W * H) == 0
    n = min(len(got), len(want))
    assert n > 0 and got[:n] == want[:n]
    assert len(got) >= len(want)  # "the file might be larger" (adder_simulproc.rs:255-257)


def test_fast_division_by_stream_constants_is_exact():
    """framer_step divides by tpf and ref_interval with a multiply-and-shift (adder_framer.hpp fast_div); it must be the
    hardware quotient for every u32 numerator -- edge numerators for a spread of divisors, plus random ones."""
    import ctypes as C
    import sim_py
    L = sim_py.lib()
    L.sim_fast_div_check.restype = C.c_uint64
    L.sim_fast_div_check.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(0)
    divisors = [1, 2, 3, 5, 7, 255, 256, 257, 1000, 5000, 7650, 65535, 65536, 65537, 2**31 - 1, 2**31, 2**31 + 1,
                2**32 - 2, 2**32 - 1] + [int(x) for x in rng.integers(1, 2**32, 200)]
    for d in divisors:
        ks = rng.integers(0, max(1, 2**32 // d), 2000, dtype=np.uint64)
        edge = np.concatenate([ks * d, ks * d + d - 1, np.minimum(ks * d + 1, 2**32 - 1)])
        n = np.concatenate([edge[edge < 2**32], rng.integers(0, 2**32, 4000, dtype=np.uint64),
                            np.array([0, 1, d - 1, d, 2**32 - 1], np.uint64)]).astype(np.uint32)
        n = np.ascontiguousarray(n)
        assert L.sim_fast_div_check(d, n.ctypes.data, len(n)) == 0, d


def _view_stream(rng, W, H, T, abs_t):
    """T raster-ordered segments: every pixel fires at least every few frames, with D_EMPTY fillers and repeats."""
    segs, clock = [], np.zeros(W * H, np.int64)
    for k in range(T):
        hit = np.flatnonzero(rng.random(W * H) < 0.6)
        runs = rng.integers(1, 4, len(hit))
        units = np.repeat(hit, runs)
        ev = np.zeros(len(units), O.EVENT_DTYPE)
        ev["c"], ev["x"], ev["y"] = 0xFF, units % W, units // W
        ev["d"] = rng.choice(np.array([0, 2, 5, 7, 9, 13, 20, 128, 255], np.uint8), len(units))
        dt = rng.choice(np.array([1, 40, 254, 255, 300, 900]), len(units))
        if abs_t:
            first = np.r_[True, units[1:] != units[:-1]]
            seg_start = np.maximum.accumulate(np.where(first, np.arange(len(units)), 0))
            cs = np.cumsum(dt)
            t = clock[units] + cs - (cs[seg_start] - dt[seg_start])
            np.maximum.at(clock, units, t)
            ev["t"] = t
        else:
            ev["t"] = dt
        segs.append(ev)
    return np.concatenate(segs)


@pytest.mark.parametrize("abs_t", [False, True])
@pytest.mark.parametrize("view,source,dmax", [(1, 0, 12.99), (1, 0, 0.0), (2, 0, 0.0), (3, 0, 0.0), (0, 1, 0.0),
                                              (0, 2, 0.0), (0, 3, 0.0)])
def test_view_modes_and_source_types_of_the_device_step_equal_the_oracle(abs_t, view, source, dmax):
    """The other arms of <u8 as FrameValue>::get_frame_value (scale_intensity.rs:73-109: D, DeltaT and SAE views,
    U16 / U32 / U64 sources): the device header's framer_value_u8 against the oracle's literal restatement.  No
    reference vector exists for them (its tests use the U8 Intensity view only), so this is oracle parity."""
    import sim_py
    rng = np.random.default_rng(view * 7 + source + 3 * abs_t)
    W, H, T = 9, 7, 40
    ev = _view_stream(rng, W, H, T, abs_t)
    fr = O.Framer(W, H, 1, chunk_rows=64, tps=7650, ref_interval=255, delta_t_max=2550, output_fps=30.0,
                  codec_version=3, time_mode=O.ABSOLUTE_T if abs_t else O.DELTA_T, source_camera=O.FRAMED_U8)
    fr.set_view(view, source, dmax)
    want = fr.ingest_events(ev)
    got = sim_py.framer_run(ev, W, H, 1, tpf=255, ref_interval=255, abs_t=abs_t, round_up=True, max_frames=1 << 14,
                            view_mode=view, source_type=source, practical_d_max=dmax, delta_t_max=2550)
    assert len(want) > 10 * W * H and got[: len(want)] == want
    if view or source:  # and the view really differs from the U8 intensity one
        fr0 = O.Framer(W, H, 1, chunk_rows=64, tps=7650, ref_interval=255, delta_t_max=2550, output_fps=30.0,
                       codec_version=3, time_mode=O.ABSOLUTE_T if abs_t else O.DELTA_T, source_camera=O.FRAMED_U8)
        assert fr0.ingest_events(ev) != want


@pytest.mark.parametrize("abs_t", [False, True])
@pytest.mark.parametrize("value_type", [1, 2])
@pytest.mark.parametrize("view,source,dmax", [(0, 0, 0.0), (0, 1, 0.0), (0, 2, 0.0), (0, 3, 0.0), (1, 0, 12.99), (2, 0, 0.0)])
def test_u16_u32_frame_elements_of_the_device_step_equal_the_oracle(abs_t, value_type, view, source, dmax):
    """<u16 / u32 as FrameValue>::get_frame_value (scale_intensity.rs:111-209): the device header's framer_value_wide
    against the oracle's restatement; the frames are the big-endian bincode bytes FrameSequence<T> writes
    (driver.rs:279,395-398).  No reference vector exists (its framer tests use u8 frames), so this is oracle parity --
    and the Intensity arm is checked below against the formula itself."""
    import sim_py
    rng = np.random.default_rng(100 + view * 7 + source + 3 * abs_t + 31 * value_type)
    W, H, T = 9, 7, 40
    ev = _view_stream(rng, W, H, T, abs_t)
    fr = O.Framer(W, H, 1, chunk_rows=64, tps=7650, ref_interval=255, delta_t_max=2550, output_fps=30.0,
                  codec_version=3, time_mode=O.ABSOLUTE_T if abs_t else O.DELTA_T, source_camera=O.FRAMED_U8)
    fr.set_view(view, source, dmax)
    fr.set_value_type(value_type)
    want = fr.ingest_events(ev)
    got = sim_py.framer_run(ev, W, H, 1, tpf=255, ref_interval=255, abs_t=abs_t, round_up=True, max_frames=1 << 14,
                            view_mode=view, source_type=source, practical_d_max=dmax, delta_t_max=2550,
                            value_type=value_type)
    elem = 1 << value_type
    assert len(want) > 10 * W * H * elem and len(want) % (W * H * elem) == 0 and got[: len(want)] == want
    vals = np.frombuffer(want, ">u2" if value_type == 1 else ">u4")
    assert len(np.unique(vals)) > 4 or source == 3  # not a degenerate image (a U64 source scales everything to 0)


@pytest.mark.parametrize("value_type", [1, 2])
@pytest.mark.parametrize("source", [0, 1, 2])
def test_u16_u32_intensity_values_follow_the_reference_formula(value_type, source):
    """One DeltaT event per pixel: frame 0 holds (2^d / delta_t [/ source max] * tpf [* T::MAX]) as T, the saturating
    truncating cast of Rust (scale_intensity.rs:126-137,175-186), computed here in float64 numpy."""
    W, H = 16, 8
    rng = np.random.default_rng(value_type * 5 + source)
    ev = np.zeros(W * H, O.EVENT_DTYPE)
    ev["c"], ev["x"], ev["y"] = 0xFF, np.arange(W * H) % W, np.arange(W * H) // W
    ev["d"] = rng.choice(np.array([0, 1, 3, 7, 8, 11, 16, 24, 31], np.uint8), W * H)
    ev["t"] = rng.integers(255, 2000, W * H)
    fr = O.Framer(W, H, 1, chunk_rows=64, tps=7650, ref_interval=255, delta_t_max=2550, output_fps=30.0,
                  codec_version=3, time_mode=O.DELTA_T, source_camera=O.FRAMED_U8)
    fr.set_view(0, source, 0.0)
    fr.set_value_type(value_type)
    out = fr.ingest_events(ev)
    elem, tmax = 1 << value_type, float((1 << (8 << value_type)) - 1)
    frame0 = np.frombuffer(out[: W * H * elem], ">u2" if value_type == 1 else ">u4").astype(np.float64)
    inten = 2.0 ** ev["d"].astype(np.float64) / ev["t"].astype(np.float64)
    smax = [255.0, 65535.0, 4294967295.0][source]
    x = inten * 255.0 if source == value_type else inten / smax * 255.0 * tmax
    want = np.clip(np.trunc(x), 0.0, tmax)
    assert np.array_equal(frame0, want)
    assert 0.0 < want.min() + want.max() and (want == tmax).any() == bool((x >= tmax).any())
