"""The C++ host mirror (adder-codec-rs_amd/host): same operator surface as the reference for this
path -- Framed::consume -> Video::integrate_matrix -> Encoder/RawOutput -- over the C-ABI."""
import gzip
import hashlib
import os

import numpy as np
import pytest

import adder_stream_np as S
import host_py as Hst


def test_crf_table_and_default_quality():
    # rate_controller.rs:5-21 ; feature radius = (CRF[q][3] * min_resolution) as u16
    assert Hst.crf_parameters(0, 200, 50)[:3] == (0, 0, 10)
    assert Hst.crf_parameters(3, 200, 50) == (2, 7, 7, int(np.float32(1.0 / 15.0) * np.float32(50)))
    assert Hst.crf_parameters(9, 1920, 1080)[:3] == (15, 25, 1)
    assert Hst.crf_parameters(-1, 64, 64) == Hst.crf_parameters(3, 64, 64)  # DEFAULT_CRF_QUALITY


def test_encoder_known_sizes():
    import adder_amd as A
    e = np.zeros(1, A.EVENT_DTYPE)
    assert len(Hst.encode_raw(3, 1, 1, 3, 1, 1, 1, 0, 1, e)) == 59          # encoder.rs:401-448 raw3
    assert len(Hst.encode_raw(0, 50, 100, 1, 53000, 4000, 50000, 0, 1, e[:0])) == 36
    assert len(Hst.encode_raw(1, 50, 100, 1, 53000, 4000, 50000, 0, 1, e[:0])) == 40
    assert len(Hst.encode_raw(2, 50, 100, 1, 53000, 4000, 50000, 0, 1, e[:0], close=False)) == 33


@pytest.mark.parametrize("name", ["sample_3_ordered.adder", "bunny_v2_dt.adder", "nyc_v1_1px.adder",
                                  "adder_info_test_sample.adder"])
def test_decoder_reads_reference_samples_and_encoder_rewrites_them(golden_dir, name):
    raw = open(os.path.join(golden_dir, name), "rb").read()
    meta, ev = Hst.decode_raw(raw)
    m2, ev2, closed = S.read_adder(raw)
    assert (int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3])) == (m2["version"], m2["width"], m2["height"], m2["channels"])
    assert np.array_equal(ev, ev2)
    blob = Hst.encode_raw(int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3]), int(meta[4]), int(meta[5]),
                          int(meta[6]), int(meta[8]) & 0xFF, int(meta[8]) >> 8, ev, close=closed)
    assert blob == raw[: len(blob)]


@pytest.mark.gpu
def test_framed_transcode_reproduces_reference_golden(golden_dir, tmp_path):
    """adder_simulproc.rs:170-268 `dark`: Framed (gray) .crf(0) .auto_time_parameters(255, 6120)
    .write_out(FramedU8, DeltaT, Normal, Raw, Crf::new(Some(0))) then consume() per frame -> the checked-in event file."""
    raw = gzip.open(os.path.join(golden_dir, "lake_scaled_hd_out.adder.gz")).read()
    frames = np.load(os.path.join(golden_dir, "lake_scaled_hd_frames_reconstructed.npz"))["frames"][:, :, :, None]
    out = str(tmp_path / "lake.adder")
    # fps chosen so that tps = (255 * fps) as u32 = 6113, the value in the golden's header
    fps = float(np.float32(6113.5 / 255.0))
    assert int(np.float32(255.0) * np.float32(fps)) == 6113
    n, chunks = Hst.transcode_raw(frames, fps=fps, crf=0, ref_time=255, delta_t_max=6120, time_mode=0,
                                  multi_mode=0, encoder_crf=0, out_path=out)
    assert n == 201_620 and chunks == 50  # chunk_rows = 1 -> one Vec<Event> per row
    blob = open(out, "rb").read()
    assert hashlib.sha256(blob).hexdigest() == "b3ceb84fbef8c6f0f054c521b3d66fdda397219befc8201d6e3e1dd64cb80967"
    assert blob == raw


@pytest.mark.gpu
def test_simulproc_dark_reproduces_both_reference_goldens(golden_dir, tmp_path):
    """The whole `dark` test (adder_simulproc.rs:170-268) through the C++ mirror: Framed -> Video -> Encoder
    AND SimulProcessor's framer; the event file equals lake_scaled_hd_out.adder and the reconstructed
    frames equal lake_scaled_out, which is exactly what the reference test compares."""
    raw = gzip.open(os.path.join(golden_dir, "lake_scaled_hd_out.adder.gz")).read()
    want_frames = gzip.open(os.path.join(golden_dir, "lake_scaled_out.gz")).read()
    frames = np.load(os.path.join(golden_dir, "lake_scaled_hd_frames_reconstructed.npz"))["frames"]
    ev_path, fr_path = str(tmp_path / "lake.adder"), str(tmp_path / "lake_frames")
    fps = float(np.float32(24000.0 / 1001.0))  # the clip's rate: tps = (255 * fps) as u32 = 6113
    n = Hst.simulproc(frames, fps=fps, crf=0, ref_time=255, delta_t_max=6120, time_mode=0, multi_mode=0,
                      out_events=ev_path, out_frames=fr_path)
    assert open(ev_path, "rb").read() == raw
    got = open(fr_path, "rb").read()
    assert n == len(got) // (200 * 50) and len(got) % (200 * 50) == 0
    # "the file might be larger ... should still pass if all the frames before that are identical" (:255-257)
    assert len(got) >= len(want_frames) and got[: len(want_frames)] == want_frames


@pytest.mark.gpu
def test_framed_color_to_gray_and_errors(tmp_path):
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (20, 9, 14, 3), dtype=np.uint8)
    out = str(tmp_path / "c.adder")
    n, _ = Hst.transcode_raw(frames, color_input=False, crf=0, ref_time=255, delta_t_max=510, time_mode=1,
                             multi_mode=1, out_path=out)
    # handle_color (utils/cv.rs:215-232) then the gray path; check against the oracle on the same gray frames
    gray = (frames[..., 0].astype(np.float64) * 0.114 + frames[..., 1].astype(np.float64) * 0.587
            + frames[..., 2].astype(np.float64) * 0.299).astype(np.uint8)
    v = O.Video(14, 9, 1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=510)
    v.set_crf_parameters(7, 7)  # write_out(EncoderOptions::default) -> quality-3 max/velocity
    v.reset_c_thresh(0)         # .crf(0) reset the pixels before that (SURVEY 8(a) note 6)
    want = np.concatenate([v.integrate_matrix(g) for g in gray])
    meta, ev = Hst.decode_raw(open(out, "rb").read())
    assert n == len(want) and np.array_equal(ev, want)
    with pytest.raises(RuntimeError, match="multiple of ref_time"):
        Hst.transcode_raw(frames, crf=0, ref_time=255, delta_t_max=600, out_path=out)


@pytest.mark.gpu
def test_framed_transcode_to_compressed_sink_config5_mode(tmp_path):
    """BASELINE configs[4]'s mode end to end through the C++ mirror on a small RGB plane: Framed (color) .crf(3)
    .auto_time_parameters(255, 7650) .write_out(FramedU8, AbsoluteT, Collapse, adu_interval = 30, Compressed) ->
    consume() per frame (GPU integration) -> Encoder::new_compressed (CPU sink).  The file must equal what the
    oracle's compressed restatement makes of the raw transcode's events, and decode back to them within the
    codec's tolerance."""
    import adder_amd as A
    import clips
    from oracle import compressed_oracle as CO
    T, H, W = 75, 36, 52
    frames = clips.make_clip("jitter", T, H, W, 3, seed=31)
    raw_path, cmp_path = str(tmp_path / "a.adder"), str(tmp_path / "a.addec")
    kw = dict(color_input=True, fps=30.0, crf=3, ref_time=255, delta_t_max=7650, time_mode=1, multi_mode=1)
    n_raw, _ = Hst.transcode_raw(frames, out_path=raw_path, **kw)
    n_cmp = Hst.transcode_compressed(frames, adu_interval=30, out_path=cmp_path, **kw)
    assert n_raw == n_cmp > 0
    meta, ev = Hst.decode_raw(open(raw_path, "rb").read())
    blob = open(cmp_path, "rb").read()
    assert blob[:5] == b"addec" and len(blob) < n_raw * 11
    co = CO.CompressedOutput(W, H, 3, tps=7650, ref_interval=255, delta_t_max=7650, adu_interval=30, time_mode=1,
                             c_thresh_max=7)
    for e in ev:
        co.ingest_event(int(e["x"]), int(e["y"]), int(e["c"]), int(e["d"]), int(e["t"]))
    assert blob == co.close()
    dec, p = A.compressed_decode(blob)
    assert (p.width, p.height, p.channels, p.adu_interval) == (W, H, 3, 30) and 0 < len(dec) <= len(ev)
    # the mirror's Decoder::new_compressed reads the file back: header fields and every event of the source
    meta2, ev2 = Hst.decode_raw(blob)  # meta: version, w, h, c, tps, ref, dtm, event_size, camera | time_mode << 8, adu
    assert (int(meta2[1]), int(meta2[2]), int(meta2[3]), int(meta2[9])) == (W, H, 3, 30)
    assert int(meta2[8]) >> 8 == 1 and np.array_equal(ev2, dec)


@pytest.mark.gpu
def test_framed_transcode_with_feature_rate_control_and_roi(tmp_path):
    """SURVEY 8(f)4 through the C++ mirror: Framed .crf(6) ... write_out(Raw), then update_detect_features(true, Off,
    true, false) and update_roi on the source's Video (what adder-viz's transcoder tab does), consume() per frame.
    The quality's own parameters drive the feedback: c_thresh_baseline 7 -> reset value 2, feature_c_radius =
    min_resolution / 25."""
    from oracle import oracle as O
    import adder_amd as A
    import clips
    T, H, W = 30, 50, 75
    frames = clips.make_clip("corners", T, H, W, 1, seed=6)
    for roi in (None, (30, 20, 60, 45)):
        out = str(tmp_path / "f.adder")
        n, fs = Hst.transcode_features(frames, crf=6, ref_time=255, delta_t_max=7650, time_mode=1, multi_mode=1,
                                       chunk_rows=4, detect=True, rate_adjustment=True, roi=roi, out_path=out)
        v = O.Video(W, H, 1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=7650, chunk_rows=4)
        v.ensure_capacity(40)
        v.set_crf_parameters(13, 4)
        v.reset_c_thresh(7)
        radius = A.crf_feature_radius(6, W, H)
        assert radius == 2
        v.update_detect_features(True, True, 7, radius)
        v.set_roi(roi, 7)
        want = np.concatenate([v.integrate_matrix(f) for f in frames])
        meta, ev = Hst.decode_raw(open(out, "rb").read())
        assert n == len(want) and np.array_equal(ev, want)
        assert np.array_equal(fs, v.feature_set()) and fs.sum() > 0


def test_decoder_new_compressed_reads_what_the_compressed_encoder_wrote():
    """Decoder::new_compressed (decoder.rs:35-51) in the mirror: the "addec" header and every event of a stream written
    by the compressed sink, equal to what the C-ABI source returns (the compressed codec is CPU code: no device)."""
    import adder_amd as A
    rng = np.random.default_rng(0)
    for Cn in (1, 3):
        W, H, n = 40, 30, 4000
        enc = A.CompressedEncoder(W, H, Cn, tps=7650, ref_interval=255, delta_t_max=7650, adu_interval=10, time_mode=1,
                                  c_thresh_max=0)
        ev = np.zeros(n, A.EVENT_DTYPE)
        ev["x"], ev["y"] = rng.integers(0, W, n), rng.integers(0, H, n)
        ev["c"] = 0xFF if Cn == 1 else rng.integers(0, 3, n)
        ev["d"] = rng.integers(0, 12, n)
        ev["t"] = np.sort(rng.integers(1, 7650 * 3, n))
        enc.ingest(ev)
        blob = enc.close()
        dec, p = A.compressed_decode(blob)
        meta, ev2 = Hst.decode_raw(blob)
        assert [int(v) for v in meta[:7]] == [3, W, H, Cn, 7650, 255, 7650] and int(meta[9]) == 10
        assert int(meta[8]) >> 8 == 1 and len(ev2) == len(dec) > 0 and np.array_equal(ev2, dec)
    with pytest.raises(AssertionError):
        Hst.decode_raw(b"addec" + bytes(3))  # a truncated header is an error, not a crash


def _prophesee_restatement(dvs, W, H, ref_time):
    """Prophesee::new + consume() until the input ends + end_events (prophesee.rs:56-372), restated over the oracle's
    Continuous Video and its sparse driver.  f64 exp / ln_1p are libm's on both sides."""
    import math
    from oracle import oracle as O
    v = O.Video(W, H, 1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=2 * ref_time)
    v.set_pixel_mode(1)
    v.ensure_capacity(40)
    v.set_crf_parameters(7, 7)  # EncoderOptions::default -> quality 3 (the pixels keep PixelArena::new's 10 / 1)
    last_t = np.full(W * H, 2, np.int64)
    last_ln = np.full(W * H, math.log1p(128.0 / 255.0))
    theta, view = 0.02, 1000000 // 60
    out, running_t, pos, calls = [], 0, 0, 0

    def as_u8(x):
        return 0 if not x > 0.0 else (255 if x >= 255.0 else int(x))

    def step(x, y, val, intensity, time, no_side=0):
        # pad bit 0 (ADDER_SPARSE_NO_SIDE): the side plane is sampled once per camera event, after its LAST step
        return (x, y, 0xFF, as_u8(val), no_side, np.float32(intensity), np.float32(time))

    while True:
        if running_t == 0:
            start = np.full((H, W, 1), 128, np.uint8)
            v.integrate_matrix(start, time_spanned=float(ref_time))
            assert len(v.integrate_matrix(start, time_spanned=float(ref_time))) == W * H
            running_t = 2
        batch, start_t, ended = [], running_t, False
        while True:
            if pos >= len(dvs):
                ended = True
                break
            e = dvs[pos]
            pos += 1
            running_t = max(running_t, int(e["t"]))
            batch.append(e)
            if int(e["t"]) > start_t + view:
                break
        if ended:
            steps = []
            for y in range(H):
                for x in range(W):
                    p = y * W + x
                    val = (math.exp(last_ln[p]) - 1.0) * 255.0
                    assert running_t - last_t[p] > 0
                    span = (running_t - int(last_t[p])) * ref_time
                    steps.append(step(x, y, val, val * float(span), span, 1))
            out.append(v.integrate_sparse(np.array(steps, O.SPARSE_STEP_DTYPE)))
            break
        steps = []
        for e in batch:
            x, y, t = int(e["x"]), int(e["y"]), int(e["t"])
            p = y * W + x
            if t < last_t[p]:
                continue
            ln = last_ln[p]
            if t > last_t[p] + 1:
                val = (math.exp(ln) - 1.0) * 255.0
                if val < 0.0 or val > 255.0:
                    val, ln = 128.0, math.log1p(128.0 / 255.0)
                gap = t - int(last_t[p]) - 1
                steps.append(step(x, y, val, val * float(gap), gap * ref_time, 1))
            new_ln = ln - theta if int(e["p"]) == 0 else ln + theta
            last_ln[p] = new_ln
            was = int(last_t[p])
            last_t[p] = t
            if t > was:
                val = (math.exp(new_ln) - 1.0) * 255.0
                if val < 0.0 or val > 255.0:
                    val, new_ln = 128.0, math.log1p(128.0 / 255.0)
                last_ln[p] = new_ln
                steps.append(step(x, y, val, val, ref_time))
        out.append(v.integrate_sparse(np.array(steps, O.SPARSE_STEP_DTYPE)) if steps else np.zeros(0, O.EVENT_DTYPE))
        calls += 1
    return np.concatenate(out), calls


@pytest.mark.gpu
def test_prophesee_source_dvs_events_to_adder(tmp_path):
    """SURVEY 8(f)3 end to end through the C++ mirror: a DVS recording (decoded events; the `.dat` reader is file
    parsing and not built) -> Prophesee::consume per 1/60 s of camera time -> sparse steps on the device
    (adder_hip_integrate_sparse) -> ADDER events, then end_events.  Against a restatement of the same driver over the
    oracle.  The driver itself has no reference vector: oracle parity."""
    rng = np.random.default_rng(12)
    W, H, n = 46, 30, 60000
    dvs = np.zeros(n, Hst.DVS_DTYPE)
    dvs["t"] = np.sort(rng.integers(3, 200000, n))  # 0.2 s of camera time: a dozen consume() calls
    hot = rng.integers(0, W * H, 40)
    pix = np.where(rng.random(n) < 0.5, hot[rng.integers(0, 40, n)], rng.integers(0, W * H, n))
    dvs["x"], dvs["y"] = pix % W, pix // W
    dvs["p"] = rng.integers(0, 2, n)
    dvs["t"][100:140] = dvs["t"][100]  # a burst with one timestamp: t == last_t skips the new-intensity step
    got, calls = Hst.prophesee(dvs, W, H, 20)
    want, want_calls = _prophesee_restatement(dvs, W, H, 20)
    assert calls == want_calls >= 10
    assert len(got) == len(want) > n // 2 and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("view,source,dmax,mode,vt", [(0, 0, 0.0, 1, 0), (1, 0, 12.99, 0, 0), (2, 0, 0.0, 0, 0), (3, 0, 0.0, 1, 0),
                                                      (0, 1, 0.0, 0, 0), (0, 0, 0.0, 0, 1), (0, 1, 0.0, 1, 2), (2, 0, 0.0, 0, 1)])
def test_framer_builder_views_and_integration_mode(view, source, dmax, mode, vt):
    """FramerBuilder::mode / view_mode / source of the C++ mirror (driver.rs:98-115): FramerMode::INTEGRATION is stored
    and never read by the reference, so it must frame exactly like INSTANTANEOUS; the views go through
    get_frame_value (scale_intensity.rs:54-109).  ingest_events_events + write_multi_frame_bytes + 3 flushes against
    the framer oracle on the same chunk division."""
    from oracle import oracle as O
    from test_gpu_framer import _synthetic_stream
    rng = np.random.default_rng(40 + view + source)
    W, H, T, rows = 45, 31, 40, 8
    ev, _ = _synthetic_stream(rng, W, H, 1, T, abs_t=True, density=0.7)
    ev = ev.view(O.EVENT_DTYPE) if ev.dtype != O.EVENT_DTYPE else ev
    chunk = ev["y"].astype(np.int64) // rows
    order = np.argsort(chunk, kind="stable")
    ev = ev[order]
    n_chunks = (H + rows - 1) // rows
    offs = np.searchsorted(chunk[order], np.arange(n_chunks + 1)).astype(np.uint64)
    kw = dict(tps=7650, ref_interval=255, delta_t_max=2550, output_fps=30.0, codec_version=3)
    ofr = O.Framer(W, H, 1, chunk_rows=rows, time_mode=O.ABSOLUTE_T, source_camera=O.FRAMED_U8, **kw)
    ofr.set_view(view, source, dmax)
    ofr.set_value_type(vt)  # finish::<u8 / u16 / u32>()
    want = ofr.write_multi_frame_bytes() if ofr.ingest_events_events(ev, offs) else b""
    for _ in range(3):
        ofr.flush_frame_buffer()
        want += ofr.write_frame_bytes()
    got = Hst.frame_events(ev, offs, W, H, 1, time_mode=1, chunk_rows=rows, framer_mode=mode, view_mode=view,
                           source_type=source, practical_d_max=dmax, flushes=3, value_type=vt, **kw)
    assert len(want) > (20 * W * H) << vt and got == want


def _davis_restatement(packets, W, H, mode, *, tps, ref_time, dtm, crf):
    """Davis::consume until the input ends (davis.rs:233-897), restated over the oracle's Video and its sparse driver:
    per packet the DVS events left over from after the previous frame, those before this frame, the frame gaps of every
    pixel, then the deblurred frame itself; at the end every pixel's events are popped.  Returns every event in the
    order the reference feeds them to its encoder."""
    import math
    from oracle import oracle as O
    continuous = mode != 0
    v = O.Video(W, H, 1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm, chunk_rows=H // 4)
    if continuous:
        v.set_pixel_mode(1)
    v.ensure_capacity(40)
    if crf is not None:
        base, cmax, vel = [(0, 0, 10), (0, 1, 9), (1, 3, 8), (2, 7, 7)][crf]
        v.set_crf_parameters(cmax, vel)
        v.reset_c_thresh(base)
    else:
        v.set_crf_parameters(7, 7)
    f32 = np.float32
    tpm = f32(tps) / f32(1e6)
    last_ts = np.zeros(W * H, np.int64)
    last_ln = np.zeros(W * H, np.float64)
    out = []
    state = {"c": 0.15, "start": None, "end": None, "last_after": None, "end_of_last": None}

    def as_u8(x):
        return 0 if not x > 0.0 else (255 if x >= 255.0 else int(x))

    def clamp(val, p):
        if val <= 0.0:
            last_ln[p] = math.log1p(0.0)
            return 0.0
        if val > 255.0:
            last_ln[p] = math.log1p(1.0)
            return 255.0
        return val

    def run(steps):
        if steps:
            out.append(v.integrate_sparse(np.array(steps, O.SPARSE_STEP_DTYPE)))

    def dvs_events(events, ts1, ts2, after2):
        steps = []
        for chunk in range(4):
            for e in events:
                x, y, t = int(e["x"]), int(e["y"]), int(e["t"])
                if y // (H // 4) != chunk:
                    continue
                if not t < ts1:
                    continue
                if ts2 is not None and not (t > ts2 if after2 else t < ts2):
                    continue
                p = y * W + x
                last_val = (math.exp(last_ln[p]) - 1.0) * 255.0
                dmicro = t - int(last_ts[p])
                if dmicro == t:
                    continue
                dticks = f32(dmicro) * tpm
                if dticks < 0:
                    continue
                first = max(f32(f32(last_val) / f32(ref_time)) * dticks, f32(0.0))
                steps.append((x, y, 0xFF, 0, 1 | 2, f32(first), f32(dticks)))
                last_ln[p] *= math.exp(state["c"] if e["on"] else -state["c"])
                fv = clamp((math.exp(last_ln[p]) - 1.0) * 255.0, p)
                steps.append((x, y, 0xFF, as_u8(fv), 1 | 4, f32(fv), f32(0.0)))
                last_ts[p] = t
        run(steps)

    for pk in packets:
        if continuous:
            state["start"], state["end"], state["c"] = int(pk["start"]), int(pk["end"]), float(pk["c"])
            if mode == 2:
                state["end"] = state["start"] + 1
        start = state["start"] if state["start"] is not None else 0
        end = state["end"] if state["end"] is not None else ref_time
        if continuous:
            if state["last_after"] is not None and state["end_of_last"] is not None:
                dvs_events(state["last_after"], start, state["end_of_last"] if mode != 2 else None, True)
            dvs_events(pk["before"], start, None, False)
            steps = []
            for y in range(H):
                for x in range(W):
                    p = y * W + x
                    lv = clamp((math.exp(last_ln[p]) - 1.0) * 255.0, p)
                    dmicro = start - int(last_ts[p])
                    if dmicro == start:
                        continue
                    dticks = f32(dmicro) * tpm
                    if dticks <= 0:
                        continue
                    integ = max((lv / float(ref_time)) * float(dticks), 0.0)
                    steps.append((x, y, 0xFF, as_u8(lv), 0, f32(integ), f32(dticks)))
            run(steps)
        img = np.clip(np.rint(pk["frame"] * 255.0), 0, 255).astype(np.uint8)
        span = f32(ref_time)
        if mode == 1:
            span = f32(end - start)
        if mode == 2:
            state["c"] = 0.15
            img[...] = 0
            span = f32(0.0)
        if not continuous:
            out.append(v.integrate_matrix(img.reshape(H, W, 1), time_spanned=float(span)))
        else:
            run([(x, y, 0xFF, int(img[y, x]), 0, f32(img[y, x]), span) for y in range(H) for x in range(W)])
        last_ln[:] = math.log1p(0.5) if mode == 2 else np.log1p(pk["frame"].reshape(-1))
        if continuous:
            state["last_after"], state["end_of_last"] = pk["after"], end
            last_ts[:] = end
    if continuous:
        run([(x, y, 0xFF, 0, 1 | 8, f32(0.0), f32(0.0)) for y in range(H) for x in range(W)])
    return np.concatenate(out) if out else np.zeros(0, O.EVENT_DTYPE)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_davis_source_frames_and_dvs_events_interleaved(mode):
    """SURVEY 8(f)3, the DAVIS half (davis.rs:233-897) end to end through the C++ mirror: EDI output (deblurred frames
    + the DVS events before / after each exposure; the reconstructor itself is an un-vendored file-parsing crate) ->
    Davis::consume in Framed / RawDavis / RawDvs mode -> sparse steps in the reference's order on the device (the
    split steps of ADDER_SPARSE_INTEGRATE_ONLY / _TEST_ONLY / _FLUSH) -> ADDER events.  Against a restatement of the
    same driver over the oracle.  The driver has no reference vector: oracle parity."""
    rng = np.random.default_rng(20 + mode)
    W, H, P = 38, 28, 7
    tps, ref_time, dtm = 1_000_000, 5000, 500_000
    packets, t = [], 40_000
    base = rng.random((H, W))
    for k in range(P):
        start, end = t, t + int(rng.integers(3000, 9000))
        nxt = end + int(rng.integers(10_000, 30_000))

        def events(lo, hi, n):
            ev = np.zeros(n, Hst.DAVIS_DVS_DTYPE)
            ev["t"] = np.sort(rng.integers(lo, hi, n))
            hot = rng.integers(0, W * H, 25)
            pix = np.where(rng.random(n) < 0.6, hot[rng.integers(0, 25, n)], rng.integers(0, W * H, n))
            ev["x"], ev["y"], ev["on"] = pix % W, pix // W, rng.integers(0, 2, n)
            return ev
        frame = np.clip(base + rng.normal(0, 0.08, (H, W)) + (0.3 if k == 4 else 0.0), -0.05, 1.1)
        # (events that fall inside an exposure or beyond the next frame's start are there on purpose: the checks drop them)
        packets.append({"frame": frame, "c": 0.1 + 0.05 * rng.random(), "start": start, "end": end,
                        "before": events(start - 25_000, start + 2000, 400), "after": events(end - 2000, nxt + 1000, 400)})
        t = nxt
    got, returned = Hst.davis(packets, W, H, mode, tps=tps, ref_time=ref_time, delta_t_max=dtm, crf=2 if mode else None)
    want = _davis_restatement(packets, W, H, mode, tps=tps, ref_time=ref_time, dtm=dtm, crf=2 if mode else None)
    assert len(got) == len(want) > 2 * W * H and np.array_equal(got, want)
    assert 0 < returned <= len(got)
