"""The device header (adder_pixel.hpp), compiled for the host by tests/cpu_sim, must
reproduce the oracle event-for-event in every mode -- this validates the compact
[fired levels | tail] state representation and the count-then-emit split on CPU,
before any GPU time is spent.  (The HIP path itself is tested in test_gpu_parity.py.)
"""
import os
import zlib

import numpy as np
import pytest

from oracle import oracle as O
import clips
from sim_py import Sim


def run_pair(clip, *, time_mode, multi_mode, dtm, ref_time=255, crf=None, default_pixels=False,
             time_spanned=None, max_depth=20):
    frames, H, W, Cn = clip.shape
    ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=ref_time, delta_t_max=dtm)
    sv = Sim(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=ref_time, delta_t_max=dtm,
             max_depth=max_depth)
    ov.ensure_capacity(max_depth + 2)
    if crf is not None:
        base, cmax, vel = crf
        ov.set_crf_parameters(cmax, vel)
        sv.set_crf_parameters(cmax, vel)
        if not default_pixels:
            ov.reset_c_thresh(base)
            sv.reset_c_thresh(base)
    ts = float(ref_time) if time_spanned is None else time_spanned
    total = 0
    for k in range(frames):
        a = ov.integrate_matrix(clip[k], time_spanned=ts)
        rc, b = sv.integrate(clip[k], ts)
        assert rc == 0, (k, rc)
        assert len(a) == len(b), (k, len(a), len(b))
        assert np.array_equal(a, b), k
        total += len(a)
    assert sv.plan_mismatches == 0
    return total, sv.max_m


CRFS = {0: (0, 0, 10), 3: (2, 7, 7), 6: (7, 13, 4), 9: (15, 25, 1)}


@pytest.mark.parametrize("kind", ["noise", "static", "dark", "jitter", "runs", "steps"])
@pytest.mark.parametrize("multi_mode", [O.NORMAL, O.COLLAPSE])
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_modes_crf0(kind, multi_mode, time_mode):
    clip = clips.make_clip(kind, 70, 6, 9, 1, seed=zlib.crc32(f"{kind}-{multi_mode}-{time_mode}".encode()) & 0xFFFF)
    for dtm in (255, 1020, 7650):
        n, _ = run_pair(clip, time_mode=time_mode, multi_mode=multi_mode, dtm=dtm, crf=CRFS[0])
        if kind not in ("static",):
            assert n > 0


@pytest.mark.parametrize("crf", [3, 6, 9])
@pytest.mark.parametrize("multi_mode", [O.NORMAL, O.COLLAPSE])
def test_lossy_crf(crf, multi_mode):
    for kind in ("jitter", "runs", "dark", "steps"):
        clip = clips.make_clip(kind, 90, 5, 8, 1, seed=crf * 100 + multi_mode)
        for tm in (O.DELTA_T, O.ABSOLUTE_T):
            run_pair(clip, time_mode=tm, multi_mode=multi_mode, dtm=7650, crf=CRFS[crf])
            run_pair(clip, time_mode=tm, multi_mode=multi_mode, dtm=255, crf=CRFS[crf])


def test_construction_default_pixels():
    # pixels keep c_thresh 10 / counter 1 when crf() was never called (SURVEY 8(a) note 6)
    clip = clips.make_clip("jitter", 80, 6, 6, 1, seed=5)
    for mm in (O.NORMAL, O.COLLAPSE):
        run_pair(clip, time_mode=O.ABSOLUTE_T, multi_mode=mm, dtm=7650, crf=(2, 7, 7), default_pixels=True)
        run_pair(clip, time_mode=O.DELTA_T, multi_mode=mm, dtm=510, crf=None)


def test_rgb_interleaved():
    clip = clips.make_clip("runs", 60, 4, 5, 3, seed=11)
    for mm in (O.NORMAL, O.COLLAPSE):
        for tm in (O.DELTA_T, O.ABSOLUTE_T):
            run_pair(clip, time_mode=tm, multi_mode=mm, dtm=1020, crf=CRFS[0])


@pytest.mark.parametrize("ref_time,dtm", [(5000, 240000), (1000, 2000), (20, 10000), (255, 6120)])
def test_other_tick_rates(ref_time, dtm):
    clip = clips.make_clip("runs", 80, 4, 6, 1, seed=ref_time)
    for mm in (O.NORMAL, O.COLLAPSE):
        for tm in (O.DELTA_T, O.ABSOLUTE_T):
            run_pair(clip, time_mode=tm, multi_mode=mm, dtm=dtm, ref_time=ref_time, crf=CRFS[0])
            run_pair(clip, time_mode=tm, multi_mode=mm, dtm=dtm, ref_time=ref_time, crf=CRFS[3])


def test_long_static_deep_arena():
    clip = clips.make_clip("static", 600, 3, 4, 1, seed=2)
    n, max_m = run_pair(clip, time_mode=O.DELTA_T, multi_mode=O.NORMAL, dtm=255, crf=CRFS[0])
    assert max_m >= 8  # the arena really does get deep in Normal mode
    # a flush at the very end drains the deep arena
    clip2 = clip.copy()
    clip2[-1] = 255 - clip2[-1]
    run_pair(clip2, time_mode=O.ABSOLUTE_T, multi_mode=O.NORMAL, dtm=255, crf=CRFS[0])
    run_pair(clip2, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, dtm=7650, crf=CRFS[0])


def test_lake_fixture(golden_dir):
    frames = np.load(os.path.join(golden_dir, "lake_scaled_hd_frames_reconstructed.npz"))["frames"]
    clip = frames[:, :, :, None]
    n, _ = run_pair(clip, time_mode=O.DELTA_T, multi_mode=O.NORMAL, dtm=6120, crf=CRFS[0])
    assert n == 201_620


def test_depth_overflow_is_reported():
    clip = clips.make_clip("static", 200, 2, 2, 1, seed=3)
    sv = Sim(2, 2, 1, time_mode=O.DELTA_T, multi_mode=O.NORMAL, delta_t_max=255, max_depth=3)
    sv.set_crf_parameters(0, 10)
    sv.reset_c_thresh(0)
    rcs = [sv.integrate(clip[k], 255.0)[0] for k in range(200)]
    assert -5 in rcs


def _pair(W, H, Cn, tm, mm, dtm, ref_time=255, max_depth=20):
    ov = O.Video(W, H, Cn, time_mode=tm, multi_mode=mm, ref_time=ref_time, delta_t_max=dtm)
    sv = Sim(W, H, Cn, time_mode=tm, multi_mode=mm, ref_time=ref_time, delta_t_max=dtm, max_depth=max_depth)
    ov.ensure_capacity(max_depth + 2)
    return ov, sv


def _same(ov, sv, frame, ts):
    a = ov.integrate_matrix(frame, time_spanned=ts)
    rc, b = sv.integrate(frame, ts)
    assert rc == 0
    assert len(a) == len(b) and np.array_equal(a, b)
    return len(a)


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("crf", [0, 3, 9])
def test_lean_step_is_the_path_of_the_headline_mode(time_mode, crf):
    """Collapse with delta_t_max <= time_spanned runs lean_step + lean_decode (16-byte records) only."""
    for kind in ("noise", "dark", "jitter", "runs", "steps", "static"):
        clip = clips.make_clip(kind, 120, 5, 7, 1, seed=crf * 7 + time_mode)
        ov, sv = _pair(7, 5, 1, time_mode, O.COLLAPSE, 255)
        base, cmax, vel = CRFS[crf]
        for v in (ov, sv):
            v.set_crf_parameters(cmax, vel)
            v.reset_c_thresh(base)
        for k in range(len(clip)):
            _same(ov, sv, clip[k], 255.0)
        assert sv.lean_steps == 120 * 35 and sv.fast_steps == 0 and sv.generic_steps == 0
        if kind == "static":  # lean_quiet / lean_step_quiet (the blocked kernel's quiet-frame loop) took most of them
            assert sv.lean_quiet_steps > 0.8 * sv.lean_steps, (kind, sv.lean_quiet_steps)
        if kind == "dark":
            assert sv.lean_quiet_steps > 0


def test_lean_step_time_spanned_above_delta_t_max_and_long_runs():
    # time_spanned = 2 * ref_time (c_increase_counter advances by 2, :408-410); a 700-frame static
    # stretch drives t past 2^17
    clip = clips.make_clip("steps", 700, 3, 4, 1, seed=77)
    clip[100:680] = clip[100]
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, sv = _pair(4, 3, 1, tm, O.COLLAPSE, 255)
        for v in (ov, sv):
            v.set_crf_parameters(7, 7)
            v.reset_c_thresh(2)
        for k in range(len(clip)):
            _same(ov, sv, clip[k], 510.0 if k % 3 == 0 else 255.0)
        assert sv.generic_steps == 0


def test_quality_change_mid_stream_keeps_parity():
    """update_quality_manual mid-stream (video.rs:1264-1287): after generic batches the lean variant
    must not be chosen again even when delta_t_max drops to ref_time (pixels may hold m >= 2)."""
    clip = clips.make_clip("runs", 150, 6, 6, 1, seed=99)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, sv = _pair(6, 6, 1, tm, O.COLLAPSE, 7650)
        for v in (ov, sv):
            v.set_crf_parameters(0, 10)
            v.reset_c_thresh(0)
        for k in range(60):
            _same(ov, sv, clip[k], 255.0)
        assert sv.max_m >= 2
        for v in (ov, sv):  # update_quality_manual(c_thresh_baseline=1, max=3, multiplier=1, velocity=4)
            v.set_crf_parameters(3, 4)
            v.set_delta_t_max(255)
            v.reset_c_thresh(1)
        for k in range(60, 150):
            _same(ov, sv, clip[k], 255.0)
        assert sv.lean_steps == 0


@pytest.mark.parametrize("multi_mode", [O.NORMAL, O.COLLAPSE])
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_continuous_mode_general_arena_step(multi_mode, time_mode):
    """Mode::Continuous: the device header's general arena step (cont_step: remainders handed to the
    child, zero events, set_d_for_continuous, pop_top on a root without a best event) against the oracle, whose
    Continuous paths the reference's own PixelArena unit tests pin (tests/test_oracle_kat.py)."""
    for kind in ("noise", "dark", "jitter", "runs", "steps", "static"):
        clip = clips.make_clip(kind, 90, 5, 7, 1, seed=len(kind) * 10 + multi_mode)
        for dtm, crf in ((255, 0), (1020, 0), (7650, 3), (7650, 9)):
            ov, sv = _pair(7, 5, 1, time_mode, multi_mode, dtm, max_depth=24)
            ov.set_pixel_mode(1)
            sv.set_continuous()
            base, cmax, vel = CRFS[crf]
            for v in (ov, sv):
                v.set_crf_parameters(cmax, vel)
                v.reset_c_thresh(base)
            total = 0
            for k in range(len(clip)):
                total += _same(ov, sv, clip[k], 255.0 if k % 7 else 510.0)
            assert total > 0 or kind == "static"


def _sparse_steps(rng, W, H, Cn, n, *, hot=0.3):
    """Steps of an event-camera source: a few hot pixels fire again and again, long and short spans, values around
    and far from the previous one.  The contrast test uses the pixel's PERSISTED base_val: the sources' `let mut
    base_val = 0` is an out parameter that integrate_for_px overwrites (video.rs:1336).  pad bit 0 marks steps that
    are not followed by a sampling of the running-intensities side plane (prophesee.rs:259-283)."""
    st = np.zeros(n, O.SPARSE_STEP_DTYPE)
    hotpx = rng.integers(0, W * H, max(2, int(W * H * 0.05)))
    pix = np.where(rng.random(n) < hot, hotpx[rng.integers(0, len(hotpx), n)], rng.integers(0, W * H, n))
    st["x"], st["y"] = pix % W, pix // W
    st["c"] = 0xFF if Cn == 1 else rng.integers(0, Cn, n)
    val = rng.choice(np.array([0, 1, 3, 9, 40, 128, 200, 255]), n)
    span = rng.choice(np.array([1, 1, 1, 2, 5, 40, 700]), n)
    st["frame_val"] = val
    # bit 0: no side-plane sample after the step; bits 1-3: the DAVIS source's partial steps (integrate only, contrast
    # test only, flush only -- davis.rs:331-395, 654-661)
    st["pad"] = rng.integers(0, 2, n) | rng.choice(np.array([0, 0, 0, 0, 2, 2, 4, 4, 8]), n)
    st["intensity"] = (val * span).astype(np.float32)
    st["time"] = (span * 20).astype(np.float32)
    return st


@pytest.mark.parametrize("multi_mode", [O.NORMAL, O.COLLAPSE])
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_sparse_steps_of_event_camera_sources(multi_mode, time_mode):
    """SURVEY 8(f)3, the sparse half: integrate_for_px(px, &mut base_val, frame_val, intensity, time) pixel by pixel in the
    order of a camera's events (prophesee.rs:170-258), after the two dense start-up frames of Prophesee::consume
    (:117-131).  The device flow (cont_step with c_thresh, its counter and running_t per unit) on the host against
    the oracle: every step's events, in order."""
    rng = np.random.default_rng(7 * multi_mode + time_mode)
    W, H = 11, 7
    for Cn, crf in ((1, (7, 7)), (3, (0, 10))):
        ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=20, delta_t_max=40)
        ov.set_pixel_mode(1)
        sv = Sim(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=20, delta_t_max=40, max_depth=24)
        sv.set_continuous()
        ov.ensure_capacity(30)
        for v in (ov, sv):
            v.set_crf_parameters(*crf)
        start = np.full((H, W, Cn), 128, np.uint8)
        for _ in range(2):
            a = ov.integrate_matrix(start, time_spanned=20.0)
            rc, b = sv.integrate(start, 20.0)
            assert rc == 0 and np.array_equal(a, b)
        total = 0
        for k in range(6):
            st = _sparse_steps(rng, W, H, Cn, 400)
            a = ov.integrate_sparse(st)
            rc, b = sv.integrate_sparse(st)
            assert rc == 0, (k, rc)
            assert len(a) == len(b) and np.array_equal(a, b), k
            total += len(a)
        assert total > 500


def test_sparse_step_contrast_test_uses_the_pixels_persisted_base_val():
    """Known answer read off the reference: integrate_for_px sets `*base_val = px.base_val` before the contrast test
    (video.rs:1336), so the `let mut base_val = 0` of prophesee.rs:206,244,334 never reaches it.  Two consecutive
    steps of one pixel with the same frame_val > c_thresh: the first flushes (base_val 128 -> 200), the second finds
    frame_val == base_val and must NOT run pop_best_events again."""
    W, H = 3, 2
    for time_mode in (O.DELTA_T, O.ABSOLUTE_T):
        ov = O.Video(W, H, 1, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=20, delta_t_max=4000)
        ov.set_pixel_mode(1)
        sv = Sim(W, H, 1, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=20, delta_t_max=4000, max_depth=24)
        sv.set_continuous()
        ov.ensure_capacity(30)
        start = np.full((H, W, 1), 128, np.uint8)
        for v in (ov, sv):
            v.set_crf_parameters(0, 10)
            v.reset_c_thresh(0)
        for _ in range(2):
            a = ov.integrate_matrix(start, time_spanned=20.0)
            rc, b = sv.integrate(start, 20.0)
            assert rc == 0 and np.array_equal(a, b)
        st = np.zeros(1, O.SPARSE_STEP_DTYPE)
        st["x"], st["y"], st["c"], st["frame_val"], st["intensity"], st["time"] = 1, 1, 0xFF, 200, 200.0, 20.0
        first = ov.integrate_sparse(st)
        rc, b = sv.integrate_sparse(st)
        assert rc == 0 and np.array_equal(first, b)
        assert len(first) >= 1  # 200 is outside 128 +- 0: the arena is flushed
        second = ov.integrate_sparse(st)
        rc, b = sv.integrate_sparse(st)
        assert rc == 0 and np.array_equal(second, b)
        # base_val is now 200: no pop_best_events.  With the d = 7 root of the flush holding 200 < 256, the second
        # step fires the root once (200 + 200 >= 256) and FramePerfect-free Continuous hands the rest to a child:
        # nothing is emitted at all (delta_t_max 4000 is far away).
        assert len(second) == 0


# ---- the bounded Collapse step (cb_step / cb_emit / cb_pop: Collapse with delta_t_max > time_spanned) ----
def _cb_pair(W, H, Cn, tm, dtm, ref_time=255, max_depth=20, crf=None):
    ov = O.Video(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm)
    sv = Sim(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm, max_depth=max_depth)
    ov.ensure_capacity(max_depth + 2)
    if crf is not None:
        base, cmax, vel = crf
        for v in (ov, sv):
            v.set_crf_parameters(cmax, vel)
            v.reset_c_thresh(base)
    return ov, sv


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("crf", [0, 3, 9])
def test_cb_blocked_launches_match_the_oracle(time_mode, crf):
    """The reference's default mode (Collapse, delta_t_max = 30 frames) through the bounded step, in temporally
    blocked launches of every length that meets a pop (frame 30), a flush or a chunk edge differently: the levels
    stay in prefix coordinates for the whole launch and go back to their resident form at its end."""
    rng = np.random.default_rng(17 + crf + time_mode)
    for kind in ("scene", "runs", "jitter", "static", "dark", "noise", "steps"):
        frames = 150
        clip = (O.synth_clip(O.CONTENT_SCENE, 12, 7, 1, frames) if kind == "scene"
                else clips.make_clip(kind, frames, 7, 12, 1, seed=crf * 7 + len(kind)))
        ov, sv = _cb_pair(12, 7, 1, time_mode, 7650, crf=CRFS[crf])
        k, total = 0, 0
        while k < frames:
            nb = int(rng.choice([1, 2, 3, 7, 29, 30, 31, 64]))
            nb = min(nb, frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            rc, got = sv.integrate_cb_block(clip[k:k + nb], 255.0)
            assert rc == 0, (kind, k, rc)
            assert len(want) == len(got) and np.array_equal(want, got), (kind, k, nb)
            total += len(got)
            k += nb
        assert sv.plan_mismatches == 0 and sv.cb_steps == frames * 12 * 7
        assert total > 0


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_cb_quiet_frames_reduce_to_the_root_update(time_mode):
    """cb_quiet / cb_step_quiet (the device's fast path for waves whose units are all popped down to their root and pass
    the contrast test): long static and lossy runs -- roots that fire at every power of two, black pixels whose d = 128
    root "fires" every frame without advancing delta_t, jitter inside the threshold band, rare flushes -- with the
    reduction on and off, both against the oracle, state included (a later flush emits what the quiet frames left)."""
    frames = 420
    rng = np.random.default_rng(5 + time_mode)
    base = rng.integers(0, 256, (1, 9, 8, 1))
    base[0, :2] = 0                                  # black rows: d = 128 roots
    base[0, 2, :4] = 1
    clip = np.repeat(base, frames, axis=0)
    clip[:, 3:] = np.clip(clip[:, 3:] + rng.integers(-1, 2, (frames, 6, 8, 1)), 0, 255)   # jitter inside crf-3's band
    clip[200:, 5] = 255 - clip[200:, 5]              # a flush of rows that were quiet for 170 frames
    clip[390:] = rng.integers(0, 256, (30, 9, 8, 1))
    clip = clip.astype(np.uint8)
    for crf, quiet in ((3, True), (3, False), (0, True), (9, True)):
        ov, sv = _cb_pair(8, 9, 1, time_mode, 7650, crf=CRFS[crf])
        sv.set_cb_quiet_path(quiet)
        k = 0
        while k < frames:
            nb = min(int(rng.choice([1, 5, 30, 64])), frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            rc, got = sv.integrate_cb_block(clip[k:k + nb], 255.0)
            assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (crf, quiet, k, nb)
            k += nb
        assert sv.plan_mismatches == 0
        if quiet and crf:
            assert sv.cb_quiet_steps > 0.5 * sv.cb_steps, (sv.cb_quiet_steps, sv.cb_steps)   # most unit-frames took it
        if not quiet:
            assert sv.cb_quiet_steps == 0


def test_cb_and_generic_steps_are_interchangeable_mid_stream():
    """Both steps keep the same resident state (level 0 planes + deep planes): alternating them frame by frame -- as
    a context does when a mode switch makes the bounded step ineligible -- must not change a single event."""
    clip = clips.make_clip("runs", 200, 6, 8, 3, seed=77)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, sv = _cb_pair(8, 6, 3, tm, 7650, crf=CRFS[3])
        rng = np.random.default_rng(3)
        for k in range(len(clip)):
            sv.set_use_cb(bool(rng.integers(0, 2)))
            a = ov.integrate_matrix(clip[k])
            rc, b = sv.integrate(clip[k], 255.0)
            assert rc == 0 and np.array_equal(a, b), k
        assert sv.cb_steps > 0 and sv.generic_steps + sv.fast_steps > 0


def test_cb_deep_levels_beyond_the_fast_four():
    """delta_t_max of 500 frames: a static pixel's arena reaches seven levels before the pop, so levels 5+ live in the
    deep planes in prefix form during a launch; a flush then drains all of them, and a too small max_depth is
    reported, not overrun."""
    frames = 560
    clip = clips.make_clip("static", frames, 3, 5, 1, seed=4)
    clip[300:] = 255 - clip[300:]      # one flush of a deep arena
    clip[520:] = clip[0]               # ... and one after the second run's pop
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, sv = _cb_pair(5, 3, 1, tm, 255 * 500, crf=CRFS[0])
        k = 0
        while k < frames:
            nb = min(64, frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            rc, got = sv.integrate_cb_block(clip[k:k + nb], 255.0)
            assert rc == 0 and np.array_equal(want, got), k
            k += nb
        assert sv.max_m >= 6  # root + levels 1..5 at least: past the four fast slots
    sv = Sim(5, 3, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, delta_t_max=255 * 500, max_depth=3)
    sv.set_crf_parameters(0, 10)
    sv.reset_c_thresh(0)
    rcs = [sv.integrate(clip[k], 255.0)[0] for k in range(40)]
    assert -5 in rcs


def test_cb_zero_intensity_quirk_inside_levels():
    """A node that fires at d = 128 (zero intensity) keeps its integration and delta_t (event_pixel_tree.rs:449): with
    crf > 0 zeros arrive INSIDE a run (base_val 3, c_thresh 7), at the root and at deeper levels."""
    rng = np.random.default_rng(9)
    frames = 400
    clip = rng.choice(np.array([0, 0, 0, 1, 2, 3, 5, 7], np.uint8), size=(frames, 4, 6, 1))
    clip[::37] = 200  # a flush now and then
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        for dtm in (510, 2040, 7650):
            ov, sv = _cb_pair(6, 4, 1, tm, dtm, crf=(7, 7, 7))
            k = 0
            while k < frames:
                nb = min(int(rng.integers(1, 40)), frames - k)
                want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
                rc, got = sv.integrate_cb_block(clip[k:k + nb], 255.0)
                assert rc == 0 and np.array_equal(want, got), (dtm, k)
                k += nb


def test_cb_other_tick_rates_and_quality_change():
    clip = clips.make_clip("runs", 160, 4, 6, 1, seed=31)
    for ref_time, dtm in ((5000, 240000), (1000, 2000), (20, 10000), (255, 6120)):
        for tm in (O.DELTA_T, O.ABSOLUTE_T):
            ov, sv = _cb_pair(6, 4, 1, tm, dtm, ref_time=ref_time, crf=CRFS[3])
            for k in range(0, 160, 16):
                if k == 80:  # update_quality_manual mid-stream: thresholds restart, delta_t_max changes
                    for v in (ov, sv):
                        v.set_crf_parameters(13, 4)
                        v.reset_c_thresh(7)
                        v.set_delta_t_max(dtm * 2)
                want = np.concatenate([ov.integrate_matrix(clip[k + i], time_spanned=float(ref_time)) for i in range(16)])
                rc, got = sv.integrate_cb_block(clip[k:k + 16], float(ref_time))
                assert rc == 0 and np.array_equal(want, got), (ref_time, k)
    # a fractional time_spanned is not exact in prefix coordinates: the bounded step must refuse it
    sv = Sim(6, 4, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, delta_t_max=7650)
    assert sv.integrate_cb_block(clip[:2], 254.5)[0] == -7


# ---- the constant-run step (cr_step / cr_emit / cr_pop / cr_materialize: the bounded regime at c_thresh 0) ----
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_cr_blocked_launches_match_the_oracle(time_mode):
    """crf 0 (c_thresh_baseline = c_thresh_max = 0): every change of value flushes, a run integrates ONE intensity, and
    the arena is a function of (intensity, frames since the flush).  cr_* keep only the root and work the levels out when
    a flush or pop_top wants them; between launches the levels go back to the planes in their resident form.  Launches
    of every length against the oracle, on every kind of content, black runs and runs past the pop included."""
    rng = np.random.default_rng(41 + time_mode)
    for kind in ("scene", "runs", "jitter", "static", "dark", "noise", "steps"):
        frames = 170
        clip = (O.synth_clip(O.CONTENT_SCENE, 12, 7, 1, frames) if kind == "scene"
                else clips.make_clip(kind, frames, 7, 12, 1, seed=3 + len(kind)))
        ov, sv = _cb_pair(12, 7, 1, time_mode, 7650, crf=CRFS[0])
        k, total = 0, 0
        while k < frames:
            nb = min(int(rng.choice([1, 2, 3, 7, 29, 30, 31, 64])), frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            rc, got = sv.integrate_cr_block(clip[k:k + nb], 255.0)
            assert rc == 0, (kind, k, rc)
            assert len(want) == len(got) and np.array_equal(want, got), (kind, k, nb)
            total += len(got)
            k += nb
        assert sv.plan_mismatches == 0 and total > 0


def test_cr_every_intensity_and_run_length():
    """The closed forms (last firing = ceil(2^e / I), level k+1's run = level k's run minus its last firing) against the
    stepped oracle for EVERY intensity 0..255 and every run length 1..45 (delta_t_max = 30 and 40 frames: the pop falls
    inside), flushed by a value change: a 256-pixel row per run length, both time modes, rgb interleaving too."""
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        for dtm_frames in (30, 40, 3):
            ov, sv = _cb_pair(256, 1, 1, tm, 255 * dtm_frames, crf=CRFS[0], max_depth=12)
            frames = []
            for run in range(1, 46):
                frames += [np.arange(256, dtype=np.uint8).reshape(1, 256, 1)] * run
                frames += [((np.arange(256) + 1 + run) % 256).astype(np.uint8).reshape(1, 256, 1)]  # the flush (a 1-frame run)
            clip = np.stack(frames)
            k = 0
            while k < len(clip):
                nb = min(37, len(clip) - k)
                want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
                rc, got = sv.integrate_cr_block(clip[k:k + nb], 255.0)
                assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (tm, dtm_frames, k)
                k += nb
            assert sv.plan_mismatches == 0


def test_cr_cb_and_generic_steps_are_interchangeable_mid_stream():
    """The constant-run step reads only the roots and writes the levels back in their resident form: a launch of any of
    the three steps may follow a launch of any other (at crf 0) without a single event changing."""
    clip = clips.make_clip("runs", 260, 6, 8, 3, seed=78)
    rng = np.random.default_rng(8)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, sv = _cb_pair(8, 6, 3, tm, 7650, crf=CRFS[0])
        k = 0
        used = set()
        while k < len(clip):
            which = int(rng.integers(0, 3))
            nb = min(int(rng.choice([1, 2, 5, 17, 33])), len(clip) - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            if which == 0:
                rc, got = sv.integrate_cr_block(clip[k:k + nb], 255.0)
            elif which == 1:
                rc, got = sv.integrate_cb_block(clip[k:k + nb], 255.0)
            else:
                sv.set_use_cb(False)
                parts = [sv.integrate(clip[k + i], 255.0) for i in range(nb)]
                sv.set_use_cb(True)
                rc, got = max(p[0] for p in parts), np.concatenate([p[1] for p in parts])
            used.add(which)
            assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (tm, k, nb, which)
            k += nb
        assert used == {0, 1, 2}


def test_cr_other_tick_rates_and_its_limits():
    clip = clips.make_clip("runs", 200, 4, 6, 1, seed=32)
    for ref_time, dtm in ((5000, 240000), (1000, 2000), (20, 10000), (255, 6120), (255, 255 * 200)):
        for tm in (O.DELTA_T, O.ABSOLUTE_T):
            ov, sv = _cb_pair(6, 4, 1, tm, dtm, ref_time=ref_time, crf=CRFS[0], max_depth=14)
            for k in range(0, 200, 25):
                want = np.concatenate([ov.integrate_matrix(clip[k + i], time_spanned=float(ref_time)) for i in range(25)])
                rc, got = sv.integrate_cr_block(clip[k:k + 25], float(ref_time))
                assert rc == 0 and np.array_equal(want, got), (ref_time, dtm, k)
    # outside the regime: c_thresh > 0, a fractional time_spanned
    sv = Sim(6, 4, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, delta_t_max=7650)
    sv.set_crf_parameters(7, 7)
    sv.reset_c_thresh(2)
    assert sv.integrate_cr_block(clip[:2], 255.0)[0] == -7
    sv = Sim(6, 4, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, delta_t_max=7650)
    sv.set_crf_parameters(0, 10)
    sv.reset_c_thresh(0)
    assert sv.integrate_cr_block(clip[:2], 254.5)[0] == -7


# ---- run records (rr_step / rr_event / rr_pack: the bounded regime at crf 0 with integer state, both time modes) ----
@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_rr_blocked_launches_match_the_oracle(time_mode):
    """A unit is {base_val, n, r1, popped, last_fired_t / T}; the step parks one record per flush / collapsed flush /
    pop_top and the events are worked out from the record alone (the chain of cr_node; in AbsoluteT each event advances
    last_fired_t by its node's last firing).  Launches of every length, every content, against the oracle."""
    rng = np.random.default_rng(43 + time_mode)
    for kind in ("scene", "runs", "jitter", "static", "dark", "noise", "steps"):
        frames = 170
        clip = (O.synth_clip(O.CONTENT_SCENE, 12, 7, 1, frames) if kind == "scene"
                else clips.make_clip(kind, frames, 7, 12, 1, seed=3 + len(kind)))
        ov, sv = _cb_pair(12, 7, 1, time_mode, 7650, crf=CRFS[0])
        k, total = 0, 0
        while k < frames:
            nb = min(int(rng.choice([1, 2, 3, 7, 29, 30, 31, 64])), frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            rc, got = sv.integrate_rr_block(clip[k:k + nb], 255.0)
            assert rc == 0, (kind, k, rc)
            assert len(want) == len(got) and np.array_equal(want, got), (kind, k, nb)
            total += len(got)
            k += nb
        assert sv.plan_mismatches == 0 and total > 0


def test_rr_every_intensity_and_run_length():
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        for dtm_frames in (30, 40, 3, 2):
            ov, sv = _cb_pair(256, 1, 1, tm, 255 * dtm_frames, crf=CRFS[0], max_depth=12)
            frames = []
            for run in range(1, 46):
                frames += [np.arange(256, dtype=np.uint8).reshape(1, 256, 1)] * run
                frames += [((np.arange(256) + 1 + run) % 256).astype(np.uint8).reshape(1, 256, 1)]  # the flush (a 1-frame run)
            clip = np.stack(frames)
            k = 0
            while k < len(clip):
                nb = min(37, len(clip) - k)
                want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
                rc, got = sv.integrate_rr_block(clip[k:k + nb], 255.0)
                assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (tm, dtm_frames, k)
                k += nb
            assert sv.plan_mismatches == 0


def test_rr_interchangeable_with_the_other_steps_mid_stream():
    clip = clips.make_clip("runs", 260, 6, 8, 3, seed=79)
    rng = np.random.default_rng(9)
    for tm in (O.DELTA_T, O.ABSOLUTE_T):
        ov, sv = _cb_pair(8, 6, 3, tm, 7650, crf=CRFS[0])
        k = 0
        used = set()
        while k < len(clip):
            which = int(rng.integers(0, 4))
            nb = min(int(rng.choice([1, 2, 5, 17, 33])), len(clip) - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            if which == 0:
                rc, got = sv.integrate_cr_block(clip[k:k + nb], 255.0)
            elif which == 1:
                rc, got = sv.integrate_cb_block(clip[k:k + nb], 255.0)
            elif which == 2:
                rc, got = sv.integrate_rr_block(clip[k:k + nb], 255.0)
            else:
                sv.set_use_cb(False)
                parts = [sv.integrate(clip[k + i], 255.0) for i in range(nb)]
                sv.set_use_cb(True)
                rc, got = max(p[0] for p in parts), np.concatenate([p[1] for p in parts])
            used.add(which)
            assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (tm, k, nb, which)
            k += nb
        assert used == {0, 1, 2, 3}


def test_rr_other_tick_rates_and_its_limits():
    clip = clips.make_clip("runs", 200, 4, 6, 1, seed=33)
    for ref_time, dtm in ((5000, 240000), (1000, 2000), (20, 10000), (255, 6120), (255, 255 * 200), (255, 300)):
        for tm in (O.DELTA_T, O.ABSOLUTE_T):
            ov, sv = _cb_pair(6, 4, 1, tm, dtm, ref_time=ref_time, crf=CRFS[0], max_depth=14)
            if tm == O.ABSOLUTE_T and ref_time < 255:  # T q >= 1 is what keeps last_fired_t on multiples of T
                assert sv.integrate_rr_block(clip[:2], float(ref_time))[0] == -7
                continue
            for k in range(0, 200, 25):
                want = np.concatenate([ov.integrate_matrix(clip[k + i], time_spanned=float(ref_time)) for i in range(25)])
                rc, got = sv.integrate_rr_block(clip[k:k + 25], float(ref_time))
                assert rc == 0 and np.array_equal(want, got), (ref_time, dtm, tm, k)
    sv = Sim(6, 4, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, delta_t_max=7650)
    sv.set_crf_parameters(7, 7)
    sv.reset_c_thresh(2)
    assert sv.integrate_rr_block(clip[:2], 255.0)[0] == -7


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_rr_in_mode_normal(time_mode):
    """Mode Normal under the same conditions: pop_top leaves [level 1, level 2, ..] = the arena of a run of n - j frames
    (the levels go on being visited), a flush after the pop emits the whole chain.  Every content, blocked launches, and
    launches of the generic step in between (the same planes)."""
    rng = np.random.default_rng(47 + time_mode)
    for dtm in (7650, 255 * 3, 255 * 40, 255):  # (255 = time_spanned: a flush and the new root's pop in one frame, one record)
        for kind in ("scene", "runs", "jitter", "static", "dark", "noise"):
            frames = 150
            clip = (O.synth_clip(O.CONTENT_SCENE, 12, 7, 1, frames) if kind == "scene"
                    else clips.make_clip(kind, frames, 7, 12, 1, seed=3 + len(kind)))
            ov = O.Video(12, 7, 1, time_mode=time_mode, multi_mode=O.NORMAL, ref_time=255, delta_t_max=dtm)
            sv = Sim(12, 7, 1, time_mode=time_mode, multi_mode=O.NORMAL, ref_time=255, delta_t_max=dtm, max_depth=20)
            ov.ensure_capacity(24)
            for v in (ov, sv):
                v.set_crf_parameters(0, 10)
                v.reset_c_thresh(0)
            k, total, used = 0, 0, set()
            while k < frames:
                nb = min(int(rng.choice([1, 2, 3, 7, 29, 31, 64])), frames - k)
                want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
                if rng.integers(0, 4) == 0:
                    parts = [sv.integrate(clip[k + i], 255.0) for i in range(nb)]
                    rc, got = max(p[0] for p in parts), np.concatenate([p[1] for p in parts])
                    used.add("generic")
                else:
                    rc, got = sv.integrate_rr_block(clip[k:k + nb], 255.0)
                    used.add("rr")
                assert rc == 0, (dtm, kind, k, rc)
                assert len(want) == len(got) and np.array_equal(want, got), (dtm, kind, k, nb)
                total += len(got)
                k += nb
            assert sv.plan_mismatches == 0 and total > 0 and "rr" in used


# ---- lean runs (lr_step / lr_decode8 / lr_pack: the headline regime at crf 0, a unit = {base_val, rho, popped}) ----
def _lean_pair(W, H, Cn, dtm=255, ref_time=255, time_mode=O.DELTA_T):
    ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm)
    sv = Sim(W, H, Cn, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm)
    ov.ensure_capacity(4)
    for v in (ov, sv):
        v.set_crf_parameters(0, 10)
        v.reset_c_thresh(0)
    return ov, sv


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_lr_blocked_launches_match_the_oracle_and_interleave_with_the_lean_step(time_mode):
    """BASELINE configs 2-4's mode at crf 0: launches of the lean-runs step of every length on every content, mixed at
    random with frames of the ordinary lean step (both keep the same resident planes) -- every event equals the oracle's.
    In AbsoluteT last_fired_t / T rides along as an integer (lr_step_lq / lr_decode12)."""
    rng = np.random.default_rng(61 + time_mode)
    for kind in ("scene", "runs", "jitter", "static", "dark", "noise", "steps"):
        frames = 260
        clip = (O.synth_clip(O.CONTENT_SCENE, 12, 7, 1, frames) if kind == "scene"
                else clips.make_clip(kind, frames, 7, 12, 1, seed=5 + len(kind)))
        ov, sv = _lean_pair(12, 7, 1, time_mode=time_mode)
        k, used = 0, set()
        while k < frames:
            nb = min(int(rng.choice([1, 2, 3, 7, 31, 64])), frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            if rng.integers(0, 3):
                rc, got = sv.integrate_lr_block(clip[k:k + nb], 255.0)
                used.add("lr")
            else:
                parts = [sv.integrate(clip[k + i], 255.0) for i in range(nb)]
                rc, got = max(p[0] for p in parts), np.concatenate([p[1] for p in parts])
                used.add("lean")
            assert rc == 0, (kind, k, rc)
            assert len(want) == len(got) and np.array_equal(want, got), (kind, k, nb)
            k += nb
        assert used == {"lr", "lean"}


def test_lr_every_intensity_and_run_length_rgb_and_tick_rates():
    """Event A worked out from (base_val, rho) for every intensity 0..255 and runs of 1..70 frames (and one of 700), the
    time_spanned > delta_t_max case, other tick rates, three channels."""
    for ref_time, dtm, T in ((255, 255, 255.0), (255, 255, 510.0), (1000, 1000, 1000.0), (20, 20, 20.0)):
        ov, sv = _lean_pair(256, 1, 1, dtm=dtm, ref_time=ref_time)
        frames = []
        for run in list(range(1, 71, 3)) + [700]:
            frames += [np.arange(256, dtype=np.uint8).reshape(1, 256, 1)] * run
            frames += [((np.arange(256) + 1 + run) % 256).astype(np.uint8).reshape(1, 256, 1)]
        clip = np.stack(frames)
        k = 0
        while k < len(clip):
            nb = min(64, len(clip) - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i], time_spanned=T) for i in range(nb)])
            rc, got = sv.integrate_lr_block(clip[k:k + nb], T)
            assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (ref_time, T, k)
            k += nb
    clip = clips.make_clip("runs", 120, 5, 6, 3, seed=9)
    ov, sv = _lean_pair(6, 5, 3)
    for k in range(0, 120, 40):
        want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(40)])
        rc, got = sv.integrate_lr_block(clip[k:k + 40], 255.0)
        assert rc == 0 and np.array_equal(want, got), k
    # AbsoluteT: every intensity and run length too, three channels
    ov, sv = _lean_pair(256, 1, 1, time_mode=O.ABSOLUTE_T)
    frames = []
    for run in list(range(1, 40, 2)) + [300]:
        frames += [np.arange(256, dtype=np.uint8).reshape(1, 256, 1)] * run
        frames += [((np.arange(256) + 1 + run) % 256).astype(np.uint8).reshape(1, 256, 1)]
    clipa = np.stack(frames)
    for k in range(0, len(clipa), 64):
        nb = min(64, len(clipa) - k)
        want = np.concatenate([ov.integrate_matrix(clipa[k + i]) for i in range(nb)])
        rc, got = sv.integrate_lr_block(clipa[k:k + nb], 255.0)
        assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), k
    ov, sv = _lean_pair(6, 5, 3, time_mode=O.ABSOLUTE_T)
    for k in range(0, 120, 40):
        want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(40)])
        rc, got = sv.integrate_lr_block(clip[k:k + 40], 255.0)
        assert rc == 0 and np.array_equal(want, got), k
    # outside the regime: AbsoluteT at a tick rate whose events need not advance last_fired_t by whole frames
    sv = Sim(6, 5, 3, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=20, delta_t_max=20)
    sv.set_crf_parameters(0, 10)
    sv.reset_c_thresh(0)
    assert sv.integrate_lr_block(clip[:2], 20.0)[0] == -7
    sv = Sim(6, 5, 3, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, delta_t_max=255)
    assert sv.integrate_lr_block(clip[:2], 255.0)[0] == -7  # construction-default pixels: c_thresh 10


def test_lean_quiet_groups_only_take_the_closed_form_at_integer_time_steps():
    """adder_lean_kernel also runs at fractional time_spanned, where the closed form's ONE rounding of delta_t + n T is not the
    reference's n rounded additions (event_pixel_tree.rs:449-451): the group form must step there (and still apply, bit for bit,
    at integer steps)."""
    from sim_py import lean_group_check
    for T in (255.0, 1000.0, 20.0):
        bad, applied = lean_group_check(T, 20000, seed=int(T))
        assert bad == 0 and applied > 5000, (T, bad, applied)
    for T in (300.7, 1000.3, 33333.332, 0.5):
        bad, applied = lean_group_check(T, 20000, seed=7)
        assert bad == 0, (T, bad, applied)   # whatever it applies is exact ...
        assert applied == 0, (T, applied)     # ... and at a fractional step it applies nothing: the units step their frames


# ---- lean runs, packed (lp_step: four units per word, adder_lp_kernel + the expansion's format 7) ----
def test_lp_packed_step_matches_the_oracle_and_interleaves_with_the_other_lean_steps():
    """The headline regime through the packed step: launches of every length on every content (planes whose unit count is
    no multiple of four included), mixed at random with lr launches and frames of the ordinary lean step -- all three keep
    the same resident planes; every event equals the oracle's, every frame's masks equal lr_step's (rc -12 / -13)."""
    rng = np.random.default_rng(77)
    used = set()
    for kind in ("scene", "runs", "jitter", "static", "dark", "noise", "steps"):
        for (W, H, Cn) in ((12, 7, 1), (5, 3, 3), (9, 2, 1)):
            frames = 200
            clip = (O.synth_clip(O.CONTENT_SCENE, W, H, Cn, frames) if kind == "scene"
                    else clips.make_clip(kind, frames, H, W, Cn, seed=5 + len(kind)))
            ov, sv = _lean_pair(W, H, Cn)
            k = 0
            while k < frames:
                nb = min(int(rng.choice([1, 2, 3, 7, 16, 31, 64])), frames - k)
                want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
                pick = int(rng.integers(0, 4))
                if pick >= 2:
                    rc, got = sv.integrate_lp_block(clip[k:k + nb], 255.0)
                    used.add("lp")
                elif pick == 1:
                    rc, got = sv.integrate_lr_block(clip[k:k + nb], 255.0)
                    used.add("lr")
                else:
                    parts = [sv.integrate(clip[k + i], 255.0) for i in range(nb)]
                    rc, got = max(p[0] for p in parts), np.concatenate([p[1] for p in parts])
                    used.add("lean")
                assert rc == 0, (kind, W, k, rc)
                assert len(want) == len(got) and np.array_equal(want, got), (kind, W, k, nb)
                k += nb
    assert used == {"lp", "lr", "lean"}


def test_lp_every_intensity_and_run_length_and_tick_rates():
    """Every intensity 0..255 against runs of 1..70 frames (and one of 700) through the packed step's start-frame form
    of rho, at other tick rates and with time_spanned > delta_t_max; refused outside its regime."""
    for ref_time, dtm, T in ((255, 255, 255.0), (255, 255, 510.0), (1000, 1000, 1000.0), (20, 20, 20.0)):
        ov, sv = _lean_pair(256, 1, 1, dtm=dtm, ref_time=ref_time)
        frames = []
        for run in list(range(1, 71, 3)) + [700]:
            frames += [np.arange(256, dtype=np.uint8).reshape(1, 256, 1)] * run
            frames += [((np.arange(256) + 1 + run) % 256).astype(np.uint8).reshape(1, 256, 1)]
        clip = np.stack(frames)
        k = 0
        while k < len(clip):
            nb = min(64, len(clip) - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i], time_spanned=T) for i in range(nb)])
            rc, got = sv.integrate_lp_block(clip[k:k + nb], T)
            assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (ref_time, T, k)
            k += nb
    clip = clips.make_clip("runs", 120, 5, 6, 3, seed=9)
    sv = Sim(6, 5, 3, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
    sv.set_crf_parameters(0, 10)
    sv.reset_c_thresh(0)
    assert sv.integrate_lp_block(clip[:2], 255.0)[0] == -7  # AbsoluteT: the lr kernel's
    sv = Sim(6, 5, 3, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, delta_t_max=255)
    assert sv.integrate_lp_block(clip[:2], 255.0)[0] == -7  # construction-default pixels: c_thresh 10


# ---- quiet GROUPS: 16 frames of a quiet unit decided at once (quiet_group_apply / lr_quiet_run) ----
_quiet_group_clip = clips.quiet_group_clip


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_cb_quiet_groups_break_at_every_position_with_ramp_and_firings(time_mode):
    """The group form of the bounded Collapse kernel's quiet path: quiet broken at every position of a 16-frame group, the
    c_thresh ramp stepping inside groups (crf 3: 2 -> 7 over the first 35 frames after each reset of the thresholds),
    roots firing inside groups (every power of two), dark pixels that fire twice in a group and black roots that wake up
    (both stepped by the unit), launches of every length -- bit-exact against the oracle with the group form on and off."""
    frames, H, W = 1088, 10, 8
    rng = np.random.default_rng(23 + time_mode)
    clip, breaks = _quiet_group_clip(frames, H, W, rng, jitter=1)
    assert sorted({b % 16 for b in breaks}) == list(range(16))
    for crf, groups_on, lens in ((3, True, [64]), (3, True, [1, 5, 16, 30, 47, 64]), (3, False, [64]), (6, True, [64, 33]), (9, True, [64])):
        ov, sv = _cb_pair(W, H, 1, time_mode, 7650, crf=CRFS[crf])
        sv.set_quiet_group_path(groups_on)
        k = 0
        while k < frames:
            nb = min(int(rng.choice(lens)), frames - k)
            if k and k % 320 == 0:   # update_crf mid-stream: every pixel's c_thresh back to the baseline, the ramp starts again
                for v in (ov, sv):
                    v.reset_c_thresh(CRFS[crf][0])
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            rc, got = sv.integrate_cb_block(clip[k:k + nb], 255.0)
            assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (crf, groups_on, k, nb)
            k += nb
        assert sv.plan_mismatches == 0
        done, fires, slow = sv.quiet_groups
        if groups_on:
            assert done > 0.4 * frames * H * W / 16 and fires > 50 and slow > 0, (crf, done, fires, slow)
        else:
            assert done == fires == slow == 0


def test_quiet_group_statistics_and_closed_form_against_the_stepped_root():
    """quiet_group_apply against cb_step_quiet frame by frame on random roots and random groups (through the sim's cb block
    on a popped plane this is implied; here the corner cases are forced: thresholds about to be reached in the first and the
    last frame of a group, sums one short of the threshold, groups of fewer than 16 frames)."""
    rng = np.random.default_rng(77)
    H, W = 4, 64
    for trial in range(6):
        frames = 30 + 16 * 6 + int(rng.integers(0, 16))
        base = rng.integers(1, 256, (1, H, W, 1))
        clip = np.repeat(base, frames, axis=0).astype(np.int64)
        clip[31:] += rng.integers(-2, 3, (frames - 31, H, W, 1))
        clip = np.clip(clip, 1, 255).astype(np.uint8)
        ov, sv = _cb_pair(W, H, 1, O.DELTA_T, 7650, crf=CRFS[3])
        want = np.concatenate([ov.integrate_matrix(f) for f in clip[:32]])
        rc, got = sv.integrate_cb_block(clip[:32], 255.0)   # frame 30 pops every root; frame 31 is quiet
        assert rc == 0 and np.array_equal(want, got)
        k = 32
        while k < frames:
            nb = min(16 * int(rng.integers(1, 4)) - int(rng.integers(0, 2)) * int(rng.integers(0, 15)), frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            rc, got = sv.integrate_cb_block(clip[k:k + nb], 255.0)
            assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (trial, k, nb)
            k += nb
        # what the quiet frames left behind shows in the next flush: every root's best event
        flush = ((clip[-1].astype(np.int64) + 128) % 256).astype(np.uint8)   # 128 away from every base_val
        want = ov.integrate_matrix(flush)
        rc, got = sv.integrate_cb_block(flush[None], 255.0)
        assert rc == 0 and len(want) == 2 * H * W and np.array_equal(want, got), trial
        assert sv.quiet_groups[0] > 0 and sv.quiet_groups[1] > 0


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_lr_quiet_groups_break_at_every_position(time_mode):
    """The lean-runs kernel's group form (every byte of a 16-frame group equals base_val: rho += 16, or the black root stays
    one frame old): static content with a change at every position of a group, launches of every length, both time modes."""
    frames, H, W = 1088, 10, 8
    rng = np.random.default_rng(41 + time_mode)
    clip, breaks = _quiet_group_clip(frames, H, W, rng, jitter=0)
    assert sorted({b % 16 for b in breaks}) == list(range(16))
    for groups_on, lens in ((True, [64]), (True, [1, 2, 16, 31, 60, 64]), (False, [64])):
        ov, sv = _lean_pair(W, H, 1, time_mode=time_mode)
        sv.set_quiet_group_path(groups_on)
        k = 0
        while k < frames:
            nb = min(int(rng.choice(lens)), frames - k)
            want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
            rc, got = sv.integrate_lr_block(clip[k:k + nb], 255.0)
            assert rc == 0 and len(want) == len(got) and np.array_equal(want, got), (groups_on, k, nb)
            k += nb
        assert (sv.quiet_groups[0] > 0.8 * frames * H * W / 16) == groups_on
