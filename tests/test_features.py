"""Feature-driven rate control and ROI (utils/cv.rs:56-212, video.rs:865-1112) on CPU.

The reference holds no unit test or fixture for `is_feature` / `handle_features`, so the oracle's LITERAL restatement
of the OpenCV-style scan (oracle/adder_oracle.c fast_is_feature) is pinned here by a second opinion instead: the
textbook definition of FAST 9_16 (an arc of >= 9 contiguous ring pixels all brighter than centre + t or all darker
than centre - t), written independently in numpy.  The device header's bit-mask formulation is checked against
both, and the device-side flow of the feature pass (tests/cpu_sim: per-pixel c_thresh, look-at filter, membership
plane, neighbourhood reset, ROI) against the oracle's handle_features / handle_roi, event for event.
"""
import numpy as np
import pytest

from oracle import oracle as O
import clips
import sim_py
from sim_py import Sim

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast9_textbook(img, t=30):
    """FAST 9_16 from its definition, vectorised over the plane; border 3 excluded."""
    img = img.astype(np.int32)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    c = img[3:h - 3, 3:w - 3]
    ring = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])  # [16][h-6][w-6]
    for flags in (ring > c + t, ring < c - t):
        ff = np.concatenate([flags, flags[:8]])  # circular: 16 starts, 9 long
        run = np.ones_like(flags)
        for k in range(9):
            run &= ff[k:k + 16]
        out[3:h - 3, 3:w - 3] |= run.any(axis=0).astype(np.uint8)
    return out


def oracle_fast_plane(img):
    L = O.lib()
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.zeros((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            out[y, x] = L.oracle_fast_is_feature(img.ctypes.data, w, h, ch, x, y)
    return out


def _images():
    rng = np.random.default_rng(7)
    yield rng.integers(0, 256, (40, 52), dtype=np.uint8)
    yield (rng.integers(0, 2, (40, 52)) * 200 + rng.integers(0, 30, (40, 52))).astype(np.uint8)
    yield (rng.integers(0, 3, (33, 47)) * 100 + rng.integers(0, 40, (33, 47))).astype(np.uint8)
    for k in range(4):  # rectangles on flat ground: real corners
        yield clips.make_clip("corners", 1, 48, 64, 1, seed=k)[0, :, :, 0]
    img = np.full((20, 20), 100, np.uint8)  # thresholds are strict: +-30 is not a corner, +-31 is
    img[10:, 10:] = 130
    yield img
    img = img.copy()
    img[10:, 10:] = 131
    yield img
    yield np.zeros((7, 7), np.uint8)  # one interior pixel
    yield np.zeros((6, 9), np.uint8)  # no interior at all


def test_fast_literal_restatement_equals_the_textbook_definition_and_the_device_form():
    positives = 0
    for img in _images():
        want = fast9_textbook(img)
        assert np.array_equal(oracle_fast_plane(img), want)
        assert np.array_equal(sim_py.fast9_plane(img), want)
        positives += int(want.sum())
    assert positives > 200


def test_fast_looks_at_channel_zero_of_interleaved_planes():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (24, 30, 3), dtype=np.uint8)
    want = fast9_textbook(img[:, :, 0])
    assert np.array_equal(oracle_fast_plane(img), want)
    assert np.array_equal(sim_py.fast9_plane(img), want)


def _run_pair(clip, *, detect, adjust, radius, baseline, roi=None, chunk_rows=1, multi_mode=O.COLLAPSE,
              time_mode=O.ABSOLUTE_T, dtm=7650, crf=(7, 7), check_every=1):
    frames, H, W, Cn = clip.shape
    ov = O.Video(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=255, delta_t_max=dtm,
                 chunk_rows=chunk_rows)
    sv = Sim(W, H, Cn, time_mode=time_mode, multi_mode=multi_mode, ref_time=255, delta_t_max=dtm, max_depth=40)
    ov.ensure_capacity(42)
    ov.set_crf_parameters(*crf)
    sv.set_crf_parameters(*crf)
    ov.reset_c_thresh(baseline)
    sv.reset_c_thresh(baseline)
    ov.update_detect_features(detect, adjust, baseline, radius)
    sv.update_detect_features(detect, adjust, baseline, radius, chunk_rows)
    ov.set_roi(roi, baseline)
    sv.update_roi(roi, baseline)
    total = new = 0
    for k in range(frames):
        a = ov.integrate_matrix(clip[k], time_spanned=255.0)
        rc, b = sv.integrate(clip[k], 255.0)
        assert rc == 0
        assert np.array_equal(a, b), k
        total += len(a)
        new += len(ov.new_features()) if detect else 0
        if k % check_every == 0 or k == frames - 1:
            assert np.array_equal(ov.c_thresh_plane().ravel(), sv.c_thresh_plane()), k
            if detect:
                assert np.array_equal(ov.feature_set(), sv.feature_set(W, H)), k
    if detect:
        assert sv.new_features == new
    return total, new, ov


@pytest.mark.parametrize("multi_mode,dtm", [(O.COLLAPSE, 255), (O.COLLAPSE, 7650), (O.NORMAL, 2550)])
@pytest.mark.parametrize("channels", [1, 3])
def test_feature_feedback_matches_handle_features(multi_mode, dtm, channels):
    clip = clips.make_clip("corners", 36, 40, 56, channels, seed=11 + channels)
    total, new, _ = _run_pair(clip, detect=True, adjust=True, radius=3, baseline=6, multi_mode=multi_mode, dtm=dtm,
                              crf=(13, 4))
    assert new > 20 and total > 0


def test_feature_feedback_changes_the_stream_and_detection_alone_does_not():
    clip = clips.make_clip("corners", 30, 40, 56, 1, seed=5)
    plain, _, _ = _run_pair(clip, detect=False, adjust=False, radius=0, baseline=8, crf=(16, 3))
    seen, new, _ = _run_pair(clip, detect=True, adjust=False, radius=4, baseline=8, crf=(16, 3))
    fed, _, _ = _run_pair(clip, detect=True, adjust=True, radius=4, baseline=8, crf=(16, 3))
    assert new > 0
    assert seen == plain  # no feedback without feature_rate_adjustment
    assert fed > plain    # lowered thresholds around features fire more events


def test_chunk_rows_only_changes_which_event_closes_a_window():
    clip = clips.make_clip("corners", 16, 30, 40, 1, seed=9)
    for chunk_rows in (1, 4, 64):
        _run_pair(clip, detect=True, adjust=True, radius=2, baseline=5, chunk_rows=chunk_rows, crf=(9, 6))


def test_single_pixel_chunks_never_pair_an_event_with_another_pixel():
    # a 7x7 plane with one interior pixel; chunk_rows = 1 and one firing pixel per row makes every window wrap
    clip = np.zeros((12, 7, 7, 1), np.uint8)
    clip[:, :, 3, 0] = (np.arange(12)[:, None] * 37 + np.arange(7)[None, :] * 11) % 251
    _run_pair(clip, detect=True, adjust=True, radius=1, baseline=1, chunk_rows=1, crf=(3, 8))


def test_radius_zero_and_default_radius():
    clip = clips.make_clip("corners", 12, 32, 48, 1, seed=2)
    _, new0, ov0 = _run_pair(clip, detect=True, adjust=True, radius=0, baseline=6, crf=(13, 4))
    assert new0 > 0
    # radius 0: `feature_c_radius > 0` fails (video.rs:1089), thresholds stay uniform
    assert len(np.unique(ov0.c_thresh_plane())) == 1


def test_roi_keeps_its_pixels_at_the_low_threshold():
    clip = clips.make_clip("jitter", 24, 20, 28, 3, seed=4)
    roi = (5, 3, 17, 11)
    total, _, ov = _run_pair(clip, detect=False, adjust=False, radius=0, baseline=9, roi=roi, crf=(20, 2))
    plane = ov.c_thresh_plane()
    assert (plane[3:12, 5:18] == 2).all()  # min(baseline, 2) after every frame
    assert (plane[:3] > 2).all() and (plane[12:] > 2).all()
    plain, _, _ = _run_pair(clip, detect=False, adjust=False, radius=0, baseline=9, crf=(20, 2))
    assert total > plain


def test_roi_and_features_together_and_roi_beyond_the_plane():
    clip = clips.make_clip("corners", 14, 24, 32, 1, seed=8)
    _run_pair(clip, detect=True, adjust=True, radius=3, baseline=7, roi=(20, 10, 500, 400), crf=(13, 4))
