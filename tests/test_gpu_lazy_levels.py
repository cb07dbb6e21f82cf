"""GPU parity of the lazy-levels kernel (adder_cz_kernel: the bounded Collapse regime with only the roots stepped and the
levels replayed from the last 32 frames' input bytes when a flush, pop_top or the end of a batch wants them) -- through
the C-ABI, against the CPU oracle, bit for bit; every test asserts the kernel it means to test
(adder_hip_last_batch_kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
import clips

CRFS = {0: (0, 0, 10), 3: (2, 7, 7), 6: (7, 13, 4), 9: (15, 25, 1)}


def _pair(W, H, Cn, tm, dtm, crf, ref_time=255, **kw):
    import adder_amd as A
    ov = O.Video(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm)
    hv = A.HipVideo(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=ref_time, delta_t_max=dtm, **kw)
    ov.ensure_capacity(24)
    for v in (ov, hv):
        v.set_crf_parameters(crf[1], crf[2])
        v.reset_c_thresh(crf[0])
    return ov, hv


def _batches(ov, hv, clip, lens, rng, kernel, k=0, end=None, T=None):
    end = len(clip) if end is None else end
    total = 0
    kw = {} if T is None else {"time_spanned": T}
    while k < end:
        nb = min(int(rng.choice(lens)), end - k)
        want = [ov.integrate_matrix(clip[k + i], **kw) for i in range(nb)]
        got, offs = hv.integrate_batch(clip[k:k + nb], **kw)
        assert hv.last_batch_kernel() == kernel, (k, hv.last_batch_kernel())
        assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == [len(w) for w in want], (k, nb)
        assert np.array_equal(got, np.concatenate(want)), (k, nb)
        total += len(got)
        k += nb
    return total


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("crf", [3, 6, 9])
def test_lazy_levels_every_content_every_launch_length_and_the_handover(time_mode, crf, monkeypatch):
    """The reference's default mode (Collapse, delta_t_max = 30 frames) at crf > 0: every content kind (jitter inside and
    across the contrast band, runs, steps, dark pixels and zeros inside runs, noise, static, the bench's scene), batches of
    every length (the history of the 32 frames before a launch: one-frame batches, launches that end inside a group), the
    c_thresh ramp; then the planes it leaves are handed to the bounded Collapse kernel (ADDER_HIP_NO_CZ), which continues
    from the materialised levels -- and the lazy kernel does not come back (its history is no longer its own)."""
    import adder_amd as A
    rng = np.random.default_rng(101 + crf + 7 * time_mode)
    for kind in ("jitter", "runs", "steps", "dark", "noise", "static", "scene"):
        frames, H, W = 330, 9, 256   # two full 128-unit segments per row
        clip = (O.synth_clip(O.CONTENT_SCENE, W, H, 1, frames) if kind == "scene"
                else clips.make_clip(kind, frames, H, W, 1, seed=31 + len(kind) + crf))
        ov, hv = _pair(W, H, 1, time_mode, 7650, CRFS[crf])
        stop = 200 + int(rng.integers(0, 40))   # (the handover lands at any position of a group / of a window)
        total = _batches(ov, hv, clip, [1, 2, 5, 16, 29, 30, 31, 64, 100], rng, A.KERNEL_LAZY_LEVELS, end=stop)
        monkeypatch.setenv("ADDER_HIP_NO_CZ", "1")
        total += _batches(ov, hv, clip, [1, 7, 60], rng, A.KERNEL_BOUNDED, k=stop, end=stop + 50)
        monkeypatch.delenv("ADDER_HIP_NO_CZ")
        total += _batches(ov, hv, clip, [1, 7, 60], rng, A.KERNEL_BOUNDED, k=stop + 50)
        if kind != "static":
            assert total > 0
        hv.close()


def test_lazy_levels_other_windows_rgb_ragged_planes_and_deep_chains():
    """delta_t_max of 2, 3, 8 and 32 frames (the longest window the history holds), three channels on a ragged plane (the
    register staging path, padding units), other tick rates; a window beyond the history runs the bounded Collapse kernel."""
    import adder_amd as A
    rng = np.random.default_rng(55)
    for ref_time, dtm_frames, Cn, W, H in ((255, 2, 1, 128, 6), (255, 3, 3, 51, 7), (255, 8, 1, 130, 5), (255, 32, 1, 128, 6),
                                           (1000, 20, 1, 64, 9), (20, 30, 3, 37, 5)):
        clip = clips.make_clip("jitter", 150, H, W, Cn, seed=dtm_frames)
        clip[60:] = clips.make_clip("runs", 90, H, W, Cn, seed=dtm_frames + 1)
        for tm in (O.DELTA_T, O.ABSOLUTE_T):
            ov, hv = _pair(W, H, Cn, tm, ref_time * dtm_frames, CRFS[3], ref_time=ref_time)
            assert _batches(ov, hv, clip, [1, 7, 33, 64], rng, A.KERNEL_LAZY_LEVELS, T=float(ref_time)) > 0
            hv.close()
    ov, hv = _pair(128, 4, 1, O.DELTA_T, 255 * 33, CRFS[3])
    _batches(ov, hv, clips.make_clip("runs", 40, 4, 128, 1, seed=1), [40], rng, A.KERNEL_BOUNDED)
    hv.close()


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_lazy_levels_rollback_restores_the_history(time_mode):
    """A batch whose event buffer is too small is rolled back -- the header's age field, the history ring and the frame
    count with it -- and the retry (and everything behind it) matches the oracle: one-frame calls and batches that end in
    the middle of a window, so that the retry's replays read the frames of the batches before."""
    import adder_amd as A
    W, H = 128, 5
    clip = clips.make_clip("runs", 120, H, W, 1, seed=17)
    clip[50] = 255 - clip[49]   # a scene cut: every pixel flushes its arena (levels replayed from the batches before)
    clip[77] = 255 - clip[76]
    ov, hv = _pair(W, H, 1, time_mode, 7650, CRFS[3])
    want = [ov.integrate_matrix(f) for f in clip]
    got, _ = hv.integrate_batch(clip[:45])
    assert np.array_equal(got, np.concatenate(want[:45])) and hv.last_batch_kernel() == A.KERNEL_LAZY_LEVELS
    for k in range(45, 60):
        if len(want[k]) > 1:
            with pytest.raises(A.AdderHipError) as ei:
                hv.integrate_matrix(clip[k], out_cap=len(want[k]) // 2)
            assert ei.value.code == A.E_OUT_CAPACITY and hv.last_required == len(want[k]), k
        assert np.array_equal(hv.integrate_matrix(clip[k], out_cap=max(len(want[k]), 1)), want[k]), k
    need = sum(len(w) for w in want[60:90])
    with pytest.raises(A.AdderHipError) as ei:
        hv.integrate_batch(clip[60:90], out_cap=need - 1)
    assert ei.value.code == A.E_OUT_CAPACITY and hv.last_required == need
    got, _ = hv.integrate_batch(clip[60:90], out_cap=need)
    assert np.array_equal(got, np.concatenate(want[60:90]))
    got, _ = hv.integrate_batch(clip[90:])
    assert np.array_equal(got, np.concatenate(want[90:])) and hv.last_batch_kernel() == A.KERNEL_LAZY_LEVELS
    hv.close()


def test_lazy_levels_mid_stream_changes():
    """update_quality_manual lowers delta_t_max mid-stream (an unpopped root older than the new window pops at once), the
    c_thresh ramp is restarted, and a reset starts the kernel afresh; a longer window than the history hands over to the
    bounded Collapse kernel for good."""
    import adder_amd as A
    rng = np.random.default_rng(5)
    W, H = 128, 6
    clip = clips.make_clip("jitter", 260, H, W, 1, seed=3)
    clip[100:180] = clips.make_clip("runs", 80, H, W, 1, seed=4)
    ov, hv = _pair(W, H, 1, O.DELTA_T, 7650, CRFS[6])
    _batches(ov, hv, clip, [13, 30], rng, A.KERNEL_LAZY_LEVELS, end=70)
    for v in (ov, hv):
        v.set_delta_t_max(255 * 6)
        v.reset_c_thresh(CRFS[6][0])
    _batches(ov, hv, clip, [13, 30], rng, A.KERNEL_LAZY_LEVELS, k=70, end=150)
    for v in (ov, hv):
        v.set_delta_t_max(255 * 40)
    _batches(ov, hv, clip, [13, 30], rng, A.KERNEL_BOUNDED, k=150, end=200)
    for v in (ov, hv):
        v.set_delta_t_max(255 * 10)   # (delta_t_max_seen keeps the longest window: no way back without a reset)
    _batches(ov, hv, clip, [13, 30], rng, A.KERNEL_BOUNDED, k=200)
    hv.close()
    ov, hv = _pair(W, H, 1, O.ABSOLUTE_T, 7650, CRFS[3])
    _batches(ov, hv, clip, [64], rng, A.KERNEL_LAZY_LEVELS, end=100)
    ov = O.Video(W, H, 1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=7650)
    ov.ensure_capacity(24)
    hv.reset()
    for v in (ov, hv):
        v.set_crf_parameters(CRFS[3][1], CRFS[3][2])
        v.reset_c_thresh(CRFS[3][0])
    _batches(ov, hv, clip, [64], rng, A.KERNEL_LAZY_LEVELS, k=100)
    hv.close()


def test_lazy_levels_full_size_default_quality_1080p_and_the_frame_ring():
    """1080p at the reference's default quality, 150 frames across chunk boundaries in one batch, then the same clip through
    the per-frame ring (one-frame batches: the history ring carries every replay), against the oracle."""
    import adder_amd as A
    W, H, T = 1920, 1080, 150
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, 1, T)
    ov, hv = _pair(W, H, 1, O.ABSOLUTE_T, 7650, CRFS[3])
    want = [ov.integrate_matrix(f) for f in clip]
    got, offs = hv.integrate_batch(clip)
    assert hv.last_batch_kernel() == A.KERNEL_LAZY_LEVELS
    assert [int(offs[i + 1] - offs[i]) for i in range(T)] == [len(w) for w in want]
    assert np.array_equal(got, np.concatenate(want))
    hv.reset()
    hv.set_crf_parameters(CRFS[3][1], CRFS[3][2])
    hv.reset_c_thresh(CRFS[3][0])
    hv.frames_configure(4, max(len(w) for w in want) + 16)
    done = 0
    for k in range(60):
        hv.frame_submit(clip[k])
        if hv.frames_in_flight() == 4:
            assert np.array_equal(hv.frame_collect(), want[done]), done
            done += 1
    while hv.frames_in_flight():
        assert np.array_equal(hv.frame_collect(), want[done]), done
        done += 1
    assert done == 60 and hv.last_batch_kernel() == A.KERNEL_LAZY_LEVELS
    hv.close()
