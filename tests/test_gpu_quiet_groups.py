"""GPU parity of the quiet-GROUP paths (16 frames of a quiet wave decided at once instead of stepped): the lean-runs
kernel (every byte of the group equals base_val) and the bounded Collapse kernel (min / max / sum of the group against the
group's smallest c_thresh, at most one firing in closed form) -- through the C-ABI, against the CPU oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
import clips

CRFS = {0: (0, 0, 10), 3: (2, 7, 7), 6: (7, 13, 4), 9: (15, 25, 1)}


def _hipmod():
    import adder_amd as A
    return A


def _pair(W, H, Cn, tm, dtm, crf, **kw):
    import adder_amd as A
    ov = O.Video(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
    hv = A.HipVideo(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm, **kw)
    ov.ensure_capacity(24)
    for v in (ov, hv):
        v.set_crf_parameters(crf[1], crf[2])
        v.reset_c_thresh(crf[0])
    return ov, hv


def _run_batches(ov, hv, clip, lens, rng, reset_every=0, crf=None, kernel=None):
    k, total = 0, 0
    while k < len(clip):
        nb = min(int(rng.choice(lens)), len(clip) - k)
        if reset_every and k and k % reset_every == 0:
            for v in (ov, hv):   # update_crf mid-stream: every pixel's c_thresh back to the baseline, the ramp starts again
                v.reset_c_thresh(crf[0])
        want = [ov.integrate_matrix(clip[k + i]) for i in range(nb)]
        got, offs = hv.integrate_batch(clip[k:k + nb])
        if kernel is not None and nb > 1:   # (the test means THIS kernel: adder_hip_last_batch_kernel; one-frame batches run the one-frame kernels)
            assert hv.last_batch_kernel() == kernel, (k, nb, hv.last_batch_kernel())
        assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == [len(w) for w in want], (k, nb)
        assert np.array_equal(got, np.concatenate(want)), (k, nb)
        total += len(got)
        k += nb
    return total


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
def test_lean_runs_quiet_groups_break_at_every_position(monkeypatch, time_mode):
    """adder_lr_kernel: one 128-unit wave per row; static rows (black, dark, mid, bright) whose groups are skipped, a change at
    every position of a 16-frame group, batches of every length (launches that end inside a group, short last groups)."""
    frames, H, W = 1088, 12, 128
    rng = np.random.default_rng(3 + time_mode)
    clip, breaks = clips.quiet_group_clip(frames, H, W, rng, jitter=0)
    assert sorted({b % 16 for b in breaks}) == list(range(16))
    A = _hipmod()
    for packed in ((True, False) if time_mode == O.DELTA_T else (False,)):   # DeltaT: adder_lp_kernel (packed bytes), then adder_lr_kernel
        for lens in ([frames], [64, 60, 37, 16, 100, 1, 2]):
            ov, hv = _pair(W, H, 1, time_mode, 255, CRFS[0])
            monkeypatch.setenv("ADDER_HIP_NO_LP", "0" if packed else "1")
            assert _run_batches(ov, hv, clip, lens, rng, kernel=A.KERNEL_LEAN_RUNS_PACKED if packed else A.KERNEL_LEAN_RUNS) > 0
            hv.close()
    monkeypatch.delenv("ADDER_HIP_NO_LP")
    # ragged plane (the register staging path, padding units), three channels
    clip3, _ = clips.quiet_group_clip(200, 7, 51, rng, jitter=0, C=3)
    ov, hv = _pair(51, 7, 3, time_mode, 255, CRFS[0])
    _run_batches(ov, hv, clip3, [200], rng)
    hv.close()


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("crf", [3, 6, 9])
def test_bounded_collapse_quiet_groups_break_at_every_position_with_ramp_and_firings(time_mode, crf):
    """adder_cb_kernel: quiet broken at every position of a group, the c_thresh ramp stepping inside groups (and restarted by
    a mid-stream update_crf), roots firing inside groups, dark pixels that fire twice in a group and black roots that wake
    up (stepped by the unit), launches of every length."""
    frames, H, W = 1088, 12, 128
    rng = np.random.default_rng(11 + time_mode + crf)
    clip, breaks = clips.quiet_group_clip(frames, H, W, rng, jitter=1)
    for lens in ([frames], [64, 60, 37, 16, 100, 1, 5]):
        ov, hv = _pair(W, H, 1, time_mode, 7650, CRFS[crf], max_depth=20)
        assert _run_batches(ov, hv, clip, lens, rng, reset_every=320, crf=CRFS[crf], kernel=_hipmod().KERNEL_BOUNDED) > 0
        hv.close()
    clip3, _ = clips.quiet_group_clip(260, 7, 51, rng, jitter=2, C=3)
    ov, hv = _pair(51, 7, 3, time_mode, 7650, CRFS[crf], max_depth=20)
    _run_batches(ov, hv, clip3, [260], rng)
    hv.close()


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("crf", [3, 9])
def test_lean_kernel_quiet_groups_break_at_every_position_with_ramp_and_firings(time_mode, crf):
    """adder_lean_kernel (Collapse with delta_t_max = ref_time at crf > 0: the lean step, not the lean-runs one): the same
    group form in front of its quiet loop -- static rows with jitter inside the band, the ramp, firings, dark and black
    rows, a flush at every position of a group."""
    frames, H, W = 1088, 12, 128
    rng = np.random.default_rng(29 + time_mode + crf)
    clip, breaks = clips.quiet_group_clip(frames, H, W, rng, jitter=1)
    for lens in ([frames], [64, 60, 37, 16, 100, 1, 5]):
        ov, hv = _pair(W, H, 1, time_mode, 255, CRFS[crf])
        assert _run_batches(ov, hv, clip, lens, rng, reset_every=320, crf=CRFS[crf], kernel=_hipmod().KERNEL_LEAN) > 0
        hv.close()
    clip3, _ = clips.quiet_group_clip(260, 7, 51, rng, jitter=2, C=3)
    ov, hv = _pair(51, 7, 3, time_mode, 255, CRFS[crf])
    _run_batches(ov, hv, clip3, [260], rng)
    hv.close()


@pytest.mark.parametrize("T", [300.7, 1000.3, 255.0])
def test_lean_kernel_quiet_groups_at_fractional_time_steps(T):
    """adder_lean_kernel accepts a fractional time_spanned (only the integer-state kernels refuse it).  The group form's closed
    form rounds delta_t + n T once where the reference adds T n times (event_pixel_tree.rs:449-451): at a fractional step the
    groups must be stepped (lean_group_apply) -- static rows with jitter inside the band, the ramp, firings, DeltaT: every
    best delta_t and truncated t against the oracle."""
    frames, H, W = 400, 12, 128
    rng = np.random.default_rng(131)
    clip, _ = clips.quiet_group_clip(frames, H, W, rng, jitter=1)
    A = _hipmod()
    for crf in (3, 9):
        dtm = 255
        ov = O.Video(W, H, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
        hv = A.HipVideo(W, H, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
        ov.ensure_capacity(24)
        for v in (ov, hv):
            v.set_crf_parameters(CRFS[crf][1], CRFS[crf][2])
            v.reset_c_thresh(CRFS[crf][0])
        k = 0
        for nb in (64, 60, 37, 16, 100, 64, 59):
            want = [ov.integrate_matrix(clip[k + i], time_spanned=T) for i in range(nb)]
            got, offs = hv.integrate_batch(clip[k:k + nb], time_spanned=T)
            assert hv.last_batch_kernel() == A.KERNEL_LEAN
            assert [int(offs[i + 1] - offs[i]) for i in range(nb)] == [len(w) for w in want], (T, crf, k)
            assert np.array_equal(got, np.concatenate(want)), (T, crf, k)
            k += nb
        hv.close()


def test_quiet_groups_full_size_static_and_default_quality_1080p():
    """1080p: static content through the lean-runs kernel and the reference's default mode at its default quality through the
    bounded Collapse kernel, 150 frames across chunk boundaries (the pop at frame 30, then quiet groups), against the oracle."""
    import adder_amd as A
    W, H, T = 1920, 1080, 150
    for content, tm, dtm, crf in ((O.CONTENT_STATIC, O.DELTA_T, 255, CRFS[0]), (O.CONTENT_SCENE, O.ABSOLUTE_T, 7650, CRFS[3])):
        clip = O.synth_clip(content, W, H, 1, T)
        ov, hv = _pair(W, H, 1, tm, dtm, crf)
        want = [ov.integrate_matrix(f) for f in clip]
        got, offs = hv.integrate_batch(clip)
        assert [int(offs[i + 1] - offs[i]) for i in range(T)] == [len(w) for w in want]
        assert np.array_equal(got, np.concatenate(want))
        hv.close()
