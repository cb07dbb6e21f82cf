"""Stress of the batch submission path: tens of thousands of SHORT batches, two contexts taking turns, two output forms.

Round 5 saw, twice in eight full runs and never again, a 3-frame batch leave its offsets untouched "as if it had run under the
previous batch's description" while a 4-byte memset sat in front of the description's host-to-device copy.  Every batch here
has its frame offsets and its event count asserted against the oracle's, so a batch that runs under a stale description --
another length, another frame table, another output buffer -- cannot pass: its offsets would be the previous batch's or its
events would land in the other buffer.  (video.rs:651-778: one integrate_matrix per frame, nothing carried between calls but
the pixel state.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
import clips  # noqa: E402

pytestmark = pytest.mark.gpu


def _oracle_counts_and_events(clip, tm, dtm, crf0=True):
    T, H, W, Cn = clip.shape
    ov = O.Video(W, H, Cn, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
    ov.ensure_capacity(24)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    evs = [ov.integrate_matrix(f) for f in clip]
    return np.array([len(e) for e in evs], np.int64), evs


@pytest.mark.parametrize("batches", [int(os.environ.get("ADDER_STRESS_BATCHES", "20000"))])
def test_twenty_thousand_short_batches_two_contexts_two_output_forms(batches):
    import torch
    import adder_amd as A
    rng = np.random.default_rng(2026)
    W, H = 96, 24
    frames = 1200  # (the contexts are reset and the clip starts over when it runs out)
    clip_a = clips.make_clip("runs", frames, H, W, 1, seed=3)
    clip_a[300:420] = clip_a[300]          # a static stretch: batches without a single event
    clip_b = clips.make_clip("jitter", frames, H, W, 1, seed=4)
    cnt_a, ev_a = _oracle_counts_and_events(clip_a, O.DELTA_T, 255)
    cnt_b, ev_b = _oracle_counts_and_events(clip_b, O.ABSOLUTE_T, 7650)
    st = torch.cuda.current_stream().cuda_stream
    ctx = []
    for clip, tm, dtm, cnt, evs in ((clip_a, A.TIME_DELTA_T, 255, cnt_a, ev_a), (clip_b, A.TIME_ABSOLUTE_T, 7650, cnt_b, ev_b)):
        # (pixels constructed at c_thresh 0 / counter 0 -- what `.crf(0)` leaves -- so that reset() restores exactly that)
        hv = A.HipVideo(W, H, 1, time_mode=tm, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=dtm, max_depth=24,
                        c_thresh_start=0, c_counter_start=0)
        hv.set_crf_parameters(0, 10)
        d_frames = torch.from_numpy(clip.reshape(frames, -1)).cuda()
        cap = 5 * 4 * W * H
        ctx.append(dict(hv=hv, d_frames=d_frames, cnt=cnt, evs=evs, k=0,
                        d_ev=torch.zeros((cap, 3), dtype=torch.int32, device="cuda"),
                        d_wire=torch.zeros(cap * 9 + 16, dtype=torch.uint8, device="cuda"),
                        d_off=torch.full((8,), -1, dtype=torch.int64, device="cuda")))
    full_checks = 0
    for i in range(batches):
        c = ctx[i & 1] if rng.random() < 0.9 else ctx[int(rng.integers(0, 2))]   # mostly taking turns, sometimes twice in a row
        nb = int(rng.integers(1, 6))
        if c["k"] + nb > frames:
            c["hv"].reset()
            c["k"] = 0
        k, hv = c["k"], c["hv"]
        want = c["cnt"][k:k + nb]
        wire = bool(rng.integers(0, 2))
        c["d_off"].fill_(-1)               # a batch that does not write its offsets shows
        if wire:
            hv.integrate_wire_device(c["d_frames"][k:k + nb], c["d_wire"], c["d_off"], stream=st)
        else:
            hv.integrate_device(c["d_frames"][k:k + nb], c["d_ev"], c["d_off"], stream=st)
        n = hv.finish()
        offs = c["d_off"][:nb + 1].cpu().numpy()
        assert n == int(want.sum()), (i, k, nb, wire, n, int(want.sum()))
        assert offs[0] == 0 and np.array_equal(np.diff(offs), want), (i, k, nb, wire, offs, want)
        if i % 97 == 0 and n:              # ... and the events themselves, now and then
            exp = np.concatenate(c["evs"][k:k + nb])
            if wire:
                got = c["d_wire"][: 9 * n].cpu().numpy().tobytes()
                assert got == O.raw_events(exp, 1), (i, k, nb)
            else:
                got = np.frombuffer(c["d_ev"][:n].cpu().numpy().tobytes(), dtype=O.EVENT_DTYPE)
                assert np.array_equal(got, exp), (i, k, nb)
            full_checks += 1
        c["k"] = k + nb
    assert full_checks > batches // 200
    for c in ctx:
        c["hv"].close()


def test_a_null_stream_batch_is_ordered_behind_the_default_stream():
    """Round 5's flake, on purpose: the offsets tensor is zero-filled on the legacy default stream BEHIND a long kernel, then a
    3-frame batch is submitted with stream = NULL.  The context's own stream is non-blocking -- without the join of
    adder_hip_integrate_device the short batch would write its offsets first and the zero-fill would land on top of them
    ("a 3-frame batch left its offsets untouched").  Now the batch waits for the default stream's work queued so far."""
    import torch
    import adder_amd as A
    W, H, nb = 96, 24, 3
    clip = clips.make_clip("runs", nb, H, W, 1, seed=8)
    cnt, _ = _oracle_counts_and_events(clip, O.DELTA_T, 255)
    hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255,
                    c_thresh_start=0, c_counter_start=0)
    hv.set_crf_parameters(0, 10)
    d_frames = torch.from_numpy(clip.reshape(nb, -1)).cuda()
    d_ev = torch.zeros((4 * W * H * nb, 3), dtype=torch.int32, device="cuda")
    big = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    for rep in range(20):
        hv.reset()
        for _ in range(6):
            big = (big @ big).clamp_(-1.0, 1.0)          # tens of milliseconds of work on the default stream ...
        d_off = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")   # ... and the zero-fill queued behind it
        hv.integrate_device(d_frames, d_ev, d_off, stream=None)
        n = hv.finish()
        offs = d_off.cpu().numpy()
        assert n == int(cnt.sum()) and np.array_equal(np.diff(offs), cnt), (rep, n, offs, cnt)
    hv.close()


def test_the_round_5_flake_is_that_race():
    """The same arrangement with the join switched off (ADDER_HIP_DBG_NO_NULL_JOIN=1, a child process): the zero-fill lands on top
    of the batch's offsets -- the symptom round 5 recorded.  (Shown, not relied upon: a race may also not happen.)"""
    import subprocess
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import adder_amd as A
W, H, nb = 96, 24, 3
rng = np.random.default_rng(1)
clip = rng.integers(1, 255, (nb, H * W), dtype=np.uint8)
hv = A.HipVideo(W, H, 1, time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
hv.set_crf_parameters(0, 10)
d_frames = torch.from_numpy(clip).cuda(); d_ev = torch.zeros((4 * W * H * nb, 3), dtype=torch.int32, device="cuda")
big = torch.randn(4096, 4096, device="cuda"); torch.cuda.synchronize()
lost = 0
for rep in range(10):
    hv.reset()
    for _ in range(6): big = (big @ big).clamp_(-1.0, 1.0)
    d_off = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
    hv.integrate_device(d_frames, d_ev, d_off, stream=None)
    n = hv.finish()
    if n and int(d_off.cpu()[-1]) == 0: lost += 1
print("LOST", lost)
""" % (ROOT, os.path.join(ROOT, "adder-codec-rs_amd"))
    env = dict(os.environ, ADDER_HIP_DBG_NO_NULL_JOIN="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lost = int(out.stdout.strip().split("LOST")[-1])
    print("batches whose offsets the late zero-fill overwrote, without the join:", lost, "of 10")
    assert lost >= 1, "the race did not show in 10 tries (it is a race) -- the join is the fix either way"
