"""Randomised parity: hypothesis draws the plane, the modes, the rate-control numbers, the batch split and
the clip; the HIP path (through the C-ABI) must produce the oracle's events, offsets and chunk offsets."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import oracle as O
import clips

pytestmark = pytest.mark.gpu

KINDS = ["noise", "static", "dark", "jitter", "runs"]


@st.composite
def cases(draw):
    W = draw(st.integers(1, 300))
    H = draw(st.integers(1, 40))
    C = draw(st.sampled_from([1, 1, 3]))
    T = draw(st.integers(1, 70))
    ref = draw(st.sampled_from([255, 255, 100, 1000, 5000]))
    dtm = ref * draw(st.sampled_from([1, 1, 2, 4, 30]))
    return dict(
        W=W, H=H, C=C, T=T, ref=ref, dtm=dtm,
        time_mode=draw(st.sampled_from([O.DELTA_T, O.ABSOLUTE_T])),
        multi_mode=draw(st.sampled_from([O.COLLAPSE, O.NORMAL])),
        crf=(draw(st.integers(0, 20)), draw(st.integers(0, 30)), draw(st.integers(1, 12))),
        default_pixels=draw(st.booleans()),
        kind=draw(st.sampled_from(KINDS)), seed=draw(st.integers(0, 2**31 - 1)),
        split=draw(st.integers(1, 70)), chunk_rows=draw(st.sampled_from([1, 1, 3, 64])),
    )


@settings(max_examples=300, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(cases())
def test_random_configurations_match_the_oracle(c):
    import adder_amd as A
    W, H, C, T = c["W"], c["H"], c["C"], c["T"]
    clip = clips.make_clip(c["kind"], T, H, W, C, seed=c["seed"])
    ov = O.Video(W, H, C, time_mode=c["time_mode"], multi_mode=c["multi_mode"], ref_time=c["ref"], delta_t_max=c["dtm"],
                 chunk_rows=c["chunk_rows"])
    hv = A.HipVideo(W, H, C, time_mode=c["time_mode"], multi_mode=c["multi_mode"], ref_time=c["ref"],
                    delta_t_max=c["dtm"], max_depth=24, chunk_rows=c["chunk_rows"])
    ov.ensure_capacity(26)
    base, cmax, vel = c["crf"]
    ov.set_crf_parameters(cmax, vel)
    hv.set_crf_parameters(cmax, vel)
    if not c["default_pixels"]:
        ov.reset_c_thresh(base)
        hv.reset_c_thresh(base)
    k = 0
    first = True
    while k < T:
        n = min(c["split"], T - k)
        sub = clip[k:k + n]
        if first and n == 1:  # also through the per-frame entry point, with chunk offsets
            a, ca = ov.integrate_matrix(sub[0], time_spanned=float(c["ref"]), want_chunks=True)
            b, cb = hv.integrate_matrix(sub[0], want_chunks=True)
            assert np.array_equal(a, b) and np.array_equal(ca, cb)
        else:
            want = [ov.integrate_matrix(f, time_spanned=float(c["ref"])) for f in sub]
            got, offs = hv.integrate_batch(sub, time_spanned=float(c["ref"]))
            assert [int(offs[i + 1] - offs[i]) for i in range(n)] == [len(w) for w in want]
            assert np.array_equal(got, np.concatenate(want))
        first = False
        k += n
