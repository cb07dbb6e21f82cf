"""Pins the CPU oracle to the reference's own unit-test known answers.

Each test restates the *inputs and asserted values* of one test in
adder-codec-rs/src/transcoder/event_pixel_tree.rs:534-1259 (the reference's 13
PixelArena unit tests) and checks oracle/adder_oracle.c against them.
"""
import numpy as np
import pytest

from oracle import oracle as O

EPS = np.float32(1.1920929e-7)


def slack(a, b):  # f32_slack, event_pixel_tree.rs:1005-1011
    return np.float32(b) - EPS <= np.float32(a) <= np.float32(b) + EPS


def ulps(a, b):
    a = np.float32(a).view(np.int32).astype(np.int64)
    b = np.float32(b).view(np.int32).astype(np.int64)
    return abs(int(a) - int(b))


def make_tree():  # :541-639
    dtm = 10_000
    t = O.Pixel(100.0)
    t.time_mode(O.DELTA_T)
    assert t.node(0)["d"] == 6
    t.integrate(100.0, 20.0, O.CONTINUOUS, dtm, 20, 0, 255, O.NORMAL)
    n0 = t.node(0)
    assert n0["has_best"] and n0["best_d"] == 6 and int(n0["best_delta_t"]) == 12
    assert n0["d"] == 7 and slack(n0["integration"], 100.0) and slack(n0["delta_t"], 20.0) and n0["alt"]
    n1 = t.node(1)
    assert not n1["has_best"] and n1["d"] == 6 and n1["integration"] == 36.0
    assert ulps(n1["delta_t"], 7.2) <= 2
    t.integrate(100.0, 20.0, O.CONTINUOUS, dtm, 20, 0, 255, O.NORMAL)
    n0, n1, n2 = t.node(0), t.node(1), t.node(2)
    assert n0["best_d"] == 7 and ulps(n0["best_delta_t"], 25.6) <= 1
    assert n0["d"] == 8 and slack(n0["integration"], 200.0) and slack(n0["delta_t"], 40.0) and n0["alt"]
    assert n1["d"] == 7 and slack(n1["integration"], 72.0) and ulps(n1["delta_t"], 14.4) <= 1
    assert n1["best_d"] == 6 and ulps(n1["best_delta_t"], 12.8) <= 2 and n1["alt"]
    assert n2["d"] == 6 and not n2["has_best"] and not n2["alt"] and slack(n2["integration"], 8.0)
    assert abs(float(n2["delta_t"]) - 1.6) <= 0.2e-5
    return t


def make_tree2():  # :641-709
    dtm = 10_000
    t = make_tree()
    t.integrate(30.0, 34.0, O.CONTINUOUS, dtm, 34, 0, 255, O.NORMAL)
    n0, n1, n2 = t.node(0), t.node(1), t.node(2)
    assert n0["d"] == 8 and slack(n0["integration"], 230.0) and slack(n0["delta_t"], 74.0)
    assert n1["d"] == 7 and slack(n1["integration"], 102.0) and slack(n1["delta_t"], 48.4)
    assert n2["d"] == 6 and slack(n2["integration"], 38.0) and slack(n2["delta_t"], 35.6)
    t.integrate(26.0, 34.0, O.CONTINUOUS, dtm, 34, 0, 255, O.NORMAL)
    n0, n1 = t.node(0), t.node(1)
    assert n0["d"] == 9 and slack(n0["integration"], 256.0) and slack(n0["delta_t"], 108.0)
    assert n0["best_d"] == 8 and n0["best_delta_t"] == 108.0
    assert n1["d"] == 4 and slack(n1["integration"], 0.0) and slack(n1["delta_t"], 0.0)
    assert not n1["has_best"] and not n1["alt"]
    return t


def test_make_tree():
    make_tree()


def test_make_tree2():
    make_tree2()


def test_pop_best_states():  # :721-741
    t = make_tree()
    ev = t.pop_best_events(O.CONTINUOUS, O.NORMAL, 20, 0.0)
    assert [(int(e["d"]), int(e["t"])) for e in ev] == [(7, 25), (6, 12)]
    n0 = t.node(0)
    assert n0["d"] == 6 and slack(n0["integration"], 8.0) and abs(float(n0["delta_t"]) - 1.6) <= 0.2e-5


def test_pop_best_states2():  # :743-755
    t = make_tree2()
    ev = t.pop_best_events(O.CONTINUOUS, O.NORMAL, 34, 0.0)
    assert [(int(e["d"]), int(e["t"])) for e in ev] == [(8, 108)]
    n0 = t.node(0)
    assert n0["d"] == 4 and slack(n0["integration"], 0.0) and slack(n0["delta_t"], 0.0)


def test_d_max():  # :757-794
    dtm = 100_000_000
    big = float(np.float32(2.0 ** 126))
    t = O.Pixel(big)
    t.integrate(float(np.float32(2.0 ** 126) + np.float32(5.0)), 100_000.0, O.CONTINUOUS, dtm, 100_000, 0, 255, O.NORMAL)
    assert t.need_to_pop_top
    ev = t.pop_best_events(O.CONTINUOUS, O.NORMAL, 100_000, 0.0)
    assert not t.need_to_pop_top
    assert [(int(e["d"]), int(e["t"])) for e in ev] == [(126, 100_000)]
    assert slack(t.node(0)["integration"], 0.0)


def test_dtm():  # :796-834
    dtm = 240_000
    t = O.Pixel(245.0)
    for _ in range(48):
        t.integrate(245.0, 5_000.0, O.FRAME_PERFECT, dtm, 5_000, 0, 255, O.NORMAL)
    assert t.need_to_pop_top
    t.pop_top_event(245.0, O.FRAME_PERFECT, 5_000)
    assert not t.need_to_pop_top
    assert t.node(0)["delta_t"] == 70_000.0


def test_new_dtm():  # :836-925
    dtm = 2_000
    t = O.Pixel(245.0)
    t.integrate(245.0, 1_000.0, O.FRAME_PERFECT, dtm, 5_000, 0, 255, O.NORMAL)
    assert not t.need_to_pop_top
    t.integrate(245.0, 1_000.0, O.FRAME_PERFECT, dtm, 5_000, 0, 255, O.NORMAL)
    assert t.need_to_pop_top
    t.pop_top_event(245.0, O.FRAME_PERFECT, 5_000)
    assert not t.need_to_pop_top
    for _ in range(48):
        t.integrate(245.0, 1_000.0, O.FRAME_PERFECT, dtm, 5_000, 0, 255, O.NORMAL)
    assert not t.need_to_pop_top
    assert t.node(0)["delta_t"] == 48000.0
    t.pop_best_events(O.FRAME_PERFECT, O.COLLAPSE, 5_000, 0.0)
    t.integrate(600.0, 3_000.0, O.FRAME_PERFECT, dtm, 5_000, 0, 255, O.NORMAL)
    assert t.need_to_pop_top


def test_big_integration():  # :927-966
    dtm = 1_000_000
    t = O.Pixel(146.0)
    t.integrate(146.0, 2_000.0, O.CONTINUOUS, dtm, 2_000, 0, 255, O.NORMAL)
    t.integrate(float(np.float32(2_790.863)), 38231.0, O.CONTINUOUS, dtm, 38231, 0, 255, O.NORMAL)
    h = t.node(0)
    assert h["integration"] == np.float32(2_790.863) + np.float32(146.0)
    assert h["delta_t"] == np.float32(38231.0) + np.float32(2_000.0)
    assert h["best_d"] == h["d"] - 1


def test_big_integration2():  # :968-1003
    dtm = 10_000_000
    t = O.Pixel(255.0)
    for _ in range(100_000):
        t.integrate(255.0, 2_000.0, O.CONTINUOUS, dtm, 2_000, 0, 255, O.NORMAL)
        if t.need_to_pop_top:
            break
    h = t.node(0)
    assert h["integration"] == np.float32(1.275e6)
    assert h["delta_t"] == np.float32(dtm)
    assert h["best_d"] == h["d"] - 1


def test_paper_example():  # :1020-1060
    dtm = 10_000
    t = O.Pixel(101.0)
    assert t.node(0)["d"] == 6
    t.integrate(101.0, 20.0, O.CONTINUOUS, dtm, 20, 0, 255, O.NORMAL)
    assert t.node(0)["has_best"]
    t.integrate(40.0, 30.0, O.CONTINUOUS, dtm, 30, 0, 255, O.NORMAL)
    assert t.node(0)["best_d"] == 7
    assert slack(t.node(1)["delta_t"], 9.75)


def _four(time_mode, last):
    dtm = 10_000
    t = O.Pixel(101.0)
    t.time_mode(time_mode)
    assert t.node(0)["d"] == 6
    t.integrate(101.0, 20.0, O.CONTINUOUS, dtm, 20, 0, 255, O.NORMAL)
    assert t.node(0)["has_best"]
    t.integrate(40.0, 30.0, O.CONTINUOUS, dtm, 30, 0, 255, O.NORMAL)
    t.integrate(140.0, 30.0, O.CONTINUOUS, dtm, 30, 0, 255, O.NORMAL)
    t.integrate(last, 30.0, O.CONTINUOUS, dtm, 30, 0, 255, O.NORMAL)
    return t


def test_absolute_mode_1():  # :1062-1126
    t = _four(O.ABSOLUTE_T, 103.0)
    ev = t.pop_best_events(O.CONTINUOUS, O.COLLAPSE, 30, 0.0)
    assert (int(ev[0]["d"]), int(ev[0]["t"])) == (8, 74)
    assert (int(ev[1]["d"]), int(ev[1]["t"])) == (7, 110)


def test_set_d_continuous_delta():  # :1128-1192
    t = _four(O.DELTA_T, 107.0)
    t.pop_best_events(O.CONTINUOUS, O.COLLAPSE, 30, 0.0)
    ev = t.set_d_for_continuous(10.0, 30)
    assert ev is not None and int(ev["t"]) == 1 and int(ev["d"]) == 255


def test_set_d_continuous_absolute():  # :1194-1258
    t = _four(O.ABSOLUTE_T, 107.0)
    t.pop_best_events(O.CONTINUOUS, O.COLLAPSE, 30, 0.0)
    ev = t.set_d_for_continuous(10.0, 30)
    assert ev is not None and int(ev["t"]) == 110 and int(ev["d"]) == 255
