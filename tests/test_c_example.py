"""examples/transcode_raw.c: the C-ABI from plain C.  CPU: it compiles and links against the library;
GPU: it runs, and the `.adder` file it writes decodes to the events the oracle produces for the same clip."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "adder-codec-rs_amd")


def _build(tmp_path):
    import adder_amd
    adder_amd.load()  # makes sure the library exists
    exe = str(tmp_path / "transcode_raw")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "transcode_raw.c"), "-L", LIBDIR, "-ladder_hip",
                           "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def test_c_example_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_example_runs_and_matches_the_oracle(tmp_path):
    from oracle import oracle as O
    import adder_stream_np as S
    exe = _build(tmp_path)
    ev_path, fr_path = str(tmp_path / "out.adder"), str(tmp_path / "out.gray")
    r = subprocess.run([exe, ev_path, fr_path], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    W, H, T = 320, 180, 48
    k, y, x = np.meshgrid(np.arange(T), np.arange(H), np.arange(W), indexing="ij")
    clip = (((x + 2 * k) * 255 // W + y) & 255).astype(np.uint8)
    sq = (x > 100) & (x < 140) & (y > 60) & (y < 100)
    clip[sq] = np.where((k[sq] // 6) % 2 == 1, 250, 5).astype(np.uint8)
    ov = O.Video(W, H, 1, time_mode=O.ABSOLUTE_T, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255 * 8)
    ov.set_crf_parameters(7, 7)
    ov.reset_c_thresh(2)
    want = np.concatenate([ov.integrate_matrix(f) for f in clip])
    meta, got, closed = S.read_adder(open(ev_path, "rb").read())
    assert closed and (meta["width"], meta["height"], meta["tps"], meta["delta_t_max"]) == (W, H, 7650, 2040)
    assert np.array_equal(got, want)
    rec = np.fromfile(fr_path, np.uint8)
    assert rec.size % (W * H) == 0 and rec.size >= W * H
