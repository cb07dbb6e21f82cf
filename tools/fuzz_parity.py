"""Randomised GPU-vs-oracle parity soak (not a test: run it for as long as you like on the GPU box).
Small planes, contents that mix static stretches, jitter inside and outside the contrast band, black regions, scene cuts and
noise; Collapse / Normal, DeltaT / AbsoluteT, delta_t_max 255 / 1020 / 7650, crf 0 / 3 / 6 / 9 numbers, gray / RGB, random batch
lengths and launch depths, some batches with an event buffer that is too small (rollback + retry).
usage: python tools/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np
import adder_amd as A
from oracle import oracle as O

CRF = {0: (0, 0, 10), 3: (2, 7, 7), 6: (7, 13, 4), 9: (15, 25, 1)}

def make_clip(rng, T, H, W, C):
    base = rng.integers(0, 256, (1, H, W, C))
    if rng.random() < 0.5:
        base[0, : max(1, H // 4)] = 0
    clip = np.repeat(base, T, axis=0).astype(np.int64)
    amp = int(rng.choice([0, 0, 1, 1, 2, 5]))
    if amp:
        y0 = int(rng.integers(0, H))
        clip[:, y0:] += rng.integers(-amp, amp + 1, (T, H - y0, W, C))
    for _ in range(int(rng.integers(0, 4))):          # scene cuts / local flips
        k = int(rng.integers(1, T)); y = int(rng.integers(0, H)); x = int(rng.integers(0, W))
        clip[k:, y:y + int(rng.integers(1, H + 1)), x:x + int(rng.integers(1, W + 1))] = int(rng.integers(0, 256))
    if rng.random() < 0.3:
        k = int(rng.integers(0, T)); clip[k:k + int(rng.integers(1, 12))] = rng.integers(0, 256, clip[k:k + 1].shape)
    return np.clip(clip, 0, 255).astype(np.uint8)

def one(rng):
    W, H = int(rng.choice([7, 33, 64, 128, 256, 300])), int(rng.integers(1, 24))
    C = int(rng.choice([1, 1, 3])); T = int(rng.integers(20, 220))
    tm = int(rng.choice([O.DELTA_T, O.ABSOLUTE_T])); mm = int(rng.choice([O.COLLAPSE, O.COLLAPSE, O.COLLAPSE, O.NORMAL]))
    dtm = int(rng.choice([255, 7650, 7650, 1020])); crf = int(rng.choice([0, 3, 3, 6, 9]))
    depth = int(rng.choice([1, 3, 16, 64, 64]))
    clip = make_clip(rng, T, H, W, C)
    ov = O.Video(W, H, C, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm)
    hv = A.HipVideo(W, H, C, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm, max_depth=24)
    ov.ensure_capacity(26)
    base, cmax, vel = CRF[crf]
    for v in (ov, hv):
        v.set_crf_parameters(cmax, vel); v.reset_c_thresh(base)
    hv.set_frames_per_launch(depth)
    k = 0
    while k < T:
        nb = min(int(rng.choice([1, 2, 17, 64, 65, 130])), T - k)
        want = np.concatenate([ov.integrate_matrix(clip[k + i]) for i in range(nb)])
        if len(want) > 2 and rng.random() < 0.15:     # too small a buffer first: rollback, then the retry
            try:
                hv.integrate_batch(clip[k:k + nb], out_cap=len(want) // 2)
                raise SystemExit("overflow not reported")
            except A.AdderHipError as e:
                assert e.code == A.E_OUT_CAPACITY and hv.last_required == len(want), (e.code, hv.last_required, len(want))
            got, _ = hv.integrate_batch(clip[k:k + nb], out_cap=len(want))
        else:
            got, _ = hv.integrate_batch(clip[k:k + nb])
        if len(got) != len(want) or not np.array_equal(got, want):
            raise SystemExit(f"MISMATCH W{W} H{H} C{C} T{T} tm{tm} mm{mm} dtm{dtm} crf{crf} depth{depth} at frame {k}+{nb}")
        k += nb
    hv.close()
    return W * H * C * T

if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    t0, n, units = time.time(), 0, 0
    while time.time() - t0 < secs:
        units += one(rng); n += 1
    print(f"fuzz_parity: {n} clips, {units} unit-frames, seed {seed}: all bit-exact")
