"""Long-stream soak of the integer-state kernels' run bound (BatchResult::max_run): streams of 66 000 - 72 000 frames on a small
plane, every pixel changing with a period of its own, some pixels going static at a random frame (their run then crosses
65 793 frames at a random point of the stream, or not at all), random batch lengths, both time modes and regimes -- GPU events ==
oracle events batch by batch, and the kernel that ran is logged.  usage: python tools/probes/long_stream_fuzz.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
import numpy as np
import adder_amd as A
from oracle import oracle as O

cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_start = time.time()
for case in range(cases):
    W, H = int(rng.choice([128, 160, 256])), int(rng.integers(1, 3))
    T = int(rng.integers(66000, 72000))
    tm = int(rng.choice([O.DELTA_T, O.DELTA_T, O.ABSOLUTE_T]))
    dtm = int(rng.choice([255, 7650]))
    period = rng.integers(200, 6000, (H, W, 1))
    phase = rng.integers(0, 6000, (H, W, 1))
    vals = rng.integers(0, 256, (64, H, W, 1)).astype(np.uint8)
    vals[:, 0, :4] = 0                                    # a few black pixels (their run counts one frame)
    n_static = int(rng.choice([0, 0, 1, 3]))
    static_from = {}
    for _ in range(n_static):
        static_from[(int(rng.integers(0, H)), int(rng.integers(4, W)))] = int(rng.integers(0, 8000))
    clip = np.empty((T, H, W, 1), np.uint8)
    for k in range(T):
        step = (k + phase) // period
        clip[k] = np.take_along_axis(vals, (step % 64)[None], axis=0)[0]
    for (y, x), f0 in static_from.items():
        clip[f0:, y, x] = clip[f0, y, x] if clip[f0, y, x, 0] != 0 else 77
    ov = O.Video(W, H, 1, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm)
    hv = A.HipVideo(W, H, 1, time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=dtm, max_depth=24)
    ov.ensure_capacity(26)
    for v in (ov, hv):
        v.set_crf_parameters(0, 10)
        v.reset_c_thresh(0)
    k, kernels, total = 0, [], 0
    while k < T:
        nb = min(int(rng.choice([1, 7, 64, 300, 2000, 5000])), T - k)
        want = np.concatenate([ov.integrate_matrix(f) for f in clip[k:k + nb]])
        got, offs = hv.integrate_batch(clip[k:k + nb])
        if not (len(got) == len(want) and np.array_equal(got, want)):
            print(f"MISMATCH case {case} seed {seed}: W {W} H {H} T {T} tm {tm} dtm {dtm} static {static_from} at frame {k} (+{nb})")
            sys.exit(1)
        if not kernels or kernels[-1][1] != hv.last_batch_kernel():
            kernels.append((k, hv.last_batch_kernel()))
        total += len(got)
        k += nb
    hv.close()
    print(f"case {case}: {W}x{H} T {T} tm {tm} dtm {dtm} static {sorted(static_from.values())} events {total} kernels {[(f, A.KERNEL_NAMES[q]) for f, q in kernels]}", flush=True)
print(f"long_stream_fuzz: {cases} streams bit-exact, seed {seed}, {time.time() - t_start:.0f} s")
