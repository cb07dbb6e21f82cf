"""Randomised GPU-vs-oracle parity soak (not a test: run it for as long as you like on the GPU box).
Small planes, contents that mix static stretches, jitter inside and outside the contrast band, black regions, scene cuts and
noise; Collapse / Normal, DeltaT / AbsoluteT, delta_t_max 255 / 1020 / 7650, crf 0 / 3 / 6 / 9 numbers, gray / RGB, random batch
lengths and launch depths, some batches with an event buffer that is too small (rollback + retry).
usage: python tools/fuzz_parity.py [seconds] [seed]   (FUZZ_CRF0=1: crf 0 only; FUZZ_CONTINUOUS=0/1)"""
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
    if os.environ.get("FUZZ_CRF0") == "1":  # the integer-state kernels (lean runs, run records): crf 0 only
        crf = 0
    depth = int(rng.choice([1, 3, 16, 64, 64]))
    cont = os.environ.get("FUZZ_CONTINUOUS") == "1" or (os.environ.get("FUZZ_CONTINUOUS") is None and rng.random() < 0.12)
    clip = make_clip(rng, T, H, W, C)
    ov = O.Video(W, H, C, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm)
    if cont:  # Mode::Continuous (the event-camera sources' mode, fed frames): the general arena step
        ov.set_pixel_mode(1)
        hv = A.HipVideo(W, H, C, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm, max_depth=24, pixel_mode=1)
    else:
        hv = A.HipVideo(W, H, C, time_mode=tm, multi_mode=mm, ref_time=255, delta_t_max=dtm, max_depth=24)
    ov.ensure_capacity(30)
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
            try:
                got, _ = hv.integrate_batch(clip[k:k + nb], out_cap=len(want))
            except A.AdderHipError as e:
                raise SystemExit(f"RETRY FAILED ({e}) want {len(want)} required {hv.last_required}: W{W} H{H} C{C} T{T} tm{tm} mm{mm} "
                                 f"dtm{dtm} crf{crf} depth{depth} at frame {k}+{nb}")
        else:
            try:
                got, _ = hv.integrate_batch(clip[k:k + nb])
            except A.AdderHipError as e:
                # the plumbing's default buffer (4 events per unit and frame) can be too small for a Normal-mode scene cut
                # that flushes deep arenas: reported with the size needed, state rolled back -- retry like a caller would
                if e.code != A.E_OUT_CAPACITY or hv.last_required != len(want):
                    raise SystemExit(f"DEFAULT-CAPACITY BATCH FAILED ({e}) want {len(want)}: W{W} H{H} C{C} T{T} tm{tm} mm{mm} "
                                     f"dtm{dtm} crf{crf} depth{depth} at frame {k}+{nb}")
                got, _ = hv.integrate_batch(clip[k:k + nb], out_cap=hv.last_required)
        if len(got) != len(want) or not np.array_equal(got, want):
            raise SystemExit(f"MISMATCH W{W} H{H} C{C} T{T} tm{tm} mm{mm} dtm{dtm} crf{crf} depth{depth} at frame {k}+{nb}")
        k += nb
    hv.close()
    return W * H * C * T

def one_records(rng):
    """Records over the wire on one GPU: a random plane in 1..18 bands (contexts), random chunk lengths; every band
    hands out its records, band 0 expands them all -- against one context over the whole plane."""
    import torch
    from adder_amd import sharding
    W, H = int(rng.choice([5, 33, 64, 130, 256])), int(rng.integers(2, 40))
    C = int(rng.choice([1, 1, 3])); T = int(rng.integers(5, 150))
    tm = int(rng.choice([O.DELTA_T, O.ABSOLUTE_T])); crf = int(rng.choice([0, 0, 3, 9]))
    nb = int(min(H, rng.choice([1, 2, 3, 5, 8, 16, 17, 18])))
    clip = make_clip(rng, T, H, W, C)
    base, cmax, vel = CRF[crf]
    kw = dict(time_mode=tm, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255, c_thresh_start=base, c_counter_start=0)
    st = torch.cuda.current_stream().cuda_stream
    whole = A.HipVideo(W, H, C, **kw); whole.set_crf_parameters(cmax, vel)
    d_all = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_ev = torch.empty((d_all.numel() * 3 + 64, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    whole.integrate_device(d_all, d_ev, d_off, stream=st)
    n_want = whole.finish()
    if rng.random() < 0.5:
        bands = sharding.row_bands(H, nb)
    else:
        bands = sharding.row_bands_root_heavy(H, nb, float(rng.uniform(0.02, 1.0 / nb)))
    bands = [b for b in bands if b[1] > b[0]]
    ctxs, fr = [], []
    for (y0, y1) in bands:
        hv = A.HipVideo(W, H, C, row_begin=y0, row_end=y1, **kw); hv.set_crf_parameters(cmax, vel); ctxs.append(hv)
        fr.append(torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda())
    d_m = torch.full((n_want + 8, 3), -1, dtype=torch.int32, device="cuda")
    d_mo = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    boffs = [torch.zeros(65, dtype=torch.int64, device="cuda") for _ in bands]
    f0, base_ev = 0, 0
    while f0 < T:
        nf = min(int(rng.choice([1, 3, 16, 37, 64])), T - f0)
        recs = []
        for r, hv in enumerate(ctxs):
            recs.append(hv.integrate_records_device(fr[r][f0:f0 + nf], boffs[r], stream=st)); hv.finish()
        ctxs[0].expand_records_device(recs, d_m, base_ev, d_mo[f0:], stream=st)
        ctxs[0].expand_status(st)
        base_ev = int(d_mo[f0 + nf].item())
        f0 += nf
    ok = base_ev == n_want and torch.equal(d_mo, d_off) and torch.equal(d_m[:n_want], d_ev[:n_want]) and int((d_m[n_want:] != -1).sum()) == 0
    if not ok:
        raise SystemExit(f"RECORDS MISMATCH W{W} H{H} C{C} T{T} tm{tm} crf{crf} bands{bands}")
    for hv in ctxs + [whole]:
        hv.close()
    return W * H * C * T


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    t0, n, units = time.time(), 0, 0
    records = os.environ.get("FUZZ_RECORDS") == "1"
    while time.time() - t0 < secs:
        units += one_records(rng) if records else one(rng); n += 1
    print(f"fuzz_parity: {n} clips, {units} unit-frames, seed {seed}: all bit-exact")
