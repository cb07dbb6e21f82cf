"""The multi-rank protocols of libadder_rccl.so (include/adder_gather.h) with MORE THAN ONE RANK on a one-GPU box.

RCCL refuses two ranks on one device, so the ranks here are threads of this process, one HipVideo + HipGather each, over
the library's in-process transport (adder_gather_local_*: blocking rendezvous + device copies).  Everything above the
transport is the production code: the streamed records gather (adder_gather_records_begin / _push / _end: sizes of chunk
k gathered while chunk k-1's payload moves, no host wait per chunk), the chunked events gather with its agreement round,
and the sink per rank (adder_gather_host_sink_*: every rank stores its own wire bytes into the one .adder image).
Reference: the row split and ordered concatenation of video.rs:677-691,742-765; the raw sink of raw/stream.rs:101-120."""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "adder-codec-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _eager_contexts(monkeypatch):
    """The ranks are THREADS here: a context that captures its launch graph (hipStreamBeginCapture) while another thread
    allocates or copies on the legacy stream has its capture invalidated by this HIP runtime.  One process per GPU -- the
    product's arrangement -- never meets that; the threaded harness submits eagerly (read at adder_hip_create)."""
    monkeypatch.setenv("ADDER_HIP_NO_GRAPH", "1")


def _hip():
    import adder_amd as A
    return A


def _run_ranks(world, fn):
    """fn(rank) on `world` threads; re-raises the first failure (the local transport times out instead of hanging)."""
    errs, outs = [None] * world, [None] * world

    def body(r):
        try:
            outs[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            errs[r] = e

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    for e in errs:
        if e is not None:
            raise e
    return outs


def _whole_plane(A, clip, W, H, Cn, kw):
    import torch
    T = clip.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    whole = A.HipVideo(W, H, Cn, **kw)
    whole.set_crf_parameters(0, 10)
    d_all = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_ev = torch.empty((int(d_all.numel() * 1.3) + 1024, 3), dtype=torch.int32, device="cuda")
    d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    whole.integrate_device(d_all, d_ev, d_off, stream=st)
    n = whole.finish()
    ev, off = d_ev[:n].clone(), d_off.clone()
    whole.close()
    del d_ev, d_all
    return ev, off


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("W,H,Cn,world,T,chunk", [(333, 41, 1, 3, 150, 37), (50, 24, 3, 4, 150, 64), (3840, 2160, 1, 8, 70, 64)])
def test_streamed_records_gather_writes_the_raw_sinks_records_on_root(time_mode, W, H, Cn, world, T, chunk, monkeypatch):
    """adder_gather_records_begin_wire / _push / _end: the bands run the lean-runs kernel and ship {rho, word} records, root
    expands every band straight into 9 / 11-byte raw-sink records (expansion format 6, WIRE) -- the multi-GPU path does what
    the single-GPU headline does.  Root's bytes == adder_hip_integrate_wire_device's of the whole plane, and one band's
    records picked out of them == O.raw_events over the oracle's events of that band (raw/stream.rs:101-120); 3 / 4 ranks on
    ragged and RGB planes and BASELINE config 4's 3840x2160 in 8 bands of 270 rows."""
    import torch
    A = _hip()
    from adder_amd import sharding
    from adder_amd.gather import HipGather, LocalGroup
    runs = Cn == 1   # (the RGB case keeps the lean kernel's records in logs -- expansion format 3 -- with the same output)
    if not runs:
        monkeypatch.setenv("ADDER_HIP_NO_LR", "1")
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, Cn, T)
    if W < 1000:  # quiet stretches (the bands' quiet groups), a cut, black rows
        clip[40:90] = clip[40]
        clip[:, : H // 6] = 0
        clip[120:] = 255 - clip[120:]
    kw = dict(time_mode=time_mode, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    rec = 9 if Cn == 1 else 11
    st0 = torch.cuda.current_stream().cuda_stream
    whole = A.HipVideo(W, H, Cn, **kw)
    whole.set_crf_parameters(0, 10)
    d_all = torch.from_numpy(clip.reshape(T, -1)).cuda()
    d_want = torch.zeros(int(d_all.numel() * 1.3) * rec + 64, dtype=torch.uint8, device="cuda")
    want_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
    whole.integrate_wire_device(d_all, d_want, want_off, stream=st0)
    n_want = whole.finish()
    whole.close()
    del d_all
    bands = sharding.row_bands(H, world)
    grp = LocalGroup(world)
    d_wire = torch.full((n_want * rec + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    d_moff = torch.full((T + 1,), -7, dtype=torch.int64, device="cuda")
    kinds = [None] * world

    def rank_fn(r):
        y0, y1 = bands[r]
        hv = A.HipVideo(W, H, Cn, row_begin=y0, row_end=y1, **kw)
        hv.set_crf_parameters(0, 10)
        g = HipGather(hv, None, r, world, local=grp)
        st, side = torch.cuda.Stream(), torch.cuda.Stream()
        d_fr = torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda()
        d_boff = torch.zeros(chunk + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        g.records_begin_wire(0, d_wire if r == 0 else None, 0, d_moff if r == 0 else None, stream=side.cuda_stream)
        for f0 in range(0, T, chunk):
            nf = min(chunk, T - f0)
            rc = hv.integrate_records_device(d_fr[f0:f0 + nf], d_boff, stream=st.cuda_stream)
            n_k = hv.finish()
            kinds[r] = int(rc.record_bytes)
            g.records_push(rc, hv.last_batch_records(), n_k)
        n_merged, sent = g.records_end()
        assert n_merged == (n_want if r == 0 else 0)
        g.close()
        hv.close()

    _run_ranks(world, rank_fn)
    grp.close()
    assert all(k == ((8 if time_mode == O.DELTA_T else 12) | (0x100 if runs else 0)) for k in kinds), kinds   # lean-runs records on the wire
    assert torch.equal(d_moff, want_off)
    assert torch.equal(d_wire[:n_want * rec], d_want[:n_want * rec]) and int((d_wire[n_want * rec:] != 0xAB).sum()) == 0
    # one band against the oracle's raw sink: its records picked out of the merged bytes by their y (bytes 2-3, big-endian)
    r = world // 2
    y0, y1 = bands[r]
    ov = O.Video(W, y1 - y0, Cn, row_begin=y0, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
    ov.ensure_capacity(8)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    ora = O.raw_events(np.concatenate([ov.integrate_matrix(f[y0:y1]) for f in clip]), Cn)
    recs = d_wire[:n_want * rec].cpu().numpy().reshape(n_want, rec)
    ys = recs[:, 2].astype(np.int64) * 256 + recs[:, 3]
    assert recs[(ys >= y0) & (ys < y1)].tobytes() == ora


@pytest.mark.parametrize("time_mode", [O.DELTA_T, O.ABSOLUTE_T])
@pytest.mark.parametrize("W,H,Cn,world,T,chunk", [(333, 41, 1, 3, 150, 37), (50, 24, 3, 4, 150, 64), (3840, 2160, 1, 8, 70, 64),
                                                  (3840, 2160, 1, 8, 140, 64)])   # (the last: config 4's geometry over THREE chunks)
def test_streamed_records_gather_equals_the_whole_plane_stream(time_mode, W, H, Cn, world, T, chunk):
    """adder_gather_records_begin / _push / _end over 3, 4 and 8 ranks -- the last case is BASELINE config 4's geometry,
    3840x2160 in 8 bands of 270 rows -- : root's merged stream and offsets equal the whole-plane context's byte for
    byte, one band also equals the oracle's stream of that band, the peers report the bytes they sent."""
    import torch
    A = _hip()
    from adder_amd import sharding
    from adder_amd.gather import HipGather, LocalGroup
    if T > 100 and W > 1000 and time_mode == O.ABSOLUTE_T:
        pytest.skip("the three-chunk 4K case runs in DeltaT (config 4's time mode): 4.5 GB of events per run")
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, Cn, T)
    if W < 1000:  # quiet stretches, a cut, black rows
        clip[40:90] = clip[40]
        clip[:, : H // 6] = 0
        clip[120:] = 255 - clip[120:]
    kw = dict(time_mode=time_mode, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    want_ev, want_off = _whole_plane(A, clip, W, H, Cn, kw)
    n_want = want_ev.shape[0]
    bands = sharding.row_bands(H, world)
    assert world != 8 or all(b[1] - b[0] == 270 for b in bands)
    grp = LocalGroup(world)
    d_merged = torch.full((n_want + 16, 3), -1, dtype=torch.int32, device="cuda")
    d_moff = torch.full((T + 1,), -7, dtype=torch.int64, device="cuda")
    band_counts = [None] * world

    def rank_fn(r):
        y0, y1 = bands[r]
        hv = A.HipVideo(W, H, Cn, row_begin=y0, row_end=y1, **kw)
        hv.set_crf_parameters(0, 10)
        g = HipGather(hv, None, r, world, local=grp)
        st, side = torch.cuda.Stream(), torch.cuda.Stream()
        d_fr = torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda()
        d_boff = torch.zeros(chunk + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        sent_total, n_band = 0, 0
        for rnd in range(2):  # a second clip through the same objects: reset + begin again (no agreement round this time)
            hv.reset()
            g.records_begin(0, d_merged if r == 0 else None, 0, d_moff if r == 0 else None, stream=side.cuda_stream)
            n_band = 0
            for f0 in range(0, T, chunk):
                nf = min(chunk, T - f0)
                rec = hv.integrate_records_device(d_fr[f0:f0 + nf], d_boff, stream=st.cuda_stream)
                n_k = hv.finish()
                g.records_push(rec, hv.last_batch_records(), n_k)
                n_band += n_k
            n_merged, sent = g.records_end()
            sent_total += sent
            if r == 0:
                assert n_merged == n_want, (rnd, n_merged, n_want)
            else:
                assert n_merged == 0 and 0 < sent < 12 * n_band
        band_counts[r] = n_band
        g.close()
        hv.close()
        return sent_total

    _run_ranks(world, rank_fn)
    grp.close()
    assert sum(band_counts) == n_want
    assert torch.equal(d_moff, want_off)
    assert torch.equal(d_merged[:n_want], want_ev) and int((d_merged[n_want:] != -1).sum()) == 0
    # one band against the oracle (the merged stream restricted to that band's rows IS the band's stream)
    r = world // 2
    y0, y1 = bands[r]
    ov = O.Video(W, y1 - y0, Cn, row_begin=y0, time_mode=time_mode, multi_mode=O.COLLAPSE, ref_time=255, delta_t_max=255)
    ov.ensure_capacity(8)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    ora = np.concatenate([ov.integrate_matrix(f[y0:y1]) for f in clip])
    got = np.frombuffer(d_merged[:n_want].cpu().numpy().tobytes(), dtype=O.EVENT_DTYPE)
    got = got[(got["y"] >= y0) & (got["y"] < y1)]
    assert len(ora) == band_counts[r] and np.array_equal(got, ora)


def test_streamed_records_gather_past_the_run_bound_with_a_static_band():
    """The integer-state kernel is left where a run could reach 2^24 / time_spanned frames.  In the records gather every rank
    must leave it in the SAME chunk (root expands one record kind per chunk) -- whatever its band shows: here band 0 is static
    (its runs are the stream), band 1 keeps changing, the tick is 60 000 per frame so that the bound is 279 frames, and the
    clip runs 400: the merged stream equals the whole-plane context's and the oracle's across the switch."""
    import torch
    A = _hip()
    from adder_amd import sharding
    from adder_amd.gather import HipGather, LocalGroup
    W, H, T, chunk, world, tick = 96, 16, 400, 64, 2, 60000
    clip = O.synth_clip(O.CONTENT_NOISE, W, H, 1, T)
    clip[:, : H // 2] = clip[0, : H // 2]          # band 0: static
    kw = dict(time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, ref_time=tick, delta_t_max=tick, c_thresh_start=0, c_counter_start=0)
    ov = O.Video(W, H, 1, time_mode=O.DELTA_T, multi_mode=O.COLLAPSE, ref_time=tick, delta_t_max=tick)
    ov.ensure_capacity(8)
    ov.set_crf_parameters(0, 10)
    ov.reset_c_thresh(0)
    want = [ov.integrate_matrix(f, time_spanned=float(tick)) for f in clip]
    want_ev = np.concatenate(want)
    n_want = len(want_ev)
    bands = sharding.row_bands(H, world)
    grp = LocalGroup(world)
    d_merged = torch.full((n_want + 16, 3), -1, dtype=torch.int32, device="cuda")
    d_moff = torch.full((T + 1,), -7, dtype=torch.int64, device="cuda")
    kernels = [None] * world

    def rank_fn(r):
        y0, y1 = bands[r]
        hv = A.HipVideo(W, H, 1, row_begin=y0, row_end=y1, **kw)
        hv.set_crf_parameters(0, 10)
        g = HipGather(hv, None, r, world, local=grp)
        st, side = torch.cuda.Stream(), torch.cuda.Stream()
        d_fr = torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda()
        d_boff = torch.zeros(chunk + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        g.records_begin(0, d_merged if r == 0 else None, 0, d_moff if r == 0 else None, stream=side.cuda_stream)
        ks = []
        for f0 in range(0, T, chunk):
            nf = min(chunk, T - f0)
            rec = hv.integrate_records_device(d_fr[f0:f0 + nf], d_boff, time_spanned=float(tick), stream=st.cuda_stream)
            n_k = hv.finish()
            ks.append(hv.last_batch_kernel())
            g.records_push(rec, hv.last_batch_records(), n_k)
        n_merged, _ = g.records_end()
        if r == 0:
            assert n_merged == n_want, (n_merged, n_want)
        kernels[r] = ks
        g.close()
        hv.close()

    _run_ranks(world, rank_fn)
    grp.close()
    assert kernels[0] == kernels[1], kernels                       # the ranks change kernels together ...
    assert kernels[0][0] == A.KERNEL_LEAN_RUNS and kernels[0][-1] == A.KERNEL_LEAN, kernels   # ... and they do change
    got = np.frombuffer(d_merged[:n_want].cpu().numpy().tobytes(), dtype=O.EVENT_DTYPE)
    assert np.array_equal(got, want_ev)
    offs = d_moff.cpu().numpy()
    assert [int(offs[i + 1] - offs[i]) for i in range(T)] == [len(w) for w in want]


def test_streamed_records_gather_reports_a_merged_buffer_that_is_too_small_on_root_only():
    """The expansion drops what does not fit, end() says so on root; the peers are not left in a send, and the objects
    stay usable for the next clip."""
    import torch
    A = _hip()
    from adder_amd import sharding
    from adder_amd.gather import HipGather, LocalGroup
    W, H, T, world = 96, 40, 48, 2
    clip = O.synth_clip(O.CONTENT_NOISE, W, H, 1, T)
    kw = dict(time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    want_ev, want_off = _whole_plane(A, clip, W, H, 1, kw)
    n_want = want_ev.shape[0]
    bands = sharding.row_bands(H, world)
    grp = LocalGroup(world)
    d_small = torch.empty((n_want // 2, 3), dtype=torch.int32, device="cuda")
    d_full = torch.empty((n_want, 3), dtype=torch.int32, device="cuda")
    d_moff = torch.zeros(T + 1, dtype=torch.int64, device="cuda")

    def rank_fn(r):
        y0, y1 = bands[r]
        hv = A.HipVideo(W, H, 1, row_begin=y0, row_end=y1, **kw)
        hv.set_crf_parameters(0, 10)
        g = HipGather(hv, None, r, world, local=grp)
        st, side = torch.cuda.Stream(), torch.cuda.Stream()
        d_fr = torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda()
        d_boff = torch.zeros(17, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for dst in (d_small, d_full):
            hv.reset()
            g.records_begin(0, dst if r == 0 else None, 0, d_moff if r == 0 else None, stream=side.cuda_stream)
            for f0 in range(0, T, 16):
                rec = hv.integrate_records_device(d_fr[f0:f0 + 16], d_boff, stream=st.cuda_stream)
                n_k = hv.finish()  # (before last_batch_records(): finish is what reads the batch's record count)
                g.records_push(rec, hv.last_batch_records(), n_k)
            if r == 0 and dst is d_small:
                with pytest.raises(A.AdderHipError) as ei:
                    g.records_end()
                assert ei.value.code == A.E_OUT_CAPACITY
            else:
                g.records_end()
        g.close()
        hv.close()

    _run_ranks(world, rank_fn)
    grp.close()
    assert torch.equal(d_full, want_ev) and torch.equal(d_moff, want_off)


def test_events_gather_with_agreement_round_over_three_ranks():
    """adder_gather_events_at over the local transport: three chunks appended one after the other, each with the offsets
    all-gather, the agreement all-reduce, the grouped send / recv and the merge kernel."""
    import torch
    A = _hip()
    from adder_amd import sharding
    from adder_amd.gather import HipGather, LocalGroup
    W, H, T, world = 130, 45, 40, 3
    import clips
    clip = clips.make_clip("runs", T, H, W, 1, seed=12)
    kw = dict(time_mode=A.TIME_ABSOLUTE_T, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=7650, c_thresh_start=0, c_counter_start=0)
    want_ev, want_off = _whole_plane(A, clip, W, H, 1, kw)
    n_want = want_ev.shape[0]
    bands = sharding.row_bands(H, world)
    grp = LocalGroup(world)
    d_merged = torch.full((n_want + 8, 3), -1, dtype=torch.int32, device="cuda")
    d_moff = torch.full((T + 1,), -7, dtype=torch.int64, device="cuda")

    def rank_fn(r):
        y0, y1 = bands[r]
        hv = A.HipVideo(W, H, 1, row_begin=y0, row_end=y1, **kw)
        hv.set_crf_parameters(0, 10)
        g = HipGather(hv, None, r, world, local=grp)
        st = torch.cuda.Stream()
        d_fr = torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda()
        d_ev = torch.empty((d_fr.numel() * 3 + 16, 3), dtype=torch.int32, device="cuda")
        d_off = torch.zeros(T + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        hv.integrate_device(d_fr, d_ev, d_off, stream=st.cuda_stream)
        hv.finish()
        pos = 0
        for f0, nf in ((0, 16), (16, 16), (32, 8)):
            pos += g.gather_events_at(d_ev, d_off, f0, nf, 0, d_merged if r == 0 else None, pos,
                                      d_moff if r == 0 else None, stream=st.cuda_stream)
        merged, base = g.layout(d_off, T, stream=st.cuda_stream)  # the layout-only exchange agrees with the merge
        g.close()
        hv.close()
        return pos, merged, base

    outs = _run_ranks(world, rank_fn)
    grp.close()
    assert outs[0][0] == n_want
    assert torch.equal(d_moff, want_off) and torch.equal(d_merged[:n_want], want_ev)
    for r in range(world):
        assert np.array_equal(outs[r][1].astype(np.int64), want_off.cpu().numpy())


@pytest.mark.parametrize("Cn,world", [(1, 2), (1, 4), (3, 3)])
def test_sink_per_rank_writes_the_single_gpu_adder_file(Cn, world):
    """adder_gather_host_sink_*: every rank serialises its own band's events and stores them at their final bytes of ONE
    image in shared memory (its own mapping of /dev/shm/<name>, registered with HIP); with the header in front and the EOF
    behind, the image must be the .adder file the single-GPU raw sink writes -- 9-byte records on one channel, 11-byte ones
    on three, segments that start at every byte phase, chunks of 16 frames with a short last one."""
    import torch
    A = _hip()
    from adder_amd import sharding
    from adder_amd.gather import HipGather, HostImage, LocalGroup
    W, H, T = 150, 60, 50
    clip = O.synth_clip(O.CONTENT_SCENE, W, H, Cn, T)
    clip[20:30] = clip[20]
    kw = dict(time_mode=A.TIME_DELTA_T, multi_mode=A.MULTI_COLLAPSE, ref_time=255, delta_t_max=255, c_thresh_start=0, c_counter_start=0)
    # expected: the single-GPU raw file (header + device wire sink + EOF)
    whole = A.HipVideo(W, H, Cn, **kw)
    whole.set_crf_parameters(0, 10)
    body, n_events, _ = whole.integrate_batch_raw(clip)
    whole.close()
    header = A.raw_header(3, W, H, Cn, 7650, 255, 255, 0, A.TIME_DELTA_T, 0)
    want = header + bytes(body) + A.raw_eof()
    rec = 9 if Cn == 1 else 11
    assert len(body) == n_events * rec
    bands = sharding.row_bands(H, world)
    grp = LocalGroup(world)
    name = f"/adder_sink_test_{os.getpid()}_{Cn}_{world}"
    cap = len(header) + n_events * rec + 64
    ready = threading.Barrier(world)
    images = [None] * world

    def rank_fn(r):
        y0, y1 = bands[r]
        hv = A.HipVideo(W, H, Cn, row_begin=y0, row_end=y1, **kw)
        hv.set_crf_parameters(0, 10)
        g = HipGather(hv, None, r, world, local=grp)
        if r == 0:
            images[0] = HostImage(name, cap, create=True)
            images[0].host_array()[:] = 0xEE
            images[0].host_array()[:len(header)] = np.frombuffer(header, np.uint8)
        ready.wait(60)
        if r != 0:
            images[r] = HostImage(name, cap, create=False)  # (its own mapping of the same file, like another process)
        st = torch.cuda.Stream()
        d_fr = torch.from_numpy(np.ascontiguousarray(clip[:, y0:y1]).reshape(T, -1)).cuda()
        d_ev = torch.empty((d_fr.numel() * 3 + 16, 3), dtype=torch.int32, device="cuda")
        d_off = torch.zeros(17, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        g.host_sink_open(images[r], len(header), stream=st.cuda_stream)
        for f0 in range(0, T, 16):
            nf = min(16, T - f0)
            hv.integrate_device(d_fr[f0:f0 + nf], d_ev, d_off[:nf + 1], stream=st.cuda_stream)
            hv.finish()
            g.host_sink_chunk(d_ev, d_off, nf, stream=st.cuda_stream)
            st.synchronize()  # (d_ev is reused by the next chunk: a caller with one buffer per chunk would not wait)
        total = g.host_sink_close(stream=st.cuda_stream)
        g.close()
        hv.close()
        return total

    totals = _run_ranks(world, rank_fn)
    grp.close()
    assert totals == [n_events] * world
    img = images[0].host_array()
    end = len(header) + n_events * rec
    got = bytes(img[:end]) + A.raw_eof()
    assert int((img[end:] != 0xEE).sum()) == 0  # nothing written past the stream
    for r in range(1, world):
        images[r].close()
    images[0].close(final_bytes=end, unlink=False)
    with open("/dev/shm" + name, "rb") as fh:  # the file on tmpfs is the stream (truncated to its length)
        on_disk = fh.read()
    os.unlink("/dev/shm" + name)
    assert on_disk == got[:end]
    assert got == want
