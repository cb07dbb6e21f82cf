#!/usr/bin/env python3
"""bench.py -- framed->ADDER transcode throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one clip resident in HBM: configs[1] of
BASELINE.json -- 1920x1080 gray 8-bit, 300 frames, delta_t_max = 255 -- with the
synthetic "scene" content of SURVEY.md 8(d), crf-0 numbers, FramePerfect, Collapse,
DeltaT, raw 12-byte events written to an HBM buffer.  Each step starts from a freshly
reset transcoder so every step does identical work.

--gpus N (N > 1): the SAME 1080p clip is split into N contiguous row bands
(adder_amd.sharding.row_bands; the reference's own split is video.rs:677-691), one rank per
GPU, no collective while integrating; inside the timed step the bands are then gathered to rank 0
over RCCL/xGMI into the single ordered stream -- by default as parked RECORDS (0.35x the events'
bytes) that rank 0 expands, chunk by chunk behind the integration (--gather; strong scaling:
`value` = the clip's pixels / wall time).  Started without a launcher, bench.py re-executes
itself under torch.distributed.run.

Prints ONE JSON line on rank 0.  Besides the contract's fields it carries
  roofline     : algorithmic HBM bytes of one chunk of frames / the duration of the chunk's
                 kernels (frame kernel + scan + offsets + expansion), HIP event pairs on the
                 launch stream
  cpu_baseline : the CPU oracle (a literal port of the reference's rayon loop + serial raw
                 sink) timed on this box's host cores over a bounded sample, thread sweep
  end_to_end   : host-buffer legs (PCIe-inclusive), never `value`
"""
import argparse
import json
import os
import socket
import sys
import time

# thread placement of the CPU-baseline leg (must be in the environment before libgomp starts)
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "adder-codec-rs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, C, FRAMES = 1920, 1080, 1, 300
REF_TIME, DTM = 255, 255
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
STATE_BYTES = 16       # level 0 of a unit: hdr + integration + delta_t + best_delta_t
REC_BYTES = 12         # one parked record per unit with events (lean variants, AbsoluteT; 8 in DeltaT)

from bench_legs import (secondary_legs, end_to_end, end_to_end_default_quality, end_to_end_config5,  # noqa: E402
                        end_to_end_host_image)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="timed steps (64 x 1.6 ms: a timed region of ~100 ms)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--content", default="scene", choices=["static", "noise", "scene"])
    ap.add_argument("--multi-mode", default="collapse", choices=["collapse", "normal"])
    ap.add_argument("--time-mode", default="delta_t", choices=["delta_t", "absolute_t"])
    ap.add_argument("--delta-t-max", type=int, default=DTM)
    ap.add_argument("--crf-numbers", default="0,0,10",
                    help="c_thresh baseline,max,velocity the pixels start with (rate_controller.rs:5-18: crf 0 = 0,0,10 -- the "
                         "headline; crf 3 = 2,7,7 -- the reference's default quality; used by the profiling scripts for the quiet legs)")
    ap.add_argument("--channels", type=int, default=C)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other SURVEY 8(d) configurations")
    ap.add_argument("--secondary-ms", type=float, default=120.0, help="timed region of each secondary leg")
    ap.add_argument("--gather", default="records", choices=["torch", "cabi", "layout", "records", "records-torch", "host"],
                    help="N>1: how the bands' streams become one inside the timed step: 'records' (default) = the bands "
                         "ship their parked records, 0.35x the events' bytes, and rank 0 expands every band, through "
                         "libadder_rccl.so's streamed C-ABI (adder_gather_records_begin / _push / _end: what a Rust host "
                         "calls; torch only ships the ncclUniqueId; lean regime only, other modes fall back to 'torch'); "
                         "'records-torch' = the same over torch.distributed (adder_amd.records); 'torch' = events over "
                         "torch.distributed + the HIP merge kernel; 'cabi' = events through adder_gather_events_at; "
                         "'host' = a sink per rank: every rank stores its own wire bytes into the one .adder image in "
                         "shared memory over its own PCIe link (adder_gather_host_sink_*); 'layout' = all-gather of the "
                         "per-frame counts only")
    ap.add_argument("--output", default="wire", choices=["events", "wire"],
                    help="N=1: what the timed step leaves in HBM: the raw sink's 9 / 11-byte records written by the expansion "
                         "itself (adder_hip_integrate_wire_device: the bytes of the .adder file between header and EOF -- what "
                         "north_star's sink takes and VERDICT r3 asked the expansion to emit; the default) or 12-byte AdderEvents "
                         "(adder_hip_integrate_device; N>1 always: the gathers move events or records).  The other form is "
                         "timed in the same process as `output_check` either way")
    ap.add_argument("--skip-roofline", action="store_true",
                    help="no per-launch timing passes (used under rocprofv3 so that only default launches are seen)")
    return ap.parse_args()


def respawn_under_launcher(args):
    """`python bench.py --gpus N` from a bare shell: one process per GPU via torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn_under_launcher(args)  # does not return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Debug hook for boxes with ONE GPU: ADDER_BENCH_SHARE_DEVICE=1 runs every rank on cuda:0 over gloo --
    # RCCL refuses two ranks on one device.  Never set by the driver; it only lets the N>1 code path be
    # exercised where a single GPU is available.
    share = os.environ.get("ADDER_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import adder_amd as A
    from adder_amd import sharding

    T, Cn, Wd, Ht = args.frames, args.channels, args.width, args.height
    # strong scaling: the one plane, split by rows.  With records on the wire (the default gather) rank 0 -- which expands
    # every band and receives over ONE link per peer -- takes more rows than the peers, so that the peers' transfers and
    # root's work take equally long (sharding.gather_peer_share); the other gathers split evenly
    gather_mode = args.gather if world > 1 else "none"
    if gather_mode in ("records", "records-torch") and not (args.multi_mode == "collapse" and args.delta_t_max <= REF_TIME):
        gather_mode = "torch"  # records exist in the lean regime only (adder_hip_integrate_records_device): events then
    if gather_mode == "records" and share:
        gather_mode = "records-torch"  # RCCL cannot put two ranks on one device: the torch transport over gloo
    if gather_mode == "host" and share:
        gather_mode = "layout"
    if gather_mode in ("records", "records-torch"):
        peer_share = sharding.gather_peer_share(world, units=Wd * Ht * Cn)
        bands = sharding.row_bands_root_heavy(Ht, world, peer_share)
    else:
        bands = sharding.row_bands(Ht, world)
    y0, y1 = bands[rank]
    rows = y1 - y0
    units = rows * Wd * Cn
    content = {"static": A.CONTENT_STATIC, "noise": A.CONTENT_NOISE, "scene": A.CONTENT_SCENE}[args.content]
    multi = A.MULTI_COLLAPSE if args.multi_mode == "collapse" else A.MULTI_NORMAL
    tmode = A.TIME_DELTA_T if args.time_mode == "delta_t" else A.TIME_ABSOLUTE_T

    # the step's batches run on a stream of their own (a NULL stream would mean "behind the default stream": the library then
    # joins the legacy default stream per batch, include/adder_hip.h)
    bench_stream = torch.cuda.Stream(device=dev)
    stream = bench_stream.cuda_stream
    d_frames = torch.empty((T, units), dtype=torch.uint8, device=dev)
    A.synth_clip_device(d_frames, content, Wd, Ht, Cn, row_begin=y0, rows=rows, frame_begin=0,
                        num_frames=T, stream=stream)
    cap = int(units * T * (1.25 if args.content == "noise" else 0.75)) + 1024
    d_events = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    d_offsets = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    crf = [int(x) for x in args.crf_numbers.split(",")]
    hv = A.HipVideo(Wd, Ht, Cn, row_begin=y0, row_end=y1, time_mode=tmode, multi_mode=multi,
                    ref_time=REF_TIME, delta_t_max=args.delta_t_max, device_id=local_rank,
                    c_thresh_start=crf[0], c_counter_start=0)
    # CRF[0] = (0, 0, 10) (rate_controller.rs:9); pixels start at c_thresh 0 / counter 0, the
    # state `.crf(0)` leaves them in (video.rs:1247-1250), so reset() restores exactly that
    hv.set_crf_parameters(crf[1], crf[2])

    if share and gather_mode == "cabi":
        gather_mode = "torch"  # RCCL cannot put two ranks on one device
    hg = None
    d_merged = d_merged_offs = None
    image = None
    header = b""
    if gather_mode in ("cabi", "records", "host"):
        # the C-ABI a Rust host binds (include/adder_gather.h); torch only carries the ncclUniqueId to the other ranks
        from adder_amd.gather import HipGather, HostImage, unique_id
        uid = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        hg = HipGather(hv, uid[0], rank, world)
        if rank == 0 and gather_mode != "host":
            d_merged = torch.empty((cap * world, 3), dtype=torch.int32, device=dev)
            d_merged_offs = torch.zeros(T + 1, dtype=torch.int64, device=dev)

        def open_host_image():
            # the .adder image every rank stores into: /dev/shm/<name>, made by rank 0, mapped + registered by all.  Every rank
            # takes every collective below whatever happens to it, and the ranks AGREE on the outcome before anyone uses the image
            hdr = A.raw_header(3, Wd, Ht, Cn, REF_TIME * 30, REF_TIME, args.delta_t_max, 0, tmode, 0)
            rec_b = 9 if Cn == 1 else 11
            img_bytes = len(hdr) + int(Wd * Ht * Cn * T * (1.25 if args.content == "noise" else 0.45)) * rec_b + 4096
            name = [f"/adder_bench_{os.getpid()}_{len(opened_images)}" if rank == 0 else None]
            dist.broadcast_object_list(name, src=0)
            img, err = None, None
            try:
                if rank == 0:
                    img = HostImage(name[0], img_bytes, create=True)
                    img.host_array()[:len(hdr)] = __import__("numpy").frombuffer(hdr, dtype="uint8")
            except Exception as exc:
                err = exc
            dist.barrier()
            try:
                if rank != 0 and err is None:
                    img = HostImage(name[0], img_bytes, create=False)
            except Exception as exc:
                err = exc
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int64, device=torch.device("cpu") if share else dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if img is not None:
                    img.close(unlink=(rank == 0))
                raise RuntimeError(f"the shared .adder image could not be opened on every rank: {err!r}")
            opened_images.append(name[0])
            return img, hdr

        opened_images = []
        if gather_mode == "host":
            image, header = open_host_image()

    # N > 1: the band is integrated one chunk of frames at a time, and every finished chunk is exchanged and merged on a
    # SIDE stream while the next chunk integrates (sharding.ChunkPipelinedGather / adder_gather_events_at): the gather
    # of SURVEY 8(e) "after each batch of T frames", not one funnel after the whole clip.
    gchunk = int(os.environ.get("ADDER_BENCH_GATHER_CHUNK", "64"))
    pg = None
    d_chunk_offs = None
    side = None
    rg = None
    if world > 1 and gather_mode in ("records", "records-torch"):
        # records over the wire: the ranks ship their parked records (0.35x the events' bytes), root expands every band
        d_chunk_offs = torch.zeros(T, gchunk + 1, dtype=torch.int64, device=dev)  # (rows for any chunk length)
        if gather_mode == "records-torch":
            from adder_amd.records import RecordsPipelinedGather
            rg = RecordsPipelinedGather(T, hv, merged_cap_events=cap * world if rank == 0 else 0, dst=0, device=dev)
        else:
            side = torch.cuda.Stream(device=dev)
    if world > 1 and gather_mode in ("torch", "cabi", "host"):
        d_chunk_offs = torch.zeros((T + gchunk - 1) // gchunk, gchunk + 1, dtype=torch.int64, device=dev)
        if gather_mode == "torch":
            pg = sharding.ChunkPipelinedGather(T, merged_cap_events=cap * world if rank == 0 else 0, dst=0, video=hv, device=dev)
        else:
            side = torch.cuda.Stream(device=dev)

    wire = {"bytes": 0}

    def step(mode):
        hv.reset()
        if mode == "records":
            # adder_gather_records_begin / _push / _end: per chunk the host only waits for its OWN batch (finish); the
            # sizes of chunk k are gathered while chunk k-1's payload moves and root expands on the side stream
            if gather_wire:  # root writes the raw sink's records, like the single-GPU step (adder_gather_records_begin_wire)
                hg.records_begin_wire(0, None if d_merged is None else d_merged.view(torch.uint8).reshape(-1), 0, d_merged_offs,
                                      stream=side.cuda_stream)
            else:
                hg.records_begin(0, d_merged, 0, d_merged_offs, stream=side.cuda_stream)
            pos, nrec = 0, 0
            gc = wire.get("chunk", gchunk)
            for k, f0 in enumerate(range(0, T, gc)):
                nf = min(gc, T - f0)
                rec = hv.integrate_records_device(d_frames[f0:f0 + nf], d_chunk_offs[k, :nf + 1], stream=stream)
                n_k = hv.finish()
                nrec_k = hv.last_batch_records()
                hg.records_push(rec, nrec_k, n_k)
                pos += n_k
                nrec += nrec_k
            n_merged, sent = hg.records_end()
            wire["bytes"], wire["records"] = sent, nrec
            wire["push_host_us"] = hg.records_host_us() / max(1, k + 1)
            return pos, (n_merged if rank == 0 else pos)
        if mode == "host":
            hg.host_sink_open(image, len(header), stream=side.cuda_stream)
            pos = 0
            for k, f0 in enumerate(range(0, T, gchunk)):
                nf = min(gchunk, T - f0)
                offs_k = d_chunk_offs[k, :nf + 1]
                hv.integrate_device(d_frames[f0:f0 + nf], d_events[pos:], offs_k, stream=stream)
                n_k = hv.finish()
                side.wait_stream(bench_stream)
                hg.host_sink_chunk(d_events[pos:], offs_k, nf, stream=side.cuda_stream)  # PCIe stores beside the next chunk
                pos += n_k
            return pos, hg.host_sink_close(stream=side.cuda_stream)
        if mode == "records-torch":
            rg.reset()
            pos, sent, nrec = 0, 0, 0
            gc = wire.get("chunk", gchunk)  # (a records batch holds at most one chunk of the scratch ring: agreed below)
            for k, f0 in enumerate(range(0, T, gc)):
                nf = min(gc, T - f0)
                rec = hv.integrate_records_device(d_frames[f0:f0 + nf], d_chunk_offs[k, :nf + 1], stream=stream)
                n_k = hv.finish()
                nrec_k = hv.last_batch_records()
                sent += rg.push(rec, nrec_k, n_k)  # side streams: overlap the next chunk's integration
                pos += n_k
                nrec += nrec_k
            out = rg.result()
            wire["bytes"], wire["records"] = sent, nrec
            return pos, (int(out[1][T]) if rank == 0 else pos)
        if mode in ("torch", "cabi"):
            if pg is not None:
                pg.reset()
            pos, merged_pos = 0, 0
            for k, f0 in enumerate(range(0, T, gchunk)):
                nf = min(gchunk, T - f0)
                offs_k = d_chunk_offs[k, :nf + 1]
                hv.integrate_device(d_frames[f0:f0 + nf], d_events[pos:], offs_k, stream=stream)
                n_k = hv.finish()
                if mode == "torch":
                    pg.push(d_events[pos:pos + n_k], offs_k)  # side stream: overlaps the next chunk's integration
                else:
                    side.wait_stream(bench_stream)
                    d_offsets[f0:f0 + nf + 1] = offs_k + pos  # (the rank's own whole-clip offsets, for the record)
                    merged_pos += hg.gather_events_at(d_events[pos:], offs_k, 0, nf, 0, d_merged, merged_pos,
                                                      None if d_merged_offs is None else d_merged_offs[f0:],
                                                      stream=side.cuda_stream)
                pos += n_k
            if mode == "torch":
                out = pg.result()
                merged_total = int(out[1][-1]) if rank == 0 else pos
            else:
                side.synchronize()
                merged_total = merged_pos if rank == 0 else pos
            return pos, merged_total
        if wire_out:
            hv.integrate_wire_device(d_frames, d_events.view(torch.uint8).reshape(-1), d_offsets, stream=stream)
        else:
            hv.integrate_device(d_frames, d_events, d_offsets, stream=stream)
        n = hv.finish()
        merged_total = n
        if mode == "layout":
            lay = sharding.exchange_stream_layout(d_offsets.cpu() if share else d_offsets)
            merged_total = int(lay[0][-1])
        return n, merged_total

    wire_out = world == 1 and args.output == "wire"
    # N > 1 with records on the wire: the bands run the lean-runs kernel, root expands every band straight into the raw
    # sink's records -- the same kernels and the same output as the N = 1 step
    gather_wire = gather_mode == "records" and args.output == "wire"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(mode, steps, warmup):
        for _ in range(warmup):
            step(mode)
        barrier()
        t0 = time.perf_counter()
        res = (0, 0)
        for _ in range(steps):
            res = step(mode)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            cdev = torch.device("cpu") if share else dev
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, res

    # One-time set-up of the context, like its allocations and the graph capture: the first batches of a new length try
    # a few instances of the captured graph and keep the fastest (include/adder_hip.h, adder_hip_launch_plan_settled).
    # Not timed and not counted as warmup; every rank runs the same number of them (a fixed bound, no data-dependent
    # exit, so multi-rank runs stay in step).
    plan_steps = 0
    for _ in range(int(os.environ.get("ADDER_BENCH_PLAN_STEPS", "14"))):
        step("none")
        plan_steps += 1
    plan_settled = hv.launch_plan_settled()
    if gather_mode in ("records", "records-torch"):  # every rank cuts the clip into the same chunks: the smallest scratch ring decides
        cdev0 = torch.device("cpu") if share else dev
        tc = torch.tensor([min(gchunk, hv.chunk_frames() or gchunk)], dtype=torch.int64, device=cdev0)
        dist.all_reduce(tc, op=dist.ReduceOp.MIN)
        wire["chunk"] = int(tc.item())

    elapsed, (n_events, merged_total) = timed(gather_mode, args.steps, args.warmup)
    kernel_ms = hv.last_batch_ms()  # HIP events around the last step's frame loop
    records = wire.get("records") if gather_mode in ("records", "records-torch") else hv.last_batch_records()

    output_check = None
    if world == 1 and not args.skip_roofline:
        # the same step with the OTHER output form, in this process (same context, same buffers), and the whole stream's
        # bytes checked: the expansion's records == the events serialised by the separate pass (adder_hip_wire_events_device)
        rec_b = 9 if Cn == 1 else 11
        main_wire = wire_out
        alt_steps = max(4, args.steps // 4)
        wire_out = not main_wire
        for _ in range(plan_steps):  # (another batch variant: its launch plan settles first, untimed like the headline's)
            step("none")
        alt_elapsed, _ = timed("none", alt_steps, 2)
        wire_out = False
        hv.reset()
        step("none")
        n_ev2 = int(d_offsets[-1].item())
        offs_ev = d_offsets.clone()
        d_wire_ref = torch.empty(n_ev2 * rec_b + 16, dtype=torch.uint8, device=dev)
        hv.wire_events_device(d_events, n_ev2, d_wire_ref, stream=stream)
        d_wire = torch.empty(n_ev2 * rec_b + 16, dtype=torch.uint8, device=dev)
        hv.reset()
        hv.integrate_wire_device(d_frames, d_wire, d_offsets, stream=stream)
        n_w = hv.finish()
        torch.cuda.synchronize()
        output_check = {
            ("events_output_ms_per_step" if main_wire else "wire_output_ms_per_step"): round(alt_elapsed / alt_steps * 1e3, 3),
            "wire_bytes": int(n_w * rec_b),
            "wire_equals_serialised_events_whole_stream": bool(n_w == n_ev2 and torch.equal(d_wire[:n_w * rec_b], d_wire_ref[:n_w * rec_b])
                                                               and torch.equal(offs_ev, d_offsets)),
            "note": "adder_hip_integrate_wire_device: the expansion serialises the events itself (25 % fewer bytes stored, no "
                    "second pass); where the process's buffers landed decides which form is faster (DESIGN section 5)",
        }
        del d_wire, d_wire_ref
        wire_out = main_wire
        hv.reset()

    total_events = n_events
    layout_elapsed, layout_steps = None, max(2, args.steps // 2)
    if world > 1:
        cdev = torch.device("cpu") if share else dev
        te = torch.tensor([n_events, wire["bytes"]], dtype=torch.int64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.SUM)
        total_events = int(te[0].item())
        wire["bytes"] = int(te[1].item())
        if (rank == 0 or gather_mode == "host") and gather_mode != "layout":
            assert merged_total == total_events, (merged_total, total_events)
        if gather_mode != "layout":  # the cheaper exchange, as an extra key
            layout_elapsed, _ = timed("layout", layout_steps, 1)

    # ---- N > 1: the other two forms of the exchange as first-class legs of their own (each timed like the headline) ----
    # `value` times ONE form (config.sharding says which); the first scaling curve should show all three side by side:
    # the sink per rank and the layout-only exchange are the forms whose cost does not funnel through rank 0.
    scale_legs = {}
    if world > 1 and layout_elapsed is not None:
        scale_legs["layout_only"] = {"ms_per_step": round(layout_elapsed / layout_steps * 1e3, 3),
                                     "value": round(Wd * Ht * T / (layout_elapsed / layout_steps) / 1e6, 1), "unit": "Mpixels/s",
                                     "bytes_sent_per_rank_per_step": 8 * (T + 1),
                                     "what": "every rank keeps its band's stream in its own HBM; only the per-frame counts are all-gathered"}
    if world > 1 and hg is not None and gather_mode in ("records", "cabi") and not share and os.environ.get("ADDER_BENCH_SCALE_LEGS", "1") == "1":
        try:
            leg_steps = max(2, args.steps // 2)
            image, header = open_host_image()
            if side is None:
                side = torch.cuda.Stream(device=dev)
            if d_chunk_offs is None or d_chunk_offs.shape[0] < (T + gchunk - 1) // gchunk:
                d_chunk_offs = torch.zeros((T + gchunk - 1) // gchunk, gchunk + 1, dtype=torch.int64, device=dev)
            sink_elapsed, (sink_n, sink_total) = timed("host", leg_steps, 1)
            rec_b = 9 if Cn == 1 else 11
            scale_legs["sink_per_rank"] = {"ms_per_step": round(sink_elapsed / leg_steps * 1e3, 3),
                                           "value": round(Wd * Ht * T / (sink_elapsed / leg_steps) / 1e6, 1), "unit": "Mpixels/s",
                                           "bytes_stored_by_this_rank_per_step": int(sink_n) * rec_b,
                                           "what": "every rank serialises its band and stores the bytes at their final place of the one "
                                                   ".adder image in shared memory over its own PCIe link (adder_gather_host_sink_*)"}
        except Exception as exc:  # (never at the price of the headline line)
            scale_legs["sink_per_rank"] = {"error": repr(exc)[:300]}

    # ---- roofline: one extra step with HIP event pairs around the launches ----
    k1_us = post_us = k1_one_us = k1_one_pair_us = post_one_us = 0.0
    post_one_chunks = 1
    k1_frames = 1.0
    post_chunks = 1
    chunk_frames = hv.chunk_frames()
    records1 = records
    if not args.skip_roofline:
        hv.set_launch_timing(True)
        step("none")
        k1_us, k1_frames = hv.last_launch_avg_us(), hv.last_launch_frames() or 1.0
        post_us = hv.last_post_avg_us()
        post_chunks = hv.last_post_chunks()
        default_depth = int(k1_frames + 0.999)
        # the frame kernel alone with one frame per launch (the per-frame `consume` contract, state
        # streamed from HBM every frame): the HBM-bound regime of SURVEY 8(d)
        hv.set_frames_per_launch(1)
        step("none")
        k1_one_pair_us = hv.last_launch_avg_us()
        # ... and with ONE event pair around each chunk's run of launches (no event packets between the
        # kernels: what is left on top of the kernel time is the gap between two dependent launches)
        hv.set_launch_timing(2)
        step("none")
        k1_one_us = hv.last_launch_avg_us()
        post_one_us = hv.last_post_avg_us()          # scan + offsets + expansion of the same step, per chunk of its launches
        post_one_chunks = hv.last_post_chunks()
        records1 = hv.last_batch_records()
        hv.set_frames_per_launch(int(os.environ.get("ADDER_HIP_FRAMES_PER_LAUNCH", "0")) or default_depth)
        hv.set_launch_timing(False)

    if image is not None:
        if rank != 0:
            image.close()
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    host_sink_check = None
    if image is not None:  # the image is the .adder file: a look at its first and last record, then the file goes
        import numpy as np
        rec_b = 9 if Cn == 1 else 11
        end = len(header) + total_events * rec_b
        img = image.host_array()
        first = bytes(img[len(header):len(header) + rec_b])
        host_sink_check = {"file_bytes": int(end + 11), "first_record_hex": first.hex(),
                           "header_ok": bytes(img[:5]) == b"adder"}
        image.close(final_bytes=end, unlink=True)

    pixels_per_step = Wd * Ht * T
    ms_per_step = elapsed / args.steps * 1e3
    value = pixels_per_step / (elapsed / args.steps) / 1e6
    e_all = total_events / float(Wd * Ht * Cn * T)
    e0 = n_events / float(units * T)
    r0 = records / float(units * T)
    abs_t = tmode == A.TIME_ABSOLUTE_T
    lean = multi == A.MULTI_COLLAPSE and args.delta_t_max <= REF_TIME
    # B = 1 (input) + (S_in + S_out) / T_launch + 12 e bytes per pixel-channel-frame, with the state
    # this implementation really keeps (S = 16 B, +4 B last_fired_t in AbsoluteT) and T_launch = frames per launch.
    S = STATE_BYTES + (4 if abs_t else 0)
    # kernels of one (full) chunk of frames: its frame-kernel launches + scan + offsets + expansion.  Both averages
    # include the clip's last, shorter chunk and are scaled to a full one.
    post_frames = T / max(post_chunks, 1)
    post_us_full = post_us * (chunk_frames / max(post_frames, 1.0))
    chunk_us = k1_us * (chunk_frames / max(k1_frames, 1.0)) + post_us_full
    alg_b = 1 + 2 * S / max(k1_frames, 1.0) + 12 * e0
    achieved = alg_b * units * chunk_frames / (chunk_us * 1e-6) / 1e9 if chunk_us > 0 else 0.0
    # the frame kernel's own traffic (what it really moves): input + state + parked records
    rec_bytes_one = REC_BYTES if abs_t or not lean else 8      # the one-frame kernels' records
    # blocked launches at crf 0 in DeltaT run adder_lp_kernel: 4-byte records (adder_pixel.hpp lp_park4)
    rec_bytes = 4 if (lean and not abs_t and crf == [0, 0, 10]) else rec_bytes_one
    k1_b = 1 + 2 * S / max(k1_frames, 1.0) + rec_bytes * r0
    k1_gbs = k1_b * units * k1_frames / (k1_us * 1e-6) / 1e9 if k1_us > 0 else 0.0
    # the rest of the chunk: parked records in, 12-byte events out
    post_b = rec_bytes * r0 + 12 * e0
    post_gbs = post_b * units * chunk_frames / (post_us_full * 1e-6) / 1e9 if post_us > 0 else 0.0
    r1 = records1 / float(units * T)
    one_b = 1 + 2 * S + rec_bytes_one * r1
    one_gbs = one_b * units / (k1_one_us * 1e-6) / 1e9 if k1_one_us > 0 else 0.0
    # ... and the WHOLE one-frame-per-launch pipeline (the literal `consume` contract of north_star): a chunk = chunk_frames
    # launches of the frame kernel + scan + offsets + expansion, priced with SURVEY's bytes at T = 1 (1 + 2 S + 12 e)
    one_post_full = post_one_us * (chunk_frames / max(T / max(post_one_chunks, 1), 1.0))
    one_chunk_us = k1_one_us * chunk_frames + one_post_full
    one_alg_b = 1 + 2 * S + 12 * e0
    one_pipe_gbs = one_alg_b * units * chunk_frames / (one_chunk_us * 1e-6) / 1e9 if one_chunk_us > 0 else 0.0

    # HBM bytes really moved by the chunk's kernels: PMC counters cannot be read inside this process (rocprofv3 collects
    # them in separate passes), so the figure comes from the committed collection of the SAME command
    # (tools/profile_round.sh -> profiles/<round>_traffic_default.json: 2 x FETCH_SIZE + WRITE_SIZE per launch, KiB,
    # MI355X_MICROARCH.md HBM section) when this run is the default workload; null otherwise.  The collection's launches
    # covered `frames_per_launch` frames each (its JSON says so; round 4's ran 160 frames = 64 + 64 + 32, 53.3 on
    # average, and its bytes were set against a 64-frame chunk's algorithmic bytes: the quoted 1.22x was really 1.45x):
    # the bytes are scaled to one full chunk, and the ratio is taken against the algorithmic bytes of the same frames.
    traffic, traffic_ratio, traffic_note = None, None, "no committed PMC collection for this workload (tools/profile_round.sh)"
    try:
        default_workload = (Wd, Ht, Cn, T, args.delta_t_max, args.content, args.multi_mode, args.time_mode) == \
            (W, H, C, FRAMES, DTM, "scene", "collapse", "delta_t") and crf == [0, 0, 10]
        cands = [("r06_traffic_default.json" if wire_out else "r06_traffic_events_output.json")]
        tname = next((n for n in cands if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        if default_workload and world == 1 and tname:
            tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
            tk = tj["kernels"]
            fpl = float(tj.get("frames_per_launch", 160.0 / 3.0))  # (round 4's collection carries no such key)
            per_launch = {k: v["hbm_bytes_per_launch"] for k, v in tk.items()}
            lean_b = sum(v for k, v in per_launch.items() if "lean_kernel" in k or "lr_kernel" in k or "lp_kernel" in k)
            exp_b = sum(v for k, v in per_launch.items() if "expand_kernel" in k or "lpx_kernel" in k)
            scan_b = sum(v for k, v in per_launch.items() if "scan_kernel" in k or "offsets_kernel" in k)
            measured = lean_b + exp_b + scan_b  # bytes of one measured launch set (fpl frames)
            traffic = int(measured * chunk_frames / fpl)
            alg_same = alg_b * units * fpl
            traffic_ratio = round(measured / alg_same, 3) if alg_same > 0 else None
            traffic_note = (f"HBM bytes of the frame kernel + scan + offsets + expansion launches from profiles/{tname} "
                            f"(separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, 2 x FETCH_SIZE + "
                            f"WRITE_SIZE, KiB): {int(measured)} bytes per launch set of {fpl:.1f} frames, scaled here to one "
                            f"{chunk_frames}-frame chunk; algorithmic bytes of the same {fpl:.1f} frames: {int(alg_same)} "
                            f"(traffic_over_algorithmic = their ratio)")
    except Exception as exc:
        traffic_note = f"could not read the committed PMC collection: {exc}"

    out = {
        "metric": "Mpixels/s framed->ADDER transcode (1080p, delta_t_max=255)",
        "value": round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{Wd}x{Ht}{'x3 RGB' if Cn == 3 else ' gray'} 8-bit, {T} frames, "
                        f"delta_t_max={args.delta_t_max}, ref_time={REF_TIME}, content={args.content}, crf "
                        f"numbers ({args.crf_numbers}), FramePerfect, {args.multi_mode}, {args.time_mode}, " +
                        ("the raw sink's 9 / 11-byte records to HBM (serialised by the expansion)" if (wire_out or gather_wire) else "raw events to HBM"),
            "output": "raw-sink records" if (wire_out or gather_wire) else "AdderEvents (12 B)",
            "plane": [Wd, Ht, Cn],
            "rows_per_gpu": rows,
            "row_bands": [list(b) for b in bands],
            "frames_per_step": T,
            "sharding": ("single GPU" if world == 1 else
                         f"{world} row bands of the one plane; per step the bands' " +
                         ("lean-runs RECORDS are shipped to rank 0, which expands every band into the one ordered stream"
                          if gather_mode in ("records", "records-torch") else
                          "wire bytes are stored by every rank itself into the one .adder image in shared memory (a sink per rank)"
                          if gather_mode == "host" else "event streams are gathered to rank 0") +
                         f" ({gather_mode}) inside the timed region, chunk by chunk ({gchunk} frames) on side "
                         f"streams while the next chunk integrates"),
            "world_size_seen": world,
            "world_size_seen_by_rccl": (hg.world() if (hg is not None and hasattr(hg, "world")) else None),
            "value_times": ("one GPU, no exchange" if world == 1 else gather_mode),
            "backend": "none" if world == 1 else ("gloo (shared-device debug)" if share else "nccl (RCCL)"),
        },
        "output_check": output_check,
        "events_per_s": round(total_events / (elapsed / args.steps), 1),
        "events_per_pixel_frame": round(e_all, 5),
        "records_per_unit_frame": round(r0, 5),
        "frame_loop_ms_hip_events": round(kernel_ms, 3),
        "launch_plan": {"setup_steps_untimed": plan_steps, "settled": bool(plan_settled),
                        "note": "the captured graph of a batch length is instantiated up to 6 times on the first batches "
                                "and the fastest instance kept (the runtime binds the graph's two branches to hardware "
                                "queues at instantiation; measured 1.86 vs 2.08 ms per step between instances)"},
        "roofline": {
            "bound": "hbm",
            "kernel": ("one chunk of frames: adder_lp_kernel (the lean-runs step in packed bytes: crf 0, DeltaT; adder_lr_kernel in "
                       "AbsoluteT, adder_lean_kernel otherwise) + adder_scan_kernel (its blocks chain the frame offsets; adder_offsets_kernel for the other frame kernels) + adder_lpx_kernel "
                       "(adder_expand_kernel for the other record formats)") if lean else
                      "one chunk of frames: adder_frame_kernel + adder_scan_kernel + adder_offsets_kernel + "
                      "adder_expand_kernel",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_over_algorithmic": traffic_ratio,
            "traffic_over_record_bytes": (round(traffic_ratio * alg_b / (alg_b - (12 - (9 if Cn == 1 else 11)) * e0), 3)
                                          if traffic_ratio and wire_out else None),
            "traffic_note": traffic_note,
            "bytes_per_unit_frame": round(alg_b, 3),
            "bytes_per_unit_frame_note": "SURVEY 8(d)'s canonical figure: 1 input + state / T_launch + 12 per event (E = 12 B, "
                                         "the device buffer's AdderEvent)" + (
                "; this run's expansion writes the events as 9 / 11-byte records instead -- frac_at_record_bytes prices the "
                "same time with E = the record's bytes" if wire_out else ""),
            "frac_at_record_bytes": (round((alg_b - (12 - (9 if Cn == 1 else 11)) * e0) * units * chunk_frames / (chunk_us * 1e-6) / 1e9
                                           / HBM_PEAK_GBS, 4) if wire_out and chunk_us > 0 else None),
            # the same algorithmic bytes over the STEP's wall time (the timed region of `value`): the tuned graph may run chunk
            # k + 1's frame kernel beside chunk k's scan and expansion, so a step can be shorter than its kernels' eager sum
            "step_wall_frac": round(alg_b * units * T / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if world == 1 else None,
            "step_wall_frac_at_record_bytes": (round((alg_b - (12 - (9 if Cn == 1 else 11)) * e0) * units * T / (ms_per_step * 1e-3) / 1e9
                                                     / HBM_PEAK_GBS, 4) if wire_out and world == 1 else None),
            "frames_per_launch": k1_frames,
            "frames_per_chunk": chunk_frames,
            "units_per_chunk": int(units * chunk_frames),
            "chunk_us": round(chunk_us, 3),
            "frame_kernel_launch_us": round(k1_us, 3),
            "scan_offsets_expand_us": round(post_us_full, 3),
            "frame_kernel_actual_GBs": round(k1_gbs, 1),
            "frame_kernel_actual_bytes_per_unit_frame": round(k1_b, 3),
            "expansion_actual_GBs": round(post_gbs, 1),
            "note": "algorithmic bytes (SURVEY 8(d) with the real 16-byte state: 1 + 32/T_launch + 12 e per unit-frame) "
                    "over ALL kernels of a chunk; the pipeline really moves the parked records twice on top of that "
                    "(frame_kernel_actual / expansion_actual are those kernels' own bytes over their own durations)",
        },
        "roofline_one_frame_per_launch": {
            "bound": "hbm",
            "kernel": "adder_lean1w_kernel" if lean else "adder_frame_kernel",
            "pipeline": {"what": "the whole one-frame-per-launch chunk: %d frame-kernel launches + scan + offsets + expansion, "
                                 "SURVEY 8(d)'s bytes at T = 1 (1 + 2 x state + 12 e per unit-frame)" % chunk_frames,
                         "chunk_us": round(one_chunk_us, 3), "frame_kernels_us": round(k1_one_us * chunk_frames, 3),
                         "scan_offsets_expand_us": round(one_post_full, 3), "bytes_per_unit_frame": round(one_alg_b, 3),
                         "achieved": round(one_pipe_gbs, 1), "frac": round(one_pipe_gbs / HBM_PEAK_GBS, 4)},
            "achieved": round(one_gbs, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(one_gbs / HBM_PEAK_GBS, 4),
            "frac_is": "the FRAME KERNEL alone on the bytes it moves; `pipeline.frac` is the regime's figure",
            "bytes_per_unit_frame": round(one_b, 3),
            "units_per_launch": units,
            "launch_avg_us": round(k1_one_us, 3),
            "launch_avg_us_event_pair_per_launch": round(k1_one_pair_us, 3),
            "note": "the frame kernel alone with the state streamed from HBM every frame (per-frame consume contract): "
                    "bytes it really moves = 1 input + 16 state in + 16 state out + 12 per parked record.  launch_avg_us "
                    "= one HIP-event pair around each chunk's run of 64 back-to-back launches / 64 (kernel + the gap "
                    "to the next dependent launch); with a pair around EVERY launch the event packets themselves add "
                    "~2 us (launch_avg_us_event_pair_per_launch).  rocprofv3's kernel duration of the same command is "
                    "in profiles/ (kernel stats, adder_lean1w_kernel)",
        },
    }
    if gather_mode in ("records", "records-torch"):
        out["records_over_the_wire"] = {
            "transport": "libadder_rccl.so C-ABI, streamed (adder_gather_records_begin / _push / _end)"
                         if gather_mode == "records" else "torch.distributed (adder_amd.records)",
            "bytes_per_step_all_peers": wire["bytes"], "events_bytes_per_step_all_peers": 12 * (total_events - n_events),
            "push_host_us_per_chunk_rank0": round(wire.get("push_host_us", 0.0), 1) if gather_mode == "records" else None,
            "note": "the peers ship their parked records + three per-segment tables instead of their events; root expands "
                    "every band (adder_hip_expand_records_device)"}
    if gather_mode == "host":
        rec_b = 9 if Cn == 1 else 11
        out["sink_per_rank"] = {
            "bytes_per_step": total_events * rec_b, "GBs_total": round(total_events * rec_b / (elapsed / args.steps) / 1e9, 2),
            "GBs_per_link": round(total_events * rec_b / (elapsed / args.steps) / 1e9 / world, 2), "check": host_sink_check,
            "note": "every rank serialises its band's events to wire records on the device and stores them at their final "
                    "bytes of /dev/shm/<image> over its own PCIe link (adder_gather_host_sink_*); no xGMI funnel, no "
                    "host wait per chunk"}
    if world > 1:
        wire_b = wire.get("bytes", 0)
        out["scale"] = {
            "value_form": gather_mode,
            "value_ms_per_step": round(ms_per_step, 3),
            "bytes_sent_per_step_all_peers": int(wire_b), "bytes_sent_per_step_per_peer": int(wire_b // max(world - 1, 1)),
            "legs": scale_legs,
            "predicted_ms_per_step": {"records (root expands every band)": {"2": 1.4, "4": 1.3, "8": 1.1},
                                      "note": "DESIGN section 6's prediction: the bands of a gather still run round 5's kernels (adder_lr_kernel, root's "
                                              "adder_expand_bands_kernel: 1.25 ms at N = 1) while N = 1 runs round 6's packed pair at ~1.0 ms, so the "
                                              "default form is predicted SLOWER than one GPU at N = 2-4 and level at 8 -- root writes every output "
                                              "byte, flat by construction; sink_per_rank and layout_only are the forms that can scale (their legs above)"}}
    if layout_elapsed is not None:
        out["layout_only_exchange"] = {
            "value": round(pixels_per_step / (layout_elapsed / layout_steps) / 1e6, 1),
            "unit": "Mpixels/s",
            "note": "same step with only the per-frame counts all-gathered (the payload stays sharded in HBM)"}

    if world == 1 and not args.no_secondary:
        # the other configurations of SURVEY 8(d), each with its own context and clip (the headline's stay alive)
        out["secondary"] = secondary_legs(args, torch, A)
    if world == 1 and not args.no_end_to_end:
        out["end_to_end"] = end_to_end(hv, d_frames, T, units, Wd, Ht, Cn)
        d_chunk = torch.zeros((T + 63) // 64, 65, dtype=torch.int64, device=dev)
        out["end_to_end"]["raw_file_image_sink"] = end_to_end_host_image(torch, A, hv, d_frames, d_events, d_chunk, T, Wd, Ht,
                                                                         Cn, tmode, args.delta_t_max, total_events)
        out["end_to_end"]["default_quality_raw"] = end_to_end_default_quality(torch, A, Wd, Ht)
        out["end_to_end"]["compressed_sink_config5"] = end_to_end_config5(torch, A)
    if world == 1 and not args.no_cpu_baseline:
        hv.reset()
        hv.integrate_device(d_frames, d_events, d_offsets, stream=stream)
        hv.finish()
        out["cpu_baseline"] = cpu_baseline(d_frames, d_events, d_offsets, T, Cn, Wd, Ht, multi, tmode, args)
    # (RCCL writes its version banner to the C stdout buffer: out with it first, so that the JSON is the LAST line)
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(d_frames, d_events, d_offsets, T, Cn, Wd, Ht, multi, tmode, args):
    """The oracle (a port of the reference CPU path: per-row-chunk tasks whose events stay in per-chunk buffers like the
    reference's Vec<Vec<Event>>, then the serial 9/11-byte raw sink, video.rs:677-740) on this box's host cores over a
    bounded prefix of the SAME clip.  The team lives for the clip (one parallel region, a barrier per frame), every
    thread steps the rows it first touched (static schedule, OMP_PLACES=cores, OMP_PROC_BIND=spread): what a rayon pool
    over `chunks` converges to.  Thread sweep x chunk_rows, best-of reported with its thread count, the STREAM triad of
    the same team beside it.  It is also the checker: the GPU events of the frames it processes must equal it bit for bit."""
    import numpy as np
    from oracle import oracle as O

    max_threads = O.max_threads()
    # what this process may really use: its affinity mask and its cgroup's CPU quota (a pod of a shared node often has
    # fewer CPUs than the node shows; a team larger than that spins against itself at every barrier)
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = max_threads
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        except Exception:
            quota = None
    usable = int(min(max_threads, affinity, quota if quota else max_threads))
    sweep_threads = sorted({t for t in (1, 8, 16, 32, 64, 128, usable) if t <= max(usable, 1)})
    n_points = len(sweep_threads) + 5
    budget = args.cpu_seconds / n_points
    max_frames = min(T, 64)
    host = np.ascontiguousarray(d_frames[:max_frames].cpu().numpy())
    sink = np.zeros(Wd * Ht * Cn * 3 * 11 + 64, np.uint8)
    offs = d_offsets.cpu().numpy()
    units = Wd * Ht * Cn

    def fresh(threads, chunk_rows=1):
        v = O.Video(Wd, Ht, Cn, time_mode=tmode, multi_mode=multi, ref_time=REF_TIME,
                    delta_t_max=args.delta_t_max, chunk_rows=chunk_rows, threads=threads)
        v.set_crf_parameters(0, 10)
        v.reset_c_thresh(0)
        return v

    def run_clip(threads, chunk_rows, with_sink, est_mpx):
        """one persistent team over as many frames as the budget allows (sized from an estimate of the rate)"""
        frames = int(max(4, min(max_frames, budget * est_mpx * 1e6 / (Wd * Ht))))
        v = fresh(threads, chunk_rows)
        t0 = time.perf_counter()
        ev = v.integrate_clip(host.ctypes.data, frames, units, Wd * Cn, float(REF_TIME), sink.ctypes.data if with_sink else None)
        el = time.perf_counter() - t0
        return Wd * Ht * frames / el / 1e6, frames, ev

    def run_per_frame(threads, chunk_rows, check):
        """the per-frame entry point (a fork / join per frame, dynamic schedule): round 3's baseline, and the checker"""
        v = fresh(threads, chunk_rows)
        t_total, frames_done, events, ok = 0.0, 0, 0, True
        while frames_done < max_frames and t_total < budget:
            f = host[frames_done]
            t0 = time.perf_counter()
            n = v.integrate_matrix_chunks(f.ctypes.data, Wd * Cn, float(REF_TIME))
            v.chunks_raw_events(sink.ctypes.data)  # the serial sink stage
            t_total += time.perf_counter() - t0
            if check:  # outside the timed region
                lo, hi = int(offs[frames_done]), int(offs[frames_done + 1])
                gpu = np.frombuffer(d_events[lo:hi].cpu().numpy().tobytes(), dtype=O.EVENT_DTYPE)
                cpu = v.chunks_copy_out(n)
                ok = ok and len(gpu) == len(cpu) and bool(np.array_equal(gpu, cpu))
            events += n
            frames_done += 1
        return Wd * Ht * frames_done / t_total / 1e6, frames_done, events, ok

    sweep, est = [], 40.0
    for th in sweep_threads:
        mp, fr, _ = run_clip(th, 1, True, est * max(1.0, min(th, 16) / 2.0) if th > 1 else est)
        sweep.append({"threads": th, "value": round(mp, 2), "frames": fr})
        est = max(est, mp / max(1.0, min(th, 16) / 2.0))
    best = max(sweep, key=lambda s: s["value"])
    bt = best["threads"]
    by_rows = {1: best["value"]}
    for cr in (8, 64):
        by_rows[cr] = round(run_clip(bt, cr, True, best["value"])[0], 2)
    best_rows = max(by_rows, key=by_rows.get)
    mp_nosink = run_clip(bt, best_rows, False, best["value"])[0]
    mp_forkjoin, _, _, _ = run_per_frame(bt, 1, False)
    _, fr_chk, ev_chk, ok = run_per_frame(bt, 1, True)
    triad = {}
    for th in sorted({1, bt, usable}):
        triad[str(th)] = round(O.stream_triad_GBs(64 << 20, th), 1)  # 3 x 256 MB
    value = by_rows[best_rows]
    # the AoS state the port streams per frame (sizeof(PixelArena) in and out) + input + events: what `value` means in GB/s
    aos_bytes = 2 * O.sizeof_pixel_arena() + 1
    return {
        "value": value,
        "unit": "Mpixels/s",
        "cores": bt,
        "kind": "port",
        "sample": f"first {best['frames']} frames of the same clip per sweep point (about {budget:.1f} s each); one OpenMP "
                  f"team for the clip (barrier per frame), static row chunks first-touched by their thread (chunk_rows="
                  f"{best_rows}, OMP_PLACES=cores, OMP_PROC_BIND=spread) + serial raw sink after every frame; events stay "
                  f"in per-chunk buffers like the reference's Vec<Vec<Event>>",
        "host_cores": max_threads,
        "cpus_usable": {"omp_max_threads": max_threads, "affinity": affinity, "cgroup_quota_cpus": quota, "swept_up_to": usable},
        "caveat": "a stated baseline, not a target: a C restatement of the reference's rayon loop (the Rust original cannot "
                  "be built here).  Round 3 forked and joined a team per frame with a dynamic schedule and no placement "
                  "(fork_join_per_frame below, kept for comparison); this run keeps the team and the rows' placement for "
                  "the clip.  stream_triad_GBs is what the same team gets out of the box's memory system; value x "
                  f"{aos_bytes} bytes per pixel-frame of AoS state is the port's own traffic",
        "thread_sweep": sweep,
        "chunk_rows_sweep_at_best_threads": {str(k): v for k, v in by_rows.items()},
        "one_thread": next((s["value"] for s in sweep if s["threads"] == 1), None),
        "without_sink": round(mp_nosink, 2),
        "fork_join_per_frame": round(mp_forkjoin, 2),
        "stream_triad_GBs": triad,
        "port_traffic_GBs_at_value": round(value * aos_bytes / 1e3, 1),
        "checked_frames": fr_chk,
        "checked_events": ev_chk,
        "gpu_events_match_bit_exact": ok,
    }


if __name__ == "__main__":
    main()
