#!/usr/bin/env python3
"""bench.py -- framed->ADDER transcode throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one clip resident in HBM: config[1] of
BASELINE.json -- 1920x1080 gray 8-bit, 300 frames, delta_t_max = 255 -- with the
synthetic "scene" content of SURVEY.md 8(d), crf-0 numbers, FramePerfect, Collapse,
DeltaT, raw 12-byte events written to an HBM buffer.  Each step starts from a freshly
reset transcoder so every step does identical work.

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): the plane is
1920 x (1080*N) and rank r owns rows [1080 r, 1080 (r+1)) (weak scaling); no collective
runs while integrating, then the per-rank event segments are gathered to rank 0 in
frame-major raster order (adder_amd/sharding.py) inside the timed region.

Prints ONE JSON line on rank 0.  Besides the contract's fields it carries
  roofline     : algorithmic HBM bytes of the frame kernel / its mean launch duration
                 (HIP event pair around every launch on the launch stream)
  cpu_baseline : the CPU oracle (a literal port of the reference's rayon loop + serial
                 raw sink) timed on this box's host cores over a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "adder-codec-rs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

W, H_BAND, C, FRAMES = 1920, 1080, 1, 300
REF_TIME, DTM = 255, 255
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--content", default="scene", choices=["static", "noise", "scene"])
    ap.add_argument("--multi-mode", default="collapse", choices=["collapse", "normal"])
    ap.add_argument("--time-mode", default="delta_t", choices=["delta_t", "absolute_t"])
    ap.add_argument("--delta-t-max", type=int, default=DTM)
    ap.add_argument("--channels", type=int, default=C)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-payload", action="store_true",
                    help="N>1: also funnel every rank's events to rank 0 over RCCL inside the timed step "
                         "(default: all-gather of the per-frame counts only; the payload stays sharded)")
    ap.add_argument("--skip-roofline", action="store_true",
                    help="no per-launch timing passes (used under rocprofv3 --pmc so that only default-depth launches are counted)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # Debug hook for boxes with ONE GPU: ADDER_BENCH_SHARE_DEVICE=1 runs every rank on cuda:0 and does the
    # (tiny) layout exchange over gloo on host copies -- RCCL refuses two ranks on one device.  Never set
    # by the driver; it only lets the N>1 code path be exercised where a single GPU is available.
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

    T, Cn = args.frames, args.channels
    H_total = H_BAND * world
    y0, y1 = rank * H_BAND, (rank + 1) * H_BAND
    units = H_BAND * W * Cn
    content = {"static": A.CONTENT_STATIC, "noise": A.CONTENT_NOISE, "scene": A.CONTENT_SCENE}[args.content]
    multi = A.MULTI_COLLAPSE if args.multi_mode == "collapse" else A.MULTI_NORMAL
    tmode = A.TIME_DELTA_T if args.time_mode == "delta_t" else A.TIME_ABSOLUTE_T

    stream = torch.cuda.current_stream().cuda_stream
    d_frames = torch.empty((T, units), dtype=torch.uint8, device=dev)
    A.synth_clip_device(d_frames, content, W, H_total, Cn, row_begin=y0, rows=H_BAND, frame_begin=0,
                        num_frames=T, stream=stream)
    cap = int(units * T * (1.25 if args.content == "noise" else 0.75)) + 1024
    d_events = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    d_offsets = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    hv = A.HipVideo(W, H_total, Cn, row_begin=y0, row_end=y1, time_mode=tmode, multi_mode=multi,
                    ref_time=REF_TIME, delta_t_max=args.delta_t_max, device_id=local_rank,
                    c_thresh_start=0, c_counter_start=0)
    # CRF[0] = (0, 0, 10) (rate_controller.rs:9); pixels start at c_thresh 0 / counter 0, the
    # state `.crf(0)` leaves them in (video.rs:1247-1250), so reset() restores exactly that
    hv.set_crf_parameters(0, 10)

    def step():
        hv.reset()
        hv.integrate_device(d_frames, d_events, d_offsets, stream=stream)
        n = hv.finish()
        merged = None
        if world > 1:
            if args.gather_payload:
                merged = sharding.gather_event_stream(d_events[:n], d_offsets, dst=0)
            else:
                # the ordered concatenation is fixed by this exchange; rank r's segment of frame f
                # belongs at my_base[f] of the merged stream
                merged = sharding.exchange_stream_layout(d_offsets.cpu() if share else d_offsets)
        return n, merged

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    n_events = 0
    for _ in range(args.steps):
        n_events, merged = step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = hv.last_batch_ms()  # HIP events around the last step's frame loop

    total_events = n_events
    if world > 1:
        cdev = torch.device("cpu") if share else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        te = torch.tensor([n_events], dtype=torch.int64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.SUM)
        total_events = int(te.item())
        if args.gather_payload:
            if rank == 0:
                assert merged is not None and merged[0].shape[0] == total_events
        else:
            assert int(merged[0][-1]) == total_events  # every rank knows the merged stream's layout

    # ---- roofline of the dominant kernel: one extra step with an event pair per launch ----
    launch_us, launch_frames, launch1_us = 0.0, 1.0, 0.0
    if not args.skip_roofline:
        hv.set_launch_timing(True)
        step()
        launch_us = hv.last_launch_avg_us()
        launch_frames = hv.last_launch_frames() or 1.0
        # the same kernel with one frame per launch (the per-frame `consume` contract, state
        # streamed from HBM every frame): this is the HBM-bound regime of SURVEY 8(d)
        default_depth = int(launch_frames + 0.999)
        hv.set_frames_per_launch(1)
        step()
        launch1_us = hv.last_launch_avg_us()
        hv.set_frames_per_launch(int(os.environ.get("ADDER_HIP_FRAMES_PER_LAUNCH", "0")) or default_depth)
        hv.set_launch_timing(False)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    pixels_per_step = W * H_total * T
    ms_per_step = elapsed / args.steps * 1e3
    value = pixels_per_step / (elapsed / args.steps) / 1e6
    e = total_events / float(units * world * T)
    # B = 1 + (S_in + S_out)/T_launch + 12 e per pixel-channel-frame;
    # S = 20 B (Collapse/DeltaT), +4 B with AbsoluteT; T_launch = frames one launch steps.
    S = 20 + (4 if tmode == A.TIME_ABSOLUTE_T else 0)
    e_rank0 = n_events / float(units * T)
    bytes_per_unit = 1 + 2 * S / launch_frames + 12 * e_rank0
    units_per_launch = units * launch_frames
    achieved = bytes_per_unit * units_per_launch / (launch_us * 1e-6) / 1e9 if launch_us > 0 else 0.0
    # at depth 1 the expansion is NOT fused (it runs as its own kernel on a second stream): the frame
    # kernel's launch then moves the input, the state and the parked records (4 bytes each in the
    # DeltaT / delta_t_max <= ref_time variants, 8 otherwise)
    park_bytes = 4 if (tmode == A.TIME_DELTA_T and multi == A.MULTI_COLLAPSE and args.delta_t_max <= REF_TIME) else 8
    bytes1 = 1 + 2 * S + park_bytes * e_rank0
    achieved1 = bytes1 * units / (launch1_us * 1e-6) / 1e9 if launch1_us > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "Mpixels/s framed->ADDER transcode (1080p, delta_t_max=255)",
        "value": round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{W}x{H_BAND}{'x3 RGB' if Cn == 3 else ' gray'} 8-bit per GPU, {T} frames, "
                        f"delta_t_max={args.delta_t_max}, ref_time={REF_TIME}, content={args.content}, crf0 "
                        f"numbers (0,0,10), FramePerfect, {args.multi_mode}, {args.time_mode}, raw events to HBM",
            "plane": [W, H_total, Cn],
            "rows_per_gpu": H_BAND,
            "frames_per_step": T,
            "sharding": ("single GPU" if world == 1 else
                         "row bands; ordered RCCL gather of the event payload to rank 0" if args.gather_payload else
                         "row bands; RCCL all-gather of per-frame event counts fixes the ordered concatenation, "
                         "the event payload stays sharded in HBM (each rank delivers its segments itself)"),
        },
        "events_per_s": round(total_events / (elapsed / args.steps), 1),
        "events_per_pixel_frame": round(e, 5),
        "frame_loop_ms_hip_events": round(kernel_ms, 3),
        "roofline": {
            "bound": "hbm",
            "kernel": "adder_frame_kernel",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "bytes_per_unit_frame": round(bytes_per_unit, 3),
            "frames_per_launch": launch_frames,
            "units_per_launch": int(units_per_launch),
            "launch_avg_us": round(launch_us, 3),
            "note": "default batches step 16 frames per launch with the state in registers: the kernel is then "
                    "VALU-bound, not HBM-bound (DESIGN.md 4); roofline_one_frame_per_launch is the HBM-bound regime",
        },
        "roofline_one_frame_per_launch": {
            "bound": "hbm",
            "kernel": "adder_frame_kernel",
            "achieved": round(achieved1, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved1 / HBM_PEAK_GBS, 4),
            "bytes_per_unit_frame": round(bytes1, 3),
            "units_per_launch": units,
            "launch_avg_us": round(launch1_us, 3),
            "note": "state streamed from HBM every frame (per-frame consume contract); this launch parks compact "
                    "records (1 + 2*20 + 4e bytes per unit; 8e in the AbsoluteT / generic variants), the 12-byte "
                    "events are written by the separate expansion kernel overlapped on a second stream",
        },
    }

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(d_frames, d_events, d_offsets, T, Cn, multi, tmode, args)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(d_frames, d_events, d_offsets, T, Cn, multi, tmode, args):
    """The oracle (a port of the reference CPU path: per-row-chunk tasks collected in order,
    then the serial 9/11-byte raw sink) on all host cores, over a bounded prefix of the SAME
    clip; also used as the checker: the GPU events of those frames must equal it bit-for-bit."""
    from oracle import oracle as O

    threads = O.max_threads()
    v = O.Video(W, H_BAND, Cn, time_mode=tmode, multi_mode=multi, ref_time=REF_TIME,
                delta_t_max=args.delta_t_max, chunk_rows=1, threads=threads)
    v.set_crf_parameters(0, 10)
    v.reset_c_thresh(0)
    v.ensure_capacity(3)
    L = O.lib()
    import ctypes as Ct
    n = Ct.c_size_t(0)
    sink = np.zeros(W * H_BAND * Cn * 3 * 11, np.uint8)
    offs = d_offsets.cpu().numpy()
    frames_done, events, parity_ok = 0, 0, True
    chunk = 8
    t_total = 0.0
    while frames_done < T and t_total < args.cpu_seconds:
        k1 = min(T, frames_done + chunk)
        host = d_frames[frames_done:k1].cpu().numpy()
        counts = []
        t0 = time.perf_counter()
        got_parts = []
        for f in host:
            L.oracle_video_integrate_matrix(v.h, f.ctypes.data, W * Cn, float(REF_TIME), v._out.ctypes.data,
                                            v._cap, Ct.byref(n), None)
            L.oracle_raw_events(sink.ctypes.data, v._out.ctypes.data, n.value, Cn)  # serial sink stage
            counts.append(n.value)
            got_parts.append(v._out[: n.value].copy())
        t_total += time.perf_counter() - t0
        # checker: same events as the GPU produced for these frames
        lo, hi = int(offs[frames_done]), int(offs[k1])
        gpu = np.frombuffer(d_events[lo:hi].cpu().numpy().tobytes(), dtype=O.EVENT_DTYPE)
        cpu = np.concatenate(got_parts) if got_parts else np.zeros(0, O.EVENT_DTYPE)
        parity_ok = parity_ok and len(gpu) == len(cpu) and bool(np.array_equal(gpu, cpu))
        events += sum(counts)
        frames_done = k1
    return {
        "value": round(W * H_BAND * frames_done / t_total / 1e6, 2),
        "unit": "Mpixels/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {frames_done} of {T} frames of the same clip, OpenMP over row chunks "
                  f"(chunk_rows=1) + serial raw sink; includes copies into its own buffers only",
        "events": events,
        "gpu_events_match_bit_exact": parity_ok,
    }


if __name__ == "__main__":
    main()
