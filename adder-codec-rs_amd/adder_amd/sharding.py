"""Row-band sharding of a plane across ranks and the ordered gather of the event stream.

The reference already splits a frame by rows (rayon row chunks, video.rs:677-691) and,
with feature detection off, pixels never interact (integrate_for_px touches only `px`,
video.rs:1318-1380).  So each rank owns a contiguous band of rows for the whole clip
and NO collective is needed while integrating.  The only exchange is the one that
concatenates the emitted events before the (unchanged, serial) sink: per frame the
segments must appear in rank order, which is raster order.  Two forms:
exchange_stream_layout() all-gathers the per-frame counts only (the payload stays sharded
and each rank delivers its segments to their final position itself -- the default of
bench.py), gather_event_stream() also funnels the payload to one rank over xGMI.

Works on CUDA tensors over RCCL (backend "nccl") and on CPU tensors over gloo (tests).
"""
import torch
import torch.distributed as dist


def row_bands(height, world, chunk_rows=1):
    """Contiguous bands [y0, y1) per rank, each a multiple of chunk_rows (except the last)."""
    chunks = (height + chunk_rows - 1) // chunk_rows
    base, extra = divmod(chunks, world)
    bands, c0 = [], 0
    for r in range(world):
        c1 = c0 + base + (1 if r < extra else 0)
        bands.append((min(c0 * chunk_rows, height), min(c1 * chunk_rows, height)))
        c0 = c1
    return bands


def gather_peer_share(world, *, link_GBs=120.0, wire_bytes_per_unit_frame=1.45, root_chunk_us_per_plane=(98.0, 150.0),
                      units=1920 * 1080, chunk=64):
    """Fraction of the rows every PEER should own when the bands' records are gathered to rank 0 over one xGMI link per
    peer (adder_amd.records): the peers' transfers (share * wire bytes / link) and root's own work (its band's frame kernel
    + the expansion of the whole plane) take equally long.  Defaults: what was measured on one MI355X for the 1080p headline
    (0.159 records * 8 B + 3 tables * 4 B / 128 units per unit-frame; round 5: 98 us of the lean-runs frame kernel and 150 us
    of expansion per 64-frame chunk of the whole plane -- round 3 priced 160 + 170) and a conservative link rate.  The times scale
    with the plane (they are given for `units` = 1080p).  Never more than an even split."""
    if world <= 1:
        return 1.0
    wire_us = units * chunk * wire_bytes_per_unit_frame / (link_GBs * 1e3)  # a whole plane's chunk over one link
    scale = units / (1920.0 * 1080.0)  # (the kernels' times grow with the plane like the wire bytes do)
    lean_us, expand_us = root_chunk_us_per_plane[0] * scale, root_chunk_us_per_plane[1] * scale
    # share * wire_us == (1 - (world - 1) * share) * lean_us + expand_us
    share = (lean_us + expand_us) / (wire_us + (world - 1) * lean_us)
    return min(share, 1.0 / world)


def row_bands_root_heavy(height, world, peer_share, chunk_rows=1):
    """Contiguous bands with rank 0 (the gather's root) on top taking what the peers leave: every peer owns
    round(peer_share * height) rows (a multiple of chunk_rows, at least one chunk)."""
    if world == 1:
        return [(0, height)]
    chunks = (height + chunk_rows - 1) // chunk_rows
    per = max(1, min(int(round(peer_share * chunks)), chunks // world))
    root = chunks - per * (world - 1)
    bands, c0 = [], 0
    for r in range(world):
        c1 = c0 + (root if r == 0 else per)
        bands.append((min(c0 * chunk_rows, height), min(c1 * chunk_rows, height)))
        c0 = c1
    return bands


def merge_frame_major(segments):
    """segments[r] = (events int32 [n_r, 3], offsets int64 [T+1]) of rank r, each frame-major.
    Returns (events [sum n_r, 3], offsets [T+1]) with, per frame, rank 0's events first, then
    rank 1's, ... -- the stream a single context over the whole plane would have produced."""
    dev = segments[0][0].device
    T = segments[0][1].numel() - 1
    counts = torch.stack([s[1][1:] - s[1][:-1] for s in segments]).to(dev)  # [R, T]
    frame_tot = counts.sum(0)
    frame_base = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    frame_base[1:] = torch.cumsum(frame_tot, 0)
    before = torch.cumsum(counts, 0) - counts  # events of lower ranks in the same frame
    total = int(frame_base[-1])
    out = torch.empty((total, 3), dtype=torch.int32, device=dev)
    for r, (ev, offs) in enumerate(segments):
        n = ev.shape[0]
        if n == 0:
            continue
        offs = offs.to(dev)
        frame_id = torch.repeat_interleave(torch.arange(T, device=dev), counts[r])
        shift = frame_base[:-1] + before[r] - offs[:-1]
        dest = torch.arange(n, device=dev) + shift[frame_id]
        out[dest] = ev
    return out, frame_base


def exchange_stream_layout(offsets, group=None):
    """The exchange that DEFINES the ordered concatenation without moving the payload: an all-gather of
    every rank's frame offsets (T+1 int64 each).  Returns (frame_base [T+1], my_base [T]): the merged
    stream's frame offsets, and for each frame the position of this rank's segment inside the merged
    stream (= frame_base[f] + events of lower ranks in frame f).  A consumer that copies rank r's
    segment of frame f to my_base_r[f] -- e.g. every rank's own D2H into one pinned host buffer, 8 PCIe
    links instead of one xGMI funnel -- obtains exactly the single-context stream."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    offsets = offsets.contiguous()
    if world == 1:
        return offsets.clone(), offsets[:-1].clone()
    all_offs = [torch.empty_like(offsets) for _ in range(world)]
    dist.all_gather(all_offs, offsets, group=group)
    counts = torch.stack([o[1:] - o[:-1] for o in all_offs])  # [R, T]
    frame_base = torch.zeros_like(offsets)
    frame_base[1:] = torch.cumsum(counts.sum(0), 0)
    before = torch.cumsum(counts, 0) - counts
    return frame_base, frame_base[:-1] + before[rank]


def place_segments(out, events, offsets, my_base):
    """Copies this rank's per-frame segments to their place in the merged stream `out` ([N, 3] int32 or
    any row-indexable buffer).  Reference implementation of the consumer side (tests / host sink)."""
    T = offsets.numel() - 1
    for f in range(T):
        a, b = int(offsets[f]), int(offsets[f + 1])
        if b > a:
            d = int(my_base[f])
            out[d:d + (b - a)] = events[a:b]
    return out


def gather_event_stream(events, offsets, dst=0, group=None, video=None):
    """Ordered variable-length gather of every rank's frame-major event segment to `dst`.
    events: int32 [n, 3] (the 12-byte records), offsets: int64 [T+1].  Returns the merged
    (events, offsets) on dst and None elsewhere.

    Transport: an all-gather of the offsets, then grouped point-to-point transfers (RCCL over xGMI
    on CUDA tensors) straight into one staging buffer on dst, rank after rank.  Merge: on CUDA tensors
    the HIP merge kernel of libadder_hip.so through `video` (a HipVideo on dst's device;
    adder_hip_merge_streams_device); the torch index arithmetic of merge_frame_major only serves CPU
    tensors (gloo tests)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return events, offsets
    dev = events.device
    # gloo has no device point-to-point: CUDA tensors then travel through host copies (single-GPU debug
    # boxes, where RCCL cannot place two ranks on one device); the merge still runs on the device
    via_host = events.is_cuda and dist.get_backend(group) == "gloo"
    tdev = torch.device("cpu") if via_host else dev
    offsets = offsets.contiguous().to(tdev)
    T = offsets.numel() - 1
    all_offs = torch.empty((world, T + 1), dtype=offsets.dtype, device=tdev)
    if tdev.type == "cuda":
        dist.all_gather_into_tensor(all_offs, offsets, group=group)
    else:
        dist.all_gather(list(all_offs.unbind(0)), offsets, group=group)
    totals = all_offs[:, -1].tolist()
    ops, stage = [], None
    if rank == dst:
        stage = torch.empty((int(sum(totals)), 3), dtype=torch.int32, device=tdev)
        pos = 0
        for r in range(world):
            n_r = int(totals[r])
            if r == dst:
                stage[pos:pos + n_r] = events[:n_r].to(tdev)
            elif n_r:
                ops.append(dist.P2POp(dist.irecv, stage[pos:pos + n_r], r, group))
            pos += n_r
    elif events.shape[0]:
        ops.append(dist.P2POp(dist.isend, events.contiguous().to(tdev), dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    if via_host:
        stage, all_offs = stage.to(dev), all_offs.to(dev)
    if events.is_cuda:
        if video is None:
            raise ValueError("gather_event_stream on CUDA tensors needs video= (the HIP merge kernel's context)")
        out = torch.empty_like(stage)
        merged_offs = torch.empty(T + 1, dtype=torch.int64, device=events.device)
        video.merge_streams_device(stage, all_offs, world, T, out, merged_offs,
                                   stream=torch.cuda.current_stream().cuda_stream)
        return out, merged_offs
    pos, segs = 0, []
    for r in range(world):
        n_r = int(totals[r])
        segs.append((stage[pos:pos + n_r], all_offs[r]))
        pos += n_r
    return merge_frame_major(segs)


class ChunkPipelinedGather:
    """The ordered gather, chunk by chunk behind the integration ("after each frame (or batch of T frames)").

    The caller integrates its band one chunk of frames at a time and hands every finished chunk to push(); the chunk's
    exchange (all-gather of its per-frame offsets, point-to-point payload to `dst`, merge into the growing merged
    stream) is issued on a SIDE stream, so it runs while the next chunk is being integrated on the caller's stream.
    result() waits for the last chunk and returns the merged (events, offsets) on dst.  Works on CUDA tensors (RCCL, or
    gloo through host copies on single-GPU debug boxes) and on CPU tensors over gloo (tests)."""

    def __init__(self, total_frames, merged_cap_events=0, dst=0, group=None, video=None, device=None):
        self.group, self.dst, self.video = group, dst, video
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.T = total_frames
        self.device = torch.device("cpu") if device is None else torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.frame_pos, self.merged_pos = 0, 0
        self.merged = self.merged_offs = None
        self._keep = []  # tensors the side stream still reads
        if self.rank == dst:
            self.merged = torch.empty((max(merged_cap_events, 1), 3), dtype=torch.int32, device=self.device)
            self.merged_offs = torch.zeros(total_frames + 1, dtype=torch.int64, device=self.device)

    def reset(self):
        self.frame_pos, self.merged_pos = 0, 0
        self._keep = []

    def push(self, events, offsets):
        """events: this rank's events of the chunk (int32 [n, 3]); offsets: int64 [Tc + 1] (any start value).  The
        tensors must stay untouched until result() (the exchange reads them asynchronously)."""
        Tc = offsets.numel() - 1
        if self.world == 1:
            n = int(offsets[-1] - offsets[0])
            self.merged[self.merged_pos:self.merged_pos + n] = events[:n]
            self.merged_offs[self.frame_pos:self.frame_pos + Tc + 1] = offsets - offsets[0] + self.merged_pos
            self.merged_pos += n
            self.frame_pos += Tc
            return
        via_host = self.cuda and dist.get_backend(self.group) == "gloo"
        tdev = torch.device("cpu") if (via_host or not self.cuda) else self.device
        ctx = torch.cuda.stream(self.side) if self.cuda else _NullCtx()
        if self.cuda:
            self.side.wait_stream(torch.cuda.current_stream(self.device))  # the chunk's events are complete behind this
        with ctx:
            # every row carries one more word: the room left in the rank's merged buffer (dst's is the one that counts).
            # A chunk that does not fit is then refused by EVERY rank, before any point-to-point operation is posted --
            # an error raised on dst alone would leave the peers in the next chunk's all-gather (adder_gather_events_at
            # agrees on such failures with an all-reduce in front of the payload for the same reason)
            room = (self.merged.shape[0] - self.merged_pos) if self.rank == self.dst else -1
            row = torch.cat([(offsets - offsets[0]).to(tdev), torch.tensor([room], dtype=torch.int64, device=tdev)]).contiguous()
            all_rows = torch.empty((self.world, Tc + 2), dtype=torch.int64, device=tdev)
            if tdev.type == "cuda":
                dist.all_gather_into_tensor(all_rows, row, group=self.group)
            else:
                dist.all_gather(list(all_rows.unbind(0)), row, group=self.group)
            all_offs = all_rows[:, :Tc + 1].contiguous()
            totals = all_offs[:, -1].tolist()  # (waits for the all-gather only: the next chunk keeps integrating)
            total = int(sum(totals))
            dst_room = int(all_rows[self.dst, Tc + 1])
            if total > dst_room:
                raise RuntimeError(f"merged buffer too small: need {total} more events, room for {dst_room} "
                                   f"(every rank refuses the chunk; nothing was sent)")
            ops, stage = [], None
            if self.rank == self.dst:
                stage = torch.empty((max(total, 1), 3), dtype=torch.int32, device=tdev)
                pos = 0
                for r in range(self.world):
                    n_r = int(totals[r])
                    if r == self.dst:
                        stage[pos:pos + n_r] = events[:n_r].to(tdev)
                    elif n_r:
                        ops.append(dist.P2POp(dist.irecv, stage[pos:pos + n_r], r, self.group))
                    pos += n_r
            elif int(totals[self.rank]):
                send = events[:int(totals[self.rank])].contiguous().to(tdev)
                self._keep.append(send)
                ops.append(dist.P2POp(dist.isend, send, self.dst, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if self.rank == self.dst:
                if self.cuda:
                    if via_host:
                        stage, all_offs = stage.to(self.device), all_offs.to(self.device)
                    self._keep += [stage, all_offs]
                    self.video.merge_streams_device(stage, all_offs, self.world, Tc, self.merged[self.merged_pos:],
                                                    self.merged_offs[self.frame_pos:], stream=self.side.cuda_stream,
                                                    merged_base=self.merged_pos)
                else:
                    pos, segs = 0, []
                    for r in range(self.world):
                        n_r = int(totals[r])
                        segs.append((stage[pos:pos + n_r], all_offs[r]))
                        pos += n_r
                    ev, mo = merge_frame_major(segs)
                    self.merged[self.merged_pos:self.merged_pos + total] = ev
                    self.merged_offs[self.frame_pos:self.frame_pos + Tc + 1] = mo + self.merged_pos
            self.merged_pos += total
            self.frame_pos += Tc

    def result(self):
        if self.cuda:
            self.side.synchronize()
            if self.rank == self.dst and self.video is not None and self.world > 1:
                self.video.check_status(self.side.cuda_stream)
        self._keep = []
        if self.rank != self.dst:
            return None
        return self.merged[:self.merged_pos], self.merged_offs


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class FeatureBands:
    """Feature-driven rate control / ROI over row bands held by ONE process (several contexts on one device, or one per
    device with peer access): drives the per-frame protocol of include/adder_hip.h -- integrate the frame on every
    band, exchange the 3-row halos of the running intensities, run the feature step per band, hand every band the
    other bands' new features.  A multi-process host does the same with RCCL send / recv between its ranks."""

    def __init__(self, videos):
        import torch
        self.videos = list(videos)  # top to bottom
        self.torch = torch
        hb = self.videos[0].L.adder_hip_feature_halo_bytes(self.videos[0].h)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.top = [torch.zeros(hb, dtype=torch.uint8, device=dev) for _ in self.videos]
        self.bottom = [torch.zeros(hb, dtype=torch.uint8, device=dev) for _ in self.videos]
        self.lists = [torch.zeros(1 << 16, dtype=torch.int32, device=dev) for _ in self.videos]
        self.new_features = 0

    def integrate_matrix(self, frame, time_spanned=None):
        """frame: the whole plane [H, W, C]; returns the merged events of the frame (bands in order = raster order)."""
        import ctypes as C
        import numpy as np
        from . import _native as N
        vs = self.videos
        out = [v.integrate_matrix(frame[v.row_begin:v.row_end], time_spanned=time_spanned) for v in vs]
        for k, v in enumerate(vs):
            N.check(v.h, v.L.adder_hip_feature_halo_export(v.h, self.top[k].data_ptr(), self.bottom[k].data_ptr(), None))
        self.torch.cuda.synchronize()
        for k, v in enumerate(vs):
            above = self.bottom[k - 1].data_ptr() if k > 0 else None
            below = self.top[k + 1].data_ptr() if k + 1 < len(vs) else None
            N.check(v.h, v.L.adder_hip_feature_halo_import(v.h, above, below, None))
        counts = []
        for k, v in enumerate(vs):
            n = C.c_uint32(0)
            N.check(v.h, v.L.adder_hip_feature_detect(v.h, self.lists[k].data_ptr(), self.lists[k].numel(), C.byref(n), None))
            counts.append(n.value)
        for k, v in enumerate(vs):
            for j in range(len(vs)):
                if j != k and counts[j]:
                    N.check(v.h, v.L.adder_hip_feature_apply(v.h, self.lists[j].data_ptr(), counts[j], None))
        self.torch.cuda.synchronize()
        self.new_features = sum(counts)
        return np.concatenate(out)
