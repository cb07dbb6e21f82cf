"""Records over the wire: the multi-GPU gather that ships every band's PARKED RECORDS (0.35x the bytes of its events) and
lets root expand them (include/adder_hip.h: adder_hip_integrate_records_device / adder_hip_expand_records_device /
adder_hip_records_to_wire).  torch.distributed is the transport (backend "nccl" = RCCL over xGMI; gloo through host copies
on single-GPU debug boxes); nothing here computes."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _native as N


def wire_bytes(num_frames, num_segments, record_bytes, n_records):
    return int(N.load().adder_hip_records_wire_bytes(num_frames, num_segments, record_bytes, int(n_records)))


def wire_sections(num_frames, num_segments, record_bytes):
    sec = (C.c_size_t * 6)()
    N.load().adder_hip_records_wire_sections(num_frames, num_segments, record_bytes, sec)
    return [int(x) for x in sec]


def records_from_wire(d_img, num_frames, num_segments, record_bytes, row_begin, rows, d_frame_table=None):
    """AdderBandRecords over a received image (uint8 CUDA tensor, adder_hip_records_to_wire's layout)."""
    sec = wire_sections(num_frames, num_segments, record_bytes)
    base = d_img.data_ptr()
    rec = N.AdderBandRecords()
    rec.num_frames, rec.num_segments, rec.record_bytes = num_frames, num_segments, record_bytes
    rec.row_begin, rec.rows = row_begin, rows
    rec.d_frame_offsets = base + sec[0]
    rec.d_frame_table = (base + sec[1]) if d_frame_table is None else d_frame_table
    rec.d_counts, rec.d_prefix, rec.d_runs, rec.d_records = base + sec[2], base + sec[3], base + sec[4], base + sec[5]
    return rec


class RecordsPipelinedGather:
    """The ordered gather of SURVEY 8(e), chunk by chunk behind the integration, with records instead of events on the wire.

    Every rank integrates its band one chunk (<= chunk_frames() frames) at a time with integrate_records_device + finish and
    hands the chunk to push(): the batch is copied into one contiguous image on the caller's stream (the context is then
    free for its next chunk), and on a SIDE stream the ranks exchange the images' sizes, the peers send their images to
    `dst`, and dst expands every band's records -- its own included -- into the growing merged frame-major stream.
    result() waits for the last chunk and returns (merged events, merged offsets) on dst."""

    def __init__(self, total_frames, video, merged_cap_events=0, dst=0, group=None, device=None):
        self.group, self.dst, self.video = group, dst, video
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.T = total_frames
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # two side streams in turn: chunk k + 1's images arrive while chunk k is being expanded (one in-order stream
        # would put every transfer behind the expansion before it)
        self.sides = [torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)]
        self.chunk_no = 0
        self.frame_pos, self.merged_pos = 0, 0
        self.merged = self.merged_offs = None
        self._keep = []
        if self.rank == dst:
            self.merged = torch.empty((max(merged_cap_events, 1), 3), dtype=torch.int32, device=self.device)
            self.merged_offs = torch.zeros(total_frames + 1, dtype=torch.int64, device=self.device)

    def reset(self):
        self.frame_pos, self.merged_pos = 0, 0
        self.chunk_no = 0
        self._keep = []

    def push(self, rec, n_records, n_events):
        """rec: what integrate_records_device returned for the chunk (after finish()); n_records =
        last_batch_records(), n_events = finish()'s count.  Returns the bytes this rank put on the wire."""
        v = self.video
        nf, nseg, rb = int(rec.num_frames), int(rec.num_segments), int(rec.record_bytes)
        nbytes = wire_bytes(nf, nseg, rb, n_records)
        img = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        # the image is copied on the stream the BATCH ran on (which may be the context's own, not a torch stream: the
        # handle of torch's default stream is 0 = "the context's"), and waited for here: a copy of this band's records and
        # tables, tens of microseconds -- after it the scratch is free for the next chunk and the side stream may send
        v.records_to_wire(rec, n_records, img, stream=None)
        v.sync_last_batch_stream()
        self._keep.append(img)
        sent = 0
        side = self.sides[self.chunk_no % 2]
        self.chunk_no += 1
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # (the last word: the room left in the rank's merged buffer -- dst's is the one that counts; a chunk that does
            # not fit is refused by EVERY rank before any point-to-point operation is posted, see ChunkPipelinedGather)
            room = (self.merged.shape[0] - self.merged_pos) if self.rank == self.dst else -1
            meta = torch.tensor([nf, nseg, rb, int(n_records), int(rec.row_begin), int(rec.rows), int(n_events), room],
                                dtype=torch.int64)
            if self.world > 1:
                via_host = dist.get_backend(self.group) == "gloo"
                tdev = torch.device("cpu") if via_host else self.device
                all_meta = torch.empty((self.world, 8), dtype=torch.int64, device=tdev)
                m = meta.to(tdev)
                if tdev.type == "cuda":
                    dist.all_gather_into_tensor(all_meta, m, group=self.group)
                else:
                    dist.all_gather(list(all_meta.unbind(0)), m, group=self.group)
                metas = all_meta.tolist()
            else:
                via_host, tdev, metas = False, self.device, [meta.tolist()]
            if any(int(mm[0]) != nf for mm in metas):
                raise RuntimeError("the ranks pushed chunks of different lengths")
            total = int(sum(int(mm[6]) for mm in metas))
            if total > int(metas[self.dst][7]):
                raise RuntimeError(f"merged buffer too small: need {total} more events, room for {int(metas[self.dst][7])} "
                                   f"(every rank refuses the chunk; nothing was sent)")
            ops, imgs = [], [None] * self.world
            if self.rank == self.dst:
                for r in range(self.world):
                    if r == self.dst:
                        imgs[r] = img
                    else:
                        b_r = wire_bytes(int(metas[r][0]), int(metas[r][1]), int(metas[r][2]), int(metas[r][3]))
                        imgs[r] = torch.empty(b_r, dtype=torch.uint8, device=tdev)
                        ops.append(dist.P2POp(dist.irecv, imgs[r], r, self.group))
            else:
                send = img.to(tdev) if via_host else img
                self._keep.append(send)
                ops.append(dist.P2POp(dist.isend, send, self.dst, self.group))
                sent = nbytes
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if self.rank == self.dst:
                if via_host:
                    imgs = [im if im.device.type == "cuda" else im.to(self.device) for im in imgs]
                self._keep += imgs
                sec_own = wire_sections(nf, nseg, rb)
                ftab = imgs[self.dst].data_ptr() + sec_own[1]  # root's own frame table of these frames
                bands = [records_from_wire(imgs[r], int(metas[r][0]), int(metas[r][1]), int(metas[r][2]), int(metas[r][4]),
                                           int(metas[r][5]), d_frame_table=ftab) for r in range(self.world)]
                v.expand_records_device(bands, self.merged, self.merged_pos, self.merged_offs[self.frame_pos:],
                                        stream=side.cuda_stream)
            self.merged_pos += total
            self.frame_pos += nf
        return sent

    def result(self):
        for sd in self.sides:
            sd.synchronize()
        if self.rank == self.dst:
            self.video.expand_status(self.sides[0].cuda_stream)
        self._keep = []
        if self.rank != self.dst:
            return None
        return self.merged[:self.merged_pos], self.merged_offs
