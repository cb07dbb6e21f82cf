/* adder_gather.h -- C-ABI of the multi-GPU event-stream gather (libadder_rccl.so).
 *
 * a plane is split into contiguous row bands, one AdderHipCtx per GPU / rank
 * (the reference's own split is the rayon row chunking of video.rs:677-691; pixels never interact
 * with feature detection off, video.rs:1318-1380).  Nothing is exchanged while integrating.  This
 * library does the one exchange the path has: the ordered concatenation of the ranks' event
 * streams on one rank before the unchanged CPU sink (video.rs:742-765 writes the chunks' events in
 * row order) -- an RCCL all-gather of the per-frame offsets, grouped ncclSend / ncclRecv of the
 * payloads over xGMI, and the merge kernel of libadder_hip.so (adder_hip_merge_streams_device).
 *
 * A Rust host binds this with `extern "C"` and passes the ncclComm_t it created with RCCL; no torch,
 * no Python.  The library links librccl only; libadder_hip.so has no RCCL dependency. */
#ifndef ADDER_GATHER_H
#define ADDER_GATHER_H

#include <stddef.h>
#include <stdint.h>

#include "adder_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct AdderGather AdderGather;

#define ADDER_GATHER_UNIQUE_ID_BYTES 128

/* ncclGetUniqueId for hosts that have no RCCL binding of their own: rank 0 calls this, ships the 128
 * bytes to the other ranks by any means, every rank calls adder_gather_create_from_id. */
int adder_gather_unique_id(uint8_t id_out[ADDER_GATHER_UNIQUE_ID_BYTES]);

/* `nccl_comm`: an ncclComm_t owned by the caller (not destroyed by adder_gather_destroy).
 * `ctx`: this rank's integration context (its device, its status word, its merge scratch). */
int adder_gather_create(AdderHipCtx *ctx, void *nccl_comm, int rank, int world, AdderGather **out);
/* Creates (and owns) a communicator from a unique id: ncclCommInitRank(world, id, rank). */
int adder_gather_create_from_id(AdderHipCtx *ctx, const uint8_t id[ADDER_GATHER_UNIQUE_ID_BYTES], int rank,
                                int world, AdderGather **out);

/* The transport under the gather: what it needs from a communicator, as a table of functions.  RCCL is one
 * implementation (the two constructors above); a host with another communicator (MPI, its own xGMI / shared-memory
 * code) supplies its own.  All buffers are device memory, all sizes bytes; operations are ordered by `stream` like RCCL's
 * and may block.  send / recv are only called between group_start and group_end, and group_end(stream) posts them all.
 * Every function returns ADDER_OK or a negative AdderStatus; `error` (may be null) describes the last failure. */
typedef struct AdderTransport {
    void *self;
    int (*all_gather)(void *self, const void *send, void *recv, size_t bytes_per_rank, void *stream);
    int (*all_reduce_max)(void *self, int32_t *d_word, void *stream); /* one int32, in place */
    int (*group_start)(void *self);
    int (*send)(void *self, const void *buf, size_t bytes, int peer, void *stream);
    int (*recv)(void *self, void *buf, size_t bytes, int peer, void *stream);
    int (*group_end)(void *self, void *stream);
    const char *(*error)(void *self);
} AdderTransport;
/* The table is copied; `self` must outlive the gather object. */
int adder_gather_create_with_transport(AdderHipCtx *ctx, const AdderTransport *transport, int rank, int world,
                                       AdderGather **out);

/* An in-process transport: the ranks are THREADS of one process, one context each (any devices, also all on one --
 * where RCCL refuses two ranks per GPU).  Every operation is a blocking rendezvous with synchronous device copies: slow,
 * but the protocol above it is the production one (tests/test_gpu_gather_local.py).  A rank that does not turn up within
 * 30 s fails the operation on every rank (ADDER_E_TIMEOUT) instead of hanging the process. */
typedef struct AdderLocalGroup AdderLocalGroup;
int adder_gather_local_group_create(int world, AdderLocalGroup **out);
void adder_gather_local_group_destroy(AdderLocalGroup *group); /* after every gather object made from it */
int adder_gather_create_local(AdderHipCtx *ctx, AdderLocalGroup *group, int rank, AdderGather **out);

void adder_gather_destroy(AdderGather *g);
const char *adder_gather_last_error(const AdderGather *g);
int adder_gather_world(const AdderGather *g);

/* Ordered gather of one batch.  Every rank passes its own stream (d_events: n events, device;
 * d_frame_offsets: device uint64[T+1], [T] == n) -- what adder_hip_integrate_device produced.  On `root`
 * the merged frame-major stream is written to d_merged (capacity merged_cap events, device) and its
 * offsets to d_merged_offsets (device uint64[T+1]); *n_merged = its length.  Other ranks pass NULL / 0
 * for the merged buffers.  Collective: every rank of the communicator must call it with the same T
 * and root.  Synchronises `stream` before returning (the totals are needed on the host). */
int adder_gather_events(AdderGather *g, const AdderEvent *d_events, const uint64_t *d_frame_offsets,
                        uint32_t num_frames, int root, AdderEvent *d_merged, size_t merged_cap,
                        uint64_t *d_merged_offsets, size_t *n_merged, void *stream);

/* The same for ONE CHUNK of the streams, so that the exchange of chunk k runs (on its own stream) while chunk k+1 is
 * integrated: d_frame_offsets is the entry of the chunk's first frame in the rank's offsets (uint64[T+1], values as
 * adder_hip_integrate_device left them: they need not start at 0), d_events the rank's whole event buffer, and the
 * chunk's merged events are appended to d_merged behind merged_base events; d_merged_offsets is the entry of the
 * chunk's first frame in the merged offsets, which continue from merged_base.  *n_merged = the chunk's merged events.
 * A failure only one rank can see (root's staging allocation) is agreed on before the payload moves: every rank
 * returns an error, none hangs. */
int adder_gather_events_at(AdderGather *g, const AdderEvent *d_events, const uint64_t *d_frame_offsets,
                           uint32_t num_frames, int root, AdderEvent *d_merged, size_t merged_cap, uint64_t merged_base,
                           uint64_t *d_merged_offsets, size_t *n_merged, void *stream);

/* RECORDS over the wire, one chunk per call, complete when the call returns (two host waits per chunk: the sizes, the
 * agreement; the streamed form below has none) (include/adder_hip.h: AdderBandRecords).  The events of a band cross ONE xGMI link on their way to
 * root; its parked records are 0.35x the bytes.  Every rank integrates a chunk of at most adder_hip_chunk_frames() frames
 * with adder_hip_integrate_records_device + adder_hip_finish and passes the description here with
 * adder_hip_last_batch_records() and the finish count: the ranks exchange the sizes, every peer sends one contiguous image
 * of its chunk (adder_hip_records_to_wire), root expands every band's records -- its own included -- into d_merged behind
 * merged_base events (adder_hip_expand_records_device) and writes the chunk's merged offsets (absolute: [0] =
 * merged_base).  The image is this object's (two chunks' worth in turn): on return the caller's context is free for its
 * next chunk while the transfers and root's expansion are still queued on `stream`; a merged buffer found too small by
 * the expansion itself is reported by adder_hip_expand_status(ctx, stream).  Lean regime only (the records call refuses
 * the others: gather events then).  replaces the same lines of video.rs:677-734 / SURVEY 8(e) as adder_gather_events_at. */
int adder_gather_records_at(AdderGather *g, const AdderBandRecords *rec, uint64_t n_records, uint64_t n_events, int root,
                            AdderEvent *d_merged, size_t merged_cap, uint64_t merged_base, uint64_t *d_merged_offsets,
                            size_t *n_merged, void *stream);

/* The same, STREAMED: the host is out of the per-chunk loop.  begin() names the destination of a whole clip (root: the
 * merged buffer, where the clip starts in it, the clip's merged offsets [T + 1]; the others pass nulls) and the stream
 * every exchange is queued on; push() takes one chunk exactly like adder_gather_records_at, but only copies the chunk's
 * image, queues the all-gather of ITS sizes (device -> pinned host memory, an event behind them) and then posts the
 * PAYLOAD OF THE CHUNK BEFORE, whose sizes arrived a chunk ago -- it waits for nothing that is not long done; end() posts
 * the last chunk's payload, waits for the stream and reports (root: the clip's merged events, a merged buffer that was too
 * small -- the expansion drops what does not fit --; the others: the bytes they sent).  No agreement round per chunk: the
 * image buffers hold a chunk's worst case (one record per unit and frame) from the object's first push on -- that push
 * alone agrees on the allocations with a blocking all-reduce --, a rank's local failure travels in its own row of the
 * sizes, and rows that do not fit together are seen by every rank alike, so every rank takes the same way out and none
 * is left in a send.  One clip at a time per object; adder_gather_records_at must not be mixed into an open stream. */
int adder_gather_records_begin(AdderGather *g, int root, AdderEvent *d_merged, size_t merged_cap, uint64_t merged_base,
                               uint64_t *d_merged_offsets, void *stream);
/* The same clip with the raw sink's records as root's output: d_wire receives the merged stream as 9 / 11-byte records back
 * to back (what adder_hip_integrate_wire_device leaves on one GPU -- the bytes of the .adder file between header and EOF,
 * raw/stream.rs:101-120); merged_base, the offsets and end()'s count are in events as before. */
int adder_gather_records_begin_wire(AdderGather *g, int root, uint8_t *d_wire, size_t wire_cap_bytes, uint64_t merged_base,
                                    uint64_t *d_merged_offsets, void *stream);
/* Failure rules of a streamed clip: push(k + 1) completes chunk k -- checks its gathered rows, which every rank reads alike --
 * BEFORE it queues anything of chunk k + 1, so a rank that finds a failed row (a peer's failure flag, chunks of different
 * lengths or record kinds, a chunk beyond the agreed worst case) returns the error without having posted a collective
 * the others would wait for, and every rank returns it at the same chunk.  A rank whose own push fails locally has
 * queued that chunk's all-gather (its row carries the flag) and must post nothing more: further pushes of the clip return
 * ADDER_E_BAD_PARAMS at once, end() returns the failure and leaves the object ready for the next begin. */
int adder_gather_records_push(AdderGather *g, const AdderBandRecords *rec, uint64_t n_records, uint64_t n_events);
int adder_gather_records_end(AdderGather *g, size_t *n_merged, uint64_t *bytes_sent);
/* Host time spent inside adder_gather_records_push since the last begin, microseconds (diagnostics). */
double adder_gather_records_host_us(const AdderGather *g);

/* SINK PER RANK ("each GPU D2H's its own segment and the host concatenates -- 8 PCIe links vs one"; the
 * consumer is the raw sink, video.rs:736-740 -> encoder.rs:233-273 -> raw/stream.rs:101-120).  Nothing is funnelled into
 * one GPU: `image` is the .adder file itself, mapped into every rank's process and registered with HIP (a POSIX
 * shared-memory file, hipHostRegister(.., hipHostRegisterMapped): pass the DEVICE pointer), header_bytes of it reserved
 * for the header the host writes.  Per chunk every rank passes what adder_hip_integrate_device left it (events, frame
 * offsets -- any start value); the ranks all-gather the chunk's offsets, a kernel derives where this rank's segment of
 * every frame belongs, and a second one serialises the segments to 9 / 11-byte wire records and stores them straight to
 * their final bytes -- over the rank's OWN PCIe link.  Everything is queued on `stream`; the file position lives on the
 * device, the host waits for nothing until close(), which returns the events the image holds (the host then appends
 * the EOF record and truncates the file).  Bytes past image_bytes are dropped and reported by close(). */
int adder_gather_host_sink_open(AdderGather *g, void *image, uint64_t image_bytes, uint64_t header_bytes, void *stream);
int adder_gather_host_sink_chunk(AdderGather *g, const AdderEvent *d_events, const uint64_t *d_frame_offsets,
                                 uint32_t num_frames, void *stream);
int adder_gather_host_sink_close(AdderGather *g, uint64_t *total_events, void *stream);

/* The image for the sinks: a POSIX shared-memory file (shm_open(name): /dev/shm/<name> IS the .adder file) of `bytes`
 * bytes, mapped into this process and registered with HIP so that kernels store into it.  create != 0: the rank that
 * makes the file (the others open it after that rank says so, by any means).  close(): final_bytes >= 0 truncates the
 * file to its real length; unlink_file removes it (benchmarks). */
typedef struct AdderHostImage AdderHostImage;
int adder_host_image_open(const char *name, uint64_t bytes, int create, AdderHostImage **out);
void *adder_host_image_host_ptr(const AdderHostImage *image);
void *adder_host_image_device_ptr(const AdderHostImage *image);
int adder_host_image_close(AdderHostImage *image, int64_t final_bytes, int unlink_file);

/* Layout-only exchange: all-gathers the per-frame offsets and returns, on every rank, the merged
 * stream's frame offsets (h_merged_offsets, host uint64[T+1]) and where this rank's segment of each
 * frame belongs in it (h_my_base, host uint64[T]).  The payload stays sharded: every rank can deliver
 * its own segments (8 PCIe links instead of one xGMI funnel). */
int adder_gather_layout(AdderGather *g, const uint64_t *d_frame_offsets, uint32_t num_frames,
                        uint64_t *h_merged_offsets, uint64_t *h_my_base, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ADDER_GATHER_H */
