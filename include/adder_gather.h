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

/* RECORDS over the wire (include/adder_hip.h: AdderBandRecords).  The events of a band cross ONE xGMI link on their way to
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
