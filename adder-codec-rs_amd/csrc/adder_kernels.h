// adder_kernels.h -- launch interface between the C-ABI (adder_hip_api.cpp) and the
// gfx950 kernels (adder_kernels.hip).  Internal; the public boundary is include/adder_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "adder_pixel.hpp"

namespace adder {

constexpr uint32_t kBlockThreads = 256;
#ifndef ADDER_UNITS_PER_LANE
#define ADDER_UNITS_PER_LANE 2
#endif
#ifndef ADDER_EXPAND_SEGS
#define ADDER_EXPAND_SEGS 16
#endif
#ifndef ADDER_LEAN1_WIDE
#define ADDER_LEAN1_WIDE 1  // one frame per launch: the 4-units-per-lane kernel (16-byte accesses)
#endif
#ifndef ADDER_SCAN_THREADS
#define ADDER_SCAN_THREADS 256
#endif
#ifndef ADDER_SCAN_TILE_WAVES
#define ADDER_SCAN_TILE_WAVES 16384  // segments one scan workgroup takes (16 uint4 groups per thread: from registers)
#endif
#ifndef ADDER_LEAN_WAVES_PER_SIMD
#define ADDER_LEAN_WAVES_PER_SIMD 8
#endif
constexpr uint32_t kExpandSegs = ADDER_EXPAND_SEGS;                // segments one expansion wave takes of a frame
constexpr uint32_t kUnitsPerLane = ADDER_UNITS_PER_LANE;           // pixel-channels per lane (2 or 4)
constexpr uint32_t kWaveUnits = 64 * kUnitsPerLane;                // units per wave segment
constexpr uint32_t kTileUnits = kBlockThreads * kUnitsPerLane;     // units per K1 block
constexpr uint32_t kLeanWavesPerSimd = ADDER_LEAN_WAVES_PER_SIMD;  // register budget of the lean K1
#ifndef ADDER_MAX_FRAMES_PER_LAUNCH
#define ADDER_MAX_FRAMES_PER_LAUNCH 64
#endif
constexpr uint32_t kMaxFramesPerLaunch = ADDER_MAX_FRAMES_PER_LAUNCH;  // temporal blocking depth of K1 (<= 64)
constexpr uint32_t kMaxChunk = kMaxFramesPerLaunch;                // frames per scan/expand launch
// parked-record scratch of one segment of one frame, in BYTES
constexpr uint32_t kLeanRecBytes = 12;                             // LeanRec {ta, tc, w}: at most one per unit (AbsoluteT)
constexpr uint32_t kLeanRec8Bytes = 8;                             // {ta, w8}: DeltaT batches (adder_pixel.hpp lean_decode8)
constexpr uint32_t kLeanParkBytes = kWaveUnits * kLeanRecBytes;
constexpr uint32_t kLeanPark8Bytes = kWaveUnits * kLeanRec8Bytes;
__host__ __device__ constexpr uint32_t lean_rec_bytes(bool abs_t) { return abs_t ? kLeanRecBytes : kLeanRec8Bytes; }
constexpr uint32_t kGenRecBytes = 8;                               // generic variants: one per EVENT

// bits of the device status word
constexpr uint32_t kStatusCapacity = 1u;  // an event did not fit into the output buffer
constexpr uint32_t kStatusWire = 4u;      // wire serialisation met c = None on a multi-channel plane
constexpr uint32_t kStatusDepth = 2u;     // a pixel needed more than max_depth stored levels
constexpr uint32_t kStatusSparse = 8u;    // a sparse step names a pixel outside the plane / band
constexpr uint32_t kStatusScratch = 16u;  // a segment's record log did not hold its bound (an internal error)
constexpr uint32_t kStatusLeanRuns = 32u; // adder_lr_kernel met a unit whose popped_dtm is not (base_val != 0) (an internal error)

struct AdderEventPod {  // same layout as AdderEvent (include/adder_hip.h)
    uint16_t x, y;
    uint8_t c, d;
    uint16_t pad;
    uint32_t t;
};

// what is uniform across the plane but changes per frame
struct FrameTab {
    float running_t;  // PixelArena::running_t BEFORE the frame's integrate
    uint32_t cth;     // c_thresh of every pixel while the frame is tested (adder_pixel.hpp header comment)
};

// Everything the per-frame kernels need.  `f` = frame index inside the batch.
struct FrameArgs {
    // structure-of-arrays pixel state, resident in HBM across frames.  Level 0 of a unit is the 16 bytes
    // {hdr, integ0, dt0, bdt0}; levels k >= 1 (generic variants only) live in the deep planes at k - 1.
    uint32_t *hdr;      // [n_pad] base_val | best_d(level 0) << 8 | m << 16 | popped << 21
    float *integ0;      // [n_pad]
    float *dt0;         // [n_pad]
    float *bdt0;        // [n_pad]
    float *lastf;       // [n_pad] last_fired_t (AbsoluteT)
    float *dv_integ;    // [max_depth - 1][n_pad]   (nullptr until the first generic batch)
    float *dv_dt;
    float *dv_bdt;
    uint8_t *dv_bd;     // best_d (d = fired_d(best_d))
    // Mode::Continuous contexts: the general arena, node k of a unit at [k][unit] (hdr then holds the arena header)
    float *cn_integ, *cn_dt, *cn_bdt;
    uint32_t *cn_meta;  // d | best_d << 8 | has_best << 16
    uint32_t max_nodes; // max_depth + 1
    uint32_t stage_events;  // staged records per unit (max_depth + 3), see adder_cont_kernel
    uint8_t *running;   // optional running_intensities side plane, or nullptr
    // feature-driven rate control / ROI: c_thresh and c_increase_counter per unit, generic K1 only;
    // nullptr while they are uniform (FrameTab::cth)
    uint8_t *cth_px, *cctr_px;
    uint32_t c_max, c_vel;  // CrfParameters::{c_thresh_max, c_increase_velocity} of the per-unit adaptation
    size_t plane_stride;  // n_pad
    // this frame
    const uint8_t *frame;  // n_units bytes, packed [rows][width][channels]
    AdderEventPod *out;
    uint64_t out_cap;
    uint64_t *frame_offsets;  // [frame_idx] = first event of the frame, [frame_idx+1] = end
    uint32_t frame_idx;
    // ordered compaction, stage 1 (frame kernel): per wave segment
    uint8_t *park;        // the frame's parked records of segment 0 (LeanRec, or {t, d | unit<<8 | offset<<16}); see park_offset
    uint32_t *wtot;       // [num_waves] events of the segment (low 16) | parked records (high 16)
    // stage 2 (scan kernel): exclusive prefix of the low halves of wtot, frame total
    uint32_t *wpref;      // [num_waves]
    uint32_t *ftot;       // [1] events of the frame
    uint32_t *status;
    uint32_t n_units;
    uint32_t num_waves;
    uint32_t width, channels, rowlen, row_begin;
    uint32_t lean;        // 1: the batch runs the lean K1 (LeanRec records); 2: lean runs ({rho, base_val ..} records); 0: generic
    uint32_t abs_t;       // TimeMode::AbsoluteT (record decoding)
    uint32_t wire_rec;    // 0: `out` takes 12-byte AdderEvents; 9 / 11: the raw sink's records (bytes, back to back), out_cap in records
    StepConsts sc;        // running_t / cth are filled per frame from the table
};

struct ParkLayout {
    uint32_t group_shift;   // log2 of the segments per group (31: one group = frame-major)
    uint32_t group_stride;  // bytes between consecutive groups
    uint32_t frame_stride;  // bytes between consecutive frame slots of one segment
    uint32_t seg_stride;    // bytes between consecutive segments inside a group
    // Rotation of the frame slots (segment-major blocked layout only): segment s keeps frame slot fi at
    // (fi + (s >> rot_shift)) & rot_mask.  Without it every wave of the chip is, at any moment, writing (frame kernel)
    // or reading (expansion: all waves work on the same frame) addresses that agree in their low 17 bits -- which
    // memory channels that hits is then left to how the hardware hashes the upper bits of wherever the ring happened
    // to be mapped.  rot_shift >= 4 keeps the 16 segments of one expansion wave a constant stride apart.
    // Off: rot_shift = 31, rot_mask = 0xffffffff.
    uint32_t rot_shift, rot_mask;
};

constexpr uint32_t kScanTileWaves = ADDER_SCAN_TILE_WAVES;  // (a multiple of 4)
// What adder_lean1w_kernel takes as kernel arguments (by value): the level-0 planes and the band's size.
struct Lean1wArgs {
    uint32_t *hdr;
    float *integ0, *dt0, *bdt0, *lastf;
    uint32_t n_units, num_waves;
};

// Device-resident description of one batch of frames.  The kernels take (BatchArgs*, f), so
// a captured hipGraph of T frames can be replayed for ANY batch of T frames: the host only
// rewrites this struct (and the per-frame table) before launching the graph.
struct BatchArgs {
    FrameArgs base;           // per-frame fields (frame, frame_idx, park, wtot, wpref, ftot, sc.running_t, sc.cth) are derived
    const uint8_t *frames;    // packed [T][n_units]
    const FrameTab *ftab;     // [T]
    // compaction scratch: a ring of `slots` frames
    uint8_t *park_ring;       // [slots][num_waves][park_bytes]
    uint32_t park_bytes;      // scratch of one segment (kLeanParkBytes, or kGenRecBytes * parked-event capacity)
    ParkLayout park_layout;   // where (frame slot, segment) lies inside a chunk of the ring (park_offset)
    uint32_t *wtot_ring;      // [slots][num_waves]
    uint32_t *wpref_ring;     // [slots][num_waves]
    uint32_t *ftot_ring;      // [2][slots]: events per frame, then parked records per frame
    // Per-event records (the generic and the bounded Collapse frame kernels) are APPENDED to one log per segment and
    // chunk: region (chunk in ring, segment) holds log_cap 8-byte records at park_ring + ((cir * num_waves + seg) *
    // log_cap) * 8, a frame's run of records starts at wofs[slot][seg] and the cursor survives from one launch of a
    // chunk to the next in wcur.  log_cap is the hard bound of what a segment can emit over a chunk (log_capacity()),
    // so nothing can overflow; a fixed slot per frame would have to hold the per-FRAME bound (a whole arena per unit).
    uint32_t log_cap;         // records per region; 0: fixed slots (park_layout)
    uint32_t *wofs_ring;      // [slots][num_waves] first record of the segment's run, relative to its region
    uint32_t *wcur;           // [ring chunks][num_waves] records appended to the region so far
    uint32_t slots;
    uint32_t chunk;           // frames per chunk; slots = chunks in the ring * chunk
    const uint8_t *rr_tab;    // [256][kRrTabRows] chain lengths (adder_pixel.hpp rr_build_tab); run-record batches only
    const uint32_t *lr_tab;   // [kLrTabWords] events A and C by (base_val, rho) / input (lr_build_tab); lean-runs batches only
    uint64_t *rec_total;      // parked records of the batch so far (diagnostics; may be null)
    // Undo copy of the levels >= 1 (a batch whose event buffer is below its worst case keeps one: adder_hip_finish can
    // roll back).  Not null: the batch's FIRST launch (frame 0) stores every unit's LIVE levels 1 .. m-1 here before
    // anything overwrites them -- the generic / bounded Collapse kernels hold them in their hands anyway; a host-side
    // copy of the whole planes moved (max_depth - 1) * 13 bytes per unit and batch, 6 GB for a 4K RGB plane.
    float *snap_dv_integ, *snap_dv_dt, *snap_dv_bdt;
    uint8_t *snap_dv_bd;
    // diagnostics (null in normal operation): first start / last end of every kernel of the batch on the 100 MHz
    // constant clock, [kernel kind 0 frame, 1 scan, 2 offsets, 3 expansion][chunk][2]  (tools/timeline_probe.py)
    unsigned long long *timeline;
    // the integer-state kernels (lean runs, run records) report the longest run -- frames a root has accumulated -- their units
    // hold at the end of a launch: the host's bound on "rho * 255 and rho * time_spanned stay exact in binary32" (may be null)
    uint32_t *run_max;
};
constexpr uint32_t kRunReportMin = 16384;  // runs shorter than this are not reported (BatchResult::max_run 0 = "below it")
constexpr uint32_t kTimelineChunks = 64;
constexpr uint32_t kMaxBands = 16;  // bands one adder_expand_bands_kernel launch takes (more: one launch per band)

// Where (frame slot, segment) parks its records, in bytes from park_ring.  Within one chunk of the ring, segments
// come in groups of G = 2^group_shift; a group holds [frame][segment of the group][park_bytes]:
//   offset = chunk_in_ring * (num_waves * chunk * park_bytes) + (seg >> shift) * group_stride + frame * frame_stride
//            + (seg & (G - 1)) * seg_stride
//   G = 1          segment-major [segment][frame]: the frames a wave steps in ONE launch are contiguous;
//   G = 16         the 16 segments ONE expansion wave reads of a frame are contiguous (24 KiB), a frame-kernel wave's
//                  frames lie 24 KiB apart inside the group's 768 KiB;
//   G >= num_waves frame-major [frame][segment]: batches launched one frame at a time.  There a launch writes one
//                  short run of records per segment, and segment-major puts consecutive segments 48 KiB apart: 16 200
//                  scattered partial lines per 1080p frame cost the one-frame kernel 2.6 of its 14.7 us, and its
//                  expansion 10 of 64 us per chunk.
__host__ __device__ __forceinline__ size_t park_offset(uint32_t slot, uint32_t seg, uint32_t chunk, uint32_t num_waves,
                                                       uint32_t park_bytes, const ParkLayout &l) {
    const uint32_t cir = slot / chunk;
    const uint32_t fi = (slot - cir * chunk + (seg >> l.rot_shift)) & l.rot_mask;
    const uint32_t group = seg >> l.group_shift, r = seg - (group << l.group_shift);
    return (size_t)cir * num_waves * chunk * park_bytes + (size_t)group * l.group_stride + (size_t)fi * l.frame_stride +
           (size_t)r * l.seg_stride;
}

// Records a segment can emit over `chunk` consecutive frames, per unit.  A flush emits at most (frames since the
// previous flush + 1) events -- every frame adds at most one level to the arena, a Collapse flush of a popped arena
// emits 2 and needs >= 2 frames -- and max_depth at most; pop_top adds one per frame at most, and none in a frame that
// flushed when delta_t_max >= 2 * time.  Summed: 2 * chunk + max_depth + 2 (Collapse with delta_t_max > time), 3 *
// chunk + max_depth + 2 in general.
__host__ __device__ __forceinline__ uint32_t log_capacity(uint32_t chunk, uint32_t max_depth, bool pops_exclude_flushes) {
    return kWaveUnits * ((pops_exclude_flushes ? 2u : 3u) * chunk + max_depth + 2u);
}

// what the host reads after a batch (adder_publish_kernel), in page-locked host memory
struct BatchResult {
    uint64_t total_events;  // frame_offsets[num_frames]
    uint64_t records;       // parked records (diagnostics)
    uint32_t status;
    uint32_t valid;         // set last
    uint32_t max_run;       // the longest run any unit held at the end of a launch of the batch (BatchArgs::run_max), 0 = not reported
    uint32_t pad;
};

// result header of one frame handed to the host (adder_frame_out_kernel), in page-locked host memory
struct FrameResult {
    uint64_t produced;  // events the frame produced (may exceed the slot's capacity: then status has kStatusCapacity)
    uint32_t status;
    uint32_t new_features;  // feature-driven rate control: features this frame found new
};

// One integrate_for_px call of an event-camera source (same layout as AdderSparseStep, include/adder_hip.h)
struct SparseStep {
    uint16_t x, y;
    uint8_t c, frame_val;
    uint16_t pad;
    float intensity, time;
};
// what the sparse kernels need of a Mode::Continuous context (adder_sparse.hip)
struct SparseArgs {
    uint32_t *hdr;
    float *lastf;
    float *cn_integ, *cn_dt, *cn_bdt;
    uint32_t *cn_meta;
    size_t plane_stride;
    uint8_t *cth_px, *cctr_px;  // c_thresh / c_increase_counter per unit
    float *rt_px;               // PixelArena::running_t per unit
    uint8_t *running;           // side plane or nullptr
    uint32_t *status;
    uint32_t c_max, c_vel, max_nodes, stage_events;
    uint32_t width, channels, row_begin, rows;
    StepConsts sc;              // (time_spanned, running_t, cth are set per step)
};

// handle_features / handle_roi of one context (video.rs:865-1112)
struct FeatureArgs {
    uint8_t *fset;        // [rows][width] membership of VideoState::features (0 / 1)
    uint32_t *counters;   // [0] features found new since the batch began
    uint32_t chunk_rows;  // the event windows are circular per row chunk
    uint32_t detect;      // VideoState::feature_detection
    uint32_t radius;      // feature_c_radius if feature_rate_adjustment, else 0
    uint32_t low;         // min(c_thresh_baseline, 2)
    uint32_t roi_on, rx0, ry0, rx1, ry1;  // Roi {start, end}, inclusive, plane coordinates
    // row-band contexts (multi-GPU): the running-intensities plane carries kFeatureHalo rows of the neighbouring bands
    // above and below the context's own rows (adder_hip_feature_halo_import), the corner test runs in PLANE coordinates,
    // and the new features are also listed (x | plane y << 16) so that the neighbours can reset THEIR rows around them
    uint32_t plane_h;     // PlaneSize height
    uint32_t *new_xy;     // list of this frame's new features, or nullptr
    uint32_t new_cap;     // its capacity; counters[1] counts the entries
};
constexpr uint32_t kFeatureHalo = 3;  // rows the FAST 9_16 ring reaches over (adder_pixel.hpp kFastBorder)

__device__ __forceinline__ FrameArgs frame_args(const BatchArgs *b, uint32_t f) {
    FrameArgs a = b->base;
    const uint32_t slot = f % b->slots;
    a.frame = b->frames + (size_t)f * a.n_units;
    a.frame_idx = f;
    a.sc.running_t = b->ftab[f].running_t;
    a.sc.running_t_u32 = f32_as_u32(a.sc.running_t);
    a.sc.cth = b->ftab[f].cth;
    a.park = b->park_ring + park_offset(slot, 0u, b->chunk, a.num_waves, b->park_bytes, b->park_layout);  // segment 0
    a.wtot = b->wtot_ring + (size_t)slot * a.num_waves;
    a.wpref = b->wpref_ring + (size_t)slot * a.num_waves;
    a.ftot = b->ftot_ring + slot;
    return a;
}

}  // namespace adder

extern "C" {
// variant = collapse | abs_t << 1 | generic << 2 | continuous << 3 (host copy of what BatchArgs holds) | the band
// has >= 4 units << 4 | bounded Collapse step << 5 (with generic: the per-event record format)
// K1: frames [f, f + nb) in one launch (nb > 1 = temporal blocking).  grid_cap (0 = none) bounds the number of
// workgroups of the lean kernel / of the expansion: the workgroups then walk their work items, which leaves room
// for the other kernel to be resident on the same CUs
hipError_t adder_launch_frame(const adder::BatchArgs *b, uint32_t f, uint32_t nb, uint32_t variant,
                              uint32_t num_waves, uint32_t grid_cap, hipStream_t stream,
                              const adder::Lean1wArgs *wide);  // (host copy of the level-0 planes, or null)
// the lean-runs step in packed bytes (adder_lp_kernels.hip; variant bit 4096): a wave per PAIR of segments
hipError_t adder_launch_lp(const adder::BatchArgs *b, uint32_t f, uint32_t nb, uint32_t lazy, uint32_t num_waves,
                           uint32_t grid_cap, hipStream_t stream);
// ... and its expansion (adder_lpx_kernel): rec = 9 / 11 (the raw sink's records) or 12 (AdderEvents)
// (host_b: the host's copy of *b; frames [f0, f0 + nf) lie in ONE chunk of the scratch ring)
hipError_t adder_launch_lpx(const adder::BatchArgs *b, const adder::BatchArgs *host_b, uint32_t f0, uint32_t nf, uint32_t rec,
                            hipStream_t stream);
hipError_t adder_launch_divtest(unsigned long long *d_bad, hipStream_t stream);
hipError_t adder_launch_wire(const adder::AdderEventPod *ev, uint64_t n, uint32_t rec, uint8_t *out, uint32_t *status,
                             hipStream_t stream);
// frames [f0, f0 + nf): per-frame scan, frame_offsets chain, expansion of the parked records
hipError_t adder_launch_scan(const adder::BatchArgs *b, uint32_t f0, uint32_t nf, uint32_t num_waves, hipStream_t stream, uint32_t whole_batch = 0u,
                             uint32_t rec_prefix = 0u, uint32_t chain = 0u);  // whole_batch: a batch of one frame -- the scan writes the frame
                                                                              // offsets too; chain: ... of every batch (adder_scan_kernel)
size_t adder_sparse_temp_bytes(uint32_t n);
hipError_t adder_sparse_run(const adder::SparseArgs *args, const adder::SparseStep *d_steps, uint32_t n, uint32_t *keys0,
                            uint32_t *keys1, uint32_t *idx0, uint32_t *idx1, void *d_temp, size_t temp_bytes, uint2 *stage,
                            uint32_t *count, uint32_t *offs, adder::AdderEventPod *d_out, uint64_t out_cap,
                            unsigned long long *d_total, hipStream_t stream);
hipError_t adder_launch_publish(const adder::BatchArgs *b, uint32_t num_frames, adder::BatchResult *h, hipStream_t stream);
hipError_t adder_launch_offsets(const adder::BatchArgs *b, uint32_t f0, uint32_t nf, hipStream_t stream);
// (host_b: the host's copy of *b -- the packed lean-runs expansion takes its context-constant part as kernel arguments)
hipError_t adder_launch_expand(const adder::BatchArgs *b, uint32_t f0, uint32_t nf, uint32_t num_waves,
                               uint32_t variant, uint32_t grid_cap, hipStream_t stream, const adder::BatchArgs *host_b = nullptr);
hipError_t adder_launch_fill_u32(uint32_t *p, size_t n, uint32_t v, hipStream_t stream);
// records over the wire (adder_kernels.hip: adder_log_pack_kernel, adder_band_layout_kernel)
hipError_t adder_launch_log_pack(const uint8_t *logs, uint32_t log_cap, uint32_t rec_bytes, const uint32_t *wcur,
                                 uint32_t *pbase, uint32_t num_waves, uint32_t nf, uint32_t *wofs_rows, uint8_t *packed,
                                 uint64_t packed_cap_bytes, uint64_t *d_total, uint32_t *status, hipStream_t stream);
hipError_t adder_launch_expand_bands(const uint8_t *descs, uint32_t stride, uint32_t n_bands, const uint32_t *num_waves,
                                     uint32_t nf, uint32_t abs_t, hipStream_t stream, uint32_t runs = 0u, uint32_t wire = 0u);
hipError_t adder_launch_sink_layout(const uint64_t *all_offs, uint32_t world, uint32_t rank, uint32_t nf, uint64_t *file_pos,
                                    uint64_t *dest, uint64_t *merged_offs, hipStream_t stream);
hipError_t adder_launch_wire_scatter(const adder::AdderEventPod *ev, const uint64_t *offs, uint32_t nf, const uint64_t *dest,
                                     uint32_t rec, uint8_t *out, uint64_t out_cap, uint64_t header, uint32_t *status,
                                     uint32_t grid, hipStream_t stream, uint64_t src_cap_events = ~0ull);  // src_cap_events: what `ev` holds
hipError_t adder_launch_slot_pack(const adder::BatchArgs *b, uint32_t nf, uint32_t num_waves, uint32_t rec_bytes, uint8_t *packed,
                                  uint64_t packed_cap_bytes, uint32_t *status, hipStream_t stream);
hipError_t adder_launch_band_layout(const uint64_t *const *offs, uint32_t n_bands, uint32_t nf, uint64_t merged_base,
                                    uint64_t *merged_offsets, uint64_t *dest, hipStream_t stream);
hipError_t adder_launch_chunk_offsets(const adder::AdderEventPod *ev, uint32_t n, uint32_t row_begin,
                                      uint32_t chunk_rows, uint32_t num_chunks, uint32_t *offsets,
                                      hipStream_t stream);
// merge of `world` frame-major streams laid back to back in `stage` (offs = [world][T+1]); work = (T+1) + world*T uint64
hipError_t adder_launch_merge(const adder::AdderEventPod *stage, const uint64_t *offs, uint32_t world, uint32_t T,
                              uint64_t *work, adder::AdderEventPod *out, uint64_t out_cap, uint64_t *merged_offsets,
                              uint64_t merged_base, uint32_t *status, hipStream_t stream);
hipError_t adder_launch_frame_out(const adder::AdderEventPod *d_ev, const uint64_t *d_offsets, uint64_t cap,
                                  adder::AdderEventPod *h_ev, adder::FrameResult *h_res, uint32_t *h_chunks,
                                  const uint32_t *status, const uint32_t *counters, uint32_t row_begin, uint32_t chunk_rows,
                                  uint32_t num_chunks, hipStream_t stream, uint32_t wire_rec = 0u);
// after frame f's events are in place: FAST features at the events' pixels -> membership plane, c_thresh reset
// around the new ones, ROI (video.rs:865-1112)
hipError_t adder_launch_features(const adder::BatchArgs *b, uint32_t f, const adder::FeatureArgs *fa,
                                 hipStream_t stream);
// the c_thresh resets around features found by OTHER row bands (x | plane y << 16), clipped to this band's rows
hipError_t adder_launch_feature_apply(const adder::BatchArgs *b, const adder::FeatureArgs *fa, const uint32_t *xy, uint32_t n,
                                      hipStream_t stream);
hipError_t adder_launch_synth(uint8_t *dst, int content, uint64_t seed, uint32_t W, uint32_t H, uint32_t C,
                              uint32_t y0, uint32_t rows, uint32_t k0, uint32_t nframes, hipStream_t stream);
}
