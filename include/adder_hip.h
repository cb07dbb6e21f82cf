/*
 * adder_hip.h -- C-ABI of the MI355X-native framed->ADDER integration path.
 *
 * This is the drop-in boundary for ONE region of the reference
 * (ac-freeman/adder-codec-rs): the rayon per-pixel loop of
 * Video::integrate_matrix, adder-codec-rs/src/transcoder/source/video.rs:677-734
 * (per pixel: integrate_for_px, video.rs:1318-1380, which drives
 * PixelArena::{integrate,pop_best_events,pop_top_event},
 * adder-codec-rs/src/transcoder/event_pixel_tree.rs:139-413).  Everything a
 * Rust `extern "C"` shim inside Video<W> would need is here: plain pointers and
 * sizes, no C++ or torch types.  See INTEGRATION.md for the Rust-side binding.
 *
 * Threading: a context is NOT thread-safe (mirrors `consume(&mut self)`); use one
 * context per device / row band; different contexts may be driven concurrently.
 * No C++ exception or abort crosses this boundary: every entry point returns an
 * int status (0 = ok, negative = AdderStatus) and adder_hip_last_error() gives
 * the message.
 */
#ifndef ADDER_HIP_H
#define ADDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADDER_HIP_ABI_VERSION 1

/* adder-codec-core/src/lib.rs:72-83 (TimeMode), :196-213 (Mode, PixelMultiMode) */
enum { ADDER_TIME_DELTA_T = 0, ADDER_TIME_ABSOLUTE_T = 1, ADDER_TIME_MIXED = 2 };
enum { ADDER_MULTI_NORMAL = 0, ADDER_MULTI_COLLAPSE = 1 };
/* Mode (lib.rs:196-205): framed sources always use FramePerfect (framed.rs:67); Continuous is the mode of the
 * event-camera sources (prophesee.rs:65, davis.rs:116-117) and of Video::integrate_matrix when they feed it frames */
enum { ADDER_MODE_FRAME_PERFECT = 0, ADDER_MODE_CONTINUOUS = 1 };

/* adder-codec-core/src/lib.rs:181-193 */
#define ADDER_D_MAX 127
#define ADDER_D_ZERO_INTEGRATION 128
#define ADDER_D_EMPTY 255
#define ADDER_C_NONE 0xFF /* Coord::c == None (single-channel plane) */

typedef enum AdderStatus {
    ADDER_OK = 0,
    ADDER_E_BAD_PARAMS = -1,   /* SourceError::BadParams analogue (video.rs:55-122) */
    ADDER_E_HIP = -2,          /* a HIP runtime call failed */
    ADDER_E_NO_DEVICE = -3,    /* no usable gfx950 device: there is NO CPU fallback */
    ADDER_E_OUT_CAPACITY = -4, /* event buffer too small; *n_out holds the required count.
                                  Pixel state HAS advanced: the context is poisoned. */
    ADDER_E_ARENA_DEPTH = -5,  /* a pixel needed more stored nodes than max_depth; poisoned */
    ADDER_E_TIMEOUT = -6,      /* reserved (no kernel of this library waits on another workgroup) */
    ADDER_E_POISONED = -7      /* a previous call failed after state had advanced */
} AdderStatus;

/* One ADDER event in host byte order: lib.rs:371-377 `Event{coord{x,y,c},d,t}`.
 * Serialisation to the 9/11-byte big-endian wire form (raw/stream.rs:101-120)
 * is the sink's job (adder_raw_* below), not the kernel's. */
typedef struct AdderEvent {
    uint16_t x;
    uint16_t y;
    uint8_t c; /* ADDER_C_NONE for 1-channel planes */
    uint8_t d;
    uint16_t pad; /* always 0 */
    uint32_t t;
} AdderEvent;

/* What Video::new (video.rs:350-438) + time_parameters (:493-537) + write_out
 * (:546-636) + chunk_rows (:471-479) + update_crf/update_quality_manual
 * (:1241-1287) establish before the first consume(). */
typedef struct AdderHipParams {
    uint32_t abi_version;  /* = ADDER_HIP_ABI_VERSION */
    uint16_t width;        /* full plane width  (PlaneSize, lib.rs:86-118) */
    uint16_t height;       /* full plane height */
    uint8_t channels;      /* 1 (c = None) or 3 */
    uint8_t time_mode;     /* ADDER_TIME_* ; pixels default to AbsoluteT (event_pixel_tree.rs:76) */
    uint8_t multi_mode;    /* ADDER_MULTI_* ; write_out defaults to Collapse (video.rs:598) */
    uint8_t pixel_mode;    /* ADDER_MODE_FRAME_PERFECT or ADDER_MODE_CONTINUOUS (a first, untuned kernel) */
    uint32_t row_begin;    /* this context integrates rows [row_begin,row_end) of the plane */
    uint32_t row_end;      /*   (0,0 => the whole plane); events carry absolute y          */
    uint32_t ref_time;     /* ticks per input frame (VideoStateParams::ref_time)            */
    uint32_t delta_t_max;  /* >= ref_time (video.rs:527-533)                               */
    uint8_t c_thresh_max;         /* CrfParameters (rate_controller.rs:40-53); EncoderOptions */
    uint8_t c_increase_velocity;  /*   ::default => quality 3 => (7, 7); must be >= 1         */
    uint8_t c_thresh_start;       /* per-pixel c_thresh at construction: 10 (event_pixel_tree.rs:82) */
    uint8_t c_counter_start;      /* per-pixel c_increase_counter at construction: 1 (:83)      */
    uint32_t chunk_rows;   /* rows per output chunk (video.rs:230); 0 => 1                  */
    uint32_t max_depth;    /* stored arena nodes per pixel, 1..31; 0 => 16                  */
    int32_t device_id;     /* HIP device ordinal; -1 => current device                     */
} AdderHipParams;

typedef struct AdderHipCtx AdderHipCtx;

/* Fills *p with the reference's construction defaults for a w x h x c plane
 * (Video::new + VideoStateParams::default: Collapse, AbsoluteT, ref 255, dtm 7650,
 * c_thresh 10 / counter 1, crf parameters of quality 3). */
void adder_hip_default_params(AdderHipParams *p, uint16_t width, uint16_t height, uint8_t channels);

int adder_hip_create(const AdderHipParams *params, AdderHipCtx **out);
void adder_hip_destroy(AdderHipCtx *ctx);
const char *adder_hip_last_error(const AdderHipCtx *ctx); /* ctx may be NULL: last create() error */

/* Encoder-side Crf replacement done by write_out (video.rs:629): c_thresh_max and
 * c_increase_velocity change, per-pixel c_thresh does NOT. */
int adder_hip_set_crf_parameters(AdderHipCtx *ctx, uint8_t c_thresh_max, uint8_t c_increase_velocity);
/* Per-pixel reset performed by update_crf / update_quality_manual (video.rs:1247-1250,
 * :1283-1286): every pixel's c_thresh = baseline, c_increase_counter = 0. */
int adder_hip_reset_c_thresh(AdderHipCtx *ctx, uint8_t c_thresh_baseline);
/* ---- Feature-driven rate control and ROI.  The only cross-pixel coupling of the transcoder: after
 * each frame's events exist, the FAST 9_16 corner test (utils/cv.rs:56-212) runs on the running intensities at the
 * pixels that fired (video.rs:883-1085), VideoState::features is updated, and -- with feature_rate_adjustment
 * and a radius > 0 -- every pixel within feature_c_radius of a NEW feature gets c_thresh = min(c_thresh_baseline,
 * 2) (:1089-1105); handle_roi (:866-882) does the same for the region of interest after every frame.  From then
 * on c_thresh differs between pixels, so the context switches (until adder_hip_reset / adder_hip_reset_c_thresh) to
 * the generic kernels with one (c_thresh, c_increase_counter) pair per pixel, and steps frame by frame.  A context
 * that owns the whole plane does all of it inside the integrate call; a ROW BAND (multi-GPU) follows the protocol of
 * adder_hip_feature_detect below.
 * Video::update_detect_features (video.rs:825-837; show_features / feature_cluster only drive displays): */
int adder_hip_update_detect_features(AdderHipCtx *ctx, int detect_features, int feature_rate_adjustment);
/* CrfParameters::{c_thresh_baseline, feature_c_radius} (rate_controller.rs:40-53; update_quality_manual
 * video.rs:1264-1279).  Defaults: Crf::new(None) = quality 3: baseline 2, radius min(width, height) / 15. */
int adder_hip_set_feature_parameters(AdderHipCtx *ctx, uint8_t c_thresh_baseline, uint16_t feature_c_radius);
/* Video::update_roi (video.rs:1291-1293); Roi {start, end} inclusive, plane coordinates. */
int adder_hip_update_roi(AdderHipCtx *ctx, int enable, uint16_t start_x, uint16_t start_y, uint16_t end_x,
                         uint16_t end_y);
/* ---- Feature mode across row bands ("needs a halo exchange between row bands").  The corner test at a
 * pixel reads 3 rows above and below it, and the reset square around a NEW feature (radius feature_c_radius) reaches
 * into the neighbouring bands.  A band context in feature / ROI mode therefore integrates ONE frame per call and leaves
 * the feature step of video.rs:744-778 to the caller's per-frame exchange:
 *   1. adder_hip_integrate / _integrate_device (one frame) on every band: the frame's events; running intensities updated;
 *   2. adder_hip_feature_halo_export on every band -> its first / last 3 rows (adder_hip_feature_halo_bytes each, device
 *      memory); ship them to the band above / below (peer copy, RCCL send/recv) and hand them to
 *      adder_hip_feature_halo_import (d_above = the upper neighbour's bottom rows, d_below = the lower neighbour's top);
 *   3. adder_hip_feature_detect on every band: the corner tests at its own events' pixels in plane coordinates, its
 *      part of VideoState::features, the resets inside its own rows, the ROI; the frame's new features come back as a
 *      device list of x | plane_y << 16 (n_new of them);
 *   4. adder_hip_feature_apply on every band with the OTHER bands' lists: their reset squares cut to this band's rows.
 * The bands together then hold exactly the state of one whole-plane context (tests: two and three bands on one device
 * against the whole plane and the oracle).  Every reset writes the same value, so the order of 3 and 4 across bands is free. */
size_t adder_hip_feature_halo_bytes(const AdderHipCtx *ctx);
int adder_hip_feature_halo_export(AdderHipCtx *ctx, uint8_t *d_top_rows, uint8_t *d_bottom_rows, void *stream);
int adder_hip_feature_halo_import(AdderHipCtx *ctx, const uint8_t *d_above, const uint8_t *d_below, void *stream);
int adder_hip_feature_detect(AdderHipCtx *ctx, uint32_t *d_new_xy, uint32_t cap, uint32_t *n_new, void *stream);
int adder_hip_feature_apply(AdderHipCtx *ctx, const uint32_t *d_xy, uint32_t n, void *stream);
/* VideoState::features as a membership plane: dst = [rows][width] bytes, 1 = the pixel is a feature. */
int adder_hip_feature_set(AdderHipCtx *ctx, uint8_t *dst);
/* every pixel's c_thresh as the NEXT frame will test it: dst = [rows][width][channels] bytes */
int adder_hip_c_thresh_plane(AdderHipCtx *ctx, uint8_t *dst);
/* features the last finished batch found new (diagnostics) */
uint32_t adder_hip_last_new_features(const AdderHipCtx *ctx);
/* update_quality_manual's delta_t_max = multiplier * ref_time (video.rs:1280). */
int adder_hip_set_delta_t_max(AdderHipCtx *ctx, uint32_t delta_t_max);
/* time_parameters / write_out `px.time_mode(..)` (video.rs:499-503,632-634); only
 * legal before the first frame has been integrated. */
int adder_hip_set_time_mode(AdderHipCtx *ctx, uint8_t time_mode);

/* Number of row chunks of this context's band: ceil(rows / chunk_rows) (video.rs:473-477). */
uint32_t adder_hip_num_chunks(const AdderHipCtx *ctx);
/* Upper bound on events one frame can emit for this context (units * (max_depth + 2)). */
size_t adder_hip_max_events_per_frame(const AdderHipCtx *ctx);

/* Page-locked host memory for frame / event buffers handed to the host-buffer entry points
 * below: with pinned buffers the PCIe copies run at link speed and asynchronously; pageable
 * buffers work too but are staged by the runtime.  NULL on failure. */
void *adder_hip_alloc_pinned(size_t bytes);
void adder_hip_free_pinned(void *p);

/* Event-buffer capacity.  Every integrate entry point takes the capacity of the caller's event buffer.  If a
 * batch emits more, the call (or adder_hip_finish for the device-pointer form) returns ADDER_E_OUT_CAPACITY with
 * the number of events the batch needs in *n_out, and the pixel state is ROLLED BACK to what it was before the
 * call: retry with a buffer of that size.  (A buffer of adder_hip_max_events_per_frame() x frames can never
 * overflow and costs no undo copy.) */

/* --- one frame: the region video.rs:677-734 -------------------------------------
 * frame_hwc: host pointer to this band's rows, [rows][width][channels] u8 with
 * row_stride_bytes between rows.  Events come back in the reference's order
 * (y, x, c, per-pixel emission order); chunk_offsets (may be NULL) receives
 * num_chunks+1 prefix offsets so the caller can rebuild Vec<Vec<Event>>. */
int adder_hip_integrate(AdderHipCtx *ctx, const uint8_t *frame_hwc, size_t row_stride_bytes,
                        float time_spanned, AdderEvent *out, size_t out_cap, size_t *n_out,
                        uint32_t *chunk_offsets);

/* With out_cap >= adder_hip_max_events_per_frame() the call cannot overflow and takes the short route: upload,
 * integrate, and a device kernel that stores the events straight into `out` when `out` is page-locked
 * (adder_hip_alloc_pinned, or hipHostRegister'ed by the caller), else into a page-locked slot that is then copied. */

/* --- one frame, without the blocking round trip: the `consume` loop of a live source (framed.rs:127-157) ------
 * adder_hip_frame_submit queues frame k (upload, integration, hand-over of events + row-chunk offsets to a slot
 * of page-locked host memory) and returns; adder_hip_frame_collect waits for the OLDEST frame in flight and
 * returns pointers into its slot, valid until `slots` more frames have been submitted.  Frame k's transfer
 * overlaps frame k+1's integration.  `frame_hwc` should be page-locked (adder_hip_alloc_pinned) -- a pageable
 * frame makes the upload synchronous.  At most `slots` frames are in flight (default 3); a slot holds the mode's
 * worst case of events, at most 2 GiB (adder_hip_frames_configure(ctx, slots, events_per_slot), 0 = default).
 * There is no rollback on this route: a frame that overflows its slot fails with ADDER_E_OUT_CAPACITY in collect
 * and poisons the context.  Other entry points refuse to run while frames are in flight. */
int adder_hip_frames_configure(AdderHipCtx *ctx, uint32_t slots, size_t events_per_slot);
int adder_hip_frame_submit(AdderHipCtx *ctx, const uint8_t *frame_hwc, size_t row_stride_bytes, float time_spanned);
int adder_hip_frame_collect(AdderHipCtx *ctx, const AdderEvent **events, size_t *n_events,
                            const uint32_t **chunk_offsets);
uint32_t adder_hip_frames_in_flight(const AdderHipCtx *ctx);
/* The ring can hand out what the raw sink WRITES instead of AdderEvents (RawOutput::ingest_event, raw/stream.rs:101-120:
 * 9-byte EventSingle records on a 1-channel plane, 11-byte Event records otherwise, bincode fixint big-endian): the
 * hand-over kernel serialises on the device and stores the bytes into the slot -- 25 % fewer bytes over PCIe, and the
 * caller's sink is a write().  adder_hip_frames_set_format(ctx, 1) while no frame is in flight; collect with
 * adder_hip_frame_collect_wire (n_bytes = n_events x record size; chunk_offsets stay in events). */
int adder_hip_frames_set_format(AdderHipCtx *ctx, int wire_records);
int adder_hip_frame_collect_wire(AdderHipCtx *ctx, const uint8_t **bytes, size_t *n_bytes, size_t *n_events,
                                 const uint32_t **chunk_offsets);

/* --- sparse sources (event cameras; Mode::Continuous contexts) ------------------------------------------------
 * One step = one `integrate_for_px(px, &mut 0, frame_val, intensity, time, &mut events, ..)` call of the reference's
 * event-camera sources (prophesee.rs:196-254, 343-358; davis.rs), i.e. what a camera event makes of its pixel: the
 * host keeps the camera-side state (last timestamp, log intensity) and hands over the steps in the camera's order.
 * All events go to ONE buffer in step order (the sources return `vec![events]`).  A pixel's c_thresh, its counter
 * and running_t advance per call, so after the first sparse call they differ between pixels: dense frames are then
 * refused until adder_hip_reset (the dense start-up frames of Prophesee::consume, :117-131, come first).  A buffer
 * of n * (max_depth + 3) events cannot overflow; an overflow poisons the context (no rollback on this route). */
/* The running-intensities side plane is sampled once per CAMERA event, after the last of its (one or two)
 * integrate_for_px calls (prophesee.rs:259-283): flag the first step of a two-step camera event, and the steps of
 * end_events (:330-372, which never sample it). */
#define ADDER_SPARSE_NO_SIDE 1u
/* The DAVIS source (davis.rs:331-395) splits the step: it integrates the pixel's OLD intensity over the time since its
 * last event without a contrast test (ADDER_SPARSE_INTEGRATE_ONLY: pop_top if flagged, integrate, pop_top if flagged),
 * then tests the NEW value against base_val +- c_thresh and, outside, flushes and restarts the arena for it WITHOUT
 * integrating it (ADDER_SPARSE_TEST_ONLY: pop_best_events(intensity), base_val = frame_val, set_d_for_continuous).
 * At the end of its input it pops every pixel's events (ADDER_SPARSE_FLUSH: pop_best_events(intensity) alone, :654-661). */
#define ADDER_SPARSE_INTEGRATE_ONLY 2u
#define ADDER_SPARSE_TEST_ONLY 4u
#define ADDER_SPARSE_FLUSH 8u
typedef struct AdderSparseStep {
    uint16_t x, y;      /* plane coordinates */
    uint8_t c;          /* channel, ADDER_C_NONE on a 1-channel plane */
    uint8_t frame_val;  /* compared with the pixel's base_val +- c_thresh (video.rs:1336-1340; the sources' `&mut 0` is an
                         * out parameter that integrate_for_px overwrites with px.base_val) */
    uint16_t pad;       /* flags: ADDER_SPARSE_NO_SIDE, else 0 */
    float intensity;    /* intensity to integrate */
    float time;         /* over this many ticks */
} AdderSparseStep;
int adder_hip_integrate_sparse(AdderHipCtx *ctx, const AdderSparseStep *steps, size_t n, AdderEvent *out, size_t out_cap,
                               size_t *n_out);
int adder_hip_integrate_sparse_device(AdderHipCtx *ctx, const AdderSparseStep *d_steps, size_t n, AdderEvent *d_out,
                                      size_t out_cap, size_t *n_out, void *stream);

/* --- T frames, host buffers: same stream, frame-major; frame_offsets gets T+1 entries. */
int adder_hip_integrate_batch(AdderHipCtx *ctx, const uint8_t *frames_hwc, uint32_t num_frames,
                              size_t frame_stride_bytes, size_t row_stride_bytes,
                              float time_spanned, AdderEvent *out, size_t out_cap, size_t *n_out,
                              uint64_t *frame_offsets);

/* --- T frames resident in HBM (clip and event buffer are DEVICE pointers) -----------
 * d_frames: packed [T][rows][width][channels] u8.  d_out: device AdderEvent[out_cap].
 * d_frame_offsets: device uint64[T+1] (prefix offsets into d_out; [0] = 0).
 * stream: hipStream_t.  NULL = the context's own (non-blocking) stream, ordered BEHIND whatever the caller has queued on the
 * legacy default stream so far (a NULL stream means "the default stream" to the caller: buffers it prepared there -- zeroed,
 * generated, copied -- are seen as prepared).  Asynchronous: call adder_hip_finish() (or synchronise the stream and call
 * it) to collect status; work queued on another stream AFTER the call is not ordered against the batch. */
int adder_hip_integrate_device(AdderHipCtx *ctx, const uint8_t *d_frames, uint32_t num_frames,
                               float time_spanned, AdderEvent *d_out, size_t out_cap,
                               uint64_t *d_frame_offsets, void *stream);

/* The same batch with the raw sink's RECORDS as its output: the expansion serialises every event as
 * RawOutput::ingest_event does (raw/stream.rs:101-120: bincode fixint big-endian; 9 bytes {x u16, y u16, d u8, t u32} on
 * a 1-channel plane, 11 bytes {x, y, 0x01, c, d, t} otherwise) and stores the records back to back -- the bytes between a
 * raw .adder file's header and its EOF event, in HBM, without a second pass (and 25 % fewer bytes than AdderEvents).
 * d_frame_offsets counts EVENTS as for adder_hip_integrate_device (frame f's records start at byte
 * d_frame_offsets[f] * record_bytes); wire_cap_bytes / record_bytes is the capacity in events, with the same overflow
 * semantics (adder_hip_finish reports the size needed and rolls back).  Dense FramePerfect batches without feature mode. */
int adder_hip_integrate_wire_device(AdderHipCtx *ctx, const uint8_t *d_frames, uint32_t num_frames, float time_spanned,
                                    uint8_t *d_wire, size_t wire_cap_bytes, uint64_t *d_frame_offsets, void *stream);
/* Waits for the work queued by adder_hip_integrate_device and reports its status;
 * *n_out (may be NULL) = total events of the last device batch. */
int adder_hip_finish(AdderHipCtx *ctx, size_t *n_out);

/* Per-chunk offsets of one frame's events on the device (binary search on y);
 * d_chunk_offsets: device uint32[num_chunks+1], relative to d_events. */
int adder_hip_chunk_offsets_device(AdderHipCtx *ctx, const AdderEvent *d_events, size_t n_events,
                                   uint32_t *d_chunk_offsets, void *stream);

/* Optional side plane Video::running_intensities (video.rs:713-730): copies the
 * band's [rows][width][channels] u8 plane to a host buffer. */
int adder_hip_running_intensities(AdderHipCtx *ctx, uint8_t *dst_host);
int adder_hip_enable_running_intensities(AdderHipCtx *ctx, int enable);

/* Duration in milliseconds of the kernels of the last adder_hip_integrate_device
 * batch, measured with HIP events on the launch stream (0 if none). */
float adder_hip_last_batch_ms(AdderHipCtx *ctx);
/* Device batches of more than one chunk are replayed from a captured graph whose two branches (the frame kernel of
 * chunk k+1 beside scan / offsets / expansion of chunk k) the runtime binds to hardware queues when the graph is
 * instantiated -- well or badly, for the life of the instance.  The first batches of a given length therefore try
 * a few instances (2 batches each, 6 instances -- the first on one stream, measured twice more at the end) and keep the fastest; this returns 1 once the last batch's length
 * has settled (always 1 for single-chunk batches and with ADDER_HIP_NO_GRAPH). */
int adder_hip_launch_plan_settled(const AdderHipCtx *ctx);
/* Which frame kernel the batch queued last ran (diagnostics; the parity tests assert the kernel they mean to test). */
#define ADDER_KERNEL_LEAN 0u          /* adder_lean_kernel / adder_lean1*_kernel */
#define ADDER_KERNEL_GENERIC 1u       /* adder_frame_kernel */
#define ADDER_KERNEL_CONTINUOUS 2u    /* adder_cont_kernel */
#define ADDER_KERNEL_BOUNDED 3u       /* adder_cb_kernel: the bounded Collapse step with stepped levels */
#define ADDER_KERNEL_CONSTANT_RUNS 4u /* adder_cr_kernel */
#define ADDER_KERNEL_RUN_RECORDS 5u   /* adder_rr_kernel */
#define ADDER_KERNEL_LEAN_RUNS 6u     /* adder_lr_kernel */
#define ADDER_KERNEL_LEAN_RUNS_PACKED 7u /* adder_lp_kernel: the same step on four units per lane (DeltaT) */
unsigned adder_hip_last_batch_kernel(const AdderHipCtx *ctx);
/* Diagnostics (environment ADDER_HIP_TIMELINE=1): first start / last end of the kernels of the last batch, in 10 ns
 * ticks of the device's constant clock: dst[4 kinds: frame, scan, offsets, expansion][64 chunks][2]. */
int adder_hip_debug_timeline(AdderHipCtx *ctx, unsigned long long *dst);

/* Mean duration in microseconds of the frame-kernel launches of the last device batch,
 * measured with one HIP event pair around EVERY launch on the launch stream; only
 * collected while adder_hip_set_launch_timing(ctx, 1) is in effect (the extra event
 * packets slow the batch down, so throughput runs leave it off).  enable = 2: one pair around
 * each chunk's RUN of frame-kernel launches instead (no event packets between the launches;
 * the mean then holds the gaps between consecutive kernels, not the events' own cost). */
int adder_hip_set_launch_timing(AdderHipCtx *ctx, int enable);
float adder_hip_last_launch_avg_us(AdderHipCtx *ctx);
/* Same run: mean duration of the scan + offsets + expansion launches of one chunk of frames (the other
 * half of the frame loop), the number of chunks timed, and the frames one chunk holds. */
float adder_hip_last_post_avg_us(AdderHipCtx *ctx);
uint32_t adder_hip_last_post_chunks(AdderHipCtx *ctx);
uint32_t adder_hip_chunk_frames(const AdderHipCtx *ctx);
/* Units (pixel-channels) of one segment -- the granularity of the records' tables; a band's worst case of one chunk is one
 * record per unit and frame: adder_hip_band_segments() * adder_hip_segment_units() * frames records. */
uint32_t adder_hip_segment_units(void);
/* Bytes of one raw-sink record of this plane: 9 (one channel: EventSingle) or 11 (raw/stream.rs:101-120). */
uint32_t adder_hip_wire_record_bytes(const AdderHipCtx *ctx);
/* Parked records of the last device batch (diagnostics: the bytes the frame kernel really moved). */
uint64_t adder_hip_last_batch_records(AdderHipCtx *ctx);
/* ---- records over the wire (multi-GPU gather of SURVEY 8(e); protocol in include/adder_gather.h / INTEGRATION.md) ----
 * The ordered gather to one device moves every event over ONE xGMI link per peer.  The frame kernel's parked records
 * carry the same information in 0.35x the bytes (8 bytes per unit with events instead of 12 per event), so a band can
 * ship those and let root expand them: adder_hip_integrate_records_device runs a batch of at most
 * adder_hip_chunk_frames() frames up to its scan (no expansion: nothing is written to an event buffer) and describes
 * what root needs; adder_hip_expand_records_device, on root, expands the bands' records -- its own included -- into the
 * merged frame-major stream, band after band inside every frame = raster order (video.rs:677-734).  Contexts in the lean
 * regime only (Collapse, delta_t_max <= time_spanned, no feature mode): others fail with ADDER_E_BAD_PARAMS and the
 * caller gathers events (adder_gather_events).  The tables and the records live in the band context's scratch until its
 * next batch; a peer's copies of them (received over RCCL) are described by the same struct with root's pointers. */
#define ADDER_RECORDS_RUNS 0x100u
typedef struct AdderBandRecords {
    uint32_t num_frames;        /* rows of the three tables */
    uint32_t num_segments;      /* columns: the band's 128-unit segments, padded (adder_hip_band_segments) */
    uint32_t record_bytes;      /* 8 (DeltaT) or 12 (AbsoluteT), | ADDER_RECORDS_RUNS when the batch ran the lean-runs kernel:
                                 * {rho, [last_fired_t / T,] unit | base_val << 8 | input << 16} records, from which root works the
                                 * events out (constant runs: crf 0, one integer time step since the reset); every band of a
                                 * gathered chunk carries the same value */
    uint32_t row_begin, rows;   /* the band */
    const uint32_t *d_counts;   /* [num_frames][num_segments] events | records << 16 of the segment in the frame */
    const uint32_t *d_prefix;   /* [num_frames][num_segments] events of the frame before the segment, inside the band */
    const uint32_t *d_runs;     /* [num_frames][num_segments] first record of the segment's run in d_records */
    const uint8_t *d_records;   /* the batch's records, adder_hip_last_batch_records() of them after adder_hip_finish */
    const uint64_t *d_frame_offsets; /* [num_frames + 1] the band's own event offsets of the batch */
    const void *d_frame_table;  /* root only, may be null: [num_frames] x 8 bytes, the frames' {running_t, c_thresh} as
                                 * root's own batch had them (null: root's current table -- valid until ITS next batch) */
} AdderBandRecords;
uint32_t adder_hip_band_segments(const AdderHipCtx *ctx);
/* Like adder_hip_integrate_device, without an event buffer: d_frame_offsets ([num_frames + 1], device) receives the
 * band's event offsets, *out the description (device pointers into the context's scratch, valid until its next batch).
 * adder_hip_finish completes the batch as usual (its count is the events the batch WOULD emit). */
int adder_hip_integrate_records_device(AdderHipCtx *ctx, const uint8_t *d_frames, uint32_t num_frames, float time_spanned,
                                       uint64_t *d_frame_offsets, void *stream, AdderBandRecords *out);
/* On root, after its own adder_hip_integrate_records_device of the SAME frames (root's frame table gives the frames'
 * running_t): bands[0 .. n_bands) in raster order, every pointer in root's memory.  Appends the frames' events to
 * d_merged at event index merged_base and writes d_merged_offsets[0 .. num_frames] (absolute: [0] = merged_base).
 * Queues on `stream`; a d_merged too small is reported by the context's status at the next adder_hip_finish /
 * adder_hip_expand_status (events past merged_cap are dropped). */
int adder_hip_expand_records_device(AdderHipCtx *root, const AdderBandRecords *bands, uint32_t n_bands, AdderEvent *d_merged,
                                    size_t merged_cap, uint64_t merged_base, uint64_t *d_merged_offsets, void *stream);
/* The same with the raw sink's records as root's output (what adder_hip_integrate_wire_device leaves on one GPU; replaces
 * video.rs:736-740 -> raw/stream.rs:101-120 for the merged stream): d_wire receives 9 / 11-byte records back to back, event
 * k of the merged stream at byte k * record size; merged_base and the offsets still count events. */
int adder_hip_expand_records_wire_device(AdderHipCtx *root, const AdderBandRecords *bands, uint32_t n_bands, uint8_t *d_wire,
                                         size_t wire_cap_bytes, uint64_t merged_base, uint64_t *d_merged_offsets, void *stream);
/* One contiguous image of a band's batch, for the transport: sections at 256-byte multiples,
 *   frame offsets (num_frames + 1) x 8 | frame table num_frames x 8 | counts | prefix | runs (num_frames x num_segments x 4
 *   each) | n_records x record_bytes.
 * adder_hip_records_wire_bytes gives its size, adder_hip_records_wire_sections the byte offset of each of the six sections
 * (so that the receiver rebuilds an AdderBandRecords from a received image + the five numbers), and
 * adder_hip_records_to_wire queues the device-to-device copies of `rec` (this context's last records batch) into d_dst on
 * `stream` -- after it the context may run its next batch while the image travels. */
size_t adder_hip_records_wire_bytes(uint32_t num_frames, uint32_t num_segments, uint32_t record_bytes, uint64_t n_records);
void adder_hip_records_wire_sections(uint32_t num_frames, uint32_t num_segments, uint32_t record_bytes, size_t sections[6]);
int adder_hip_records_to_wire(AdderHipCtx *ctx, const AdderBandRecords *rec, uint64_t n_records, void *d_dst, size_t dst_bytes,
                              void *stream);
/* The stream the context's last device batch was queued on (what a null `stream` of adder_hip_records_to_wire means: the
 * copies are then ordered before the context's next batch, whatever stream the transport uses). */
void *adder_hip_last_batch_stream(AdderHipCtx *ctx);
/* Waits for it (a transport that cannot order its own stream behind that one -- e.g. one that only knows torch streams
 * while the batch ran on the context's own -- calls this after adder_hip_records_to_wire and before it sends). */
int adder_hip_sync_last_batch_stream(AdderHipCtx *ctx);
/* Waits for `stream` and returns ADDER_OK or the failure the expansions since the last call ran into (capacity). */
int adder_hip_expand_status(AdderHipCtx *root, void *stream);

/* ---- sink per rank ("each GPU D2H's its own segment and the host concatenates -- 8 PCIe links vs one";
 * consumer: video.rs:736-740 -> encoder.rs:233-273 -> raw/stream.rs:101-120).  The building blocks
 * adder_gather_host_sink_* (include/adder_gather.h) are made of; usable on their own with any means of exchanging the
 * ranks' per-frame offsets.
 * adder_hip_sink_layout_device: d_all_offsets = the ranks' frame offsets of ONE chunk, device uint64 [world][num_frames + 1]
 * (values as adder_hip_integrate_device left them).  *d_file_pos (device) is the number of events the image holds before
 * the chunk; the call writes d_dest[f] = the event index at which THIS rank's events of frame f start in the image, advances
 * *d_file_pos by the chunk's total and, if d_merged_offsets is not null, writes the merged offsets [num_frames + 1].
 * adder_hip_wire_scatter_device: this rank's events of the chunk (d_events indexed by d_frame_offsets' values) as 9 / 11-byte
 * wire records at out + header_bytes + d_dest[f] * record_bytes.  `out` is anything the device can store to: HBM, or host
 * memory mapped into the device (hipHostRegister / hipHostMalloc) -- then the stores cross this GPU's own PCIe link and
 * the bytes land where the file has them.  Bytes past out_cap_bytes are dropped and reported by adder_hip_expand_status.
 * Both queue on `stream`; nothing waits on the host. */
int adder_hip_sink_layout_device(AdderHipCtx *ctx, const uint64_t *d_all_offsets, uint32_t world, uint32_t rank,
                                 uint32_t num_frames, uint64_t *d_file_pos, uint64_t *d_dest, uint64_t *d_merged_offsets,
                                 void *stream);
int adder_hip_wire_scatter_device(AdderHipCtx *ctx, const AdderEvent *d_events, const uint64_t *d_frame_offsets,
                                  uint32_t num_frames, const uint64_t *d_dest, uint8_t *out, uint64_t out_cap_bytes,
                                  uint64_t header_bytes, void *stream);

/* Mean number of frames one timed frame-kernel launch stepped (see frames_per_launch). */
float adder_hip_last_launch_frames(AdderHipCtx *ctx);

/* Temporal blocking depth of the frame kernel: how many consecutive frames of a batch one
 * launch steps with the pixel state held in registers (1..32, default 32).  Results do not
 * depend on it.  Scratch is handled in chunks of at most 16 frames and a launch never spans
 * two chunks; batches that want the running-intensities side plane run one frame per launch. */
int adder_hip_set_frames_per_launch(AdderHipCtx *ctx, uint32_t frames);

/* Back to the state right after adder_hip_create (Video::new): every pixel pristine,
 * c_thresh/counter = c_thresh_start/c_counter_start, running_t = 0, poison cleared.
 * Parameters set since (crf parameters, delta_t_max, time mode) are kept. */
int adder_hip_reset(AdderHipCtx *ctx);

/* --- raw sink on the device ---------------------------------------------------------
 * Replaces the per-event `RawOutput::ingest_event` loop (adder-codec-core/src/codec/raw/stream.rs:101-120,
 * reached from video.rs:736-740 through encoder.rs:233-273): the events are serialised to the
 * `.adder` wire form (bincode fixint big-endian: 9-byte EventSingle on a 1-channel plane, 11-byte
 * Event {x, y, Some(c), d, t} otherwise) by a kernel, so the host writes the bytes as they are.
 * Header and EOF stay on the host (adder_raw_header / adder_raw_eof).
 *
 * adder_hip_wire_events_device: `n_events` events of this context at d_events -> d_out (device
 * memory), asynchronously on `stream`; *n_bytes = n_events * 9 (or 11).
 * adder_hip_integrate_batch_raw: adder_hip_integrate_batch with the serialisation appended; out_bytes
 * (host memory, pinned for full speed) receives the records of all T frames in stream order;
 * frame_offsets (optional, T+1) are EVENT indices as in adder_hip_integrate_batch. */
int adder_hip_wire_events_device(AdderHipCtx *ctx, const AdderEvent *d_events, size_t n_events,
                                 uint8_t *d_out, size_t out_cap_bytes, size_t *n_bytes, void *stream);
int adder_hip_integrate_batch_raw(AdderHipCtx *ctx, const uint8_t *frames, uint32_t num_frames,
                                  size_t frame_stride_bytes, size_t row_stride_bytes, float time_spanned,
                                  uint8_t *out_bytes, size_t out_cap_bytes, size_t *n_bytes,
                                  size_t *n_events, uint64_t *frame_offsets);

/* Pipelined form for whole clips (what the reference's SimulProcessor / Framed::consume loop does frame
 * by frame, utils/simulproc.rs:233): submit(k) uploads and integrates batch k, then queues its
 * serialisation and download on a second stream and returns; collect() waits for the oldest batch in
 * flight and hands out its wire bytes (a pinned host buffer owned by the context, valid until the batch
 * after next is submitted).  With submit(k+1) called before collect(k) the download of batch k overlaps
 * the upload + integration of batch k+1.  At most two batches are in flight.  frames should be pinned
 * (adder_hip_alloc_pinned) for the upload to run at link speed. */
int adder_hip_stream_submit(AdderHipCtx *ctx, const uint8_t *frames, uint32_t num_frames,
                            size_t frame_stride_bytes, size_t row_stride_bytes, float time_spanned,
                            size_t out_cap_events);
int adder_hip_stream_collect(AdderHipCtx *ctx, const uint8_t **bytes, size_t *n_bytes, size_t *n_events,
                             const uint64_t **frame_offsets);

/* --- self-test ----------------------------------------------------------------------
 * The lean step replaces the one f32 division of integrate_main (event_pixel_tree.rs:431,445)
 * by a 4-instruction sequence that is correctly rounded on the domain it is used on
 * (integer numerator in [1, 2^24], denominator = u8 intensity in [1, 255]).  This runs both
 * on all 4.3e9 pairs on the current device and returns the number of differing results
 * (0 on gfx950). */
int adder_hip_selftest_division(uint64_t *mismatches);

/* --- deterministic synthetic clips generated directly in HBM ------ */
enum { ADDER_CONTENT_STATIC = 0, ADDER_CONTENT_NOISE = 1, ADDER_CONTENT_SCENE = 2 };
int adder_hip_synth_clip_device(uint8_t *d_dst, int content, uint64_t seed, uint32_t width,
                                uint32_t height, uint32_t channels, uint32_t row_begin,
                                uint32_t rows, uint32_t frame_begin, uint32_t num_frames,
                                void *stream);

/* --- raw `.adder` sink (encoder.rs:170-229, raw/stream.rs:79-120), host side -------- */
size_t adder_raw_header(uint8_t *dst, uint8_t codec_version, uint16_t width, uint16_t height,
                        uint8_t channels, uint32_t tps, uint32_t ref_interval, uint32_t delta_t_max,
                        uint32_t source_camera, uint32_t time_mode, uint32_t adu_interval);
size_t adder_raw_events(uint8_t *dst, const AdderEvent *events, size_t n, uint8_t channels);
size_t adder_raw_eof(uint8_t *dst);

/* --- multi-GPU: ordered merge of the row bands' streams -----------------------
 * The reference splits a frame by rows (video.rs:677-691) and concatenates the chunks' events in row
 * order (video.rs:742-765).  With one context per GPU / row band, rank r produces a frame-major stream
 * with offsets offs_r[0..T]; the single-context stream is, per frame, rank 0's events, then rank 1's, ...
 * d_stage: the ranks' streams back to back (rank r at the sum of the lower ranks' totals), e.g. the
 * receive buffer of an ordered gather (include/adder_gather.h does that over RCCL);
 * d_rank_offsets: device uint64 [world][T+1]; d_work: adder_hip_merge_work_bytes() bytes of device
 * scratch; d_merged_offsets (may be NULL): device uint64 [T+1].  Asynchronous on `stream`; a capacity
 * overflow is reported by adder_hip_check_status (which synchronises the stream). */
size_t adder_hip_merge_work_bytes(uint32_t world, uint32_t num_frames);
int adder_hip_merge_streams_device(AdderHipCtx *ctx, const AdderEvent *d_stage, const uint64_t *d_rank_offsets,
                                   uint32_t world, uint32_t num_frames, void *d_work, AdderEvent *d_out,
                                   size_t out_cap, uint64_t *d_merged_offsets, void *stream);
/* The same for a CHUNK of a longer stream (the gather is pipelined chunk by chunk behind the integration): a rank's
 * offsets may start anywhere (they are taken relative to their first entry), d_out is where the chunk's first merged
 * event goes, d_merged_offsets the entry of the chunk's first frame, and the offsets written continue from merged_base
 * = the events of the merged stream before this chunk. */
int adder_hip_merge_streams_device_at(AdderHipCtx *ctx, const AdderEvent *d_stage, const uint64_t *d_rank_offsets,
                                      uint32_t world, uint32_t num_frames, void *d_work, AdderEvent *d_out,
                                      size_t out_cap, uint64_t *d_merged_offsets, uint64_t merged_base, void *stream);
int adder_hip_check_status(AdderHipCtx *ctx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ADDER_HIP_H */
