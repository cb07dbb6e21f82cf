/* adder_framer.h -- C-ABI of the MI355X-native framer (ADDER events -> u8 / u16 / u32 frames).
 *
 * Replaces, for T = u8 / u16 / u32 (value_type), the
 * reference's FrameSequence<u8> (adder-codec-rs/src/framer/driver.rs):
 *   FramerBuilder::new / time_parameters / codec_version / source / finish   driver.rs:55-138
 *   Framer::ingest_event / ingest_events_events                               driver.rs:437-626 (+ :984-1133)
 *   Framer::flush_frame_buffer                                                driver.rs:632-677
 *   is_frame_filled(0) + write_multi_frame_bytes / write_frame_bytes          driver.rs:807-825, 935-981
 * i.e. the consumer the reference's SimulProcessor feeds with the transcoder's events
 * (utils/simulproc.rs:161-176) and its `dark` test compares byte for byte.
 *
 * Frame content does not depend on WHEN complete frames are popped (a frame is only handed out once
 * every pixel has a value in it, and each (pixel, frame) is written once), so this ABI ingests whole
 * batches and pops complete frames afterwards; the bytes equal the reference's output stream.
 *
 * Views: every arm of <u8 as FrameValue>::get_frame_value (scale_intensity.rs:54-109) -- Intensity for U8 / U16 /
 * U32 / U64 sources, D, DeltaT, SAE -- through view_mode / source_type below.  FramerMode::INTEGRATION needs no
 * switch: the reference stores the mode (driver.rs:270, 383) and never reads it, both modes ingest alike.
 * Frame element types: u8, and u16 / u32 (<u16 / u32 as FrameValue>, scale_intensity.rs:111-209) whose frames are popped
 * as the big-endian bincode bytes the reference's writers produce (driver.rs:279,395-398,944).  FrameSequence<u64> cannot
 * be instantiated in the reference (its methods need T: Into<f64>).  Not built: EventCoordless frames, feature detection
 * on frames, buffer_limit.
 */
#ifndef ADDER_FRAMER_H
#define ADDER_FRAMER_H

#include <stddef.h>
#include <stdint.h>

#include "adder_hip.h" /* AdderEvent, status codes, ADDER_TIME_* */

#ifdef __cplusplus
extern "C" {
#endif

#define ADDER_FRAMER_ABI_VERSION 2u

typedef struct AdderFramerParams {
    uint32_t abi_version;   /* ADDER_FRAMER_ABI_VERSION */
    uint16_t width, height; /* PlaneSize (driver.rs:57) */
    uint8_t channels;
    uint8_t codec_version;  /* .codec_version(v, time_mode) (driver.rs:117-124) */
    uint8_t time_mode;      /* ADDER_TIME_*; AbsoluteT handling needs codec_version >= 2 (:1001) */
    uint8_t reserved0;
    uint32_t row_begin, row_end; /* row band owned by this context (multi-GPU sharding); 0, height for all */
    uint32_t tps, ref_interval, delta_t_max; /* .time_parameters(..) (driver.rs:77-90) */
    float output_fps;       /* Some(fps): tpf = (tps as f32 / fps) as u32; <= 0: None, tpf = ref_interval (:357-361) */
    uint32_t source_camera; /* SourceCamera discriminant (adder-codec-core/src/lib.rs:35-47); 0..5 = framed */
    uint32_t ring_frames;   /* frames kept on the device; 0 = delta_t_max / tpf + 80 */
    int32_t device_id;
    /* what a frame's byte shows: <u8 as FrameValue>::get_frame_value (framer/scale_intensity.rs:54-109) */
    uint8_t view_mode;      /* FramedViewMode (video.rs:144-158): 0 Intensity, 1 D, 2 DeltaT, 3 SAE */
    uint8_t source_type;    /* SourceType of the intensities (Intensity view): 0 U8, 1 U16, 2 U32, 3 U64 */
    uint8_t value_type;     /* the frame element type T: ADDER_FRAME_U8 (0, default) / _U16 / _U32; pops hand out
                             * [n][rows][width][channels] elements of 1 / 2 / 4 bytes, big-endian */
    uint8_t reserved1;
    float practical_d_max;  /* D view: fast_math::log2_raw(255.0 * (delta_t_max / ref_interval) as f32), computed by
                             * the caller (driver.rs:1020-1021; the approximation belongs to a third-party crate) */
} AdderFramerParams;

#define ADDER_VIEW_INTENSITY 0
#define ADDER_VIEW_D 1
#define ADDER_VIEW_DELTA_T 2
#define ADDER_VIEW_SAE 3
#define ADDER_FRAME_U8 0
#define ADDER_FRAME_U16 1
#define ADDER_FRAME_U32 2

typedef struct AdderFramer AdderFramer;

void adder_framer_default_params(AdderFramerParams *p, uint16_t width, uint16_t height, uint8_t channels);
int adder_framer_create(const AdderFramerParams *p, AdderFramer **out);
void adder_framer_destroy(AdderFramer *fr);
const char *adder_framer_last_error(const AdderFramer *fr);
uint32_t adder_framer_tpf(const AdderFramer *fr);             /* FrameSequenceState::tpf */
int64_t adder_framer_frames_written(const AdderFramer *fr);   /* FrameSequenceState::frames_written */

/* Ingest events (Framer::ingest_event for each, in order).  The stream is given as `num_segments`
 * consecutive segments, events [seg_offsets[s], seg_offsets[s+1]); INSIDE a segment all events of one
 * pixel-channel must be contiguous (their order is kept).  Each per-frame segment produced by
 * adder_hip_integrate* has that shape: pass its frame_offsets.  seg_offsets is host memory.
 * d_events is device memory (ingest_device, asynchronous on `stream`) or host memory (ingest). */
int adder_framer_ingest_device(AdderFramer *fr, const AdderEvent *d_events, const uint64_t *seg_offsets,
                               uint32_t num_segments, void *stream);
int adder_framer_ingest(AdderFramer *fr, const AdderEvent *events, const uint64_t *seg_offsets,
                        uint32_t num_segments);

/* The same for the transcoder's own output -- `num_frames` per-frame segments, each in RASTER order
 * (events sorted by y, then x, then c: what adder_hip_integrate* emits; frame_offsets = its frame_offsets,
 * host memory).  One launch for the whole batch: every workgroup owns a few rows for all frames.  A
 * stream that is not in raster order inside its segments is reported (ADDER_E_BAD_PARAMS), use
 * adder_framer_ingest_device for those. */
int adder_framer_ingest_frames_device(AdderFramer *fr, const AdderEvent *d_events, const uint64_t *frame_offsets,
                                      uint32_t num_frames, void *stream);

/* The same with frame_offsets in DEVICE memory (where adder_hip_integrate_device leaves them): the transcoder's batch
 * and its framing queue back to back on one stream, nothing crosses the bus and the host does not wait.  Offsets that
 * decrease are reported by the next call that synchronises (frames_ready / pop / flush). */
int adder_framer_ingest_frames_device_offsets(AdderFramer *fr, const AdderEvent *d_events,
                                              const uint64_t *d_frame_offsets, uint32_t num_frames, void *stream);

/* Number of complete frames waiting (is_frame_filled(0), (1), ...).  Synchronises. */
int adder_framer_frames_ready(AdderFramer *fr, uint32_t *n_ready);

/* write_multi_frame_bytes: pops up to max_frames complete frames, [n][rows][width][channels] u8, into
 * device (pop_device) or host (pop) memory. */
int adder_framer_pop_device(AdderFramer *fr, uint8_t *d_out, uint32_t max_frames, uint32_t *n_popped, void *stream);
int adder_framer_pop(AdderFramer *fr, uint8_t *out, uint32_t max_frames, uint32_t *n_popped);

/* write_frame_bytes: pops frame 0 whether or not it is complete (pixels without a value read 0). */
int adder_framer_write_frame(AdderFramer *fr, uint8_t *out);

/* flush_frame_buffer: if any pixel has reached beyond frame 0, fill frame 0's missing pixels with their
 * last intensity (*frame0_ready = 1); the frame must be popped before the next ingest. */
int adder_framer_flush(AdderFramer *fr, int *frame0_ready);

#ifdef __cplusplus
}
#endif
#endif /* ADDER_FRAMER_H */
