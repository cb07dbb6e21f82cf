/* adder_compressed.h -- C-ABI of the CPU compressed ADDER sink / source.
 *
 * What the reference does in Encoder::new_compressed + CompressedOutput / CompressedInput
 * (adder-codec-core/src/codec/compressed/stream.rs:126-424): events are grouped into ADUs of
 * `adu_interval` reference intervals, every ADU is a grid of 16x16 EventCubes
 * (source_model/event_structure/event_adu.rs:83-117, event_cube.rs:309-517) whose first events are
 * intra-coded and later events inter-coded as (D residual, bit shift, t residual) symbols through an
 * adaptive Fenwick-tree model with four contexts (source_model/cabac_contexts.rs:26-134,
 * fenwick/context_switching.rs:10-100) into a 33-bit integer range coder
 * (arithmetic-coding-adder-dep/src/encoder.rs).  The stream is the "addec" header followed, per ADU,
 * by a 32-bit big-endian byte count and the ADU's bytes.
 *
 * This stage stays on the CPU (BASELINE.json north_star: "the unchanged CPU arithmetic-coding stage"):
 * libadder_hip.so carries it as plain C++ (csrc/adder_compressed.cpp), ADUs are compressed on worker
 * threads exactly as the reference spawns one thread per ADU.  Integer-only except the f64 intensity test
 * of the lossy bit shift (cabac_contexts.rs:75-150): byte-identical to the restatement in
 * oracle/compressed_oracle.py, which is pinned by the reference's own round-trip tests.
 *
 * A Rust host binds these with `extern "C"` in place of CompressedOutput::ingest_event / into_writer. */
#ifndef ADDER_COMPRESSED_H
#define ADDER_COMPRESSED_H

#include <stddef.h>
#include <stdint.h>

#include "adder_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ADDER_COMPRESSED_ABI_VERSION 1u

typedef struct AdderCompressedParams {
    uint32_t abi_version;     /* ADDER_COMPRESSED_ABI_VERSION */
    uint16_t width, height;   /* CodecMetadata::plane */
    uint8_t channels;         /* 1 or 3 */
    uint8_t codec_version;    /* header version (LATEST_CODEC_VERSION = 3) */
    uint8_t time_mode;        /* ADDER_TIME_* (header extension V2) */
    uint8_t write_header;     /* 1: Encoder::new_compressed (header first); 0: a bare CompressedOutput */
    uint32_t tps, ref_interval, delta_t_max;
    uint32_t adu_interval;    /* reference intervals per ADU (header extension V3) */
    uint32_t source_camera;   /* SourceCamera discriminant (header extension V1) */
    uint8_t c_thresh_max;     /* EncoderOptions.crf.get_parameters().c_thresh_max: the lossy t tolerance */
    uint8_t reserved[3];
    uint32_t threads;         /* ADU compression workers; 0 = a default */
} AdderCompressedParams;

typedef struct AdderCompressedEncoder AdderCompressedEncoder;

void adder_compressed_default_params(AdderCompressedParams *p, uint16_t width, uint16_t height, uint8_t channels);
int adder_compressed_encoder_create(const AdderCompressedParams *p, AdderCompressedEncoder **out);
void adder_compressed_encoder_destroy(AdderCompressedEncoder *e);
const char *adder_compressed_last_error(const AdderCompressedEncoder *e);
/* Encoder::ingest_events -> CompressedOutput::ingest_event (stream.rs:268-319), n events in stream order. */
int adder_compressed_encoder_ingest(AdderCompressedEncoder *e, const AdderEvent *events, size_t n);
/* Encoder::close_writer -> CompressedOutput::into_writer (stream.rs:179-262): compresses the partial last ADU,
 * waits for every ADU and hands out the whole stream (header + ADUs); the pointer stays valid until destroy. */
int adder_compressed_encoder_close(AdderCompressedEncoder *e, const uint8_t **bytes, size_t *n_bytes);
/* ADUs finished (compressed and appended to the stream) so far, and the stream's current size. */
int adder_compressed_encoder_progress(AdderCompressedEncoder *e, uint32_t *adus_written, size_t *n_bytes);

/* CompressedInput::digest_event until the data runs out (stream.rs:377-424).  `params`: the stream's meta (from the
 * header when header_size > 0 is found, else taken from *params as given: a bare CompressedOutput stream).
 * out may be NULL to count: *n_out = events in the stream.  Returns ADDER_E_OUT_CAPACITY if out_cap is too small. */
int adder_compressed_decode(const uint8_t *data, size_t size, int has_header, AdderCompressedParams *params,
                            AdderEvent *out, size_t out_cap, size_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* ADDER_COMPRESSED_H */
