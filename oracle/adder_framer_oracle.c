/* adder_framer_oracle.c -- CPU restatement of the reference's event -> frame reconstruction
 * (FrameSequence<u8>, FramerMode::INSTANTANEOUS, FramedViewMode::Intensity, SourceType::U8).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/oracle.py): used by tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py as the checker; the product never links or calls it.
 *
 * Follows, function by function:
 *   FramerBuilder / FrameSequence::new           adder-codec-rs/src/framer/driver.rs:55-138, 302-399
 *   Framer::ingest_event                          driver.rs:437-562 (feature detection off)
 *   Framer::ingest_events_events                  driver.rs:564-626
 *   Framer::flush_frame_buffer                    driver.rs:632-677
 *   is_frame_filled / is_frame_0_filled           driver.rs:807-844
 *   pop_next_frame_for_chunk                      driver.rs:903-925
 *   write_frame_bytes / write_multi_frame_bytes   driver.rs:935-981
 *   ingest_event_for_chunk                        driver.rs:984-1133
 *   <u8 as FrameValue>::get_frame_value           framer/scale_intensity.rs:54-109 (Intensity arm)
 *   event_to_intensity                            framer/scale_intensity.rs:262-270
 *
 * Pinned by (tests/test_framer_oracle.py): the reference's own unit tests get_frame_bytes_u8 and
 * test_get_empty_frame (tests/integration_tests.rs:555-611, 782-820), its sample_3_{ordered,
 * unordered}.adder -> sample_3.gray vectors (405 frames, :822-975) and the `dark` test's
 * lake_scaled_hd_out.adder -> lake_scaled_out (src/bin/adder_simulproc.rs:170-268).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define D_EMPTY 255u

typedef struct {
    uint16_t x, y;
    uint8_t c, d; /* c = 0xFF: None */
    uint16_t pad;
    uint32_t t;
} OracleEvent;

typedef struct {
    uint32_t *val; /* Option<T> payload, T = u8 / u16 / u32 (value_type) */
    uint8_t *some; /* Option<T> discriminant */
    size_t filled_count;
} OFrame;

typedef struct {
    OFrame *q; /* VecDeque<Frame<Option<u8>>>, front = q[0] */
    size_t len, cap;
    size_t px; /* array.len() of this chunk */
} ODeque;

typedef struct {
    /* FrameSequenceState (driver.rs:223-240) */
    int64_t frames_written;
    uint16_t w, h;
    uint8_t c;
    uint32_t tpf, tps;
    uint8_t codec_version;
    uint32_t source_camera; /* 0..5 framed, >= 6 event cameras (lib.rs:35-47) */
    uint32_t ref_interval, source_dtm;
    int time_mode; /* 0 DeltaT, 1 AbsoluteT, 2 Mixed */
    /* FrameSequence (driver.rs:261-287) */
    size_t chunk_rows, num_chunks;
    ODeque *frames;
    int64_t *frame_idx_offsets;
    uint64_t **pixel_ts;
    int64_t **last_filled;
    uint32_t **last_intensity;
    uint8_t *chunk_filled;
    int has_buffer_limit;
    uint32_t buffer_limit;
    /* FramerBuilder::view_mode / ::source (driver.rs:104-107, 132-140); practical_d_max as the caller computes it
     * (fast_math::log2_raw, a third-party approximation that is not restated here) */
    int view_mode, source_type;
    float practical_d_max;
    /* the frame element type T of FrameSequence<T>: 0 u8, 1 u16, 2 u32 (scale_intensity.rs:54-211).  FrameSequence<u64>
     * cannot be instantiated in the reference (its methods need T: Into<f64>, driver.rs:135,298,689,985, which u64 is not) */
    int value_type;
} OracleFramer;

static void frame_init(OFrame *f, size_t px) {
    f->val = (uint32_t *)calloc(px ? px : 1, sizeof(uint32_t));
    f->some = (uint8_t *)calloc(px ? px : 1, 1);
    f->filled_count = 0;
}
static void frame_free(OFrame *f) {
    free(f->val);
    free(f->some);
}
static void deque_push_back_empty(ODeque *d, size_t n) {
    if (d->len + n > d->cap) {
        size_t nc = d->cap ? d->cap : 4;
        while (nc < d->len + n) nc *= 2;
        d->q = (OFrame *)realloc(d->q, nc * sizeof(OFrame));
        d->cap = nc;
    }
    for (size_t i = 0; i < n; ++i) frame_init(&d->q[d->len++], d->px);
}

/* FrameSequence::new (driver.rs:302-399); output_fps < 0 means None */
OracleFramer *oracle_framer_new(uint16_t w, uint16_t h, uint8_t c, uint32_t chunk_rows, uint32_t tps,
                                uint32_t ref_interval, uint32_t delta_t_max, float output_fps,
                                uint8_t codec_version, int time_mode, uint32_t source_camera) {
    if (!w || !h || !c || !chunk_rows) return NULL;
    OracleFramer *f = (OracleFramer *)calloc(1, sizeof *f);
    f->w = w;
    f->h = h;
    f->c = c;
    f->chunk_rows = chunk_rows;
    f->num_chunks = (h + chunk_rows - 1) / chunk_rows; /* ceil(h / chunk_rows) (:309) */
    const size_t last_rows = h - (f->num_chunks - 1) * chunk_rows;
    f->frames = (ODeque *)calloc(f->num_chunks, sizeof(ODeque));
    f->frame_idx_offsets = (int64_t *)calloc(f->num_chunks, sizeof(int64_t));
    f->pixel_ts = (uint64_t **)calloc(f->num_chunks, sizeof(void *));
    f->last_filled = (int64_t **)calloc(f->num_chunks, sizeof(void *));
    f->last_intensity = (uint32_t **)calloc(f->num_chunks, sizeof(void *));
    f->chunk_filled = (uint8_t *)calloc(f->num_chunks, 1);
    for (size_t k = 0; k < f->num_chunks; ++k) {
        const size_t rows = k + 1 == f->num_chunks ? last_rows : chunk_rows;
        const size_t px = rows * w * c;
        f->frames[k].px = px;
        deque_push_back_empty(&f->frames[k], 1);
        f->pixel_ts[k] = (uint64_t *)calloc(px, sizeof(uint64_t));
        f->last_filled[k] = (int64_t *)malloc(px * sizeof(int64_t));
        for (size_t i = 0; i < px; ++i) f->last_filled[k][i] = -1; /* :351-355 */
        f->last_intensity[k] = (uint32_t *)calloc(px, sizeof(uint32_t));
    }
    /* :357-361: tpf = (tps as f32 / output_fps) as u32, or ref_interval */
    if (output_fps >= 0.0f) {
        const float q = (float)tps / output_fps;
        f->tpf = q >= 4294967296.0f ? 0xffffffffu : (q > 0.0f ? (uint32_t)q : 0u);
    } else {
        f->tpf = ref_interval;
    }
    f->tps = tps;
    f->codec_version = codec_version;
    f->source_camera = source_camera;
    f->ref_interval = ref_interval;
    f->source_dtm = delta_t_max;
    f->time_mode = time_mode;
    return f;
}

void oracle_framer_free(OracleFramer *f) {
    if (!f) return;
    for (size_t k = 0; k < f->num_chunks; ++k) {
        for (size_t i = 0; i < f->frames[k].len; ++i) frame_free(&f->frames[k].q[i]);
        free(f->frames[k].q);
        free(f->pixel_ts[k]);
        free(f->last_filled[k]);
        free(f->last_intensity[k]);
    }
    free(f->frames);
    free(f->frame_idx_offsets);
    free(f->pixel_ts);
    free(f->last_filled);
    free(f->last_intensity);
    free(f->chunk_filled);
    free(f);
}

void oracle_framer_buffer_limit(OracleFramer *f, int has, uint32_t limit) {
    f->has_buffer_limit = has;
    f->buffer_limit = limit;
}
void oracle_framer_set_view(OracleFramer *f, int view_mode, int source_type, float practical_d_max) {
    f->view_mode = view_mode;
    f->source_type = source_type;
    f->practical_d_max = practical_d_max;
}
int oracle_framer_set_value_type(OracleFramer *f, int value_type) {
    if (value_type < 0 || value_type > 2) return -1;
    f->value_type = value_type;
    return 0;
}
uint32_t oracle_framer_tpf(const OracleFramer *f) { return f->tpf; }
int64_t oracle_framer_frames_written(const OracleFramer *f) { return f->frames_written; }
size_t oracle_framer_num_chunks(const OracleFramer *f) { return f->num_chunks; }

/* event_to_intensity (scale_intensity.rs:262-270) and the U8 / Intensity arm of get_frame_value
 * (:69-72): (intensity * tpf) as u8, with `tpf` = ref_interval as f64 at the call site (:1034) */
static uint8_t frame_value_u8(uint8_t d, uint32_t t, double tpf) {
    double intensity;
    if (d >= 129) {
        intensity = 0.0;
    } else {
        double shift = 0.0; /* D_SHIFT_F64[128] = 0 */
        if (d < 128) {
            shift = 1.0;
            for (unsigned i = 0; i < d; ++i) shift *= 2.0;
        }
        intensity = t == 0 ? shift : shift / (double)t;
    }
    const double v = intensity * tpf;
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

/* Rust's `as u8` from f32: saturating, NaN -> 0 */
static uint8_t f32_as_u8(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
static uint8_t f64_as_u8(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}
static double event_to_intensity(uint8_t d, uint32_t t) { /* scale_intensity.rs:262-270 */
    if (d >= 129) return 0.0;
    double shift = 0.0; /* D_SHIFT_F64[128] = 0 */
    if (d < 128) {
        shift = 1.0;
        for (unsigned i = 0; i < d; ++i) shift *= 2.0;
    }
    return t == 0 ? shift : shift / (double)t;
}
/* <u8 as FrameValue>::get_frame_value, every arm (scale_intensity.rs:54-109); px = SaeTime {running_t, last_fired_t}.
 * view: 0 Intensity, 1 D, 2 DeltaT, 3 SAE (FramedViewMode, video.rs:144-158); source: 0 U8, 1 U16, 2 U32, 3 U64 */
static uint8_t get_frame_value_u8(uint8_t d, uint32_t t, int source, double tpf, float practical_d_max,
                                  uint32_t delta_t_max, int view, uint32_t running_t, uint32_t last_fired_t) {
    switch (view) {
    case 1: return f32_as_u8(((float)d / practical_d_max) * 255.0f);
    case 2: return f32_as_u8(((float)t / (float)delta_t_max) * 255.0f);
    case 3: return f32_as_u8(((float)(running_t - last_fired_t) / (float)delta_t_max) * 255.0f);
    default: break;
    }
    const double intensity = event_to_intensity(d, t);
    switch (source) {
    case 0: return f64_as_u8(intensity * tpf);
    case 1: return f64_as_u8(intensity / 65535.0 * tpf * 255.0);
    case 2: return f64_as_u8(intensity / 4294967295.0 * tpf * 255.0);
    default: return f64_as_u8(intensity / 18446744073709551615.0 * tpf * 255.0); /* u64::MAX as f64 */
    }
}

/* Rust's `as u16` / `as u32` from a float: saturating, NaN -> 0 */
static uint32_t f64_as_uint(double v, double max) {
    if (!(v > 0.0)) return 0;
    if (v >= max) return (uint32_t)max;
    return (uint32_t)v;
}
static uint32_t f32_as_uint(float v, double max) {
    if (!(v > 0.0f)) return 0;
    if ((double)v >= max) return (uint32_t)max;
    return (uint32_t)v;
}
/* <u16 as FrameValue>::get_frame_value (scale_intensity.rs:111-160) and <u32 ...> (:162-209); SAE is todo!() there.
 * value_type 1: u16, 2: u32.  Returns -1 for the arms the reference does not implement. */
static int get_frame_value_wide(int value_type, uint8_t d, uint32_t t, int source, double tpf, float practical_d_max,
                                uint32_t delta_t_max, int view, uint32_t *out) {
    const double tmax = value_type == 1 ? 65535.0 : 4294967295.0;
    /* f32::from(u16::MAX) = 65535.0; u32::MAX as f32 = 4294967296.0 (rounded to the nearest f32) */
    const float tmax_f32 = value_type == 1 ? 65535.0f : 4294967296.0f;
    switch (view) {
    case 1: *out = f32_as_uint(((float)d / practical_d_max) * tmax_f32, tmax); return 0;
    case 2: *out = f32_as_uint(((float)t / (float)delta_t_max) * tmax_f32, tmax); return 0;
    case 3: return -1; /* todo!() */
    default: break;
    }
    const double intensity = event_to_intensity(d, t);
    if (source == value_type) { /* the source's own type: (intensity * tpf) as T */
        *out = f64_as_uint(intensity * tpf, tmax);
        return 0;
    }
    const double smax = source == 0 ? 255.0 : source == 1 ? 65535.0 : source == 2 ? 4294967295.0 : 18446744073709551615.0;
    *out = f64_as_uint(intensity / smax * tpf * tmax, tmax);
    return 0;
}

static int is_framed_camera(uint32_t cam) { return cam <= 5u; } /* FramedU8..FramedF64 (lib.rs:35-47) */

/* ingest_event_for_chunk (driver.rs:984-1133).  ev->y is already chunk-local.  Returns filled;
 * *grew_out as in the reference. */
static int ingest_event_for_chunk(OracleFramer *f, OracleEvent *ev, ODeque *chunk, uint64_t *running_ts,
                                  int64_t *frame_idx_offset, int64_t *last_filled, uint32_t *last_intensity,
                                  int *grew_out) {
    const uint8_t channel = ev->c == 0xFF ? 0 : ev->c;
    int grew = 0;
    const int64_t prev_last_filled = *last_filled;
    const uint64_t prev_running_ts = *running_ts;
    const int abs_t = f->codec_version >= 2 && f->time_mode == 1;

    if (abs_t) {
        if (prev_running_ts >= (uint64_t)ev->t) {
            if (grew_out) *grew_out = 0;
            return chunk->q[0].filled_count == chunk->px;
        }
        *running_ts = ev->t;
    } else {
        *running_ts += (uint64_t)ev->t;
    }

    const uint64_t rm1 = *running_ts ? *running_ts - 1 : 0; /* saturating_sub(1) */
    if ((int64_t)rm1 / (int64_t)f->tpf > *last_filled) {
        if (ev->d != D_EMPTY) {
            if (abs_t && f->view_mode != 3) { /* :1022-1028 */
                const uint32_t p = (uint32_t)prev_running_ts;
                ev->t = ev->t > p ? ev->t - p : 0u; /* saturating_sub */
            }
            if (f->value_type != 0) {
                if (get_frame_value_wide(f->value_type, ev->d, ev->t, f->source_type, (double)f->ref_interval,
                                         f->practical_d_max, f->source_dtm, f->view_mode, last_intensity) != 0)
                    abort(); /* todo!() in the reference */
            } else if (f->view_mode == 0 && f->source_type == 0)
                *last_intensity = frame_value_u8(ev->d, ev->t, (double)f->ref_interval);
            else
                *last_intensity = get_frame_value_u8(ev->d, ev->t, f->source_type, (double)f->ref_interval,
                                                     f->practical_d_max, f->source_dtm, f->view_mode,
                                                     (uint32_t)*running_ts, (uint32_t)prev_running_ts);
        }
        *last_filled = (int64_t)rm1 / (int64_t)f->tpf;

        const int64_t a = *last_filled - *frame_idx_offset;
        if (a > 0) {
            deque_push_back_empty(chunk, (size_t)a);
            *frame_idx_offset += a;
            grew = 1;
        }
        const size_t off = ((size_t)ev->y * f->w + ev->x) * f->c + channel;
        for (int64_t i = prev_last_filled; i < *last_filled; ++i) {
            const int64_t idx = i - f->frames_written + 1;
            if (idx >= 0) {
                OFrame *fr = &chunk->q[(size_t)idx]; /* an index past the deque panics in the reference */
                if (!fr->some[off]) {
                    fr->some[off] = 1;
                    fr->val[off] = *last_intensity;
                    fr->filled_count += 1;
                }
            }
        }
    }

    /* framed sources: round the pixel's clock up to the next reference interval (:1093-1111) */
    if (f->codec_version >= 1 && is_framed_camera(f->source_camera) && *running_ts % (uint64_t)f->ref_interval > 0)
        *running_ts = ((*running_ts / (uint64_t)f->ref_interval) + 1) * (uint64_t)f->ref_interval;

    if (f->has_buffer_limit) {
        if (*last_filled > f->frames_written + (int64_t)f->buffer_limit) chunk->q[0].filled_count = chunk->px;
    }
    if (chunk->q[0].filled_count > chunk->px) chunk->q[0].filled_count = chunk->px;
    if (grew_out) *grew_out = grew;
    return chunk->q[0].filled_count == chunk->px;
}

static int all_chunks_filled(const OracleFramer *f) {
    for (size_t k = 0; k < f->num_chunks; ++k)
        if (!f->chunk_filled[k]) return 0;
    return 1;
}

/* is_frame_0_filled (driver.rs:828-843) */
int oracle_framer_is_frame_0_filled(const OracleFramer *f) {
    if (f->has_buffer_limit)
        for (size_t k = 0; k < f->num_chunks; ++k)
            if (f->frames[k].len > (size_t)f->buffer_limit) return 1;
    return all_chunks_filled(f);
}

/* Framer::ingest_event (driver.rs:437-562), detect_features = false.  Returns 1 when frame 0 of
 * every chunk is filled. */
int oracle_framer_ingest_event(OracleFramer *f, OracleEvent *ev) {
    const size_t chunk_num = (size_t)ev->y / f->chunk_rows;
    if (chunk_num >= f->num_chunks) return 0; /* silently handle malformed event (:442-444) */
    ev->y = (uint16_t)(ev->y - chunk_num * f->chunk_rows);
    const uint8_t channel = ev->c == 0xFF ? 0 : ev->c;
    const size_t off = ((size_t)ev->y * f->w + ev->x) * f->c + channel;
    const int filled =
        ingest_event_for_chunk(f, ev, &f->frames[chunk_num], &f->pixel_ts[chunk_num][off],
                               &f->frame_idx_offsets[chunk_num], &f->last_filled[chunk_num][off],
                               &f->last_intensity[chunk_num][off], NULL);
    f->chunk_filled[chunk_num] = (uint8_t)filled;
    return all_chunks_filled(f);
}

/* Framer::ingest_events_events (driver.rs:564-626): events[chunk_offsets[k] .. chunk_offsets[k+1])
 * is the k-th inner Vec.  The reference asserts events.len() == num_chunks: the caller passes
 * num_chunks + 1 offsets.  Chunks are independent (rayon there, a plain loop here). */
int oracle_framer_ingest_events_events(OracleFramer *f, OracleEvent *events, const uint64_t *chunk_offsets) {
    for (size_t k = 0; k < f->num_chunks; ++k) {
        for (uint64_t i = chunk_offsets[k]; i < chunk_offsets[k + 1]; ++i) {
            OracleEvent *ev = &events[i];
            const uint8_t channel = ev->c == 0xFF ? 0 : ev->c;
            const size_t chunk_num = (size_t)ev->y / f->chunk_rows;
            ev->y = (uint16_t)(ev->y - chunk_num * f->chunk_rows);
            const size_t off = ((size_t)ev->y * f->w + ev->x) * f->c + channel;
            /* the trackers indexed are those of the k-th zip element (:584-593) */
            f->chunk_filled[k] = (uint8_t)ingest_event_for_chunk(
                f, ev, &f->frames[k], &f->pixel_ts[k][off], &f->frame_idx_offsets[k], &f->last_filled[k][off],
                &f->last_intensity[k][off], NULL);
        }
    }
    return oracle_framer_is_frame_0_filled(f);
}

/* Framer::flush_frame_buffer (driver.rs:632-677) */
int oracle_framer_flush_frame_buffer(OracleFramer *f) {
    int any_nonempty = 0;
    for (size_t k = 0; k < f->num_chunks; ++k)
        if (f->frames[k].len > 1) any_nonempty = 1;
    if (any_nonempty) {
        for (size_t k = 0; k < f->num_chunks; ++k) {
            OFrame *fr = &f->frames[k].q[0];
            for (size_t i = 0; i < f->frames[k].px; ++i) {
                if (!fr->some[i]) {
                    fr->some[i] = 1;
                    fr->val[i] = f->last_intensity[k][i];
                    fr->filled_count += 1;
                    f->last_filled[k][i] += 1;
                }
            }
            f->chunk_filled[k] = 1;
        }
    } else {
        f->chunk_filled[0] = 0;
    }
    return oracle_framer_is_frame_0_filled(f);
}

/* is_frame_filled (driver.rs:807-825): 1 filled, 0 not, -1 InvalidIndex, -2 BadFillCount */
int oracle_framer_is_frame_filled(const OracleFramer *f, size_t frame_idx) {
    for (size_t k = 0; k < f->num_chunks; ++k) {
        if (f->frames[k].len <= frame_idx) return -1;
        const size_t a = f->frames[k].q[frame_idx].filled_count;
        if (a == f->frames[k].px) continue;
        if (a > f->frames[k].px) return -2;
        return 0;
    }
    return 1;
}

/* pop_next_frame_for_chunk (driver.rs:903-925): the popped frame's Option<u8>s into val / some */
static void pop_next_frame_for_chunk(OracleFramer *f, size_t k, OFrame *out) {
    ODeque *d = &f->frames[k];
    *out = d->q[0];
    memmove(d->q, d->q + 1, (d->len - 1) * sizeof(OFrame));
    d->len -= 1;
    if (d->len == 0) {
        deque_push_back_empty(d, 1);
        f->frame_idx_offsets[k] += 1;
    }
    f->chunk_filled[k] = d->q[0].filled_count == d->px;
}

/* write_frame_bytes (driver.rs:935-962): one byte per pixel (bincode u8), None as u8::default().
 * Returns the number of bytes written (= w*h*c). */
size_t oracle_framer_write_frame_bytes(OracleFramer *f, uint8_t *out) {
    size_t n = 0;
    for (size_t k = 0; k < f->num_chunks; ++k) {
        OFrame fr;
        pop_next_frame_for_chunk(f, k, &fr);
        for (size_t i = 0; i < f->frames[k].px; ++i) { /* bincode fixint, big-endian (driver.rs:279,395-398) */
            const uint32_t v = fr.some[i] ? fr.val[i] : 0u;
            if (f->value_type == 2) { out[n++] = (uint8_t)(v >> 24); out[n++] = (uint8_t)(v >> 16); }
            if (f->value_type >= 1) out[n++] = (uint8_t)(v >> 8);
            out[n++] = (uint8_t)v;
        }
        frame_free(&fr);
    }
    f->frames_written += 1;
    return n;
}

/* write_multi_frame_bytes (driver.rs:970-981): frames written, or -1 if is_frame_filled errs or
 * the buffer is too small; *bytes_out = bytes appended. */
int oracle_framer_write_multi_frame_bytes(OracleFramer *f, uint8_t *out, size_t cap, size_t *bytes_out) {
    const size_t frame_bytes = ((size_t)f->w * f->h * f->c) << f->value_type;
    int frames = 0;
    size_t n = 0;
    for (;;) {
        const int st = oracle_framer_is_frame_filled(f, 0);
        if (st < 0) return -1;
        if (!st) break;
        if (n + frame_bytes > cap) return -1;
        n += oracle_framer_write_frame_bytes(f, out + n);
        frames += 1;
    }
    if (bytes_out) *bytes_out = n;
    return frames;
}
