// adder_framer_api.cpp -- C-ABI of the instantaneous framer (include/adder_framer.h).
// Host side: parameter checks (FramerBuilder, driver.rs:55-138), tracker planes and the frame
// ring in HBM, one kernel launch per stream segment, frame hand-out.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <cstdlib>
#include <string>

#include "../../include/adder_framer.h"
#include "adder_framer_kernels.h"

using namespace adder;

static thread_local std::string g_framer_create_error;

constexpr uint32_t kOffsSlots = 4;
constexpr size_t kOffsSlotEntries = (size_t)kFramerRowsMaxFrames + 1u;

struct AdderFramer {
    AdderFramerParams p{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t rows = 0, n_units = 0, tpf = 0, ring_frames = 0;
    FramerPx *px = nullptr;  // per-unit trackers
    uint8_t *ring = nullptr;
    uint32_t *status = nullptr;
    int32_t *minmax = nullptr;  // device {min, max}
    AdderEvent *d_events = nullptr;
    size_t d_events_cap = 0;
    uint8_t *d_out = nullptr;
    size_t d_out_cap = 0;
    uint64_t *d_offs = nullptr;  // device copy of the frame offsets of adder_framer_ingest_frames_device
    size_t d_offs_cap = 0;
    uint64_t *h_offs = nullptr;  // pinned staging of the same: kOffsSlots slots of kOffsSlotEntries, used in turn
    hipEvent_t h_offs_done[4] = {nullptr, nullptr, nullptr, nullptr};  // ... each free again once its copy has run
    bool h_offs_busy[4] = {false, false, false, false};
    uint32_t h_offs_next = 0;
    uint64_t *h_offs_big = nullptr;  // batches of more frames than a slot holds (synchronous path)
    size_t h_offs_cap = 0;
    uint64_t *d_tile_off = nullptr;  // [frames of a launch][tiles + 1] slice offsets (adder_framer_slices_kernel)
    size_t d_tile_off_cap = 0;
    uint32_t window_rows = 8;        // LDS window of the tiles kernel: covers delta_t_max / tpf frames of lag
    hipEvent_t ingested = nullptr;  // recorded behind the last device operation (ingest / pop / flush) on its stream
    bool op_pending = false;
    int64_t frames_written = 0;
    bool flushed_pending = false;
    bool poisoned = false;
    std::string err;
};

static int ffail(AdderFramer *fr, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (fr)
        fr->err = buf;
    else
        g_framer_create_error = buf;
    return code;
}

#define FHIPCHK(fr, expr)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return ffail(fr, ADDER_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                    \
    } while (0)

static void framer_free(AdderFramer *fr) {
    if (!fr) return;
    (void)hipSetDevice(fr->device);
    if (fr->stream) (void)hipStreamSynchronize(fr->stream);
    for (void *p : {(void *)fr->px, (void *)fr->ring, (void *)fr->status,
                    (void *)fr->minmax, (void *)fr->d_events, (void *)fr->d_out, (void *)fr->d_offs, (void *)fr->d_tile_off})
        if (p) (void)hipFree(p);
    if (fr->h_offs) (void)hipHostFree(fr->h_offs);
    if (fr->h_offs_big) (void)hipHostFree(fr->h_offs_big);
    for (hipEvent_t e : fr->h_offs_done)
        if (e) (void)hipEventDestroy(e);
    if (fr->ingested) (void)hipEventDestroy(fr->ingested);
    if (fr->stream) (void)hipStreamDestroy(fr->stream);
    delete fr;
}

extern "C" void adder_framer_default_params(AdderFramerParams *p, uint16_t width, uint16_t height, uint8_t channels) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->abi_version = ADDER_FRAMER_ABI_VERSION;
    p->width = width;
    p->height = height;
    p->channels = channels;
    p->codec_version = 3;
    p->time_mode = ADDER_TIME_ABSOLUTE_T;
    p->row_begin = 0;
    p->row_end = height;
    p->ref_interval = 255;
    p->tps = 255 * 30;
    p->delta_t_max = 255 * 30;
    p->output_fps = 0.0f;
    p->source_camera = 0;  // FramedU8
    p->device_id = 0;
}

extern "C" int adder_framer_create(const AdderFramerParams *pp, AdderFramer **out) {
    if (out) *out = nullptr;
    if (!pp || !out) return ffail(nullptr, ADDER_E_BAD_PARAMS, "null argument");
    const AdderFramerParams p = *pp;
    if (p.abi_version != ADDER_FRAMER_ABI_VERSION) return ffail(nullptr, ADDER_E_BAD_PARAMS, "abi_version mismatch");
    if (!p.width || !p.height || (p.channels != 1 && p.channels != 3))
        return ffail(nullptr, ADDER_E_BAD_PARAMS, "bad plane %ux%ux%u", p.width, p.height, p.channels);
    if (p.row_begin >= p.row_end || p.row_end > p.height) return ffail(nullptr, ADDER_E_BAD_PARAMS, "bad row band");
    if (!p.ref_interval || !p.tps) return ffail(nullptr, ADDER_E_BAD_PARAMS, "tps and ref_interval must be non-zero");
    if (p.time_mode > ADDER_TIME_MIXED) return ffail(nullptr, ADDER_E_BAD_PARAMS, "bad time_mode");
    if (p.view_mode > ADDER_VIEW_SAE || p.source_type > 3) return ffail(nullptr, ADDER_E_BAD_PARAMS, "bad view_mode / source_type");
    // FrameSequence<T>: T = u8 / u16 / u32.  (u64 cannot be instantiated in the reference -- its methods need T: Into<f64>
    // -- and the SAE view of the wider types is todo!() there, scale_intensity.rs:154,203.)
    if (p.value_type > ADDER_FRAME_U32) return ffail(nullptr, ADDER_E_BAD_PARAMS, "value_type must be ADDER_FRAME_U8 / _U16 / _U32");
    if (p.value_type != ADDER_FRAME_U8 && p.view_mode == ADDER_VIEW_SAE)
        return ffail(nullptr, ADDER_E_BAD_PARAMS, "the SAE view exists for u8 frames only (todo!() in the reference for u16 / u32)");
    // FrameSequence::new (driver.rs:357-361)
    uint32_t tpf = p.ref_interval;
    if (p.output_fps > 0.0f) {
        const float q = (float)p.tps / p.output_fps;
        tpf = q >= 4294967296.0f ? 0xffffffffu : (uint32_t)q;
    }
    if (!tpf) return ffail(nullptr, ADDER_E_BAD_PARAMS, "ticks per output frame is zero");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ffail(nullptr, ADDER_E_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (p.device_id < 0 || p.device_id >= ndev) return ffail(nullptr, ADDER_E_BAD_PARAMS, "bad device_id");
    AdderFramer *fr = new (std::nothrow) AdderFramer();
    if (!fr) return ffail(nullptr, ADDER_E_HIP, "out of host memory");
    fr->p = p;
    fr->device = p.device_id;
    fr->rows = p.row_end - p.row_begin;
    const uint64_t units = (uint64_t)fr->rows * p.width * p.channels;
    if (units > 0x7fffffffull) {
        delete fr;
        return ffail(nullptr, ADDER_E_BAD_PARAMS, "row band too large");
    }
    fr->n_units = (uint32_t)units;
    fr->tpf = tpf;
    {   // a pixel lags the newest frame by at most delta_t_max ticks: the window should hold that many rows
        // (DeltaT clocks are per-pixel sums of event times and drift apart: a wider window for those streams)
        const uint64_t lag = (uint64_t)p.delta_t_max / tpf + (p.time_mode == ADDER_TIME_ABSOLUTE_T ? 4u : 24u);
        uint32_t k = 16;
        while (k < 32u && k < lag) k <<= 1;  // (64 rows cost more in occupancy than the stragglers they save)
        if (const char *e = getenv("ADDER_FRAMER_WINDOW")) k = (uint32_t)atoi(e);  // 8 / 16 / 32 / 64 (tuning)
        fr->window_rows = k;
    }
    fr->ring_frames = p.ring_frames ? p.ring_frames : (uint32_t)std::min<uint64_t>(p.delta_t_max / tpf + 80u, 1u << 20);
    auto setup = [&]() -> int {
        FHIPCHK(fr, hipSetDevice(fr->device));
        FHIPCHK(fr, hipStreamCreateWithFlags(&fr->stream, hipStreamNonBlocking));
        FHIPCHK(fr, hipEventCreateWithFlags(&fr->ingested, hipEventDisableTiming));
        FHIPCHK(fr, hipMalloc(reinterpret_cast<void **>(&fr->px), (size_t)fr->n_units * sizeof(FramerPx)));
        FHIPCHK(fr, hipMalloc(reinterpret_cast<void **>(&fr->ring), ((size_t)fr->ring_frames * fr->n_units) << p.value_type));
        FHIPCHK(fr, hipMalloc(reinterpret_cast<void **>(&fr->status), sizeof(uint32_t)));
        FHIPCHK(fr, hipMalloc(reinterpret_cast<void **>(&fr->minmax), 2 * sizeof(int32_t)));
        // everything a batch of <= 64 frames needs exists from here on (longer ones grow the slice table once): an
        // ingest call neither allocates nor waits for the device
        FHIPCHK(fr, hipMalloc(reinterpret_cast<void **>(&fr->d_offs), kOffsSlotEntries * sizeof(uint64_t)));
        fr->d_offs_cap = kOffsSlotEntries * sizeof(uint64_t);
        fr->d_tile_off_cap = (size_t)std::min(kFramerRowsMaxFrames, 64u) * (adder_framer_num_tiles(fr->n_units) + 1u) * sizeof(uint64_t);
        FHIPCHK(fr, hipMalloc(reinterpret_cast<void **>(&fr->d_tile_off), fr->d_tile_off_cap));
        FHIPCHK(fr, hipHostMalloc(reinterpret_cast<void **>(&fr->h_offs), kOffsSlots * kOffsSlotEntries * sizeof(uint64_t),
                                  hipHostMallocDefault));
        for (hipEvent_t &e : fr->h_offs_done) FHIPCHK(fr, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        FHIPCHK(fr, hipMemsetAsync(fr->status, 0, sizeof(uint32_t), fr->stream));
        FHIPCHK(fr, hipMemsetAsync(fr->ring, 0, ((size_t)fr->ring_frames * fr->n_units) << p.value_type, fr->stream));
        FHIPCHK(fr, adder_framer_launch_init(fr->px, fr->n_units, fr->stream));
        FHIPCHK(fr, hipStreamSynchronize(fr->stream));
        return ADDER_OK;
    };
    const int rc = setup();
    if (rc != ADDER_OK) {
        g_framer_create_error = fr->err;
        framer_free(fr);
        return rc;
    }
    *out = fr;
    return ADDER_OK;
}

extern "C" void adder_framer_destroy(AdderFramer *fr) { framer_free(fr); }
extern "C" const char *adder_framer_last_error(const AdderFramer *fr) {
    return fr ? fr->err.c_str() : g_framer_create_error.c_str();
}
extern "C" uint32_t adder_framer_tpf(const AdderFramer *fr) { return fr ? fr->tpf : 0; }
extern "C" int64_t adder_framer_frames_written(const AdderFramer *fr) { return fr ? fr->frames_written : 0; }

template <class T>
static int fensure(AdderFramer *fr, T **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return ADDER_OK;
    if (*p) FHIPCHK(fr, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    FHIPCHK(fr, hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(need, 16)));
    *cap = need;
    return ADDER_OK;
}

static FramerArgs make_args(const AdderFramer *fr) {
    FramerArgs a{};
    a.px = fr->px;
    a.ring = fr->ring;
    a.status = fr->status;
    a.n_units = fr->n_units;
    a.width = fr->p.width;
    a.channels = fr->p.channels;
    a.row_begin = fr->p.row_begin;
    a.rows = fr->rows;
    a.ring_frames = fr->ring_frames;
    a.by_ring = fast_div_make(fr->ring_frames);
    a.frames_written = (int32_t)fr->frames_written;
    a.k = framer_consts(fr->tpf, fr->p.ref_interval,
                        (fr->p.codec_version >= 2 && fr->p.time_mode == ADDER_TIME_ABSOLUTE_T) ? 1u : 0u,
                        (fr->p.codec_version >= 1 && fr->p.source_camera <= 5u) ? 1u : 0u, fr->p.view_mode,
                        fr->p.source_type, fr->p.practical_d_max, fr->p.delta_t_max, fr->p.value_type);
    return a;
}

// Operations may be queued on different streams (the caller's, the context's): each one is ordered
// behind the previous one through this event.
static int after_last_op(AdderFramer *fr, hipStream_t s) {
    if (fr->op_pending) FHIPCHK(fr, hipStreamWaitEvent(s, fr->ingested, 0));
    return ADDER_OK;
}
static int mark_op(AdderFramer *fr, hipStream_t s) {
    FHIPCHK(fr, hipEventRecord(fr->ingested, s));
    fr->op_pending = true;
    return ADDER_OK;
}

static int check_status(AdderFramer *fr) {
    {
        const int rc_ = after_last_op(fr, fr->stream);
        if (rc_ != ADDER_OK) return rc_;
    }
    uint32_t st = 0;
    FHIPCHK(fr, hipMemcpyAsync(&st, fr->status, sizeof st, hipMemcpyDeviceToHost, fr->stream));
    FHIPCHK(fr, hipStreamSynchronize(fr->stream));
    if (!st) return ADDER_OK;
    fr->poisoned = true;
    if (st & kFramerStatusMalformed) return ffail(fr, ADDER_E_BAD_PARAMS, "an event lies outside the plane / row band");
    if (st & kFramerStatusRange) return ffail(fr, ADDER_E_BAD_PARAMS, "frame index out of range");
    return ffail(fr, ADDER_E_OUT_CAPACITY,
                 "the frame ring (%u frames) is too small for this stream: raise AdderFramerParams.ring_frames or pop "
                 "complete frames more often",
                 fr->ring_frames);
}

extern "C" int adder_framer_ingest_device(AdderFramer *fr, const AdderEvent *d_events, const uint64_t *seg_offsets,
                                          uint32_t num_segments, void *stream) {
    if (!fr) return ADDER_E_BAD_PARAMS;
    if (fr->poisoned) return ffail(fr, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", fr->err.c_str());
    if (fr->flushed_pending) return ffail(fr, ADDER_E_BAD_PARAMS, "pop the flushed frame before ingesting more events");
    if (num_segments && (!seg_offsets || (!d_events && seg_offsets[num_segments] > seg_offsets[0])))
        return ffail(fr, ADDER_E_BAD_PARAMS, "null argument");
    FHIPCHK(fr, hipSetDevice(fr->device));
    {
        const int rc_ = after_last_op(fr, (hipStream_t)stream);
        if (rc_ != ADDER_OK) return rc_;
    }
    const FramerArgs a = make_args(fr);
    for (uint32_t s = 0; s < num_segments; ++s) {
        if (seg_offsets[s + 1] < seg_offsets[s]) return ffail(fr, ADDER_E_BAD_PARAMS, "segment offsets must not decrease");
        FHIPCHK(fr, adder_framer_launch_segment(d_events, seg_offsets[s], seg_offsets[s + 1], &a, (hipStream_t)stream));
    }
    return mark_op(fr, (hipStream_t)stream);
}

static int ingest_frames_launch(AdderFramer *fr, const AdderEvent *d_events, const uint64_t *d_offs, uint32_t num_frames,
                                hipStream_t s);

extern "C" int adder_framer_ingest_frames_device(AdderFramer *fr, const AdderEvent *d_events,
                                                 const uint64_t *frame_offsets, uint32_t num_frames, void *stream) {
    if (!fr) return ADDER_E_BAD_PARAMS;
    if (fr->poisoned) return ffail(fr, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", fr->err.c_str());
    if (fr->flushed_pending) return ffail(fr, ADDER_E_BAD_PARAMS, "pop the flushed frame before ingesting more events");
    if (!num_frames) return ADDER_OK;
    if (!frame_offsets || (!d_events && frame_offsets[num_frames] > frame_offsets[0]))
        return ffail(fr, ADDER_E_BAD_PARAMS, "null argument");
    for (uint32_t s = 0; s < num_frames; ++s)
        if (frame_offsets[s + 1] < frame_offsets[s]) return ffail(fr, ADDER_E_BAD_PARAMS, "frame offsets must not decrease");
    // u16 / u32 frames: the per-event kernel (the tile kernel's LDS window holds bytes)
    if (fr->p.value_type != ADDER_FRAME_U8) return adder_framer_ingest_device(fr, d_events, frame_offsets, num_frames, stream);
    FHIPCHK(fr, hipSetDevice(fr->device));
    hipStream_t s = (hipStream_t)stream;
    {
        const int rc_ = after_last_op(fr, s);
        if (rc_ != ADDER_OK) return rc_;
    }
    const size_t bytes = ((size_t)num_frames + 1) * sizeof(uint64_t);
    int rc = fensure(fr, &fr->d_offs, &fr->d_offs_cap, bytes);
    if (rc != ADDER_OK) return rc;
    if (num_frames + 1u <= kOffsSlotEntries) {
        // pinned staging slots in turn: a slot is reused four calls later, when its copy has long run
        const uint32_t slot = fr->h_offs_next;
        fr->h_offs_next = (slot + 1u) % kOffsSlots;
        if (fr->h_offs_busy[slot]) FHIPCHK(fr, hipEventSynchronize(fr->h_offs_done[slot]));
        uint64_t *stage = fr->h_offs + (size_t)slot * kOffsSlotEntries;
        memcpy(stage, frame_offsets, bytes);
        FHIPCHK(fr, hipMemcpyAsync(fr->d_offs, stage, bytes, hipMemcpyHostToDevice, s));
        FHIPCHK(fr, hipEventRecord(fr->h_offs_done[slot], s));
        fr->h_offs_busy[slot] = true;
    } else {
        FHIPCHK(fr, hipStreamSynchronize(s));  // the big staging buffer may still feed an earlier copy
        if (fr->h_offs_cap < bytes) {
            if (fr->h_offs_big) FHIPCHK(fr, hipHostFree(fr->h_offs_big));
            fr->h_offs_big = nullptr;
            fr->h_offs_cap = 0;
            FHIPCHK(fr, hipHostMalloc(reinterpret_cast<void **>(&fr->h_offs_big), bytes, hipHostMallocDefault));
            fr->h_offs_cap = bytes;
        }
        memcpy(fr->h_offs_big, frame_offsets, bytes);
        FHIPCHK(fr, hipMemcpyAsync(fr->d_offs, fr->h_offs_big, bytes, hipMemcpyHostToDevice, s));
    }
    return ingest_frames_launch(fr, d_events, fr->d_offs, num_frames, s);
}

// frame_offsets in device memory (what adder_hip_integrate_device leaves there): nothing crosses the bus, the
// transcoder's batch and its framing queue back to back.  The offsets are checked on the device.
extern "C" int adder_framer_ingest_frames_device_offsets(AdderFramer *fr, const AdderEvent *d_events,
                                                         const uint64_t *d_frame_offsets, uint32_t num_frames,
                                                         void *stream) {
    if (!fr) return ADDER_E_BAD_PARAMS;
    if (fr->poisoned) return ffail(fr, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", fr->err.c_str());
    if (fr->flushed_pending) return ffail(fr, ADDER_E_BAD_PARAMS, "pop the flushed frame before ingesting more events");
    if (!num_frames) return ADDER_OK;
    if (!d_frame_offsets || !d_events) return ffail(fr, ADDER_E_BAD_PARAMS, "null argument");
    if (fr->p.value_type != ADDER_FRAME_U8)
        return ffail(fr, ADDER_E_BAD_PARAMS, "u16 / u32 frames take their segment offsets from host memory "
                     "(adder_framer_ingest_frames_device / adder_framer_ingest_device)");
    FHIPCHK(fr, hipSetDevice(fr->device));
    hipStream_t s = (hipStream_t)stream;
    const int rc = after_last_op(fr, s);
    if (rc != ADDER_OK) return rc;
    return ingest_frames_launch(fr, d_events, d_frame_offsets, num_frames, s);
}

static int ingest_frames_launch(AdderFramer *fr, const AdderEvent *d_events, const uint64_t *d_offs, uint32_t num_frames,
                                hipStream_t s) {
    int rc;
    const FramerArgs a = make_args(fr);
    const uint32_t per_launch = std::min(kFramerRowsMaxFrames, num_frames);
    rc = fensure(fr, &fr->d_tile_off, &fr->d_tile_off_cap,
                 (size_t)per_launch * (adder_framer_num_tiles(fr->n_units) + 1u) * sizeof(uint64_t));
    if (rc != ADDER_OK) return rc;
    for (uint32_t f0 = 0; f0 < num_frames; f0 += kFramerRowsMaxFrames) {
        const uint32_t nf = std::min(kFramerRowsMaxFrames, num_frames - f0);
        FHIPCHK(fr, adder_framer_launch_tiles(d_events, d_offs + f0, nf, fr->d_tile_off, fr->window_rows, &a, s));
    }
    return mark_op(fr, s);
}

extern "C" int adder_framer_ingest(AdderFramer *fr, const AdderEvent *events, const uint64_t *seg_offsets,
                                   uint32_t num_segments) {
    if (!fr) return ADDER_E_BAD_PARAMS;
    if (!num_segments) return ADDER_OK;
    if (!seg_offsets) return ffail(fr, ADDER_E_BAD_PARAMS, "null argument");
    const uint64_t e0 = seg_offsets[0], e1 = seg_offsets[num_segments];
    if (e1 < e0 || (!events && e1 > e0)) return ffail(fr, ADDER_E_BAD_PARAMS, "bad segment offsets");
    FHIPCHK(fr, hipSetDevice(fr->device));
    int rc = fensure(fr, &fr->d_events, &fr->d_events_cap, (size_t)(e1 - e0) * sizeof(AdderEvent));
    if (rc != ADDER_OK) return rc;
    if (e1 > e0)
        FHIPCHK(fr, hipMemcpyAsync(fr->d_events, events + e0, (size_t)(e1 - e0) * sizeof(AdderEvent),
                                   hipMemcpyHostToDevice, fr->stream));
    // the device copy starts at event e0: shift the offsets
    std::string keep;
    uint64_t small[64];
    uint64_t *offs = num_segments + 1 <= 64 ? small : new (std::nothrow) uint64_t[num_segments + 1];
    if (!offs) return ffail(fr, ADDER_E_HIP, "out of host memory");
    for (uint32_t s = 0; s <= num_segments; ++s) offs[s] = seg_offsets[s] - e0;
    rc = adder_framer_ingest_device(fr, fr->d_events, offs, num_segments, fr->stream);
    if (offs != small) delete[] offs;
    if (rc != ADDER_OK) return rc;
    return check_status(fr);
}

// complete frames = min(last_filled) + 1 - frames_written; also the max (flush needs it)
static int minmax_sync(AdderFramer *fr, int32_t *mn, int32_t *mx, hipStream_t s) {
    const int32_t init[2] = {0x7fffffff, -0x7fffffff - 1};
    FHIPCHK(fr, hipMemcpyAsync(fr->minmax, init, sizeof init, hipMemcpyHostToDevice, s));
    FHIPCHK(fr, adder_framer_launch_minmax(fr->px, fr->n_units, fr->minmax, s));
    int32_t h[2];
    FHIPCHK(fr, hipMemcpyAsync(h, fr->minmax, sizeof h, hipMemcpyDeviceToHost, s));
    FHIPCHK(fr, hipStreamSynchronize(s));
    *mn = h[0];
    *mx = h[1];
    return ADDER_OK;
}

static int ready_count(AdderFramer *fr, uint32_t *n, hipStream_t s) {
    *n = 0;
    if (fr->flushed_pending) {  // flush made frame 0 complete (driver.rs:665-667)
        *n = 1;
        return ADDER_OK;
    }
    int32_t mn, mx;
    int rc = after_last_op(fr, s);
    if (rc != ADDER_OK) return rc;
    rc = minmax_sync(fr, &mn, &mx, s);
    if (rc != ADDER_OK) return rc;
    const int64_t r = (int64_t)mn + 1 - fr->frames_written;
    *n = r > 0 ? (uint32_t)std::min<int64_t>(r, fr->ring_frames) : 0u;
    return ADDER_OK;
}

extern "C" int adder_framer_frames_ready(AdderFramer *fr, uint32_t *n_ready) {
    if (!fr || !n_ready) return ADDER_E_BAD_PARAMS;
    if (fr->poisoned) return ffail(fr, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", fr->err.c_str());
    FHIPCHK(fr, hipSetDevice(fr->device));
    const int rc = check_status(fr);
    if (rc != ADDER_OK) return rc;
    return ready_count(fr, n_ready, fr->stream);
}

extern "C" int adder_framer_pop_device(AdderFramer *fr, uint8_t *d_out, uint32_t max_frames, uint32_t *n_popped,
                                       void *stream) {
    if (!fr || !n_popped) return ADDER_E_BAD_PARAMS;
    *n_popped = 0;
    if (fr->poisoned) return ffail(fr, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", fr->err.c_str());
    if (!d_out && max_frames) return ffail(fr, ADDER_E_BAD_PARAMS, "null output");
    FHIPCHK(fr, hipSetDevice(fr->device));
    hipStream_t s = (hipStream_t)stream;
    uint32_t ready = 0;
    const int rc = ready_count(fr, &ready, s);
    if (rc != ADDER_OK) return rc;
    const uint32_t n = std::min(ready, max_frames);
    if (!n) return ADDER_OK;
    FHIPCHK(fr, adder_framer_launch_pop(fr->ring, fr->px, fr->n_units, fr->ring_frames, (int32_t)fr->frames_written, n,
                                        0u, d_out, fr->p.value_type, s));
    {
        const int rc_ = mark_op(fr, s);
        if (rc_ != ADDER_OK) return rc_;
    }
    fr->frames_written += n;
    fr->flushed_pending = false;
    *n_popped = n;
    return ADDER_OK;
}

extern "C" int adder_framer_pop(AdderFramer *fr, uint8_t *out, uint32_t max_frames, uint32_t *n_popped) {
    if (!fr || !n_popped) return ADDER_E_BAD_PARAMS;
    *n_popped = 0;
    if (!out && max_frames) return ffail(fr, ADDER_E_BAD_PARAMS, "null output");
    FHIPCHK(fr, hipSetDevice(fr->device));
    int rc = check_status(fr);
    if (rc != ADDER_OK) return rc;
    uint32_t ready = 0;
    rc = ready_count(fr, &ready, fr->stream);
    if (rc != ADDER_OK) return rc;
    const uint32_t want = std::min(ready, max_frames);
    if (!want) return ADDER_OK;
    rc = fensure(fr, &fr->d_out, &fr->d_out_cap, ((size_t)want * fr->n_units) << fr->p.value_type);
    if (rc != ADDER_OK) return rc;
    uint32_t n = 0;
    rc = adder_framer_pop_device(fr, fr->d_out, want, &n, fr->stream);
    if (rc != ADDER_OK) return rc;
    FHIPCHK(fr, hipMemcpyAsync(out, fr->d_out, ((size_t)n * fr->n_units) << fr->p.value_type, hipMemcpyDeviceToHost, fr->stream));
    FHIPCHK(fr, hipStreamSynchronize(fr->stream));
    *n_popped = n;
    return ADDER_OK;
}

extern "C" int adder_framer_write_frame(AdderFramer *fr, uint8_t *out) {
    if (!fr || !out) return ADDER_E_BAD_PARAMS;
    if (fr->poisoned) return ffail(fr, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", fr->err.c_str());
    FHIPCHK(fr, hipSetDevice(fr->device));
    int rc = check_status(fr);
    if (rc != ADDER_OK) return rc;
    rc = fensure(fr, &fr->d_out, &fr->d_out_cap, (size_t)fr->n_units << fr->p.value_type);
    if (rc != ADDER_OK) return rc;
    // frame 0 as it is: pixels that have no value yet read 0 (driver.rs:946-950); after a flush
    // every pixel has one
    FHIPCHK(fr, adder_framer_launch_pop(fr->ring, fr->px, fr->n_units, fr->ring_frames, (int32_t)fr->frames_written, 1u,
                                        fr->flushed_pending ? 0u : 1u, fr->d_out, fr->p.value_type, fr->stream));
    FHIPCHK(fr, hipMemcpyAsync(out, fr->d_out, (size_t)fr->n_units << fr->p.value_type, hipMemcpyDeviceToHost, fr->stream));
    FHIPCHK(fr, hipStreamSynchronize(fr->stream));
    fr->frames_written += 1;
    fr->flushed_pending = false;
    return ADDER_OK;
}

extern "C" int adder_framer_flush(AdderFramer *fr, int *frame0_ready) {
    if (!fr) return ADDER_E_BAD_PARAMS;
    if (frame0_ready) *frame0_ready = 0;
    if (fr->poisoned) return ffail(fr, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", fr->err.c_str());
    FHIPCHK(fr, hipSetDevice(fr->device));
    int rc = check_status(fr);
    if (rc != ADDER_OK) return rc;
    if (fr->flushed_pending) {
        if (frame0_ready) *frame0_ready = 1;
        return ADDER_OK;
    }
    int32_t mn, mx;
    rc = minmax_sync(fr, &mn, &mx, fr->stream);
    if (rc != ADDER_OK) return rc;
    // `any chunk.len() > 1` (driver.rs:635-639): some pixel has reached beyond frame 0
    if ((int64_t)mx > fr->frames_written) {
        FHIPCHK(fr, adder_framer_launch_flush(fr->ring, fr->px, fr->n_units, fr->ring_frames,
                                              (int32_t)fr->frames_written, fr->p.value_type, fr->stream));
        FHIPCHK(fr, hipStreamSynchronize(fr->stream));
        fr->flushed_pending = true;
        if (frame0_ready) *frame0_ready = 1;
    }
    return ADDER_OK;
}
