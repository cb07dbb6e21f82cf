// adder_host.cpp -- see adder_host.hpp.  Reference file:line citations are in the header.
#include "adder_host.hpp"

#include <cmath>

#include "../../include/adder_compressed.h"

#include <chrono>

#include <string.h>

#include <algorithm>

namespace adder_host {

// ---------------------------------------------------------------- PlaneSize
PlaneSize::PlaneSize(uint16_t width, uint16_t height, uint8_t channels)
    : width_(width), height_(height), channels_(channels) {
    if (width == 0 || height == 0 || channels == 0)  // PlaneError::InvalidPlane (lib.rs:103-110)
        throw SourceError(SourceError::BadParams, "invalid plane " + std::to_string(width) + "x" +
                                                      std::to_string(height) + "x" + std::to_string(channels));
}

// ---------------------------------------------------------------- Crf
// baseline C, max C, C increase velocity, feature radius (x min resolution)
const float CRF[10][4] = {
    {0.0f, 0.0f, 10.0f, 1E-9f},         {0.0f, 1.0f, 9.0f, 1.0f / 12.0f},  {1.0f, 3.0f, 8.0f, 1.0f / 14.0f},
    {2.0f, 7.0f, 7.0f, 1.0f / 15.0f},   {5.0f, 9.0f, 6.0f, 1.0f / 18.0f},  {6.0f, 10.0f, 5.0f, 1.0f / 20.0f},
    {7.0f, 13.0f, 4.0f, 1.0f / 25.0f},  {8.0f, 16.0f, 3.0f, 1.0f / 30.0f}, {10.0f, 20.0f, 2.0f, 1.0f / 30.0f},
    {15.0f, 25.0f, 1.0f, 1.0f / 30.0f},
};

static CrfParameters crf_row(uint8_t q, PlaneSize plane) {
    if (q > 9) throw SourceError(SourceError::BadParams, "crf must be in 0..9");
    CrfParameters p;
    p.c_thresh_baseline = (uint8_t)CRF[q][0];
    p.c_thresh_max = (uint8_t)CRF[q][1];
    p.c_increase_velocity = (uint8_t)CRF[q][2];
    p.feature_c_radius = (uint16_t)(CRF[q][3] * (float)plane.min_resolution());
    return p;
}

Crf::Crf(std::optional<uint8_t> crf, PlaneSize plane_)
    : plane(plane_), quality_(crf), parameters_(crf_row(crf.value_or(DEFAULT_CRF_QUALITY), plane_)) {}

void Crf::update_quality(uint8_t crf) {
    parameters_ = crf_row(crf, plane);
    quality_ = crf;
}

// ---------------------------------------------------------------- Encoder
Encoder::Encoder(CodecMetadata meta, std::ostream *w, EncoderOptions o, EncoderType t)
    : options(o), meta_(meta), writer_(w), type_(t) {
    // EncoderState::default(): last_event_ts = Instant::now() (encoder.rs:45-53)
    last_event_ts_ = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Encoder Encoder::new_raw(CodecMetadata meta, std::ostream *writer, EncoderOptions options) {
    if (!writer) throw CodecError(CodecError::Io, "raw encoder needs a writer");
    meta.event_size = meta.plane.c() == 1 ? 9 : 11;  // RawOutput::new (raw/stream.rs:33-46)
    Encoder e(meta, writer, options, EncoderType::Raw);
    e.encode_header();
    return e;
}

Encoder Encoder::new_compressed(CodecMetadata meta, std::ostream *writer, EncoderOptions options) {
    if (!writer) throw CodecError(CodecError::Io, "compressed encoder needs a writer");
    meta.event_size = meta.plane.c() == 1 ? 9 : 11;  // EventStreamHeader::new (header.rs:75-81)
    Encoder e(meta, writer, options, EncoderType::Compressed);
    AdderCompressedParams p;
    adder_compressed_default_params(&p, meta.plane.w(), meta.plane.h(), meta.plane.c());
    p.codec_version = meta.codec_version;
    p.time_mode = (uint8_t)meta.time_mode;
    p.tps = meta.tps;
    p.ref_interval = meta.ref_interval;
    p.delta_t_max = meta.delta_t_max;
    p.adu_interval = (uint32_t)meta.adu_interval;
    p.source_camera = (uint32_t)meta.source_camera;
    p.c_thresh_max = options.crf.get_parameters().c_thresh_max;  // CompressedOutput::with_options (stream.rs:169-171)
    if (adder_compressed_encoder_create(&p, &e.compressed_) != ADDER_OK)
        throw CodecError(CodecError::BadFile, std::string("compressed encoder: ") + adder_compressed_last_error(nullptr));
    e.meta_.header_size = 25 + 4 * (size_t)std::min<int>(meta.codec_version, 3);
    return e;
}

Encoder::~Encoder() {
    if (compressed_) adder_compressed_encoder_destroy(compressed_);
}

Encoder::Encoder(Encoder &&o) noexcept
    : options(o.options), clock(std::move(o.clock)), meta_(o.meta_), writer_(o.writer_), type_(o.type_),
      scratch_(std::move(o.scratch_)), current_event_rate_(o.current_event_rate_), last_event_ts_(o.last_event_ts_),
      queue_(std::move(o.queue_)), compressed_(o.compressed_) {
    o.compressed_ = nullptr;
    o.writer_ = nullptr;
}

Encoder Encoder::new_empty(CodecMetadata meta, EncoderOptions options) {
    Encoder e(meta, nullptr, options, EncoderType::Empty);
    e.encode_header();
    return e;
}

void Encoder::encode_header() {
    uint8_t buf[64];
    const size_t n = adder_raw_header(buf, meta_.codec_version, meta_.plane.w(), meta_.plane.h(), meta_.plane.c(),
                                      meta_.tps, meta_.ref_interval, meta_.delta_t_max,
                                      (uint32_t)meta_.source_camera, (uint32_t)meta_.time_mode,
                                      (uint32_t)meta_.adu_interval);
    if (meta_.codec_version > LATEST_CODEC_VERSION) throw CodecError(CodecError::BadFile, "bad codec version");
    if (writer_) writer_->write(reinterpret_cast<const char *>(buf), (std::streamsize)n);
    meta_.header_size = n;
}

void Encoder::output_events(const Event *events, size_t n) {
    if (compressed_) {  // CompressedOutput::ingest_event (compressed/stream.rs:268-319)
        if (adder_compressed_encoder_ingest(compressed_, events, n) != ADDER_OK)
            throw CodecError(CodecError::Io, std::string("compressed encoder: ") + adder_compressed_last_error(compressed_));
        return;
    }
    if (!writer_ || n == 0) return;  // EmptyOutput swallows events
    scratch_.resize(n * 11);
    const size_t bytes = adder_raw_events(scratch_.data(), events, n, meta_.plane.c());
    writer_->write(reinterpret_cast<const char *>(scratch_.data()), (std::streamsize)bytes);
    if (!*writer_) throw CodecError(CodecError::Io, "write failed");
}

// std::collections::BinaryHeap, restated so that ties on t pop in the reference's order: `Ord for Event`
// compares other.t with self.t (lib.rs:424-430), i.e. a > b  <=>  a.t < b.t.
//   push = Vec::push + sift_up(0, old_len): the hole climbs while element > parent;
//   pop  = swap the last element into the root, sift_down_to_bottom (always towards the greater child,
//          the RIGHT one when left <= right), then sift_up from there.
static inline bool ord_le(const Event &a, const Event &b) { return a.t >= b.t; }  // a <= b in `Ord for Event`
void Encoder::heap_push(const Event &e) {
    queue_.push_back(e);
    size_t pos = queue_.size() - 1;
    const Event elem = queue_[pos];
    while (pos > 0) {
        const size_t parent = (pos - 1) / 2;
        if (ord_le(elem, queue_[parent])) break;
        queue_[pos] = queue_[parent];
        pos = parent;
    }
    queue_[pos] = elem;
}
Event Encoder::heap_pop() {
    Event item = queue_.back();
    queue_.pop_back();
    if (queue_.empty()) return item;
    std::swap(item, queue_[0]);
    const size_t end = queue_.size();
    const Event elem = queue_[0];
    size_t pos = 0, child = 1;
    while (child <= (end >= 2 ? end - 2 : 0)) {
        if (ord_le(queue_[child], queue_[child + 1])) child += 1;
        queue_[pos] = queue_[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) {
        queue_[pos] = queue_[child];
        pos = child;
    }
    // sift_up(0, pos)
    while (pos > 0) {
        const size_t parent = (pos - 1) / 2;
        if (ord_le(elem, queue_[parent])) break;
        queue_[pos] = queue_[parent];
        pos = parent;
    }
    queue_[pos] = elem;
    return item;
}

void Encoder::ingest_event(const Event &e) {
    if (options.event_drop.kind == EventDrop::Manual) {  // encoder.rs:237-250
        const double now = clock ? clock()
                                 : std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const double t_diff = now - last_event_ts_;
        const double alpha = options.event_drop.alpha;
        const double new_event_rate = alpha * current_event_rate_ + (1.0 - alpha) / t_diff;
        if (new_event_rate > options.event_drop.target_event_rate) {
            current_event_rate_ *= alpha;
            return;  // skip this event
        }
        last_event_ts_ = now;
        current_event_rate_ = new_event_rate;
    }
    if (options.event_order == EventOrder::Unchanged) {
        output_events(&e, 1);
        return;
    }
    // EventOrder::Interleaved (:257-270): push, then release the earliest event once it is more than
    // delta_t_max older than the one just pushed.  At most ONE event leaves per ingest, and whatever is
    // still queued when the writer is closed is never written (close_writer, :152-157).
    const uint32_t dt = e.t;
    heap_push(e);
    const uint32_t horizon = dt > meta_.delta_t_max ? dt - meta_.delta_t_max : 0u;  // saturating_sub
    if (queue_[0].t < horizon) {
        const Event first = heap_pop();
        output_events(&first, 1);
    }
}

void Encoder::ingest_events(const Event *events, size_t n) {
    if (options.event_drop.kind == EventDrop::None && options.event_order == EventOrder::Unchanged) {
        output_events(events, n);  // the default path: one serialisation call for the whole slice
        return;
    }
    for (size_t i = 0; i < n; ++i) ingest_event(events[i]);
}

void Encoder::ingest_events_events(const std::vector<std::vector<Event>> &v) {
    for (const auto &chunk : v) ingest_events(chunk.data(), chunk.size());
}

std::ostream *Encoder::close_writer() {
    if (compressed_) {  // CompressedOutput::into_writer (compressed/stream.rs:179-262): the partial last ADU, no EOF event
        const uint8_t *bytes = nullptr;
        size_t n = 0;
        if (adder_compressed_encoder_close(compressed_, &bytes, &n) != ADDER_OK)
            throw CodecError(CodecError::Io, std::string("compressed encoder: ") + adder_compressed_last_error(compressed_));
        writer_->write(reinterpret_cast<const char *>(bytes), (std::streamsize)n);
        writer_->flush();
        adder_compressed_encoder_destroy(compressed_);
        compressed_ = nullptr;
        std::ostream *w = writer_;
        writer_ = nullptr;
        return w;
    }
    if (!writer_) return nullptr;  // EmptyOutput::into_writer -> None
    uint8_t eof[16];
    const size_t n = adder_raw_eof(eof);
    writer_->write(reinterpret_cast<const char *>(eof), (std::streamsize)n);
    writer_->flush();
    std::ostream *w = writer_;
    writer_ = nullptr;
    return w;
}

// ---------------------------------------------------------------- Decoder
static uint16_t rd16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }
static uint32_t rd32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

Decoder::Decoder(const uint8_t *data, size_t size) : data_(data), size_(size) {
    if (size < 25) throw CodecError(CodecError::Deserialize, "stream shorter than a header");
    if (memcmp(data, "adder", 5) != 0) throw CodecError(CodecError::WrongMagic, "not a raw ADDER stream");
    meta_.codec_version = data[5];
    meta_.plane = PlaneSize(rd16(data + 7), rd16(data + 9), data[24]);
    meta_.tps = rd32(data + 11);
    meta_.ref_interval = rd32(data + 15);
    meta_.delta_t_max = rd32(data + 19);
    meta_.event_size = data[23] == 10 ? 11 : data[23];  // "manual fix for malformed files" (decoder.rs:132-135)
    meta_.time_mode = TimeMode::AbsoluteT;
    meta_.source_camera = SourceCamera::FramedU8;
    meta_.adu_interval = 0;
    pos_ = 25;
    auto ext = [&]() -> uint32_t {
        if (pos_ + 4 > size_) throw CodecError(CodecError::Deserialize, "truncated header extension");
        const uint32_t v = rd32(data_ + pos_);
        pos_ += 4;
        return v;
    };
    if (meta_.codec_version >= 1) meta_.source_camera = (SourceCamera)ext();
    if (meta_.codec_version >= 2) meta_.time_mode = (TimeMode)ext();
    if (meta_.codec_version >= 3) meta_.adu_interval = ext();
    if (meta_.codec_version > 3)
        throw CodecError(CodecError::UnsupportedVersion, "codec version " + std::to_string(meta_.codec_version));
    meta_.header_size = pos_;
}

Decoder Decoder::new_compressed(const uint8_t *data, size_t size) {
    if (size < 5 || memcmp(data, "addec", 5) != 0) throw CodecError(CodecError::WrongMagic, "not a compressed ADDER stream");
    Decoder d;
    d.data_ = data;
    d.size_ = size;
    d.compressed_ = true;
    AdderCompressedParams p;
    adder_compressed_default_params(&p, 1, 1, 1);
    size_t n = 0;
    int rc = adder_compressed_decode(data, size, 1, &p, nullptr, 0, &n);  // a counting pass: fills p from the header
    if (rc != ADDER_OK && rc != ADDER_E_OUT_CAPACITY) throw CodecError(CodecError::Deserialize, "malformed compressed stream");
    d.decoded_.resize(n);
    if (n) {
        rc = adder_compressed_decode(data, size, 1, &p, d.decoded_.data(), n, &n);
        if (rc != ADDER_OK) throw CodecError(CodecError::Deserialize, "malformed compressed stream");
    }
    d.meta_.codec_version = p.codec_version;
    d.meta_.plane = PlaneSize(p.width, p.height, p.channels);
    d.meta_.tps = p.tps;
    d.meta_.ref_interval = p.ref_interval;
    d.meta_.delta_t_max = p.delta_t_max;
    d.meta_.event_size = 0;  // (compressed streams carry no fixed-size records: header.rs writes 0)
    d.meta_.time_mode = (TimeMode)p.time_mode;
    d.meta_.source_camera = (SourceCamera)p.source_camera;
    d.meta_.adu_interval = p.adu_interval;
    return d;
}

bool Decoder::digest_event(Event *out) {
    if (compressed_) {
        if (pos_ >= decoded_.size()) return false;
        *out = decoded_[pos_++];
        return true;
    }
    const size_t es = meta_.event_size;
    if (pos_ + es > size_) return false;
    const uint8_t *p = data_ + pos_;
    Event e;
    e.x = rd16(p);
    e.y = rd16(p + 2);
    e.pad = 0;
    if (meta_.plane.c() == 1) {
        e.c = ADDER_C_NONE;
        e.d = p[4];
        e.t = rd32(p + 5);
    } else {
        if (p[4] != 1) throw CodecError(CodecError::Deserialize, "event without a channel in a multi-channel stream");
        e.c = p[5];
        e.d = p[6];
        e.t = rd32(p + 7);
    }
    if (e.x == 0xFFFF && e.y == 0xFFFF) return false;  // Coord::is_eof -> CodecError::Eof
    pos_ += es;
    *out = e;
    return true;
}

// ---------------------------------------------------------------- Video
static void hip_check(AdderHipCtx *ctx, int rc) {
    if (rc == ADDER_OK) return;
    const char *m = adder_hip_last_error(ctx);
    throw SourceError(rc == ADDER_E_BAD_PARAMS ? SourceError::BadParams : SourceError::Hip,
                      std::string("adder_hip: ") + (m ? m : "") + " (" + std::to_string(rc) + ")");
}

Video::Video(PlaneSize plane, std::ostream *writer, int device_id, Mode pixel_tree_mode)
    : plane_(plane), device_id_(device_id), pixel_tree_mode_(pixel_tree_mode) {
    CodecMetadata meta;  // video.rs:391-402
    meta.codec_version = LATEST_CODEC_VERSION;
    meta.header_size = 0;
    meta.time_mode = TimeMode::AbsoluteT;
    meta.plane = plane;
    meta.tps = tps_;
    meta.ref_interval = ref_time_;
    meta.delta_t_max = delta_t_max_;
    meta.event_size = 0;
    meta.source_camera = SourceCamera::FramedU8;
    meta.adu_interval = 0;
    EncoderOptions opts = EncoderOptions::default_(plane);
    encoder_.reset(new Encoder(writer ? Encoder::new_raw(meta, writer, opts) : Encoder::new_empty(meta, opts)));
}

Video::~Video() {
    if (ctx_) adder_hip_destroy(ctx_);
}

Video &Video::chunk_rows(size_t n) {
    if (n == 0) throw SourceError(SourceError::BadParams, "chunk_rows must be > 0");
    if (ctx_) throw SourceError(SourceError::BadParams, "chunk_rows must be set before the first frame");
    chunk_rows_ = n;
    return *this;
}

Video &Video::time_parameters(uint32_t tps, uint32_t ref_time, uint32_t delta_t_max, std::optional<TimeMode> tm) {
    if (tm) px_time_mode_ = *tm;  // px.time_mode(time_mode) for every pixel (:499-503)
    if (delta_t_max < ref_time) return *this;  // reference prints a warning and keeps the current values (:527-533)
    if (ctx_) {
        if (ref_time != ref_time_) throw SourceError(SourceError::BadParams, "ref_time cannot change mid-stream here");
        hip_check(ctx_, adder_hip_set_delta_t_max(ctx_, delta_t_max));
        if (tm) hip_check(ctx_, adder_hip_set_time_mode(ctx_, (uint8_t)*tm));
    }
    delta_t_max_ = delta_t_max;
    ref_time_ = ref_time;
    tps_ = tps;
    return *this;
}

Video &Video::write_out(std::optional<SourceCamera> source_camera, std::optional<TimeMode> time_mode,
                        std::optional<PixelMultiMode> pixel_multi_mode, std::optional<size_t> adu_interval,
                        EncoderType encoder_type, EncoderOptions encoder_options, std::ostream *write) {
    if (ctx_ && pixel_multi_mode.value_or(PixelMultiMode::Collapse) != multi_mode_)
        throw SourceError(SourceError::BadParams, "pixel_multi_mode cannot change after the first frame");
    multi_mode_ = pixel_multi_mode.value_or(PixelMultiMode::Collapse);
    CodecMetadata meta;
    meta.codec_version = LATEST_CODEC_VERSION;
    meta.header_size = 0;
    meta.time_mode = time_mode.value_or(TimeMode::AbsoluteT);
    meta.plane = plane_;
    meta.tps = tps_;
    meta.ref_interval = ref_time_;
    meta.delta_t_max = delta_t_max_;
    meta.event_size = 0;
    meta.source_camera = source_camera.value_or(SourceCamera::FramedU8);
    // Default::default() for Raw / Empty (:612, :630); adu_interval.unwrap_or_default() for Compressed (:573)
    meta.adu_interval = encoder_type == EncoderType::Compressed ? adu_interval.value_or(0) : 0;
    if (encoder_type == EncoderType::Compressed)  // video.rs:556-579 (the reference's "compression" feature)
        encoder_.reset(new Encoder(Encoder::new_compressed(meta, write, encoder_options)));
    else
        encoder_.reset(new Encoder(encoder_type == EncoderType::Raw ? Encoder::new_raw(meta, write, encoder_options)
                                                                   : Encoder::new_empty(meta, encoder_options)));
    if (time_mode) {  // px.time_mode(time_mode) (:632-634)
        px_time_mode_ = *time_mode;
        if (ctx_) hip_check(ctx_, adder_hip_set_time_mode(ctx_, (uint8_t)*time_mode));
    }
    if (ctx_)  // the encoder's Crf is replaced, the pixels' c_thresh is not (:629)
        hip_check(ctx_, adder_hip_set_crf_parameters(ctx_, encoder_options.crf.get_parameters().c_thresh_max,
                                                     encoder_options.crf.get_parameters().c_increase_velocity));
    return *this;
}

std::ostream *Video::end_write_stream() {
    // pending per-pixel events are dropped, like the reference (only the writer is closed)
    std::ostream *w = encoder_->close_writer();
    CodecMetadata meta;
    encoder_.reset(new Encoder(Encoder::new_empty(meta, encoder_->options)));
    return w;
}

void Video::update_crf(uint8_t crf) {
    encoder_->options.crf = Crf(crf, plane_);
    const CrfParameters &p = encoder_->options.crf.get_parameters();
    px_c_thresh_reset_ = p.c_thresh_baseline;
    if (ctx_) {
        hip_check(ctx_, adder_hip_set_crf_parameters(ctx_, p.c_thresh_max, p.c_increase_velocity));
        hip_check(ctx_, adder_hip_reset_c_thresh(ctx_, p.c_thresh_baseline));
        px_c_thresh_reset_.reset();
        sync_feature_controls();
    }
}

void Video::update_detect_features(bool detect_features, ShowFeatureMode, bool feature_rate_adjustment, bool) {
    feature_detection_ = detect_features;
    feature_rate_adjustment_ = feature_rate_adjustment;
    if (ctx_) sync_feature_controls();
}

void Video::update_roi(std::optional<Roi> roi) {
    roi_ = roi;
    if (ctx_) sync_feature_controls();
}

// the device context's copy of {feature_detection, feature_rate_adjustment, roi} and of the two CrfParameters the
// feedback reads (video.rs:1085-1105, 866-882)
void Video::sync_feature_controls() {
    const CrfParameters &p = encoder_->options.crf.get_parameters();
    hip_check(ctx_, adder_hip_set_feature_parameters(ctx_, p.c_thresh_baseline, p.feature_c_radius));
    hip_check(ctx_, adder_hip_update_detect_features(ctx_, feature_detection_, feature_rate_adjustment_));
    const Roi r = roi_.value_or(Roi{});
    hip_check(ctx_, adder_hip_update_roi(ctx_, roi_.has_value(), r.start_x, r.start_y, r.end_x, r.end_y));
}

std::vector<uint8_t> Video::feature_set() {
    ensure_ctx();
    std::vector<uint8_t> out((size_t)plane_.w() * plane_.h());
    hip_check(ctx_, adder_hip_feature_set(ctx_, out.data()));
    return out;
}

std::vector<uint8_t> Video::running_intensities() {
    ensure_ctx();
    std::vector<uint8_t> out(plane_.volume());
    hip_check(ctx_, adder_hip_running_intensities(ctx_, out.data()));
    return out;
}

void Video::update_quality_manual(uint8_t c_thresh_baseline, uint8_t c_thresh_max, uint32_t delta_t_max_multiplier,
                                  uint8_t c_increase_velocity, float feature_c_radius) {
    Crf &crf = encoder_->options.crf;
    crf.override_c_thresh_baseline(c_thresh_baseline);
    crf.override_c_thresh_max(c_thresh_max);
    crf.override_c_increase_velocity(c_increase_velocity);
    crf.override_feature_c_radius((uint16_t)feature_c_radius);
    delta_t_max_ = delta_t_max_multiplier * ref_time_;
    px_c_thresh_reset_ = c_thresh_baseline;
    if (ctx_) {
        hip_check(ctx_, adder_hip_set_crf_parameters(ctx_, c_thresh_max, c_increase_velocity));
        hip_check(ctx_, adder_hip_set_delta_t_max(ctx_, delta_t_max_));
        hip_check(ctx_, adder_hip_reset_c_thresh(ctx_, c_thresh_baseline));
        px_c_thresh_reset_.reset();
        sync_feature_controls();
    }
}

void Video::ensure_ctx() {
    if (ctx_) return;
    AdderHipParams p;
    adder_hip_default_params(&p, plane_.w(), plane_.h(), plane_.c());
    p.time_mode = (uint8_t)px_time_mode_;
    p.multi_mode = (uint8_t)multi_mode_;
    p.ref_time = ref_time_;
    p.delta_t_max = delta_t_max_;
    p.c_thresh_max = encoder_->options.crf.get_parameters().c_thresh_max;
    p.c_increase_velocity = encoder_->options.crf.get_parameters().c_increase_velocity;
    p.chunk_rows = (uint32_t)chunk_rows_;
    p.device_id = device_id_;
    p.pixel_mode = (uint8_t)pixel_tree_mode_;
    const int rc = adder_hip_create(&p, &ctx_);
    if (rc != ADDER_OK) {
        ctx_ = nullptr;
        hip_check(nullptr, rc);
    }
    if (px_c_thresh_reset_) {
        hip_check(ctx_, adder_hip_reset_c_thresh(ctx_, *px_c_thresh_reset_));
        px_c_thresh_reset_.reset();
    }
    if (pixel_tree_mode_ == Mode::FramePerfect) sync_feature_controls();
}

std::vector<std::vector<Event>> Video::integrate_matrix(const Frame &matrix, float time_spanned) {
    if (matrix.size() != plane_.volume())
        throw SourceError(SourceError::BadParams, "frame does not match the plane");
    ensure_ctx();
    // in_interval_count starts at 1 (video.rs:231), so set_initial_d (:656-658) is never taken here
    in_interval_count_ += 1;
    const uint32_t num_chunks = adder_hip_num_chunks(ctx_);
    // one frame through the ring: the events come back as a view of a page-locked slot that holds the mode's
    // worst case, so nothing can overflow and nothing is copied before the per-chunk vectors are built
    hip_check(ctx_, adder_hip_frame_submit(ctx_, matrix.data(), (size_t)plane_.w() * plane_.c(), time_spanned));
    const AdderEvent *ev = nullptr;
    const uint32_t *offs = nullptr;
    size_t n = 0;
    hip_check(ctx_, adder_hip_frame_collect(ctx_, &ev, &n, &offs));
    const Event *events = reinterpret_cast<const Event *>(ev);
    std::vector<std::vector<Event>> big_buffer(num_chunks);
    for (uint32_t ch = 0; ch < num_chunks; ++ch) big_buffer[ch].assign(events + offs[ch], events + offs[ch + 1]);
    encoder_->ingest_events(events, n);  // for events in &big_buffer { encoder.ingest_event } (:736-740)
    return big_buffer;
}

std::vector<Event> Video::integrate_frames(const uint8_t *frames, uint32_t num_frames, float time_spanned,
                                           std::vector<uint64_t> *frame_offsets) {
    ensure_ctx();
    in_interval_count_ += num_frames;
    // start from a typical event density; a batch that needs more (a scene cut flushes every pixel's arena)
    // comes back with ADDER_E_OUT_CAPACITY, the state rolled back and the size needed in n: retry with that
    std::vector<Event> out(std::min<size_t>(adder_hip_max_events_per_frame(ctx_), plane_.volume() + 64) * num_frames);
    std::vector<uint64_t> offs(num_frames + 1);
    size_t n = 0;
    int rc = adder_hip_integrate_batch(ctx_, frames, num_frames, 0, 0, time_spanned, out.data(), out.size(), &n, offs.data());
    if (rc == ADDER_E_OUT_CAPACITY) {
        out.resize(n);
        rc = adder_hip_integrate_batch(ctx_, frames, num_frames, 0, 0, time_spanned, out.data(), out.size(), &n, offs.data());
    }
    hip_check(ctx_, rc);
    out.resize(n);
    encoder_->ingest_events(out.data(), n);
    if (frame_offsets) *frame_offsets = offs;
    return out;
}

std::vector<Event> Video::integrate_sparse(const std::vector<AdderSparseStep> &steps) {
    ensure_ctx();
    std::vector<Event> out(steps.size() * (size_t)(adder_hip_max_events_per_frame(ctx_) / plane_.volume() + 2));
    size_t n = 0;
    hip_check(ctx_, adder_hip_integrate_sparse(ctx_, steps.data(), steps.size(), out.data(), out.size(), &n));
    out.resize(n);
    encoder_->ingest_events(out.data(), n);
    return out;
}

// ---------------------------------------------------------------- Prophesee (prophesee.rs)
static void mid_clamp_u8(double &frame_val, double &last_val_ln) {  // utils/cv.rs:444-449
    if (frame_val < 0.0 || frame_val > 255.0) {
        frame_val = 128.0;
        last_val_ln = std::log1p(128.0 / 255.0);
    }
}
static uint8_t f64_as_u8(double v) { return !(v > 0.0) ? 0 : (v >= 255.0 ? 255 : (uint8_t)v); }  // `as u8`

Prophesee::Prophesee(uint32_t ref_time, uint16_t width, uint16_t height, std::function<bool(DvsEvent &)> decode_event,
                     int device_id)
    : decode_event_(std::move(decode_event)), video_(PlaneSize(width, height, 1), nullptr, device_id, Mode::Continuous) {
    video_.chunk_rows(1);
    video_.time_parameters(ref_time * PROPHESEE_SOURCE_TPS, ref_time, ref_time * 2, TimeMode::AbsoluteT);
    const size_t n = video_.plane().volume();
    dvs_last_timestamps_.assign(n, 2u);
    dvs_last_ln_val_.assign(n, std::log1p(128.0 / 255.0));
}

std::vector<std::vector<Event>> Prophesee::consume() {
    const uint32_t ref_time = video_.get_ref_time();
    const size_t W = video_.plane().w();
    if (running_t_ == 0) {  // :117-131: two frames of the start intensities
        const Frame start(video_.plane().volume(), 128);
        video_.integrate_matrix(start, (float)ref_time);
        size_t first = 0;
        for (auto &v : video_.integrate_matrix(start, (float)ref_time)) first += v.size();
        if (first != video_.plane().volume()) throw SourceError(SourceError::Codec, "assert_eq!(first_events.len(), volume)");
        running_t_ = 2;
    }
    const uint32_t view_interval = PROPHESEE_SOURCE_TPS / 60;  // :136
    std::vector<DvsEvent> dvs_events;
    const uint32_t start_running_t = running_t_;
    for (;;) {  // :142-168
        DvsEvent e;
        if (!decode_event_(e)) {
            end_events();
            throw SourceError(SourceError::NoData, "End of input file");
        }
        e.t -= t_subtract_;
        if (e.t > running_t_) running_t_ = e.t;
        dvs_events.push_back(e);
        if (e.t > start_running_t + view_interval) break;
    }
    // :174-258 -- for every DVS event, the pixel's previous intensity over the time since its last event, then one
    // source time unit of the new one
    std::vector<AdderSparseStep> steps;
    steps.reserve(dvs_events.size() * 2);
    for (const DvsEvent &e : dvs_events) {
        if (e.x >= W || e.y >= video_.plane().h()) throw SourceError(SourceError::BadParams, "DVS event outside the plane");
        const size_t px = (size_t)e.y * W + e.x;
        const uint32_t t = e.t, last_t = dvs_last_timestamps_[px];
        if (t < last_t) continue;
        double last_ln_val = dvs_last_ln_val_[px];
        if (t > last_t + 1) {
            double last_val = (std::exp(last_ln_val) - 1.0) * 255.0;
            mid_clamp_u8(last_val, last_ln_val);
            const uint32_t time_spanned = (t - last_t - 1) * ref_time;
            const double intensity_to_integrate = last_val * (double)(t - last_t - 1);
            // (t > last_t + 1 implies the second step below: the side plane is sampled after that one, :259-283)
            steps.push_back(AdderSparseStep{e.x, e.y, ADDER_C_NONE, f64_as_u8(last_val), ADDER_SPARSE_NO_SIDE,
                                            (float)intensity_to_integrate, (float)time_spanned});
        }
        if (e.p > 1) throw SourceError(SourceError::BadParams, "Invalid polarity");
        double new_ln_val = e.p == 0 ? last_ln_val - camera_theta_ : last_ln_val + camera_theta_;
        dvs_last_ln_val_[px] = new_ln_val;
        dvs_last_timestamps_[px] = t;
        if (t > last_t) {
            double new_val = (std::exp(new_ln_val) - 1.0) * 255.0;
            mid_clamp_u8(new_val, new_ln_val);
            dvs_last_ln_val_[px] = new_ln_val;
            steps.push_back(AdderSparseStep{e.x, e.y, ADDER_C_NONE, f64_as_u8(new_val), 0, (float)new_val, (float)ref_time});
        }
    }
    // "just wrap the vec in another vec" (:302-304); Video::integrate_sparse has pushed them through the encoder (:308-312)
    std::vector<std::vector<Event>> nested(1);
    nested[0] = video_.integrate_sparse(steps);
    return nested;
}

void Prophesee::end_events() {  // :332-372: every pixel's last intensity up to the end of the recording
    const uint32_t ref_time = video_.get_ref_time();
    const size_t W = video_.plane().w(), H = video_.plane().h();
    std::vector<AdderSparseStep> steps;
    steps.reserve(W * H);
    for (size_t y = 0; y < H; ++y)
        for (size_t x = 0; x < W; ++x) {
            const size_t px = y * W + x;
            const double last_val = (std::exp(dvs_last_ln_val_[px]) - 1.0) * 255.0;
            if (!(running_t_ - dvs_last_timestamps_[px] > 0))
                throw SourceError(SourceError::Codec, "assert!(running_t - dvs_last_timestamps > 0)");
            const uint32_t time_spanned = (running_t_ - dvs_last_timestamps_[px]) * ref_time;
            const double intensity_to_integrate = last_val * (double)time_spanned;
            steps.push_back(AdderSparseStep{(uint16_t)x, (uint16_t)y, ADDER_C_NONE, f64_as_u8(last_val), ADDER_SPARSE_NO_SIDE,
                                            (float)intensity_to_integrate, (float)time_spanned});
        }
    last_end_events_ = video_.integrate_sparse(steps);
}

// ---------------------------------------------------------------- Davis
static void clamp_u8(double &frame_val, double &last_val_ln) {  // utils/cv.rs:432-440
    if (frame_val <= 0.0) {
        frame_val = 0.0;
        last_val_ln = std::log1p(0.0);
    } else if (frame_val > 255.0) {
        frame_val = 255.0;
        last_val_ln = std::log1p(1.0);
    }
}

Davis::Davis(uint16_t width, uint16_t height, TranscoderMode mode, std::function<bool(DavisPacket &)> next_packet, int device_id)
    : next_packet_(std::move(next_packet)),
      video_(PlaneSize(width, height, 1), nullptr, device_id, mode == TranscoderMode::Framed ? Mode::FramePerfect : Mode::Continuous),
      mode_(mode) {
    if (height % 4 != 0) throw SourceError(SourceError::BadParams, "Davis: the height must be a multiple of 4");
    video_.chunk_rows(height / 4);
    dvs_last_timestamps_.assign(video_.plane().volume(), 0);
    dvs_last_ln_val_.assign(video_.plane().volume(), 0.0);
}

double Davis::get_running_input_bitrate() const {
    const double volume = (double)video_.plane().volume();
    if (mode_ == TranscoderMode::Framed)
        return volume * 8.0 * ((double)video_.get_tps() / (ref_time_divisor_ * (double)video_.get_ref_time()));
    const double time_mult = 1e6 / time_change_;
    const double event_bits = (double)num_dvs_events_ * time_mult * 9.0 * 8.0;
    return mode_ == TranscoderMode::RawDavis ? event_bits + time_mult * volume * 8.0 : event_bits;
}

// :233-466.  The events of the four row chunks are handled chunk after chunk, each in arrival order (:254-258, the
// rayon map is collected in chunk order); per event that passes the checks: the pixel's OLD intensity over the time
// since its last event (no contrast test), then the NEW value against base_val without integrating it.
std::vector<Event> Davis::integrate_dvs_events(const std::vector<DavisDvsEvent> &ev, int64_t frame_timestamp, bool check2_after,
                                               bool has_ts2, int64_t frame_timestamp_2) {
    const size_t W = video_.plane().w(), H = video_.plane().h();
    const size_t chunk_h = H / 4;
    const float ticks_per_micro = (float)video_.get_tps() / 1e6f;
    const float ref_time_f = (float)video_.get_ref_time();
    std::vector<AdderSparseStep> steps;
    steps.reserve(ev.size() * 2);
    for (size_t chunk = 0; chunk < 4; ++chunk)
        for (const DavisDvsEvent &e : ev) {
            if (e.x >= W || e.y >= H) throw SourceError(SourceError::BadParams, "DVS event outside the plane");
            if (e.y / chunk_h != chunk) continue;
            // "Ignore events occuring during the deblurred frame's effective exposure time" (:287-294)
            if (!(e.t < frame_timestamp)) continue;  // check_dvs_before
            if (has_ts2 && !(check2_after ? e.t > frame_timestamp_2 : e.t < frame_timestamp_2)) continue;
            const size_t px = (size_t)e.y * W + e.x;
            double &last_val_ln = dvs_last_ln_val_[px];
            const double last_val = (std::exp(last_val_ln) - 1.0) * 255.0;
            const int64_t delta_t_micro = e.t - dvs_last_timestamps_[px];
            if (delta_t_micro == e.t) continue;
            const float delta_t_ticks = (float)delta_t_micro * ticks_per_micro;
            if (delta_t_ticks < 0.0f) continue;
            const float first_integration = std::max((float)last_val / ref_time_f * delta_t_ticks, 0.0f);
            steps.push_back(AdderSparseStep{e.x, e.y, ADDER_C_NONE, 0, ADDER_SPARSE_NO_SIDE | ADDER_SPARSE_INTEGRATE_ONLY,
                                            first_integration, delta_t_ticks});
            last_val_ln *= std::exp(e.on ? dvs_c_ : -dvs_c_);
            double frame_val = (std::exp(last_val_ln) - 1.0) * 255.0;
            clamp_u8(frame_val, last_val_ln);
            steps.push_back(AdderSparseStep{e.x, e.y, ADDER_C_NONE, f64_as_u8(frame_val), ADDER_SPARSE_NO_SIDE | ADDER_SPARSE_TEST_ONLY,
                                            (float)frame_val, 0.0f});
            dvs_last_timestamps_[px] = e.t;
        }
    return video_.integrate_sparse(steps);
}

// :468-598 -- every pixel's last intensity over the time from its last event to the start of the frame
std::vector<Event> Davis::integrate_frame_gaps() {
    if (!have_start_) throw SourceError(SourceError::Codec, "UninitializedData");
    const size_t W = video_.plane().w(), H = video_.plane().h();
    const float ticks_per_micro = (float)video_.get_tps() / 1e6f;
    std::vector<AdderSparseStep> steps;
    steps.reserve(W * H);
    for (size_t y = 0; y < H; ++y)
        for (size_t x = 0; x < W; ++x) {
            const size_t px = y * W + x;
            double last_val = (std::exp(dvs_last_ln_val_[px]) - 1.0) * 255.0;
            clamp_u8(last_val, dvs_last_ln_val_[px]);
            const int64_t delta_t_micro = start_of_frame_timestamp_ - dvs_last_timestamps_[px];
            if (delta_t_micro == start_of_frame_timestamp_) continue;
            const float delta_t_ticks = (float)delta_t_micro * ticks_per_micro;
            if (delta_t_ticks <= 0.0f) continue;
            const double integration = std::max((last_val / (double)video_.get_ref_time()) * (double)delta_t_ticks, 0.0);
            steps.push_back(AdderSparseStep{(uint16_t)x, (uint16_t)y, ADDER_C_NONE, f64_as_u8(last_val), 0, (float)integration,
                                            delta_t_ticks});
        }
    return video_.integrate_sparse(steps);
}

std::vector<std::vector<Event>> Davis::consume() {
    const bool with_events = mode_ != TranscoderMode::Framed;
    const size_t W = video_.plane().w(), H = video_.plane().h(), n = W * H;
    std::vector<std::vector<Event>> ret;
    auto log = [&](const std::vector<Event> &v) { ingested_.insert(ingested_.end(), v.begin(), v.end()); };
    if (have_cached_) {
        have_cached_ = false;
        if (cached_end_) {  // :641-667 "We've reached the end of the input. Forcibly pop the last event from each pixel."
            if (mode_ != TranscoderMode::Framed) {
                std::vector<AdderSparseStep> steps;
                steps.reserve(n);
                for (size_t y = 0; y < H; ++y)
                    for (size_t x = 0; x < W; ++x)
                        steps.push_back(AdderSparseStep{(uint16_t)x, (uint16_t)y, ADDER_C_NONE, 0,
                                                        ADDER_SPARSE_NO_SIDE | ADDER_SPARSE_FLUSH, 0.0f, 0.0f});
                log(video_.integrate_sparse(steps));
            }
            throw SourceError(SourceError::NoData, "End of input");
        }
        DavisPacket pk = std::move(cached_);
        if (pk.frame.size() != n) throw SourceError(SourceError::BadParams, "Davis: a frame of the wrong size");
        if (pk.has_events) {  // :668-699
            time_change_ = (double)(pk.img_start_ts - (have_start_ ? start_of_frame_timestamp_ : 0));
            num_dvs_events_ = pk.events_before.size() + pk.events_after.size();
            start_of_frame_timestamp_ = pk.img_start_ts;
            end_of_frame_timestamp_ = pk.img_end_ts;
            have_start_ = have_end_ = true;
            if (mode_ == TranscoderMode::RawDvs) end_of_frame_timestamp_ = pk.img_start_ts + 1;
            dvs_c_ = pk.c;
            events_before_ = std::move(pk.events_before);
            events_after_ = std::move(pk.events_after);
            ref_time_divisor_ = (double)(pk.img_end_ts - pk.img_start_ts) / (double)video_.get_ref_time();
        }
        const int64_t start_ts = have_start_ ? start_of_frame_timestamp_ : 0;
        const int64_t end_ts = have_end_ ? end_of_frame_timestamp_ : (int64_t)video_.get_ref_time();
        if (temp_first_frame_start_timestamp_ == 0) temp_first_frame_start_timestamp_ = start_ts;
        if (with_events) {
            // (VideoState::in_interval_count starts at 1 and only grows, video.rs:231,662: the "very beginning" arm of
            // :722-728 is dead; the pixels' last timestamps are 0 until the first frame and every event before it is skipped)
            if (have_last_after_ && have_end_of_last_)
                log(integrate_dvs_events(events_last_after_, start_ts, true, mode_ != TranscoderMode::RawDvs,
                                         end_of_last_frame_timestamp_));
            log(integrate_dvs_events(events_before_, start_ts, false, false, 0));
            log(integrate_frame_gaps());
        }
        // :775-838 -- convert_to(CV_8U, 255.0): saturate_cast<uchar>(cvRound(v * 255)), round half to even
        Frame image_8u(n);
        for (size_t i = 0; i < n; ++i) {
            const double v = std::nearbyint(pk.frame[i] * 255.0);
            image_8u[i] = (uint8_t)(v <= 0.0 ? 0.0 : (v >= 255.0 ? 255.0 : v));
        }
        float mat_integration_time = (float)video_.get_ref_time();
        if (mode_ == TranscoderMode::RawDavis) mat_integration_time = (float)(end_ts - start_ts);
        if (mode_ == TranscoderMode::RawDvs) {
            dvs_c_ = 0.15;
            std::fill(image_8u.begin(), image_8u.end(), 0);
            mat_integration_time = 0.0f;
        }
        if (mode_ == TranscoderMode::Framed) {
            ret = video_.integrate_matrix(image_8u, mat_integration_time);
            for (auto &v : ret) log(v);
        } else {
            // the frame as one integrate_for_px per pixel in raster order (video.rs:697-731): a context that has taken
            // sparse steps keeps c_thresh, its counter and running_t per pixel
            std::vector<AdderSparseStep> steps;
            steps.reserve(n);
            for (size_t y = 0; y < H; ++y)
                for (size_t x = 0; x < W; ++x)
                    steps.push_back(AdderSparseStep{(uint16_t)x, (uint16_t)y, ADDER_C_NONE, image_8u[y * W + x], 0,
                                                    (float)image_8u[y * W + x], mat_integration_time});
            const std::vector<Event> ev = video_.integrate_sparse(steps);
            log(ev);
            const size_t chunk_rows = video_.get_chunk_rows();
            ret.assign((H + chunk_rows - 1) / chunk_rows, {});
            for (const Event &e : ev) ret[e.y / chunk_rows].push_back(e);
        }
        for (size_t i = 0; i < n; ++i)  // :840-864
            dvs_last_ln_val_[i] = mode_ == TranscoderMode::RawDvs ? std::log1p(0.5) : std::log1p(pk.frame[i]);
        if (with_events) {
            events_last_after_ = events_after_;
            have_last_after_ = true;
            end_of_last_frame_timestamp_ = end_ts;
            have_end_of_last_ = have_end_;
            std::fill(dvs_last_timestamps_.begin(), dvs_last_timestamps_.end(), end_ts);
        }
    }
    cached_ = DavisPacket();
    cached_end_ = !next_packet_(cached_);  // (the reference fetches on a thread while it integrates)
    have_cached_ = true;
    return ret;
}

// ---------------------------------------------------------------- Framed
Frame handle_color(const Frame &input, uint32_t width, uint32_t height, uint32_t channels, bool color) {
    if (color || channels == 1) return input;
    Frame out((size_t)width * height);
    for (size_t i = 0; i < out.size(); ++i) {
        const double v = (double)input[3 * i] * 0.114 + (double)input[3 * i + 1] * 0.587 + (double)input[3 * i + 2] * 0.299;
        out[i] = v >= 255.0 ? 255 : (v <= 0.0 ? 0 : (uint8_t)v);  // `as u8`
    }
    return out;
}

Framed::Framed(FrameProvider provider, bool color_input, int device_id)
    : source_fps(provider.frame_rate), cap_(provider), color_input_(color_input),
      video_(PlaneSize((uint16_t)provider.width, (uint16_t)provider.height, color_input ? 3 : 1), nullptr, device_id) {
    if (color_input && provider.channels != 3)
        throw SourceError(SourceError::BadParams, "color input needs a 3-channel frame provider");
}

Framed &Framed::frame_start(uint32_t idx) {
    if (idx >= cap_.frame_count) throw SourceError(SourceError::StartOutOfBounds, "frame_start out of bounds");
    frame_idx_start = idx;
    next_frame_ = idx;
    return *this;
}

Framed &Framed::auto_time_parameters(uint32_t ref_time, uint32_t delta_t_max, std::optional<TimeMode> time_mode) {
    if (delta_t_max % ref_time != 0)
        throw SourceError(SourceError::BadParams, "delta_t_max must be a multiple of ref_time");
    const uint32_t tps = (uint32_t)((float)ref_time * source_fps);  // framed.rs:101
    video_.time_parameters(tps, ref_time, delta_t_max, time_mode);
    return *this;
}

Framed &Framed::time_parameters(uint32_t tps, uint32_t ref_time, uint32_t delta_t_max, std::optional<TimeMode> tm) {
    if (delta_t_max % ref_time == 0) video_.time_parameters(tps, ref_time, delta_t_max, tm);
    return *this;  // otherwise: "delta_t_max must be a multiple of ref_time" is only printed (framed.rs:229-232)
}

Framed &Framed::write_out(SourceCamera source_camera, TimeMode time_mode, PixelMultiMode pixel_multi_mode,
                          std::optional<size_t> adu_interval, EncoderType encoder_type,
                          EncoderOptions encoder_options, std::ostream *write) {
    video_.write_out(source_camera, time_mode, pixel_multi_mode, adu_interval, encoder_type, encoder_options, write);
    return *this;
}

std::vector<std::vector<Event>> Framed::consume() {
    Frame raw;
    if (!cap_.decode || !cap_.decode(next_frame_, raw)) throw SourceError(SourceError::NoData, "no more frames");
    ++next_frame_;
    input_frame_ = handle_color(raw, cap_.width, cap_.height, cap_.channels, color_input_);
    return video_.integrate_matrix(input_frame_, (float)video_.get_ref_time());
}

double Framed::get_running_input_bitrate() const {
    return (double)video_.get_tps() / (double)video_.get_ref_time() * (double)video_.plane().volume() * 8.0;
}

}  // namespace adder_host

// ================================================================ framer (framer/driver.rs)
namespace adder_host {

std::unique_ptr<FrameSequence> FramerBuilder::finish(FrameElement element) { return std::make_unique<FrameSequence>(*this, element); }

FrameSequence::FrameSequence(const FramerBuilder &b, FrameElement element) : chunk_rows(b.chunk_rows_) {
    if (b.chunk_rows_ == 0) throw SourceError(SourceError::BadParams, "chunk_rows must be > 0");  // assert (:306)
    (void)b.mode_;  // driver.rs:270,383: stored and never read, INTEGRATION ingests like INSTANTANEOUS
    AdderFramerParams p;
    adder_framer_default_params(&p, b.plane_.w(), b.plane_.h(), b.plane_.c());
    p.codec_version = b.codec_version_;
    p.time_mode = (uint8_t)b.time_mode_;
    p.tps = b.tps_;
    p.ref_interval = b.ref_interval_;
    p.delta_t_max = b.delta_t_max_;
    p.output_fps = b.output_fps_ ? *b.output_fps_ : 0.0f;
    p.source_camera = (uint32_t)b.source_camera_;
    p.device_id = b.device_id_ < 0 ? 0 : b.device_id_;
    p.ring_frames = b.ring_frames_;
    p.view_mode = (uint8_t)b.view_mode_;
    p.source_type = (uint8_t)b.source_type_;
    p.value_type = (uint8_t)element;
    p.practical_d_max = b.practical_d_max_ ? *b.practical_d_max_
                                           : std::log2f(255.0f * (float)(b.delta_t_max_ / (b.ref_interval_ ? b.ref_interval_ : 1u)));
    if (adder_framer_create(&p, &fr_) != ADDER_OK)
        throw SourceError(SourceError::BadParams, std::string("framer: ") + adder_framer_last_error(nullptr));
    num_chunks_ = (b.plane_.h() + b.chunk_rows_ - 1) / b.chunk_rows_;
    frame_units_ = (size_t)b.plane_.w() * b.plane_.h() * b.plane_.c();
    frame_bytes_ = frame_units_ << (unsigned)element;
    width_ = b.plane_.w();
    height_ = b.plane_.h();
    channels_ = b.plane_.c();
}

FrameSequence::~FrameSequence() { adder_framer_destroy(fr_); }
uint32_t FrameSequence::tpf() const { return adder_framer_tpf(fr_); }
int64_t FrameSequence::frames_written() const { return adder_framer_frames_written(fr_); }

static void framer_check(AdderFramer *fr, int rc) {
    if (rc != ADDER_OK) throw SourceError(SourceError::BadParams, std::string("framer: ") + adder_framer_last_error(fr));
}

bool FrameSequence::is_frame_0_filled() {
    uint32_t n = 0;
    framer_check(fr_, adder_framer_frames_ready(fr_, &n));
    return n > 0;
}

bool FrameSequence::ingest_event(Event &event) {
    const uint64_t offs[2] = {0, 1};
    framer_check(fr_, adder_framer_ingest(fr_, &event, offs, 1));
    return is_frame_0_filled();
}

bool FrameSequence::ingest_events_events(const std::vector<std::vector<Event>> &events) {
    // "Make sure that the chunk division is aligned between the source and the framer" (:566)
    if (events.size() != num_chunks_) throw SourceError(SourceError::BadParams, "events.len() != number of framer chunks");
    flat_.clear();
    for (const auto &v : events) flat_.insert(flat_.end(), v.begin(), v.end());
    // The device wants segments in which a pixel-channel's events are contiguous.  One source frame's events (what
    // SimulProcessor hands over) are one such segment; any other list is cut where a pixel comes back.
    seg_offs_.assign(1, 0);
    seen_.assign(frame_units_, 0xffffffffu);
    uint32_t seg = 0;
    size_t prev = SIZE_MAX;
    for (size_t i = 0; i < flat_.size(); ++i) {
        const Event &e = flat_[i];
        const size_t ch = e.c == 0xff ? 0 : e.c;
        if (e.x >= width_ || e.y >= height_ || ch >= channels_) continue;  // reported by the device
        const size_t u = ((size_t)e.y * width_ + e.x) * channels_ + ch;
        if (u != prev && seen_[u] == seg) {
            seg_offs_.push_back(i);
            ++seg;
        }
        seen_[u] = seg;
        prev = u;
    }
    seg_offs_.push_back(flat_.size());
    framer_check(fr_, adder_framer_ingest(fr_, flat_.data(), seg_offs_.data(), (uint32_t)seg_offs_.size() - 1));
    return is_frame_0_filled();
}

bool FrameSequence::flush_frame_buffer() {
    int ready = 0;
    framer_check(fr_, adder_framer_flush(fr_, &ready));
    return ready != 0;
}

void FrameSequence::write_frame_bytes(std::ostream &writer) {
    out_.resize(frame_bytes_);
    framer_check(fr_, adder_framer_write_frame(fr_, out_.data()));
    writer.write(reinterpret_cast<const char *>(out_.data()), (std::streamsize)frame_bytes_);
}

int FrameSequence::write_multi_frame_bytes(std::ostream &writer) {
    int frames = 0;
    for (;;) {
        out_.resize(frame_bytes_ * 64);
        uint32_t n = 0;
        framer_check(fr_, adder_framer_pop(fr_, out_.data(), 64, &n));
        if (!n) break;
        writer.write(reinterpret_cast<const char *>(out_.data()), (std::streamsize)(frame_bytes_ * n));
        frames += (int)n;
        if (n < 64) break;
    }
    return frames;
}

// ---------------------------------------------------------------- SimulProcessor (utils/simulproc.rs)
SimulProcessor::SimulProcessor(Framed &source, uint32_t ref_time, std::ostream &frames_out, int32_t frame_max,
                               uint8_t codec_version, TimeMode time_mode, int device_id)
    : source_(source), out_(frames_out), frame_max_(frame_max) {
    const float reconstructed_frame_rate = source.source_fps;
    // "For instantaneous reconstruction, make sure the frame rate matches the source video rate" (:146-150)
    if (source.get_video_ref().get_tps() / ref_time != (uint32_t)reconstructed_frame_rate)
        throw SourceError(SourceError::BadParams, "tps / ref_time does not match the source frame rate");
    const Video &v = source.get_video_ref();
    framer_ = FramerBuilder(v.plane(), v.get_chunk_rows())
                  .codec_version(codec_version, time_mode)
                  .time_parameters(v.get_tps(), ref_time, v.get_delta_t_max(), reconstructed_frame_rate)
                  .mode(FramerMode::INSTANTANEOUS)
                  .source(SourceType::U8, SourceCamera::FramedU8)
                  .device(device_id)
                  .finish();
}

void SimulProcessor::run(uint32_t frame_max) {
    int frame_count = 1;  // :174
    bool framing = true;
    uint32_t consumed = 0;
    for (;;) {
        std::vector<std::vector<Event>> events;
        try {
            events = source_.consume();
        } catch (const SourceError &) {
            break;  // Err(e) => break (:246-249)
        }
        ++consumed;
        if (framing && framer_->ingest_events_events(events)) {
            const int n = framer_->write_multi_frame_bytes(out_);
            frame_count += n;
            frames_written += n;
            if (frame_count >= frame_max_ && frame_max_ > 0) framing = false;  // "Wrote max frames" (:208-211)
        }
        if (frame_max > 0 && consumed >= frame_max) break;  // :264-267
    }
    out_.flush();
    source_.get_video_mut().end_write_stream();
}

}  // namespace adder_host
