// adder_host.hpp -- host side of the framed->ADDER path, in C++ (the reference's host is Rust;
// this image has no Rust toolchain, see INTEGRATION.md for the Rust binding of the same ABI).
//
// Mirrors, with the same names, argument meaning and error behaviour, the part of the
// reference's operator surface this path lives behind:
//   PlaneSize / Event / TimeMode / PixelMultiMode / SourceCamera   adder-codec-core/src/lib.rs
//   Crf / CrfParameters / CRF table                                codec/rate_controller.rs
//   EncoderOptions / EncoderType / CodecMetadata                   codec/mod.rs
//   Encoder + RawOutput + EmptyOutput (sink)                       codec/encoder.rs, raw/stream.rs
//   Decoder + RawInput (reader, used by tests/tools)               codec/decoder.rs, raw/stream.rs
//   Video  (integrate_matrix & builders)                           transcoder/source/video.rs
//   Source trait + Framed                                          transcoder/source/{video,framed}.rs
// The per-pixel work of Video::integrate_matrix (video.rs:677-734) is NOT done here: it is
// one call into the C-ABI of libadder_hip.so (include/adder_hip.h).  There is no CPU path.
#pragma once
#include <stdint.h>

#include <functional>
#include <memory>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/adder_framer.h"
#include "../../include/adder_hip.h"

struct AdderCompressedEncoder;  // include/adder_compressed.h

namespace adder_host {

// ---------------------------------------------------------------- errors (video.rs:55-122, codec/mod.rs:209-256)
struct SourceError : std::runtime_error {
    enum Kind { BadParams, StartOutOfBounds, BufferEmpty, NoData, Hip, Codec } kind;
    SourceError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};
struct CodecError : std::runtime_error {
    enum Kind { WrongMagic, Deserialize, Eof, BadFile, UnsupportedVersion, Io } kind;
    CodecError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};

// ---------------------------------------------------------------- core types (lib.rs)
enum class TimeMode : uint32_t { DeltaT = 0, AbsoluteT = 1, Mixed = 2 };  // lib.rs:72-83 (default AbsoluteT)
enum class PixelMultiMode : uint8_t { Normal = 0, Collapse = 1 };         // lib.rs:207-213 (default Collapse)
enum class SourceCamera : uint32_t {                                      // lib.rs:35-47
    FramedU8 = 0, FramedU16, FramedU32, FramedU64, FramedF32, FramedF64, Dvs, DavisU8, Atis, Asint
};
enum class EncoderType { Compressed, Raw, Empty };  // codec/mod.rs
enum class Mode { FramePerfect = 0, Continuous = 1 };  // lib.rs:196-205
constexpr uint8_t LATEST_CODEC_VERSION = 3;         // codec/mod.rs:74

struct PlaneSize {  // lib.rs:86-178
    PlaneSize() = default;
    PlaneSize(uint16_t width, uint16_t height, uint8_t channels);  // throws like PlaneSize::new
    uint16_t w() const { return width_; }
    uint16_t h() const { return height_; }
    uint8_t c() const { return channels_; }
    size_t volume() const { return (size_t)width_ * height_ * channels_; }
    uint16_t min_resolution() const { return width_ < height_ ? width_ : height_; }

  private:
    uint16_t width_ = 1, height_ = 1;
    uint8_t channels_ = 1;
};

using Event = AdderEvent;  // lib.rs:371-377; c == ADDER_C_NONE <=> Coord::c == None
using Frame = std::vector<uint8_t>;  // [h][w][c] u8, row-major (ndarray::Array3<u8>)

// ---------------------------------------------------------------- rate control (rate_controller.rs)
extern const float CRF[10][4];                 // rate_controller.rs:5-18
constexpr uint8_t DEFAULT_CRF_QUALITY = 3;     // :21
struct CrfParameters {                         // :40-53
    uint8_t c_thresh_baseline, c_thresh_max, c_increase_velocity;
    uint16_t feature_c_radius;
};
class Crf {  // :24-37, :55-118
  public:
    Crf(std::optional<uint8_t> crf, PlaneSize plane);
    void update_quality(uint8_t crf);
    void override_c_thresh_baseline(uint8_t v) { parameters_.c_thresh_baseline = v; quality_.reset(); }
    void override_c_thresh_max(uint8_t v) { parameters_.c_thresh_max = v; quality_.reset(); }
    void override_c_increase_velocity(uint8_t v) { parameters_.c_increase_velocity = v; quality_.reset(); }
    void override_feature_c_radius(uint16_t v) { parameters_.feature_c_radius = v; quality_.reset(); }
    const CrfParameters &get_parameters() const { return parameters_; }
    std::optional<uint8_t> get_quality() const { return quality_; }
    PlaneSize plane;

  private:
    std::optional<uint8_t> quality_;
    CrfParameters parameters_;
};

// codec/mod.rs:285-317
struct EventDrop {
    enum Kind { None, Manual } kind = None;  // (`Auto` is `todo!()` in the reference, encoder.rs:251-253)
    double target_event_rate = 0.0;          // Manual
    double alpha = 0.0;                      // Manual: the decay rate in [0, 1]
    static EventDrop manual(double rate, double a) { return EventDrop{Manual, rate, a}; }
};
enum class EventOrder { Unchanged, Interleaved };

struct EncoderOptions {  // codec/mod.rs:264-283
    EventDrop event_drop;
    EventOrder event_order = EventOrder::Unchanged;
    Crf crf;
    static EncoderOptions default_(PlaneSize plane) { return EncoderOptions{EventDrop{}, EventOrder::Unchanged, Crf(std::nullopt, plane)}; }
};

struct CodecMetadata {  // codec/mod.rs:76-107
    uint8_t codec_version = LATEST_CODEC_VERSION;
    size_t header_size = 24;
    TimeMode time_mode = TimeMode::AbsoluteT;
    PlaneSize plane;
    uint32_t tps = 2550, ref_interval = 255, delta_t_max = 255;
    uint8_t event_size = 9;
    SourceCamera source_camera = SourceCamera::FramedU8;
    size_t adu_interval = 1;
};

// ---------------------------------------------------------------- sink: Encoder over RawOutput / EmptyOutput
// W of the reference (any std::io::Write) is a std::ostream here; the Encoder does not own it.
class Encoder {  // encoder.rs:29-37
  public:
    static Encoder new_raw(CodecMetadata meta, std::ostream *writer, EncoderOptions options);  // :94-108
    static Encoder new_empty(CodecMetadata meta, EncoderOptions options);                      // :57-72
    // :74-92 + CompressedOutput::new (compressed/stream.rs:126-166): the CPU compressed sink of libadder_hip.so
    // (include/adder_compressed.h); the stream is written to `writer` when the writer is closed
    static Encoder new_compressed(CodecMetadata meta, std::ostream *writer, EncoderOptions options);
    ~Encoder();
    Encoder(Encoder &&o) noexcept;
    Encoder &operator=(Encoder &&) = delete;
    Encoder(const Encoder &) = delete;
    const CodecMetadata &meta() const { return meta_; }
    void ingest_event(const Event &e);                                   // :233-273
    void ingest_events(const Event *events, size_t n);                   // :281-286
    void ingest_events_events(const std::vector<std::vector<Event>> &v); // :289-294
    std::ostream *close_writer();  // :154-168 -> RawOutput::into_writer writes the 11-byte EOF and flushes
    EncoderOptions options;
    // EventDrop::Manual reads the wall clock (`Instant::now()`, :243); tests inject their own (seconds)
    std::function<double()> clock;
    size_t queued() const { return queue_.size(); }  // EventOrder::Interleaved: events still in the heap
    void reset_clock_origin(double t) { last_event_ts_ = t; }  // tests: EncoderState::last_event_ts

  private:
    Encoder(CodecMetadata meta, std::ostream *w, EncoderOptions o, EncoderType t);
    void encode_header();  // :170-229
    void output_events(const Event *events, size_t n);  // self.output.ingest_event
    void heap_push(const Event &e);                     // std::collections::BinaryHeap::push
    Event heap_pop();                                   // ... ::pop
    CodecMetadata meta_;
    std::ostream *writer_;
    EncoderType type_;
    std::vector<uint8_t> scratch_;
    // EncoderState (encoder.rs:39-53)
    double current_event_rate_ = 0.0;
    double last_event_ts_ = 0.0;
    std::vector<Event> queue_;  // BinaryHeap<Event>: `Ord for Event` is reversed on t (lib.rs:424-436) => min-heap on t
    ::AdderCompressedEncoder *compressed_ = nullptr;  // EncoderType::Compressed
};

// ---------------------------------------------------------------- reader: Decoder over RawInput
class Decoder {  // decoder.rs:22-29
  public:
    Decoder(const uint8_t *data, size_t size);  // Decoder::new_raw + decode_header (:102-203)
    // Decoder::new_compressed (:35-51): the "addec" header, then CompressedInput::digest_event (compressed/stream.rs:
    // 377-424) over the ADUs -- decoded by the compressed source of libadder_hip.so (adder_compressed_decode)
    static Decoder new_compressed(const uint8_t *data, size_t size);
    const CodecMetadata &meta() const { return meta_; }
    bool digest_event(Event *out);  // raw/stream.rs:177-201 ; false at the EOF event / end of data
    size_t position() const { return pos_; }

  private:
    Decoder() = default;
    const uint8_t *data_ = nullptr;
    size_t size_ = 0, pos_ = 0;
    CodecMetadata meta_;
    bool compressed_ = false;
    std::vector<Event> decoded_;  // compressed: all events of the stream, in stream order
};

// utils/viz.rs:76-86 -- display only; kept so that update_detect_features reads like the reference's
enum class ShowFeatureMode { Off, Instant, Hold };
struct Roi {  // video.rs:219-223: start / end coordinates, inclusive
    uint16_t start_x = 0, start_y = 0, end_x = 0, end_y = 0;
};

// ---------------------------------------------------------------- Video (video.rs:322-346)
class Video {
  public:
    // ::new :350-438 -- Video::new(plane, pixel_tree_mode, writer): framed sources pass Mode::FramePerfect
    // (framed.rs:67), the event-camera sources Mode::Continuous (prophesee.rs:65)
    Video(PlaneSize plane, std::ostream *writer /* may be null: EmptyOutput */, int device_id = -1,
          Mode pixel_tree_mode = Mode::FramePerfect);
    ~Video();
    Video(const Video &) = delete;
    Video &operator=(const Video &) = delete;

    Video &chunk_rows(size_t chunk_rows);  // :471-479
    // :493-537 -- like the reference, an invalid delta_t_max/ref_time keeps the current values
    Video &time_parameters(uint32_t tps, uint32_t ref_time, uint32_t delta_t_max, std::optional<TimeMode> time_mode);
    // :546-636
    Video &write_out(std::optional<SourceCamera> source_camera, std::optional<TimeMode> time_mode,
                     std::optional<PixelMultiMode> pixel_multi_mode, std::optional<size_t> adu_interval,
                     EncoderType encoder_type, EncoderOptions encoder_options, std::ostream *write);
    std::ostream *end_write_stream();  // :641-648
    void update_crf(uint8_t crf);      // :1241-1251
    void update_quality_manual(uint8_t c_thresh_baseline, uint8_t c_thresh_max, uint32_t delta_t_max_multiplier,
                               uint8_t c_increase_velocity, float feature_c_radius);  // :1264-1287
    // :825-837 -- feature detection on the running intensities + c_thresh reset around new features
    // (show_features / feature_cluster drive displays and are not kept)
    void update_detect_features(bool detect_features, ShowFeatureMode show_features, bool feature_rate_adjustment,
                                bool feature_cluster);
    void update_roi(std::optional<Roi> roi);  // :1291-1293
    // VideoState::features as a membership plane [h][w] (1 = feature), running_intensities [h][w][c]
    std::vector<uint8_t> feature_set();
    std::vector<uint8_t> running_intensities();
    EncoderOptions get_encoder_options() const { return encoder_->options; }
    TimeMode get_time_mode() const { return encoder_->meta().time_mode; }
    uint8_t get_event_size() const { return encoder_->meta().event_size; }
    uint32_t get_tps() const { return tps_; }
    uint32_t get_ref_time() const { return ref_time_; }
    uint32_t get_delta_t_max() const { return delta_t_max_; }
    size_t get_chunk_rows() const { return chunk_rows_; }  // VideoState::chunk_rows
    PlaneSize plane() const { return plane_; }

    // :651-778 -- one frame in; Vec<Vec<Event>> out (one vector per row chunk, raster order);
    // the events are also pushed through the encoder, exactly like the reference.
    std::vector<std::vector<Event>> integrate_matrix(const Frame &matrix, float time_spanned);
    // the same for T packed frames in one boundary call (events of all frames, frame-major)
    std::vector<Event> integrate_frames(const uint8_t *frames, uint32_t num_frames, float time_spanned,
                                        std::vector<uint64_t> *frame_offsets);
    // what the event-camera sources do with their pixels: one `integrate_for_px(px, &mut 0, frame_val, intensity,
    // time, &mut events, ..)` per step, in order (prophesee.rs:196-254); the events are also pushed through the encoder
    std::vector<Event> integrate_sparse(const std::vector<AdderSparseStep> &steps);

  private:
    void ensure_ctx();  // (re)creates the device context once every builder call has been made
    void sync_feature_controls();
    bool feature_detection_ = false, feature_rate_adjustment_ = false;
    std::optional<Roi> roi_;
    PlaneSize plane_;
    int device_id_;
    Mode pixel_tree_mode_ = Mode::FramePerfect;
    uint32_t tps_ = 7650, ref_time_ = 255, delta_t_max_ = 7650;  // VideoState/VideoStateParams::default
    size_t chunk_rows_ = 1;
    PixelMultiMode multi_mode_ = PixelMultiMode::Collapse;
    TimeMode px_time_mode_ = TimeMode::AbsoluteT;  // PixelArena::time_mode default (event_pixel_tree.rs:76)
    std::optional<uint8_t> px_c_thresh_reset_;     // pending per-pixel reset from update_crf / quality_manual
    uint32_t in_interval_count_ = 1;
    std::unique_ptr<Encoder> encoder_;
    AdderHipCtx *ctx_ = nullptr;
    std::vector<Event> buf_;
};

// ---------------------------------------------------------------- Source + Framed (video.rs:1419-1442, framed.rs)
class Source {
  public:
    virtual ~Source() = default;
    virtual std::vector<std::vector<Event>> consume() = 0;
    virtual void crf(uint8_t crf) = 0;
    virtual Video &get_video_mut() = 0;
    virtual const Video &get_video_ref() const = 0;
    virtual const Frame *get_input() const = 0;
    virtual double get_running_input_bitrate() const = 0;
};

// Stands in for video_rs_adder_dep::Decoder (ffmpeg, un-vendored, out of scope): yields
// decoded 8-bit frames [h][w][3] (or [h][w][1] for an already-gray provider).
struct FrameProvider {
    uint32_t width = 0, height = 0, channels = 3;
    float frame_rate = 30.0f;
    uint64_t frame_count = 0;
    std::function<bool(uint64_t index, Frame &out)> decode;  // false = end of stream (-> SourceError::NoData)
};

class Framed : public Source {  // framed.rs:22-39
  public:
    Framed(FrameProvider provider, bool color_input, int device_id = -1);  // ::new :44-78 (scale handled upstream)
    Framed &frame_start(uint32_t frame_idx_start);                          // :81-91
    Framed &auto_time_parameters(uint32_t ref_time, uint32_t delta_t_max, std::optional<TimeMode> time_mode);  // :94-111
    uint32_t get_ref_time() const { return video_.get_ref_time(); }
    const Frame &get_last_input_frame() const { return input_frame_; }
    // VideoBuilder (framed.rs:187-280)
    Framed &crf_builder(uint8_t crf) { video_.update_crf(crf); return *this; }
    Framed &quality_manual(uint8_t base, uint8_t max, uint32_t dtm_mult, uint8_t velocity, float radius) {
        video_.update_quality_manual(base, max, dtm_mult, velocity, radius);
        return *this;
    }
    Framed &chunk_rows(size_t n) { video_.chunk_rows(n); return *this; }
    Framed &time_parameters(uint32_t tps, uint32_t ref_time, uint32_t delta_t_max, std::optional<TimeMode> tm);
    Framed &write_out(SourceCamera source_camera, TimeMode time_mode, PixelMultiMode pixel_multi_mode,
                      std::optional<size_t> adu_interval, EncoderType encoder_type, EncoderOptions encoder_options,
                      std::ostream *write);
    // Source
    std::vector<std::vector<Event>> consume() override;  // :127-157
    void crf(uint8_t crf) override { video_.update_crf(crf); }
    Video &get_video_mut() override { return video_; }
    const Video &get_video_ref() const override { return video_; }
    const Frame *get_input() const override { return &input_frame_; }
    double get_running_input_bitrate() const override;

    uint32_t frame_idx_start = 0;
    float source_fps;

  private:
    FrameProvider cap_;
    uint64_t next_frame_ = 0;
    Frame input_frame_;
    bool color_input_;
    Video video_;
};

// ---------------------------------------------------------------- Prophesee (prophesee.rs:18-372)
// The DVS-to-ADDER source.  The `.dat` reader (parse_header / decode_event, :374-452: file parsing) is replaced by a
// provider of decoded events; the camera-side state -- last timestamp and log intensity per pixel -- lives here,
// the pixels' arenas on the device.
struct DvsEvent {  // :45-51
    uint32_t t;
    uint16_t x, y;
    uint8_t p;
};
constexpr uint32_t PROPHESEE_SOURCE_TPS = 1000000;  // :22
class Prophesee : public Source {
  public:
    // ::new :56-113 -- Video::new(plane, Continuous, None).chunk_rows(1).time_parameters(ref_time * 1e6, ref_time,
    // ref_time * 2, Some(AbsoluteT)); running_intensities = 128, last timestamps = 2, last ln = ln_1p(128 / 255)
    Prophesee(uint32_t ref_time, uint16_t width, uint16_t height, std::function<bool(DvsEvent &)> decode_event,
              int device_id = -1);
    std::vector<std::vector<Event>> consume() override;  // :116-330; throws SourceError::NoData at the end of the input
    void crf(uint8_t crf) override { video_.update_crf(crf); }
    Video &get_video_mut() override { return video_; }
    const Video &get_video_ref() const override { return video_; }
    const Frame *get_input() const override { return nullptr; }
    double get_running_input_bitrate() const override { return 0.0; }

    // the events end_events() produced when the input ran out (the reference only feeds them to the encoder)
    const std::vector<Event> &last_end_events() const { return last_end_events_; }

  private:
    void end_events();  // :332-372
    std::vector<Event> last_end_events_;
    std::function<bool(DvsEvent &)> decode_event_;
    Video video_;
    uint32_t running_t_ = 0, t_subtract_ = 0;
    std::vector<uint32_t> dvs_last_timestamps_;
    std::vector<double> dvs_last_ln_val_;
    double camera_theta_ = 0.02;  // "A fixed assumption" (:107)
};

// ---------------------------------------------------------------- Davis (transcoder/source/davis.rs)
// The DAVIS source interleaves deblurred APS frames with the DVS events between them.  Its input is what the EDI
// reconstructor (the un-vendored davis-edi-rs crate: file parsing + deblurring, not part of this path) hands out per
// `next()`: a frame of f64 intensities in [0, 1] and, in the raw modes, the DVS events before and after the frame's
// exposure with its start / end timestamps (in microseconds) and the contrast threshold c.
struct DavisDvsEvent {
    int64_t t;
    uint16_t x, y;
    bool on;
};
struct DavisPacket {                       // davis-edi-rs IterVal
    std::vector<double> frame;             // [h][w]
    bool has_events = false;               // Some((c, events_before, events_after, img_start_ts, img_end_ts))
    double c = 0.15;
    std::vector<DavisDvsEvent> events_before, events_after;
    int64_t img_start_ts = 0, img_end_ts = 0;
};
enum class TranscoderMode { Framed, RawDavis, RawDvs };  // davis.rs:41-50
class Davis : public Source {
  public:
    // ::new :109-177 -- Video::new(plane, FramePerfect | Continuous, None).chunk_rows(h / 4); `next_packet` stands for
    // reconstructor.next(with_events): false at the end of the input.  The plane's height must be a multiple of 4 (the
    // reference indexes four event chunks by y / (h / 4), :254-258).
    Davis(uint16_t width, uint16_t height, TranscoderMode mode, std::function<bool(DavisPacket &)> next_packet,
          int device_id = -1);
    // :601-897 -- the frame's events; the first call only fetches (the reference keeps one packet cached); throws
    // SourceError::NoData after the last pixel flush at the end of the input
    std::vector<std::vector<Event>> consume() override;
    void crf(uint8_t crf) override { video_.update_crf(crf); }
    Video &get_video_mut() override { return video_; }
    const Video &get_video_ref() const override { return video_; }
    const Frame *get_input() const override { return nullptr; }
    double get_running_input_bitrate() const override;  // :917-941
    // every event the source fed to the encoder, in order (DVS events, frame gaps, frames, the final flush): the
    // reference returns only the frames' events from consume() and ingests the others itself
    const std::vector<Event> &ingested() const { return ingested_; }

  private:
    std::vector<Event> integrate_dvs_events(const std::vector<DavisDvsEvent> &ev, int64_t frame_timestamp, bool check2_after,
                                            bool has_ts2, int64_t frame_timestamp_2);  // :233-466
    std::vector<Event> integrate_frame_gaps();                                          // :468-598
    std::function<bool(DavisPacket &)> next_packet_;
    Video video_;
    TranscoderMode mode_;
    bool have_cached_ = false, cached_end_ = false;
    DavisPacket cached_;
    double dvs_c_ = 0.15;
    bool have_last_after_ = false;
    std::vector<DavisDvsEvent> events_before_, events_after_, events_last_after_;
    int64_t temp_first_frame_start_timestamp_ = 0, start_of_frame_timestamp_ = 0, end_of_frame_timestamp_ = 0,
            end_of_last_frame_timestamp_ = 0;
    bool have_start_ = false, have_end_ = false, have_end_of_last_ = false;
    std::vector<int64_t> dvs_last_timestamps_;
    std::vector<double> dvs_last_ln_val_;
    double time_change_ = 0.0, ref_time_divisor_ = 1.0;
    size_t num_dvs_events_ = 0;
    std::vector<Event> ingested_;
};

// ---------------------------------------------------------------- framer (framer/driver.rs)
enum class FramerMode { INSTANTANEOUS, INTEGRATION };  // driver.rs:20-28; stored, never read by the reference's ingest: both behave alike
enum class SourceType { U8, U16, U32, U64 };           // what the Intensity view divides by (scale_intensity.rs:68-75); F32 / F64 panic there
enum class FramedViewMode { Intensity, D, DeltaT, SAE };  // video.rs:144-158

class FrameSequence;
using FrameSequenceU8 = FrameSequence;
// the T of FramerBuilder::finish::<T>() (driver.rs:126-138; T: FrameValue, scale_intensity.rs:54-209)
enum class FrameElement : uint8_t { U8 = 0, U16 = 1, U32 = 2 };

class FramerBuilder {  // driver.rs:36-147
  public:
    FramerBuilder(PlaneSize plane, size_t chunk_rows) : plane_(plane), chunk_rows_(chunk_rows) {}  // :57-75
    FramerBuilder &time_parameters(uint32_t tps, uint32_t ref_interval, uint32_t delta_t_max,
                                   std::optional<float> output_fps) {  // :77-90
        tps_ = tps; ref_interval_ = ref_interval; delta_t_max_ = delta_t_max; output_fps_ = output_fps;
        return *this;
    }
    FramerBuilder &mode(FramerMode m) { mode_ = m; return *this; }                                   // :98-102
    FramerBuilder &view_mode(FramedViewMode v) { view_mode_ = v; return *this; }                      // :104-108
    FramerBuilder &source(SourceType t, SourceCamera cam) { source_type_ = t; source_camera_ = cam; return *this; }  // :110-115
    // D view only: the reference divides by fast_math::log2_raw(255 * (delta_t_max / ref_interval) as f32) (:1020-1021),
    // a third-party approximation of log2; a caller that wants the reference's bytes passes that crate's value, the
    // default is the exact log2f
    FramerBuilder &practical_d_max(float v) { practical_d_max_ = v; return *this; }
    FramerBuilder &codec_version(uint8_t v, TimeMode tm) { codec_version_ = v; time_mode_ = tm; return *this; }  // :117-124
    FramerBuilder &device(int device_id) { device_id_ = device_id; return *this; }
    // frames the device ring holds (the reference's VecDeque grows without bound, :1066-1090); 0 = delta_t_max / tpf + 80
    FramerBuilder &ring_frames(uint32_t n) { ring_frames_ = n; return *this; }
    // :126-138.  finish() is finish::<u8>(); u16 / u32 frames are written as the big-endian bincode elements of
    // driver.rs:279,944.  (finish::<u64>() does not compile in the reference: FrameSequence<T> needs T: Into<f64>.)
    std::unique_ptr<FrameSequence> finish(FrameElement element = FrameElement::U8);

  private:
    friend class FrameSequence;
    PlaneSize plane_;
    size_t chunk_rows_;
    uint32_t tps_ = 150000, ref_interval_ = 5000, delta_t_max_ = 5000;  // FramerBuilder::new defaults (:60-65)
    std::optional<float> output_fps_;
    FramerMode mode_ = FramerMode::INSTANTANEOUS;
    FramedViewMode view_mode_ = FramedViewMode::Intensity;
    SourceType source_type_ = SourceType::U8;
    std::optional<float> practical_d_max_;
    SourceCamera source_camera_ = SourceCamera::FramedU8;
    uint8_t codec_version_ = 3;
    TimeMode time_mode_ = TimeMode::AbsoluteT;
    int device_id_ = -1;
    uint32_t ring_frames_ = 0;
};

// FrameSequence<T> (driver.rs:261-981), T = u8 / u16 / u32; the per-pixel work runs behind include/adder_framer.h
class FrameSequence {
  public:
    explicit FrameSequence(const FramerBuilder &b, FrameElement element = FrameElement::U8);
    ~FrameSequence();
    FrameSequence(const FrameSequence &) = delete;
    FrameSequence &operator=(const FrameSequence &) = delete;
    bool ingest_event(Event &event);                                          // :437-562 -> frame 0 filled?
    bool ingest_events_events(const std::vector<std::vector<Event>> &events); // :564-626
    bool flush_frame_buffer();                                                // :632-677
    bool is_frame_0_filled();                                                 // :828-843
    void write_frame_bytes(std::ostream &writer);                             // :935-962
    int write_multi_frame_bytes(std::ostream &writer);                        // :970-981
    uint32_t tpf() const;
    int64_t frames_written() const;
    size_t chunk_rows;

  private:
    AdderFramer *fr_ = nullptr;
    size_t num_chunks_ = 0, frame_units_ = 0, frame_bytes_ = 0;
    size_t width_ = 0, height_ = 0, channels_ = 0;
    std::vector<Event> flat_;
    std::vector<uint64_t> seg_offs_;
    std::vector<uint32_t> seen_;
    std::vector<uint8_t> out_;
};

// utils/simulproc.rs:33-277 -- transcode a framed source and reconstruct frames from the events
// at the same time.  The reference runs the framer on a second thread behind a channel; the
// order of operations per source frame is the same here (consume -> ingest_events_events ->
// write_multi_frame_bytes).
class SimulProcessor {
  public:
    SimulProcessor(Framed &source, uint32_t ref_time, std::ostream &frames_out, int32_t frame_max,
                   uint8_t codec_version, TimeMode time_mode, int device_id = -1);  // ::new :113-206
    void run(uint32_t frame_max);                                                    // :233-277
    int frames_written = 0;

  private:
    Framed &source_;
    std::ostream &out_;
    int32_t frame_max_;
    std::unique_ptr<FrameSequenceU8> framer_;
};

// utils/cv.rs:215-232
Frame handle_color(const Frame &input, uint32_t width, uint32_t height, uint32_t channels, bool color);

}  // namespace adder_host
