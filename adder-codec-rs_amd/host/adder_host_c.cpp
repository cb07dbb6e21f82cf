// adder_host_c.cpp -- a small extern "C" facade over the C++ host mirror so that the pytest
// suite can drive Framed -> Video -> Encoder exactly the way the reference's own transcode
// test does (adder-codec-rs/src/bin/adder_simulproc.rs:170-268).  No exception crosses it.
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>

#include "adder_host.hpp"

using namespace adder_host;

static thread_local std::string g_err;

extern "C" {

const char *adder_host_last_error() { return g_err.c_str(); }

// Framed(gray) over `frames` ([T][h][w][channels_in] u8), builder calls in the order of the
// reference's transcode test: crf -> (auto_)time_parameters -> write_out(raw, options) -> consume()
// x T -> end_write_stream.  Returns the number of events, or -1.
// encoder_type: 0 = Compressed (adu_interval reference intervals per ADU), 1 = Raw
long long adder_host_transcode(const uint8_t *frames, uint32_t num_frames, uint32_t width, uint32_t height,
                               uint32_t channels_in, int color_input, float fps, int crf /* <0: none */,
                               uint32_t ref_time, uint32_t delta_t_max, int time_mode, int multi_mode,
                               uint32_t chunk_rows, int encoder_crf /* <0: EncoderOptions::default */,
                               int encoder_type, uint32_t adu_interval, const char *out_path,
                               uint32_t *num_chunks_out) {
    try {
        FrameProvider cap;
        cap.width = width;
        cap.height = height;
        cap.channels = channels_in;
        cap.frame_rate = fps;
        cap.frame_count = num_frames;
        const size_t fsz = (size_t)width * height * channels_in;
        cap.decode = [=](uint64_t idx, Frame &out) {
            if (idx >= num_frames) return false;
            out.assign(frames + idx * fsz, frames + (idx + 1) * fsz);
            return true;
        };
        Framed source(cap, color_input != 0);
        source.chunk_rows(chunk_rows ? chunk_rows : 1);
        if (crf >= 0) source.crf_builder((uint8_t)crf);
        source.auto_time_parameters(ref_time, delta_t_max, std::nullopt);
        std::ofstream file(out_path, std::ios::binary);
        if (!file) throw SourceError(SourceError::BadParams, "cannot open output file");
        const PlaneSize plane = source.get_video_ref().plane();
        EncoderOptions opts = EncoderOptions::default_(plane);
        if (encoder_crf >= 0) opts.crf = Crf((uint8_t)encoder_crf, plane);
        source.write_out(SourceCamera::FramedU8, (TimeMode)time_mode, (PixelMultiMode)multi_mode,
                         encoder_type == 0 ? std::optional<size_t>(adu_interval) : std::nullopt,
                         encoder_type == 0 ? EncoderType::Compressed : EncoderType::Raw, opts, &file);
        long long total = 0;
        uint32_t chunks = 0;
        for (uint32_t k = 0; k < num_frames; ++k) {
            auto events = source.consume();
            chunks = (uint32_t)events.size();
            for (auto &v : events) total += (long long)v.size();
        }
        source.get_video_mut().end_write_stream();
        if (num_chunks_out) *num_chunks_out = chunks;
        return total;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// The same flow with the controls of adder-viz's transcoder tab (adder-viz/src/transcoder/adder.rs: update_crf /
// update_detect_features / update_roi on the source's Video before the frames are consumed): feature-driven rate
// control and an optional region of interest.  roi = {start_x, start_y, end_x, end_y} or null.  Writes a raw
// file; feature_set_out (may be null) receives VideoState::features as [h][w] membership bytes.
long long adder_host_transcode_features(const uint8_t *frames, uint32_t num_frames, uint32_t width, uint32_t height,
                                        uint32_t channels_in, int color_input, float fps, int crf,
                                        uint32_t ref_time, uint32_t delta_t_max, int time_mode, int multi_mode,
                                        uint32_t chunk_rows, int detect_features, int feature_rate_adjustment,
                                        const uint16_t *roi, const char *out_path, uint8_t *feature_set_out) {
    try {
        FrameProvider cap;
        cap.width = width;
        cap.height = height;
        cap.channels = channels_in;
        cap.frame_rate = fps;
        cap.frame_count = num_frames;
        const size_t fsz = (size_t)width * height * channels_in;
        cap.decode = [=](uint64_t idx, Frame &out) {
            if (idx >= num_frames) return false;
            out.assign(frames + idx * fsz, frames + (idx + 1) * fsz);
            return true;
        };
        Framed source(cap, color_input != 0);
        source.chunk_rows(chunk_rows ? chunk_rows : 1);
        source.crf_builder((uint8_t)crf);
        source.auto_time_parameters(ref_time, delta_t_max, std::nullopt);
        std::ofstream file(out_path, std::ios::binary);
        if (!file) throw SourceError(SourceError::BadParams, "cannot open output file");
        const PlaneSize plane = source.get_video_ref().plane();
        EncoderOptions opts = EncoderOptions::default_(plane);
        opts.crf = Crf((uint8_t)crf, plane);
        source.write_out(SourceCamera::FramedU8, (TimeMode)time_mode, (PixelMultiMode)multi_mode, std::nullopt,
                         EncoderType::Raw, opts, &file);
        Video &video = source.get_video_mut();
        video.update_detect_features(detect_features != 0, ShowFeatureMode::Off, feature_rate_adjustment != 0, false);
        if (roi) video.update_roi(Roi{roi[0], roi[1], roi[2], roi[3]});
        long long total = 0;
        for (uint32_t k = 0; k < num_frames; ++k)
            for (auto &v : source.consume()) total += (long long)v.size();
        if (feature_set_out) {
            const std::vector<uint8_t> fs = video.feature_set();
            memcpy(feature_set_out, fs.data(), fs.size());
        }
        video.end_write_stream();
        return total;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Prophesee::new(ref_time, file) + consume() until the input runs out (prophesee.rs), with the decoded DVS events handed
// over in memory instead of a `.dat` file: dvs = n records {t u32, x u16, y u16, p u8, pad u8}.  out receives the
// events of every consume() in order, then those of end_events(); *n_consumes = consume() calls that returned events.
long long adder_host_prophesee(const uint8_t *dvs, size_t n, uint16_t width, uint16_t height, uint32_t ref_time,
                               AdderEvent *out, size_t cap, uint32_t *n_consumes) {
    try {
        size_t pos = 0;
        auto decode = [&](DvsEvent &e) {
            if (pos >= n) return false;
            const uint8_t *r = dvs + pos * 10;
            memcpy(&e.t, r, 4);
            memcpy(&e.x, r + 4, 2);
            memcpy(&e.y, r + 6, 2);
            e.p = r[8];
            ++pos;
            return true;
        };
        Prophesee source(ref_time, width, height, decode);
        long long total = 0;
        uint32_t calls = 0;
        auto take = [&](const std::vector<Event> &v) {
            for (const Event &e : v) {
                if ((size_t)total < cap) out[total] = e;
                ++total;
            }
        };
        for (;;) {
            try {
                for (auto &v : source.consume()) take(v);
                ++calls;
            } catch (const SourceError &err) {
                if (err.kind != SourceError::NoData) throw;
                take(source.last_end_events());
                break;
            }
        }
        if (n_consumes) *n_consumes = calls;
        return total;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Davis::new + consume() until the input runs out (davis.rs), over EDI reconstructor output handed over in memory:
// frames = [P][h * w] f64; meta = [P] records {c f64, img_start_ts i64, img_end_ts i64, n_before u32, n_after u32};
// dvs = the packets' before-events then after-events, one packet after the other, as {t i64, x u16, y u16, on u8, pad[3]}.
// out receives every event the source ingests, in order; returned = their number; *n_returned = events consume() returned.
long long adder_host_davis(const double *frames, const uint8_t *meta, const uint8_t *dvs, uint32_t packets, uint16_t width,
                           uint16_t height, int mode, uint32_t tps, uint32_t ref_time, uint32_t delta_t_max, int time_mode,
                           int crf /* <0: none */, AdderEvent *out, size_t cap, unsigned long long *n_returned) {
    try {
        uint32_t pos = 0;
        size_t dvs_pos = 0;
        const size_t n = (size_t)width * height;
        auto read_events = [&](uint32_t count, std::vector<DavisDvsEvent> &dst) {
            dst.resize(count);
            for (uint32_t i = 0; i < count; ++i, ++dvs_pos) {
                const uint8_t *r = dvs + dvs_pos * 16;
                memcpy(&dst[i].t, r, 8);
                memcpy(&dst[i].x, r + 8, 2);
                memcpy(&dst[i].y, r + 10, 2);
                dst[i].on = r[12] != 0;
            }
        };
        auto next = [&](DavisPacket &pk) {
            if (pos >= packets) return false;
            pk.frame.assign(frames + (size_t)pos * n, frames + (size_t)(pos + 1) * n);
            const uint8_t *m = meta + (size_t)pos * 32;
            uint32_t nb, na;
            memcpy(&pk.c, m, 8);
            memcpy(&pk.img_start_ts, m + 8, 8);
            memcpy(&pk.img_end_ts, m + 16, 8);
            memcpy(&nb, m + 24, 4);
            memcpy(&na, m + 28, 4);
            pk.has_events = mode != 0;
            read_events(nb, pk.events_before);
            read_events(na, pk.events_after);
            ++pos;
            return true;
        };
        Davis source(width, height, mode == 0 ? TranscoderMode::Framed : mode == 1 ? TranscoderMode::RawDavis : TranscoderMode::RawDvs,
                     next);
        source.get_video_mut().time_parameters(tps, ref_time, delta_t_max, time_mode == 1 ? TimeMode::AbsoluteT : TimeMode::DeltaT);
        if (crf >= 0) source.crf((uint8_t)crf);
        unsigned long long returned = 0;
        for (;;) {
            try {
                for (auto &v : source.consume()) returned += v.size();
            } catch (const SourceError &err) {
                if (err.kind != SourceError::NoData) throw;
                break;
            }
        }
        const std::vector<Event> &all = source.ingested();
        for (size_t i = 0; i < all.size() && i < cap; ++i) out[i] = all[i];
        if (n_returned) *n_returned = returned;
        return (long long)all.size();
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

long long adder_host_transcode_raw(const uint8_t *frames, uint32_t num_frames, uint32_t width, uint32_t height,
                                   uint32_t channels_in, int color_input, float fps, int crf /* <0: none */,
                                   uint32_t ref_time, uint32_t delta_t_max, int time_mode, int multi_mode,
                                   uint32_t chunk_rows, int encoder_crf /* <0: EncoderOptions::default */,
                                   const char *out_path, uint32_t *num_chunks_out) {
    return adder_host_transcode(frames, num_frames, width, height, channels_in, color_input, fps, crf, ref_time,
                                delta_t_max, time_mode, multi_mode, chunk_rows, encoder_crf, 1, 0, out_path,
                                num_chunks_out);
}

// The reference's `dark` test end to end (src/bin/adder_simulproc.rs:170-268): Framed(gray) ->
// crf(0) -> frame_start -> time_parameters((ref_time * fps) as u32, ref_time, dtm, None) ->
// write_out(FramedU8, time_mode, multi_mode, None, Raw, {crf: Crf::new(Some(0))}) ->
// SimulProcessor::new::<u8>(source, ref_time, frames path, frame_count_max, 1, 1, TimeMode::default())
// -> run(0).  Writes the event file and the reconstructed frames; returns the frames written, or -1.
long long adder_host_simulproc(const uint8_t *frames, uint32_t num_frames, uint32_t width, uint32_t height, float fps,
                               int crf, uint32_t ref_time, uint32_t delta_t_max, int time_mode, int multi_mode,
                               uint32_t chunk_rows, int32_t frame_count_max, uint8_t framer_codec_version,
                               int framer_time_mode, const char *out_events_path, const char *out_frames_path) {
    try {
        FrameProvider cap;
        cap.width = width;
        cap.height = height;
        cap.channels = 1;
        cap.frame_rate = fps;
        cap.frame_count = num_frames;
        const size_t fsz = (size_t)width * height;
        cap.decode = [=](uint64_t idx, Frame &out) {
            if (idx >= num_frames) return false;
            out.assign(frames + idx * fsz, frames + (idx + 1) * fsz);
            return true;
        };
        Framed source(cap, false);
        source.chunk_rows(chunk_rows ? chunk_rows : 1);
        source.crf_builder((uint8_t)crf);
        source.time_parameters((uint32_t)((double)ref_time * (double)source.source_fps), ref_time, delta_t_max,
                               std::nullopt);
        std::ofstream ev_file(out_events_path, std::ios::binary);
        std::ofstream fr_file(out_frames_path, std::ios::binary);
        if (!ev_file || !fr_file) throw SourceError(SourceError::BadParams, "cannot open output file");
        const PlaneSize plane = source.get_video_ref().plane();
        EncoderOptions opts = EncoderOptions::default_(plane);
        opts.crf = Crf((uint8_t)crf, plane);
        source.write_out(SourceCamera::FramedU8, (TimeMode)time_mode, (PixelMultiMode)multi_mode, std::nullopt,
                         EncoderType::Raw, opts, &ev_file);
        SimulProcessor proc(source, source.get_ref_time(), fr_file, frame_count_max, framer_codec_version,
                            (TimeMode)framer_time_mode);
        proc.run(0);
        return proc.frames_written;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// FramerBuilder -> FrameSequence<u8> over events in memory: one ingest_events_events call (chunk_offsets has
// n_chunks + 1 entries, the framer's chunk division), write_multi_frame_bytes, then `flushes` rounds of
// flush_frame_buffer + write_frame_bytes.  params = {tps, ref_interval, delta_t_max, codec_version, time_mode,
// framer mode, view mode, source type, chunk_rows, frame element type (0 u8 / 1 u16 / 2 u32)}.  Returns the bytes
// written to out (needs <= cap), or -1.
long long adder_host_frame_events(const AdderEvent *events, const uint64_t *chunk_offsets, uint32_t n_chunks,
                                  uint16_t width, uint16_t height, uint8_t channels, const uint32_t *params,
                                  float output_fps, float practical_d_max, uint32_t flushes, uint8_t *out, size_t cap) {
    try {
        PlaneSize plane(width, height, channels);
        FramerBuilder b(plane, params[8]);
        b.time_parameters(params[0], params[1], params[2], output_fps > 0.0f ? std::optional<float>(output_fps) : std::nullopt)
            .codec_version((uint8_t)params[3], (TimeMode)params[4])
            .mode((FramerMode)params[5])
            .view_mode((FramedViewMode)params[6])
            .source((SourceType)params[7], SourceCamera::FramedU8)
            .ring_frames(1u << 14);
        if (practical_d_max > 0.0f) b.practical_d_max(practical_d_max);
        auto fr = b.finish((FrameElement)params[9]);
        std::vector<std::vector<Event>> chunks(n_chunks);
        for (uint32_t k = 0; k < n_chunks; ++k) chunks[k].assign(events + chunk_offsets[k], events + chunk_offsets[k + 1]);
        std::ostringstream os;
        if (fr->ingest_events_events(chunks)) fr->write_multi_frame_bytes(os);
        for (uint32_t i = 0; i < flushes; ++i) {
            fr->flush_frame_buffer();
            fr->write_frame_bytes(os);
        }
        const std::string bytes = os.str();
        if (bytes.size() > cap) throw SourceError(SourceError::BadParams, "output buffer too small");
        memcpy(out, bytes.data(), bytes.size());
        return (long long)bytes.size();
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Decoder over a raw stream in memory: fills meta[10] = {version, width, height, channels, tps,
// ref_interval, delta_t_max, event_size, source_camera|time_mode<<8, adu_interval} and up to cap
// events; returns the number of events in the stream, or -1.
long long adder_host_decode_raw(const uint8_t *data, size_t size, uint32_t *meta, AdderEvent *events, size_t cap) {
    try {
        // (Decoder::new_raw or Decoder::new_compressed by the magic, as the reference's players do: adder-viz
        // player/adder.rs tries the compressed decoder first)
        Decoder dec = size >= 5 && memcmp(data, "addec", 5) == 0 ? Decoder::new_compressed(data, size) : Decoder(data, size);
        const CodecMetadata &m = dec.meta();
        meta[0] = m.codec_version;
        meta[1] = m.plane.w();
        meta[2] = m.plane.h();
        meta[3] = m.plane.c();
        meta[4] = m.tps;
        meta[5] = m.ref_interval;
        meta[6] = m.delta_t_max;
        meta[7] = m.event_size;
        meta[8] = (uint32_t)m.source_camera | ((uint32_t)m.time_mode << 8);
        meta[9] = (uint32_t)m.adu_interval;
        long long n = 0;
        Event e;
        while (dec.digest_event(&e)) {
            if ((size_t)n < cap) events[n] = e;
            ++n;
        }
        return n;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Crf table / EncoderOptions::default checks (no device needed): out[4] = baseline, max, velocity, radius
int adder_host_crf_parameters(int crf /* <0: default */, uint16_t width, uint16_t height, uint32_t *out) {
    try {
        Crf c(crf < 0 ? std::nullopt : std::optional<uint8_t>((uint8_t)crf), PlaneSize(width, height, 1));
        const CrfParameters &p = c.get_parameters();
        out[0] = p.c_thresh_baseline;
        out[1] = p.c_thresh_max;
        out[2] = p.c_increase_velocity;
        out[3] = p.feature_c_radius;
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Encoder over an in-memory writer: header + events + (optionally) EOF -> bytes.  Returns size or -1.
long long adder_host_encode_raw(uint8_t codec_version, uint16_t width, uint16_t height, uint8_t channels,
                                uint32_t tps, uint32_t ref_interval, uint32_t delta_t_max, uint32_t source_camera,
                                uint32_t time_mode, const AdderEvent *events, size_t n, int close, uint8_t *dst,
                                size_t cap) {
    try {
        std::ostringstream os;
        CodecMetadata meta;
        meta.codec_version = codec_version;
        meta.plane = PlaneSize(width, height, channels);
        meta.tps = tps;
        meta.ref_interval = ref_interval;
        meta.delta_t_max = delta_t_max;
        meta.source_camera = (SourceCamera)source_camera;
        meta.time_mode = (TimeMode)time_mode;
        meta.adu_interval = 0;
        Encoder enc = Encoder::new_raw(meta, &os, EncoderOptions::default_(meta.plane));
        enc.ingest_events(events, n);
        if (close) enc.close_writer();
        const std::string s = os.str();
        if (s.size() > cap) throw CodecError(CodecError::Io, "destination too small");
        memcpy(dst, s.data(), s.size());
        return (long long)s.size();
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Encoder with EventOrder::Interleaved and/or EventDrop::Manual over an in-memory writer: ingests `n`
// events (for Manual, event i arrives at clock_s[i] seconds; the encoder is created at clock 0), closes the
// writer and copies the file bytes out.  Returns the byte count, or -1.
long long adder_host_encode_events(const AdderEvent *events, size_t n, uint16_t width, uint16_t height, uint8_t channels,
                                   uint32_t delta_t_max, int interleaved, int manual_drop, double target_event_rate,
                                   double alpha, const double *clock_s, uint8_t *out, size_t out_cap,
                                   size_t *still_queued) {
    try {
        CodecMetadata meta;
        meta.plane = PlaneSize(width, height, channels);
        meta.delta_t_max = delta_t_max;
        meta.time_mode = TimeMode::AbsoluteT;
        std::ostringstream os(std::ios::binary);
        EncoderOptions opts = EncoderOptions::default_(meta.plane);
        if (interleaved) opts.event_order = EventOrder::Interleaved;
        if (manual_drop) opts.event_drop = EventDrop::manual(target_event_rate, alpha);
        size_t idx = 0;
        double now = 0.0;
        Encoder enc = Encoder::new_raw(meta, &os, opts);
        enc.clock = [&]() { return now; };
        // (the constructor stamped last_event_ts with the steady clock; the first ingest below overrides
        // the notion of "now", and t_diff of the first event is measured from 0 as the test's model does)
        enc.reset_clock_origin(0.0);
        for (idx = 0; idx < n; ++idx) {
            now = clock_s ? clock_s[idx] : 0.0;
            enc.ingest_event(events[idx]);
        }
        if (still_queued) *still_queued = enc.queued();
        enc.close_writer();
        const std::string bytes = os.str();
        if (bytes.size() > out_cap) throw CodecError(CodecError::Io, "output buffer too small");
        memcpy(out, bytes.data(), bytes.size());
        return (long long)bytes.size();
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}
}
