// adder_raw_sink.cpp -- raw `.adder` wire form, host side of the path's sink.
//
// Byte-exact restatement of what the reference's Encoder + RawOutput write:
//   header  : adder-codec-core/src/codec/encoder.rs:170-229, header.rs:14-25,52-84
//             (bincode fixint, big-endian: 25-byte base + V1/V2/V3 u32 extensions)
//   events  : raw/stream.rs:101-120 (EventSingle 9 B for 1-channel planes, Event 11 B
//             otherwise; Option<u8> c = 0x01 + value, or a lone 0x00 for None)
//   EOF     : raw/stream.rs:79-92 (always the 11-byte Event form)
#include <string.h>

#include "../../include/adder_hip.h"

static inline uint8_t *put16(uint8_t *p, uint16_t v) {
    p[0] = (uint8_t)(v >> 8);
    p[1] = (uint8_t)v;
    return p + 2;
}
static inline uint8_t *put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
    return p + 4;
}

extern "C" size_t adder_raw_header(uint8_t *dst, uint8_t codec_version, uint16_t width, uint16_t height,
                                   uint8_t channels, uint32_t tps, uint32_t ref_interval, uint32_t delta_t_max,
                                   uint32_t source_camera, uint32_t time_mode, uint32_t adu_interval) {
    uint8_t *p = dst;
    memcpy(p, "adder", 5);  // MAGIC_RAW, header.rs:5
    p += 5;
    *p++ = codec_version;
    *p++ = 98;  // 'b': big endian, header.rs:69
    p = put16(p, width);
    p = put16(p, height);
    p = put32(p, tps);
    p = put32(p, ref_interval);
    p = put32(p, delta_t_max);
    *p++ = channels == 1 ? 9 : 11;  // event_size, header.rs:76-81
    *p++ = channels;
    if (codec_version >= 1) p = put32(p, source_camera);  // SourceCamera as u32 variant index
    if (codec_version >= 2) p = put32(p, time_mode);
    if (codec_version >= 3) p = put32(p, adu_interval);
    return (size_t)(p - dst);
}

extern "C" size_t adder_raw_events(uint8_t *dst, const AdderEvent *ev, size_t n, uint8_t channels) {
    uint8_t *p = dst;
    if (channels == 1) {
        for (size_t i = 0; i < n; ++i) {
            p = put16(p, ev[i].x);
            p = put16(p, ev[i].y);
            *p++ = ev[i].d;
            p = put32(p, ev[i].t);
        }
    } else {
        for (size_t i = 0; i < n; ++i) {
            p = put16(p, ev[i].x);
            p = put16(p, ev[i].y);
            if (ev[i].c == ADDER_C_NONE) {
                *p++ = 0;
            } else {
                *p++ = 1;
                *p++ = ev[i].c;
            }
            *p++ = ev[i].d;
            p = put32(p, ev[i].t);
        }
    }
    return (size_t)(p - dst);
}

extern "C" size_t adder_raw_eof(uint8_t *dst) {
    static const uint8_t eof[11] = {0xff, 0xff, 0xff, 0xff, 0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00};
    memcpy(dst, eof, sizeof eof);
    return sizeof eof;
}
