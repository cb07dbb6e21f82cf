// adder_framer.hpp -- per-pixel-channel step of the instantaneous framer (events -> u8 frames).
//
// One call = what ingest_event_for_chunk (adder-codec-rs/src/framer/driver.rs:984-1133) does to
// the pixel's trackers for one event of that pixel: advance the pixel's clock, decide which
// output frames the event covers, compute the intensity they get
// (<u8 as FrameValue>::get_frame_value, Intensity arm, framer/scale_intensity.rs:54-72).
// A pixel's frames are filled front to back, each exactly once, so the reference's
// Option<u8> + filled_count bookkeeping reduces to `last_filled` per pixel: frame f holds a
// value for the pixel iff last_filled >= f, and frame f is complete iff min over pixels of
// last_filled >= f.  Compiled for the device and, by tests/cpu_sim, for the host.
#pragma once
#include <stdint.h>

#include "adder_pixel.hpp"  // ADDER_HD, frame_value_u8

namespace adder {

// Exact n / d for every u32 n and an invariant d >= 1 (Granlund & Montgomery's round-up method): the two
// divisions of the step are by per-stream constants, and a u32 division is a ~40-instruction routine on the GPU.
struct FastDivU32 {
    uint32_t d, magic, sh1, sh2;
};
ADDER_HD FastDivU32 fast_div_make(uint32_t d) {
    FastDivU32 f;
    f.d = d;
    uint32_t l = 0;
    while (l < 32u && ((uint64_t)1 << l) < (uint64_t)d) ++l;  // ceil(log2 d)
    f.magic = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1u);
    f.sh1 = l < 1u ? l : 1u;
    f.sh2 = l > 1u ? l - 1u : 0u;
    return f;
}
ADDER_HD uint32_t fast_div(uint32_t n, const FastDivU32 &f) {
    const uint32_t t1 = (uint32_t)(((uint64_t)f.magic * n) >> 32);
    return (t1 + ((n - t1) >> f.sh1)) >> f.sh2;
}

struct FramerConsts {
    uint32_t tpf;           // ticks per output frame (driver.rs:357-361)
    uint32_t ref_interval;  // ticks per source frame; also the `tpf` argument of get_frame_value (:1034)
    uint32_t abs_t;         // codec_version >= 2 && TimeMode::AbsoluteT (:1001, :1024-1030)
    uint32_t round_up;      // codec_version >= 1 && framed source camera (:1093-1107)
    FastDivU32 by_tpf, by_ref;
    // what a frame's byte shows (<u8 as FrameValue>::get_frame_value, scale_intensity.rs:54-109)
    uint32_t view_mode;      // FramedViewMode: 0 Intensity, 1 D, 2 DeltaT, 3 SAE (video.rs:144-158)
    uint32_t source_type;    // SourceType of the intensities: 0 U8, 1 U16, 2 U32, 3 U64 (Intensity view)
    float practical_d_max;   // D view: log2_raw(255 * (delta_t_max / ref_interval)), computed by the caller (:1020-1021)
    uint32_t delta_t_max;    // DeltaT / SAE views
    uint32_t value_type;     // the frame element type T of FrameSequence<T>: 0 u8, 1 u16, 2 u32 (scale_intensity.rs:54-209)
};
constexpr uint32_t kViewIntensity = 0, kViewD = 1, kViewDeltaT = 2, kViewSae = 3;

ADDER_HD FramerConsts framer_consts(uint32_t tpf, uint32_t ref_interval, uint32_t abs_t, uint32_t round_up,
                                    uint32_t view_mode = 0, uint32_t source_type = 0, float practical_d_max = 0.0f,
                                    uint32_t delta_t_max = 0, uint32_t value_type = 0) {
    FramerConsts k;
    k.tpf = tpf;
    k.ref_interval = ref_interval;
    k.abs_t = abs_t;
    k.round_up = round_up;
    k.by_tpf = fast_div_make(tpf);
    k.by_ref = fast_div_make(ref_interval);
    k.view_mode = view_mode;
    k.source_type = source_type;
    k.practical_d_max = practical_d_max;
    k.delta_t_max = delta_t_max;
    k.value_type = value_type;
    return k;
}

// <u8 as FrameValue>::get_frame_value for an event that sets a pixel's intensity (driver.rs:1017-1044): `te` = event.t
// as the framer hands it over (AbsoluteT streams: minus the pixel's previous clock, except in the SAE view, :1022-1028),
// clock / prev_clock = the pixel's running timestamp after / before the event, as u32 (`as DeltaT`).
ADDER_HD uint32_t framer_value_u8(uint32_t d, uint32_t te, uint32_t clock, uint32_t prev_clock, const FramerConsts &k) {
    if (k.view_mode == kViewD) return f32_as_u8((float)d / k.practical_d_max * 255.0f);
    if (k.view_mode == kViewDeltaT) return f32_as_u8((float)te / (float)k.delta_t_max * 255.0f);
    if (k.view_mode == kViewSae) return f32_as_u8((float)(clock - prev_clock) / (float)k.delta_t_max * 255.0f);
    const double intensity = event_intensity_f64(d, te);
    const double tpf = (double)k.ref_interval;
    if (k.source_type == 0u) return f64_as_u8(intensity * tpf);
    const double full = k.source_type == 1u ? 65535.0 : k.source_type == 2u ? 4294967295.0 : 18446744073709551616.0;
    return f64_as_u8(intensity / full * tpf * 255.0);
}

// Rust's `as u16` / `as u32` from a float: saturating, NaN -> 0
ADDER_HD uint32_t f64_as_uint(double v, double max) {
    if (!(v > 0.0)) return 0u;
    if (v >= max) return (uint32_t)max;
    return (uint32_t)v;
}
ADDER_HD uint32_t f32_as_uint(float v, double max) {
    if (!(v > 0.0f)) return 0u;
    if ((double)v >= max) return (uint32_t)max;
    return (uint32_t)v;
}
// <u16 as FrameValue>::get_frame_value (scale_intensity.rs:111-160) and <u32 ..> (:162-209): the same expressions as u8's
// with the target type's maximum (f32::from(u16::MAX) = 65535, u32::MAX as f32 = 2^32 after rounding).  Their SAE arm is
// todo!() in the reference: the C-ABI refuses that combination at create.  FrameSequence<u64> cannot be instantiated in
// the reference (its methods need T: Into<f64>).
ADDER_HD uint32_t framer_value_wide(uint32_t d, uint32_t te, const FramerConsts &k) {
    const double tmax = k.value_type == 1u ? 65535.0 : 4294967295.0;
    const float tmax_f32 = k.value_type == 1u ? 65535.0f : 4294967296.0f;
    if (k.view_mode == kViewD) return f32_as_uint((float)d / k.practical_d_max * tmax_f32, tmax);
    if (k.view_mode == kViewDeltaT) return f32_as_uint((float)te / (float)k.delta_t_max * tmax_f32, tmax);
    const double intensity = event_intensity_f64(d, te);
    const double tpf = (double)k.ref_interval;
    if (k.source_type == k.value_type) return f64_as_uint(intensity * tpf, tmax);
    const double smax = k.source_type == 0u ? 255.0 : k.source_type == 1u ? 65535.0 : k.source_type == 2u ? 4294967295.0 : 18446744073709551616.0;
    return f64_as_uint(intensity / smax * tpf * tmax, tmax);
}
ADDER_HD uint32_t framer_value(uint32_t d, uint32_t te, uint32_t clock, uint32_t prev_clock, const FramerConsts &k) {
    return k.value_type == 0u ? framer_value_u8(d, te, clock, prev_clock, k) : framer_value_wide(d, te, k);
}

struct FramerPx {
    uint64_t ts;     // pixel_ts_tracker
    int32_t lastf;   // last_filled_tracker (-1: none)
    uint32_t lasti;  // last_frame_intensity_tracker
};

static_assert(sizeof(FramerPx) == 16, "one 16-byte record per unit");

constexpr int32_t kFramerMaxFrame = 0x7ffffff0;

// Returns true when frames (fill_from, fill_to] (absolute indices) take the value p.lasti.
// `overflow` is raised when the frame index leaves the supported range.
ADDER_HD bool framer_step(FramerPx &p, uint32_t d, uint32_t t, const FramerConsts &k, int32_t &fill_from,
                          int32_t &fill_to, bool &overflow) {
    const uint64_t prev_ts = p.ts;
    bool fills = false;
    if (k.abs_t) {
        if (prev_ts >= (uint64_t)t) return false;  // an event from the pixel's past (:1002-1007)
        p.ts = t;
    } else {
        p.ts = prev_ts + (uint64_t)t;
    }
    const uint64_t rm1 = p.ts ? p.ts - 1u : 0u;  // saturating_sub(1)
    // 64-bit clocks, but they stay below 2^32 for ~150 hours of 30 fps video at 255 ticks per
    // frame: divide in 32 bits then (a 64-bit division is a ~130-instruction routine on the GPU)
    const bool small = (p.ts >> 32) == 0u;
    const uint64_t q = small ? (uint64_t)fast_div((uint32_t)rm1, k.by_tpf) : rm1 / (uint64_t)k.tpf;
    if (q > (uint64_t)kFramerMaxFrame) {
        overflow = true;
    } else if ((int64_t)q > (int64_t)p.lastf) {
        if (d != 255u) {  // D_EMPTY repeats the last intensity (:1017-1019)
            uint32_t te = t;
            if (k.abs_t && k.view_mode != kViewSae) {
                const uint32_t pr = (uint32_t)prev_ts;
                te = t > pr ? t - pr : 0u;  // event.t.saturating_sub(prev_running_ts as u32)
            }
            p.lasti = framer_value(d, te, (uint32_t)p.ts, (uint32_t)prev_ts, k);
        }
        fill_from = p.lastf;
        fill_to = (int32_t)q;
        p.lastf = (int32_t)q;
        fills = true;
    }
    if (k.round_up) {
        if (small) {
            const uint32_t ts32 = (uint32_t)p.ts, qr = fast_div(ts32, k.by_ref);
            if (ts32 - qr * k.ref_interval > 0u) p.ts = ((uint64_t)qr + 1u) * (uint64_t)k.ref_interval;
        } else if (p.ts % (uint64_t)k.ref_interval > 0u) {
            p.ts = (p.ts / (uint64_t)k.ref_interval + 1u) * (uint64_t)k.ref_interval;
        }
    }
    return fills;
}

}  // namespace adder
