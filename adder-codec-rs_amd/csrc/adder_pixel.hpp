// adder_pixel.hpp -- per-pixel-channel FramePerfect integrate / decimate / fire step.
//
// Device code of the MI355X path (included by adder_kernels.hip).  It is written as
// __host__ __device__ so that tests/cpu_sim can compile the *same* functions with g++
// and diff them against the oracle without a GPU; nothing in the product calls the
// host instantiation.
//
// What it computes is what the reference's integrate_for_px does for one pixel and
// one frame (adder-codec-rs/src/transcoder/source/video.rs:1318-1380, driving
// PixelArena::{pop_best_events:213-287, integrate:317-413, integrate_main:418-479,
// pop_top_event:139-210, delta_t_to_absolute_t:113-137} of
// adder-codec-rs/src/transcoder/event_pixel_tree.rs), for Mode::FramePerfect only
// (the mode every framed source uses, framed.rs:67).
//
// STATE REPRESENTATION.  The reference keeps a SmallVec arena of `length` nodes.  In
// FramePerfect mode the arena always has the shape
//        [ fired_0, fired_1, ..., fired_{m-1}, tail ]        (length = m + 1)
// where every fired_k carries a best_event and the tail carries none (a node gets
// a best_event exactly when it fires, and firing creates a new tail behind it:
// event_pixel_tree.rs:342-356; pop_top shifts left :199-207; pop_best leaves only
// the old tail or a fresh node :249-273).  A tail with integration == 0 and
// delta_t == 0 is "pristine": its d is rewritten from the next intensity before
// it is used (:332-335), so nothing about it needs storing.  We therefore keep
//   hdr    : base_val | c_thresh<<8 | c_increase_counter<<16 | flags<<24
//            flags = m (5 bits) | popped_dtm<<5 | tail_live<<6
//   tail   : (d, integration, delta_t), meaningful iff tail_live
//   level k: (d, integration, delta_t, best_d, best_delta_t) for k < m
//   last_fired_t (AbsoluteT only);  running_t is identical for every pixel and is
//   passed in.
// need_to_pop_top is never set between frames (integrate_for_px pops at its end),
// dtm_reached is recomputed by every integrate, `alt` is only asserted on.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ADDER_HD __host__ __device__ __forceinline__
#else
#define ADDER_HD inline
#endif

namespace adder {

constexpr uint32_t kDMax = 127;
constexpr uint32_t kDZero = 128;
constexpr uint32_t kDEmpty = 255;

constexpr uint32_t kFlagMMask = 0x1f;
constexpr uint32_t kFlagPopped = 0x20;
constexpr uint32_t kFlagTailLive = 0x40;

constexpr uint32_t kMaxDepthLimit = 31;

// Uniform (per launch) constants of one frame step.
struct StepConsts {
    float time_spanned;  // `time` of PixelArena::integrate
    float running_t;     // PixelArena::running_t BEFORE this frame's integrate
    float dtm_f;         // delta_t_max as f32 (event_pixel_tree.rs:394)
    uint32_t ref_time;
    uint32_t c_thresh_max;
    uint32_t velocity_m1;  // (c_increase_velocity - 1) as u8
    uint32_t c_inc;        // ((time as u32) / ref_time) as u8   (:408-410)
    uint32_t collapse;     // PixelMultiMode::Collapse
    uint32_t abs_t;        // TimeMode::AbsoluteT
    uint32_t max_depth;    // stored levels available
};

struct Node {
    float integ;
    float dt;
    float bdt;    // best_event.delta_t
    uint32_t d;
    uint32_t bd;  // best_event.d
};

// Always-resident part of one pixel-channel.
struct PxState {
    uint32_t hdr;
    Node n0;  // level 0, valid iff m > 0
    float tinteg, tdt;
    uint32_t td;
    float lastf;
};

ADDER_HD float bits_to_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
ADDER_HD uint32_t f32_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// D_SHIFT_F32[d]: 2^d for d < 128, 0 for d == 128 (adder-codec-core/src/lib.rs:220-235)
ADDER_HD float pow2_d(uint32_t d) { return d >= 128u ? 0.0f : bits_to_f32((d + 127u) << 23); }

// get_d_from_intensity (event_pixel_tree.rs:482-499): floor(log2(trunc(x))) clamped to
// D_MAX, 128 if x < 1.  For x >= 1 that is the unbiased binary32 exponent.
ADDER_HD uint32_t get_d(float x) {
    if (x < 1.0f) return kDZero;
    uint32_t e = ((f32_to_bits(x) >> 23) & 0xffu) - 127u;
    return e > kDMax ? kDMax : e;
}

// rustc `f32 as u32`: truncate toward zero, saturate, NaN -> 0
ADDER_HD uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

// Unfused multiply / add / correctly rounded divide.  The reference is compiled by
// rustc, which never contracts a*b+c; hipcc would (-ffp-contract=fast) unless told
// otherwise, so the one rounding-sensitive expression (:431,:445) uses these.
#if defined(__HIP_DEVICE_COMPILE__)
ADDER_HD float fmul(float a, float b) { return __fmul_rn(a, b); }
ADDER_HD float fadd(float a, float b) { return __fadd_rn(a, b); }
ADDER_HD float fsub(float a, float b) { return __fsub_rn(a, b); }
ADDER_HD float fdiv(float a, float b) { return __fdiv_rn(a, b); }
#else
ADDER_HD float fmul(float a, float b) { return a * b; }
ADDER_HD float fadd(float a, float b) { return a + b; }
ADDER_HD float fsub(float a, float b) { return a - b; }
ADDER_HD float fdiv(float a, float b) { return a / b; }
#endif

// integrate_main (event_pixel_tree.rs:418-479), FramePerfect.  Returns true if the
// node fired (the caller then creates the fresh child / truncates the arena).
ADDER_HD bool node_integrate(Node &n, float intensity, float time) {
    const float s = fadd(n.integ, intensity);
    if (s >= pow2_d(n.d)) {
        const uint32_t nd = get_d(s);
        float prop = fdiv(fsub(pow2_d(nd), n.integ), intensity);
        if (nd == kDZero || n.d == kDZero || intensity < 1.1920929e-7f) prop = 1.0f;
        n.bd = nd;
        n.bdt = fadd(n.dt, fmul(time, prop));
        if (nd < kDMax) {
            n.integ = s;
            n.dt = fadd(n.dt, time);
            // smallest k > nd with 2^k > trunc(integration): nd = floor(log2(s)) so k = nd+1
            n.d = nd + 1u;
        } else {
            n.d = nd;
        }
        return true;
    }
    n.integ = s;
    n.dt = fadd(n.dt, time);
    return false;
}

// delta_t_to_absolute_t (event_pixel_tree.rs:113-137), FramePerfect.  Returns the
// event's `t` and updates last_fired_t.
ADDER_HD uint32_t event_time(float ev_dt, float &lastf, const StepConsts &sc) {
    if (sc.abs_t) {
        ev_dt = fadd(ev_dt, lastf);
        lastf = ev_dt;
        const uint32_t lf = f32_as_u32(lastf);
        const uint32_t q = lf / sc.ref_time;
        const uint32_t r = lf - q * sc.ref_time;
        lastf = (r == 0u) ? (float)lf : (float)((q + 1u) * sc.ref_time);
    }
    return f32_as_u32(ev_dt);
}

ADDER_HD bool contrast_exceeded(uint32_t v, uint32_t hdr) {
    const uint32_t base = hdr & 0xffu;
    const uint32_t cth = (hdr >> 8) & 0xffu;
    const uint32_t lo = base > cth ? base - cth : 0u;  // saturating_sub
    uint32_t hi = base + cth;                          // saturating_add
    hi = hi > 255u ? 255u : hi;
    return v < lo || v > hi;  // video.rs:1338-1340
}

// Phase A: how many events will this pixel emit this frame?  Needs only the resident
// part of the state (hdr, level 0, tail), so every lane can run it before the block's
// ordered compaction assigns output positions.
ADDER_HD uint32_t plan_count(const PxState &s, uint32_t v, const StepConsts &sc) {
    const float I = (float)v;
    const uint32_t flags = s.hdr >> 24;
    const uint32_t m = flags & kFlagMMask;
    bool popped = (flags & kFlagPopped) != 0u;
    const bool live = (flags & kFlagTailLive) != 0u;
    float tinteg = live ? s.tinteg : 0.0f;
    float tdt = live ? s.tdt : 0.0f;

    uint32_t count = 0;
    float r_integ, r_dt;
    uint32_t r_d;
    bool r_from_tail;
    if (contrast_exceeded(v, s.hdr)) {
        const uint32_t tz = (tinteg == 0.0f && tdt > 0.0f) ? 1u : 0u;
        const uint32_t nloc = m + tz;
        if (popped && sc.collapse && nloc > 0u) {
            count = 2u;
            tinteg = 0.0f;
            tdt = 0.0f;
        } else {
            count = nloc;
            if (tz) tdt = 0.0f;
        }
        popped = false;
        r_from_tail = true;
    } else {
        r_from_tail = (m == 0u);
    }
    if (r_from_tail) {
        r_integ = tinteg;
        r_dt = tdt;
        r_d = (tinteg == 0.0f && tdt == 0.0f) ? get_d(I) : s.td;
    } else {
        r_integ = s.n0.integ;
        r_dt = s.n0.dt;
        r_d = s.n0.d;
    }
    // root's integrate_main outcome -> need_to_pop_top (event_pixel_tree.rs:394-396)
    const float sum = fadd(r_integ, I);
    uint32_t d_after;
    float dt_after;
    if (sum >= pow2_d(r_d)) {
        const uint32_t nd = get_d(sum);
        if (nd < kDMax) {
            d_after = nd + 1u;
            dt_after = fadd(r_dt, sc.time_spanned);
        } else {
            d_after = nd;
            dt_after = r_dt;
        }
    } else {
        d_after = r_d;
        dt_after = fadd(r_dt, sc.time_spanned);
    }
    if (d_after == kDMax || (dt_after >= sc.dtm_f && !popped)) count += 1u;
    return count;
}

// Phase B: the full step.  `deep` gives access to levels k >= 1 of this pixel
// (load(k, Node&), store(k, const Node&)); `emit(d, t)` appends one event of this
// pixel (in order).  Returns false if the pixel needed more than sc.max_depth levels.
template <class Deep, class Emit>
ADDER_HD bool exec_step(PxState &s, uint32_t v, const StepConsts &sc, Deep &deep, Emit &emit) {
    const float I = (float)v;
    const float T = sc.time_spanned;
    uint32_t flags = s.hdr >> 24;
    uint32_t m = flags & kFlagMMask;
    bool popped = (flags & kFlagPopped) != 0u;
    if (!(flags & kFlagTailLive)) {
        s.tinteg = 0.0f;
        s.tdt = 0.0f;
    }
    uint32_t base = s.hdr & 0xffu;
    uint32_t cth = (s.hdr >> 8) & 0xffu;
    uint32_t cctr = (s.hdr >> 16) & 0xffu;
    bool ok = true;

    // ---- pop_best_events (event_pixel_tree.rs:213-287) when the contrast test trips ----
    if (contrast_exceeded(v, s.hdr)) {
        const bool tz = (s.tinteg == 0.0f && s.tdt > 0.0f);  // zero event of the tail (:225-230)
        const uint32_t nloc = m + (tz ? 1u : 0u);
        if (popped && sc.collapse && nloc > 0u) {
            // :249-265  keep the first local event, then the D_EMPTY filler at running_t
            uint32_t fd;
            float fdt;
            if (m > 0u) {
                fd = s.n0.bd;
                fdt = s.n0.bdt;
            } else {
                fd = kDZero;
                fdt = s.tdt;
            }
            if (sc.abs_t) fdt = fadd(fdt, s.lastf);
            emit(fd, f32_as_u32(fdt));
            s.lastf = sc.running_t;  // :257 (overrides the chain's last_fired_t updates)
            emit(kDEmpty, f32_as_u32(sc.running_t));
            s.tinteg = 0.0f;  // arena[0] = PixelNode::new(intensity)
            s.tdt = 0.0f;
        } else {
            if (m > 0u) emit(s.n0.bd, event_time(s.n0.bdt, s.lastf, sc));
            for (uint32_t k = 1; k < m; ++k) {
                Node nk;
                deep.load(k, nk);
                emit(nk.bd, event_time(nk.bdt, s.lastf, sc));
            }
            if (tz) {
                emit(kDZero, event_time(s.tdt, s.lastf, sc));
                s.tdt = 0.0f;  // get_zero_event (:103)
            }
            // arena.swap(0, length-1): the old tail becomes the root (:269)
        }
        m = 0u;
        popped = false;
        base = v;  // video.rs:1350
    }

    // ---- integrate (event_pixel_tree.rs:317-413) ----
    if (s.tinteg == 0.0f && s.tdt == 0.0f) s.td = get_d(I);  // pristine tail (:332-335)
    bool stop = false;
    if (m > 0u) {
        if (node_integrate(s.n0, I, T)) {
            m = 1u;  // length = idx + 2, fresh child
            s.tinteg = 0.0f;
            s.tdt = 0.0f;
            stop = true;
        } else if (popped && sc.collapse) {
            stop = true;  // :360-362 only the root keeps integrating
        }
        for (uint32_t k = 1; !stop && k < m; ++k) {
            Node nk;
            deep.load(k, nk);
            const bool fired = node_integrate(nk, I, T);
            deep.store(k, nk);
            if (fired) {
                m = k + 1u;
                s.tinteg = 0.0f;
                s.tdt = 0.0f;
                stop = true;
            }
        }
    }
    if (!stop) {
        Node t;
        t.integ = s.tinteg;
        t.dt = s.tdt;
        t.d = s.td;
        t.bd = 0u;
        t.bdt = 0.0f;
        if (node_integrate(t, I, T)) {
            if (m >= sc.max_depth) {
                ok = false;  // would need another stored level
            } else {
                if (m == 0u)
                    s.n0 = t;
                else
                    deep.store(m, t);
                m += 1u;
            }
            s.tinteg = 0.0f;
            s.tdt = 0.0f;
        } else {
            s.tinteg = t.integ;
            s.tdt = t.dt;
        }
    }
    const float root_dt = m > 0u ? s.n0.dt : s.tdt;
    const uint32_t root_d = m > 0u ? s.n0.d : s.td;
    const bool need_pop = root_d == kDMax || (root_dt >= sc.dtm_f && !popped);  // :394-396

    // contrast-threshold adaptation (:402-412), u8 saturating arithmetic
    if (cth < sc.c_thresh_max) {
        if (cctr >= sc.velocity_m1) {
            cth = cth >= 255u ? 255u : cth + 1u;
            cctr = 0u;
        } else {
            cctr += sc.c_inc;
            cctr = cctr > 255u ? 255u : cctr;
        }
    }

    // ---- pop_top_event (event_pixel_tree.rs:139-210) ----
    if (need_pop) {
        uint32_t ed;
        float edt;
        if (m > 0u) {
            ed = s.n0.bd;
            edt = s.n0.bdt;
            for (uint32_t k = 1; k < m; ++k) {  // shift the arena left by one
                Node nk;
                deep.load(k, nk);
                if (k == 1u)
                    s.n0 = nk;
                else
                    deep.store(k - 1u, nk);
            }
            m -= 1u;
        } else if (s.tinteg == 0.0f && s.tdt > 0.0f) {
            ed = kDZero;  // get_zero_event(0, Some(next_intensity))
            edt = s.tdt;
            s.tdt = 0.0f;
            s.td = get_d(I);
        } else {
            ed = s.tinteg < 1.0f ? kDZero : get_d(s.tinteg);  // synthesised best (:164-185)
            edt = s.tdt;
            s.tinteg = 0.0f;  // root <- fresh child
            s.tdt = 0.0f;
        }
        popped = true;
        emit(ed, event_time(edt, s.lastf, sc));
    }

    const bool live = !(s.tinteg == 0.0f && s.tdt == 0.0f);
    flags = m | (popped ? kFlagPopped : 0u) | (live ? kFlagTailLive : 0u);
    s.hdr = base | (cth << 8) | (cctr << 16) | (flags << 24);
    return ok;
}

// u8::get_frame_value for the running_intensities side plane (video.rs:713-730,
// framer/scale_intensity.rs:58-72,262-270): ((2^d / t) * ref_time) as u8 in f64.
ADDER_HD uint32_t frame_value_u8(uint32_t d, uint32_t t, double tpf) {
    double intensity;
    if (d >= 129u) {
        intensity = 0.0;
    } else {
        const double shift =
            d == 128u ? 0.0 : __builtin_bit_cast(double, (uint64_t)(d + 1023u) << 52);
        intensity = t == 0u ? shift : shift / (double)t;
    }
    const double val = intensity * tpf;
    if (!(val > 0.0)) return 0u;
    if (val >= 255.0) return 255u;
    return (uint32_t)val;
}

}  // namespace adder
