// adder_pixel.hpp -- per-pixel-channel FramePerfect integrate / decimate / fire step.
//
// Device code of the MI355X path (included by adder_kernels.hip).  It is written as
// __host__ __device__ so that tests/cpu_sim can compile the *same* functions with g++
// and diff them against the CPU reference restatement without a GPU; nothing in the
// product calls the host instantiation.
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
// where every fired_k carries a best_event and the tail carries none: a node gets a
// best_event exactly when it fires, and firing creates a fresh tail behind it
// (event_pixel_tree.rs:342-356); pop_top shifts left (:199-207); pop_best leaves only
// the old tail or a fresh node (:249-273).  Moreover the tail is ALWAYS pristine
// (integration 0, delta_t 0): a pristine tail gets d = floor(log2 I) (128 for I < 1)
// right before it is visited (:332-335), so `0 + I >= 2^d` holds and it fires the first
// time the walk reaches it (:427) -- it never accumulates, and FramePerfect hands the
// child nothing (:468-469).  Consequently the zero-event and synthesised-event branches
// of pop_top (:156-197) and the tail's zero event in pop_best (:225-230) are
// unreachable here, and a pixel is fully described by
//   hdr    : base_val | c_thresh<<8 | c_increase_counter<<16 | flags<<24
//            flags = m (5 bits) | popped_dtm<<5
//   level k: (integration, delta_t, best_d, best_delta_t) for k < m; the node's d is a
//            function of best_d (fired_d below)
//   last_fired_t (AbsoluteT only);  running_t is identical for every pixel and is
//   passed in.
// need_to_pop_top is never set between frames (integrate_for_px pops at its end),
// dtm_reached is recomputed by every integrate, `alt` is only asserted on.
// tests/cpu_sim + tests/test_device_logic_cpu.py check all of this against the literal
// restatement on randomised clips in every mode.
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

constexpr uint32_t kMaxDepthLimit = 31;

// Uniform (per launch) constants of one frame step.
struct StepConsts {
    float time_spanned;  // `time` of PixelArena::integrate
    float running_t;     // PixelArena::running_t BEFORE this frame's integrate
    uint32_t running_t_u32;  // running_t as u32 (the t of a D_EMPTY event)
    float dtm_f;         // delta_t_max as f32 (event_pixel_tree.rs:394)
    uint32_t ref_time;
    uint32_t c_thresh_max;
    uint32_t velocity_m1;  // (c_increase_velocity - 1) as u8
    uint32_t c_inc;        // ((time as u32) / ref_time) as u8   (:408-410)
    uint32_t collapse;     // PixelMultiMode::Collapse
    uint32_t abs_t;        // TimeMode::AbsoluteT
    uint32_t max_depth;    // stored levels available
    uint32_t ref_magic;    // floor(2^32 / ref_time) (ref_time >= 2), see ceil_to_ref()
};

// One fired node.
struct Node {
    float integ;
    float dt;
    float bdt;    // best_event.delta_t
    uint32_t bd;  // best_event.d
};

// Always-resident part of one pixel-channel.
struct PxState {
    uint32_t hdr;
    Node n0;  // level 0, valid iff m > 0
    float lastf;
};

ADDER_HD float bits_to_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
ADDER_HD uint32_t f32_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// D_SHIFT_F32[d]: 2^d for d < 128, 0 for d == 128 (adder-codec-core/src/lib.rs:220-235)
ADDER_HD float pow2_d(uint32_t d) { return d >= 128u ? 0.0f : bits_to_f32((d + 127u) << 23); }

// get_d_from_intensity (event_pixel_tree.rs:482-499): floor(log2(trunc(x))) clamped to
// D_MAX, 128 if x < 1.  For x >= 1 that is the unbiased binary32 exponent.
ADDER_HD uint32_t get_d(float x) {
    const uint32_t e = ((f32_to_bits(x) >> 23) & 0xffu) - 127u;
    return x < 1.0f ? kDZero : (e > kDMax ? kDMax : e);
}

// d of a FIRED node as a function of its best_event.d: firing sets best_d = nd and
// d = nd + 1 (nd < D_MAX) or d = nd (nd = 127 / 128), and nothing but the next firing
// changes either (event_pixel_tree.rs:438-461).  So stored levels keep only best_d.
ADDER_HD uint32_t fired_d(uint32_t bd) { return bd < kDMax ? bd + 1u : bd; }

// rustc `f32 as u32`: truncate toward zero, saturate, NaN -> 0.  On gfx950 that is exactly
// v_cvt_u32_f32 (round toward zero, out-of-range clamped, NaN -> 0); C++'s cast is undefined
// out of range, so the instruction is requested explicitly.
ADDER_HD uint32_t f32_as_u32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#else
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
#endif
}

// Unfused multiply / add / correctly rounded divide.  The reference is compiled by
// rustc, which never contracts a*b+c; hipcc would (-ffp-contract=fast) unless told
// otherwise, so the one rounding-sensitive expression (:431,:445) uses these.
#if defined(__HIP_DEVICE_COMPILE__)
ADDER_HD float fmul(float a, float b) { return __fmul_rn(a, b); }
ADDER_HD float fadd(float a, float b) { return __fadd_rn(a, b); }
ADDER_HD float fsub(float a, float b) { return __fsub_rn(a, b); }
ADDER_HD float fdiv(float a, float b) { return __fdiv_rn(a, b); }
// Correctly rounded a / b for INTEGER-valued a in [1, 2^24] and b in [1, 255]: hardware
// reciprocal, one multiply, one exact-remainder correction (4 instructions instead of the 12
// of the general IEEE sequence).  Equal to __fdiv_rn on that whole domain on gfx950 --
// adder_hip_selftest_division() checks all 4.3e9 pairs.  Outside the domain the result is
// unspecified (but finite or inf/nan without trapping); step_fast only consumes it inside.
ADDER_HD float fdiv_small(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = __fmul_rn(a, r);
    return __fmaf_rn(__fmaf_rn(-q, b, a), r, q);
}
#else
ADDER_HD float fmul(float a, float b) { return a * b; }
ADDER_HD float fadd(float a, float b) { return a + b; }
ADDER_HD float fsub(float a, float b) { return a - b; }
ADDER_HD float fdiv(float a, float b) { return a / b; }
ADDER_HD float fdiv_small(float a, float b) { return a / b; }
#endif

// The firing arm of integrate_main (event_pixel_tree.rs:427-473), FramePerfect: node with
// (integ, dt) and current d fires on intensity I over `time`; s = integ + I >= 2^d.
ADDER_HD void node_fire(Node &n, uint32_t d, float s, float intensity, float time) {
    const uint32_t nd = get_d(s);
    float prop = fdiv(fsub(pow2_d(nd), n.integ), intensity);
    if (nd == kDZero || d == kDZero || intensity < 1.1920929e-7f) prop = 1.0f;
    n.bd = nd;
    n.bdt = fadd(n.dt, fmul(time, prop));
    if (nd < kDMax) {  // otherwise the node keeps (integ, dt) and d = nd
        n.integ = s;
        n.dt = fadd(n.dt, time);
    }
}

// integrate_main for a stored level.  Returns true if it fired.
ADDER_HD bool node_integrate(Node &n, float intensity, float time) {
    const uint32_t d = fired_d(n.bd);
    const float s = fadd(n.integ, intensity);
    if (s >= pow2_d(d)) {
        node_fire(n, d, s, intensity, time);
        return true;
    }
    n.integ = s;
    n.dt = fadd(n.dt, time);
    return false;
}

// The (always pristine) tail being visited: it fires and becomes a stored level.
ADDER_HD Node tail_fire(float intensity, float time) {
    Node n;
    n.integ = 0.0f;
    n.dt = 0.0f;
    n.bdt = 0.0f;
    n.bd = 0u;
    node_fire(n, get_d(intensity), intensity, intensity, time);
    return n;
}

// floor-to-multiple helper for the FramePerfect last_fired_t rounding without a hardware
// divide: magic = floor(2^32 / ref_time) (ref_time >= 2) gives q or q-1.
ADDER_HD uint32_t ceil_to_ref(uint32_t lf, uint32_t ref_time, uint32_t magic) {
    if (ref_time == 1u) return lf;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t q = __umulhi(lf, magic);
#else
    uint32_t q = (uint32_t)(((uint64_t)lf * magic) >> 32);
#endif
    uint32_t r = lf - q * ref_time;
    if (r >= ref_time) {
        r -= ref_time;
        q += 1u;
    }
    return r == 0u ? lf : (q + 1u) * ref_time;
}

// delta_t_to_absolute_t (event_pixel_tree.rs:113-137), FramePerfect.  Returns the
// event's `t` and updates last_fired_t.
template <bool ABS_T>
ADDER_HD uint32_t event_time(float ev_dt, float &lastf, const StepConsts &sc) {
    if (ABS_T) {
        ev_dt = fadd(ev_dt, lastf);
        lastf = (float)ceil_to_ref(f32_as_u32(ev_dt), sc.ref_time, sc.ref_magic);
    }
    return f32_as_u32(ev_dt);
}

// video.rs:1338-1340: frame_val outside [base (-sat) c_thresh, base (+sat) c_thresh].
// v, base, c_thresh are u8, so the saturating bounds reduce to |v - base| > c_thresh.
ADDER_HD bool contrast_exceeded(uint32_t v, uint32_t hdr) {
    const uint32_t base = hdr & 0xffu;
    const uint32_t cth = (hdr >> 8) & 0xffu;
    const uint32_t diff = v > base ? v - base : base - v;
    return diff > cth;
}

// need_to_pop_top after an integrate (event_pixel_tree.rs:394-396); the root is level 0.
ADDER_HD bool root_needs_pop(const Node &root, bool popped, const StepConsts &sc) {
    return fired_d(root.bd) == kDMax || (root.dt >= sc.dtm_f && !popped);
}

// ---------------------------------------------------------------------------------------
// FAST PATH: pixels whose arena has at most ONE fired level before the step and at most
// one after it.  That is every pixel, always, when PixelMultiMode::Collapse is combined
// with delta_t_max <= time_spanned (the headline configuration: the root pops as soon as
// it has accumulated once), and the common case otherwise.
// ---------------------------------------------------------------------------------------

// Events of one fast step, in emission order: A (pop_best's first event), B (the D_EMPTY
// filler of a collapsed pop_best), C (pop_top's event).
struct FastEvents {
    uint32_t mask;  // bit0 A, bit1 B, bit2 C
    uint32_t da, ta, db, tb, dc, tc;
};

template <bool COLLAPSE>
ADDER_HD bool fast_eligible(const PxState &s, uint32_t v) {
    const uint32_t flags = s.hdr >> 24;
    const uint32_t m = flags & kFlagMMask;
    if (m >= 2u) return false;
    if (m == 0u) return true;
    // m == 1: the walk must stop at level 0 (else the tail fires and creates level 1)
    if (contrast_exceeded(v, s.hdr)) return true;        // arena restarts from the tail
    if (COLLAPSE && (flags & kFlagPopped)) return true;  // only the root integrates
    return fadd(s.n0.integ, (float)v) >= pow2_d(fired_d(s.n0.bd));
}

// Unpacked form of PxState for the frame kernel: with temporal blocking the header word is
// unpacked once per launch instead of once per frame.
struct FastPx {
    uint32_t base, cth, cctr;
    uint32_t has0;    // m == 1 (0/1)
    uint32_t popped;  // 0/1
    Node n0;
    float lastf;
};

ADDER_HD FastPx unpack_px(const PxState &s) {
    FastPx p;
    p.base = s.hdr & 0xffu;
    p.cth = (s.hdr >> 8) & 0xffu;
    p.cctr = (s.hdr >> 16) & 0xffu;
    p.has0 = (s.hdr >> 24) & 1u;  // fast path: m is 0 or 1
    p.popped = (s.hdr >> 29) & 1u;
    p.n0 = s.n0;
    p.lastf = s.lastf;
    return p;
}
ADDER_HD uint32_t pack_hdr(const FastPx &p) {
    return p.base | (p.cth << 8) | (p.cctr << 16) | ((p.has0 | (p.popped << 5)) << 24);
}

// Written branch-free on purpose: on CDNA a divergent `if` costs scalar exec-mask
// bookkeeping per region, and with several pixels per lane and ~10% of the pixels flushing
// every wave takes every path anyway.  Everything is computed unconditionally and selected.
template <bool COLLAPSE, bool ABS_T>
ADDER_HD void step_fast(FastPx &p, uint32_t v, const StepConsts &sc, FastEvents &ev) {
    const float I = (float)v;
    const float T = sc.time_spanned;
    bool has0 = p.has0 != 0u;
    bool popped = p.popped != 0u;
    float lastf = p.lastf;

    // ---- pop_best_events (event_pixel_tree.rs:213-287): A = level 0's best event; with
    // Collapse after a delta_t_max pop it is followed by the D_EMPTY filler B (:249-265) ----
    const uint32_t diff = v > p.base ? v - p.base : p.base - v;
    const bool flush = diff > p.cth;  // == the saturating-bounds test of video.rs:1338-1340
    const bool a_valid = flush && has0;
    const bool b_valid = COLLAPSE && a_valid && popped;
    {
        float evdt = p.n0.bdt;
        if (ABS_T) {
            evdt = fadd(evdt, lastf);
            const float chained = (float)ceil_to_ref(f32_as_u32(evdt), sc.ref_time, sc.ref_magic);
            lastf = a_valid ? (b_valid ? sc.running_t : chained) : lastf;  // :122-129 / :257
        }
        ev.da = p.n0.bd;
        ev.ta = f32_as_u32(evdt);
        ev.db = kDEmpty;
        ev.tb = sc.running_t_u32;
    }
    has0 = has0 && !flush;
    popped = popped && !flush;
    p.base = flush ? v : p.base;

    // ---- integrate (:317-413): arena index 0 is level 0 if present, else the pristine
    // tail, which always fires; the walk stops there (fast_eligible) ----
    const float integ = has0 ? p.n0.integ : 0.0f;
    const float dt = has0 ? p.n0.dt : 0.0f;
    const float sum = fadd(integ, I);
    // d of the node: fired_d(best_d) for level 0; for the tail floor(log2 I), which is get_d of
    // the same `sum` (integ is 0 there)
    const uint32_t nd = get_d(sum);
    const uint32_t d = has0 ? fired_d(p.n0.bd) : nd;
    const bool fire = sum >= pow2_d(d);
    // When prop is consumed (the node fires, I >= 1, no d = 128 involved) the numerator is an
    // integer with 1 <= 2^nd - integ <= I + 1 <= 256 (2^d <= 2^nd <= integ + I), I an integer in
    // [1, 255]: fdiv_small's domain.  Everywhere else the quotient is discarded.
    float prop = fdiv_small(fsub(pow2_d(nd), integ), I);
    prop = (nd == kDZero || d == kDZero || I < 1.1920929e-7f) ? 1.0f : prop;
    const float bdt_fire = fadd(dt, fmul(T, prop));
    const bool acc = !fire || nd < kDMax;  // a node that fires at nd >= D_MAX keeps (integ, dt)
    p.n0.integ = acc ? sum : integ;
    p.n0.dt = acc ? fadd(dt, T) : dt;
    p.n0.bd = fire ? nd : p.n0.bd;
    p.n0.bdt = fire ? bdt_fire : p.n0.bdt;
    const bool need_pop = fired_d(p.n0.bd) == kDMax || (p.n0.dt >= sc.dtm_f && !popped);  // :394-396

    if (sc.c_thresh_max != 0u) {  // :402-412, u8 saturating; uniform branch
        const bool adapt = p.cth < sc.c_thresh_max;
        const bool bump = p.cctr >= sc.velocity_m1;
        uint32_t cinc = p.cctr + sc.c_inc;
        cinc = cinc > 255u ? 255u : cinc;
        const uint32_t cth1 = p.cth >= 255u ? 255u : p.cth + 1u;
        p.cth = (adapt && bump) ? cth1 : p.cth;
        p.cctr = adapt ? (bump ? 0u : cinc) : p.cctr;
    }

    // ---- pop_top_event (:139-210): C = the root's best event; the arena shifts left ----
    {
        float evdt = p.n0.bdt;
        if (ABS_T) {
            evdt = fadd(evdt, lastf);
            const float chained = (float)ceil_to_ref(f32_as_u32(evdt), sc.ref_time, sc.ref_magic);
            lastf = need_pop ? chained : lastf;
        }
        ev.dc = p.n0.bd;
        ev.tc = f32_as_u32(evdt);
    }
    ev.mask = (a_valid ? 1u : 0u) | (b_valid ? 2u : 0u) | (need_pop ? 4u : 0u);
    p.has0 = need_pop ? 0u : 1u;
    p.popped = (popped || need_pop) ? 1u : 0u;
    p.lastf = lastf;
}

// PxState front end (generic-capable kernels, CPU harness)
template <bool COLLAPSE, bool ABS_T>
ADDER_HD void step_fast(PxState &s, uint32_t v, const StepConsts &sc, FastEvents &ev) {
    FastPx p = unpack_px(s);
    step_fast<COLLAPSE, ABS_T>(p, v, sc, ev);
    s.n0 = p.n0;
    s.lastf = p.lastf;
    s.hdr = pack_hdr(p);
}

// ---------------------------------------------------------------------------------------
// GENERIC PATH: any arena depth.  plan_count tells how many events the step will emit
// (so the ordered compaction can reserve the slots before the step runs), exec_step runs it.
// ---------------------------------------------------------------------------------------
ADDER_HD uint32_t plan_count(const PxState &s, uint32_t v, const StepConsts &sc) {
    const float I = (float)v;
    const uint32_t flags = s.hdr >> 24;
    const uint32_t m = flags & kFlagMMask;
    bool popped = (flags & kFlagPopped) != 0u;
    uint32_t count = 0;
    bool from_tail = (m == 0u);
    if (contrast_exceeded(v, s.hdr)) {
        count = (popped && sc.collapse && m > 0u) ? 2u : m;
        popped = false;
        from_tail = true;
    }
    Node r = s.n0;
    if (from_tail)
        r = tail_fire(I, sc.time_spanned);
    else
        node_integrate(r, I, sc.time_spanned);
    if (root_needs_pop(r, popped, sc)) count += 1u;
    return count;
}

// `deep` gives access to levels k >= 1 of this pixel (load(k, Node&), store(k, const
// Node&)); `emit(d, t)` appends one event of this pixel (in order).  Returns false if the
// pixel needed more than sc.max_depth levels.
template <bool ABS_T, class Deep, class Emit>
ADDER_HD bool exec_step_t(PxState &s, uint32_t v, const StepConsts &sc, Deep &deep, Emit &emit) {
    const float I = (float)v;
    const float T = sc.time_spanned;
    uint32_t flags = s.hdr >> 24;
    uint32_t m = flags & kFlagMMask;
    bool popped = (flags & kFlagPopped) != 0u;
    uint32_t base = s.hdr & 0xffu;
    uint32_t cth = (s.hdr >> 8) & 0xffu;
    uint32_t cctr = (s.hdr >> 16) & 0xffu;
    bool ok = true;

    // ---- pop_best_events ----
    if (contrast_exceeded(v, s.hdr)) {
        if (popped && sc.collapse && m > 0u) {
            emit(s.n0.bd, f32_as_u32(ABS_T ? fadd(s.n0.bdt, s.lastf) : s.n0.bdt));
            s.lastf = sc.running_t;
            emit(kDEmpty, f32_as_u32(sc.running_t));
        } else {
            if (m > 0u) emit(s.n0.bd, event_time<ABS_T>(s.n0.bdt, s.lastf, sc));
            for (uint32_t k = 1; k < m; ++k) {
                Node nk;
                deep.load(k, nk);
                emit(nk.bd, event_time<ABS_T>(nk.bdt, s.lastf, sc));
            }
        }
        m = 0u;
        popped = false;
        base = v;
    }

    // ---- integrate: walk the fired levels from the root; the first one that fires
    // truncates the arena behind it; if none does, the tail fires and becomes level m ----
    bool stop = false;
    if (m > 0u) {
        if (node_integrate(s.n0, I, T)) {
            m = 1u;
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
                stop = true;
            }
        }
    }
    if (!stop) {
        const Node t = tail_fire(I, T);
        if (m >= sc.max_depth) {
            ok = false;  // would need another stored level
        } else {
            if (m == 0u)
                s.n0 = t;
            else
                deep.store(m, t);
            m += 1u;
        }
    }
    const bool need_pop = m > 0u && root_needs_pop(s.n0, popped, sc);

    if (sc.c_thresh_max != 0u && cth < sc.c_thresh_max) {
        if (cctr >= sc.velocity_m1) {
            cth = cth >= 255u ? 255u : cth + 1u;
            cctr = 0u;
        } else {
            cctr += sc.c_inc;
            cctr = cctr > 255u ? 255u : cctr;
        }
    }

    // ---- pop_top_event ----
    if (need_pop) {
        const uint32_t ed = s.n0.bd;
        const float edt = s.n0.bdt;
        for (uint32_t k = 1; k < m; ++k) {  // shift the arena left by one
            Node nk;
            deep.load(k, nk);
            if (k == 1u)
                s.n0 = nk;
            else
                deep.store(k - 1u, nk);
        }
        m -= 1u;
        popped = true;
        emit(ed, event_time<ABS_T>(edt, s.lastf, sc));
    }

    flags = m | (popped ? kFlagPopped : 0u);
    s.hdr = base | (cth << 8) | (cctr << 16) | (flags << 24);
    return ok;
}

template <class Deep, class Emit>
ADDER_HD bool exec_step(PxState &s, uint32_t v, const StepConsts &sc, Deep &deep, Emit &emit) {
    return sc.abs_t ? exec_step_t<true>(s, v, sc, deep, emit) : exec_step_t<false>(s, v, sc, deep, emit);
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
