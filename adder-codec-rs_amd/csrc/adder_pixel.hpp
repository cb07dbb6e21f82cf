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
//   hdr    : base_val | best_d(level 0) << 8 | m << 16 | popped_dtm << 21
//   level k: (integration, delta_t, best_d, best_delta_t) for k < m; the node's d is a
//            function of best_d (fired_d below).  Level 0's best_d rides in the header word
//            so that level 0 is exactly 16 bytes {hdr, integration, delta_t, best_delta_t}.
//   last_fired_t (AbsoluteT only);  running_t is identical for every pixel and is
//   passed in.
// c_thresh and c_increase_counter are NOT per-pixel state here: integrate() adapts them once
// per call from (c_thresh_max, c_increase_velocity, time) alone (:402-412), every pixel
// integrates exactly once per frame (video.rs:1318-1380), PixelArena::new gives every pixel
// the same (10, 1) (:82-83) and update_crf / update_quality_manual reset every pixel to the
// same baseline (video.rs:1247-1250,1283-1286) -- so the pair is identical in all pixels at
// all times and the context carries ONE copy (c_thresh_advance below), handing each frame its
// threshold as a uniform.  (Only the feature-driven rate control of video.rs:865-1112, which
// this path does not run, would make them differ.)
// With 8-bit input, integration never exceeds 2^33 (an f32 that large absorbs a further
// + 255), so d stays <= 34 or is 128; the D_MAX clamps and the d == D_MAX pop (:395) are kept
// in the generic step and provably idle in the lean one.
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

// header word layout
constexpr uint32_t kHdrBdShift = 8;
constexpr uint32_t kHdrMShift = 16;
constexpr uint32_t kHdrMMask = 0x1f;
constexpr uint32_t kHdrPopped = 1u << 21;
ADDER_HD uint32_t hdr_m(uint32_t hdr) { return (hdr >> kHdrMShift) & kHdrMMask; }
ADDER_HD uint32_t hdr_make(uint32_t base, uint32_t bd, uint32_t m, bool popped) {
    return base | (bd << kHdrBdShift) | (m << kHdrMShift) | (popped ? kHdrPopped : 0u);
}

constexpr uint32_t kMaxDepthLimit = 31;

// Uniform (per launch) constants of one frame step.
struct StepConsts {
    float time_spanned;  // `time` of PixelArena::integrate
    float running_t;     // PixelArena::running_t BEFORE this frame's integrate
    uint32_t running_t_u32;  // running_t as u32 (the t of a D_EMPTY event)
    float dtm_f;         // delta_t_max as f32 (event_pixel_tree.rs:394)
    uint32_t ref_time;
    uint32_t cth;          // c_thresh every pixel holds while this frame is tested (uniform, see above)
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


// Always-resident part of one pixel-channel (generic step).
struct PxState {
    uint32_t base;  // base_val
    uint32_t m;     // fired levels
    bool popped;    // popped_dtm
    Node n0;        // level 0, valid iff m > 0
    float lastf;
};
ADDER_HD PxState px_unpack(uint32_t hdr, float integ, float dt, float bdt, float lastf) {
    PxState s;
    s.base = hdr & 0xffu;
    s.m = hdr_m(hdr);
    s.popped = (hdr & kHdrPopped) != 0u;
    s.n0.integ = integ;
    s.n0.dt = dt;
    s.n0.bdt = bdt;
    s.n0.bd = (hdr >> kHdrBdShift) & 0xffu;
    s.lastf = lastf;
    return s;
}
ADDER_HD uint32_t px_hdr(const PxState &s) { return hdr_make(s.base, s.n0.bd & 0xffu, s.m, s.popped); }

// The c_thresh adaptation at the end of PixelArena::integrate (event_pixel_tree.rs:402-412), all
// u8 with saturating adds; `time` is the integrate call's time argument.  Uniform across pixels
// (see the header comment), so the host advances ONE pair per frame.
ADDER_HD void c_thresh_advance(uint8_t &c_thresh, uint8_t &counter, uint8_t c_thresh_max, uint8_t velocity,
                               float time, uint32_t ref_time) {
    if (c_thresh < c_thresh_max) {
        if (counter >= (uint8_t)(velocity - 1)) {
            c_thresh = c_thresh == 255 ? (uint8_t)255 : (uint8_t)(c_thresh + 1);
            counter = 0;
        } else {
            const uint32_t inc = (uint8_t)(f32_as_u32(time) / ref_time);
            const uint32_t sum = (uint32_t)counter + inc;
            counter = (uint8_t)(sum > 255u ? 255u : sum);
        }
    }
}

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
ADDER_HD bool contrast_exceeded(uint32_t v, uint32_t base, uint32_t cth) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t diff = __builtin_amdgcn_sad_u8(v, base, 0u);  // both below 256: |v - base| in one instruction
#else
    const uint32_t diff = v > base ? v - base : base - v;
#endif
    return diff > cth;
}

// need_to_pop_top after an integrate (event_pixel_tree.rs:394-396); the root is level 0.
ADDER_HD bool root_needs_pop(const Node &root, bool popped, const StepConsts &sc) {
    return fired_d(root.bd) == kDMax || (root.dt >= sc.dtm_f && !popped);
}

// ---------------------------------------------------------------------------------------
// LEAN STEP: PixelMultiMode::Collapse with delta_t_max <= time_spanned (BASELINE config 2-4:
// delta_t_max = ref_time = 255).  There the root pops the first time it has accumulated
// (delta_t = time >= delta_t_max, :394-396) and, popped, nothing but the root integrates
// (:360-362), so the arena never holds more than ONE fired level: m is 0 or 1 before and
// after every step.  The step below is integrate_for_px for that case, folded:
//   * the node's threshold 2^d is carried as the f32 `thr` (0 for d = 128) instead of d: the
//     firing test is one compare, 2^new_d is `sum` with its mantissa cleared, the next
//     threshold is that doubled; best_d is thr's exponent minus one (128 when thr is 0);
//   * 8-bit input keeps every d below D_MAX (header comment), so `if node.state.d < D_MAX`
//     (:449) reduces to "sum != 0" and the d == D_MAX pop (:395) never happens;
//   * a node that has only ever seen zeros keeps delta_t 0 (the d = 128 firing does not
//     accumulate, :449), so "delta_t >= delta_t_max" is exactly "sum != 0" once time >= dtm.
// Its <= 3 events (A: pop_best's root event, B: the Collapse filler, C: pop_top's event) are
// not materialised: the step hands back one 16-byte record with the raw (best_delta_t, thr)
// pairs and the expansion kernel decodes it (lean_decode_*).  All of it is checked against
// the literal oracle on the host by tests/cpu_sim.
// ---------------------------------------------------------------------------------------
// The step is written once over a "lanes" policy L: booleans are L::Mask values combined with
// L::and_/or_/andnot/not_, a comparison enters with L::from() and a mask is consumed by a select
// through L::lane().  ScalarLanes (Mask = bool) is the per-unit form the CPU harness runs;
// WaveLanes (Mask = the wave's 64-bit lane mask, device only) makes every boolean of the step a
// scalar-register mask: the logic runs on the scalar ALU, selects read the mask directly and the
// kernel popcounts the event masks without a ballot.
struct ScalarLanes {
    using Mask = bool;
    static ADDER_HD Mask from(bool c) { return c; }
    static ADDER_HD bool lane(Mask m) { return m; }
    static ADDER_HD Mask and_(Mask a, Mask b) { return a && b; }
    static ADDER_HD Mask or_(Mask a, Mask b) { return a || b; }
    static ADDER_HD Mask andnot(Mask a, Mask b) { return a && !b; }
    static ADDER_HD Mask not_(Mask a) { return !a; }
};
#if defined(__HIPCC__)
struct WaveLanes {  // every lane of the wave is live in the frame kernels (padding units included)
    using Mask = uint64_t;
    static __device__ __forceinline__ Mask from(bool c) { return __builtin_amdgcn_ballot_w64(c); }
    static __device__ __forceinline__ bool lane(Mask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
    static __device__ __forceinline__ Mask and_(Mask a, Mask b) { return a & b; }
    static __device__ __forceinline__ Mask or_(Mask a, Mask b) { return a | b; }
    static __device__ __forceinline__ Mask andnot(Mask a, Mask b) { return a & ~b; }
    static __device__ __forceinline__ Mask not_(Mask a) { return ~a; }
};
#endif

template <class L>
struct LeanPxT {
    float integ, dt, bdt;
    float thr;      // 2^d of level 0 (0.0 for d = 128); meaningful iff has0
    uint32_t base;  // base_val
    typename L::Mask has0;    // m == 1
    typename L::Mask popped;  // popped_dtm
    float lastf;    // last_fired_t (AbsoluteT)
};
using LeanPx = LeanPxT<ScalarLanes>;

// One unit's output of one frame, 12 bytes.  ta / tc = best_delta_t bits (DeltaT) or the event's absolute t
// (AbsoluteT) of events A / C; w = flag bits | unit tag | the exponent bytes of the two thresholds (a threshold
// is a power of two or 0, so its exponent field is all of it).
constexpr uint32_t kLeanA = 1u;          // event A present
constexpr uint32_t kLeanB = 2u;          // event B (D_EMPTY filler) present
constexpr uint32_t kLeanC = 4u;          // event C present
constexpr uint32_t kLeanUnitShift = 3;   // unit index inside the segment (10 bits)
constexpr uint32_t kLeanExpAShift = 13;  // exponent byte of A's threshold (thr bits >> 10)
constexpr uint32_t kLeanExpCShift = 21;  // exponent byte of C's threshold (thr bits >> 2)
struct LeanRec {
    uint32_t ta, tc, w;
    uint32_t w8;  // the 8-byte form's second word (DeltaT, see below); {ta, w8} is then the whole record
};
// DeltaT batches park 8 bytes per unit instead of 12: event C -- pop_top's event right after the root's first
// accumulation -- is a function of the frame's INPUT BYTE alone.  The root that pops is either the pristine tail
// that fired this frame (integration 0, delta_t 0: d = get_d(I), best_delta_t = T * (2^d / I)) or a root that has
// only ever seen zeros (threshold 0: the same d, best_delta_t = T), so the record carries I and one bit instead of
// C's time and exponent, and the expansion redoes the one division (lean_decode8):
//   w8 = A | B << 1 | C << 2 | (C's prop forced to 1) << 3 | unit << 4 | exponent byte of A's threshold << 11 | I << 19
// (AbsoluteT records keep both times: C's depends on last_fired_t, which the expansion does not have.)
constexpr uint32_t kLean8CUnit = 8u;
constexpr uint32_t kLean8UnitShift = 4;
constexpr uint32_t kLean8ExpAShift = 11;
constexpr uint32_t kLean8InShift = 19;

ADDER_HD float lean_thr_from_bd(uint32_t bd) { return pow2_d(fired_d(bd)); }
// best_d of a fired node from its threshold word: thr = 2^(bd+1), or 0 for bd = 128
ADDER_HD uint32_t lean_bd_from_exp(uint32_t e) { return e == 0u ? kDZero : e - 128u; }
ADDER_HD uint32_t lean_bd_from_thr(uint32_t thr_bits) { return lean_bd_from_exp((thr_bits >> 23) & 0xffu); }

template <class L>
ADDER_HD LeanPxT<L> lean_unpack(uint32_t hdr, float integ, float dt, float bdt, float lastf) {
    LeanPxT<L> p;
    p.integ = integ;
    p.dt = dt;
    p.bdt = bdt;
    p.thr = lean_thr_from_bd((hdr >> kHdrBdShift) & 0xffu);
    p.base = hdr & 0xffu;
    p.has0 = L::from(hdr_m(hdr) != 0u);  // lean batches only ever see m <= 1
    p.popped = L::from((hdr & kHdrPopped) != 0u);
    p.lastf = lastf;
    return p;
}
template <class L>
ADDER_HD uint32_t lean_hdr(const LeanPxT<L> &p) {
    return hdr_make(p.base, lean_bd_from_thr(f32_to_bits(p.thr)), L::lane(p.has0) ? 1u : 0u, L::lane(p.popped));
}

// Which events the unit produced this frame.
template <class L>
struct LeanFlagsT {
    typename L::Mask a, b, c;
};
// `cth` is the frame's uniform contrast threshold, T = time_spanned >= delta_t_max (the lean
// precondition).  The unit has a record iff flags.a || flags.c.
template <bool ABS_T, class L>
ADDER_HD LeanFlagsT<L> lean_step(LeanPxT<L> &p, uint32_t v, uint32_t cth, float T, const StepConsts &sc, uint32_t tag,
                                 LeanRec &rec) {
    using M = typename L::Mask;
    const float I = (float)v;
    // ---- pop_best_events (:213-287): A = the root's best event; popped + Collapse adds the
    // D_EMPTY filler B and restarts the arena from a fresh node (:249-265) ----
    const M flush = L::from(contrast_exceeded(v, p.base, cth));
    const M a_valid = L::and_(flush, p.has0);
    const M b_valid = L::and_(a_valid, p.popped);
    uint32_t ta = f32_to_bits(p.bdt);
    float lastf = p.lastf;
    if (ABS_T) {
        const float evdt = fadd(p.bdt, lastf);
        const uint32_t t = f32_as_u32(evdt);
        const float chained = (float)ceil_to_ref(t, sc.ref_time, sc.ref_magic);
        const float nl = L::lane(b_valid) ? sc.running_t : chained;  // :122-129 / :257
        lastf = L::lane(a_valid) ? nl : lastf;
        ta = t;
    }
    const uint32_t wa =
        (f32_to_bits(p.thr) >> 10) | tag | (L::lane(a_valid) ? kLeanA : 0u) | (L::lane(b_valid) ? kLeanB : 0u);
    const uint32_t w8a = (f32_to_bits(p.thr) >> 12) | (tag << (kLean8UnitShift - kLeanUnitShift)) |
                         (L::lane(a_valid) ? kLeanA : 0u) | (L::lane(b_valid) ? kLeanB : 0u);
    const float thr_in = p.thr;  // the root's threshold as the frame found it
    const M has0 = L::andnot(p.has0, flush);
    const M popped = L::andnot(p.popped, flush);
    p.base = L::lane(flush) ? v : p.base;

    // ---- integrate (:317-413): arena index 0 is level 0 if present, else the pristine tail,
    // which always fires; the walk stops there ----
    const float integ0 = L::lane(has0) ? p.integ : 0.0f;
    const float dt0 = L::lane(has0) ? p.dt : 0.0f;
    const float sum = fadd(integ0, I);
    const M keep = L::and_(has0, L::from(!(sum >= p.thr)));  // level 0 accumulates without firing
    const M zero = L::from(sum == 0.0f);                     // get_d(sum) == 128
    const float p2 = bits_to_f32(f32_to_bits(sum) & 0x7f800000u);  // 2^get_d(sum), 0 when zero
    // prop (:431-437).  Where it is consumed (the node fires, nothing is zero) the numerator is
    // an integer in [1, 255] and I an integer in [1, 255]: fdiv_small's domain.
    const M unit_prop = L::or_(zero, L::and_(has0, L::from(p.thr == 0.0f)));
    const float q = fdiv_small(fsub(p2, integ0), I);
    const float prop = L::lane(unit_prop) ? 1.0f : q;
    const float bdt_fire = fadd(dt0, fmul(T, prop));
    const float dt_acc = fadd(dt0, T);
    p.integ = sum;                                  // :450 (sum == integ0 when the d = 128 firing skips it)
    p.dt = L::lane(zero) ? dt0 : dt_acc;            // :451
    p.thr = L::lane(keep) ? p.thr : fadd(p2, p2);   // :454-461: the next power of two above sum
    p.bdt = L::lane(keep) ? p.bdt : bdt_fire;
    const M need_pop = L::not_(L::or_(popped, zero));  // :394-396 with delta_t = dt0 + T >= delta_t_max

    // ---- pop_top_event (:139-210): C = the root's best event; the arena shifts left ----
    uint32_t tc = f32_to_bits(p.bdt);
    if (ABS_T) {
        const float evdt = fadd(p.bdt, lastf);
        const uint32_t t = f32_as_u32(evdt);
        const float chained = (float)ceil_to_ref(t, sc.ref_time, sc.ref_magic);
        lastf = L::lane(need_pop) ? chained : lastf;
        tc = t;
    }
    rec.ta = ta;
    rec.tc = tc;
    rec.w = wa | (f32_to_bits(p.thr) >> 2) | (L::lane(need_pop) ? kLeanC : 0u);
    rec.w8 = w8a | (L::lane(need_pop) ? kLeanC : 0u) | (L::lane(L::and_(has0, L::from(thr_in == 0.0f))) ? kLean8CUnit : 0u) |
             (v << kLean8InShift);
    p.has0 = L::not_(need_pop);
    p.popped = L::or_(popped, need_pop);
    p.lastf = lastf;
    LeanFlagsT<L> fl;
    fl.a = a_valid;
    fl.b = b_valid;
    fl.c = need_pop;
    return fl;
}

// QUIET FRAMES of the lean step.  A unit with its root in place that is popped (or black: integration 0 and a zero
// input, so get_d(sum) = 128 and need_to_pop_top stays clear, :394-396 / :449) and passes the contrast test leaves no
// record: no flush, no pop -- the root accumulates, or fires and doubles its threshold.  What lean_step reduces to
// under lean_quiet; the blocked kernel runs it for waves ALL of whose units are quiet (static content, lossy content
// away from what moves).  has0, popped, base and last_fired_t stay as they are.
template <class L>
ADDER_HD typename L::Mask lean_quiet(const LeanPxT<L> &p, uint32_t v, uint32_t cth) {
    using M = typename L::Mask;
    const M calm = L::and_(p.has0, L::not_(L::from(contrast_exceeded(v, p.base, cth))));
    return L::and_(calm, L::or_(p.popped, L::from(p.integ == 0.0f && v == 0u)));
}
// the root accumulates without firing (lean_step's `keep`)
template <class L>
ADDER_HD typename L::Mask lean_quiet_keeps(const LeanPxT<L> &p, uint32_t v) {
    return L::from(!(fadd(p.integ, (float)v) >= p.thr));
}
// MAY_FIRE = false: the caller knows that every unit of its wave keeps
template <class L, bool MAY_FIRE = true>
ADDER_HD void lean_step_quiet(LeanPxT<L> &p, uint32_t v, float T) {
    using M = typename L::Mask;
    const float I = (float)v;
    const float integ0 = p.integ;
    const float sum = fadd(integ0, I);
    const M zero = L::from(sum == 0.0f);
    const float dt0 = p.dt;
    const float dt_acc = fadd(dt0, T);
    p.integ = sum;
    p.dt = L::lane(zero) ? dt0 : dt_acc;
    if (MAY_FIRE) {
        const M keep = L::from(!(sum >= p.thr));
        const float p2 = bits_to_f32(f32_to_bits(sum) & 0x7f800000u);
        const M unit_prop = L::or_(zero, L::from(p.thr == 0.0f));
        const float q = fdiv_small(fsub(p2, integ0), I);
        const float prop = L::lane(unit_prop) ? 1.0f : q;
        const float bdt_fire = fadd(dt0, fmul(T, prop));
        p.thr = L::lane(keep) ? p.thr : fadd(p2, p2);
        p.bdt = L::lane(keep) ? p.bdt : bdt_fire;
    }
}

// Decoding of a record (expansion kernel, CPU harness): the events in emission order are A, B, C.
struct LeanEvents {
    bool a, b, c;
    uint32_t da, ta;  // A: pop_best's root event
    uint32_t tb;      // B: {d: D_EMPTY, t: running_t as u32} (:259-263), absolute in every time mode
    uint32_t dc, tc;  // C: pop_top's event
};
ADDER_HD LeanEvents lean_decode(const LeanRec &r, bool abs_t, uint32_t running_t_u32) {
    LeanEvents e;
    e.a = (r.w & kLeanA) != 0u;
    e.b = (r.w & kLeanB) != 0u;
    e.c = (r.w & kLeanC) != 0u;
    e.da = lean_bd_from_exp((r.w >> kLeanExpAShift) & 0xffu);
    e.dc = lean_bd_from_exp((r.w >> kLeanExpCShift) & 0xffu);
    const uint32_t ca = f32_as_u32(bits_to_f32(r.ta)), cc = f32_as_u32(bits_to_f32(r.tc));
    e.ta = abs_t ? r.ta : ca;
    e.tc = abs_t ? r.tc : cc;
    e.tb = running_t_u32;
    return e;
}

// The 8-byte DeltaT record {ta, w8}: C is recomputed from the input byte exactly as lean_step computed it (same
// operations on the same values: best_delta_t = 0 + T * prop, prop = 2^get_d(I) / I or 1).
ADDER_HD LeanEvents lean_decode8(uint32_t ta, uint32_t w8, float T, uint32_t running_t_u32) {
    LeanEvents e;
    e.a = (w8 & kLeanA) != 0u;
    e.b = (w8 & kLeanB) != 0u;
    e.c = (w8 & kLeanC) != 0u;
    e.da = lean_bd_from_exp((w8 >> kLean8ExpAShift) & 0xffu);
    e.ta = f32_as_u32(bits_to_f32(ta));
    e.tb = running_t_u32;
    const float I = (float)((w8 >> kLean8InShift) & 0xffu);
    const float p2 = bits_to_f32(f32_to_bits(I) & 0x7f800000u);  // 2^get_d(I)
    const float prop = (w8 & kLean8CUnit) ? 1.0f : fdiv_small(fsub(p2, 0.0f), I);
    e.dc = get_d(I);
    e.tc = f32_as_u32(fadd(0.0f, fmul(T, prop)));
    return e;
}

// ---------------------------------------------------------------------------------------
// GENERIC STEP (Normal mode, or delta_t_max > time_spanned): any arena depth.  integrate_for_px
// for one unit, split into the four phases the kernel runs wave-wide:
//   gen_root  : pop_best_events' bookkeeping + the walk's visit of arena index 0, entirely on the
//               register-resident level 0 and branch-free; leaves the number of events the unit will emit
//               this frame (so the ordered compaction can place them before they exist) and whether
//               the walk goes on to deeper levels;
//   gen_emit  : the events, in emission order -- the flushed arena (root first, then levels 1.. or the
//               Collapse filler), then pop_top's event;
//   gen_walk  : integrate's loop over levels >= 1 (:340-392): levels accumulate until one fires, which
//               truncates the arena behind it; if none does the pristine tail fires and becomes a level;
//   gen_pop   : pop_top_event's arena shift (:199-207).
// `deep` gives access to levels k >= 1 of this pixel (load(k, Node&), store(k, const Node&)); `emit(d, t)`
// appends one event.  tests/cpu_sim runs exactly these functions against the literal oracle.
// ---------------------------------------------------------------------------------------
struct GenPlan {
    bool flush;        // pop_best_events ran this frame
    bool collapsed;    // ... on a popped arena in Collapse mode: root event + D_EMPTY filler only (:249-265)
    bool walk;         // the walk continues past arena index 0
    bool need_pop;     // pop_top_event follows the integrate (:394-396)
    uint32_t m_old;    // fired levels before the step
    uint32_t old_bd;   // the old root's best event (valid iff m_old > 0)
    float old_bdt;
    uint32_t count;    // events of this unit this frame
};

template <bool COLLAPSE>
ADDER_HD void gen_root(PxState &s, uint32_t v, const StepConsts &sc, GenPlan &p) {
    const float I = (float)v;
    const float T = sc.time_spanned;
    p.m_old = s.m;
    p.old_bd = s.n0.bd;
    p.old_bdt = s.n0.bdt;
    p.flush = contrast_exceeded(v, s.base, sc.cth);
    p.collapsed = COLLAPSE && p.flush && s.popped && s.m > 0u;
    const uint32_t flushed = p.flush ? (p.collapsed ? 2u : s.m) : 0u;
    const bool has0 = s.m != 0u && !p.flush;
    const bool popped = s.popped && !p.flush;
    s.base = p.flush ? v : s.base;
    // arena index 0: level 0 if present, else the pristine tail (which always fires)
    const float integ = has0 ? s.n0.integ : 0.0f;
    const float dt = has0 ? s.n0.dt : 0.0f;
    const float sum = fadd(integ, I);
    const uint32_t nd = get_d(sum);
    const uint32_t d = has0 ? fired_d(s.n0.bd) : nd;
    const bool fire = sum >= pow2_d(d);
    float prop = fdiv_small(fsub(pow2_d(nd), integ), I);  // consumed only inside fdiv_small's domain (lean_step)
    prop = (nd == kDZero || d == kDZero || I < 1.1920929e-7f) ? 1.0f : prop;
    const float bdt_fire = fadd(dt, fmul(T, prop));
    const bool acc = !fire || nd < kDMax;  // a node that fires at nd >= D_MAX keeps (integ, dt)
    s.n0.integ = acc ? sum : integ;
    s.n0.dt = acc ? fadd(dt, T) : dt;
    s.n0.bd = fire ? nd : s.n0.bd;
    s.n0.bdt = fire ? bdt_fire : s.n0.bdt;
    // a firing root truncates the arena to itself; otherwise the deeper levels are visited unless only the
    // root integrates any more (:360-362)
    p.walk = has0 && !fire && !(COLLAPSE && popped);
    s.m = (fire || !has0) ? 1u : s.m;
    s.popped = popped;
    p.need_pop = root_needs_pop(s.n0, popped, sc);
    p.count = flushed + (p.need_pop ? 1u : 0u);
}

// The unit's events of this frame.  Must run after gen_root and BEFORE gen_walk / gen_pop touch the
// deep levels of other frames' state: a flushing unit does not walk, so its old levels are still in place.
template <bool ABS_T, class Deep, class Emit>
ADDER_HD void gen_emit(PxState &s, const GenPlan &p, const StepConsts &sc, Deep &deep, Emit &emit) {
    if (p.flush && p.m_old > 0u) {
        if (p.collapsed) {
            emit(p.old_bd, f32_as_u32(ABS_T ? fadd(p.old_bdt, s.lastf) : p.old_bdt));
            s.lastf = sc.running_t;  // :257
            emit(kDEmpty, f32_as_u32(sc.running_t));
        } else {
            emit(p.old_bd, event_time<ABS_T>(p.old_bdt, s.lastf, sc));
            for (uint32_t k = 1; k < p.m_old; ++k) {
                Node nk;
                deep.load(k, nk);
                emit(nk.bd, event_time<ABS_T>(nk.bdt, s.lastf, sc));
            }
        }
    }
    if (p.need_pop) emit(s.n0.bd, event_time<ABS_T>(s.n0.bdt, s.lastf, sc));  // the root after this frame's integrate
}

// Returns false if the pixel needed more than sc.max_depth levels.
template <class Deep>
ADDER_HD bool gen_walk(PxState &s, uint32_t v, const GenPlan &p, const StepConsts &sc, Deep &deep) {
    if (!p.walk) return true;
    const float I = (float)v;
    const float T = sc.time_spanned;
    uint32_t k = 1;
    Node nf;  // the node that fires: level k, or the pristine tail when k == m
    nf.integ = 0.0f;
    nf.dt = 0.0f;
    nf.bdt = 0.0f;
    nf.bd = 0u;
    uint32_t d = get_d(I);  // the tail's d (:332-335)
    for (; k < s.m; ++k) {
        Node nk;
        deep.load(k, nk);
        const uint32_t dk = fired_d(nk.bd);
        const float sk = fadd(nk.integ, I);
        if (sk >= pow2_d(dk)) {
            nf = nk;
            d = dk;
            break;
        }
        nk.integ = sk;
        nk.dt = fadd(nk.dt, T);
        deep.store(k, nk);
    }
    if (k >= sc.max_depth) return false;  // would need another stored level
    node_fire(nf, d, fadd(nf.integ, I), I, T);
    deep.store(k, nf);
    s.m = k + 1u;
    return true;
}

template <class Deep>
ADDER_HD void gen_pop(PxState &s, const GenPlan &p, Deep &deep) {
    if (!p.need_pop) return;
    for (uint32_t k = 1; k < s.m; ++k) {  // shift the arena left by one
        Node nk;
        deep.load(k, nk);
        if (k == 1u)
            s.n0 = nk;
        else
            deep.store(k - 1u, nk);
    }
    s.m -= 1u;
    s.popped = true;
}

// ---------------------------------------------------------------------------------------
// BOUNDED COLLAPSE STEP ("cb"): PixelMultiMode::Collapse with delta_t_max > time_spanned -- the reference's
// DEFAULT configuration (Collapse, delta_t_max 7650 = 30 frames; lib.rs:207-213, video.rs:173-182) and BASELINE
// config 5.  What is special about it:
//   * until the root has accumulated delta_t_max (pop_top, :394-396) every frame adds at most one level to the
//     arena, and in steady state the levels behave like a binary counter: depth <= ~log2(delta_t_max / time) + 1;
//   * once popped, ONLY the root integrates (:360-362), a firing root truncates the arena to itself (:342-356) and a
//     flush emits the first node's event plus the D_EMPTY filler (:249-265): levels >= 1 can never matter again, so
//     pop_top here keeps at most the new root (m <= 1 while popped).
// PREFIX COORDINATES.  While the walk visits a level it sees the same intensity and time as the root
// (FramePerfect hands a child nothing, :468-469), and levels 0 .. k-1 all accumulate in any frame in which level
// k is reached.  So for k >= 1:  integration_k = S - P_k,  delta_t_k = dt0 - Q_k  with (S, dt0) the ROOT's
// integration and delta_t and (P_k, Q_k) constants fixed when the level was created.  All of these are integers
// below 2^24 here (8-bit intensities, integer time_spanned, at most delta_t_max / time + 1 frames before the pop:
// the host checks the bounds and falls back to the generic step otherwise), so the f32 subtractions are exact and
// equal the reference's running sums bit for bit.  A level that is visited without firing therefore needs NO
// update; "level k fires" is `S_new >= F_k` with the fire-at value F_k = P_k + 2^d_k, and only the one level that
// fires in a frame is rewritten.  A stored level is {F, Q, best_delta_t, best_d}.
// The zero-intensity quirk (a node firing at d = 128 keeps its integration and delta_t, :449) moves Q instead.
// tests/cpu_sim runs exactly these functions (and the accessor below) against the literal oracle.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kCbFastLevels = 4;  // levels 1..4 of a unit sit side by side (one 16-byte read finds the one that fires)

struct CbLevel {
    float F;    // fire-at: the level fires in the first frame whose root integration reaches F
    float Q;    // root delta_t minus the level's delta_t
    float bdt;  // best_event.delta_t
    float thr;  // 2^d of the level (0 for d = 128); best_event.d is its exponent minus one (lean_bd_from_thr)
};

// Levels k >= 1 of ONE unit.  Levels 1..kCbFastLevels: four consecutive floats per plane (F4, Q4) and four pairs (BT) -- the
// wave's LDS slice on the device (FP is then an LDS pointer), plain arrays in the CPU harness.  Deeper levels
// (delta_t_max beyond ~30 frames, or adversarial input) go to the context's deep planes, level k at [k - 1][u],
// which hold {F, Q, best_delta_t, best_d} instead of {integration, delta_t, ..} for the duration of a launch.
template <class FP>
struct CbLevelsT {
    FP F4, Q4, BT;  // BT: {best_delta_t, threshold} pairs (one 8-byte read per emitted level)
    float *gF, *gQ, *gB;
    uint8_t *gbd;
    size_t stride, u;
    ADDER_HD CbLevel load_fast(uint32_t k) const {  // 1 <= k <= kCbFastLevels
        CbLevel l;
        l.F = F4[k - 1u];
        l.Q = Q4[k - 1u];
        l.bdt = BT[2u * (k - 1u)];
        l.thr = BT[2u * (k - 1u) + 1u];
        return l;
    }
    ADDER_HD void store_fast(uint32_t k, const CbLevel &l) const {
        F4[k - 1u] = l.F;
        Q4[k - 1u] = l.Q;
        BT[2u * (k - 1u)] = l.bdt;
        BT[2u * (k - 1u) + 1u] = l.thr;
    }
    ADDER_HD void load_bt(float (&bt)[2u * kCbFastLevels]) const {  // the four fast {best_delta_t, threshold} pairs
        for (uint32_t i = 0; i < 2u * kCbFastLevels; ++i) bt[i] = BT[i];
    }
    ADDER_HD CbLevel load_deep(uint32_t k) const {  // k > kCbFastLevels
        const size_t i = (size_t)(k - 1u) * stride + u;
        CbLevel l;
        l.F = gF[i];
        l.Q = gQ[i];
        l.bdt = gB[i];
        l.thr = lean_thr_from_bd(gbd[i]);
        return l;
    }
    ADDER_HD void store_deep(uint32_t k, const CbLevel &l) const {
        const size_t i = (size_t)(k - 1u) * stride + u;
        gF[i] = l.F;
        gQ[i] = l.Q;
        gB[i] = l.bdt;
        gbd[i] = (uint8_t)lean_bd_from_thr(f32_to_bits(l.thr));
    }
    ADDER_HD CbLevel load(uint32_t k) const { return k <= kCbFastLevels ? load_fast(k) : load_deep(k); }
    ADDER_HD void store(uint32_t k, const CbLevel &l) const {
        if (k <= kCbFastLevels) store_fast(k, l);
        else store_deep(k, l);
    }
    // The shallowest of the four fast levels whose fire-at value S has reached, or kCbFastLevels + 1.  Slots of levels
    // that do not exist hold stale values; the caller's min(., m) makes them harmless.
    ADDER_HD uint32_t first_fast(float S) const {
        uint32_t kf = kCbFastLevels + 1u;
        kf = S >= F4[3] ? 4u : kf;
        kf = S >= F4[2] ? 3u : kf;
        kf = S >= F4[1] ? 2u : kf;
        kf = S >= F4[0] ? 1u : kf;
        return kf;
    }
    ADDER_HD uint32_t first_deep(float S, uint32_t m) const {  // levels kCbFastLevels + 1 .. m - 1, or m
        uint32_t k = kCbFastLevels + 1u;
        for (; k < m; ++k)
            if (S >= gF[(size_t)(k - 1u) * stride + u]) break;
        return k;
    }
};
using CbLevels = CbLevelsT<float *>;

// The 9-bit d code of a bounded-step record: the exponent field of the node's threshold (with the sign bit above it,
// always 0) -- 0: d = 128 (a zero threshold), e: d = e - 128 -- or kCbCodeEmpty for the Collapse filler.
constexpr uint32_t kCbCodeEmpty = 255u;  // (the exponent of inf / nan: never a threshold's)
ADDER_HD uint32_t cb_d_from_code(uint32_t code) {
    return code == kCbCodeEmpty ? kDEmpty : lean_bd_from_exp(code & 0xffu);
}

// Like the lean step, the bounded step is written over a "lanes" policy L (ScalarLanes on the host; WaveLanes on the
// device, where every boolean is a wave mask in scalar registers and the logic runs on the scalar ALU).
template <class L>
struct CbPxT {
    float S, dt0, bdt0;  // the root's integration, delta_t, best_event.delta_t   (meaningful iff m > 0)
    float thr0;          // 2^d of the root (0 for d = 128)
    uint32_t base, m;    // base_val; fired levels (arena length - 1)
    typename L::Mask popped;  // popped_dtm
    float lastf;         // last_fired_t (AbsoluteT)
};
using CbPx = CbPxT<ScalarLanes>;
template <class L>
ADDER_HD CbPxT<L> cb_unpack(uint32_t hdr, float integ, float dt, float bdt, float lastf) {
    CbPxT<L> s;
    s.S = integ;
    s.dt0 = dt;
    s.bdt0 = bdt;
    s.thr0 = lean_thr_from_bd((hdr >> kHdrBdShift) & 0xffu);
    s.base = hdr & 0xffu;
    s.m = hdr_m(hdr);
    s.popped = L::from((hdr & kHdrPopped) != 0u);
    s.lastf = lastf;
    return s;
}
template <class L>
ADDER_HD uint32_t cb_hdr(const CbPxT<L> &s) {
    return hdr_make(s.base, lean_bd_from_thr(f32_to_bits(s.thr0)), s.m, L::lane(s.popped));
}

// Levels 1 .. m-1 between their resident form {integration, delta_t, best_delta_t, best_d} (the state planes, shared
// with the generic step) and prefix coordinates.  A popped arena keeps only its root (header comment).
template <class L>
ADDER_HD CbLevel cb_level_from_node(const CbPxT<L> &s, const Node &n) {
    CbLevel l;
    l.thr = lean_thr_from_bd(n.bd);
    l.F = fadd(fsub(s.S, n.integ), l.thr);
    l.Q = fsub(s.dt0, n.dt);
    l.bdt = n.bdt;
    return l;
}
template <class L>
ADDER_HD Node cb_node_from_level(const CbPxT<L> &s, const CbLevel &l) {
    Node n;
    n.integ = fsub(s.S, fsub(l.F, l.thr));
    n.dt = fsub(s.dt0, l.Q);
    n.bdt = l.bdt;
    n.bd = lean_bd_from_thr(f32_to_bits(l.thr));
    return n;
}

template <class L>
struct CbPlanT {
    typename L::Mask flush;      // pop_best_events ran this frame
    typename L::Mask flushed;    // ... and the arena held a fired level: events leave
    typename L::Mask collapsed;  // ... and was popped: root event + D_EMPTY filler (:249-265)
    typename L::Mask need_pop;   // pop_top_event follows the integrate
    typename L::Mask depth_error;  // the unit needed more than sc.max_depth stored levels
    uint32_t m_old;   // fired levels before the step
    float old_thr0, old_bdt0;  // the old root's best event (valid iff m_old > 0)
    uint32_t count;   // events of this unit this frame
};
using CbPlan = CbPlanT<ScalarLanes>;

// pop_best_events' bookkeeping + integrate (:317-413) of one unit.  Exactly one node fires per frame while the arena
// is not popped -- arena index 0 (the root, or the pristine tail of an empty arena), else the shallowest level whose
// fire-at value the root's integration has reached, else the pristine tail behind the levels -- and none or the root
// once it is (:360-362).  The firing arm of integrate_main (:427-473) is computed ONCE, on the node's (integration,
// delta_t, threshold) gathered from wherever it lives.
// Two halves, so that a lane can run the first half of all its units before the second of any: the first issues the
// level reads (two dependent round trips to the wave's LDS slice) without any control flow, the second consumes them.
// The unit's OLD levels 1 .. m_old-1 are still in place afterwards when plan.flush is set (a flushed arena restarts
// from the tail at index 0, which touches no level).
template <class L>
struct CbMidT {
    float I, S_old, dt_old, S_new;
    uint32_t m, k, kf;    // fired levels after the flush; the level that fires if the walk gets that far; first_fast
    typename L::Mask has0, popped, root_fires, walk;
    CbLevel l;            // fast slot clamp(k, 1, kCbFastLevels) as read (stale when the firing node is a fresh tail)
};

template <class L, class Lv>
ADDER_HD void cb_step_a(CbPxT<L> &s, const Lv &lv, uint32_t v, const StepConsts &sc, CbPlanT<L> &p, CbMidT<L> &x) {
    using M = typename L::Mask;
    x.I = (float)v;
    p.m_old = s.m;
    p.old_thr0 = s.thr0;
    p.old_bdt0 = s.bdt0;
    const M has_m = L::from(s.m != 0u);
    p.flush = L::from(contrast_exceeded(v, s.base, sc.cth));
    p.flushed = L::and_(p.flush, has_m);
    p.collapsed = L::and_(p.flushed, s.popped);
    p.depth_error = L::from(false);
    p.count = L::lane(p.flush) ? (L::lane(p.collapsed) ? 2u : s.m) : 0u;  // (+ pop_top's event: cb_step_b)
    x.m = L::lane(p.flush) ? 0u : s.m;
    x.popped = L::andnot(s.popped, p.flush);
    s.base = L::lane(p.flush) ? v : s.base;
    x.has0 = L::andnot(has_m, p.flush);
    x.S_old = L::lane(x.has0) ? s.S : 0.0f;
    x.dt_old = L::lane(x.has0) ? s.dt0 : 0.0f;
    x.S_new = fadd(x.S_old, x.I);
    x.root_fires = L::or_(L::not_(x.has0), L::from(x.S_new >= s.thr0));  // (the pristine tail at index 0 always fires)
    x.walk = L::andnot(L::not_(x.root_fires), x.popped);
    x.kf = kCbFastLevels + 1u;
    x.k = 0u;
    x.l = CbLevel{0.0f, 0.0f, 0.0f, 0.0f};
}

// The level reads of the walk, whether or not THIS unit walks (when some lane of the wave does): no branch, so the reads
// of a lane's units overlap.  A wave none of whose units walks -- every arena popped, the steady state of static and of
// lossy (crf > 0) content -- skips them and runs cb_step_b<.., false>.
template <class L, class Lv>
ADDER_HD void cb_step_reads(const Lv &lv, CbMidT<L> &x) {
    x.kf = lv.first_fast(x.S_new);
    x.k = x.kf < x.m ? x.kf : x.m;
    const uint32_t ks = x.k < 1u ? 1u : (x.k > kCbFastLevels ? kCbFastLevels : x.k);
    x.l = lv.load_fast(ks);
}

// MAY_WALK = false: the caller knows that x.walk is false (no level is read, fired or stored: only the root integrates)
template <class L, class Lv, bool MAY_WALK = true>
ADDER_HD void cb_step_b(CbPxT<L> &s, const Lv &lv, float T, const StepConsts &sc, CbPlanT<L> &p, const CbMidT<L> &x_in) {
    using M = typename L::Mask;
    CbMidT<L> x = x_in;
    if (!MAY_WALK) x.walk = L::from(false);
    // ---- which node fires, and its (integration, delta_t, threshold) as offsets from the root's ----
    uint32_t k = x.k;
    M fresh_lv = L::from(k >= x.m);  // the pristine tail behind the levels: integration 0, delta_t 0, d = get_d(I) (:332-335)
    float P = fsub(x.l.F, x.l.thr), Qk = x.l.Q, thr_lv = x.l.thr;
    const M deep = L::and_(x.walk, L::from(x.kf > kCbFastLevels && x.m > kCbFastLevels));
    if (L::lane(deep)) {  // rare: the arena is deeper than the fast slots
        k = lv.first_deep(x.S_new, x.m);
        if (k < x.m && k < sc.max_depth) {
            const CbLevel l = lv.load_deep(k);
            P = fsub(l.F, l.thr);
            Qk = l.Q;
            thr_lv = l.thr;
        }
    }
    fresh_lv = L::or_(L::andnot(fresh_lv, deep), L::and_(deep, L::from(k >= x.m)));  // (a mask: merged outside the branch)
    const M over = L::and_(x.walk, L::from(k >= sc.max_depth));
    p.depth_error = over;
    k = L::lane(over) ? (sc.max_depth > 1u ? sc.max_depth - 1u : 1u) : k;  // (flagged; keep the store inside the planes)
    const M fresh = L::or_(L::and_(x.walk, fresh_lv), L::andnot(L::not_(x.has0), x.walk));
    const M use_lv = L::andnot(x.walk, fresh_lv);  // the firing node is a stored level
    P = L::lane(use_lv) ? P : (L::lane(x.walk) ? x.S_old : 0.0f);
    Qk = L::lane(use_lv) ? Qk : (L::lane(x.walk) ? x.dt_old : 0.0f);
    const float thr_old = L::lane(x.walk) ? thr_lv : s.thr0;
    // ---- the firing arm, once (:427-473) ----
    const float integ_old = fsub(x.S_old, P), dtk_old = fsub(x.dt_old, Qk);
    const float sum = fadd(integ_old, x.I);
    const M zero = L::from(sum == 0.0f);  // get_d(sum) == 128: the node keeps (integration, delta_t) (:449)
    const float p2 = bits_to_f32(f32_to_bits(sum) & 0x7f800000u);  // 2^get_d(sum)
    // the node's d before it fires is 128 (:432-437): a fresh tail with I < 1, a stored node with a zero threshold
    const M d128 = L::or_(L::and_(fresh, L::from(x.I < 1.0f)), L::andnot(L::from(thr_old == 0.0f), fresh));
    const float q = fdiv_small(fsub(p2, integ_old), x.I);
    const float prop = L::lane(L::or_(zero, d128)) ? 1.0f : q;
    const float bdt = fadd(dtk_old, fmul(T, prop));
    const float thr2 = fadd(p2, p2);
    // ---- where the result goes ----
    s.S = x.S_new;  // (a root that fires at d = 128 has S_new == S_old)
    s.dt0 = L::lane(L::and_(x.root_fires, zero)) ? x.dt_old : fadd(x.dt_old, T);
    s.bdt0 = L::lane(x.root_fires) ? bdt : s.bdt0;
    s.thr0 = L::lane(x.root_fires) ? thr2 : s.thr0;
    s.m = L::lane(x.root_fires) ? 1u : (L::lane(x.walk) ? k + 1u : x.m);
    if (L::lane(x.walk)) {
        CbLevel l;
        l.F = fadd(P, thr2);
        l.Q = L::lane(zero) ? fadd(Qk, T) : Qk;  // a d = 128 firing does not advance the level's delta_t; the root's did
        l.bdt = bdt;
        l.thr = thr2;
        lv.store(k, l);
    }
    s.popped = x.popped;
    p.need_pop = L::andnot(L::from(s.dt0 >= sc.dtm_f), x.popped);  // :394-396 (d == D_MAX cannot happen with 8-bit input)
    p.count += L::lane(p.need_pop) ? 1u : 0u;
}

template <class L, class Lv>
ADDER_HD void cb_step(CbPxT<L> &s, const Lv &lv, uint32_t v, float T, const StepConsts &sc, CbPlanT<L> &p) {
    CbMidT<L> x;
    cb_step_a(s, lv, v, sc, p, x);
    if (L::lane(x.walk)) {
        cb_step_reads(lv, x);
        cb_step_b<L, Lv, true>(s, lv, T, sc, p, x);
    } else {
        cb_step_b<L, Lv, false>(s, lv, T, sc, p, x);
    }
}

// QUIET FRAMES.  A unit whose arena is popped down to its root (popped_dtm set, one fired level) and whose contrast test
// passes takes none of the step's side roads: no flush, no walk, no pop_top (need_to_pop_top wants !popped_dtm, :396),
// no event -- only the root integrates (:360-362), and fires now and then (its threshold doubles every time, :452).
// That is the steady state of every pixel of static content and of lossy (crf > 0) content between its rare flushes:
// a wave ALL of whose units are quiet runs cb_step_quiet instead of the step, the compaction and the emission.
// cb_quiet is the predicate, cb_step_quiet what cb_step_a + cb_step_b<.., false> + cb_pop reduce to under it.
// A BLACK pixel is quiet too although it never pops: its d = 128 root fires on every zero without accumulating delta_t
// (:449), so delta_t never reaches delta_t_max -- the same root-only firing arm, no walk (the root fires), no pop.
template <class L>
ADDER_HD bool cb_quiet(const CbPxT<L> &s, uint32_t v, uint32_t cth) {
    return s.m == 1u && !contrast_exceeded(v, s.base, cth) && (L::lane(s.popped) || (s.thr0 == 0.0f && v == 0u));
}
// MAY_FIRE = false: the caller knows that S + I stays below the root's threshold (for every unit of its wave)
template <class L, bool MAY_FIRE = true>
ADDER_HD void cb_step_quiet(CbPxT<L> &s, uint32_t v, float T) {
    const float I = (float)v;
    const float S_new = fadd(s.S, I);
    if (!MAY_FIRE) {
        s.S = S_new;
        s.dt0 = fadd(s.dt0, T);
        return;
    }
    using M = typename L::Mask;
    const M fires = L::from(S_new >= s.thr0);
    // the firing arm of integrate_main on the root (:427-473), as in cb_step_b with P = Q = 0
    const M zero = L::from(S_new == 0.0f);
    const float p2 = bits_to_f32(f32_to_bits(S_new) & 0x7f800000u);
    const M d128 = L::from(s.thr0 == 0.0f);
    const float q = fdiv_small(fsub(p2, fsub(s.S, 0.0f)), I);
    const float prop = L::lane(L::or_(zero, d128)) ? 1.0f : q;
    const float bdt = fadd(fsub(s.dt0, 0.0f), fmul(T, prop));
    const float thr2 = fadd(p2, p2);
    const float dt_old = s.dt0;
    s.S = S_new;
    s.dt0 = L::lane(L::and_(fires, zero)) ? dt_old : fadd(dt_old, T);
    s.bdt0 = L::lane(fires) ? bdt : s.bdt0;
    s.thr0 = L::lane(fires) ? thr2 : s.thr0;
}
template <class L>
ADDER_HD bool cb_quiet_fires(const CbPxT<L> &s, uint32_t v) { return fadd(s.S, (float)v) >= s.thr0; }

// QUIET GROUPS: n <= kQuietGroup consecutive frames of a quiet unit decided at once instead of stepped (the blocked
// kernels stage their input in groups of that many frames).  A root that only integrates (popped, or black) does, per
// frame, S += v, delta_t += T (:449-451) and fires when S reaches its threshold (:427), which then doubles (:452-461).  Over
// a group whose bytes ALL pass the contrast test against the group's SMALLEST c_thresh (video.rs:1338-1340; the ramp of
// :402-412 is the same for every pixel, so the kernel hands the group's minimum down) that is: S += sum of the bytes,
// delta_t += n T, and at most one firing -- in the first frame whose prefix sum reaches (threshold - S) -- as long as the sum
// stays below the DOUBLED threshold; the firing arm is evaluated on that frame's (S, delta_t, v) exactly as the stepped
// form would.  Everything is an exact integer below 2^24 (checked here: larger sums are stepped), so the group's adds
// equal the frame-by-frame ones bit for bit.  What the caller gathers per unit over the group's bytes v_0 .. v_{n-1}, with
// P_i = v_0 + .. + v_i and need = quiet_group_need():
struct QuietGroupStats {
    uint32_t mn, mx;  // smallest / largest byte
    uint32_t sum;     // P_{n-1}
    uint32_t cnt;     // frames whose prefix sum P_i stays below need: the root fires in frame cnt (if cnt < n)
    uint32_t pm;      // P_{cnt-1}: what the root has accumulated of the group when it fires
    uint32_t vc;      // v_cnt (any value when cnt == n)
};
constexpr uint32_t kQuietGroup = 16;
constexpr uint32_t kQuietNo = 0u;    // some frame of the group is not quiet for this unit: the caller steps frame by frame
constexpr uint32_t kQuietDone = 1u;  // the group is applied
constexpr uint32_t kQuietSlow = 2u;  // quiet in every frame, but not a closed form (a second firing, a black root that
                                     // wakes up, sums beyond 2^24): the unit steps its n frames with *_step_quiet
// what the root still needs to fire, as a 16-bit count (65535: not within a group's reach; 0 for a black root)
ADDER_HD uint32_t quiet_group_need(float S, float thr) {
    const uint32_t d = f32_as_u32(fsub(thr, S));
    return d < 0xffffu ? d : 0xffffu;
}
// the reference computation of the statistics (the kernels gather them with packed 16-bit operations)
ADDER_HD QuietGroupStats quiet_group_stats(const uint8_t *v, size_t stride, uint32_t n, uint32_t need) {
    QuietGroupStats g{255u, 0u, 0u, 0u, 0u, 0u};
    uint32_t P = 0u;
    bool below = true;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t x = v[i * stride];
        g.mn = x < g.mn ? x : g.mn;
        g.mx = x > g.mx ? x : g.mx;
        P += x;
        if (below && P >= need) {
            below = false;
            g.vc = x;
        }
        if (below) {
            ++g.cnt;
            g.pm = P;
        }
    }
    g.sum = P;
    return g;
}
// (S, dt, bdt, thr) = the root's integration, delta_t, best delta_t and threshold 2^d; precondition: the unit holds its
// root alone and is popped or black (cb_quiet / lean_quiet without their per-frame part)
ADDER_HD uint32_t quiet_group_apply(float &S, float &dt, float &bdt, float &thr, uint32_t base, bool popped,
                                    const QuietGroupStats &g, uint32_t n, uint32_t cth_min, float T) {
    const uint32_t dmx = g.mx > base ? g.mx - base : base - g.mx, dmn = g.mn > base ? g.mn - base : base - g.mn;
    if ((dmx > dmn ? dmx : dmn) > cth_min) return kQuietNo;  // the extremes bound |v - base_val| of every frame
    if (!popped && g.mx != 0u) return kQuietNo;              // an unpopped unit is in here as a black one: quiet on zeros only
    if (thr == 0.0f) {
        // a black root (integration 0, d = 128) fires on every zero without accumulating (:449): idempotent
        if (g.sum != 0u) return kQuietSlow;
        bdt = fadd(dt, fmul(T, 1.0f));
        return kQuietDone;
    }
    const float S_new = fadd(S, (float)g.sum), dt_new = fadd(dt, fmul((float)n, T));
    if (!(S_new < 16777216.0f && dt_new < 16777216.0f)) return kQuietSlow;
    if (S_new >= thr) {  // the root fires in frame g.cnt: the firing arm of integrate_main (:427-473) on that frame
        if (g.cnt >= n || g.vc == 0u) return kQuietSlow;  // (cannot happen while thr > S; never divide by zero)
        const float S_old = fadd(S, (float)g.pm), I = (float)g.vc;
        const float p2 = bits_to_f32(f32_to_bits(fadd(S_old, I)) & 0x7f800000u);
        const float thr2 = fadd(p2, p2);
        if (S_new >= thr2) return kQuietSlow;  // a second firing inside the group
        bdt = fadd(fadd(dt, fmul((float)g.cnt, T)), fmul(T, fdiv_small(fsub(p2, S_old), I)));
        thr = thr2;
    }
    S = S_new;
    dt = dt_new;
    return kQuietDone;
}
template <class L>
ADDER_HD uint32_t cb_group_apply(CbPxT<L> &s, const QuietGroupStats &g, uint32_t n, uint32_t cth_min, float T) {
    return quiet_group_apply(s.S, s.dt0, s.bdt0, s.thr0, s.base, L::lane(s.popped), g, n, cth_min, T);
}
// (the lean step's quiet root is the same root: lean_step_quiet == cb_step_quiet on {integ, dt, bdt, thr})
// The lean step also runs at fractional time steps (the bounded Collapse step does not: cb_possible).  The closed form
// advances delta_t by n T in ONE rounding where the reference adds T n times (event_pixel_tree.rs:449-451): the same number
// only while every partial sum is exact -- T and delta_t integers (the group's sums stay below 2^24, checked inside).
// Otherwise the unit steps its frames (kQuietSlow).
template <class L>
ADDER_HD uint32_t lean_group_apply(LeanPxT<L> &p, const QuietGroupStats &g, uint32_t n, uint32_t cth_min, float T) {
    if (!(T == (float)f32_as_u32(T) && p.dt == (float)f32_as_u32(p.dt))) {
        const uint32_t dmx = g.mx > p.base ? g.mx - p.base : p.base - g.mx, dmn = g.mn > p.base ? g.mn - p.base : p.base - g.mn;
        if ((dmx > dmn ? dmx : dmn) > cth_min) return kQuietNo;
        if (!L::lane(p.popped) && g.mx != 0u) return kQuietNo;
        return kQuietSlow;
    }
    return quiet_group_apply(p.integ, p.dt, p.bdt, p.thr, p.base, L::lane(p.popped), g, n, cth_min, T);
}

// The unit's events of this frame, in emission order.  After cb_step, before cb_pop.  An event leaves as (the bits of
// the node's threshold, t): best_event.d is the threshold's exponent minus one (lean_bd_from_thr), which the consumer
// of the records works out -- emit.ev(thr_bits, t); the Collapse filler {d: D_EMPTY} as emit.filler(t).
template <bool ABS_T, class L, class Lv, class Emit>
ADDER_HD void cb_emit(CbPxT<L> &s, const CbPlanT<L> &p, const StepConsts &sc, const Lv &lv, Emit &emit) {
    if (L::lane(p.flushed)) {
        if (L::lane(p.collapsed)) {
            emit.ev(f32_to_bits(p.old_thr0), f32_as_u32(ABS_T ? fadd(p.old_bdt0, s.lastf) : p.old_bdt0));
            s.lastf = sc.running_t;  // :257
            emit.filler(sc.running_t_u32);
        } else {
            emit.ev(f32_to_bits(p.old_thr0), event_time<ABS_T>(p.old_bdt0, s.lastf, sc));
            // levels 1 .. m_old-1 in order: the four fast ones are read in one go and leave without a loop (each step
            // narrows the active lanes), the rest -- rare -- come from the deep planes
            if (p.m_old > 1u) {
                float bt[2u * kCbFastLevels];
                lv.load_bt(bt);
                emit.ev(f32_to_bits(bt[1]), event_time<ABS_T>(bt[0], s.lastf, sc));
                if (p.m_old > 2u) {
                    emit.ev(f32_to_bits(bt[3]), event_time<ABS_T>(bt[2], s.lastf, sc));
                    if (p.m_old > 3u) {
                        emit.ev(f32_to_bits(bt[5]), event_time<ABS_T>(bt[4], s.lastf, sc));
                        if (p.m_old > 4u) {
                            emit.ev(f32_to_bits(bt[7]), event_time<ABS_T>(bt[6], s.lastf, sc));
                            for (uint32_t k = kCbFastLevels + 1u; k < p.m_old; ++k) {
                                const CbLevel l = lv.load_deep(k);
                                emit.ev(f32_to_bits(l.thr), event_time<ABS_T>(l.bdt, s.lastf, sc));
                            }
                        }
                    }
                }
            }
        }
    }
    if (L::lane(p.need_pop)) emit.ev(f32_to_bits(s.thr0), event_time<ABS_T>(s.bdt0, s.lastf, sc));
}

// pop_top_event's arena shift (:199-207): level 1, if there is one, becomes the root; deeper levels are dropped
// (popped + Collapse: they would never be visited, emitted or kept again).
template <class L, class Lv>
ADDER_HD void cb_pop(CbPxT<L> &s, const CbPlanT<L> &p, const Lv &lv) {
    if (L::lane(p.need_pop)) {
        if (s.m >= 2u) {
            const CbLevel l = lv.load_fast(1);
            s.S = fsub(s.S, fsub(l.F, l.thr));
            s.dt0 = fsub(s.dt0, l.Q);
            s.bdt0 = l.bdt;
            s.thr0 = l.thr;
            s.m = 1u;
        } else {
            s.m = 0u;
        }
    }
    s.popped = L::or_(s.popped, p.need_pop);
}

// ---------------------------------------------------------------------------------------
// CONSTANT-RUN STEP: the bounded Collapse regime (above) while c_thresh is 0 in EVERY frame (crf 0: c_thresh_baseline =
// c_thresh_max = 0, rate_controller.rs:9) and time_spanned is one integer constant.  Then the contrast test
// (video.rs:1338-1340) is `frame_val != base_val`: any change of value flushes the arena, so between two flushes a pixel
// integrates ONE intensity I, frame after frame -- and the whole arena is a function of (I, frames since the flush):
//   * a node that has been visited r >= 1 times holds integration r I and delta_t r T (event_pixel_tree.rs:449-451: a
//     firing accumulates like any other visit; I = 0 is the exception below), it fires at its first visit (a fresh
//     node's d is floor(log2 I), :332-335, 503-512) and then whenever r I crosses a power of two (:427, :452: d becomes
//     floor(log2 sum) + 1), so its LAST firing was at visit j = ceil(2^e / I), e = floor(log2(r I)), and its best event
//     is the firing arm (:427-447) evaluated there: (2^e's exponent, (j-1) T + T (2^e - (j-1) I) / I);
//   * a level exists since its parent's last firing (:342-356: the firing node gets a fresh child, deeper nodes are
//     dropped) and has been visited in every frame since (a shallower node that fires drops it), so level k+1 has run
//     r_{k+1} = r_k - j_k frames: the levels are the ROOT states of shorter and shorter runs, and the arena ends at the
//     first r_k = 0 (the pristine tail).
// So nothing but the root is stored or stepped: the levels' events are WORKED OUT when a flush wants them (one division
// each, the same correctly rounded operations on the same exact integers as the stepped arm), pop_top's new root
// (:199-207) is level 1 worked out the same way, and a launch writes the levels back in their resident form for whoever
// runs next.  Per unit: the root {S, delta_t, best delta_t, threshold}, base_val, popped_dtm, and r1 = frames since the
// root last fired (= level 1's run).  I = 0: a d = 128 node fires at every visit without accumulating (:449) -- no
// level ever exists behind it, r1 stays 0.  tests/cpu_sim runs these functions against the literal oracle, launch by
// launch, mixed with the bounded and the generic step.
// ---------------------------------------------------------------------------------------
template <class L>
struct CrPxT {
    float S, dt0, bdt0, thr0;  // the root (meaningful iff has)
    uint32_t base;
    uint32_t r1;               // frames since the root last fired: the run of level 1 (0: the root's child is the pristine tail)
    typename L::Mask has;      // the arena holds a fired root (m != 0)
    typename L::Mask popped;   // popped_dtm
    float lastf;
};
using CrPx = CrPxT<ScalarLanes>;

// A node that has been visited r >= 1 times at the constant intensity I >= 1: its best event and its last firing.
struct CrNode {
    float bdt, thr;   // best_event.delta_t; 2^d of the node (its exponent minus one is best_event.d)
    uint32_t j;       // the visit at which it last fired (1 <= j <= r)
};
ADDER_HD CrNode cr_node(float I, uint32_t r, float T) {
    const float rI = fmul((float)r, I);                           // exact: below 2^24 (cb_possible)
    const float p2 = bits_to_f32(f32_to_bits(rI) & 0x7f800000u);  // 2^floor(log2(r I))
    const float qj = fdiv_small(p2, I);                           // <= r: far from the next float, so ceil() is exact
    uint32_t j = (uint32_t)qj;
    j += (float)j < qj ? 1u : 0u;
    const float jm1 = (float)(j - 1u);
    const float q = fdiv_small(fsub(p2, fmul(jm1, I)), I);        // the firing arm at visit j (:431, :445)
    CrNode n;
    n.bdt = fadd(fmul(jm1, T), fmul(T, q));
    n.thr = fadd(p2, p2);
    n.j = j;
    return n;
}
// fired levels of an unpopped arena whose root has run since its last firing for r1 frames (root included)
ADDER_HD uint32_t cr_depth(float I, uint32_t r1, float T) {
    uint32_t m = 1u;
    for (uint32_t r = r1; r != 0u; ++m) r -= cr_node(I, r, T).j;
    return m;
}

template <class L>
ADDER_HD CrPxT<L> cr_unpack(uint32_t hdr, float integ, float dt, float bdt, float lastf, float T) {
    CrPxT<L> s;
    s.S = integ;
    s.dt0 = dt;
    s.bdt0 = bdt;
    s.thr0 = lean_thr_from_bd((hdr >> kHdrBdShift) & 0xffu);
    s.base = hdr & 0xffu;
    const bool has = hdr_m(hdr) != 0u, popped = (hdr & kHdrPopped) != 0u;
    s.has = L::from(has);
    s.popped = L::from(popped);
    s.lastf = lastf;
    // r1 from the root alone: it has run n = delta_t / T frames and last fired at ceil(2^(d-1) / I)
    s.r1 = 0u;
    if (has && !popped && s.base != 0u && hdr_m(hdr) > 1u) {
        const float I = (float)s.base;
        const float qj = fdiv_small(fmul(0.5f, s.thr0), I);
        uint32_t j = (uint32_t)qj;
        j += (float)j < qj ? 1u : 0u;
        s.r1 = (uint32_t)fdiv(dt, T) - j;
    }
    return s;
}
template <class L>
ADDER_HD uint32_t cr_hdr(const CrPxT<L> &s, float T) {
    const uint32_t m = !L::lane(s.has) ? 0u : (L::lane(s.popped) ? 1u : cr_depth((float)s.base, s.r1, T));
    return hdr_make(s.base, lean_bd_from_thr(f32_to_bits(s.thr0)), m, L::lane(s.popped));
}

constexpr uint32_t kCrKept = 4;  // levels whose events the step hands to the emission (deeper ones are worked out again)
template <class L>
struct CrPlanT {
    typename L::Mask flushed;    // pop_best_events ran and the arena held a fired level: events leave
    typename L::Mask collapsed;  // ... and was popped: root event + D_EMPTY filler (:249-265)
    typename L::Mask need_pop;   // pop_top_event follows the integrate
    float old_thr0, old_bdt0;    // the old root's best event (valid iff flushed)
    uint32_t old_base;           // the flushed run's intensity
    uint32_t n_lv;               // levels behind the flushed root (0 when collapsed)
    float lv_bdt[kCrKept], lv_thr[kCrKept];  // levels 1 .. min(n_lv, kCrKept): best delta_t, threshold
    uint32_t r_rest;             // the run of level kCrKept + 1 (0: there is none)
    uint32_t count;              // events of this unit this frame
};
using CrPlan = CrPlanT<ScalarLanes>;

// integrate_for_px (video.rs:1318-1380) of one unit under the conditions above: flush bookkeeping, the root's integrate
// (the firing arm of cb_step_b with the node = the root), need_to_pop_top.  `count_levels`: the caller wants plan.count
// to include the flushed levels (it costs the depth walk; the kernels need it for the ordered compaction).
template <class L>
ADDER_HD void cr_step(CrPxT<L> &s, uint32_t v, float T, const StepConsts &sc, CrPlanT<L> &p) {
    using M = typename L::Mask;
    const float I = (float)v;
    const M flush = L::from(v != s.base);  // c_thresh == 0
    p.flushed = L::and_(flush, s.has);
    p.collapsed = L::and_(p.flushed, s.popped);
    p.old_thr0 = s.thr0;
    p.old_bdt0 = s.bdt0;
    p.old_base = s.base;
    p.n_lv = 0u;
    p.r_rest = 0u;
    p.count = 0u;
    if (L::lane(p.flushed)) {
        if (L::lane(p.collapsed)) {
            p.count = 2u;
        } else {  // the levels behind the root, worked out once: their number places the unit's events, the emission takes them
            const float Io = (float)s.base;
            uint32_t r = s.r1;
            for (uint32_t k = 0; k < kCrKept; ++k) {  // (unrolled: the kept levels stay in registers)
                if (r != 0u) {
                    const CrNode n = cr_node(Io, r, T);
                    p.lv_bdt[k] = n.bdt;
                    p.lv_thr[k] = n.thr;
                    r -= n.j;
                    p.n_lv += 1u;
                }
            }
            p.r_rest = r;
            for (; r != 0u; p.n_lv += 1u) r -= cr_node(Io, r, T).j;  // deeper than kCrKept: rare (delta_t_max beyond ~30 frames)
            p.count = 1u + p.n_lv;
        }
    }
    const M has0 = L::andnot(s.has, flush);
    const M popped = L::andnot(s.popped, flush);
    s.base = L::lane(flush) ? v : s.base;
    const float S_old = L::lane(has0) ? s.S : 0.0f;
    const float dt_old = L::lane(has0) ? s.dt0 : 0.0f;
    const float S_new = fadd(S_old, I);
    const M fires = L::or_(L::not_(has0), L::from(S_new >= s.thr0));  // (the pristine tail at index 0 always fires)
    // the firing arm (:427-473) on the root
    const M zero = L::from(S_new == 0.0f);
    const float p2 = bits_to_f32(f32_to_bits(S_new) & 0x7f800000u);
    const M d128 = L::or_(L::andnot(L::from(I < 1.0f), has0), L::and_(has0, L::from(s.thr0 == 0.0f)));
    const float q = fdiv_small(fsub(p2, S_old), I);
    const float prop = L::lane(L::or_(zero, d128)) ? 1.0f : q;
    const float bdt = fadd(dt_old, fmul(T, prop));
    s.S = S_new;
    s.dt0 = L::lane(L::and_(fires, zero)) ? dt_old : fadd(dt_old, T);
    s.bdt0 = L::lane(fires) ? bdt : s.bdt0;
    s.thr0 = L::lane(fires) ? fadd(p2, p2) : s.thr0;
    // level 1: reborn as the pristine tail when the root fires; visited (it, or a deeper node, fires or accumulates)
    // otherwise -- unless the arena is popped, where only the root is visited (:360-362) and no level is kept
    s.r1 = L::lane(L::or_(fires, popped)) ? 0u : s.r1 + 1u;
    s.has = L::from(true);
    s.popped = popped;
    p.need_pop = L::andnot(L::from(s.dt0 >= sc.dtm_f), popped);  // :394-396
    p.count += L::lane(p.need_pop) ? 1u : 0u;
}

// The unit's events of this frame, in emission order (after cr_step, before cr_pop): emit.ev(threshold bits, t) /
// emit.filler(t) as for cb_emit.
template <bool ABS_T, class L, class Emit>
ADDER_HD void cr_emit(CrPxT<L> &s, const CrPlanT<L> &p, float T, const StepConsts &sc, Emit &emit) {
    if (L::lane(p.flushed)) {
        if (L::lane(p.collapsed)) {
            emit.ev(f32_to_bits(p.old_thr0), f32_as_u32(ABS_T ? fadd(p.old_bdt0, s.lastf) : p.old_bdt0));
            s.lastf = sc.running_t;  // :257
            emit.filler(sc.running_t_u32);
        } else {
            emit.ev(f32_to_bits(p.old_thr0), event_time<ABS_T>(p.old_bdt0, s.lastf, sc));
            // levels 1, 2, ..: the root states of runs of r1, r2, .. frames (each step narrows the active lanes)
            if (p.n_lv > 0u) {
                emit.ev(f32_to_bits(p.lv_thr[0]), event_time<ABS_T>(p.lv_bdt[0], s.lastf, sc));
                if (p.n_lv > 1u) {
                    emit.ev(f32_to_bits(p.lv_thr[1]), event_time<ABS_T>(p.lv_bdt[1], s.lastf, sc));
                    if (p.n_lv > 2u) {
                        emit.ev(f32_to_bits(p.lv_thr[2]), event_time<ABS_T>(p.lv_bdt[2], s.lastf, sc));
                        if (p.n_lv > 3u) {
                            emit.ev(f32_to_bits(p.lv_thr[3]), event_time<ABS_T>(p.lv_bdt[3], s.lastf, sc));
                            const float Io = (float)p.old_base;
                            for (uint32_t r = p.r_rest; r != 0u;) {
                                const CrNode n = cr_node(Io, r, T);
                                emit.ev(f32_to_bits(n.thr), event_time<ABS_T>(n.bdt, s.lastf, sc));
                                r -= n.j;
                            }
                        }
                    }
                }
            }
        }
    }
    if (L::lane(p.need_pop)) emit.ev(f32_to_bits(s.thr0), event_time<ABS_T>(s.bdt0, s.lastf, sc));
}

// pop_top_event's arena shift (:199-207): level 1, if there is one, becomes the root; deeper levels are dropped.
template <class L>
ADDER_HD void cr_pop(CrPxT<L> &s, const CrPlanT<L> &p, float T) {
    // (masks are wave-wide under WaveLanes: they are combined outside of any per-lane branch)
    const typename L::Mask promote = L::and_(p.need_pop, L::from(s.r1 != 0u));
    if (L::lane(promote)) {
        const float I = (float)s.base;
        const CrNode n = cr_node(I, s.r1, T);
        s.S = fmul((float)s.r1, I);
        s.dt0 = fmul((float)s.r1, T);
        s.bdt0 = n.bdt;
        s.thr0 = n.thr;
    }
    s.has = L::andnot(s.has, L::andnot(p.need_pop, promote));  // a root popped off an arena without levels leaves the tail
    s.r1 = L::lane(p.need_pop) ? 0u : s.r1;
    s.popped = L::or_(s.popped, p.need_pop);
}

// ---------------------------------------------------------------------------------------
// LEAN RUNS: the lean regime (Collapse, delta_t_max <= time_spanned) under the constant-run conditions above (c_thresh
// 0 throughout, one integer time_spanned), DeltaT.  There the arena holds the root or nothing, the root's whole state
// is cr_node(base_val, rho) with rho = the frames it has accumulated -- so a unit is THREE SMALL INTEGERS {base_val, rho,
// popped_dtm}, the step is compares and a counter, and the one event that needs arithmetic (A, the flushed root's best
// event) is worked out by the EXPANSION from (base_val, rho), densely, like it already works event C out from the
// input byte.  The parked 8-byte record is {rho of the flushed root, unit | popped_dtm << 7 | flushed base_val << 8 |
// input byte << 16}.  rho I and rho T stay below 2^24 (the host switches to lean_step before they would not: it bounds
// the longest run by what the kernel reports, BatchArgs::run_max -- 65 793 frames), so the closed form's operations are the
// stepped ones bit for bit.
// ---------------------------------------------------------------------------------------
// popped_dtm is no state of its own here: in this regime a root of non-zero intensity is popped in the very frame it
// starts (it has accumulated time_spanned >= delta_t_max), a black one never is, a flush clears the flag and the same
// frame's integrate sets it again -- after every frame popped_dtm == (base_val != 0).  (adder_lr_kernel checks the planes
// it loads against this and reports a violation instead of stepping on.)
template <class L>
struct LrPxT {
    uint32_t base, rho;  // rho: frames the root has accumulated; 0 = no root (the pristine tail alone)
};
using LrPx = LrPxT<ScalarLanes>;
// record: w0 = rho of the flushed root, w8 = unit | flushed base_val << 8 | input << 16; which of the events A, B, C it
// stands for follows from these (lr_decode8) -- the frame kernel assembles no flag bits, its event count comes from the
// masks.  (An all-zero word decodes to no events.)
constexpr uint32_t kLrBaseShift = 8, kLrInShift = 16;

template <class L>
ADDER_HD LrPxT<L> lr_unpack(uint32_t hdr, float dt, float T, bool &consistent) {
    LrPxT<L> p;
    p.base = hdr & 0xffu;
    // a root of intensity I > 0 holds delta_t = rho T; a black one (it never accumulates, :449) counts as one frame
    p.rho = hdr_m(hdr) == 0u ? 0u : (p.base != 0u ? (uint32_t)fdiv(dt, T) : 1u);
    consistent = ((hdr & kHdrPopped) != 0u) == (p.base != 0u);
    return p;
}
// integrate_for_px under the conditions above: which events leave, the record's two words, the new state.
// pair = base_val << 8 | v << 16 (the kernel builds it with one byte permute of the previous and the current input
// words); nz_old = the previous frame's (v != 0), i.e. popped_dtm before this frame
template <class L>
ADDER_HD LeanFlagsT<L> lr_step(LrPxT<L> &p, uint32_t v, uint32_t pair, uint32_t unit, typename L::Mask &nz_old, uint32_t &w0,
                               uint32_t &w8) {
    using M = typename L::Mask;
    const M flush = L::from(v != p.base);
    const M has = L::from(p.rho != 0u);
    const M zero = L::from(v == 0u);  // the root's sum stays 0 exactly when the run's intensity is 0 (:449)
    LeanFlagsT<L> fl;
    fl.a = L::and_(flush, has);          // the flushed root's best event
    fl.b = L::and_(fl.a, nz_old);        // ... and the D_EMPTY filler if it had been popped
    fl.c = L::andnot(flush, zero);       // need_to_pop_top: a new root of non-zero intensity (an old one is popped already)
    w0 = p.rho;
    w8 = pair | unit;
    p.base = v;  // (unchanged without a flush)
    // rho: 0 where pop_top takes the new root, rho + 1 where a root goes on accumulating (or the tail starts one, 0 + 1),
    // 1 for a black root
    const uint32_t grown = L::lane(flush) ? 0u : p.rho + 1u;
    p.rho = L::lane(zero) ? 1u : grown;
    nz_old = L::not_(zero);
    return fl;
}
// k consecutive frames whose byte equals base_val, at once: no flush, no event (lr_step's three masks stay clear), the
// root goes on accumulating -- or stays the one-frame black root.  (nz_old and last_fired_t / T do not move either.)
template <class L>
ADDER_HD void lr_quiet_run(LrPxT<L> &p, uint32_t k) {
    p.rho = p.base != 0u ? p.rho + k : 1u;
}
// AbsoluteT (time_spanned == ref_time >= 255: the run records' condition and argument): lq = last_fired_t / T rides along
// as an integer.  A's time stamp is (best delta_t + lq T); the filler B sets last_fired_t to the frame's running_t (:257),
// a black root's one firing advances it by one frame; C (the new root popped at once, best delta_t in (T / 2, T]) by one
// more.  The 12-byte record is {rho, lq before the frame, the DeltaT record's second word}.
template <class L>
ADDER_HD void lr_step_lq(uint32_t &lq, const LeanFlagsT<L> &fl, uint32_t frame_idx, uint32_t &w1) {
    w1 = lq;
    const uint32_t after_a = L::lane(fl.b) ? frame_idx : lq + 1u;
    lq = L::lane(fl.a) ? after_a : lq;
    lq += L::lane(fl.c) ? 1u : 0u;
}
ADDER_HD LeanEvents lr_decode12(uint32_t w0, uint32_t w1, uint32_t w8, float T, uint32_t running_t_u32, uint32_t frame_idx) {
    LeanEvents e;
    const uint32_t Io = (w8 >> kLrBaseShift) & 0xffu, Iv = (w8 >> kLrInShift) & 0xffu;
    const bool flush = Io != Iv;
    e.a = flush && w0 != 0u;
    e.b = e.a && Io != 0u;
    e.c = flush && Iv != 0u;
    const CrNode n = cr_node(Io != 0u ? (float)Io : 1.0f, w0 != 0u ? w0 : 1u, T);
    e.da = Io != 0u ? lean_bd_from_thr(f32_to_bits(n.thr)) : kDZero;
    e.ta = f32_as_u32(fadd(Io != 0u ? n.bdt : T, fmul((float)w1, T)));
    e.tb = running_t_u32;
    const uint32_t lq_c = e.a ? (e.b ? frame_idx : w1 + 1u) : w1;
    const float I = (float)Iv;
    const float p2 = bits_to_f32(f32_to_bits(I) & 0x7f800000u);  // 2^get_d(I)
    e.dc = get_d(I);
    e.tc = f32_as_u32(fadd(fadd(0.0f, fmul(T, fdiv_small(fsub(p2, 0.0f), I))), fmul((float)lq_c, T)));  // (C exists only for I >= 1)
    return e;
}
// The record back into events: A from the flushed root's run, C from the input byte (lean_decode8's arithmetic).
ADDER_HD LeanEvents lr_decode8(uint32_t w0, uint32_t w8, float T, uint32_t running_t_u32) {
    LeanEvents e;
    const uint32_t Io = (w8 >> kLrBaseShift) & 0xffu, Iv = (w8 >> kLrInShift) & 0xffu;
    const bool flush = Io != Iv;
    e.a = flush && w0 != 0u;
    e.b = e.a && Io != 0u;
    e.c = flush && Iv != 0u;
    // (a record without A still decodes: rho 0 is given a frame so that the divisions stay inside their domain)
    const CrNode n = cr_node(Io != 0u ? (float)Io : 1.0f, w0 != 0u ? w0 : 1u, T);
    e.da = Io != 0u ? lean_bd_from_thr(f32_to_bits(n.thr)) : kDZero;
    e.ta = f32_as_u32(Io != 0u ? n.bdt : T);  // a black root's best event: (128, 0 + T * 1)
    e.tb = running_t_u32;
    const float I = (float)Iv;
    const float p2 = bits_to_f32(f32_to_bits(I) & 0x7f800000u);  // 2^get_d(I)
    e.dc = get_d(I);
    e.tc = f32_as_u32(fmul(T, fdiv_small(p2, I)));  // 0 + T (2^d - 0) / I: both identities on positive values (C exists only for I >= 1)
    return e;
}
// The same through a table: the expansion is bound by vector instructions, and the two events' arithmetic (three
// divisions, the reciprocals at a quarter of the rate) is most of a record's.  tab[I * kLrTabRuns + rho] = d << 24 | t of
// event A for rho < kLrTabRuns (I = 0: the black root's (128, T)), tab[256 * kLrTabRuns + v] the same of event C; every t
// fits 24 bits (rho T < 2^24 is the regime's own condition).  Longer runs are worked out (a branch no lane takes in most
// rounds).  Built by lr_build_tab with the very functions above: equal bit for bit by construction, and tests/cpu_sim
// decodes through it against the oracle.
constexpr uint32_t kLrTabRuns = 32, kLrTabWords = 256u * kLrTabRuns + 256u;
ADDER_HD void lr_build_tab(uint32_t *tab, float T) {
    for (uint32_t I = 0; I < 256u; ++I) {
        for (uint32_t rho = 0; rho < kLrTabRuns; ++rho) {
            const LeanEvents e = lr_decode8(rho, (I << kLrBaseShift) | ((I ^ 1u) << kLrInShift), T, 0u);
            tab[I * kLrTabRuns + rho] = (e.da << 24) | (e.ta & 0xffffffu);
        }
        const LeanEvents e = lr_decode8(0u, (0u << kLrBaseShift) | (I << kLrInShift), T, 0u);
        tab[256u * kLrTabRuns + I] = I != 0u ? (e.dc << 24) | (e.tc & 0xffffffu) : 0u;
    }
}
// tab_a: the event-A rows (may be null: then A is always worked out -- per-lane gathers from global memory cost the
// expansion more than the divisions they replace, so the kernel keeps only the 256 event-C words, in LDS)
ADDER_HD LeanEvents lr_decode8_tab(uint32_t w0, uint32_t w8, float T, uint32_t running_t_u32, const uint32_t *tab_a,
                                   const uint32_t *tab_c) {
    LeanEvents e;
    const uint32_t Io = (w8 >> kLrBaseShift) & 0xffu, Iv = (w8 >> kLrInShift) & 0xffu;
    const bool flush = Io != Iv;
    e.a = flush && w0 != 0u;
    e.b = e.a && Io != 0u;
    e.c = flush && Iv != 0u;
    const uint32_t xc = tab_c[Iv];
    if (tab_a != nullptr && w0 < kLrTabRuns) {
        const uint32_t xa = tab_a[Io * kLrTabRuns + w0];
        e.da = xa >> 24;
        e.ta = xa & 0xffffffu;
    } else {
        const CrNode n = cr_node(Io != 0u ? (float)Io : 1.0f, w0 != 0u ? w0 : 1u, T);
        e.da = Io != 0u ? lean_bd_from_thr(f32_to_bits(n.thr)) : kDZero;
        e.ta = f32_as_u32(Io != 0u ? n.bdt : T);
    }
    e.tb = running_t_u32;
    e.dc = xc >> 24;
    e.tc = xc & 0xffffffu;
    return e;
}
// back to the resident form {header, integration, delta_t, best delta_t}
template <class L>
ADDER_HD uint32_t lr_pack(const LrPxT<L> &p, float T, float &integ, float &dt, float &bdt) {
    const float I = (float)p.base;
    uint32_t bd = 0u;
    integ = dt = bdt = 0.0f;
    if (p.rho != 0u) {
        if (p.base != 0u) {
            const CrNode n = cr_node(I, p.rho, T);
            integ = fmul((float)p.rho, I);
            dt = fmul((float)p.rho, T);
            bdt = n.bdt;
            bd = lean_bd_from_thr(f32_to_bits(n.thr));
        } else {
            bdt = T;
            bd = kDZero;
        }
    }
    return hdr_make(p.base, bd, p.rho != 0u ? 1u : 0u, p.base != 0u);  // (popped_dtm == (base_val != 0): see LrPxT)
}

// ---------------------------------------------------------------------------------------
// LEAN RUNS, PACKED (adder_lp_kernel): lr_step on FOUR units per 32-bit word -- the lane's four input bytes of a frame
// are never taken apart.  Booleans are bit 7 of the unit's byte (kLpK8), so one vector instruction decides four units:
//   flush  h = nzb(vin ^ prev)                 (video.rs:1338-1340 with c_thresh 0: any change of value)
//   A      = h & ~hnp      hnp: the unit holds NO root (rho == 0) -- exactly the units that flushed to a non-zero value in
//                          the frame before (the new root was popped at once, lr_step) and the pristine ones of a fresh stream
//   B      = A & nzp       nzp: base_val != 0 (== popped_dtm, LrPxT)
//   C      = h & nz(vin)   and that is the next frame's hnp
// The events of a frame are the set bits of A, B and C (v_bcnt), a unit that does not change costs nothing of its own.
// rho is not counted: every unit remembers the launch-relative frame `start` of its last flush, a record carries
// rho' = i - start, which is rho + 1 for a root of non-zero intensity (start = -rho - 1 when the launch begins) and >= 1
// for a black root (start = -rho; it never accumulates, event_pixel_tree.rs:449) -- lp_rho() undoes it for lr_decode8.
// Record: w0 = rho', w8 = unit (8 bits: a wave steps a PAIR of segments) | flushed base_val << 8 | input << 16.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kLpK7 = 0x7f7f7f7fu, kLpK8 = 0x80808080u;
ADDER_HD uint32_t lp_nzb(uint32_t w) { return (((w & kLpK7) + kLpK7) | w) & kLpK8; }  // bit 7 of every non-zero byte
struct LpWord {
    uint32_t prev;      // the four base_vals (= the previous frame's input bytes)
    uint32_t nzp, hnp;  // bit 7 per byte: base_val != 0; no root
    uint32_t start[4];  // (two's complement) frame of the last flush, relative to the launch's first frame
};
struct LpMasks {
    uint32_t h, a, b, c;  // bit 7 per byte: flush, events A, B, C
};
// the word's state from the resident planes' (base_val, rho) of its four units (lr_unpack)
ADDER_HD void lp_init(LpWord &s, const LrPx (&p)[4]) {
    s.prev = s.nzp = s.hnp = 0u;
    for (uint32_t j = 0; j < 4u; ++j) {
        s.prev |= p[j].base << (8u * j);
        s.nzp |= (p[j].base != 0u ? 0x80u : 0u) << (8u * j);
        s.hnp |= (p[j].rho == 0u ? 0x80u : 0u) << (8u * j);
        s.start[j] = 0u - p[j].rho - (p[j].base != 0u ? 1u : 0u);
    }
}
ADDER_HD LpMasks lp_step(LpWord &s, uint32_t vin) {
    LpMasks m;
    m.h = lp_nzb(vin ^ s.prev);
    const uint32_t nzv = lp_nzb(vin);
    m.c = m.h & nzv;
    m.a = m.h & ~s.hnp;
    m.b = m.a & s.nzp;
    s.prev = vin;
    s.nzp = nzv;
    s.hnp = m.c;
    return m;
}
// frames whose four bytes equal the base_vals: nothing leaves, every root goes on accumulating (or comes into being)
ADDER_HD void lp_quiet(LpWord &s) { s.hnp = 0u; }
// unit j's record of frame i: base_w = the word's base_vals BEFORE the frame's lp_step, start_j = its start then (the
// caller sets start[j] = i afterwards)
ADDER_HD void lp_record(uint32_t base_w, uint32_t vin, uint32_t start_j, uint32_t j, uint32_t i, uint32_t unit8, uint32_t &w0,
                        uint32_t &w8) {
    w0 = i - start_j;
    w8 = unit8 | (((base_w >> (8u * j)) & 0xffu) << kLrBaseShift) | (((vin >> (8u * j)) & 0xffu) << kLrInShift);
}
ADDER_HD uint32_t lp_rho(uint32_t w0, uint32_t w8) { return w0 - (((w8 >> kLrBaseShift) & 0xffu) != 0u ? 1u : 0u); }
// The PARKED form is four bytes: unit | base_val << 8 | input << 16 | min(rho', 255) << 24.  A unit that flushed in this
// launch has rho' <= the launch's length; only a run carried in from earlier launches can reach 255 -- then the record says
// 255 and the full rho' is an ESCAPE word: the pair's k-th escaping record of the frame keeps it 4 (k + 1) bytes below the
// end of the pair's two slots (a pair parks at most 256 records of 4 bytes and 256 escapes: its 2 KB).
constexpr uint32_t kLpRhoEsc = 255u, kLpRhoShift = 24u;
ADDER_HD uint32_t lp_park4(uint32_t w0, uint32_t w8) { return (w8 & 0xffffffu) | ((w0 < kLpRhoEsc ? w0 : kLpRhoEsc) << kLpRhoShift); }
ADDER_HD bool lp_escapes(uint32_t w4) { return (w4 >> kLpRhoShift) == kLpRhoEsc; }
ADDER_HD void lp_unpark4(uint32_t w4, uint32_t esc, uint32_t &w0, uint32_t &w8) {
    w0 = lp_escapes(w4) ? esc : w4 >> kLpRhoShift;
    w8 = w4 & 0xffffffu;
}
// (base_val, rho) of unit j after nb frames of the launch
ADDER_HD LrPx lp_final(const LpWord &s, uint32_t j, uint32_t nb) {
    LrPx p;
    p.base = (s.prev >> (8u * j)) & 0xffu;
    const uint32_t d = nb - s.start[j];
    p.rho = p.base != 0u ? d - 1u : (d != 0u ? 1u : 0u);
    return p;
}

// ---------------------------------------------------------------------------------------
// RUN RECORDS: the bounded Collapse regime (delta_t_max > time_spanned) under the constant-run conditions, with the
// whole step in integers -- what LEAN RUNS does for the lean regime.  A unit is {base_val, n, popped_dtm} (n = the frames
// the root has accumulated) and, in AbsoluteT, lq = last_fired_t / T: nothing else is state -- an unpopped root of n frames
// last fired at j = cr_node(base_val, n).j, so level 1 has run n - j frames, and so on down the chain.  The frame kernel
// only decides WHAT happens -- a flush of (base_val, n), a collapsed flush, pop_top -- counts the events (a byte table of
// chain lengths) and parks one record per happening; the expansion works the events out, densely: the flushed arena is
// the chain r_0 = n, r_{k+1} = r_k - j_k of cr_node(base_val, r_k).
// AbsoluteT needs last_fired_t per unit, and it stays an integer too when time_spanned == ref_time >= 255 (every framed
// source, framed.rs:101-109): last_fired_t is a multiple L T, an event's best delta_t lies in ((j-1) T, j T] with
// T q >= T / 255 >= 1, so t = trunc(bdt + L T) lies in [L T + (j-1) T + 1, L T + j T] and delta_t_to_absolute_t's
// round-up (:122-129) gives (L + j) T: every event advances L by its node's last firing j.  Over a flushed chain the j_k
// telescope to n (the run's length), pop_top advances L by the root's last firing, a collapsed flush sets L to the
// frame's index (:257).  The expansion redoes the same chain to give event k its own L_k.  tests/cpu_sim checks every
// intensity, run length and both time modes against the literal oracle.
// record {n, lq, kind | unit << 2 | base_val << 9 | events << 17} (DeltaT: {n, the third word}): kind 1 flush, 2 collapsed
// flush, 3 pop_top
// ---------------------------------------------------------------------------------------
constexpr uint32_t kRrFlush = 1u, kRrCollapsed = 2u, kRrPop = 3u, kRrFlushPop = 0u;  // (kind 0 with a count: a flush, then the new root's pop)
constexpr uint32_t kRrPopInShift = 24, kRrRunMask = 0xffffffu;  // first word: n | the popped root's intensity << 24 (kRrFlushPop)
constexpr uint32_t kRrUnitShift = 2, kRrBaseShift = 9, kRrCountShift = 17;
constexpr uint32_t kRrTabRows = 32;  // chain lengths tabulated for runs below 32 frames (delta_t_max up to 32 frames; beyond: worked out)
// events of an unpopped arena whose root has accumulated r frames of intensity I (0 for r == 0; a black root holds one)
ADDER_HD uint32_t rr_chain(uint32_t I, uint32_t r, float T) { return r == 0u ? 0u : I == 0u ? 1u : cr_depth((float)I, r, T) - 1u; }
ADDER_HD void rr_build_tab(uint8_t *tab, float T) {  // tab[I * kRrTabRows + r] (the time step does not enter: j = ceil(2^e / I))
    for (uint32_t I = 0; I < 256u; ++I)
        for (uint32_t r = 0; r < kRrTabRows; ++r) tab[I * kRrTabRows + r] = (uint8_t)rr_chain(I, r, T);
}
template <class L>
struct RrPxT {
    uint32_t base, n, lq;
    typename L::Mask popped;
};
using RrPx = RrPxT<ScalarLanes>;

template <class L>
ADDER_HD RrPxT<L> rr_unpack(uint32_t hdr, float dt, float lastf, float T, bool abs_t) {
    RrPxT<L> p;
    p.base = hdr & 0xffu;
    p.popped = L::from((hdr & kHdrPopped) != 0u);
    p.n = hdr_m(hdr) == 0u ? 0u : (p.base != 0u ? (uint32_t)fdiv(dt, T) : 1u);
    p.lq = abs_t ? (uint32_t)fdiv(lastf, T) : 0u;
    return p;
}
// One frame of one unit: the record's three words (valid iff count != 0) and the number of events it stands for.
// frame_idx = running_t / T before this frame; n_pop = ceil(delta_t_max / T) >= 2; tab(I, r) = rr_chain for r < kRrTabRows.
// Mode Normal (collapse = false) differs in two places only: a flush after the pop still emits the whole chain (no
// root + filler), and the levels below a popped root go on being visited -- pop_top leaves [level 1, level 2, ..], which IS
// the arena of a run of n - j frames, so the same {base_val, n} describes it.
template <bool ABS_T, class L, class Tab>
ADDER_HD void rr_step(RrPxT<L> &p, uint32_t v, uint32_t frame_idx, uint32_t n_pop, float T, const Tab &tab, uint32_t tag,
                      uint32_t &w0, uint32_t &w1, uint32_t &w2, uint32_t &count, bool collapse = true) {
    using M = typename L::Mask;
    const M flush = L::from(v != p.base);
    const M collapsed = L::and_(L::and_(L::and_(flush, p.popped), L::from(p.n != 0u)), L::from(collapse));
    // the flushed arena's events (pop_best_events, :210-286): the chain below an unpopped root, root + filler if popped
    uint32_t chain = tab(p.base, p.n < kRrTabRows - 1u ? p.n : kRrTabRows - 1u);
    if (L::lane(L::andnot(flush, L::and_(p.popped, L::from(collapse)))) && p.n >= kRrTabRows) chain = rr_chain(p.base, p.n, T);  // (runs beyond the table)
    count = L::lane(flush) ? (L::lane(collapsed) ? 2u : chain) : 0u;
    uint32_t kind = L::lane(collapsed) ? kRrCollapsed : kRrFlush;
    uint32_t rn = p.n, rbase = p.base;
    w1 = p.lq;
    if (ABS_T) {
        const uint32_t lq_f = L::lane(collapsed) ? frame_idx : p.lq + p.n;  // (:257 / the chain's firings telescope to n)
        p.lq = L::lane(flush) ? lq_f : p.lq;
    }
    const M popped1 = L::andnot(p.popped, flush);
    const M zero = L::from(v == 0u);
    const uint32_t n1 = L::lane(L::or_(flush, zero)) ? 1u : p.n + 1u;  // (a black root holds one frame, :427-473 with intensity 0)
    const M need_pop = L::andnot(L::andnot(L::from(n1 >= n_pop), popped1), zero);  // :394-396
    uint32_t nn = n1;
    if (L::lane(need_pop)) {  // pop_top (:156-197): the root's event; level 1 becomes the root, deeper levels are dropped
        const uint32_t j = cr_node((float)v, n1, T).j;
        nn = n1 - j;
        if (ABS_T) p.lq += j;
        if (count != 0u) {
            // a frame that flushed AND pops: only with delta_t_max <= time_spanned (n_pop = 1: Mode Normal's lean case), where
            // the new run's root (v, 1) is popped at once.  One record for both: the flushed chain, then that root's event
            count += 1u;
            kind = kRrFlushPop;
            rn |= v << kRrPopInShift;
        } else {
            count = 1u;
            kind = kRrPop;
            rn = n1;
            rbase = v;
        }
    }
    p.n = nn;
    p.base = v;
    p.popped = L::or_(popped1, need_pop);
    w0 = rn;
    w2 = kind | tag | (rbase << kRrBaseShift) | (count << kRrCountShift);
}
// Event k of a record (n, lq, kind, base_val), from the record alone: k hops down the chain (one division each), then
// the node's event (two).  Event k's own last_fired_t is (lq + n - r_k) T: the firings above it telescope.
struct RrEvent {
    uint32_t d, t;
};
template <bool ABS_T>
// (w0 = the record's first word, cnt its event count: kRrFlushPop's last event is the popped root's)
ADDER_HD RrEvent rr_event_at(uint32_t kind, uint32_t Iu, uint32_t w0, uint32_t lq, uint32_t k, uint32_t cnt, float T,
                             uint32_t running_t_u32) {
    RrEvent e;
    const uint32_t n = w0 & kRrRunMask;
    if (kind == kRrCollapsed && k != 0u) {  // the D_EMPTY filler of a collapsed flush (:258-264)
        e.d = kDEmpty;
        e.t = running_t_u32;
        return e;
    }
    if (kind == kRrFlushPop && k + 1u == cnt) {  // the new run's root (v, 1), popped in the frame it started
        const CrNode nd = cr_node((float)(w0 >> kRrPopInShift), 1u, T);
        e.d = lean_bd_from_thr(f32_to_bits(nd.thr));
        e.t = f32_as_u32(ABS_T ? fadd(nd.bdt, fmul((float)(lq + n), T)) : nd.bdt);
        return e;
    }
    float bdt = T;  // a black root: (D_ZERO, time_spanned)
    uint32_t r = n;
    e.d = kDZero;
    if (Iu != 0u) {
        const float I = (float)Iu;
        for (uint32_t h = 0; h < k; ++h) r -= cr_node(I, r, T).j;
        const CrNode nd = cr_node(I, r, T);
        bdt = nd.bdt;
        e.d = lean_bd_from_thr(f32_to_bits(nd.thr));
    }
    e.t = f32_as_u32(ABS_T ? fadd(bdt, fmul((float)(lq + (n - r)), T)) : bdt);
    return e;
}
// The levels 1 .. m-1 of an unpopped arena in their resident form {integration, delta_t, best_delta_t, best_d}, for
// the planes the other steps read: store(k, Node).  Returns m.
template <class L, class Store>
ADDER_HD uint32_t cr_materialize(const CrPxT<L> &s, float T, Store &store) {
    if (!L::lane(s.has)) return 0u;
    uint32_t m = 1u;
    if (L::lane(s.popped)) return m;
    const float I = (float)s.base;
    for (uint32_t r = s.r1; r != 0u; ++m) {
        const CrNode n = cr_node(I, r, T);
        Node nd;
        nd.integ = fmul((float)r, I);
        nd.dt = fmul((float)r, T);
        nd.bdt = n.bdt;
        nd.bd = lean_bd_from_thr(f32_to_bits(n.thr));
        store(m, nd);
        r -= n.j;
    }
    return m;
}

// back to the resident form: the root's planes; the levels through store(k, Node) (cr_materialize's contract); returns the header
template <class L, class Store>
ADDER_HD uint32_t rr_pack(const RrPxT<L> &p, float T, float &integ, float &dt, float &bdt, float &lastf, Store &store,
                          bool collapse = true) {
    CrPxT<L> c;
    c.base = p.base;
    c.r1 = 0u;
    c.has = L::from(p.n != 0u);
    c.popped = L::and_(p.popped, L::from(collapse));  // (Normal: the levels below a popped root are kept and stepped)
    c.S = c.dt0 = c.bdt0 = c.thr0 = 0.0f;
    uint32_t bd = 0u;
    if (p.n != 0u) {
        if (p.base != 0u) {
            const CrNode n = cr_node((float)p.base, p.n, T);
            c.S = fmul((float)p.n, (float)p.base);
            c.dt0 = fmul((float)p.n, T);
            c.bdt0 = n.bdt;
            c.thr0 = n.thr;
            c.r1 = L::lane(c.popped) ? 0u : p.n - n.j;
            bd = lean_bd_from_thr(f32_to_bits(n.thr));
        } else {
            c.bdt0 = T;
            bd = kDZero;
        }
    }
    const uint32_t m = cr_materialize<L>(c, T, store);
    integ = c.S;
    dt = c.dt0;
    bdt = c.bdt0;
    lastf = fmul((float)p.lq, T);
    return hdr_make(p.base, bd, m < kHdrMMask ? m : kHdrMMask, L::lane(p.popped));
}

// ---------------------------------------------------------------------------------------
// GENERAL ARENA STEP -- Mode::Continuous (the mode of the event-camera sources,
// prophesee.rs:65, davis.rs:116-117, and of Video::integrate_matrix when they feed it frames).
// There a firing node hands the REST of the intensity and time to its child, which keeps integrating it
// (event_pixel_tree.rs:340-385, 468-471), so the tail is no longer pristine, a flush can meet nodes that
// hold time but no intensity (zero events, :96-111, :225-230), pop_best is followed by
// set_d_for_continuous (:289-312) and pop_top can meet a root without a best event (:156-197).  None of
// the FramePerfect reductions above hold: this step keeps the reference's node as it is
// {d, integration, delta_t, best_event} and walks the arena through an accessor
// (acc.load(k) -> ANode, acc.store(k, ANode)), emitting through emit(d, t).
// tests/cpu_sim runs it against the literal oracle; the reference's 13 PixelArena unit tests pin the oracle.
// ---------------------------------------------------------------------------------------
struct ANode {
    float integ, dt, bdt;
    uint32_t d, bd;
    bool has_best;
};
// node meta word in HBM: d | best_d << 8 | has_best << 16
ADDER_HD uint32_t anode_meta(const ANode &n) { return n.d | (n.bd << 8) | (n.has_best ? 1u << 16 : 0u); }
ADDER_HD void anode_set_meta(ANode &n, uint32_t w) {
    n.d = w & 0xffu;
    n.bd = (w >> 8) & 0xffu;
    n.has_best = ((w >> 16) & 1u) != 0u;
}
ADDER_HD ANode anode_new(float intensity) {  // PixelNode::new (:501-514)
    ANode n;
    n.integ = 0.0f;
    n.dt = 0.0f;
    n.bdt = 0.0f;
    n.d = get_d(intensity);
    n.bd = 0u;
    n.has_best = false;
    return n;
}
// arena header word in HBM (continuous contexts): base_val | length << 8 | popped_dtm << 16
struct APx {
    uint32_t base, length;
    bool popped;
    float lastf;
};
ADDER_HD uint32_t apx_hdr(const APx &s) { return s.base | (s.length << 8) | (s.popped ? 1u << 16 : 0u); }
ADDER_HD APx apx_unpack(uint32_t hdr, float lastf) {
    APx s;
    s.base = hdr & 0xffu;
    s.length = (hdr >> 8) & 0xffu;
    s.popped = ((hdr >> 16) & 1u) != 0u;
    s.lastf = lastf;
    return s;
}

// delta_t_to_absolute_t with Mode::Continuous (:113-137): no rounding of last_fired_t
template <bool ABS_T>
ADDER_HD uint32_t cont_event_time(float ev_dt, float &lastf) {
    if (ABS_T) {
        ev_dt = fadd(ev_dt, lastf);
        lastf = ev_dt;
    }
    return f32_as_u32(ev_dt);
}

// integrate_main (:418-479), Mode::Continuous.  Returns true if the node fired; the remainder for the child
// is left in (next_i, next_t).
ADDER_HD bool cont_integrate_main(ANode &n, float intensity, float time, float &next_i, float &next_t) {
    const float s = fadd(n.integ, intensity);
    if (s >= pow2_d(n.d)) {
        const uint32_t nd = get_d(s);
        float prop = fdiv(fsub(pow2_d(nd), n.integ), intensity);
        if (nd == kDZero || n.d == kDZero || intensity < 1.1920929e-7f) prop = 1.0f;
        n.d = nd;
        n.has_best = true;
        n.bd = nd;
        n.bdt = fadd(n.dt, fmul(time, prop));
        if (nd < kDMax) {
            n.integ = s;
            n.dt = fadd(n.dt, time);
            // loop { d += 1; if D_SHIFT[d] > integration as u128 { break } } (:454-460); D_SHIFT[128] = 0
            uint32_t d = nd;
            const float tr = (float)(uint64_t)n.integ;  // trunc for the values reachable here (< 2^63)
            for (;;) {
                d += 1u;
                if (d >= 128u || pow2_d(d) > tr) break;
            }
            n.d = d > 128u ? 128u : d;
        }
        const float rest = fsub(intensity, fmul(intensity, prop));
        if (rest >= 0.0f) {
            next_i = rest;
            next_t = fsub(time, fmul(time, prop));
        } else {
            next_i = 0.0f;
            next_t = 0.0f;
        }
        return true;
    }
    n.integ = s;
    n.dt = fadd(n.dt, time);
    return false;
}

// integrate_for_px (video.rs:1318-1380) with pixel_tree_mode = Continuous.  `max_nodes` = nodes the arena may
// hold; returns false if it needed more.
// `kind` selects the parts of the step a source runs (AdderSparseStep::pad): 0 = all of integrate_for_px; the DAVIS
// source (davis.rs:331-395) integrates the OLD intensity over the gap first (kSparseIntegrateOnly) and then tests the
// NEW value against base_val without integrating it (kSparseTestOnly); at the end of its input every pixel is flushed
// (kSparseFlush: pop_best_events alone, davis.rs:654-661).
constexpr uint32_t kSparseNoSide = 1u, kSparseIntegrateOnly = 2u, kSparseTestOnly = 4u, kSparseFlush = 8u;
template <bool ABS_T, class Acc, class Emit>
ADDER_HD bool cont_step(APx &s, Acc &acc, uint32_t v, float intensity, float time, const StepConsts &sc,
                        uint32_t max_nodes, Emit &emit, uint32_t kind = 0u) {
    bool ok = true;
    const bool flush_only = (kind & kSparseFlush) != 0u;
    // ---- pop_best_events (:213-287) + set_d_for_continuous (:289-312) ----
    if (flush_only || (!(kind & kSparseIntegrateOnly) && contrast_exceeded(v, s.base, sc.cth))) {
        const bool collapsed = s.popped && sc.collapse;
        uint32_t count = 0, first_d = 0, first_t = 0;
        for (uint32_t k = 0; k < s.length; ++k) {
            ANode n = acc.load(k);
            uint32_t ed;
            float edt;
            if (n.has_best) {
                ed = n.bd;
                edt = n.bdt;
            } else if (n.dt > 0.0f && n.integ == 0.0f) {  // get_zero_event(idx, None) (:96-111)
                ed = kDZero;
                edt = n.dt;
                n.dt = 0.0f;
                acc.store(k, n);
            } else {
                continue;
            }
            const uint32_t t = cont_event_time<ABS_T>(edt, s.lastf);
            if (collapsed) {
                if (count == 0u) {
                    first_d = ed;
                    first_t = t;
                }
            } else {
                emit(ed, t);
            }
            count += 1u;
        }
        if (collapsed && count != 0u) {
            emit(first_d, first_t);
            s.lastf = sc.running_t;
            emit(kDEmpty, f32_as_u32(sc.running_t));
            acc.store(0, anode_new(intensity));
        } else {
            const ANode a = acc.load(0), z = acc.load(s.length - 1u);  // swap(arena[0], arena[length - 1])
            acc.store(0, z);
            acc.store(s.length - 1u, a);
        }
        s.length = 1u;
        s.popped = false;
        if (!flush_only) {
            s.base = v;
            ANode r = acc.load(0);
            const uint32_t next_d = get_d(intensity);
            if (next_d < r.d && r.dt > 0.0f) {
                emit(kDEmpty, cont_event_time<ABS_T>(r.dt, s.lastf));
                r.dt = 0.0f;
                r.integ = 0.0f;
            }
            r.d = next_d;
            acc.store(0, r);
        }
    }
    if (kind & (kSparseTestOnly | kSparseFlush)) return ok;
    // ---- integrate (:317-413) ----
    {
        ANode tail = acc.load(s.length - 1u);
        if (tail.dt == 0.0f && tail.integ == 0.0f) {
            tail.d = get_d(intensity);
            acc.store(s.length - 1u, tail);
        }
    }
    {
        float I = intensity, T = time;
        uint32_t idx = 0;
        for (uint32_t count = 0; count < 31u; ++count) {
            ANode n = acc.load(idx);
            float next_i = 0.0f, next_t = 0.0f;
            const bool filled = cont_integrate_main(n, I, T, next_i, next_t);
            acc.store(idx, n);
            if (filled) {
                if (idx + 1u >= max_nodes) {
                    ok = false;
                    break;
                }
                acc.store(idx + 1u, anode_new(I));
                s.length = idx + 2u;
                I = next_i;
                T = next_t;
            }
            idx += 1u;
            if (s.popped && sc.collapse) break;
            if (filled) {
                if (T > (float)sc.ref_time) {
                    ANode c = acc.load(idx);
                    c.d = get_d(I);
                    acc.store(idx, c);
                }
                if (I == 0.0f) break;
            }
            if (idx >= s.length) break;
        }
    }
    // ---- pop_top_event (:139-210) ----
    ANode root = acc.load(0);
    if (root.d == kDMax || (root.dt >= sc.dtm_f && !s.popped)) {
        bool shifted = true;
        uint32_t ed;
        float edt;
        if (!root.has_best) {
            if (root.integ == 0.0f && root.dt > 0.0f) {  // get_zero_event(0, Some(next_intensity))
                ed = kDZero;
                edt = root.dt;
                root.dt = 0.0f;
                root.d = get_d(intensity);
                acc.store(0, root);
                shifted = false;
            } else {  // synthesise the root's event, start a fresh node behind it
                root.has_best = true;
                root.bd = root.integ < 1.0f ? kDZero : get_d(root.integ);
                root.bdt = root.dt;
                if (max_nodes < 2u) ok = false;
                acc.store(1, anode_new(intensity));
                s.length = 2u;
                ed = root.bd;
                edt = root.bdt;
            }
        } else {
            ed = root.bd;
            edt = root.bdt;
        }
        if (shifted) {
            for (uint32_t i = 0; i + 1u < s.length; ++i) acc.store(i, acc.load(i + 1u));
            s.length -= 1u;
        }
        s.popped = true;
        emit(ed, cont_event_time<ABS_T>(edt, s.lastf));
    }
    return ok;
}

// event_to_intensity (framer/scale_intensity.rs:262-270): 2^d / t in f64 (t = 0 counts as 1; d beyond the table: 0)
ADDER_HD double event_intensity_f64(uint32_t d, uint32_t t) {
    if (d >= 129u) return 0.0;
    const double shift = d == 128u ? 0.0 : __builtin_bit_cast(double, (uint64_t)(d + 1023u) << 52);
    return t == 0u ? shift : shift / (double)t;
}
// Rust's `as u8` from a float: saturating, NaN -> 0
ADDER_HD uint32_t f64_as_u8(double val) {
    if (!(val > 0.0)) return 0u;
    if (val >= 255.0) return 255u;
    return (uint32_t)val;
}
ADDER_HD uint32_t f32_as_u8(float val) {
    if (!(val > 0.0f)) return 0u;
    if (val >= 255.0f) return 255u;
    return (uint32_t)val;
}

// u8::get_frame_value, Intensity view of a U8 source, for the running_intensities side plane (video.rs:713-730,
// framer/scale_intensity.rs:58-72): ((2^d / t) * ref_time) as u8 in f64.
ADDER_HD uint32_t frame_value_u8(uint32_t d, uint32_t t, double tpf) { return f64_as_u8(event_intensity_f64(d, t) * tpf); }

// ------------------------------------------------------------------------------------------
// Feature-driven rate control.  FAST 9_16 corner test on the running-intensities plane
// (utils/cv.rs:56-212, the OpenCV-style scan with its quick rejects): the reference's answer is exactly
// "some arc of >= 9 contiguous pixels on the radius-3 Bresenham circle is entirely brighter than
// centre + 30, or entirely darker than centre - 30" (its rejects and the k == 17 early exit never change
// that; tests/test_features.py checks the literal restatement in the oracle against both).  Here: two 16-bit
// ring masks and an AND of rotations.
// ------------------------------------------------------------------------------------------
constexpr int kFastThreshold = 30;  // cv.rs:21 INTENSITY_THRESHOLD
constexpr uint32_t kFastArc = 9;    // cv.rs:32 STREAK_SIZE
constexpr uint32_t kFastBorder = 3; // cv.rs:57 is_border(.., 3)

ADDER_HD bool fast_arc9(uint32_t m) {  // m: 16 ring flags; true if 9 circularly contiguous bits are set
    m |= m << 16;
    m &= m >> 1;   // runs of 2
    m &= m >> 2;   // runs of 4
    m &= m >> 4;   // runs of 8
    m &= m >> 1;   // runs of 9
    return (m & 0xffffu) != 0u;
}

// img = [h][w][channels] u8, channel 0 is the one looked at (cv.rs:67-69: coord.c is None or 0)
ADDER_HD bool fast9_is_feature(const uint8_t *img, uint32_t w, uint32_t h, uint32_t channels, uint32_t x, uint32_t y) {
    if (x < kFastBorder || x + kFastBorder >= w || y < kFastBorder || y + kFastBorder >= h) return false;
    // cv.rs:25-30 CIRCLE3 as (dx, dy), nibble-packed with a +3 bias
    constexpr uint64_t kDx = 0x2100012345666543ull, kDy = 0x6543210001234566ull;
    const int c = (int)img[((size_t)y * w + x) * channels];
    uint32_t bright = 0u, dark = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        const int dx = (int)((kDx >> (4 * k)) & 0xfu) - 3, dy = (int)((kDy >> (4 * k)) & 0xfu) - 3;
        const int p = (int)img[((size_t)((int)y + dy) * w + (size_t)((int)x + dx)) * channels];
        bright |= (p > c + kFastThreshold ? 1u : 0u) << k;
        dark |= (p < c - kFastThreshold ? 1u : 0u) << k;
    }
    return fast_arc9(bright) || fast_arc9(dark);
}

// handle_features' filter (video.rs:893-906) for event i of a frame whose events are ev[begin, end), in raster
// order: the reference walks each row chunk's events as CIRCULAR pairs (e1, e2) and looks at e1 when it is on
// channel 0 / None, is not a D_EMPTY filler and e1.coord != e2.coord -- i.e. e1 is the last event of its pixel's
// run, where the chunk's last event is paired with the chunk's first.
template <class Ev>
ADDER_HD bool feature_looked_at(const Ev *ev, uint64_t begin, uint64_t end, uint64_t i, uint32_t row_begin,
                                uint32_t chunk_rows) {
    const Ev e1 = ev[i];
    if (!((e1.c == 0xffu || e1.c == 0u) && e1.d != kDEmpty)) return false;
    const uint32_t chunk = (e1.y - row_begin) / chunk_rows;
    uint64_t nxt = i + 1;
    if (nxt >= end || (ev[nxt].y - row_begin) / chunk_rows != chunk) {
        // e1 closes its chunk: the window wraps to the chunk's first event
        const uint32_t cy0 = row_begin + chunk * chunk_rows;
        uint64_t lo = begin, hi = i;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (ev[mid].y < cy0) lo = mid + 1;
            else hi = mid;
        }
        nxt = lo;
    }
    const Ev e2 = ev[nxt];
    return !(e2.x == e1.x && e2.y == e1.y && e2.c == e1.c);
}

}  // namespace adder
