// tests/cpu_sim/sim.cpp -- TEST HARNESS, not product code.
//
// Compiles the device header adder-codec-rs_amd/csrc/adder_pixel.hpp with g++ and runs
// it serially over the same structure-of-arrays state layout the HIP kernel uses
// (header word, tail planes, level planes), so the compact state representation and
// the two-phase count/emit split can be diffed against the oracle on a machine
// without a GPU.  The product never builds or loads this file.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "adder_pixel.hpp"

using namespace adder;

struct SimEvent {
    uint16_t x, y;
    uint8_t c, d;
    uint16_t pad;
    uint32_t t;
};

struct Sim {
    uint32_t W, H, C, row_begin;
    size_t N;
    uint32_t max_depth;
    uint32_t ref_time, dtm;
    uint32_t c_max, velocity;
    uint8_t c_thresh, c_counter;  // ONE pair for the whole plane (adder_pixel.hpp header comment)
    uint32_t collapse, abs_t;
    float running_t;
    // level 0 = {hdr, integ, dt, bdt}; levels k >= 1 in the deep planes at index k - 1
    std::vector<uint32_t> hdr;
    std::vector<float> integ0, dt0, bdt0;
    std::vector<float> lastf;
    std::vector<float> lv_integ, lv_dt, lv_bdt;  // [level - 1][unit]
    std::vector<uint8_t> lv_bd;
    uint64_t plan_mismatch;
    uint32_t max_m;
    int use_fast;
    int generic_sticky;  // a generic batch has run since the last reset: lean batches are no longer allowed
    int continuous;      // Mode::Continuous: the general arena step (cont_step), its own state planes
    std::vector<uint32_t> c_hdr, c_meta;           // [unit], [node][unit]
    std::vector<float> c_integ, c_dt, c_bdt;       // [node][unit]
    uint64_t fast_steps, generic_steps, lean_steps, cb_steps, cb_quiet_steps, lean_quiet_steps;
    int cb_quiet_path;  // take the device's quiet-wave reduction wherever a unit qualifies (default on)
    int quiet_group_path;  // ... and its group form (quiet_group_apply / lr_quiet_run: 16 frames decided at once) where a launch allows
    uint64_t quiet_groups, quiet_group_fires, quiet_group_slow;  // groups applied in closed form / of those with a firing / stepped by the unit
    int use_cb;          // Collapse with delta_t_max > time: the bounded step (cb_step) instead of the generic one
    int frac_time_seen;  // a non-integer time_spanned has been integrated since the last reset (cb needs exact sums)
    // feature-driven rate control (the kernel-side flow of adder_feature_kernel, serially)
    int feat_detect, feat_adjust, roi_on, perpx;
    uint32_t baseline, radius, chunk_rows, roi[4];
    std::vector<uint8_t> cth_px, cctr_px, running, fset;
    uint32_t new_features;
    // sparse steps (event-camera sources): running_t per unit as well, from the first sparse call on
    int sparse;
    std::vector<float> rt_px;
};

struct DeepAcc {
    Sim *s;
    size_t u;
    void load(uint32_t k, Node &n) {
        size_t i = (size_t)(k - 1) * s->N + u;
        n.integ = s->lv_integ[i];
        n.dt = s->lv_dt[i];
        n.bdt = s->lv_bdt[i];
        n.bd = s->lv_bd[i];
    }
    void store(uint32_t k, const Node &n) {
        size_t i = (size_t)(k - 1) * s->N + u;
        s->lv_integ[i] = n.integ;
        s->lv_dt[i] = n.dt;
        s->lv_bdt[i] = n.bdt;
        s->lv_bd[i] = (uint8_t)n.bd;
    }
};

struct ContAcc {
    Sim *s;
    size_t u;
    ANode load(uint32_t k) const {
        const size_t i = (size_t)k * s->N + u;
        ANode n;
        n.integ = s->c_integ[i];
        n.dt = s->c_dt[i];
        n.bdt = s->c_bdt[i];
        anode_set_meta(n, s->c_meta[i]);
        return n;
    }
    void store(uint32_t k, const ANode &n) const {
        const size_t i = (size_t)k * s->N + u;
        s->c_integ[i] = n.integ;
        s->c_dt[i] = n.dt;
        s->c_bdt[i] = n.bdt;
        s->c_meta[i] = anode_meta(n);
    }
};

struct Emitter {
    SimEvent *out;
    size_t cap, pos;
    uint16_t x, y;
    uint8_t c;
    void operator()(uint32_t d, uint32_t t) {
        if (pos < cap) {
            SimEvent e;
            e.x = x; e.y = y; e.c = c; e.d = (uint8_t)d; e.pad = 0; e.t = t;
            out[pos] = e;
        }
        pos++;
    }
};

extern "C" {

Sim *sim_new(uint32_t W, uint32_t H, uint32_t C, uint32_t row_begin, int time_mode, int multi_mode,
             uint32_t ref_time, uint32_t dtm, uint32_t max_depth) {
    Sim *s = new Sim();
    s->W = W; s->H = H; s->C = C; s->row_begin = row_begin;
    s->N = (size_t)W * H * C;
    s->max_depth = max_depth;
    s->ref_time = ref_time; s->dtm = dtm;
    s->c_max = 7; s->velocity = 7;
    s->c_thresh = 10; s->c_counter = 1;  // PixelArena::new
    s->collapse = multi_mode == 1; s->abs_t = time_mode == 1;
    s->running_t = 0.0f;
    s->hdr.assign(s->N, hdr_make(0u, 0u, 0u, false));
    s->integ0.assign(s->N, 0.0f);
    s->dt0.assign(s->N, 0.0f);
    s->bdt0.assign(s->N, 0.0f);
    s->lastf.assign(s->N, 0.0f);
    const size_t deep = max_depth > 1 ? max_depth - 1 : 1;
    s->lv_integ.assign(s->N * deep, 0.0f);
    s->lv_dt.assign(s->N * deep, 0.0f);
    s->lv_bdt.assign(s->N * deep, 0.0f);
    s->lv_bd.assign(s->N * deep, 0);
    s->plan_mismatch = 0;
    s->max_m = 0;
    s->use_fast = 1;
    s->generic_sticky = 0;
    s->continuous = 0;
    s->fast_steps = s->generic_steps = s->lean_steps = s->cb_steps = s->cb_quiet_steps = s->lean_quiet_steps = 0;
    s->cb_quiet_path = 1;
    s->quiet_group_path = 1;
    s->quiet_groups = s->quiet_group_fires = s->quiet_group_slow = 0;
    s->use_cb = 1;
    s->frac_time_seen = 0;
    s->feat_detect = s->feat_adjust = s->roi_on = s->perpx = 0;
    s->baseline = 2; s->radius = 0; s->chunk_rows = 1;
    s->running.assign(s->N, 0);
    s->fset.assign((size_t)W * H, 0);
    s->new_features = 0;
    s->sparse = 0;
    return s;
}
void sim_free(Sim *s) { delete s; }
// Video::new(plane, Mode::Continuous, ..): every pixel = PixelArena::new(1.0): one node {d: 0}, length 1
void sim_set_continuous(Sim *s) {
    s->continuous = 1;
    const size_t nodes = s->max_depth + 1;
    s->c_hdr.assign(s->N, apx_hdr(APx{0u, 1u, false, 0.0f}));
    s->c_meta.assign(s->N * nodes, anode_meta(anode_new(1.0f)));
    s->c_integ.assign(s->N * nodes, 0.0f);
    s->c_dt.assign(s->N * nodes, 0.0f);
    s->c_bdt.assign(s->N * nodes, 0.0f);
}
// the device header's corner test, a whole plane at once: out[y][x] = fast9_is_feature
void sim_fast9_plane(const uint8_t *img, uint32_t w, uint32_t h, uint32_t channels, uint8_t *out) {
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) out[(size_t)y * w + x] = fast9_is_feature(img, w, h, channels, x, y) ? 1 : 0;
}
void sim_set_crf_parameters(Sim *s, uint8_t c_max, uint8_t velocity) { s->c_max = c_max; s->velocity = velocity; }
void sim_reset_c_thresh(Sim *s, uint8_t baseline) {
    s->c_thresh = baseline;
    s->c_counter = 0;
    s->perpx = 0;
}
void sim_update_detect_features(Sim *s, int detect, int adjust, uint32_t baseline, uint32_t radius, uint32_t chunk_rows) {
    s->feat_detect = detect; s->feat_adjust = adjust; s->baseline = baseline; s->radius = radius;
    s->chunk_rows = chunk_rows ? chunk_rows : 1;
}
void sim_update_roi(Sim *s, int on, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t baseline) {
    s->roi_on = on; s->roi[0] = x0; s->roi[1] = y0; s->roi[2] = x1; s->roi[3] = y1; s->baseline = baseline;
}
void sim_feature_set(const Sim *s, uint8_t *out) { memcpy(out, s->fset.data(), s->fset.size()); }
void sim_c_thresh_plane(const Sim *s, uint8_t *out) {
    if (s->perpx) memcpy(out, s->cth_px.data(), s->N);
    else memset(out, s->c_thresh, s->N);
}
uint32_t sim_new_features(const Sim *s) { return s->new_features; }
void sim_set_delta_t_max(Sim *s, uint32_t dtm) { s->dtm = dtm; }
uint64_t sim_plan_mismatches(const Sim *s) { return s->plan_mismatch; }
uint32_t sim_max_m(const Sim *s) { return s->max_m; }
void sim_set_use_fast(Sim *s, int on) { s->use_fast = on; }
uint64_t sim_fast_steps(const Sim *s) { return s->fast_steps; }
uint64_t sim_generic_steps(const Sim *s) { return s->generic_steps; }
uint64_t sim_lean_steps(const Sim *s) { return s->lean_steps; }
uint64_t sim_cb_steps(const Sim *s) { return s->cb_steps; }
uint64_t sim_cb_quiet_steps(const Sim *s) { return s->cb_quiet_steps; }
uint64_t sim_lean_quiet_steps(const Sim *s) { return s->lean_quiet_steps; }
void sim_set_cb_quiet_path(Sim *s, int on) { s->cb_quiet_path = on; }
void sim_set_quiet_group_path(Sim *s, int on) { s->quiet_group_path = on; }
uint64_t sim_quiet_groups(const Sim *s) { return s->quiet_groups; }
uint64_t sim_quiet_group_fires(const Sim *s) { return s->quiet_group_fires; }
uint64_t sim_quiet_group_slow(const Sim *s) { return s->quiet_group_slow; }
void sim_set_use_cb(Sim *s, int on) { s->use_cb = on; }

// integrate_for_px(px, &mut 0, frame_val, intensity, time) per step, in order (the flow of adder_sparse_run_kernel,
// serially): Continuous contexts; c_thresh, its counter and running_t are per unit from the first call on
struct SimSparseStep {
    uint16_t x, y;
    uint8_t c, frame_val;
    uint16_t pad;
    float intensity, time;
};
int sim_integrate_sparse(Sim *s, const SimSparseStep *steps, size_t n, SimEvent *out, size_t cap, size_t *n_out) {
    if (!s->continuous) return -1;
    if (!s->sparse) {
        s->cth_px.assign(s->N, s->c_thresh);
        s->cctr_px.assign(s->N, s->c_counter);
        s->rt_px.assign(s->N, s->running_t);
        s->sparse = 1;
    }
    StepConsts sc;
    sc.dtm_f = (float)s->dtm;
    sc.ref_time = s->ref_time;
    sc.collapse = s->collapse;
    sc.abs_t = s->abs_t;
    sc.max_depth = s->max_depth;
    sc.ref_magic = s->ref_time >= 2 ? (uint32_t)(0x100000000ull / s->ref_time) : 0u;
    int rc = 0;
    Emitter em;
    em.out = out; em.cap = cap; em.pos = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t c = steps[i].c == 0xFF ? 0u : steps[i].c;
        if (steps[i].x >= s->W || steps[i].y < s->row_begin || steps[i].y - s->row_begin >= s->H || c >= s->C) return -1;
        const size_t u = ((size_t)(steps[i].y - s->row_begin) * s->W + steps[i].x) * s->C + c;
        em.x = steps[i].x; em.y = steps[i].y; em.c = steps[i].c;
        APx p = apx_unpack(s->c_hdr[u], s->lastf[u]);
        sc.time_spanned = steps[i].time;
        sc.running_t = s->rt_px[u];
        sc.running_t_u32 = f32_as_u32(sc.running_t);
        sc.cth = s->cth_px[u];
        ContAcc acc{s, u};
        const uint32_t kind = steps[i].pad;
        const bool ok = s->abs_t ? cont_step<true>(p, acc, steps[i].frame_val, steps[i].intensity, steps[i].time, sc, s->max_depth + 1, em, kind)
                                 : cont_step<false>(p, acc, steps[i].frame_val, steps[i].intensity, steps[i].time, sc, s->max_depth + 1, em, kind);
        if (!ok) rc = -5;
        s->c_hdr[u] = apx_hdr(p);
        s->lastf[u] = p.lastf;
        if (!(kind & (kSparseTestOnly | kSparseFlush))) {  // (running_t and c_thresh advance inside PixelArena::integrate)
            s->rt_px[u] += steps[i].time;
            c_thresh_advance(s->cth_px[u], s->cctr_px[u], (uint8_t)s->c_max, (uint8_t)s->velocity, steps[i].time, s->ref_time);
        }
    }
    *n_out = em.pos;
    if (em.pos > cap && rc == 0) rc = -4;
    return rc;
}

// the product's choice (adder_hip_api.cpp cb_possible): Collapse, delta_t_max > time_spanned, uniform c_thresh, and
// every sum the prefix coordinates form an exact integer below 2^24
static bool sim_cb_possible(const Sim *s, float T) {
    if (!s->use_cb || !s->collapse || s->perpx || s->continuous || s->frac_time_seen) return false;
    if (!((float)s->dtm > T)) return false;
    if (!(T >= 1.0f) || T != (float)(uint32_t)T || T > 65536.0f) return false;
    if ((double)s->dtm + 2.0 * T >= 8388608.0) return false;
    if (((double)s->dtm / T + 3.0) * 255.0 >= 8388608.0) return false;
    return true;
}

// One temporally blocked launch of the bounded Collapse step, as adder_cb_kernel runs it: per unit the levels are
// brought into prefix coordinates once, nb frames are stepped, and the state goes back in its resident form.
// frames = [nb][N]; the events come out frame-major like the product's stream.  returns 0 ok, -4 capacity, -5 depth
int sim_integrate_cb_block(Sim *s, const uint8_t *frames, uint32_t nb, float T, SimEvent *out, size_t cap, size_t *n_out) {
    if (!sim_cb_possible(s, T)) return -7;
    s->generic_sticky = 1;
    StepConsts sc;
    sc.time_spanned = T;
    sc.dtm_f = (float)s->dtm;
    sc.ref_time = s->ref_time;
    sc.collapse = 1;
    sc.abs_t = s->abs_t;
    sc.max_depth = s->max_depth;
    sc.ref_magic = s->ref_time >= 2 ? (uint32_t)(0x100000000ull / s->ref_time) : 0u;
    std::vector<float> rts(nb);
    std::vector<uint8_t> cths(nb);
    {
        float rt = s->running_t;
        uint8_t cth = s->c_thresh, cctr = s->c_counter;
        for (uint32_t i = 0; i < nb; ++i) {
            rts[i] = rt;
            cths[i] = cth;
            rt += T;
            c_thresh_advance(cth, cctr, (uint8_t)s->c_max, (uint8_t)s->velocity, T, s->ref_time);
        }
        s->running_t = rt;
        s->c_thresh = cth;
        s->c_counter = cctr;
    }
    std::vector<std::vector<SimEvent>> per_frame(nb);
    int rc = 0;
    size_t u = 0;
    for (uint32_t y = 0; y < s->H; y++)
        for (uint32_t x = 0; x < s->W; x++)
            for (uint32_t c = 0; c < s->C; c++, u++) {
                const uint32_t hdr = s->hdr[u];
                const uint32_t m0 = hdr_m(hdr);
                CbPx p = cb_unpack<ScalarLanes>(hdr, m0 ? s->integ0[u] : -12345.0f, m0 ? s->dt0[u] : -777.0f, m0 ? s->bdt0[u] : -999.0f,
                                   s->abs_t ? s->lastf[u] : -1.0f);
                if (p.popped && p.m > 1u) p.m = 1u;  // a popped arena keeps only its root
                // garbage on purpose: slots of levels >= m must never decide anything
                float F4[4] = {-3.0f, 1e30f, 0.0f, -1e30f}, Q4[4] = {7.0f, 7.0f, 7.0f, 7.0f};
                float BT[8] = {9.0f, 3.0f, 9.0f, -1.0f, 9.0f, 1e20f, 9.0f, 0.0f};
                CbLevels lv{F4, Q4, BT, s->lv_integ.data(), s->lv_dt.data(), s->lv_bdt.data(), s->lv_bd.data(), s->N, u};
                DeepAcc deep{s, u};
                for (uint32_t k = 1; k < p.m; ++k) {
                    Node n;
                    deep.load(k, n);
                    lv.store(k, cb_level_from_node(p, n));
                }
                for (uint32_t i = 0; i < nb; ++i) {
                    sc.running_t = rts[i];
                    sc.running_t_u32 = f32_as_u32(rts[i]);
                    sc.cth = cths[i];
                    struct VecEmit {
                        std::vector<SimEvent> *v;
                        uint16_t x, y;
                        uint8_t c;
                        uint32_t n;
                        void put(uint32_t d, uint32_t t) {
                            SimEvent e;
                            e.x = x; e.y = y; e.c = c; e.d = (uint8_t)d; e.pad = 0; e.t = t;
                            v->push_back(e);
                            ++n;
                        }
                        // the record's 9-bit code = the threshold's exponent field, decoded as the expansion kernel does
                        void ev(uint32_t thr_bits, uint32_t t) { put(cb_d_from_code(thr_bits >> 23), t); }
                        void filler(uint32_t t) { put(cb_d_from_code(kCbCodeEmpty), t); }
                    } em{&per_frame[i], (uint16_t)x, (uint16_t)(y + s->row_begin), s->C == 1 ? (uint8_t)0xFF : (uint8_t)c, 0u};
                    CbPlan plan;
                    const uint32_t vq = frames[(size_t)i * s->N + u];
                    if (s->cb_quiet_path && s->quiet_group_path && (i % kQuietGroup) == 0u && p.m == 1u && (p.popped || p.thr0 == 0.0f)) {
                        // the device's group form (adder_cb_kernel at the start of an input group): the whole group at once
                        const uint32_t n = nb - i < kQuietGroup ? nb - i : kQuietGroup;
                        uint32_t cth_min = 255u;
                        for (uint32_t k = 0; k < n; ++k) cth_min = cths[i + k] < cth_min ? cths[i + k] : cth_min;
                        const QuietGroupStats g = quiet_group_stats(frames + (size_t)i * s->N + u, s->N, n, quiet_group_need(p.S, p.thr0));
                        CbPx t = p;
                        const float thr_was = p.thr0;
                        const uint32_t r = cb_group_apply(t, g, n, cth_min, T);
                        if (r != kQuietNo) {
                            if (r == kQuietDone) {
                                p = t;
                                s->quiet_groups++;
                                if (p.thr0 != thr_was) s->quiet_group_fires++;
                            } else {
                                for (uint32_t k = 0; k < n; ++k) cb_step_quiet<ScalarLanes, true>(p, frames[(size_t)(i + k) * s->N + u], T);
                                s->quiet_group_slow++;
                            }
                            s->cb_quiet_steps += n;
                            s->cb_steps += n;
                            i += n - 1u;
                            continue;
                        }
                    }
                    if (s->cb_quiet_path && cb_quiet(p, vq, sc.cth)) {
                        // the device's fast path (a wave all of whose units are quiet): must equal the step, no events
                        if (cb_quiet_fires(p, vq) || (s->cb_steps & 1u)) cb_step_quiet<ScalarLanes, true>(p, vq, T);
                        else cb_step_quiet<ScalarLanes, false>(p, vq, T);
                        s->cb_quiet_steps++;
                        s->cb_steps++;
                        continue;
                    }
                    cb_step(p, lv, vq, T, sc, plan);
                    if (plan.depth_error) rc = -5;
                    if (s->abs_t) cb_emit<true>(p, plan, sc, lv, em); else cb_emit<false>(p, plan, sc, lv, em);
                    if (em.n != plan.count) s->plan_mismatch++;
                    cb_pop(p, plan, lv);
                    if (p.m > s->max_m) s->max_m = p.m;
                    s->cb_steps++;
                }
                for (uint32_t k = 1; k < p.m; ++k) deep.store(k, cb_node_from_level(p, lv.load(k)));
                s->hdr[u] = cb_hdr(p);
                if (p.m > 0) { s->integ0[u] = p.S; s->dt0[u] = p.dt0; s->bdt0[u] = p.bdt0; }
                if (s->abs_t) s->lastf[u] = p.lastf;
                if (p.m != 0u)
                    s->running[u] = (uint8_t)frame_value_u8(lean_bd_from_thr(f32_to_bits(p.thr0)), f32_as_u32(p.bdt0), (double)s->ref_time);
            }
    size_t pos = 0;
    for (uint32_t i = 0; i < nb; ++i)
        for (const SimEvent &e : per_frame[i]) {
            if (pos < cap) out[pos] = e;
            ++pos;
        }
    *n_out = pos;
    if (pos > cap && rc == 0) rc = -4;
    return rc;
}

// One temporally blocked launch of the CONSTANT-RUN step (cr_step / cr_emit / cr_pop / cr_materialize), as
// adder_cr_kernel runs it: only the roots are loaded and stepped, the levels are written back in their resident form at
// the end.  The caller vouches for the regime (c_thresh 0 in every frame since the reset, one integer time_spanned):
// -7 if this launch is not in it.
int sim_integrate_cr_block(Sim *s, const uint8_t *frames, uint32_t nb, float T, SimEvent *out, size_t cap, size_t *n_out) {
    if (!sim_cb_possible(s, T) || s->c_thresh != 0 || s->c_max != 0) return -7;
    s->generic_sticky = 1;
    StepConsts sc;
    sc.time_spanned = T;
    sc.dtm_f = (float)s->dtm;
    sc.ref_time = s->ref_time;
    sc.collapse = 1;
    sc.abs_t = s->abs_t;
    sc.max_depth = s->max_depth;
    sc.cth = 0;
    sc.ref_magic = s->ref_time >= 2 ? (uint32_t)(0x100000000ull / s->ref_time) : 0u;
    std::vector<float> rts(nb);
    {
        float rt = s->running_t;
        for (uint32_t i = 0; i < nb; ++i) {
            rts[i] = rt;
            rt += T;
        }
        s->running_t = rt;
    }
    std::vector<std::vector<SimEvent>> per_frame(nb);
    int rc = 0;
    size_t u = 0;
    for (uint32_t y = 0; y < s->H; y++)
        for (uint32_t x = 0; x < s->W; x++)
            for (uint32_t c = 0; c < s->C; c++, u++) {
                const uint32_t hdr = s->hdr[u];
                const uint32_t m0 = hdr_m(hdr);
                CrPx p = cr_unpack<ScalarLanes>(hdr, m0 ? s->integ0[u] : -12345.0f, m0 ? s->dt0[u] : -777.0f,
                                                m0 ? s->bdt0[u] : -999.0f, s->abs_t ? s->lastf[u] : -1.0f, T);
                for (uint32_t i = 0; i < nb; ++i) {
                    sc.running_t = rts[i];
                    sc.running_t_u32 = f32_as_u32(rts[i]);
                    struct VecEmit {
                        std::vector<SimEvent> *v;
                        uint16_t x, y;
                        uint8_t c;
                        uint32_t n;
                        void put(uint32_t d, uint32_t t) {
                            SimEvent e;
                            e.x = x; e.y = y; e.c = c; e.d = (uint8_t)d; e.pad = 0; e.t = t;
                            v->push_back(e);
                            ++n;
                        }
                        void ev(uint32_t thr_bits, uint32_t t) { put(cb_d_from_code(thr_bits >> 23), t); }
                        void filler(uint32_t t) { put(cb_d_from_code(kCbCodeEmpty), t); }
                    } em{&per_frame[i], (uint16_t)x, (uint16_t)(y + s->row_begin), s->C == 1 ? (uint8_t)0xFF : (uint8_t)c, 0u};
                    CrPlan plan;
                    cr_step(p, frames[(size_t)i * s->N + u], T, sc, plan);
                    if (s->abs_t) cr_emit<true>(p, plan, T, sc, em); else cr_emit<false>(p, plan, T, sc, em);
                    if (em.n != plan.count) s->plan_mismatch++;
                    cr_pop(p, plan, T);
                    s->cb_steps++;
                }
                DeepAcc deep{s, u};
                struct Store {
                    DeepAcc *d;
                    uint32_t max_depth;
                    int *rc;
                    void operator()(uint32_t k, const Node &n) {
                        if (k < max_depth) d->store(k, n);
                        else *rc = -5;
                    }
                } st{&deep, s->max_depth, &rc};
                const uint32_t m = cr_materialize(p, T, st);
                if (m > s->max_m) s->max_m = m;
                s->hdr[u] = cr_hdr(p, T);
                if (m > 0) { s->integ0[u] = p.S; s->dt0[u] = p.dt0; s->bdt0[u] = p.bdt0; }
                if (s->abs_t) s->lastf[u] = p.lastf;
                if (m != 0u)
                    s->running[u] = (uint8_t)frame_value_u8(lean_bd_from_thr(f32_to_bits(p.thr0)), f32_as_u32(p.bdt0), (double)s->ref_time);
            }
    size_t pos = 0;
    for (uint32_t i = 0; i < nb; ++i)
        for (const SimEvent &e : per_frame[i]) {
            if (pos < cap) out[pos] = e;
            ++pos;
        }
    *n_out = pos;
    if (pos > cap && rc == 0) rc = -4;
    return rc;
}

// One temporally blocked launch of the RUN-RECORDS step (rr_step / rr_event / rr_pack): integer state, one record per
// happening, the events worked out from the records as the expansion does.  -7 outside its regime (the constant-run
// conditions; AbsoluteT also wants time_spanned == ref_time >= 255).
int sim_integrate_rr_block(Sim *s, const uint8_t *frames, uint32_t nb, float T, SimEvent *out, size_t cap, size_t *n_out) {
    {   // the bounded Collapse step's conditions without the mode: Normal runs it too
        const bool was = s->collapse;
        const uint32_t dtm_was = s->dtm;
        s->collapse = true;
        if (!was && (float)s->dtm <= T) s->dtm = (uint32_t)T + 1u;  // (Normal also with delta_t_max <= time_spanned: kRrFlushPop)
        const bool ok = sim_cb_possible(s, T);
        s->collapse = was;
        s->dtm = dtm_was;
        if (!ok || s->c_thresh != 0 || s->c_max != 0) return -7;
    }
    if (s->abs_t && (T != (float)s->ref_time || s->ref_time < 255u)) return -7;
    s->generic_sticky = 1;
    const uint32_t Tu = (uint32_t)T;
    const uint32_t n_pop = (s->dtm + Tu - 1u) / Tu;
    std::vector<uint8_t> tab(256 * kRrTabRows);
    rr_build_tab(tab.data(), T);
    auto levels = [&](uint32_t I, uint32_t r) -> uint32_t { return tab[I * kRrTabRows + r]; };
    const uint32_t frame0 = (uint32_t)(s->running_t / T);
    std::vector<std::vector<SimEvent>> per_frame(nb);
    int rc = 0;
    size_t u = 0;
    for (uint32_t y = 0; y < s->H; y++)
        for (uint32_t x = 0; x < s->W; x++)
            for (uint32_t c = 0; c < s->C; c++, u++) {
                const uint32_t hdr = s->hdr[u];
                const uint32_t m0 = hdr_m(hdr);
                RrPx p = rr_unpack<ScalarLanes>(hdr, m0 ? s->dt0[u] : -777.0f, s->abs_t ? s->lastf[u] : -1.0f, T, s->abs_t != 0);
                for (uint32_t i = 0; i < nb; ++i) {
                    uint32_t w0, w1, w2, count;
                    if (s->abs_t) rr_step<true>(p, frames[(size_t)i * s->N + u], frame0 + i, n_pop, T, levels, 0u, w0, w1, w2, count, s->collapse);
                    else rr_step<false>(p, frames[(size_t)i * s->N + u], frame0 + i, n_pop, T, levels, 0u, w0, w1, w2, count, s->collapse);
                    s->cb_steps++;
                    if (count == 0u) continue;
                    // the expansion's side: everything from the three words
                    const uint32_t kind = w2 & 3u, Iu = (w2 >> kRrBaseShift) & 0xffu, cnt = w2 >> kRrCountShift;
                    if (cnt != count) s->plan_mismatch++;
                    const uint32_t rt_u32 = f32_as_u32(fmul((float)(frame0 + i), T));
                    for (uint32_t k = 0; k < cnt; ++k) {
                        const RrEvent e = s->abs_t ? rr_event_at<true>(kind, Iu, w0, w1, k, cnt, T, rt_u32) : rr_event_at<false>(kind, Iu, w0, w1, k, cnt, T, rt_u32);
                        SimEvent ev;
                        ev.x = (uint16_t)x; ev.y = (uint16_t)(y + s->row_begin); ev.c = s->C == 1 ? (uint8_t)0xFF : (uint8_t)c;
                        ev.d = (uint8_t)e.d; ev.pad = 0; ev.t = e.t;
                        per_frame[i].push_back(ev);
                    }
                    if (kind == kRrFlush && Iu != 0u && rr_chain(Iu, w0, T) != cnt) s->plan_mismatch++;  // the chain ends exactly at the count
                    if (kind == kRrFlushPop && Iu != 0u && rr_chain(Iu, w0 & kRrRunMask, T) + 1u != cnt) s->plan_mismatch++;
                }
                DeepAcc deep{s, u};
                struct Store {
                    DeepAcc *d;
                    uint32_t max_depth;
                    int *rc;
                    void operator()(uint32_t k, const Node &n) {
                        if (k < max_depth) d->store(k, n);
                        else *rc = -5;
                    }
                } st{&deep, s->max_depth, &rc};
                float integ, dt, bdt, lastf;
                const uint32_t h2 = rr_pack(p, T, integ, dt, bdt, lastf, st, s->collapse);
                const uint32_t m = hdr_m(h2);
                if (m > s->max_m) s->max_m = m;
                s->hdr[u] = h2;
                if (m > 0) { s->integ0[u] = integ; s->dt0[u] = dt; s->bdt0[u] = bdt; }
                if (s->abs_t) s->lastf[u] = lastf;
                if (m != 0u) s->running[u] = (uint8_t)frame_value_u8((h2 >> kHdrBdShift) & 0xffu, f32_as_u32(bdt), (double)s->ref_time);
            }
    s->running_t = fmul((float)(frame0 + nb), T);
    size_t pos = 0;
    for (uint32_t i = 0; i < nb; ++i)
        for (const SimEvent &e : per_frame[i]) {
            if (pos < cap) out[pos] = e;
            ++pos;
        }
    *n_out = pos;
    if (pos > cap && rc == 0) rc = -4;
    return rc;
}

// One temporally blocked launch of the LEAN-RUNS step (lr_step / lr_decode8 / lr_pack), as adder_lean_kernel's RUNS
// instantiation and the expansion run it: a unit is {base_val, rho, popped}, event A is worked out from the record.
// -7 outside its regime (Collapse, delta_t_max <= T, DeltaT, c_thresh 0 with c_thresh_max 0, integer T, no generic batch).
int sim_integrate_lr_block(Sim *s, const uint8_t *frames, uint32_t nb, float T, SimEvent *out, size_t cap, size_t *n_out) {
    if (!s->collapse || !((float)s->dtm <= T) || s->generic_sticky || s->perpx || s->continuous ||
        s->c_thresh != 0 || s->c_max != 0 || !(T >= 1.0f) || T != (float)(uint32_t)T || s->frac_time_seen)
        return -7;
    if (s->abs_t && (T != (float)s->ref_time || s->ref_time < 255u)) return -7;  // (last_fired_t on multiples of T)
    const uint32_t frame0 = (uint32_t)(s->running_t / T);
    std::vector<float> rts(nb);
    {
        float rt = s->running_t;
        for (uint32_t i = 0; i < nb; ++i) {
            rts[i] = rt;
            rt += T;
        }
        s->running_t = rt;
    }
    std::vector<std::vector<SimEvent>> per_frame(nb);
    std::vector<uint32_t> lr_tab(kLrTabWords);
    lr_build_tab(lr_tab.data(), T);
    int rc = 0;
    size_t u = 0;
    for (uint32_t y = 0; y < s->H; y++)
        for (uint32_t x = 0; x < s->W; x++)
            for (uint32_t c = 0; c < s->C; c++, u++) {
                const uint32_t m0 = hdr_m(s->hdr[u]);
                bool consistent;
                LrPx p = lr_unpack<ScalarLanes>(s->hdr[u], m0 ? s->dt0[u] : -777.0f, T, consistent);
                if (!consistent) rc = -10;  // popped_dtm != (base_val != 0): the planes are not the lean regime's
                bool nz_old = p.base != 0u;
                uint32_t lq = s->abs_t ? (uint32_t)(s->lastf[u] / T) : 0u;
                for (uint32_t i = 0; i < nb; ++i) {
                    uint32_t w0, w8, w1 = 0u;
                    const uint32_t vin = frames[(size_t)i * s->N + u];
                    if (s->quiet_group_path && (i % kQuietGroup) == 0u) {
                        // the device's group form (adder_lr_kernel at the start of an input group): every byte of the group
                        // equals base_val -- no flush, no event, the run grows by the group
                        const uint32_t n = nb - i < kQuietGroup ? nb - i : kQuietGroup;
                        bool same = true;
                        for (uint32_t k = 0; k < n; ++k) same = same && frames[(size_t)(i + k) * s->N + u] == p.base;
                        if (same) {
                            lr_quiet_run(p, n);
                            s->quiet_groups++;
                            i += n - 1u;
                            continue;
                        }
                    }
                    const LeanFlagsT<ScalarLanes> fl = lr_step<ScalarLanes>(p, vin, (p.base << kLrBaseShift) | (vin << kLrInShift), (uint32_t)(u & 127u), nz_old, w0, w8);
                    if (s->abs_t) lr_step_lq<ScalarLanes>(lq, fl, frame0 + i, w1);
                    if (fl.b && !fl.a) rc = -9;
                    if (!(fl.a || fl.c)) continue;  // (no record)
                    if ((w8 & 127u) != (u & 127u)) rc = -9;
                    const LeanEvents e = s->abs_t ? lr_decode12(w0, w1, w8, T, f32_as_u32(rts[i]), frame0 + i)
                                                  : lr_decode8_tab(w0, w8, T, f32_as_u32(rts[i]), (u & 1u) ? lr_tab.data() : nullptr, lr_tab.data() + 256u * kLrTabRuns);  // (as the expansion does; odd units through the A rows too)
                    if (!s->abs_t) {
                        const LeanEvents e2 = lr_decode8(w0, w8, T, f32_as_u32(rts[i]));
                        if ((e.a && (e.da != e2.da || e.ta != e2.ta)) || (e.c && (e.dc != e2.dc || e.tc != e2.tc))) rc = -11;
                    }
                    if (e.a != fl.a || e.b != fl.b || e.c != fl.c) rc = -9;
                    SimEvent ev;
                    ev.x = (uint16_t)x; ev.y = (uint16_t)(y + s->row_begin); ev.c = s->C == 1 ? (uint8_t)0xFF : (uint8_t)c; ev.pad = 0;
                    if (e.a) { ev.d = (uint8_t)e.da; ev.t = e.ta; per_frame[i].push_back(ev); }
                    if (e.b) { ev.d = (uint8_t)kDEmpty; ev.t = e.tb; per_frame[i].push_back(ev); }
                    if (e.c) { ev.d = (uint8_t)e.dc; ev.t = e.tc; per_frame[i].push_back(ev); }
                }
                float integ, dt, bdt;
                s->hdr[u] = lr_pack(p, T, integ, dt, bdt);
                if (s->abs_t) s->lastf[u] = fmul((float)lq, T);
                if (p.rho != 0u) {
                    s->integ0[u] = integ; s->dt0[u] = dt; s->bdt0[u] = bdt;
                    if (s->max_m < 1) s->max_m = 1;
                    s->running[u] = (uint8_t)frame_value_u8((s->hdr[u] >> kHdrBdShift) & 0xffu, f32_as_u32(bdt), (double)s->ref_time);
                }
                s->lean_steps += nb;
            }
    size_t pos = 0;
    for (uint32_t i = 0; i < nb; ++i)
        for (const SimEvent &e : per_frame[i]) {
            if (pos < cap) out[pos] = e;
            ++pos;
        }
    *n_out = pos;
    if (pos > cap && rc == 0) rc = -4;
    return rc;
}

// One temporally blocked launch of the PACKED lean-runs step (lp_init / lp_step / lp_record / lp_final, decoded through
// lp_rho + lr_decode8), as adder_lp_kernel and the expansion's format 7 run it: four units per word, booleans in bit 7 of the
// unit's byte, rho as the distance to the launch-relative frame of the last flush.  DeltaT only; -7 outside the regime.
// Every frame's masks are also checked against lr_step on the same units (rc -12), the event counts against the masks' bits.
int sim_integrate_lp_block(Sim *s, const uint8_t *frames, uint32_t nb, float T, SimEvent *out, size_t cap, size_t *n_out) {
    if (!s->collapse || !((float)s->dtm <= T) || s->generic_sticky || s->perpx || s->continuous || s->abs_t ||
        s->c_thresh != 0 || s->c_max != 0 || !(T >= 1.0f) || T != (float)(uint32_t)T || s->frac_time_seen)
        return -7;
    std::vector<float> rts(nb);
    {
        float rt = s->running_t;
        for (uint32_t i = 0; i < nb; ++i) {
            rts[i] = rt;
            rt += T;
        }
        s->running_t = rt;
    }
    std::vector<std::vector<SimEvent>> per_frame(nb);
    std::vector<uint32_t> lr_tab(kLrTabWords);
    lr_build_tab(lr_tab.data(), T);
    int rc = 0;
    const size_t N = s->N;
    const uint32_t rowlen = s->W * s->C;
    for (size_t u0 = 0; u0 < N; u0 += 4) {
        LrPx p[4], ref[4];
        bool nz_old[4];
        for (uint32_t j = 0; j < 4u; ++j) {
            const size_t u = u0 + j;
            p[j].base = 0u;
            p[j].rho = 0u;  // (padding units: pristine, fed zeros)
            if (u < N) {
                bool consistent;
                p[j] = lr_unpack<ScalarLanes>(s->hdr[u], hdr_m(s->hdr[u]) ? s->dt0[u] : -777.0f, T, consistent);
                if (!consistent) rc = -10;
            }
            ref[j] = p[j];
            nz_old[j] = p[j].base != 0u;
        }
        LpWord w;
        lp_init(w, p);
        for (uint32_t i = 0; i < nb; ++i) {
            uint32_t vin = 0u;
            for (uint32_t j = 0; j < 4u; ++j)
                if (u0 + j < N) vin |= (uint32_t)frames[(size_t)i * N + u0 + j] << (8u * j);
            const uint32_t base_w = w.prev;
            if (vin == base_w && (i & 1u)) {  // (the kernel's quiet frames and groups; odd frames here, so that both forms run)
                lp_quiet(w);
                for (uint32_t j = 0; j < 4u; ++j) {
                    uint32_t w0, w8;
                    const LeanFlagsT<ScalarLanes> fl = lr_step<ScalarLanes>(ref[j], (vin >> (8u * j)) & 0xffu, 0u, 0u, nz_old[j], w0, w8);
                    if (fl.a || fl.b || fl.c) rc = -12;
                }
                continue;
            }
            const LpMasks m = lp_step(w, vin);
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t bit = 0x80u << (8u * j);
                uint32_t r0, r8;
                const LeanFlagsT<ScalarLanes> fl = lr_step<ScalarLanes>(ref[j], (vin >> (8u * j)) & 0xffu, 0u, 0u, nz_old[j], r0, r8);
                if (fl.a != ((m.a & bit) != 0u) || fl.b != ((m.b & bit) != 0u) || fl.c != ((m.c & bit) != 0u)) rc = -12;
                if (((m.h & bit) != 0u) != (fl.a || fl.c || ((vin ^ base_w) >> (8u * j) & 0xffu) != 0u)) rc = -12;
                if (!(m.h & bit)) continue;
                const size_t u = u0 + j;
                uint32_t w0, w8;
                lp_record(base_w, vin, w.start[j], j, i, (uint32_t)(u & 255u), w0, w8);
                w.start[j] = i;
                if (u >= N) { rc = -12; continue; }  // (a padding unit never flushes)
                {   // through the parked 4-byte form and its escape word
                    const uint32_t w4 = lp_park4(w0, w8);
                    uint32_t b0, b8;
                    lp_unpark4(w4, w0, b0, b8);
                    if (b0 != w0 || b8 != w8 || lp_escapes(w4) != (w0 >= kLpRhoEsc)) rc = -14;
                }
                const uint32_t rho = lp_rho(w0, w8);
                const LeanEvents e = lr_decode8_tab(rho, w8, T, f32_as_u32(rts[i]), (u & 1u) ? lr_tab.data() : nullptr, lr_tab.data() + 256u * kLrTabRuns);
                if (e.a != fl.a || e.b != fl.b || e.c != fl.c) rc = -9;
                if (fl.a && r0 != rho && ((w8 >> kLrBaseShift) & 0xffu) != 0u) rc = -13;  // (the run lr_step counted)
                SimEvent ev;
                ev.x = (uint16_t)((u % rowlen) / s->C); ev.y = (uint16_t)(u / rowlen + s->row_begin);
                ev.c = s->C == 1 ? (uint8_t)0xFF : (uint8_t)(u % s->C); ev.pad = 0;
                if (e.a) { ev.d = (uint8_t)e.da; ev.t = e.ta; per_frame[i].push_back(ev); }
                if (e.b) { ev.d = (uint8_t)kDEmpty; ev.t = e.tb; per_frame[i].push_back(ev); }
                if (e.c) { ev.d = (uint8_t)e.dc; ev.t = e.tc; per_frame[i].push_back(ev); }
            }
        }
        for (uint32_t j = 0; j < 4u; ++j) {
            const size_t u = u0 + j;
            if (u >= N) continue;
            const LrPx q = lp_final(w, j, nb);
            if (q.base != ref[j].base || q.rho != ref[j].rho) rc = -13;
            float integ, dt, bdt;
            s->hdr[u] = lr_pack(q, T, integ, dt, bdt);
            if (q.rho != 0u) {
                s->integ0[u] = integ; s->dt0[u] = dt; s->bdt0[u] = bdt;
                if (s->max_m < 1) s->max_m = 1;
                s->running[u] = (uint8_t)frame_value_u8((s->hdr[u] >> kHdrBdShift) & 0xffu, f32_as_u32(bdt), (double)s->ref_time);
            }
            s->lean_steps += nb;
        }
    }
    size_t pos = 0;
    for (uint32_t i = 0; i < nb; ++i)
        for (const SimEvent &e : per_frame[i]) {
            if (pos < cap) out[pos] = e;
            ++pos;
        }
    *n_out = pos;
    if (pos > cap && rc == 0) rc = -4;
    return rc;
}

// The lean kernel's quiet-GROUP form against its own stepped form (lean_group_apply vs n x lean_step_quiet) on random quiet
// roots at the time step T: returns the number of groups whose (integration, delta_t, best delta_t, threshold) differ in a
// bit, *applied = groups the closed form took.  (adder_lean_kernel applies the group form at whatever T the batch has.)
uint64_t sim_lean_group_check(float T, uint32_t groups, uint32_t seed, uint64_t *applied) {
    uint64_t bad = 0, done = 0;
    uint32_t x = seed * 2654435761u + 12345u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    for (uint32_t gi = 0; gi < groups; ++gi) {
        const uint32_t base = rnd() % 256u, n = 1u + rnd() % kQuietGroup, cth = rnd() % 4u;
        uint8_t v[kQuietGroup];
        for (uint32_t i = 0; i < n; ++i) {
            int d = (int)(rnd() % (2u * cth + 1u)) - (int)cth, q = (int)base + d;
            v[i] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
        }
        LeanPx p{};
        p.base = base;
        p.has0 = true;
        p.popped = true;
        const uint32_t frames = 1u + rnd() % 300u;  // a root that has accumulated `frames` frames of about base
        p.integ = (float)(base * frames);
        p.dt = 0.0f;
        for (uint32_t i = 0; i < frames; ++i) p.dt = fadd(p.dt, T);
        p.bdt = p.dt;
        p.thr = bits_to_f32((f32_to_bits(p.integ > 1.0f ? p.integ : 1.0f) & 0x7f800000u) + 0x00800000u);
        if (p.integ == 0.0f) p.thr = 0.0f;  // a black root
        LeanPx a = p, b = p;
        const QuietGroupStats g = quiet_group_stats(v, 1, n, quiet_group_need(p.integ, p.thr));
        const uint32_t r = lean_group_apply<ScalarLanes>(a, g, n, cth, T);
        if (r != kQuietDone) continue;
        ++done;
        for (uint32_t i = 0; i < n; ++i) lean_step_quiet<ScalarLanes>(b, v[i], T);
        if (f32_to_bits(a.integ) != f32_to_bits(b.integ) || f32_to_bits(a.dt) != f32_to_bits(b.dt) ||
            f32_to_bits(a.bdt) != f32_to_bits(b.bdt) || f32_to_bits(a.thr) != f32_to_bits(b.thr))
            ++bad;
    }
    if (applied) *applied = done;
    return bad;
}

// returns 0 ok, -4 capacity, -5 depth
int sim_integrate(Sim *s, const uint8_t *frame, float time_spanned, SimEvent *out, size_t cap, size_t *n_out) {
    StepConsts sc;
    sc.time_spanned = time_spanned;
    sc.running_t = s->running_t;
    sc.running_t_u32 = f32_as_u32(s->running_t);
    sc.dtm_f = (float)s->dtm;
    sc.ref_time = s->ref_time;
    sc.cth = s->c_thresh;
    sc.collapse = s->collapse;
    sc.abs_t = s->abs_t;
    sc.max_depth = s->max_depth;
    sc.ref_magic = s->ref_time >= 2 ? (uint32_t)(0x100000000ull / s->ref_time) : 0u;
    // the product's variant choice (adder_hip_api.cpp enqueue_frames): lean iff Collapse with
    // delta_t_max <= time_spanned and no generic batch has run since the last reset
    // feature path (adder_hip_api.cpp prepare_per_unit_c_thresh): one pair per unit from the first frame that can
    // make them differ
    const bool needs_perpx = s->roi_on || (s->feat_detect && s->feat_adjust && s->radius > 0);
    if (needs_perpx && !s->perpx) {
        s->cth_px.assign(s->N, s->c_thresh);
        s->cctr_px.assign(s->N, s->c_counter);
        s->perpx = 1;
    }
    const bool lean = s->collapse && (float)s->dtm <= time_spanned && !s->generic_sticky && s->use_fast && !s->perpx;
    if (!lean && !s->continuous && !s->feat_detect && !s->roi_on && sim_cb_possible(s, time_spanned))
        return sim_integrate_cb_block(s, frame, 1u, time_spanned, out, cap, n_out);
    if (!(time_spanned >= 1.0f) || time_spanned != (float)(uint32_t)time_spanned) s->frac_time_seen = 1;
    if (!lean) s->generic_sticky = 1;
    int rc = 0;
    Emitter em;
    em.out = out; em.cap = cap; em.pos = 0;
    size_t u = 0;
    for (uint32_t y = 0; y < s->H; y++)
        for (uint32_t x = 0; x < s->W; x++)
            for (uint32_t c = 0; c < s->C; c++, u++) {
                const uint32_t hdr = s->hdr[u];
                uint32_t m = hdr_m(hdr);
                // garbage on purpose: level 0 must not be read when m == 0
                const float gi = m ? s->integ0[u] : -12345.0f, gd = m ? s->dt0[u] : -777.0f,
                            gb = m ? s->bdt0[u] : -999.0f;
                const float glf = s->abs_t ? s->lastf[u] : -1.0f;
                const uint32_t v = frame[u];
                em.x = (uint16_t)x; em.y = (uint16_t)(y + s->row_begin); em.c = s->C == 1 ? 0xFF : (uint8_t)c;
                if (s->continuous) {
                    APx p = apx_unpack(s->c_hdr[u], s->lastf[u]);
                    ContAcc acc{s, u};
                    const bool ok = s->abs_t ? cont_step<true>(p, acc, v, (float)v, time_spanned, sc, s->max_depth + 1, em)
                                             : cont_step<false>(p, acc, v, (float)v, time_spanned, sc, s->max_depth + 1, em);
                    if (!ok) rc = -5;
                    s->c_hdr[u] = apx_hdr(p);
                    s->lastf[u] = p.lastf;
                    if (p.length > s->max_m) s->max_m = p.length;
                    continue;
                }
                if (s->perpx) {
                    sc.cth = s->cth_px[u];
                    c_thresh_advance(s->cth_px[u], s->cctr_px[u], (uint8_t)s->c_max, (uint8_t)s->velocity, time_spanned,
                                     s->ref_time);
                }
                if (lean) {
                    LeanPx p = lean_unpack<ScalarLanes>(hdr, gi, gd, gb, glf);
                    if (s->cb_quiet_path && lean_quiet(p, v, sc.cth)) {
                        // the blocked kernel's quiet-frame loop: must equal the step and leave no record
                        if (!lean_quiet_keeps(p, v) || (s->lean_steps & 1u)) lean_step_quiet<ScalarLanes, true>(p, v, time_spanned);
                        else lean_step_quiet<ScalarLanes, false>(p, v, time_spanned);
                        s->lean_quiet_steps++;
                        s->lean_steps++;
                        s->hdr[u] = lean_hdr(p);
                        s->integ0[u] = p.integ; s->dt0[u] = p.dt; s->bdt0[u] = p.bdt;
                        if (s->abs_t) s->lastf[u] = p.lastf;
                        s->running[u] = (uint8_t)frame_value_u8(lean_bd_from_thr(f32_to_bits(p.thr)), f32_as_u32(p.bdt),
                                                               (double)s->ref_time);
                        continue;
                    }
                    LeanRec rec;
                    const uint32_t tag = (uint32_t)(u & 127u) << kLeanUnitShift;  // unit inside its 128-unit segment
                    const LeanFlagsT<ScalarLanes> fl = s->abs_t ? lean_step<true>(p, v, sc.cth, time_spanned, sc, tag, rec)
                                                                 : lean_step<false>(p, v, sc.cth, time_spanned, sc, tag, rec);
                    if (fl.b && !fl.a) rc = -9;
                    if (fl.a || fl.c) {
                        if (((rec.w >> kLeanUnitShift) & 255u) != (u & 127u)) rc = -9;  // the tag must survive
                        // (DeltaT batches park the 8-byte form, as the device kernels do)
                        const LeanEvents e = s->abs_t ? lean_decode(rec, true, sc.running_t_u32)
                                                      : lean_decode8(rec.ta, rec.w8, time_spanned, sc.running_t_u32);
                        if (!s->abs_t && ((rec.w8 >> kLean8UnitShift) & 127u) != (u & 127u)) rc = -9;
                        if (e.a != fl.a || e.b != fl.b || e.c != fl.c) rc = -9;
                        if (e.a) em(e.da, e.ta);
                        if (e.b) em(kDEmpty, e.tb);
                        if (e.c) em(e.dc, e.tc);
                    } else if (rec.w & (kLeanA | kLeanB | kLeanC)) {
                        rc = -9;
                    }
                    s->hdr[u] = lean_hdr(p);
                    if (p.has0) { s->integ0[u] = p.integ; s->dt0[u] = p.dt; s->bdt0[u] = p.bdt; }
                    if (s->abs_t) s->lastf[u] = p.lastf;
                    if (p.has0 && s->max_m < 1) s->max_m = 1;
                    if (p.has0)  // the side plane as adder_lean1_kernel writes it
                        s->running[u] = (uint8_t)frame_value_u8(lean_bd_from_thr(f32_to_bits(p.thr)), f32_as_u32(p.bdt),
                                                               (double)s->ref_time);
                    s->lean_steps++;
                    continue;
                }
                PxState p = px_unpack(hdr, gi, gd, gb, glf);
                size_t before = em.pos;
                DeepAcc deep{s, u};
                GenPlan plan;
                if (s->collapse) gen_root<true>(p, v, sc, plan); else gen_root<false>(p, v, sc, plan);
                if (s->abs_t) gen_emit<true>(p, plan, sc, deep, em); else gen_emit<false>(p, plan, sc, deep, em);
                if (!gen_walk(p, v, plan, sc, deep)) rc = -5;
                gen_pop(p, plan, deep);
                if (plan.walk) s->generic_steps++; else s->fast_steps++;
                if (em.pos - before != plan.count) s->plan_mismatch++;
                s->hdr[u] = px_hdr(p);
                m = p.m;
                if (m > s->max_m) s->max_m = m;
                if (m > 0) { s->integ0[u] = p.n0.integ; s->dt0[u] = p.n0.dt; s->bdt0[u] = p.n0.bdt; }
                if (s->abs_t) s->lastf[u] = p.lastf;
                if (m != 0u) s->running[u] = (uint8_t)frame_value_u8(p.n0.bd, f32_as_u32(p.n0.bdt), (double)s->ref_time);
            }
    // ---- adder_feature_kernel, serially ----
    if (s->feat_detect && em.pos <= cap) {
        const uint32_t low = s->baseline < 2 ? s->baseline : 2;
        const uint32_t radius = s->feat_adjust ? s->radius : 0u;
        for (uint64_t i = 0; i < em.pos; ++i) {
            if (!feature_looked_at(out, 0, em.pos, i, s->row_begin, s->chunk_rows)) continue;
            const uint32_t x = out[i].x, y = out[i].y - s->row_begin;
            uint8_t *member = &s->fset[(size_t)y * s->W + x];
            if (fast9_is_feature(s->running.data(), s->W, s->H, s->C, x, y)) {
                const bool is_new = *member == 0;
                *member = 1;
                if (is_new) {
                    s->new_features++;
                    if (radius) {
                        const uint32_t x0 = x > radius ? x - radius : 0u, y0 = y > radius ? y - radius : 0u;
                        const uint32_t x1 = x + radius < s->W - 1 ? x + radius : s->W - 1;
                        const uint32_t y1 = y + radius < s->H - 1 ? y + radius : s->H - 1;
                        for (uint32_t yy = y0; yy <= y1; ++yy)
                            for (uint32_t xx = x0; xx <= x1; ++xx)
                                for (uint32_t cc = 0; cc < s->C; ++cc) s->cth_px[((size_t)yy * s->W + xx) * s->C + cc] = (uint8_t)low;
                    }
                }
            } else {
                *member = 0;
            }
        }
    }
    if (s->roi_on) {
        const uint32_t low = s->baseline < 2 ? s->baseline : 2;
        for (uint32_t yy = s->roi[1]; yy <= s->roi[3] && yy < s->H; ++yy)
            for (uint32_t xx = s->roi[0]; xx <= s->roi[2] && xx < s->W; ++xx)
                for (uint32_t cc = 0; cc < s->C; ++cc) s->cth_px[((size_t)yy * s->W + xx) * s->C + cc] = (uint8_t)low;
    }
    s->running_t += time_spanned;
    c_thresh_advance(s->c_thresh, s->c_counter, (uint8_t)s->c_max, (uint8_t)s->velocity, time_spanned, s->ref_time);
    *n_out = em.pos;
    if (em.pos > cap && rc == 0) rc = -4;
    return rc;
}
}

// ------------------------------------------------------------------------------------------
// Host run of the framer's device step (csrc/adder_framer.hpp): the events are applied in
// stream order with framer_step, frames are kept as a plain [frames][units] array and the
// complete ones are min(last_filled) + 1 -- the formulation the GPU framer uses.  Compared
// with the framer oracle (a literal restatement of the reference's deque machinery) on CPU.
// ------------------------------------------------------------------------------------------
#include "adder_framer.hpp"
#include <vector>

// fast_div against the hardware division: returns the number of mismatches over n[0..count)
extern "C" uint64_t sim_fast_div_check(uint32_t d, const uint32_t *n, size_t count) {
    const adder::FastDivU32 f = adder::fast_div_make(d);
    uint64_t bad = 0;
    for (size_t i = 0; i < count; ++i) bad += adder::fast_div(n[i], f) != n[i] / d;
    return bad;
}

static uint32_t g_view_mode = 0, g_source_type = 0, g_view_dtm = 0, g_value_type = 0;
static float g_practical_d_max = 0.0f;
// what the next sim_framer_run shows (FramedViewMode / SourceType; 0, 0 = the U8 Intensity view)
extern "C" void sim_framer_set_view(uint32_t view_mode, uint32_t source_type, float practical_d_max, uint32_t delta_t_max) {
    g_view_mode = view_mode;
    g_source_type = source_type;
    g_practical_d_max = practical_d_max;
    g_view_dtm = delta_t_max;
}
// the frame element type T of the next sim_framer_run: 0 u8, 1 u16, 2 u32 (elements come out big-endian)
extern "C" void sim_framer_set_value_type(uint32_t value_type) { g_value_type = value_type; }

extern "C" int64_t sim_framer_run(const SimEvent *ev, size_t n, uint32_t width, uint32_t height, uint32_t channels,
                                  uint32_t tpf, uint32_t ref_interval, uint32_t abs_t, uint32_t round_up,
                                  uint8_t *out, size_t out_cap_frames) {
    using namespace adder;
    const size_t units = (size_t)width * height * channels;
    std::vector<FramerPx> px(units);
    for (auto &p : px) { p.ts = 0; p.lastf = -1; p.lasti = 0; }
    std::vector<uint32_t> frames;
    const FramerConsts k = framer_consts(tpf, ref_interval, abs_t, round_up, g_view_mode, g_source_type, g_practical_d_max, g_view_dtm,
                                         g_value_type);
    const size_t elem = (size_t)1 << g_value_type;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t c = ev[i].c == 0xFF ? 0u : ev[i].c;
        if (ev[i].x >= width || ev[i].y >= height || c >= channels) return -1;
        const size_t u = ((size_t)ev[i].y * width + ev[i].x) * channels + c;
        int32_t from = 0, to = 0;
        bool overflow = false;
        if (framer_step(px[u], ev[i].d, ev[i].t, k, from, to, overflow)) {
            if ((size_t)(to + 1) * units > frames.size()) frames.resize((size_t)(to + 1) * units, 0);
            for (int32_t f = from + 1; f <= to; ++f) frames[(size_t)f * units + u] = px[u].lasti;
        }
        if (overflow) return -2;
    }
    int32_t mn = 0x7fffffff;
    for (auto &p : px) mn = p.lastf < mn ? p.lastf : mn;
    const int64_t complete = (int64_t)mn + 1;
    if (complete > (int64_t)out_cap_frames) return -3;
    for (size_t i = 0; complete > 0 && i < (size_t)complete * units; ++i)
        for (size_t b = 0; b < elem; ++b) out[i * elem + b] = (uint8_t)(frames[i] >> (8 * (elem - 1 - b)));
    return complete;
}
