// adder_hip_api.cpp -- C-ABI of the MI355X framed->ADDER integration path (include/adder_hip.h).
//
// Owns what Video<W> owns for this path in the reference (video.rs:322-346): the pixel
// state (here: structure-of-arrays planes in HBM) and the step parameters; the caller
// owns frames and event buffers.  There is NO CPU fallback: without a gfx950 device
// adder_hip_create fails with ADDER_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/adder_hip.h"
#include "adder_kernels.h"

using namespace adder;

static_assert(sizeof(AdderEvent) == 12, "AdderEvent must be 12 bytes");
static_assert(sizeof(AdderEventPod) == sizeof(AdderEvent), "layout mismatch");

static thread_local std::string g_create_error;

struct AdderHipCtx {
    AdderHipParams p{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t rows = 0, n_units = 0, num_tiles = 0, num_chunks = 0, grid = 0, num_cus = 0;
    size_t n_pad = 0;
    uint32_t max_depth = 0;
    // state planes: level 0 = {hdr, integ0, dt0, bdt0}; levels >= 1 in the deep planes (allocated by the
    // first generic batch: the lean variants never touch them)
    uint32_t *hdr = nullptr;
    float *integ0 = nullptr, *dt0 = nullptr, *bdt0 = nullptr;
    float *lastf = nullptr;
    float *dv_integ = nullptr, *dv_dt = nullptr, *dv_bdt = nullptr;
    uint8_t *dv_bd = nullptr;
    // Mode::Continuous contexts: node planes of the general arena ([max_depth + 1][n_pad]); hdr holds the arena header
    bool continuous = false;
    float *cn_integ = nullptr, *cn_dt = nullptr, *cn_bdt = nullptr;
    uint32_t *cn_meta = nullptr;
    uint8_t *state_slab = nullptr;  // hdr, lastf, status, integ0, dt0, bdt0 live in here
    size_t reset_bytes = 0;         // hdr .. status: one memset per reset
    size_t slab_bytes = 0;
    uint8_t *running = nullptr;       // the context's own rows, inside running_base
    uint8_t *running_base = nullptr;  // kFeatureHalo rows of the band above + the own rows (n_pad) + kFeatureHalo rows below
    bool running_enabled = false;
    uint32_t *d_new_xy = nullptr;     // row-band feature mode: this frame's new features (x | plane y << 16)
    uint32_t new_xy_cap = 0;
    bool band_frame_pending = false;  // a frame's events await adder_hip_feature_detect
    // undo copy of the pixel state, taken before a batch whose event buffer is smaller than the batch's worst
    // case: an overflow then rolls the state back and the caller retries with the size reported
    struct Snapshot {
        uint8_t *slab = nullptr;  // copy of state_slab
        float *dv_integ = nullptr, *dv_dt = nullptr, *dv_bdt = nullptr;
        uint8_t *dv_bd = nullptr;
        float *cn_integ = nullptr, *cn_dt = nullptr, *cn_bdt = nullptr;
        uint32_t *cn_meta = nullptr;
        bool valid = false, deep = false;
        float running_t = 0.0f;
        uint8_t c_thresh = 0, c_counter = 0;
        uint64_t frames_done = 0, run_bound = 0;
        bool generic_sticky = false;
        uint8_t *cth_px = nullptr, *cctr_px = nullptr, *fset = nullptr, *running = nullptr;
        bool perpx = false, has_running = false;
    } snap;
    // feature-driven rate control + ROI (video.rs:825-837, 865-1112, 1291-1293)
    bool feat_detect = false, feat_adjust = false;
    uint8_t c_thresh_baseline = 2;   // Crf::new(None): CRF[3][0] (rate_controller.rs:21,62)
    uint16_t feature_c_radius = 0;   // CRF[3][3] * min_resolution, set in create
    bool roi_on = false;
    uint16_t roi[4] = {0, 0, 0, 0};  // start.x, start.y, end.x, end.y (inclusive)
    uint8_t *cth_px = nullptr, *cctr_px = nullptr;  // per-unit pair once it stops being uniform (perpx)
    uint8_t *fset = nullptr;         // [rows][width] VideoState::features membership
    uint32_t *d_feat_counters = nullptr;
    uint32_t last_new_features = 0;
    bool perpx = false;              // sticky until adder_hip_reset / reset_c_thresh
    // c_thresh / c_increase_counter: identical in every pixel (adder_pixel.hpp header comment)
    uint8_t c_thresh = 10, c_counter = 1;
    // a generic batch has run since create / reset: pixels may hold more than one fired level, which only
    // the generic kernels understand -- the lean variant is not chosen again until adder_hip_reset
    bool generic_sticky = false;
    // ordered-compaction scratch: a ring of two chunks of frames (a chunk is stepped while the one before
    // it is scanned and expanded)
    uint8_t *park_ring = nullptr;    // [slots][num_waves][park_bytes]
    uint32_t park_group_shift = 0;   // ring layout of temporally blocked batches (park_offset)
    // graph instances no longer wanted (tuning losers, evicted batch lengths): destroyed with the context.  This HIP
    // runtime (ROCm 7.0) crashed in hip::Graph::UpdateStreams at a later hipGraphLaunch of ANOTHER instance once a
    // few dozen instances had been destroyed mid-life (rocgdb backtrace; tests: 22 batch lengths on one context).
    std::vector<hipGraphExec_t> retired_execs;
    uint32_t park_bytes = 0;         // scratch of one segment of one frame (fixed-slot kinds)
    // what the scratch ring is laid out for: fixed slots per segment and frame (lean records, Continuous staging), or
    // one record log per segment and chunk (per-event records of the generic / bounded Collapse kernels: log_capacity)
    enum ScratchKind { kScratchNone, kScratchLean, kScratchLean8, kScratchCont, kScratchLog2, kScratchLog3 };
    ScratchKind scratch_kind = kScratchNone;
    uint32_t log_cap = 0;            // records per (segment, chunk) region (log kinds)
    uint32_t *wofs_ring = nullptr;   // [slots][num_waves] log kinds: where a segment's run of a frame starts
    uint32_t *wcur = nullptr;        // [ring_chunks][num_waves] log cursors between the launches of a chunk
    // the bounded Collapse step needs exact integer sums (adder_pixel.hpp): a fractional time_spanned or a huge
    // delta_t_max since the last reset rules it out
    bool frac_time_seen = false;
    uint32_t dtm_max_seen = 0;
    // constant runs (adder_cr_kernel): every dense frame since the last reset was tested with c_thresh 0 at ONE integer
    // time_spanned, so every arena is a function of (its run's intensity, the run's length) -- adder_pixel.hpp
    bool cr_valid = true;
    float cr_time = 0.0f;
    // The integer-state kernels need rho * max(255, time_spanned) < 2^24 for every run length rho.  run_bound = an upper bound
    // of any unit's run: it grows by every frame queued and comes down to what the kernels REPORT at the end of a batch
    // (BatchResult::max_run) -- so a stream whose pixels keep changing stays on the integer kernels for good, where "frames
    // since the reset" sent every stream to the float kernels after 65 793 frames (18 minutes of video)
    uint64_t run_bound = 0, pending_end_frames = 0;
    bool pending_reports = false;
    hipEvent_t null_join = nullptr;  // stream == NULL: the batch is ordered behind the legacy default stream (join_null_stream)
    uint32_t *d_run_max = nullptr;
    uint32_t *wtot_ring = nullptr;   // [slots][num_waves]
    uint32_t *wpref_ring = nullptr;  // [slots][num_waves]
    uint32_t *ftot_ring = nullptr;   // [slots]
    uint32_t chunk = 1, slots = 2, ring_chunks = 3;
    uint32_t lean_blocks_per_cu = 0, expand_blocks_per_cu = 0;  // walking grids of the lean path when > 0 (0 = full grids)
    uint32_t gen_blocks_per_cu = 0, gen_expand_blocks_per_cu = 0;  // generic variants: 0 = full grids
    uint32_t frames_per_launch = kMaxFramesPerLaunch;  // temporal blocking depth of the frame kernels
    uint32_t num_waves = 0;
    // device-resident batch description (kernels take {BatchArgs*, f}) + its pinned host mirror
    BatchArgs *d_batch = nullptr;
    BatchArgs *h_batch = nullptr;   // pinned
    FrameTab *d_ftab = nullptr;     // per-frame uniforms (running_t, c_thresh)
    FrameTab *h_ftab = nullptr;     // pinned
    // the batch description and the frame table share one device block and one pinned block (one upload per batch)
    uint8_t *d_desc = nullptr, *h_desc = nullptr;
    BatchResult *h_result = nullptr;  // pinned: adder_publish_kernel writes it, finish() reads it
    hipEvent_t reset_e = nullptr;     // adder_hip_reset's memsets (queued on the context's stream, not waited for)
    bool reset_pending = false;
    size_t ftab_cap = 0;            // entries
    // capture streams/events and the cache of instantiated frame-loop graphs
    hipStream_t cap_s = nullptr, cap_s2 = nullptr;
    hipEvent_t cap_e1 = nullptr, cap_e2[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // One cached launch sequence.  How well the two branches of an instantiated graph overlap is decided when it is
    // instantiated (the runtime binds the branches to hardware queues then; measured: the same graph comes out at
    // 1.86 or 2.08 ms per 300-frame step from one instantiation to the next, stable for the life of the exec), so a
    // key keeps a few candidates, runs each on real batches while it is new, and settles on the fastest.
    struct GraphTune {
        std::vector<hipGraphExec_t> cand;
        std::vector<float> ms;    // best batch time seen per candidate
        std::vector<int> runs;
        int chosen = -1;          // settled
        int last = -1;            // candidate of the batch in flight
        int recheck = 0;          // extra runs of candidate 0 after the others (it ran first, on a cold chip)
    };
    std::map<uint64_t, GraphTune> graphs;  // key: see get_graph
    uint64_t tune_key = 0;
    bool tune_pending = false;
    uint32_t graph_candidates = 6;
    bool use_graph = true;
    bool eager_two_streams = false;
    // ADDER_HIP_CU_SPLIT=n (0 < n < CUs): spatial partitioning instead of time sharing -- the frame kernel of chunk k+1
    // on a stream masked to n CUs (the same share of every XCD), scan / offsets / expansion of chunk k on a stream masked
    // to the others; a batch's first frame-kernel run and last expansion have no partner and take the unmasked stream.
    uint32_t *status = nullptr;   // device status word
    uint64_t *d_rec_total = nullptr;  // parked records of the last batch (diagnostics)
    // [0]: adder_log_pack_kernel's total (written, never read back); [1] (as u32): the status word of
    // adder_hip_expand_records_device's kernels -- they run on a side stream beside root's own batches, so they must not
    // share the batches' word (a capacity flag of theirs would be charged to an unrelated batch, and clearing theirs
    // could erase a flag a batch in flight raised)
    uint64_t *d_side_words = nullptr;
    bool wire_batch = false;      // the batch being enqueued writes the raw sink's records instead of AdderEvents (adder_hip_integrate_wire_device)
    uint8_t *d_rr_tab = nullptr;  // [256][kRrTabRows] chain lengths of the run-records step (built at its first batch)
    uint32_t *d_lr_tab = nullptr; // [kLrTabWords] the lean-runs expansion's events by (base_val, rho) / input, for lr_tab_time
    float lr_tab_time = 0.0f;
    // records over the wire (adder_hip_integrate_records_device / adder_hip_expand_records_device)
    bool records_only = false;        // the batch being queued stops after its scan
    uint32_t last_variant = 0;        // kernel choice of the batch queued last (enqueue_frames)
    float last_time_spanned = 0.0f;   // of the last batch (root's expansion uses its consts and frame table)
    // root: n_bands BatchArgs + pointer / destination tables per call, kBandDescSlots calls' worth in turn (a slot's
    // host block is reused once the upload queued from it has gone through: no wait for the stream's other work)
    static constexpr uint32_t kBandDescSlots = 4;
    uint8_t *d_band_desc = nullptr, *h_band_desc = nullptr;
    size_t band_desc_cap = 0;  // bytes of ONE slot
    uint32_t band_desc_next = 0;
    hipEvent_t band_desc_e[kBandDescSlots] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long *d_timeline = nullptr;  // ADDER_HIP_TIMELINE diagnostics
    // sparse steps (adder_hip_integrate_sparse): running_t per unit, and the work buffers of a call
    bool sparse_mode = false;  // c_thresh / counter / running_t live per unit from the first sparse call on
    float *rt_px = nullptr;
    struct SparseWork {
        SparseStep *steps = nullptr;
        uint32_t *keys0 = nullptr, *keys1 = nullptr, *idx0 = nullptr, *idx1 = nullptr, *count = nullptr, *offs = nullptr;
        uint2 *stage = nullptr;
        void *temp = nullptr;
        unsigned long long *total = nullptr;
        size_t cap = 0, temp_bytes = 0;  // steps
    } sw;
    uint64_t last_records = 0;
    std::vector<hipEvent_t> post_events;  // launch timing: pairs around scan + offsets + expand of every chunk
    uint32_t timed_posts = 0;
    float last_post_avg_us = 0.0f;
    uint64_t *d_offsets = nullptr;  // internal frame offsets (host-buffer API)
    size_t d_offsets_cap = 0;       // all *_cap below are in BYTES
    // staging for the host-buffer API
    uint8_t *d_frames = nullptr;
    size_t d_frames_cap = 0;
    AdderEvent *d_events = nullptr;
    size_t d_events_cap = 0;
    uint8_t *d_wire = nullptr;      // wire-format bytes of the last raw batch
    size_t d_wire_cap = 0;
    // pipelined raw transcode (adder_hip_stream_*): two slots, each one batch in flight
    struct StreamSlot {
        uint8_t *d_wire = nullptr;
        size_t d_wire_cap = 0;
        uint8_t *h_wire = nullptr;   // pinned
        size_t h_wire_cap = 0;
        uint64_t *h_offsets = nullptr;  // pinned, [T+1]
        size_t h_offsets_cap = 0;
        hipEvent_t done = nullptr;   // wire kernel + D2H of this slot
        hipEvent_t wired = nullptr;  // wire kernel only (d_events may be overwritten after it)
        size_t n_events = 0, n_bytes = 0;
        uint32_t status = 0;
        bool busy = false;
    } slot[2];
    hipStream_t out_s = nullptr;     // wire + D2H stream of the pipeline
    uint64_t submitted = 0, collected = 0;
    // per-frame `consume` ring (adder_hip_frame_submit / _collect): every slot owns its frame, its events on
    // the device and in page-locked host memory, and its own batch description (the host-side copies are read
    // by the DMA engine after submit has returned)
    struct FrameSlot {
        uint8_t *d_frame = nullptr;
        AdderEvent *d_events = nullptr, *h_events = nullptr;
        size_t cap = 0;                 // events
        uint8_t *h_hdr = nullptr;       // pinned: FrameResult, then chunk offsets [num_chunks + 1]
        uint64_t *d_offsets = nullptr;  // [2]
        BatchArgs *d_batch = nullptr, *h_batch = nullptr;
        FrameTab *d_ftab = nullptr, *h_ftab = nullptr;
        hipEvent_t done = nullptr;
        uint32_t *d_counters = nullptr; // feature path: this frame's counters (the shared ones are reset per enqueue)
        AdderEvent *out = nullptr;      // where the events were sent (h_events, or the caller's pinned buffer)
        size_t out_cap = 0;
        bool wire = false;              // the slot holds 9 / 11-byte wire records (the ring's format when it was submitted)
        hipGraphExec_t out_graph = nullptr;  // the slot's hand-over (wire scatter + frame_out) as ONE launch
        uint64_t out_key[6] = {0, 0, 0, 0, 0, 0};  // what that graph baked in
    } fslot[4];
    std::vector<hipStream_t> dummy_streams;  // ADDER_HIP_RING_SKIP_STREAMS (diagnostic)
    // The per-frame ring's stream arrangement is MEASURED, like the graph instances: where the HIP runtime puts a stream among
    // its hardware queues decides whether a dependency between two streams is cheap or costs tens of microseconds, and nothing
    // in the API tells (one process ran the default-quality ring at 60 us per frame, the next at 115; profiles/r05_ring_queues.txt).
    // Candidates: {second stream A, upload stream A}, {post-processing on the context's own stream, upload A}, and three more
    // pairs of streams; each gets a window of 24 frames, the fastest stays for the context's life.
    struct RingCand {
        hipStream_t out_s = nullptr, in_s = nullptr;  // (borrowed from ring_streams)
        bool post_on_main = false;
        double us = 1e30;  // measured period per frame
    };
    std::vector<RingCand> ring_cands;
    std::vector<hipStream_t> ring_streams;  // owned
    int ring_cur = 0, ring_chosen = -1;
    uint32_t ring_win_frames = 0, ring_win_waits = 0, ring_win_skip = 0;
    std::chrono::steady_clock::time_point ring_win_t0;
    uint32_t graph_slot = 0;         // 1 + the frame slot whose description the batch being queued uses (0: the context's own)
    bool frame_only = false;         // the one-frame batch being queued stops after its frame kernel: the caller queues scan and
                                     // expansion itself, on another stream (the per-frame ring)
    bool frame_only_done = false;    // ... and enqueue_frames did so (false: the whole pipeline was queued -- feature mode, timing)
    uint32_t f_slots = 3;
    bool f_wire = false;             // the ring hands out 9 / 11-byte wire records instead of AdderEvents (adder_hip_frames_set_format)
    size_t f_events_per_slot = 0;    // 0: the mode's worst case, at most 2 GiB of events
    uint64_t f_submitted = 0, f_collected = 0;
    hipEvent_t frame_e = nullptr;
    hipStream_t in_s = nullptr;      // the ring's upload stream (wire format: the next frame's H2D beside this frame's kernels)
    uint64_t f_last_produced = 0;    // events of the frame collected last
    hipEvent_t in_e = nullptr;
    bool no_snapshot = false;
    uint32_t *d_chunks = nullptr;
    // running state
    float running_t = 0.0f;  // PixelArena::running_t (identical for all pixels)
    uint64_t frames_done = 0;
    bool poisoned = false;
    std::string err;
    // pending device batch
    bool pending = false;
    hipStream_t pending_stream = nullptr;
    uint64_t *pending_offsets = nullptr;
    uint32_t pending_frames = 0;
    size_t pending_cap = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    float last_ms = 0.0f;
    // optional per-launch timing (one HIP event pair around every frame launch)
    int launch_timing = 0;  // 1: a pair around every frame-kernel launch; 2: a pair around a chunk's run of them
    std::vector<hipEvent_t> launch_events;
    uint32_t timed_launches = 0, timed_frames = 0, timed_pairs = 0;
    float last_launch_avg_us = 0.0f;
};

static int fail(AdderHipCtx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define HIPCHK(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, ADDER_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

template <class T>
static hipError_t dalloc(T **p, size_t count) {
    return hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T));
}

static void free_ctx(AdderHipCtx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *ptrs[] = {c->state_slab, c->dv_integ, c->dv_dt,
                    c->dv_bdt,  c->dv_bd,   c->running_base, c->d_new_xy, c->cn_integ, c->cn_dt, c->cn_bdt, c->cn_meta,
                    c->snap.cn_integ, c->snap.cn_dt, c->snap.cn_bdt, c->snap.cn_meta,
                    c->d_offsets, c->d_frames, c->d_events, c->d_chunks, c->d_wire,
                    c->cth_px, c->cctr_px, c->fset, c->d_feat_counters, c->snap.cth_px, c->snap.cctr_px, c->snap.fset,
                    c->snap.running};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (void *p : {(void *)c->park_ring, (void *)c->wtot_ring, (void *)c->wpref_ring, (void *)c->ftot_ring,
                    (void *)c->wofs_ring, (void *)c->wcur})
        if (p) (void)hipFree(p);
    for (auto &sl : c->slot) {
        if (sl.d_wire) (void)hipFree(sl.d_wire);
        if (sl.h_wire) (void)hipHostFree(sl.h_wire);
        if (sl.h_offsets) (void)hipHostFree(sl.h_offsets);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.wired) (void)hipEventDestroy(sl.wired);
    }
    for (auto &fs : c->fslot) {
        if (fs.out_graph) (void)hipGraphExecDestroy(fs.out_graph);
        for (void *p : {(void *)fs.d_frame, (void *)fs.d_events, (void *)fs.d_offsets, (void *)fs.d_batch,
                        (void *)fs.d_counters})
            if (p) (void)hipFree(p);
        for (void *p : {(void *)fs.h_events, (void *)fs.h_hdr, (void *)fs.h_batch})
            if (p) (void)hipHostFree(p);
        if (fs.done) (void)hipEventDestroy(fs.done);
    }
    if (c->frame_e) (void)hipEventDestroy(c->frame_e);
    if (c->out_s) (void)hipStreamDestroy(c->out_s);
    if (c->in_s && c->in_s != c->out_s) (void)hipStreamDestroy(c->in_s);
    for (hipStream_t d : c->dummy_streams) (void)hipStreamDestroy(d);
    for (hipStream_t d : c->ring_streams) (void)hipStreamDestroy(d);
    if (c->in_e) (void)hipEventDestroy(c->in_e);
    for (void *p : {(void *)c->snap.slab, (void *)c->snap.dv_integ, (void *)c->snap.dv_dt, (void *)c->snap.dv_bdt,
                    (void *)c->snap.dv_bd})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : c->launch_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->post_events) (void)hipEventDestroy(e);
    if (c->d_rec_total) (void)hipFree(c->d_rec_total);
    if (c->d_side_words) (void)hipFree(c->d_side_words);
    if (c->d_rr_tab) (void)hipFree(c->d_rr_tab);
    if (c->d_run_max) (void)hipFree(c->d_run_max);
    if (c->d_lr_tab) (void)hipFree(c->d_lr_tab);
    if (c->d_band_desc) (void)hipFree(c->d_band_desc);
    if (c->h_band_desc) (void)hipHostFree(c->h_band_desc);
    for (hipEvent_t e : c->band_desc_e)
        if (e) (void)hipEventDestroy(e);
    if (c->d_timeline) (void)hipFree(c->d_timeline);
    for (void *p : {(void *)c->rt_px, (void *)c->sw.steps, (void *)c->sw.keys0, (void *)c->sw.keys1, (void *)c->sw.idx0,
                    (void *)c->sw.idx1, (void *)c->sw.count, (void *)c->sw.offs, (void *)c->sw.stage, c->sw.temp,
                    (void *)c->sw.total})
        if (p) (void)hipFree(p);
    for (auto &kv : c->graphs)
        for (hipGraphExec_t e : kv.second.cand)
            if (e) (void)hipGraphExecDestroy(e);
    for (hipGraphExec_t e : c->retired_execs) (void)hipGraphExecDestroy(e);
    if (c->d_desc) (void)hipFree(c->d_desc);
    if (c->h_desc) (void)hipHostFree(c->h_desc);
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->reset_e) (void)hipEventDestroy(c->reset_e);
    for (hipEvent_t e : {c->cap_e1, c->cap_e2[0], c->cap_e2[1], c->cap_e2[2], c->cap_e2[3], c->cap_e2[4]})
        if (e) (void)hipEventDestroy(e);
    if (c->cap_s) (void)hipStreamDestroy(c->cap_s);
    if (c->cap_s2) (void)hipStreamDestroy(c->cap_s2);
    if (c->null_join) (void)hipEventDestroy(c->null_join);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void adder_hip_default_params(AdderHipParams *p, uint16_t width, uint16_t height, uint8_t channels) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->abi_version = ADDER_HIP_ABI_VERSION;
    p->width = width;
    p->height = height;
    p->channels = channels;
    p->time_mode = ADDER_TIME_ABSOLUTE_T;   // TimeMode::default (lib.rs:77-79)
    p->multi_mode = ADDER_MULTI_COLLAPSE;   // PixelMultiMode::default (lib.rs:211-212)
    p->pixel_mode = ADDER_MODE_FRAME_PERFECT;
    p->row_begin = 0;
    p->row_end = height;
    p->ref_time = 255;      // VideoStateParams::default (video.rs:173-182)
    p->delta_t_max = 7650;
    p->c_thresh_max = 7;    // Crf::new(None) -> quality 3 (rate_controller.rs:21,55-70)
    p->c_increase_velocity = 7;
    p->c_thresh_start = 10; // PixelArena::new (event_pixel_tree.rs:82-83)
    p->c_counter_start = 1;
    p->chunk_rows = 1;      // VideoState::default (video.rs:230)
    p->max_depth = 16;
    p->device_id = -1;
}

static StepConsts make_consts(const AdderHipCtx *c, float time_spanned) {
    StepConsts sc;
    sc.time_spanned = time_spanned;
    sc.running_t = 0.0f;  // per frame: BatchArgs::ftab
    sc.running_t_u32 = 0u;
    sc.dtm_f = (float)c->p.delta_t_max;
    sc.ref_time = c->p.ref_time;
    sc.cth = 0u;          // per frame: BatchArgs::ftab
    sc.collapse = c->p.multi_mode == ADDER_MULTI_COLLAPSE ? 1u : 0u;
    sc.abs_t = c->p.time_mode == ADDER_TIME_ABSOLUTE_T ? 1u : 0u;
    sc.max_depth = c->max_depth;
    sc.ref_magic = c->p.ref_time >= 2 ? (uint32_t)(0x100000000ull / c->p.ref_time) : 0u;
    return sc;
}

static void base_args(const AdderHipCtx *c, FrameArgs *a) {
    memset(a, 0, sizeof *a);
    a->hdr = c->hdr;
    a->integ0 = c->integ0;
    a->dt0 = c->dt0;
    a->bdt0 = c->bdt0;
    a->lastf = c->lastf;
    a->dv_integ = c->dv_integ;
    a->dv_dt = c->dv_dt;
    a->dv_bdt = c->dv_bdt;
    a->dv_bd = c->dv_bd;
    a->cn_integ = c->cn_integ;
    a->cn_dt = c->cn_dt;
    a->cn_bdt = c->cn_bdt;
    a->cn_meta = c->cn_meta;
    a->max_nodes = c->max_depth + 1u;
    a->stage_events = c->max_depth + 3u;
    a->running = c->running_enabled ? c->running : nullptr;
    a->cth_px = c->perpx ? c->cth_px : nullptr;
    a->cctr_px = c->perpx ? c->cctr_px : nullptr;
    a->c_max = c->p.c_thresh_max;
    a->c_vel = c->p.c_increase_velocity;
    a->plane_stride = c->n_pad;
    a->status = c->status;
    a->n_units = c->n_units;
    a->num_waves = c->num_waves;
    a->width = c->p.width;
    a->channels = c->p.channels;
    a->rowlen = (uint32_t)c->p.width * c->p.channels;
    a->row_begin = c->p.row_begin;
}

static size_t halo_bytes(const AdderHipCtx *c) { return (size_t)kFeatureHalo * c->p.width * c->p.channels; }
static int alloc_running(AdderHipCtx *c, hipStream_t s) {
    if (c->running) return ADDER_OK;
    const size_t bytes = c->n_pad + 2 * halo_bytes(c);
    HIPCHK(c, dalloc(&c->running_base, bytes));
    HIPCHK(c, hipMemsetAsync(c->running_base, 0, bytes, s));
    c->running = c->running_base + halo_bytes(c);
    return ADDER_OK;
}

// Video::new (video.rs:350-438): every pixel = PixelArena::new(1.0, coord): base_val 0,
// c_thresh 10, counter 1, one pristine node (m = 0), last_fired_t 0, running_t 0.
static int init_state(AdderHipCtx *c) {
    const AdderHipParams &p = c->p;
    // header word 0 = base_val 0, no fired level, not popped.  The level planes are only read where
    // the header says they are live (or are selected away: lean step), so they need no clearing.
    if (c->continuous) {
        // PixelArena::new(1.0): one node {d: 0, integration 0, delta_t 0, no best event}, length 1
        const size_t cnt = c->n_pad * (c->max_depth + 1u);
        HIPCHK(c, adder_launch_fill_u32(c->hdr, c->n_pad, apx_hdr(APx{0u, 1u, false, 0.0f}), c->stream));
        HIPCHK(c, hipMemsetAsync(c->cn_integ, 0, cnt * sizeof(float), c->stream));
        HIPCHK(c, hipMemsetAsync(c->cn_dt, 0, cnt * sizeof(float), c->stream));
        HIPCHK(c, hipMemsetAsync(c->cn_bdt, 0, cnt * sizeof(float), c->stream));
        HIPCHK(c, hipMemsetAsync(c->cn_meta, 0, cnt * sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(c->lastf, 0, c->n_pad * sizeof(float), c->stream));
        HIPCHK(c, hipMemsetAsync(c->status, 0, sizeof(uint32_t), c->stream));
    } else {
        HIPCHK(c, hipMemsetAsync(c->state_slab, 0, c->reset_bytes, c->stream));  // hdr, last_fired_t, status
    }
    if (c->running_base) HIPCHK(c, hipMemsetAsync(c->running_base, 0, c->n_pad + 2 * halo_bytes(c), c->stream));
    c->band_frame_pending = false;
    c->c_thresh = p.c_thresh_start;
    c->c_counter = p.c_counter_start;
    c->generic_sticky = false;
    c->frac_time_seen = false;
    c->cr_valid = true;
    c->cr_time = 0.0f;
    c->run_bound = 0;
    c->pending_reports = false;
#ifndef ADDER_DBG_NO_RUNMAX_RESET
    if (c->d_run_max) HIPCHK(c, hipMemsetAsync(c->d_run_max, 0, sizeof(uint32_t), c->stream));
#endif
    c->dtm_max_seen = p.delta_t_max;
    c->perpx = false;
    c->sparse_mode = false;
    if (c->fset) HIPCHK(c, hipMemsetAsync(c->fset, 0, (size_t)c->rows * p.width, c->stream));
    c->running_t = 0.0f;
    c->frames_done = 0;
    c->poisoned = false;
    return ADDER_OK;
}

// The scratch ring holds two chunks of frames: chunk k is stepped while chunk k-1 is scanned and expanded
// (launch_frame_loop makes the step of chunk k wait for the expansion of chunk k-2).


static int alloc_scratch(AdderHipCtx *c, AdderHipCtx::ScratchKind kind);
static bool env_flag(const char *name) {
    const char *e = getenv(name);
    return e && atoi(e) != 0;
}

// BatchArgs followed (256-byte aligned) by the frame table, on the device and in page-locked host memory
constexpr size_t kBatchDescBytes = (sizeof(BatchArgs) + 255) & ~(size_t)255;
static int alloc_batch_desc(AdderHipCtx *c, size_t frames) {
    uint8_t *od = c->d_desc, *oh = c->h_desc;
    c->d_desc = c->h_desc = nullptr;
    c->d_batch = c->h_batch = nullptr;
    c->d_ftab = c->h_ftab = nullptr;
    c->ftab_cap = 0;
    if (od) HIPCHK(c, hipFree(od));
    if (oh) HIPCHK(c, hipHostFree(oh));
    const size_t bytes = kBatchDescBytes + frames * sizeof(FrameTab);
    HIPCHK(c, dalloc(&c->d_desc, bytes));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_desc), bytes, hipHostMallocDefault));
    c->d_batch = reinterpret_cast<BatchArgs *>(c->d_desc);
    c->h_batch = reinterpret_cast<BatchArgs *>(c->h_desc);
    c->d_ftab = reinterpret_cast<FrameTab *>(c->d_desc + kBatchDescBytes);
    c->h_ftab = reinterpret_cast<FrameTab *>(c->h_desc + kBatchDescBytes);
    c->ftab_cap = frames;
    return ADDER_OK;
}

extern "C" int adder_hip_create(const AdderHipParams *params, AdderHipCtx **out) {
    if (!out) return fail(nullptr, ADDER_E_BAD_PARAMS, "out is null");
    *out = nullptr;
    if (!params) return fail(nullptr, ADDER_E_BAD_PARAMS, "params is null");
    AdderHipParams p = *params;
    if (p.abi_version != ADDER_HIP_ABI_VERSION)
        return fail(nullptr, ADDER_E_BAD_PARAMS, "abi_version %u != %u", p.abi_version, ADDER_HIP_ABI_VERSION);
    if (p.width == 0 || p.height == 0)  // PlaneSize::new (lib.rs:103-118)
        return fail(nullptr, ADDER_E_BAD_PARAMS, "invalid plane %ux%ux%u", p.width, p.height, p.channels);
    if (p.channels != 1 && p.channels != 3)
        return fail(nullptr, ADDER_E_BAD_PARAMS, "channels must be 1 or 3 (got %u)", p.channels);
    if (p.pixel_mode != ADDER_MODE_FRAME_PERFECT && p.pixel_mode != ADDER_MODE_CONTINUOUS)
        return fail(nullptr, ADDER_E_BAD_PARAMS, "bad pixel_mode");
    if (p.time_mode > ADDER_TIME_MIXED || p.multi_mode > ADDER_MULTI_COLLAPSE)
        return fail(nullptr, ADDER_E_BAD_PARAMS, "bad time_mode / multi_mode");
    if (p.row_begin == 0 && p.row_end == 0) p.row_end = p.height;
    if (p.row_begin >= p.row_end || p.row_end > p.height)
        return fail(nullptr, ADDER_E_BAD_PARAMS, "bad row band [%u,%u) of %u", p.row_begin, p.row_end, p.height);
    if (p.ref_time == 0) return fail(nullptr, ADDER_E_BAD_PARAMS, "ref_time must be > 0");
    if (p.delta_t_max < p.ref_time)  // video.rs:527-533
        return fail(nullptr, ADDER_E_BAD_PARAMS, "delta_t_max %u is smaller than ref_time %u", p.delta_t_max,
                    p.ref_time);
    if (p.delta_t_max % p.ref_time != 0)  // framed.rs:100-109
        return fail(nullptr, ADDER_E_BAD_PARAMS, "delta_t_max must be a multiple of ref_time");
    if (p.c_increase_velocity == 0) return fail(nullptr, ADDER_E_BAD_PARAMS, "c_increase_velocity must be >= 1");
    if (p.chunk_rows == 0) p.chunk_rows = 1;
    if (p.max_depth == 0) p.max_depth = 16;
    if (p.max_depth > kMaxDepthLimit)
        return fail(nullptr, ADDER_E_BAD_PARAMS, "max_depth must be <= %u", kMaxDepthLimit);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, ADDER_E_NO_DEVICE, "no HIP device visible; this path has no CPU fallback");
    int dev = p.device_id;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) return fail(nullptr, ADDER_E_BAD_PARAMS, "device_id %d out of range (%d devices)", dev, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return fail(nullptr, ADDER_E_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, ADDER_E_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", dev,
                    prop.gcnArchName);

    AdderHipCtx *c = new (std::nothrow) AdderHipCtx();
    if (!c) return fail(nullptr, ADDER_E_HIP, "out of host memory");
    c->p = p;
    c->device = dev;
    c->num_cus = (uint32_t)prop.multiProcessorCount;
    c->rows = p.row_end - p.row_begin;
    const uint64_t units = (uint64_t)c->rows * p.width * p.channels;
    // the kernels address a state plane with 32-bit byte offsets: 4 bytes per unit below 4 GiB
    if (units > 0x3ffffc00ull) {
        delete c;
        return fail(nullptr, ADDER_E_BAD_PARAMS, "row band too large (%llu pixel-channels)", (unsigned long long)units);
    }
    c->n_units = (uint32_t)units;
    // pad to a whole number of K1 blocks and of segment groups (the expand kernel's unit)
    {
        const uint32_t quantum = std::max<uint32_t>(kTileUnits, (uint32_t)ADDER_EXPAND_SEGS * kWaveUnits);
        c->n_pad = (size_t)((c->n_units + quantum - 1) / quantum) * quantum;
        c->num_tiles = (uint32_t)(c->n_pad / kTileUnits);  // K1 blocks
    }
    c->num_waves = (uint32_t)(c->n_pad / kWaveUnits);
    c->num_chunks = (c->rows + p.chunk_rows - 1) / p.chunk_rows;
    c->max_depth = p.max_depth;
    // Crf::new(None, plane): quality 3 -> feature_c_radius = (CRF[3][3] * min_resolution as f32) as u16
    // (rate_controller.rs:17,66-67)
    c->feature_c_radius = (uint16_t)((1.0f / 15.0f) * (float)std::min<uint32_t>(p.width, p.height));

    int rc = ADDER_OK;
    auto setup = [&]() -> int {
        HIPCHK(c, hipSetDevice(dev));
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreate(&c->ev_start));
        HIPCHK(c, hipEventCreate(&c->ev_stop));
        {
            // The five level-0 planes come from one allocation, each pushed a few KiB further than the one before:
            // separate allocations of this size all start on the same 2 MiB boundary, a wave then reads the same
            // offset of four planes at once, and whether those four streams fall on the same HBM channels was left
            // to the allocator (measured: the same build ran at 1.82 or 1.97 ms per step from context to context).
            size_t skew = 4352;
            if (const char *e = getenv("ADDER_HIP_PLANE_SKEW")) skew = (size_t)atoi(e) & ~(size_t)255;
            const size_t plane = (c->n_pad * sizeof(uint32_t) + 255) & ~(size_t)255;
            c->slab_bytes = 5 * (plane + skew) + 256;
            HIPCHK(c, dalloc(&c->state_slab, c->slab_bytes));
            // header plane, last_fired_t plane and the status word first: what a reset clears is one range
            uint8_t *q = c->state_slab;
            c->hdr = reinterpret_cast<uint32_t *>(q);
            c->lastf = reinterpret_cast<float *>(q + 1 * (plane + skew));
            c->status = reinterpret_cast<uint32_t *>(q + 2 * (plane + skew));
            c->reset_bytes = 2 * (plane + skew) + 256;
            q += 256;
            c->integ0 = reinterpret_cast<float *>(q + 2 * (plane + skew));
            c->dt0 = reinterpret_cast<float *>(q + 3 * (plane + skew));
            c->bdt0 = reinterpret_cast<float *>(q + 4 * (plane + skew));
        }
        c->continuous = p.pixel_mode == ADDER_MODE_CONTINUOUS;
        if (c->continuous) {
            const size_t cnt = c->n_pad * (c->max_depth + 1u);
            HIPCHK(c, dalloc(&c->cn_integ, cnt));
            HIPCHK(c, dalloc(&c->cn_dt, cnt));
            HIPCHK(c, dalloc(&c->cn_bdt, cnt));
            HIPCHK(c, dalloc(&c->cn_meta, cnt));
        }
        { int rc_ = alloc_scratch(c, c->continuous ? AdderHipCtx::kScratchCont
                                      : p.time_mode == ADDER_TIME_ABSOLUTE_T ? AdderHipCtx::kScratchLean : AdderHipCtx::kScratchLean8);
          if (rc_ != ADDER_OK) return rc_; }
        { int rc_ = alloc_batch_desc(c, 1024); if (rc_ != ADDER_OK) return rc_; }
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_result), sizeof(BatchResult), hipHostMallocDefault));
        memset(c->h_result, 0, sizeof(BatchResult));
        HIPCHK(c, hipEventCreateWithFlags(&c->reset_e, hipEventDisableTiming));
        HIPCHK(c, hipStreamCreateWithFlags(&c->cap_s, hipStreamNonBlocking));
        {
            // scan / offsets / expansion of chunk k share the chip with the frame kernel of chunk k+1: the short
            // scan must not queue behind a grid that fills every CU, or the expansion starts a whole kernel late
            int lo_prio = 0, hi_prio = 0;
            HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
            const char *pe = getenv("ADDER_HIP_S2_PRIORITY");
            const int prio = pe ? atoi(pe) : hi_prio;
            HIPCHK(c, hipStreamCreateWithPriority(&c->cap_s2, hipStreamNonBlocking, prio));
        }
        HIPCHK(c, hipEventCreateWithFlags(&c->cap_e1, hipEventDisableTiming));
        for (hipEvent_t &e : c->cap_e2) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (const char *ng = getenv("ADDER_HIP_NO_GRAPH")) {
            c->use_graph = atoi(ng) == 0;
            c->eager_two_streams = atoi(ng) == 2;
        }
        {
            // workgroups per CU of the lean frame kernel and of the expansion when they share the chip (0: full grids)
            const char *lb = getenv("ADDER_HIP_LEAN_BLOCKS_PER_CU"), *eb = getenv("ADDER_HIP_EXPAND_BLOCKS_PER_CU");
            // (full grids since the frame kernel runs 5 waves per SIMD and the streams carry cache hints: walking
            // grids of 4 + 3 workgroups per CU, the default until then, measured 2-3 % slower)
            c->lean_blocks_per_cu = lb ? (uint32_t)atoi(lb) : 0u;
            c->expand_blocks_per_cu = eb ? (uint32_t)atoi(eb) : 0u;
        }
        if (const char *e = getenv("ADDER_HIP_GEN_BLOCKS_PER_CU")) c->gen_blocks_per_cu = (uint32_t)atoi(e);
        if (const char *e = getenv("ADDER_HIP_GEN_EXPAND_BLOCKS_PER_CU")) c->gen_expand_blocks_per_cu = (uint32_t)atoi(e);
        if (const char *gc = getenv("ADDER_HIP_GRAPH_CANDIDATES")) c->graph_candidates = (uint32_t)std::max(1, atoi(gc));
        if (const char *fl = getenv("ADDER_HIP_FRAMES_PER_LAUNCH"))
            c->frames_per_launch = (uint32_t)std::max(1, std::min<int>(atoi(fl), kMaxFramesPerLaunch));
        HIPCHK(c, dalloc(&c->d_rec_total, 1));
        HIPCHK(c, dalloc(&c->d_side_words, 4));  // ([2]: a constant 0 -- the wire scatter's destination of a lone frame)
        HIPCHK(c, hipMemsetAsync(c->d_side_words, 0, 4 * sizeof(uint64_t), c->stream));
        HIPCHK(c, dalloc(&c->d_chunks, c->num_chunks + 1));
        { int rc_ = init_state(c); if (rc_ != ADDER_OK) return rc_; }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return ADDER_OK;
    };
    rc = setup();
    if (rc != ADDER_OK) {
        g_create_error = c->err;
        free_ctx(c);
        return rc;
    }
    *out = c;
    return ADDER_OK;
}

static void ring_timeline_report();  // (diagnostics: ADDER_HIP_RING_TIMING=2, below)
extern "C" void adder_hip_destroy(AdderHipCtx *ctx) {
    ring_timeline_report(); free_ctx(ctx); }

extern "C" const char *adder_hip_last_error(const AdderHipCtx *ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" int adder_hip_set_crf_parameters(AdderHipCtx *c, uint8_t c_thresh_max, uint8_t c_increase_velocity) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c_increase_velocity == 0) return fail(c, ADDER_E_BAD_PARAMS, "c_increase_velocity must be >= 1");
    c->p.c_thresh_max = c_thresh_max;
    c->p.c_increase_velocity = c_increase_velocity;
    return ADDER_OK;
}

extern "C" int adder_hip_reset_c_thresh(AdderHipCtx *c, uint8_t baseline) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->pending || c->f_submitted != c->f_collected || c->submitted != c->collected)
        return fail(c, ADDER_E_BAD_PARAMS, "work is in flight on this context (finish / collect it first)");
    // every pixel: c_thresh = baseline, c_increase_counter = 0 (video.rs:1247-1250,1283-1286); the pair
    // is uniform across the plane, so it lives in the context
    c->c_thresh = baseline;
    c->c_counter = 0;
    c->perpx = false;  // uniform again; the next frame of a feature / ROI batch re-creates the planes from the pair
    if (c->sparse_mode && c->cth_px) {  // sparse contexts keep the pair per unit: every unit gets (baseline, 0)
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipMemsetAsync(c->cth_px, baseline, c->n_pad, c->stream));
        HIPCHK(c, hipMemsetAsync(c->cctr_px, 0, c->n_pad, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return ADDER_OK;
}

// adder_hip_reset does not wait for its memsets: host-side readers of the state do
static int settle_reset(AdderHipCtx *c) {
    if (c->reset_pending) {
        HIPCHK(c, hipEventSynchronize(c->reset_e));
        c->reset_pending = false;
    }
    return ADDER_OK;
}

// ---- feature-driven rate control + ROI ----
static bool feature_path(const AdderHipCtx *c) { return c->feat_detect || c->roi_on; }
static bool feature_needs_perpx(const AdderHipCtx *c) {
    return c->roi_on || (c->feat_detect && c->feat_adjust && c->feature_c_radius > 0);
}
static bool is_band(const AdderHipCtx *c) { return c->p.row_begin != 0 || c->p.row_end != c->p.height; }
// A row-band context (multi-GPU) in feature mode cannot finish a frame on its own: the corner test reads the three
// rows beyond its band and the reset squares of its neighbours' new features reach into its rows.  It integrates ONE
// frame per call and leaves the feature step to adder_hip_feature_detect, between the halo exchanges (include/adder_hip.h).
static bool band_features(const AdderHipCtx *c) { return is_band(c) && (c->feat_detect || c->roi_on); }

extern "C" int adder_hip_update_detect_features(AdderHipCtx *c, int detect_features, int feature_rate_adjustment) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    if (detect_features && c->continuous) return fail(c, ADDER_E_BAD_PARAMS, "feature detection: FramePerfect contexts only");
    c->feat_detect = detect_features != 0;
    c->feat_adjust = feature_rate_adjustment != 0;
    return ADDER_OK;
}

extern "C" int adder_hip_set_feature_parameters(AdderHipCtx *c, uint8_t c_thresh_baseline, uint16_t feature_c_radius) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    c->c_thresh_baseline = c_thresh_baseline;
    c->feature_c_radius = feature_c_radius;
    return ADDER_OK;
}

extern "C" int adder_hip_update_roi(AdderHipCtx *c, int enable, uint16_t x0, uint16_t y0, uint16_t x1, uint16_t y1) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    if (enable && c->continuous) return fail(c, ADDER_E_BAD_PARAMS, "a region of interest: FramePerfect contexts only");
    c->roi_on = enable != 0;
    c->roi[0] = x0;
    c->roi[1] = y0;
    c->roi[2] = x1;
    c->roi[3] = y1;
    return ADDER_OK;
}

extern "C" int adder_hip_feature_set(AdderHipCtx *c, uint8_t *dst) {
    if (!c || !dst) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    const size_t n = (size_t)c->rows * c->p.width;
    if (!c->fset) {
        memset(dst, 0, n);
        return ADDER_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = settle_reset(c); if (rc_ != ADDER_OK) return rc_; }
    HIPCHK(c, hipMemcpy(dst, c->fset, n, hipMemcpyDeviceToHost));
    return ADDER_OK;
}

extern "C" int adder_hip_c_thresh_plane(AdderHipCtx *c, uint8_t *dst) {
    if (!c || !dst) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    if (!c->perpx) {
        memset(dst, c->c_thresh, c->n_units);
        return ADDER_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    { int rc_ = settle_reset(c); if (rc_ != ADDER_OK) return rc_; }
    HIPCHK(c, hipMemcpy(dst, c->cth_px, c->n_units, hipMemcpyDeviceToHost));
    return ADDER_OK;
}

extern "C" uint32_t adder_hip_last_new_features(const AdderHipCtx *c) { return c ? c->last_new_features : 0u; }

// planes of the feature path, on first use
static int prepare_feature_set(AdderHipCtx *c, hipStream_t s) {
    if (!c->fset) {
        HIPCHK(c, dalloc(&c->fset, (size_t)c->rows * c->p.width));
        HIPCHK(c, hipMemsetAsync(c->fset, 0, (size_t)c->rows * c->p.width, s));
    }
    if (!c->d_feat_counters) HIPCHK(c, dalloc(&c->d_feat_counters, 4));
    HIPCHK(c, hipMemsetAsync(c->d_feat_counters, 0, 4 * sizeof(uint32_t), s));
    return ADDER_OK;
}
// the per-unit (c_thresh, c_increase_counter) pair starts from the uniform one
static int prepare_per_unit_c_thresh(AdderHipCtx *c, hipStream_t s) {
    if (feature_needs_perpx(c) && !c->perpx) {
        if (!c->cth_px) {
            HIPCHK(c, dalloc(&c->cth_px, c->n_pad));
            HIPCHK(c, dalloc(&c->cctr_px, c->n_pad));
        }
        HIPCHK(c, hipMemsetAsync(c->cth_px, c->c_thresh, c->n_pad, s));
        HIPCHK(c, hipMemsetAsync(c->cctr_px, c->c_counter, c->n_pad, s));
        c->perpx = true;
    }
    return ADDER_OK;
}

static FeatureArgs feature_args(const AdderHipCtx *c);
// ---- feature mode across row bands (SURVEY 8(f)4 "needs a halo exchange between row bands") ----
extern "C" size_t adder_hip_feature_halo_bytes(const AdderHipCtx *c) { return c ? halo_bytes(c) : 0; }

extern "C" int adder_hip_feature_halo_export(AdderHipCtx *c, uint8_t *d_top, uint8_t *d_bottom, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    if (!c->running) return fail(c, ADDER_E_BAD_PARAMS, "no running-intensities plane yet (integrate a frame in feature mode first)");
    if (c->rows < kFeatureHalo) return fail(c, ADDER_E_BAD_PARAMS, "a band in feature mode needs at least %u rows", kFeatureHalo);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t hb = halo_bytes(c), rowlen = (size_t)c->p.width * c->p.channels;
    if (d_top) HIPCHK(c, hipMemcpyAsync(d_top, c->running, hb, hipMemcpyDeviceToDevice, s));
    if (d_bottom) HIPCHK(c, hipMemcpyAsync(d_bottom, c->running + (size_t)(c->rows - kFeatureHalo) * rowlen, hb, hipMemcpyDeviceToDevice, s));
    return ADDER_OK;
}

extern "C" int adder_hip_feature_halo_import(AdderHipCtx *c, const uint8_t *d_above, const uint8_t *d_below, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    if (!c->running) return fail(c, ADDER_E_BAD_PARAMS, "no running-intensities plane yet (integrate a frame in feature mode first)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t hb = halo_bytes(c), rowlen = (size_t)c->p.width * c->p.channels;
    if (d_above) HIPCHK(c, hipMemcpyAsync(c->running_base, d_above, hb, hipMemcpyDeviceToDevice, s));
    if (d_below) HIPCHK(c, hipMemcpyAsync(c->running + (size_t)c->rows * rowlen, d_below, hb, hipMemcpyDeviceToDevice, s));
    return ADDER_OK;
}

extern "C" int adder_hip_feature_detect(AdderHipCtx *c, uint32_t *d_new_xy, uint32_t cap, uint32_t *n_new, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (n_new) *n_new = 0;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    if (!c->band_frame_pending) return fail(c, ADDER_E_BAD_PARAMS, "no frame awaits its feature step");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    FeatureArgs fa = feature_args(c);
    fa.new_xy = d_new_xy;
    fa.new_cap = d_new_xy ? cap : 0u;
    HIPCHK(c, adder_launch_features(c->d_batch, 0u, &fa, s));
    uint32_t cnt[2] = {0u, 0u};
    HIPCHK(c, hipMemcpyAsync(cnt, c->d_feat_counters, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    c->band_frame_pending = false;
    c->last_new_features = cnt[0];
    if (n_new) *n_new = cnt[0];
    if (d_new_xy && cnt[0] > cap) return fail(c, ADDER_E_OUT_CAPACITY, "feature list too small: %u new features", cnt[0]);
    return ADDER_OK;
}

extern "C" int adder_hip_feature_apply(AdderHipCtx *c, const uint32_t *d_xy, uint32_t n, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending");
    if (!d_xy && n) return fail(c, ADDER_E_BAD_PARAMS, "null feature list");
    if (!c->perpx || n == 0) return ADDER_OK;  // no per-unit thresholds to reset (detection without rate adjustment)
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const FeatureArgs fa = feature_args(c);
    HIPCHK(c, adder_launch_feature_apply(c->d_batch, &fa, d_xy, n, s));
    return ADDER_OK;
}

extern "C" int adder_hip_set_delta_t_max(AdderHipCtx *c, uint32_t dtm) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (dtm < c->p.ref_time || dtm % c->p.ref_time != 0)
        return fail(c, ADDER_E_BAD_PARAMS, "delta_t_max must be a multiple of ref_time and >= ref_time");
    c->p.delta_t_max = dtm;
    c->dtm_max_seen = std::max(c->dtm_max_seen, dtm);
    return ADDER_OK;
}

extern "C" int adder_hip_set_time_mode(AdderHipCtx *c, uint8_t time_mode) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (time_mode > ADDER_TIME_MIXED) return fail(c, ADDER_E_BAD_PARAMS, "bad time_mode");
    if (c->frames_done != 0)
        return fail(c, ADDER_E_BAD_PARAMS, "time mode can only be set before the first frame is integrated");
    c->p.time_mode = time_mode;
    return ADDER_OK;
}

extern "C" uint32_t adder_hip_num_chunks(const AdderHipCtx *c) { return c ? c->num_chunks : 0; }

// The most events one frame can emit: the lean step at most 3 per unit (root event, Collapse filler, pop_top's
// event); the generic step its whole arena (<= max_depth levels) plus pop_top's event.
static bool lean_possible(const AdderHipCtx *c, float time_spanned) {
    return !c->generic_sticky && !c->perpx && !feature_needs_perpx(c) && c->p.multi_mode == ADDER_MULTI_COLLAPSE &&
           (float)c->p.delta_t_max <= time_spanned;
}
// The bounded Collapse step (adder_pixel.hpp cb_step): Collapse with delta_t_max > time_spanned, a uniform c_thresh, and
// every sum its prefix coordinates form an exact integer below 2^24 -- integer time_spanned, at most delta_t_max /
// time + 1 frames of 8-bit intensities before the pop.  Anything else takes the generic step.
static bool rr_possible(const AdderHipCtx *c, float T, bool pop_at_once_ok = false);
static bool cb_possible(const AdderHipCtx *c, float T) {
    if (c->p.multi_mode != ADDER_MULTI_COLLAPSE) return false;
    return rr_possible(c, T);
}
// (the bounded regime's conditions without the mode; pop_at_once_ok: delta_t_max <= time_spanned is fine too -- Mode Normal,
// where a new root is then popped in the frame it starts: adder_pixel.hpp kRrFlushPop)
static bool rr_possible(const AdderHipCtx *c, float T, bool pop_at_once_ok) {
    if (c->continuous || c->perpx || feature_needs_perpx(c)) return false;
    if (c->frac_time_seen) return false;
    static const bool off = [] { const char *e = getenv("ADDER_HIP_NO_CB"); return e && atoi(e) != 0; }();
    if (off) return false;
    const double dtm = (double)std::max(c->p.delta_t_max, c->dtm_max_seen);
    if (!((float)c->p.delta_t_max > T) && !pop_at_once_ok) return false;
    if (!(T >= 1.0f) || T != (float)(uint32_t)T || T > 65536.0f) return false;
    if (dtm + 2.0 * T >= 8388608.0) return false;
    if ((dtm / T + 3.0) * 255.0 >= 8388608.0) return false;
    return true;
}
static size_t worst_case_events_per_frame(const AdderHipCtx *c, float time_spanned) {
    if (c->continuous) return (size_t)c->n_units * (c->max_depth + 3u);
    return (size_t)c->n_units * (lean_possible(c, time_spanned) ? 3u : c->max_depth + 1u);
}
extern "C" size_t adder_hip_max_events_per_frame(const AdderHipCtx *c) {
    return c ? worst_case_events_per_frame(c, (float)c->p.ref_time) : 0;
}

static int status_to_code(AdderHipCtx *c, uint32_t st) {
    if (st == 0) return ADDER_OK;
    c->poisoned = true;
    if (st & kStatusDepth)
        return fail(c, ADDER_E_ARENA_DEPTH, "a pixel needed more than max_depth=%u stored nodes", c->max_depth);
    if (st & kStatusWire)
        return fail(c, ADDER_E_BAD_PARAMS, "wire serialisation: an event without a channel on a multi-channel plane");
    if (st & kStatusSparse) return fail(c, ADDER_E_BAD_PARAMS, "a sparse step names a pixel outside the plane / row band");
    if (st & kStatusScratch) return fail(c, ADDER_E_HIP, "internal error: a segment's record log exceeded its bound");
    if (st & kStatusLeanRuns)
        return fail(c, ADDER_E_HIP, "internal error: the lean-runs kernel met state planes outside its regime (popped_dtm != (base_val != 0))");
    return fail(c, ADDER_E_OUT_CAPACITY, "event buffer too small");
}

// (Re)allocates the compaction scratch ring for one kind of record.  Fixed-slot kinds: the lean step parks at most
// one 12-byte record per unit and frame (kLeanParkBytes per segment), a Continuous context stages max_depth + 3 events
// per unit.  Log kinds (per-event records): one region per segment and CHUNK holding the hard bound of what the
// segment can emit over the chunk (log_capacity: 2 or 3 events per unit and frame + one arena), instead of a slot per
// frame that would have to hold a whole arena per unit -- 18 instead of 144 bytes per unit and frame at 64 frames.
// Chunk = frames per scan launch: as many as the budget allows (ring_chunks chunks are in flight), at most kMaxChunk.
// The budget is a third of what the device has free, at most 96 GiB (a 1080p plane takes 3 - 5 GiB at 64-frame
// chunks; a 4K RGB one -- 12 times the units -- needs 88 GiB for 64-frame chunks of per-event logs and fell to 48-frame
// chunks under round 4's 64 GiB: a 64-frame batch then paid the state's round trip twice).
static size_t scratch_bytes_per_chunk(const AdderHipCtx *c, AdderHipCtx::ScratchKind kind, uint32_t chunk) {
    switch (kind) {
        case AdderHipCtx::kScratchLean: return (size_t)c->num_waves * chunk * kLeanParkBytes;
        case AdderHipCtx::kScratchLean8: return (size_t)c->num_waves * chunk * kLeanPark8Bytes;
        case AdderHipCtx::kScratchCont:
            return (size_t)c->num_waves * chunk * (kWaveUnits + kWaveUnits * (c->max_depth + 3u) * kGenRecBytes);
        case AdderHipCtx::kScratchLog2: return (size_t)c->num_waves * log_capacity(chunk, c->max_depth, true) * kGenRecBytes;
        case AdderHipCtx::kScratchLog3: return (size_t)c->num_waves * log_capacity(chunk, c->max_depth, false) * kGenRecBytes;
        default: return 0;
    }
}
static uint32_t park_group_shift_wanted() {
    if (const char *e = getenv("ADDER_HIP_PARK_GROUP_SHIFT")) {  // 0, or >= log2(segments per expansion wave)
        const int sh = atoi(e);
        static_assert(16 % ADDER_EXPAND_SEGS == 0, "a group (and a rotation group of 16 segments) must hold whole expansion waves");
        return sh <= 0 ? 0u : (uint32_t)std::max(sh, 4);
    }
    return 4u;
}
static int alloc_scratch(AdderHipCtx *c, AdderHipCtx::ScratchKind kind) {
    // (a Collapse context that has the general log can run the bounded step on it: no thrash between the two)
    if (c->park_ring && (c->scratch_kind == kind || (c->scratch_kind == AdderHipCtx::kScratchLog3 && kind == AdderHipCtx::kScratchLog2)))
        return ADDER_OK;
    for (auto &kv : c->graphs)  // they bake the chunking
        for (hipGraphExec_t e : kv.second.cand)
            if (e) c->retired_execs.push_back(e);
    c->graphs.clear();
    c->tune_pending = false;
    void *old[] = {c->park_ring, c->wtot_ring, c->wpref_ring, c->ftot_ring, c->wofs_ring, c->wcur};
    c->park_ring = nullptr;
    c->wtot_ring = c->wpref_ring = c->ftot_ring = c->wofs_ring = c->wcur = nullptr;
    c->park_bytes = 0;
    c->log_cap = 0;
    c->scratch_kind = AdderHipCtx::kScratchNone;
    for (void *p : old)
        if (p) HIPCHK(c, hipFree(p));
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::min<size_t>((size_t)96 << 30, free_b / 3);
    c->ring_chunks = 3;  // a chunk being stepped, one being scanned / expanded, one of slack between the two streams
    if (const char *e = getenv("ADDER_HIP_RING_CHUNKS")) c->ring_chunks = std::max(2, std::min(atoi(e), 4));
    // default: groups of 16 segments -- the 16 segments an expansion wave reads of one frame are contiguous; contexts
    // measured 1.515 - 1.525 ms per step (a few 1.57 - 1.58) against 1.55 / 1.62 (two modes) for the rotated
    // segment-major layout (profiles/r03_ctx_spread.txt).  (A group must divide the segment count, which is padded to a
    // multiple of 16: larger requests fall back to 16.)
    c->park_group_shift = park_group_shift_wanted();
    uint32_t ch = kMaxChunk;
    while (ch > 1u && c->ring_chunks * (scratch_bytes_per_chunk(c, kind, ch) + (size_t)c->num_waves * ch * 12u) > budget) --ch;
    c->chunk = ch;
    if (const char *e = getenv("ADDER_HIP_CHUNK")) c->chunk = std::max(1, std::min<int>(atoi(e), kMaxChunk));
    c->slots = c->ring_chunks * c->chunk;
    HIPCHK(c, dalloc(&c->park_ring, (size_t)c->ring_chunks * scratch_bytes_per_chunk(c, kind, c->chunk)));
    HIPCHK(c, dalloc(&c->wtot_ring, (size_t)c->slots * c->num_waves));
    HIPCHK(c, dalloc(&c->wpref_ring, (size_t)c->slots * c->num_waves));
    // events, then parked records per frame, then the scan's tile sums [slot][tile][2] (planes scanned in tiles)
    HIPCHK(c, dalloc(&c->ftot_ring, 2 * (size_t)c->slots + (size_t)c->slots * ((c->num_waves + kScanTileWaves - 1) / kScanTileWaves) * 2));
    if (kind != AdderHipCtx::kScratchCont) {  // (the lean records of blocked batches go to logs too: lean_log_cap)
        HIPCHK(c, dalloc(&c->wofs_ring, (size_t)c->slots * c->num_waves));
        HIPCHK(c, dalloc(&c->wcur, (size_t)c->ring_chunks * c->num_waves));
    }
    if (kind == AdderHipCtx::kScratchLog2 || kind == AdderHipCtx::kScratchLog3) {
        c->log_cap = log_capacity(c->chunk, c->max_depth, kind == AdderHipCtx::kScratchLog2);
    } else {
        c->park_bytes = (uint32_t)(scratch_bytes_per_chunk(c, kind, 1u) / c->num_waves);
    }
    c->scratch_kind = kind;
    if (getenv("ADDER_HIP_DEBUG_ADDRS"))
        fprintf(stderr, "[adder_hip] slab %p park %p wtot %p wpref %p ftot %p chunk %u\n", (void *)c->state_slab, (void *)c->park_ring,
                (void *)c->wtot_ring, (void *)c->wpref_ring, (void *)c->ftot_ring, c->chunk);
    return ADDER_OK;
}

// Levels >= 1 of the arena: only the generic kernels use them.
static int alloc_deep_planes(AdderHipCtx *c) {
    if (c->dv_integ) return ADDER_OK;
    const size_t deep = std::max<uint32_t>(c->max_depth, 2u) - 1u;
    HIPCHK(c, dalloc(&c->dv_integ, c->n_pad * deep));
    HIPCHK(c, dalloc(&c->dv_dt, c->n_pad * deep));
    HIPCHK(c, dalloc(&c->dv_bdt, c->n_pad * deep));
    HIPCHK(c, dalloc(&c->dv_bd, c->n_pad * deep));
    return ADDER_OK;
}

// The launch sequence of a batch.  Frames are handled in chunks of c->chunk; per chunk
//   * K1 launches back to back, each stepping up to frames_per_launch consecutive frames
//     (frame f+1 only needs frame f's pixel state),
//   * one scan launch (a block per frame) + the frame_offsets chain,
//   * the expansion of the chunk's parked records.
// With a second stream (graph capture), scan/offsets/expand of chunk k go to s2 behind the chunk's last
// K1, and the K1s of chunk k+2 wait for them (the scratch ring holds two chunks).  Running the
// expansion inside K1's grid (round 1) no longer pays: with the lean step both kernels are bound by the
// memory system, and a resident K1 grid leaves no wave slots for a concurrent kernel anyway.
static uint32_t launch_depth(const AdderHipCtx *c) { return c->running_enabled ? 1u : c->frames_per_launch; }
static Lean1wArgs lean1w_args(const AdderHipCtx *c) {
    return Lean1wArgs{c->hdr, c->integ0, c->dt0, c->bdt0, c->lastf, c->n_units, c->num_waves};
}

// Feature path: frame f+1's contrast thresholds depend on the features frame f's EVENTS reveal, so the whole
// pipeline runs frame by frame: step, scan, offsets, expansion, features.
static FeatureArgs feature_args(const AdderHipCtx *c) {
    FeatureArgs fa{};
    fa.fset = c->fset;
    fa.counters = c->d_feat_counters;
    fa.chunk_rows = c->p.chunk_rows;
    fa.detect = c->feat_detect ? 1u : 0u;
    fa.radius = (c->feat_detect && c->feat_adjust) ? c->feature_c_radius : 0u;
    fa.low = std::min<uint32_t>(c->c_thresh_baseline, 2u);
    fa.roi_on = c->roi_on ? 1u : 0u;
    fa.rx0 = c->roi[0];
    fa.ry0 = c->roi[1];
    fa.rx1 = c->roi[2];
    fa.ry1 = c->roi[3];
    fa.plane_h = c->p.height;
    fa.new_xy = nullptr;
    fa.new_cap = 0;
    return fa;
}

static int launch_feature_loop(AdderHipCtx *c, uint32_t num_frames, uint32_t variant, hipStream_t s) {
    const Lean1wArgs wide = lean1w_args(c);
    const FeatureArgs fa = feature_args(c);
    const bool band = band_features(c);  // the feature step waits for the halo exchange: adder_hip_feature_detect
    for (uint32_t f = 0; f < num_frames; ++f) {
        HIPCHK(c, adder_launch_frame(c->d_batch, f, 1u, variant, c->num_waves, 0u, s, &wide));
        HIPCHK(c, adder_launch_scan(c->d_batch, f, 1u, c->num_waves, s));
        HIPCHK(c, adder_launch_offsets(c->d_batch, f, 1u, s));
        HIPCHK(c, adder_launch_expand(c->d_batch, f, 1u, c->num_waves, variant, 0u, s));
        if (!band) HIPCHK(c, adder_launch_features(c->d_batch, f, &fa, s));
    }
    HIPCHK(c, adder_launch_publish(c->d_batch, num_frames, c->h_result, s));
    return ADDER_OK;
}

// The integer-state kernels (lean runs, run records) keep their whole state in the header, delta_t and last_fired_t
// planes; the other level-0 planes and the levels are derived from it (divisions, and for run records a store per level).
// A launch that is followed by another launch of the same batch leaves them stale (variant bit 2048): only the batch's last
// launch brings the planes to the resident form every other kernel, a rollback or the next batch reads.  (Run records:
// 30 us of a 160 us launch were this epilogue.)
static uint32_t lazy_state_bit(const AdderHipCtx *c, uint32_t variant, bool more_launches) {
    static const bool off = env_flag("ADDER_HIP_NO_LAZY_STATE");
    return (more_launches && (variant & (256u | 512u)) && !c->running_enabled && !off) ? 2048u : 0u;
}

static int launch_frame_loop(AdderHipCtx *c, uint32_t num_frames, uint32_t variant, hipStream_t s, hipStream_t s2,
                             bool timing) {
    // temporal blocking is off while the running-intensities side plane is wanted (per-frame
    // semantics).  Generic batches block too: levels >= 1 stay in HBM, but they are private
    // to their unit, so the lane's own program order keeps them consistent across frames.
    const uint32_t depth = launch_depth(c);
    const Lean1wArgs wide = lean1w_args(c);
    // With a second stream the expansion of chunk k runs beside the frame kernel of chunk k+1.  The frame kernel is
    // bound by instruction issue, the expansion by memory, and two grids that each fill the chip would simply run one
    // after the other (measured: the scan queued behind the resident frame kernel for a whole kernel time): both are
    // launched with a few workgroups per CU that walk their work, so that both are resident on every CU.
    const bool share = s2 != nullptr && num_frames > c->chunk;
    const bool gen = (variant & 4u) != 0u;
    const uint32_t lean_cap = share ? (gen ? c->gen_blocks_per_cu : c->lean_blocks_per_cu) * c->num_cus : 0u;
    const uint32_t expand_cap = share ? (gen ? c->gen_expand_blocks_per_cu : c->expand_blocks_per_cu) * c->num_cus : 0u;
    uint32_t k = 0;
    for (uint32_t f0 = 0; f0 < num_frames; f0 += c->chunk, ++k) {
        const uint32_t nf = std::min(c->chunk, num_frames - f0);
        if (s2 && k >= c->ring_chunks)  // scratch reuse: the expansion of the chunk that held these slots
            HIPCHK(c, hipStreamWaitEvent(s, c->cap_e2[(k - c->ring_chunks) % 5u], 0));
        const bool per_chunk = timing && c->launch_timing == 2;
        if (per_chunk) HIPCHK(c, hipEventRecord(c->launch_events[2 * c->timed_pairs], s));
        for (uint32_t f = f0; f < f0 + nf; f += depth) {
            const uint32_t nb = std::min(depth, f0 + nf - f);
            if (timing && !per_chunk) HIPCHK(c, hipEventRecord(c->launch_events[2 * c->timed_pairs], s));
            HIPCHK(c, adder_launch_frame(c->d_batch, f, nb, variant | lazy_state_bit(c, variant, f + nb < num_frames), c->num_waves, lean_cap, s, &wide));
            if (timing) {  // the pair brackets the frame kernel (K1) only
                if (!per_chunk) {
                    HIPCHK(c, hipEventRecord(c->launch_events[2 * c->timed_pairs + 1], s));
                    c->timed_pairs += 1;
                }
                c->timed_launches += 1;
                c->timed_frames += nb;
            }
        }
        if (per_chunk) {  // ... or the chunk's whole run of them (no event packets between the launches)
            HIPCHK(c, hipEventRecord(c->launch_events[2 * c->timed_pairs + 1], s));
            c->timed_pairs += 1;
        }
        // (scan + offsets on a third stream of their own -- they depend on the chunk's frame kernels only -- was
        // tried: they then run at once instead of behind the previous expansion, and the expansions, now back to
        // back, take as much longer as was gained: the two big kernels compete for the same thing; in-kernel
        // timestamps, adder_hip_debug_timeline)
        hipStream_t t = s;
        if (s2) {
            HIPCHK(c, hipEventRecord(c->cap_e1, s));
            HIPCHK(c, hipStreamWaitEvent(s2, c->cap_e1, 0));
            t = s2;
        }
        if (timing) HIPCHK(c, hipEventRecord(c->post_events[2 * c->timed_posts], t));
        // (a lean-runs batch that hands its records out: the scan also leaves the segments' record prefix, for the packing)
        // (the integer-state kernels' and the bounded Collapse kernel's batches: the scan's blocks chain the frame offsets
        // themselves, adder_scan_kernel CHAIN -- a launch less per chunk; those frame kernels zero the entries, chain_zero)
        static const bool no_chain = env_flag("ADDER_HIP_NO_SCAN_CHAIN");
        // (adder_launch_frame's order: Continuous, run records, constant runs, bounded Collapse, generic, packed / plain lean runs)
        const bool chain_kernel = (variant & 8u) == 0u && ((variant & (512u | 128u | 32u)) != 0u ||
                                                           ((variant & 4u) == 0u && (variant & (4096u | 256u)) != 0u));
        const bool chain = chain_kernel && num_frames != 1u && !c->records_only && !no_chain;
        HIPCHK(c, adder_launch_scan(c->d_batch, f0, nf, c->num_waves, t, num_frames == 1u ? 1u : 0u,
                                    (c->records_only && (variant & 256u)) ? 1u : 0u, chain ? 1u : 0u));
        if (num_frames != 1u && !chain) HIPCHK(c, adder_launch_offsets(c->d_batch, f0, nf, t));  // (one frame: done by the scan)
        if (!c->records_only) HIPCHK(c, adder_launch_expand(c->d_batch, f0, nf, c->num_waves, variant, expand_cap, t, c->h_batch));
        if (timing) {
            HIPCHK(c, hipEventRecord(c->post_events[2 * c->timed_posts + 1], t));
            c->timed_posts += 1;
        }
        if (s2) HIPCHK(c, hipEventRecord(c->cap_e2[k % 5u], s2));
    }
    if (s2 && k) HIPCHK(c, hipStreamWaitEvent(s, c->cap_e2[(k - 1) % 5u], 0));  // join (s2 is in-order)
    HIPCHK(c, adder_launch_publish(c->d_batch, num_frames, c->h_result, s));
    return ADDER_OK;
}

static int instantiate_graph(AdderHipCtx *c, uint32_t num_frames, uint32_t variant, bool one_stream, hipGraphExec_t *out) {
    hipGraph_t graph = nullptr;
    HIPCHK(c, hipStreamBeginCapture(c->cap_s, hipStreamCaptureModeThreadLocal));
    // per-event-record batches (generic / bounded Collapse kernels) are captured on ONE stream: their frame kernel is
    // bound by instruction issue at full occupancy and leaves the expansion no room to run beside it -- two branches
    // measured 10.5 us per 1080p frame against 10.0 in sequence (walking grids 11.6 - 14.2)
    static const bool gen_two = [] { const char *e = getenv("ADDER_HIP_GEN_TWO_STREAMS"); return e && atoi(e) != 0; }();
    hipStream_t s2 = (one_stream || ((variant & 4u) && !gen_two)) ? nullptr : c->cap_s2;
    int rc = launch_frame_loop(c, num_frames, variant, c->cap_s, s2, false);
    hipError_t e = hipStreamEndCapture(c->cap_s, &graph);
    if (rc != ADDER_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    HIPCHK(c, e);
    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIPCHK(c, e);
    return ADDER_OK;
}

constexpr int kTuneRunsPerCandidate = 2;  // the first run of an exec also uploads it
constexpr int kTuneRecheckRuns = 2;

static int get_graph(AdderHipCtx *c, uint32_t num_frames, uint32_t variant, hipGraphExec_t *out) {
    // everything the captured launch sequence depends on
    // (running_enabled and the lazy-state switch decide the launches' lazy bits, lazy_state_bit)
    // (graph_slot: a frame slot of the per-frame ring has its own description, so its own graphs)
    // (bits: frames 0-23, variant 24-39, launch depth 40-47, running 48, frame slot 49-)
    const uint64_t key = (uint64_t)(num_frames & 0xffffffu) | ((uint64_t)(variant & 0xffffu) << 24) | ((uint64_t)(launch_depth(c) & 0xffu) << 40) |
                         ((uint64_t)(c->running_enabled ? 1u : 0u) << 48) | ((uint64_t)c->graph_slot << 49);
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        if (c->graphs.size() >= 24) {  // keep the cache small (a ring of four slots holds a graph per slot and kernel choice)
            for (hipGraphExec_t e : c->graphs.begin()->second.cand)
                if (e) c->retired_execs.push_back(e);
            c->graphs.erase(c->graphs.begin());
        }
        it = c->graphs.emplace(key, AdderHipCtx::GraphTune{}).first;
    }
    AdderHipCtx::GraphTune &g = it->second;
    c->tune_pending = false;
    if (g.chosen >= 0) {
        *out = g.cand[g.chosen];
        return ADDER_OK;
    }
    // still choosing: the newest candidate until it has had its runs, then one more candidate
    // (a batch of a single chunk has no second branch to overlap: one candidate is all it needs)
    const uint32_t want = (num_frames > c->chunk && !(variant & 4u)) ? std::max(1u, c->graph_candidates) : 1u;
    int use = (int)g.cand.size() - 1;
    if (use > 0 && g.cand.size() >= want && g.runs[use] >= kTuneRunsPerCandidate && g.recheck < kTuneRecheckRuns) {
        // every candidate has had its runs: the first one (one stream) ran on a cold chip -- it gets two more now
        g.last = 0;
        c->tune_key = key;
        c->tune_pending = true;
        *out = g.cand[0];
        return ADDER_OK;
    }
    if (use < 0 || g.runs[use] >= kTuneRunsPerCandidate) {
        if (g.cand.size() < want) {
            hipGraphExec_t exec = nullptr;
            // the first candidate runs the chunks' kernels one after the other, the others on two branches (the frame
            // kernel of chunk k + 1 beside the scan / expansion of chunk k): with both big kernels bound by instruction
            // issue the branches only get in each other's way on a full 1080p plane (1.335 against 1.284 ms per step,
            // medians of six processes each), on small or quiet planes they hide the short kernels' launch gaps -- the
            // measured batch times decide
            int rc = instantiate_graph(c, num_frames, variant, g.cand.empty(), &exec);
            if (rc != ADDER_OK) return rc;
            g.cand.push_back(exec);
            g.ms.push_back(1e30f);
            g.runs.push_back(0);
            use = (int)g.cand.size() - 1;
        }
    }
    g.last = use;
    c->tune_key = key;
    c->tune_pending = true;
    *out = g.cand[use];
    return ADDER_OK;
}

// the batch of a graph that is still being chosen has finished in `ms`
static void graph_tune_report(AdderHipCtx *c, float ms) {
    if (!c->tune_pending) return;
    c->tune_pending = false;
    auto it = c->graphs.find(c->tune_key);
    if (it == c->graphs.end()) return;
    AdderHipCtx::GraphTune &g = it->second;
    if (g.last < 0 || g.chosen >= 0) return;
    g.runs[g.last] += 1;
    if (g.runs[g.last] > 1 || kTuneRunsPerCandidate == 1) g.ms[g.last] = std::min(g.ms[g.last], ms);
    const uint32_t want = (c->pending_frames > c->chunk && !((c->tune_key >> 24) & 4u)) ? std::max(1u, c->graph_candidates) : 1u;  // (get_graph's: variant bit 2 = generic)
    const bool all_ran = g.cand.size() >= want && g.runs.back() >= kTuneRunsPerCandidate;
    if (all_ran && g.cand.size() > 1 && g.last == 0 && g.runs[0] > kTuneRunsPerCandidate) g.recheck += 1;
    if (all_ran && (g.cand.size() == 1 || g.recheck >= kTuneRecheckRuns)) {
        // (the one-stream instance keeps the batch unless a two-branch one beats it by more than the runs' own scatter)
        int best = 0;
        for (int k = 1; k < (int)g.cand.size(); ++k)
            if (g.ms[k] < g.ms[best] * (best == 0 ? 0.985f : 1.0f)) best = k;
        for (int k = 0; k < (int)g.cand.size(); ++k)
            if (k != best) {
                c->retired_execs.push_back(g.cand[k]);
                g.cand[k] = nullptr;
            }
        g.chosen = best;
        if (getenv("ADDER_HIP_DEBUG_TUNE")) {
            fprintf(stderr, "[adder_hip] graph candidates (ms):");
            for (float v : g.ms) fprintf(stderr, " %.3f", v);
            fprintf(stderr, " -> %d\n", best);
        }
    }
}

// ---- undo copy of the state (see AdderHipCtx::Snapshot) ----
template <class T>
static hipError_t snap_copy(T **dst, const T *src, size_t count, hipStream_t s) {
    if (!*dst) {
        hipError_t e = dalloc(dst, count);
        if (e != hipSuccess) return e;
    }
    return hipMemcpyAsync(*dst, src, count * sizeof(T), hipMemcpyDeviceToDevice, s);
}
static int take_snapshot(AdderHipCtx *c, bool deep, hipStream_t s) {
    AdderHipCtx::Snapshot &n = c->snap;
    HIPCHK(c, snap_copy(&n.slab, c->state_slab, c->slab_bytes, s));  // the five level-0 planes + status, one copy
    if (c->continuous) {
        const size_t cnt = c->n_pad * (c->max_depth + 1u);
        HIPCHK(c, snap_copy(&n.cn_integ, c->cn_integ, cnt, s));
        HIPCHK(c, snap_copy(&n.cn_dt, c->cn_dt, cnt, s));
        HIPCHK(c, snap_copy(&n.cn_bdt, c->cn_bdt, cnt, s));
        HIPCHK(c, snap_copy(&n.cn_meta, c->cn_meta, cnt, s));
    }
    // levels >= 1: the batch's first launch stores every unit's LIVE levels into the copy (BatchArgs::snap_dv_*);
    // the buffers only have to exist
    n.deep = deep && c->dv_integ;
    if (n.deep && !n.dv_integ) {
        const size_t cnt = c->n_pad * (std::max<uint32_t>(c->max_depth, 2u) - 1u);
        HIPCHK(c, dalloc(&n.dv_integ, cnt));
        HIPCHK(c, dalloc(&n.dv_dt, cnt));
        HIPCHK(c, dalloc(&n.dv_bdt, cnt));
        HIPCHK(c, dalloc(&n.dv_bd, cnt));
    }
    n.perpx = c->perpx;
    if (c->perpx) {
        HIPCHK(c, snap_copy(&n.cth_px, c->cth_px, c->n_pad, s));
        HIPCHK(c, snap_copy(&n.cctr_px, c->cctr_px, c->n_pad, s));
    }
    if (c->fset) HIPCHK(c, snap_copy(&n.fset, c->fset, (size_t)c->rows * c->p.width, s));
    n.has_running = c->running != nullptr;
    if (c->running) HIPCHK(c, snap_copy(&n.running, c->running, c->n_pad, s));
    n.running_t = c->running_t;
    n.c_thresh = c->c_thresh;
    n.c_counter = c->c_counter;
    n.frames_done = c->frames_done;
    n.run_bound = c->run_bound;
    n.generic_sticky = c->generic_sticky;
    n.valid = true;
    return ADDER_OK;
}
static int restore_snapshot(AdderHipCtx *c, hipStream_t s) {
    AdderHipCtx::Snapshot &n = c->snap;
    HIPCHK(c, hipMemcpyAsync(c->state_slab, n.slab, c->slab_bytes, hipMemcpyDeviceToDevice, s));
    if (c->continuous) {
        const size_t cnt = c->n_pad * (c->max_depth + 1u);
        HIPCHK(c, hipMemcpyAsync(c->cn_integ, n.cn_integ, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->cn_dt, n.cn_dt, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->cn_bdt, n.cn_bdt, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->cn_meta, n.cn_meta, cnt * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    if (n.deep) {
        const size_t cnt = c->n_pad * (std::max<uint32_t>(c->max_depth, 2u) - 1u);
        HIPCHK(c, hipMemcpyAsync(c->dv_integ, n.dv_integ, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->dv_dt, n.dv_dt, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->dv_bdt, n.dv_bdt, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->dv_bd, n.dv_bd, cnt, hipMemcpyDeviceToDevice, s));
    }
    if (n.perpx) {
        HIPCHK(c, hipMemcpyAsync(c->cth_px, n.cth_px, c->n_pad, hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->cctr_px, n.cctr_px, c->n_pad, hipMemcpyDeviceToDevice, s));
    }
    if (c->fset && n.fset) HIPCHK(c, hipMemcpyAsync(c->fset, n.fset, (size_t)c->rows * c->p.width, hipMemcpyDeviceToDevice, s));
    c->perpx = n.perpx;
    if (c->running && n.has_running) HIPCHK(c, hipMemcpyAsync(c->running, n.running, c->n_pad, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->status, 0, sizeof(uint32_t), s));
    HIPCHK(c, hipStreamSynchronize(s));
    c->running_t = n.running_t;
    c->c_thresh = n.c_thresh;
    c->c_counter = n.c_counter;
    c->frames_done = n.frames_done;
    c->run_bound = n.run_bound;
    c->pending_reports = false;
    c->generic_sticky = n.generic_sticky;
    n.valid = false;
    return ADDER_OK;
}

// a row band in feature / ROI mode: refused before anything is queued (a caller's mistake, not a poisoned context)
static int band_precheck(AdderHipCtx *c, uint32_t num_frames) {
    if (!band_features(c)) return ADDER_OK;
    if (num_frames > 1)
        return fail(c, ADDER_E_BAD_PARAMS, "a row band in feature / ROI mode integrates one frame per call (the halo "
                    "exchange and adder_hip_feature_detect come between the frames)");
    if (c->band_frame_pending)
        return fail(c, ADDER_E_BAD_PARAMS, "the previous frame's feature step is missing (adder_hip_feature_detect)");
    return ADDER_OK;
}

// the lean-runs expansion's tables for this time step (cr_valid: one time step per reset, so once per stream)
static int ensure_lr_tab(AdderHipCtx *c, float time_spanned, hipStream_t stream) {
    if (c->d_lr_tab && c->lr_tab_time == time_spanned) return ADDER_OK;
    std::vector<uint32_t> tab(kLrTabWords);
    lr_build_tab(tab.data(), time_spanned);
    if (!c->d_lr_tab) HIPCHK(c, dalloc(&c->d_lr_tab, tab.size()));
    HIPCHK(c, hipStreamSynchronize(stream));  // (an expansion of the batch before may still read the old one)
    HIPCHK(c, hipMemcpy(c->d_lr_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->lr_tab_time = time_spanned;
    return ADDER_OK;
}

// Queues `num_frames` frames on `stream`.
static int enqueue_frames(AdderHipCtx *c, const uint8_t *d_frames, uint32_t num_frames, float time_spanned,
                          AdderEvent *d_out, size_t out_cap, uint64_t *d_offsets, hipStream_t stream) {
    // Pixels deeper than one fired level cannot occur when Collapse pops the root as soon as
    // it has accumulated once (delta_t_max <= time_spanned): then the lean kernel (adder_pixel.hpp
    // lean_step) runs.  Once a generic batch has run, pixels may hold deeper arenas (or a root that
    // the lean step's "time_spanned >= delta_t_max" folding does not describe), so the choice is sticky
    // until adder_hip_reset: update_quality_manual can lower delta_t_max mid-stream (video.rs:1264-1287).
    if (c->reset_pending) {  // adder_hip_reset queued its memsets on the context's stream
        HIPCHK(c, hipStreamWaitEvent(stream, c->reset_e, 0));
        c->reset_pending = false;
    }
    if (c->sparse_mode)
        return fail(c, ADDER_E_BAD_PARAMS, "dense frames after sparse steps: the pixels' c_thresh and running_t have "
                    "diverged (pass every pixel as a sparse step, or adder_hip_reset)");
    const bool collapse = c->p.multi_mode == ADDER_MULTI_COLLAPSE;
    const bool sticky_before = c->generic_sticky;
    const bool fpath = feature_path(c);

    const bool generic = !c->continuous && (c->generic_sticky || c->perpx || feature_needs_perpx(c) ||
                                            !(collapse && (float)c->p.delta_t_max <= time_spanned));
    if (fpath) {  // the corner test reads the running intensities (video.rs:736-744)
        c->running_enabled = true;
        int rc_ = prepare_feature_set(c, stream);
        if (rc_ != ADDER_OK) return rc_;
    }
    const bool cb = generic && cb_possible(c, time_spanned);  // the bounded Collapse step instead of the generic one
    if (!(time_spanned >= 1.0f) || time_spanned != (float)(uint32_t)time_spanned) c->frac_time_seen = true;
    // constant runs: c_thresh is 0 now and cannot grow (c_thresh_max 0: crf 0), the time step is the one of every batch
    // since the reset.  Once lost, the property stays lost until adder_hip_reset (a rolled-back batch included).
    if (c->c_thresh != 0 || c->p.c_thresh_max != 0 || c->frac_time_seen || fpath || c->perpx ||
        (c->cr_time != 0.0f && c->cr_time != time_spanned))
        c->cr_valid = false;
    c->cr_time = time_spanned;
    const bool cr_off = env_flag("ADDER_HIP_NO_CR");  // (read per batch: the tests switch kernels inside one process)
    const bool cr = cb && c->cr_valid && !cr_off;  // ... then only the roots are stepped (adder_cr_kernel)
    // how long a run can be by now: frames since the reset in AbsoluteT (last_fired_t / T is an integer of that size), in DeltaT
    // the bound the kernels' own reports keep down (AdderHipCtx::run_bound)
    // (a batch that hands its records to the multi-GPU gather takes the bound every rank shares -- the frames since the reset:
    // the kernels' reports follow each band's own content, a static band would leave the integer-state kernel where a busy one
    // stays, and root expands ONE record kind per chunk)
    const uint64_t run_frames = (c->p.time_mode == ADDER_TIME_ABSOLUTE_T || c->records_only)
                                    ? c->frames_done : std::min<uint64_t>(c->run_bound, c->frames_done);
    // run records (adder_rr_kernel): the same regime with integer state while n * 255 and n * time_spanned stay exact in
    // binary32; AbsoluteT also wants last_fired_t on multiples of time_spanned (time_spanned == ref_time >= 255)
    const bool rr_off = env_flag("ADDER_HIP_NO_RR");
    // (Mode Normal under the same conditions runs it too -- adder_pixel.hpp rr_step; the other two kernels are Collapse's)
    const bool rr_regime = cr || (generic && !collapse && c->cr_valid && rr_possible(c, time_spanned, true));
    const bool rr = rr_regime && !rr_off &&
                    (c->p.time_mode != ADDER_TIME_ABSOLUTE_T || (time_spanned == (float)c->p.ref_time && c->p.ref_time >= 255u)) &&
                    (double)(run_frames + num_frames) * std::max(255.0, (double)time_spanned) < 16777216.0;
    // lean runs (adder_lr_kernel): the lean regime in DeltaT under the same property, in blocked batches of events, while
    // rho * 255 and rho * time_spanned stay exact in binary32 (rho <= frames since the reset)
    const bool lr_off = env_flag("ADDER_HIP_NO_LR");
    // (AbsoluteT: last_fired_t / T rides along as an integer when time_spanned == ref_time >= 255, like the run records')
    const bool lr_time = c->p.time_mode == ADDER_TIME_DELTA_T ||
                         (c->p.time_mode == ADDER_TIME_ABSOLUTE_T && time_spanned == (float)c->p.ref_time && c->p.ref_time >= 255u);
    // (batches that hand their records out -- the multi-GPU gather -- run it too: the {rho, word} records are the smallest
    // payload and root's expansion works the events out of them like the single-GPU one)
    const bool lr = !generic && !c->continuous && collapse && lr_time && c->cr_valid && !lr_off &&
                    launch_depth(c) > 1u && num_frames > 1u &&
                    (double)(run_frames + num_frames) * std::max(255.0, (double)time_spanned) < 16777216.0;
    // ... in packed bytes (adder_lp_kernel, four units per lane): DeltaT batches whose records the expansion reads itself
    // (the pair's records lie in one run: the ring layout must keep a pair of segments adjacent)
    const bool lp_off = env_flag("ADDER_HIP_NO_LP");
    const bool lp = lr && !lp_off && c->p.time_mode == ADDER_TIME_DELTA_T && !c->records_only && park_group_shift_wanted() >= 1u;
    const uint32_t variant = (lp ? 4096u : 0u) | ((lp && c->p.channels == 3) ? 8192u : 0u) | (collapse ? 1u : 0u) | (c->p.time_mode == ADDER_TIME_ABSOLUTE_T ? 2u : 0u) |
                             (generic ? 4u : 0u) | (c->continuous ? 8u : 0u) |
                             (c->n_units >= 4u ? 16u : 0u) |  // 16: the 4-units-per-lane one-frame kernel may run
                             (cb ? 32u : 0u) | (cr ? 128u : 0u) | (lr ? 256u : 0u) | (rr ? 512u : 0u) | (c->wire_batch ? 1024u : 0u) |
                             ((c->records_only && !lr) ? 64u : 0u);  // 64: lean records in per-segment logs (batches that hand them out)
    c->last_variant = variant;
    if (lr) {
        int rc_ = ensure_lr_tab(c, time_spanned, stream);
        if (rc_ != ADDER_OK) return rc_;
    }
    if ((lr || rr) && !c->d_run_max) {
        HIPCHK(c, dalloc(&c->d_run_max, 1));
        HIPCHK(c, hipMemset(c->d_run_max, 0, sizeof(uint32_t)));  // (once: adder_publish_kernel clears it behind every batch)
    }
    if (rr && !c->d_rr_tab) {  // (does not depend on the time step: a node's last firing is ceil(2^e / I))
        std::vector<uint8_t> tab(256u * kRrTabRows);
        rr_build_tab(tab.data(), 255.0f);
        HIPCHK(c, dalloc(&c->d_rr_tab, tab.size()));
        HIPCHK(c, hipMemcpy(c->d_rr_tab, tab.data(), tab.size(), hipMemcpyHostToDevice));
    }
    if (c->wire_batch && (c->continuous || fpath || c->records_only))
        return fail(c, ADDER_E_BAD_PARAMS, "wire records straight from the expansion: dense FramePerfect batches without feature mode only "
                    "(otherwise integrate events and serialise them with adder_hip_wire_events_device)");
    if (c->records_only && (generic || c->continuous || fpath))
        return fail(c, ADDER_E_BAD_PARAMS, "records can be handed out in the lean regime only (Collapse, delta_t_max <= "
                    "time_spanned, no feature mode, no generic batch before): gather events instead");
    if (generic) {
        // per-event records go to a log per segment and chunk, sized by the hard bound of what a segment can emit
        // (pop_top and a flush exclude each other in one frame when delta_t_max >= 2 * time: 2 instead of 3 per frame)
        const bool two = collapse && (double)c->p.delta_t_max >= 2.0 * (double)time_spanned;
        int rc_ = alloc_scratch(c, two ? AdderHipCtx::kScratchLog2 : AdderHipCtx::kScratchLog3);
        if (rc_ == ADDER_OK) rc_ = alloc_deep_planes(c);
        if (rc_ != ADDER_OK) return rc_;
        c->generic_sticky = true;
    } else if (!c->continuous) {
        // (12-byte records in AbsoluteT, 8-byte ones otherwise: adder_pixel.hpp lean_decode8)
        int rc_ = alloc_scratch(c, c->p.time_mode == ADDER_TIME_ABSOLUTE_T ? AdderHipCtx::kScratchLean : AdderHipCtx::kScratchLean8);
        if (rc_ != ADDER_OK) return rc_;
    }
    // an event buffer below the batch's worst case can overflow: keep an undo copy of the state so that the
    // overflow is recoverable (adder_hip_finish rolls back and reports the size needed)
    if (c->running_enabled && !c->running) {
        int rc_ = alloc_running(c, c->stream);
        if (rc_ != ADDER_OK) return rc_;
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->snap.valid = false;
    {
        const size_t per_frame = (size_t)c->n_units * (c->continuous ? c->max_depth + 3u : generic ? c->max_depth + 1u : 3u);
        if (out_cap < per_frame * num_frames && !c->no_snapshot) {
            int rc_ = take_snapshot(c, generic, stream);
            if (rc_ != ADDER_OK) return rc_;
            c->snap.generic_sticky = sticky_before;
        }
    }
    if (fpath) {
        int rc_ = prepare_per_unit_c_thresh(c, stream);
        if (rc_ != ADDER_OK) return rc_;
    }

    // ---- batch description -> device ----
    if (c->ftab_cap < num_frames) {
        // (the captured graphs hold the description's address)
        for (auto &kv : c->graphs)
            for (hipGraphExec_t e : kv.second.cand)
                if (e) c->retired_execs.push_back(e);
        c->graphs.clear();
        c->tune_pending = false;
        int rc_ = alloc_batch_desc(c, std::max<size_t>(num_frames, 2 * c->ftab_cap));
        if (rc_ != ADDER_OK) return rc_;
    }
    float rt = c->running_t;
    uint8_t cth = c->c_thresh, cctr = c->c_counter;
    for (uint32_t f = 0; f < num_frames; ++f) {
        c->h_ftab[f].running_t = rt;
        c->h_ftab[f].cth = cth;  // the contrast test of frame f sees the value before its integrate
        rt += time_spanned;      // `self.running_t += time` (event_pixel_tree.rs:336), f32
        c_thresh_advance(cth, cctr, c->p.c_thresh_max, c->p.c_increase_velocity, time_spanned, c->p.ref_time);
    }
    BatchArgs &b = *c->h_batch;
    base_args(c, &b.base);
    b.base.out = reinterpret_cast<AdderEventPod *>(d_out);
    b.base.out_cap = out_cap;
    b.base.frame_offsets = d_offsets;
    b.base.lean = (variant & 512u) ? 3u : (generic || c->continuous) ? 0u : ((variant & 256u) ? 2u : 1u);  // 2: lean-runs records (lr_decode8), 3: run records (rr_event)
    b.base.abs_t = c->p.time_mode == ADDER_TIME_ABSOLUTE_T ? 1u : 0u;
    b.base.wire_rec = c->wire_batch ? (c->p.channels == 1 ? 9u : 11u) : 0u;
    b.base.sc = make_consts(c, time_spanned);
    b.frames = d_frames;
    b.ftab = c->d_ftab;
    b.park_ring = c->park_ring;
    // run records: a fixed slot per (segment, frame) like the lean records' (at most one 8 / 12-byte record per unit and
    // frame), laid out by park_offset inside the log ring (2 KB per segment and frame there: enough) -- the expansion
    // reads the sixteen segments of a wave out of one contiguous stretch instead of sixteen logs 64 KB apart
    const uint32_t pb = (variant & 512u) ? kWaveUnits * (c->p.time_mode == ADDER_TIME_ABSOLUTE_T ? 12u : 8u) : c->park_bytes;
    b.park_bytes = pb;
    // (a lean batch's region holds one record per unit and frame of the chunk: the same bytes as its fixed slots)
    b.log_cap = (variant & 512u) ? 0u : (variant & 64u) ? kWaveUnits * c->chunk : c->log_cap;
    b.rr_tab = c->d_rr_tab;
    b.lr_tab = c->d_lr_tab;
    {
        const bool sd = c->snap.valid && c->snap.deep;  // this batch keeps an undo copy of the levels >= 1
        b.snap_dv_integ = sd ? c->snap.dv_integ : nullptr;
        b.snap_dv_dt = sd ? c->snap.dv_dt : nullptr;
        b.snap_dv_bdt = sd ? c->snap.dv_bdt : nullptr;
        b.snap_dv_bd = sd ? c->snap.dv_bd : nullptr;
    }
    b.run_max = (lr || rr) ? c->d_run_max : nullptr;
    b.wofs_ring = c->wofs_ring;
    b.wcur = c->wcur;
    // ring layout (park_offset): batches launched one frame at a time park frame-major, the others in groups of
    // segments (ADDER_HIP_PARK_GROUP_SHIFT: 0 = segment-major)
    if (b.log_cap) {
        b.park_layout = ParkLayout{0u, 0u, 0u, 0u, 31u, 0xffffffffu};  // (unused: the records are appended to logs)
    } else if (launch_depth(c) == 1u && (uint64_t)c->num_waves * pb <= 0xffffffffull) {
        b.park_layout = ParkLayout{31u, 0u, c->num_waves * pb, pb, 31u, 0xffffffffu};
    } else {
        uint32_t sh = c->park_group_shift;
        while (sh && ((c->num_waves & ((1u << sh) - 1u)) || ((uint64_t)c->chunk * pb << sh) > 0xffffffffull)) --sh;
        b.park_layout = ParkLayout{sh, (c->chunk * pb) << sh, pb << sh, pb, 31u, 0xffffffffu};
        // segment-major: rotate the frame slots by the segment's group of 16 (needs a power-of-two chunk)
        static const bool rot_on = [] { const char *e = getenv("ADDER_HIP_PARK_ROT"); return !e || atoi(e) != 0; }();
        if (rot_on && sh == 0u && c->chunk >= 2u && (c->chunk & (c->chunk - 1u)) == 0u &&
            (uint64_t)c->chunk * pb <= 0xffffffffull) {
            b.park_layout.rot_shift = 4u;
            b.park_layout.rot_mask = c->chunk - 1u;
        }
    }
    b.wtot_ring = c->wtot_ring;
    b.wpref_ring = c->wpref_ring;
    b.ftot_ring = c->ftot_ring;
    if (c->frame_only && c->graph_slot != 0u) {
        // the per-frame ring: a one-frame batch always sits in scratch slot 0 -- each frame slot of the ring gets a ring
        // CHUNK of its own as "its" scratch (views of the rings shifted by whole chunks), so that the scan and expansion of
        // frame k may run beside the frame kernel of frame k + 1
        const uint32_t rk = (c->graph_slot - 1u) % c->ring_chunks;
        b.park_ring += (size_t)rk * scratch_bytes_per_chunk(c, c->scratch_kind, c->chunk);
        const size_t rows = (size_t)rk * c->chunk * c->num_waves;
        b.wtot_ring += rows;
        b.wpref_ring += rows;
        if (b.wofs_ring) b.wofs_ring += rows;
        if (b.wcur) b.wcur += (size_t)rk * c->num_waves;
        b.ftot_ring += (size_t)rk * c->chunk;  // (its second half and the tile sums move along: still inside the allocation)
    }
    b.slots = c->slots;
    b.chunk = c->chunk;
    b.rec_total = c->d_rec_total;
    b.timeline = nullptr;
    if (getenv("ADDER_HIP_TIMELINE")) {  // diagnostics (adder_hip_debug_timeline)
        if (!c->d_timeline) HIPCHK(c, dalloc(&c->d_timeline, 4 * kTimelineChunks * 2));
        std::vector<unsigned long long> init(4 * kTimelineChunks * 2);
        for (size_t i = 0; i < init.size(); ++i) init[i] = (i & 1) ? 0ull : ~0ull;
        HIPCHK(c, hipMemcpy(c->d_timeline, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        b.timeline = c->d_timeline;
    }
    // (frame_offsets[0] and the record count are started by the first chunk's offsets kernel: nothing to clear here)
    static const bool dbg_memset = env_flag("ADDER_HIP_DBG_DESC_MEMSET");  // (tests/test_gpu_stress.py: round 5's arrangement, on purpose)
    if (dbg_memset && c->d_rec_total) HIPCHK(c, hipMemsetAsync(c->d_rec_total, 0, 4, stream));
    if (reinterpret_cast<uint8_t *>(c->d_ftab) == reinterpret_cast<uint8_t *>(c->d_batch) + kBatchDescBytes &&
        reinterpret_cast<uint8_t *>(c->h_ftab) == reinterpret_cast<uint8_t *>(c->h_batch) + kBatchDescBytes) {
        HIPCHK(c, hipMemcpyAsync(c->d_batch, c->h_batch, kBatchDescBytes + num_frames * sizeof(FrameTab),
                                 hipMemcpyHostToDevice, stream));
    } else {  // a frame slot's own pair
        HIPCHK(c, hipMemcpyAsync(c->d_ftab, c->h_ftab, num_frames * sizeof(FrameTab), hipMemcpyHostToDevice, stream));
        HIPCHK(c, hipMemcpyAsync(c->d_batch, c->h_batch, sizeof(BatchArgs), hipMemcpyHostToDevice, stream));
    }

    c->timed_launches = 0;
    c->timed_pairs = 0;
    c->timed_frames = 0;
    c->timed_posts = 0;
    const bool timing = c->launch_timing != 0;
    if (timing) {
        while (c->launch_events.size() < 2 * (size_t)num_frames) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreate(&e));
            c->launch_events.push_back(e);
        }
        while (c->post_events.size() < 2 * (size_t)num_frames) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreate(&e));
            c->post_events.push_back(e);
        }
    }
    c->h_result->valid = 0u;
    HIPCHK(c, hipEventRecord(c->ev_start, stream));
    int rc = ADDER_OK;
    c->frame_only_done = false;
    if (c->frame_only && num_frames == 1u && !fpath && !timing && !c->records_only) {
        // the per-frame ring: only the frame kernel goes on this stream -- the NEXT frame's kernel needs nothing else of
        // this frame, so the ring puts scan, expansion and hand-over on its second stream, beside that kernel
        const Lean1wArgs wide = lean1w_args(c);
        HIPCHK(c, adder_launch_frame(c->d_batch, 0u, 1u, variant, c->num_waves, 0u, stream, &wide));
        c->frame_only_done = true;
    } else if (c->records_only) {
        rc = launch_frame_loop(c, num_frames, variant, stream, nullptr, false);  // (stops after scan + offsets)
    } else if (fpath) {
        rc = launch_feature_loop(c, num_frames, variant, stream);
    } else if (c->use_graph && !timing) {
        hipGraphExec_t exec = nullptr;
        rc = get_graph(c, num_frames, variant, &exec);
        if (rc == ADDER_OK) HIPCHK(c, hipGraphLaunch(exec, stream));
    } else if (c->eager_two_streams && !timing) {
        // the graph's two-stream structure, submitted eagerly (diagnostics)
        HIPCHK(c, hipEventRecord(c->cap_e1, stream));
        HIPCHK(c, hipStreamWaitEvent(c->cap_s, c->cap_e1, 0));
        rc = launch_frame_loop(c, num_frames, variant, c->cap_s, c->cap_s2, false);
        if (rc == ADDER_OK) {
            HIPCHK(c, hipEventRecord(c->cap_e1, c->cap_s));
            HIPCHK(c, hipStreamWaitEvent(stream, c->cap_e1, 0));
        }
    } else {
        rc = launch_frame_loop(c, num_frames, variant, stream, nullptr, timing);
    }
    if (rc != ADDER_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev_stop, stream));
    c->running_t = rt;
    c->c_thresh = cth;
    c->c_counter = cctr;
    c->last_time_spanned = time_spanned;
    c->frames_done += num_frames;
    c->run_bound += num_frames;  // (every unit may have gone on accumulating; adder_hip_finish takes the kernels' report)
    c->pending_end_frames = c->frames_done;
    c->pending_reports = b.run_max != nullptr;
    c->band_frame_pending = band_features(c);
    return ADDER_OK;
}

// The same batch with the raw sink's records as its output (include/adder_hip.h): out_cap in records.
extern "C" int adder_hip_integrate_wire_device(AdderHipCtx *c, const uint8_t *d_frames, uint32_t num_frames, float time_spanned,
                                               uint8_t *d_wire, size_t wire_cap_bytes, uint64_t *d_frame_offsets, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    const size_t rec = c->p.channels == 1 ? 9u : 11u;
    c->wire_batch = true;
    const int rc = adder_hip_integrate_device(c, d_frames, num_frames, time_spanned, reinterpret_cast<AdderEvent *>(d_wire),
                                              wire_cap_bytes / rec, d_frame_offsets, stream);
    c->wire_batch = false;
    return rc;
}

// stream == NULL means "the default stream" to a caller, and the context's own stream is a non-blocking one: whatever the
// caller queued on the legacy default stream -- a torch.zeros() of the offsets, the clip's generator -- is NOT ordered against
// it.  (Round 5's "a 3-frame batch left its offsets untouched" was this: the test's zero-fill of the offsets tensor, on the
// null stream, landing AFTER the short batch had written them; tests/test_gpu_stress.py holds the arrangement.)  So a batch
// submitted with stream == NULL first waits for the default stream's work queued so far.
static int join_null_stream(AdderHipCtx *c) {
    static const bool off = env_flag("ADDER_HIP_DBG_NO_NULL_JOIN");  // (tests/test_gpu_stress.py shows the race with it)
    if (off) return ADDER_OK;
    if (!c->null_join) HIPCHK(c, hipEventCreateWithFlags(&c->null_join, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->null_join, nullptr));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->null_join, 0));
    return ADDER_OK;
}

extern "C" int adder_hip_integrate_device(AdderHipCtx *c, const uint8_t *d_frames, uint32_t num_frames,
                                          float time_spanned, AdderEvent *d_out, size_t out_cap,
                                          uint64_t *d_frame_offsets, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->poisoned) return fail(c, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", c->err.c_str());
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "previous device batch not finished (call adder_hip_finish)");
    if (c->f_submitted != c->f_collected)
        return fail(c, ADDER_E_BAD_PARAMS, "frames are in flight (call adder_hip_frame_collect)");
    if (!d_frames || !d_frame_offsets || (!d_out && out_cap)) return fail(c, ADDER_E_BAD_PARAMS, "null pointer");
    if (!(time_spanned >= 0.0f)) return fail(c, ADDER_E_BAD_PARAMS, "time_spanned must be >= 0");
    { int rc_ = band_precheck(c, num_frames); if (rc_ != ADDER_OK) return rc_; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (!stream) { int rc_ = join_null_stream(c); if (rc_ != ADDER_OK) return rc_; }
    if (num_frames == 0) {
        HIPCHK(c, hipMemsetAsync(d_frame_offsets, 0, sizeof(uint64_t), s));
    } else {
        int rc = enqueue_frames(c, d_frames, num_frames, time_spanned, d_out, out_cap, d_frame_offsets, s);
        if (rc != ADDER_OK) {
            c->poisoned = true;
            return rc;
        }
    }
    c->pending = true;
    c->pending_stream = s;
    c->pending_offsets = d_frame_offsets;
    c->pending_frames = num_frames;
    c->pending_cap = out_cap;
    return ADDER_OK;
}

// ---- records over the wire (include/adder_hip.h) ----
extern "C" uint32_t adder_hip_band_segments(const AdderHipCtx *c) { return c ? c->num_waves : 0u; }

extern "C" int adder_hip_integrate_records_device(AdderHipCtx *c, const uint8_t *d_frames, uint32_t num_frames,
                                                  float time_spanned, uint64_t *d_frame_offsets, void *stream,
                                                  AdderBandRecords *out) {
    if (!c || !out) return ADDER_E_BAD_PARAMS;
    if (c->poisoned) return fail(c, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", c->err.c_str());
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "previous device batch not finished (call adder_hip_finish)");
    if (c->f_submitted != c->f_collected || c->submitted != c->collected)
        return fail(c, ADDER_E_BAD_PARAMS, "frames are in flight (collect them first)");
    if (!d_frames || !d_frame_offsets || num_frames == 0) return fail(c, ADDER_E_BAD_PARAMS, "null pointer / no frames");
    if (!(time_spanned >= 0.0f)) return fail(c, ADDER_E_BAD_PARAMS, "time_spanned must be >= 0");
    // The lean regime is a precondition the CALLER can miss (include/adder_gather.h: "gather events instead"): it is
    // checked before anything is touched -- no scratch re-laid out, no graph retired, no poisoned context.
    if (c->generic_sticky || c->perpx || c->continuous || c->sparse_mode || feature_path(c) || feature_needs_perpx(c) ||
        !(c->p.multi_mode == ADDER_MULTI_COLLAPSE && (float)c->p.delta_t_max <= time_spanned))
        return fail(c, ADDER_E_BAD_PARAMS, "records can be handed out in the lean regime only (Collapse, delta_t_max <= "
                    "time_spanned, no feature mode, no generic batch before): gather events instead");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (!stream) { int rc_ = join_null_stream(c); if (rc_ != ADDER_OK) return rc_; }
    // the scratch must exist before the chunk size is known
    {
        int rc_ = alloc_scratch(c, c->p.time_mode == ADDER_TIME_ABSOLUTE_T ? AdderHipCtx::kScratchLean : AdderHipCtx::kScratchLean8);
        if (rc_ != ADDER_OK) return rc_;
    }
    if (c->ring_chunks < 2 || num_frames > c->chunk)
        return fail(c, ADDER_E_BAD_PARAMS, "a records batch holds at most adder_hip_chunk_frames() = %u frames", c->chunk);
    c->records_only = true;
    const bool snap = c->no_snapshot;
    c->no_snapshot = true;  // (nothing can overflow: no event buffer)
    int rc = enqueue_frames(c, d_frames, num_frames, time_spanned, nullptr, (size_t)1 << 62, d_frame_offsets, s);
    c->no_snapshot = snap;
    c->records_only = false;
    if (rc != ADDER_OK) {
        c->poisoned = true;
        return rc;
    }
    // the batch started at slot 0: the first num_frames rows of the rings are its tables, chunk 0 of the ring its logs (or
    // its fixed slots: lean-runs records); the used prefixes are packed into chunk 1's (idle) region, the runs re-based
    // onto the packed buffer
    const uint32_t rb = lean_rec_bytes(c->p.time_mode == ADDER_TIME_ABSOLUTE_T);
    const uint32_t cap = kWaveUnits * c->chunk;
    const size_t chunk_bytes = (size_t)c->num_waves * cap * rb;
    uint8_t *const packed = c->park_ring + chunk_bytes;
    const bool runs = (c->last_variant & 256u) != 0u;  // adder_lr_kernel ran: {rho, word} records, frame-major packing
    if (runs) {
        HIPCHK(c, adder_launch_slot_pack(c->d_batch, num_frames, c->num_waves, rb, packed, chunk_bytes, c->status, s));
    } else {
        uint32_t *const pbase = c->wcur + c->num_waves;
        uint64_t *const d_total = c->d_side_words;  // (its own word: wcur holds ring_chunks >= 2 rows and two are in use)
        HIPCHK(c, adder_launch_log_pack(c->park_ring, cap, rb, c->wcur, pbase, c->num_waves, num_frames, c->wofs_ring, packed,
                                        chunk_bytes, d_total, c->status, s));
    }
    HIPCHK(c, hipEventRecord(c->ev_stop, s));
    out->num_frames = num_frames;
    out->num_segments = c->num_waves;
    out->record_bytes = rb | (runs ? (uint32_t)ADDER_RECORDS_RUNS : 0u);
    out->row_begin = c->p.row_begin;
    out->rows = c->rows;
    out->d_counts = c->wtot_ring;
    out->d_prefix = c->wpref_ring;
    out->d_runs = c->wofs_ring;
    out->d_records = packed;
    out->d_frame_offsets = d_frame_offsets;
    out->d_frame_table = c->d_ftab;
    c->pending = true;
    c->pending_stream = s;
    c->pending_offsets = d_frame_offsets;
    c->pending_frames = num_frames;
    c->pending_cap = (size_t)1 << 62;
    return ADDER_OK;
}

static int ensure_lr_tab(AdderHipCtx *c, float time_spanned, hipStream_t stream);
// wire: d_merged receives the raw sink's 9 / 11-byte records (merged_cap / merged_base still count events)
static int expand_records_impl(AdderHipCtx *c, const AdderBandRecords *bands, uint32_t n_bands, AdderEvent *d_merged,
                               size_t merged_cap, uint64_t merged_base, uint64_t *d_merged_offsets, void *stream, bool wire) {
    if (!c || !bands || n_bands == 0 || !d_merged_offsets || (!d_merged && merged_cap)) return ADDER_E_BAD_PARAMS;
    if (c->poisoned) return fail(c, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", c->err.c_str());
    const uint32_t nf = bands[0].num_frames;
    const bool abs_t = c->p.time_mode == ADDER_TIME_ABSOLUTE_T;
    if (nf == 0 || nf > kMaxChunk || nf > c->ftab_cap) return fail(c, ADDER_E_BAD_PARAMS, "bad num_frames (root integrates the same frames first)");
    const bool runs = (bands[0].record_bytes & (uint32_t)ADDER_RECORDS_RUNS) != 0u;  // lean-runs records (expansion format 6)
    for (uint32_t r = 0; r < n_bands; ++r)
        if (bands[r].num_frames != nf || bands[r].record_bytes != (lean_rec_bytes(abs_t) | (runs ? (uint32_t)ADDER_RECORDS_RUNS : 0u)) || !bands[r].d_counts ||
            !bands[r].d_prefix || !bands[r].d_runs || !bands[r].d_records || !bands[r].d_frame_offsets ||
            bands[r].num_segments % (uint32_t)ADDER_EXPAND_SEGS != 0u || bands[r].rows == 0)
            return fail(c, ADDER_E_BAD_PARAMS, "band %u: bad description", r);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (runs) {  // event C's table (root's own lean-runs batches build it too; a root that only expands needs it all the same)
        int rc_ = ensure_lr_tab(c, c->last_time_spanned, s);
        if (rc_ != ADDER_OK) return rc_;
    }
    // device block: n_bands BatchArgs, then the bands' offsets pointers, then the destination table [n_bands][nf]
    const size_t ptrs_at = (size_t)n_bands * kBatchDescBytes;
    const size_t dest_at = ptrs_at + (((size_t)n_bands * sizeof(uint64_t *) + 255) & ~(size_t)255);
    const size_t bytes = dest_at + (size_t)n_bands * nf * sizeof(uint64_t);
    constexpr uint32_t kSlots = AdderHipCtx::kBandDescSlots;
    if (c->band_desc_cap < bytes) {
        HIPCHK(c, hipStreamSynchronize(s));  // (the kernels queued from the old block have to be through: rare, it only grows)
        if (c->d_band_desc) HIPCHK(c, hipFree(c->d_band_desc));
        if (c->h_band_desc) HIPCHK(c, hipHostFree(c->h_band_desc));
        c->d_band_desc = c->h_band_desc = nullptr;
        c->band_desc_cap = 0;
        const size_t slot_bytes = (bytes + 4095) & ~(size_t)4095;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_band_desc), slot_bytes * kSlots));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_band_desc), slot_bytes * kSlots, hipHostMallocDefault));
        c->band_desc_cap = slot_bytes;
        for (hipEvent_t &e : c->band_desc_e) {
            if (e) HIPCHK(c, hipEventDestroy(e));
            e = nullptr;
        }
    }
    const uint32_t slot = c->band_desc_next++ % kSlots;
    if (c->band_desc_e[slot]) HIPCHK(c, hipEventSynchronize(c->band_desc_e[slot]));  // the call four calls ago is through
    else HIPCHK(c, hipEventCreateWithFlags(&c->band_desc_e[slot], hipEventDisableTiming));
    uint8_t *const d_blk = c->d_band_desc + (size_t)slot * c->band_desc_cap;
    uint8_t *const h_blk = c->h_band_desc + (size_t)slot * c->band_desc_cap;
    uint64_t *const d_dest = reinterpret_cast<uint64_t *>(d_blk + dest_at);
    const uint64_t **h_ptrs = reinterpret_cast<const uint64_t **>(h_blk + ptrs_at);
    for (uint32_t r = 0; r < n_bands; ++r) {
        BatchArgs &b = *reinterpret_cast<BatchArgs *>(h_blk + (size_t)r * kBatchDescBytes);
        memset(&b, 0, sizeof b);
        base_args(c, &b.base);
        b.base.status = reinterpret_cast<uint32_t *>(c->d_side_words + 1);  // (the expansions' own word: d_side_words)
        b.base.n_units = bands[r].rows * c->p.width * c->p.channels;
        b.base.num_waves = bands[r].num_segments;
        b.base.row_begin = bands[r].row_begin;
        b.base.out = reinterpret_cast<AdderEventPod *>(d_merged);
        b.base.out_cap = merged_cap;
        b.base.frame_offsets = d_dest + (size_t)r * nf;  // (the expansion reads [f] only: where the band's frame starts)
        b.base.lean = runs ? 2u : 1u;
        b.base.abs_t = abs_t ? 1u : 0u;
        b.base.wire_rec = wire ? (c->p.channels == 1 ? 9u : 11u) : 0u;
        b.lr_tab = c->d_lr_tab;
        b.base.sc = make_consts(c, c->last_time_spanned);
        // the frames' running_t (D_EMPTY fillers): root's own batch of the same frames, or the copy the caller kept of it
        b.ftab = bands[r].d_frame_table ? const_cast<FrameTab *>(reinterpret_cast<const FrameTab *>(bands[r].d_frame_table)) : c->d_ftab;
        b.park_ring = const_cast<uint8_t *>(bands[r].d_records);
        b.park_bytes = 0;
        b.log_cap = 0;  // (expansion format 3 with a zero region stride: the runs index ONE packed buffer)
        b.wofs_ring = const_cast<uint32_t *>(bands[r].d_runs);
        b.wtot_ring = const_cast<uint32_t *>(bands[r].d_counts);
        b.wpref_ring = const_cast<uint32_t *>(bands[r].d_prefix);
        b.park_layout = ParkLayout{0u, 0u, 0u, 0u, 31u, 0xffffffffu};
        b.slots = kMaxChunk;  // slot of frame f = f: the tables' rows
        b.chunk = kMaxChunk;
        h_ptrs[r] = bands[r].d_frame_offsets;
    }
    HIPCHK(c, hipMemcpyAsync(d_blk, h_blk, dest_at, hipMemcpyHostToDevice, s));
    HIPCHK(c, adder_launch_band_layout(reinterpret_cast<const uint64_t *const *>(d_blk + ptrs_at), n_bands, nf,
                                       merged_base, d_merged_offsets, d_dest, s));
    if (n_bands <= kMaxBands) {  // every band in ONE launch, frame-major across the bands
        uint32_t nw[kMaxBands];
        for (uint32_t r = 0; r < n_bands; ++r) nw[r] = bands[r].num_segments;
        HIPCHK(c, adder_launch_expand_bands(d_blk, (uint32_t)kBatchDescBytes, n_bands, nw, nf, abs_t ? 1u : 0u, s, runs ? 1u : 0u,
                                            wire ? 1u : 0u));
    } else {
        const uint32_t variant = 1u | (abs_t ? 2u : 0u) | 64u | (runs ? 256u : 0u) | (wire ? 1024u : 0u);
        for (uint32_t r = 0; r < n_bands; ++r)
            HIPCHK(c, adder_launch_expand(reinterpret_cast<const BatchArgs *>(d_blk + (size_t)r * kBatchDescBytes), 0u, nf,
                                          bands[r].num_segments, variant, 0u, s));
    }
    HIPCHK(c, hipEventRecord(c->band_desc_e[slot], s));  // (the device block is read by the kernels: free after them)
    return ADDER_OK;
}
extern "C" int adder_hip_expand_records_device(AdderHipCtx *c, const AdderBandRecords *bands, uint32_t n_bands,
                                               AdderEvent *d_merged, size_t merged_cap, uint64_t merged_base,
                                               uint64_t *d_merged_offsets, void *stream) {
    return expand_records_impl(c, bands, n_bands, d_merged, merged_cap, merged_base, d_merged_offsets, stream, false);
}
extern "C" int adder_hip_expand_records_wire_device(AdderHipCtx *c, const AdderBandRecords *bands, uint32_t n_bands,
                                                    uint8_t *d_wire, size_t wire_cap_bytes, uint64_t merged_base,
                                                    uint64_t *d_merged_offsets, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    return expand_records_impl(c, bands, n_bands, reinterpret_cast<AdderEvent *>(d_wire), wire_cap_bytes / (c->p.channels == 1 ? 9u : 11u),
                               merged_base, d_merged_offsets, stream, true);
}

static size_t wire_align(size_t x) { return (x + 255) & ~(size_t)255; }
extern "C" void adder_hip_records_wire_sections(uint32_t nf, uint32_t nseg, uint32_t rb, size_t sec[6]) {
    (void)rb;
    const size_t tab = wire_align((size_t)nf * nseg * sizeof(uint32_t));
    sec[0] = 0;                                                    // frame offsets
    sec[1] = sec[0] + wire_align(((size_t)nf + 1) * sizeof(uint64_t));  // frame table
    sec[2] = sec[1] + wire_align((size_t)nf * sizeof(FrameTab));   // counts
    sec[3] = sec[2] + tab;                                         // prefix
    sec[4] = sec[3] + tab;                                         // runs
    sec[5] = sec[4] + tab;                                         // records
}
extern "C" size_t adder_hip_records_wire_bytes(uint32_t nf, uint32_t nseg, uint32_t rb, uint64_t n_records) {
    size_t sec[6];
    adder_hip_records_wire_sections(nf, nseg, rb, sec);
    return sec[5] + wire_align((size_t)n_records * (rb & 0xffu));  // (record_bytes may carry ADDER_RECORDS_RUNS)
}
extern "C" int adder_hip_records_to_wire(AdderHipCtx *c, const AdderBandRecords *rec, uint64_t n_records, void *d_dst,
                                         size_t dst_bytes, void *stream) {
    if (!c || !rec || !d_dst) return ADDER_E_BAD_PARAMS;
    static_assert(sizeof(FrameTab) == 8, "the wire image's frame table rows");
    const uint32_t nf = rec->num_frames, nseg = rec->num_segments, rb = rec->record_bytes;
    if (adder_hip_records_wire_bytes(nf, nseg, rb, n_records) > dst_bytes) return fail(c, ADDER_E_BAD_PARAMS, "wire buffer too small");
    HIPCHK(c, hipSetDevice(c->device));
    // (null: the stream the context's last batch ran on -- the copies are then ordered before its next batch)
    hipStream_t s = stream ? (hipStream_t)stream : (c->pending_stream ? c->pending_stream : c->stream);
    size_t sec[6];
    adder_hip_records_wire_sections(nf, nseg, rb, sec);
    uint8_t *dst = static_cast<uint8_t *>(d_dst);
    const size_t tab = (size_t)nf * nseg * sizeof(uint32_t);
    HIPCHK(c, hipMemcpyAsync(dst + sec[0], rec->d_frame_offsets, ((size_t)nf + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dst + sec[1], rec->d_frame_table ? rec->d_frame_table : c->d_ftab, (size_t)nf * sizeof(FrameTab),
                             hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dst + sec[2], rec->d_counts, tab, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dst + sec[3], rec->d_prefix, tab, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dst + sec[4], rec->d_runs, tab, hipMemcpyDeviceToDevice, s));
    if (n_records) HIPCHK(c, hipMemcpyAsync(dst + sec[5], rec->d_records, (size_t)n_records * (rb & 0xffu), hipMemcpyDeviceToDevice, s));
    return ADDER_OK;
}

extern "C" void *adder_hip_last_batch_stream(AdderHipCtx *c) {
    return c ? (void *)(c->pending_stream ? c->pending_stream : c->stream) : nullptr;
}
extern "C" int adder_hip_sync_last_batch_stream(AdderHipCtx *c) {
    if (!c) return ADDER_E_BAD_PARAMS;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->pending_stream ? c->pending_stream : c->stream));
    return ADDER_OK;
}

extern "C" int adder_hip_expand_status(AdderHipCtx *c, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    uint32_t st = 0;
    uint32_t *const word = reinterpret_cast<uint32_t *>(c->d_side_words + 1);
    HIPCHK(c, hipMemcpyAsync(&st, word, sizeof st, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (st) HIPCHK(c, hipMemsetAsync(word, 0, sizeof(uint32_t), s));
    if (st & kStatusCapacity)
        return fail(c, ADDER_E_OUT_CAPACITY, "the merged event buffer was too small: events were dropped");
    return status_to_code(c, st);
}

extern "C" int adder_hip_finish(AdderHipCtx *c, size_t *n_out) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (!c->pending) return fail(c, ADDER_E_BAD_PARAMS, "no device batch pending");
    c->pending = false;
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t st = 0;
    uint64_t total = 0;
    c->last_new_features = 0;
    if (c->pending_frames && !(c->d_feat_counters && feature_path(c))) {
        // the batch's last node wrote {events, records, status} into page-locked memory: wait for the event behind
        // it, no copies
        // (a short spin on the flag the last node sets: an event wait parks the thread and wakes up tens of
        // microseconds late; the event is behind the flag in the same stream, so the wait below is then immediate)
        {
            volatile uint32_t *flag = &c->h_result->valid;
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
            uint32_t spins = 0;
            while (!*flag && ((++spins & 1023u) != 0u || std::chrono::steady_clock::now() < t_end)) {
            }
        }
        HIPCHK(c, hipEventSynchronize(c->ev_stop));
        if (!c->h_result->valid) return fail(c, ADDER_E_HIP, "the batch finished without publishing its result");
        st = c->h_result->status;
        total = c->h_result->total_events;
        c->last_records = c->h_result->records;
        if (c->pending_reports)  // the longest run at the batch's end (0: below kRunReportMin) + whatever was queued behind it
            c->run_bound = std::min<uint64_t>(c->run_bound, (uint64_t)std::max(c->h_result->max_run, kRunReportMin) +
                                                                (c->frames_done - c->pending_end_frames));
        c->pending_reports = false;
    } else {
        HIPCHK(c, hipMemcpyAsync(&st, c->status, sizeof st, hipMemcpyDeviceToHost, c->pending_stream));
        HIPCHK(c, hipMemcpyAsync(&total, c->pending_offsets + c->pending_frames, sizeof total, hipMemcpyDeviceToHost,
                                 c->pending_stream));
        HIPCHK(c, hipMemcpyAsync(&c->last_records, c->d_rec_total, sizeof(uint64_t), hipMemcpyDeviceToHost,
                                 c->pending_stream));
        if (c->d_feat_counters && feature_path(c))
            HIPCHK(c, hipMemcpyAsync(&c->last_new_features, c->d_feat_counters, sizeof(uint32_t), hipMemcpyDeviceToHost,
                                     c->pending_stream));
        HIPCHK(c, hipStreamSynchronize(c->pending_stream));
    }
    if (c->pending_frames)
        HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev_start, c->ev_stop));
    else
        c->last_ms = 0.0f;
    graph_tune_report(c, c->last_ms);
    c->last_launch_avg_us = 0.0f;
    if (c->timed_launches) {
        double sum = 0.0;
        for (uint32_t f = 0; f < c->timed_pairs; ++f) {
            float ms = 0.0f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->launch_events[2 * f], c->launch_events[2 * f + 1]));
            sum += ms;
        }
        c->last_launch_avg_us = (float)(sum * 1000.0 / c->timed_launches);
    }
    c->last_post_avg_us = 0.0f;
    if (c->timed_posts) {
        double sum = 0.0;
        for (uint32_t k = 0; k < c->timed_posts; ++k) {
            float ms = 0.0f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->post_events[2 * k], c->post_events[2 * k + 1]));
            sum += ms;
        }
        c->last_post_avg_us = (float)(sum * 1000.0 / c->timed_posts);
    }
    if (n_out) *n_out = (size_t)total;
    if (st == kStatusCapacity && c->snap.valid) {
        // the event buffer was too small: nothing else went wrong, and the state before the batch is at hand
        int rc = restore_snapshot(c, c->pending_stream);
        c->band_frame_pending = false;  // (the frame is gone with the rollback: the retry is a fresh integrate call)
        if (rc != ADDER_OK) {
            c->poisoned = true;
            return rc;
        }
        return fail(c, ADDER_E_OUT_CAPACITY, "event buffer too small: the batch needs %llu events (state rolled back, "
                    "retry with a larger buffer)", (unsigned long long)total);
    }
    c->snap.valid = false;
    // (a total beyond the caller's buffer with no capacity status cannot come out of a sound batch: refuse it here rather
    // than hand the caller a count it would copy by)
    if (st == 0u && total > (uint64_t)c->pending_cap) {
        c->poisoned = true;
        return fail(c, ADDER_E_HIP, "the batch reports %llu events for a buffer of %llu without a capacity status (frame offsets damaged)",
                    (unsigned long long)total, (unsigned long long)c->pending_cap);
    }
    const int rc_st = status_to_code(c, st);
    if (rc_st != ADDER_OK) c->band_frame_pending = false;  // (no feature step follows a failed frame)
    return rc_st;
}

extern "C" float adder_hip_last_batch_ms(AdderHipCtx *c) { return c ? c->last_ms : 0.0f; }

// diagnostics: the last batch's kernel timeline (ADDER_HIP_TIMELINE=1), [4 kinds][64 chunks][start, end] in 10 ns ticks
extern "C" int adder_hip_debug_timeline(AdderHipCtx *c, unsigned long long *dst) {
    if (!c || !dst || !c->d_timeline) return ADDER_E_BAD_PARAMS;
    HIPCHK(c, hipMemcpy(dst, c->d_timeline, 4 * kTimelineChunks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return ADDER_OK;
}

extern "C" unsigned adder_hip_last_batch_kernel(const AdderHipCtx *c) {
    if (!c) return ADDER_KERNEL_LEAN;
    const uint32_t v = c->last_variant;  // (the order of adder_launch_frame's tests)
    if (v & 8u) return ADDER_KERNEL_CONTINUOUS;
    if (v & 512u) return ADDER_KERNEL_RUN_RECORDS;
    if (v & 128u) return ADDER_KERNEL_CONSTANT_RUNS;
    if (v & 32u) return ADDER_KERNEL_BOUNDED;
    if (v & 4u) return ADDER_KERNEL_GENERIC;
    if (v & 4096u) return ADDER_KERNEL_LEAN_RUNS_PACKED;
    if (v & 256u) return ADDER_KERNEL_LEAN_RUNS;
    return ADDER_KERNEL_LEAN;
}
extern "C" int adder_hip_launch_plan_settled(const AdderHipCtx *c) {
    if (!c) return 1;
    auto it = c->graphs.find(c->tune_key);
    return it == c->graphs.end() || it->second.chosen >= 0 ? 1 : 0;
}

extern "C" int adder_hip_set_launch_timing(AdderHipCtx *c, int enable) {
    if (!c) return ADDER_E_BAD_PARAMS;
    c->launch_timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return ADDER_OK;
}

extern "C" float adder_hip_last_launch_avg_us(AdderHipCtx *c) { return c ? c->last_launch_avg_us : 0.0f; }

extern "C" float adder_hip_last_post_avg_us(AdderHipCtx *c) { return c ? c->last_post_avg_us : 0.0f; }
extern "C" uint32_t adder_hip_last_post_chunks(AdderHipCtx *c) { return c ? c->timed_posts : 0u; }
extern "C" uint64_t adder_hip_last_batch_records(AdderHipCtx *c) { return c ? c->last_records : 0ull; }
extern "C" uint32_t adder_hip_chunk_frames(const AdderHipCtx *c) { return c ? c->chunk : 0u; }
extern "C" uint32_t adder_hip_segment_units(void) { return kWaveUnits; }
extern "C" uint32_t adder_hip_wire_record_bytes(const AdderHipCtx *c) { return c ? (c->p.channels == 1 ? 9u : 11u) : 0u; }

extern "C" float adder_hip_last_launch_frames(AdderHipCtx *c) {
    return (c && c->timed_launches) ? (float)c->timed_frames / (float)c->timed_launches : 0.0f;
}

extern "C" int adder_hip_set_frames_per_launch(AdderHipCtx *c, uint32_t frames) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (frames < 1 || frames > kMaxFramesPerLaunch)
        return fail(c, ADDER_E_BAD_PARAMS, "frames_per_launch must be in 1..%u", kMaxFramesPerLaunch);
    c->frames_per_launch = frames;
    return ADDER_OK;
}

extern "C" int adder_hip_reset(AdderHipCtx *c) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending (call adder_hip_finish)");
    if (c->f_submitted != c->f_collected || c->submitted != c->collected)
        return fail(c, ADDER_E_BAD_PARAMS, "frames are in flight (collect them first)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = init_state(c);
    if (rc != ADDER_OK) return rc;
    // the memsets are queued on the context's stream; whoever touches the state next orders itself behind them
    // (enqueue_frames: a stream wait; the host-side accessors: settle_reset)
    HIPCHK(c, hipEventRecord(c->reset_e, c->stream));
    c->reset_pending = true;
    return ADDER_OK;
}

extern "C" int adder_hip_chunk_offsets_device(AdderHipCtx *c, const AdderEvent *d_events, size_t n_events,
                                              uint32_t *d_chunk_offsets, void *stream) {
    if (!c || !d_chunk_offsets) return ADDER_E_BAD_PARAMS;
    if (n_events > 0xffffffffull) return fail(c, ADDER_E_BAD_PARAMS, "too many events for one frame");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, adder_launch_chunk_offsets(reinterpret_cast<const AdderEventPod *>(d_events), (uint32_t)n_events,
                                         c->p.row_begin, c->p.chunk_rows, c->num_chunks, d_chunk_offsets, s));
    return ADDER_OK;
}

static int ensure(AdderHipCtx *c, void **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return ADDER_OK;
    if (*p) HIPCHK(c, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    HIPCHK(c, hipMalloc(p, std::max<size_t>(need, 16)));
    *cap = need;
    return ADDER_OK;
}

// Host frames -> device, integrate, finish: the events of the batch end up in c->d_events
// (capacity out_cap events), the frame offsets in c->d_offsets; *total = number of events.
static int batch_to_device(AdderHipCtx *c, const uint8_t *frames, uint32_t num_frames, size_t frame_stride,
                           size_t row_stride, float time_spanned, size_t out_cap, size_t *total) {
    *total = 0;
    if (c->poisoned) return fail(c, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", c->err.c_str());
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending (call adder_hip_finish)");
    if (!frames && num_frames) return fail(c, ADDER_E_BAD_PARAMS, "frames is null");
    const size_t rowlen = (size_t)c->p.width * c->p.channels;
    if (row_stride == 0) row_stride = rowlen;
    if (row_stride < rowlen) return fail(c, ADDER_E_BAD_PARAMS, "row_stride_bytes smaller than a row");
    if (frame_stride == 0) frame_stride = row_stride * c->rows;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    void *vp = c->d_frames;
    if ((rc = ensure(c, &vp, &c->d_frames_cap, (size_t)num_frames * c->n_units + 16)) != ADDER_OK) return rc;
    c->d_frames = (uint8_t *)vp;
    vp = c->d_events;
    if ((rc = ensure(c, &vp, &c->d_events_cap, out_cap * sizeof(AdderEvent))) != ADDER_OK) return rc;
    c->d_events = (AdderEvent *)vp;
    vp = c->d_offsets;
    if ((rc = ensure(c, &vp, &c->d_offsets_cap, ((size_t)num_frames + 1) * sizeof(uint64_t))) != ADDER_OK) return rc;
    c->d_offsets = (uint64_t *)vp;

    if (row_stride == rowlen && frame_stride == rowlen * c->rows) {  // packed: one linear copy
        HIPCHK(c, hipMemcpyAsync(c->d_frames, frames, (size_t)num_frames * c->n_units, hipMemcpyHostToDevice, c->stream));
    } else {
        for (uint32_t f = 0; f < num_frames; ++f)
            HIPCHK(c, hipMemcpy2DAsync(c->d_frames + (size_t)f * c->n_units, rowlen, frames + (size_t)f * frame_stride,
                                       row_stride, rowlen, c->rows, hipMemcpyHostToDevice, c->stream));
    }
    rc = adder_hip_integrate_device(c, c->d_frames, num_frames, time_spanned, c->d_events, out_cap, c->d_offsets,
                                    c->stream);
    if (rc != ADDER_OK) return rc;
    rc = adder_hip_finish(c, total);
    if (rc != ADDER_OK) return rc;  // ADDER_E_OUT_CAPACITY: rolled back, *total = the size needed
    return ADDER_OK;
}

extern "C" int adder_hip_integrate_batch(AdderHipCtx *c, const uint8_t *frames, uint32_t num_frames,
                                         size_t frame_stride, size_t row_stride, float time_spanned,
                                         AdderEvent *out, size_t out_cap, size_t *n_out,
                                         uint64_t *frame_offsets) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (n_out) *n_out = 0;
    if (!out && out_cap) return fail(c, ADDER_E_BAD_PARAMS, "out is null");
    size_t total = 0;
    int rc = batch_to_device(c, frames, num_frames, frame_stride, row_stride, time_spanned, out_cap, &total);
    if (n_out) *n_out = total;
    if (rc != ADDER_OK) return rc;
    if (total) HIPCHK(c, hipMemcpy(out, c->d_events, total * sizeof(AdderEvent), hipMemcpyDeviceToHost));
    if (frame_offsets)
        HIPCHK(c, hipMemcpy(frame_offsets, c->d_offsets, ((size_t)num_frames + 1) * sizeof(uint64_t),
                            hipMemcpyDeviceToHost));
    return ADDER_OK;
}

static uint32_t wire_record_bytes(const AdderHipCtx *c) { return c->p.channels == 1 ? 9u : 11u; }

extern "C" int adder_hip_wire_events_device(AdderHipCtx *c, const AdderEvent *d_events, size_t n_events,
                                            uint8_t *d_out, size_t out_cap_bytes, size_t *n_bytes, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (n_bytes) *n_bytes = 0;
    if ((!d_events || !d_out) && n_events) return fail(c, ADDER_E_BAD_PARAMS, "null device buffer");
    const size_t need = n_events * wire_record_bytes(c);
    if (n_bytes) *n_bytes = need;
    if (need > out_cap_bytes) return fail(c, ADDER_E_OUT_CAPACITY, "wire buffer too small: need %zu bytes", need);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, adder_launch_wire(reinterpret_cast<const AdderEventPod *>(d_events), n_events, wire_record_bytes(c), d_out,
                                c->status, (hipStream_t)stream));
    return ADDER_OK;
}

// ---- sink per rank (include/adder_hip.h; SURVEY 8(e): every GPU delivers its own segments) ----
extern "C" int adder_hip_sink_layout_device(AdderHipCtx *c, const uint64_t *d_all_offsets, uint32_t world, uint32_t rank,
                                            uint32_t num_frames, uint64_t *d_file_pos, uint64_t *d_dest,
                                            uint64_t *d_merged_offsets, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (!d_all_offsets || !d_file_pos || !d_dest || world == 0 || rank >= world || num_frames == 0)
        return fail(c, ADDER_E_BAD_PARAMS, "sink layout: null pointer / bad rank / no frames");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, adder_launch_sink_layout(d_all_offsets, world, rank, num_frames, d_file_pos, d_dest, d_merged_offsets,
                                       (hipStream_t)stream));
    return ADDER_OK;
}
extern "C" int adder_hip_wire_scatter_device(AdderHipCtx *c, const AdderEvent *d_events, const uint64_t *d_frame_offsets,
                                             uint32_t num_frames, const uint64_t *d_dest, uint8_t *out, uint64_t out_cap_bytes,
                                             uint64_t header_bytes, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (!d_events || !d_frame_offsets || !d_dest || !out) return fail(c, ADDER_E_BAD_PARAMS, "wire scatter: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    // (the events per frame are known on the device only: a fixed grid of workgroups walks every frame's blocks)
    HIPCHK(c, adder_launch_wire_scatter(reinterpret_cast<const AdderEventPod *>(d_events), d_frame_offsets, num_frames, d_dest,
                                        wire_record_bytes(c), out, out_cap_bytes, header_bytes,
                                        reinterpret_cast<uint32_t *>(c->d_side_words + 1), c->num_cus * 4u, (hipStream_t)stream));
    return ADDER_OK;
}

extern "C" int adder_hip_integrate_batch_raw(AdderHipCtx *c, const uint8_t *frames, uint32_t num_frames,
                                             size_t frame_stride, size_t row_stride, float time_spanned,
                                             uint8_t *out_bytes, size_t out_cap_bytes, size_t *n_bytes,
                                             size_t *n_events, uint64_t *frame_offsets) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (n_bytes) *n_bytes = 0;
    if (n_events) *n_events = 0;
    if (!out_bytes && out_cap_bytes) return fail(c, ADDER_E_BAD_PARAMS, "out_bytes is null");
    const uint32_t rec = wire_record_bytes(c);
    const size_t out_cap = out_cap_bytes / rec;
    size_t total = 0;
    int rc = batch_to_device(c, frames, num_frames, frame_stride, row_stride, time_spanned, out_cap, &total);
    if (n_events) *n_events = total;
    if (n_bytes) *n_bytes = total * rec;
    if (rc != ADDER_OK) return rc;
    void *vp = c->d_wire;
    if ((rc = ensure(c, &vp, &c->d_wire_cap, std::max<size_t>(total * rec, 16))) != ADDER_OK) return rc;
    c->d_wire = (uint8_t *)vp;
    HIPCHK(c, adder_launch_wire(reinterpret_cast<const AdderEventPod *>(c->d_events), total, rec, c->d_wire, c->status,
                                c->stream));
    if (total) HIPCHK(c, hipMemcpyAsync(out_bytes, c->d_wire, total * rec, hipMemcpyDeviceToHost, c->stream));
    if (frame_offsets)
        HIPCHK(c, hipMemcpyAsync(frame_offsets, c->d_offsets, ((size_t)num_frames + 1) * sizeof(uint64_t),
                                 hipMemcpyDeviceToHost, c->stream));
    uint32_t st = 0;
    HIPCHK(c, hipMemcpyAsync(&st, c->status, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return status_to_code(c, st);
}

static int ensure_pinned(AdderHipCtx *c, void **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return ADDER_OK;
    if (*p) HIPCHK(c, hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
    HIPCHK(c, hipHostMalloc(p, std::max<size_t>(need, 16), hipHostMallocDefault));
    *cap = need;
    return ADDER_OK;
}

// Pipelined raw transcode.  submit(k) uploads and integrates batch k (blocking for exactly
// that), then queues its serialisation and download on a second stream and returns; the
// download of batch k therefore overlaps the upload + integration of batch k+1 when the
// caller submits k+1 before collecting k.  At most two batches are in flight.
extern "C" int adder_hip_stream_submit(AdderHipCtx *c, const uint8_t *frames, uint32_t num_frames,
                                       size_t frame_stride, size_t row_stride, float time_spanned,
                                       size_t out_cap_events) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->submitted - c->collected >= 2)
        return fail(c, ADDER_E_BAD_PARAMS, "two batches are in flight: call adder_hip_stream_collect first");
    HIPCHK(c, hipSetDevice(c->device));
    AdderHipCtx::StreamSlot &sl = c->slot[c->submitted & 1u];
    AdderHipCtx::StreamSlot &prev = c->slot[(c->submitted & 1u) ^ 1u];
    if (!c->out_s) HIPCHK(c, hipStreamCreateWithFlags(&c->out_s, hipStreamNonBlocking));
    if (!sl.done) {
        HIPCHK(c, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.wired, hipEventDisableTiming));
    }
    // the previous batch's serialisation still reads c->d_events
    if (prev.busy && prev.wired) HIPCHK(c, hipStreamWaitEvent(c->stream, prev.wired, 0));
    size_t total = 0;
    int rc = batch_to_device(c, frames, num_frames, frame_stride, row_stride, time_spanned, out_cap_events, &total);
    if (rc != ADDER_OK) return rc;
    const uint32_t rec = wire_record_bytes(c);
    void *vp = sl.d_wire;
    if ((rc = ensure(c, &vp, &sl.d_wire_cap, total * rec)) != ADDER_OK) return rc;
    sl.d_wire = (uint8_t *)vp;
    vp = sl.h_wire;
    if ((rc = ensure_pinned(c, &vp, &sl.h_wire_cap, total * rec)) != ADDER_OK) return rc;
    sl.h_wire = (uint8_t *)vp;
    vp = sl.h_offsets;
    if ((rc = ensure_pinned(c, &vp, &sl.h_offsets_cap, ((size_t)num_frames + 1) * sizeof(uint64_t))) != ADDER_OK) return rc;
    sl.h_offsets = (uint64_t *)vp;
    sl.n_events = total;
    sl.n_bytes = total * rec;
    // batch_to_device has synchronised c->stream: the events are complete
    HIPCHK(c, adder_launch_wire(reinterpret_cast<const AdderEventPod *>(c->d_events), total, rec, sl.d_wire, c->status,
                                c->out_s));
    HIPCHK(c, hipMemcpyAsync(sl.h_offsets, c->d_offsets, ((size_t)num_frames + 1) * sizeof(uint64_t),
                             hipMemcpyDeviceToHost, c->out_s));
    HIPCHK(c, hipEventRecord(sl.wired, c->out_s));
    if (total) HIPCHK(c, hipMemcpyAsync(sl.h_wire, sl.d_wire, total * rec, hipMemcpyDeviceToHost, c->out_s));
    HIPCHK(c, hipMemcpyAsync(&sl.status, c->status, sizeof(uint32_t), hipMemcpyDeviceToHost, c->out_s));
    HIPCHK(c, hipEventRecord(sl.done, c->out_s));
    sl.busy = true;
    c->submitted += 1;
    return ADDER_OK;
}

extern "C" int adder_hip_stream_collect(AdderHipCtx *c, const uint8_t **bytes, size_t *n_bytes, size_t *n_events,
                                        const uint64_t **frame_offsets) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->collected == c->submitted) return fail(c, ADDER_E_BAD_PARAMS, "no batch in flight");
    AdderHipCtx::StreamSlot &sl = c->slot[c->collected & 1u];
    HIPCHK(c, hipEventSynchronize(sl.done));
    sl.busy = false;
    c->collected += 1;
    if (bytes) *bytes = sl.h_wire;
    if (n_bytes) *n_bytes = sl.n_bytes;
    if (n_events) *n_events = sl.n_events;
    if (frame_offsets) *frame_offsets = sl.h_offsets;
    return status_to_code(c, sl.status);
}

// ------------------------------------------------------------------------------------------
// Per-frame `consume` contract without a blocking round trip (framed.rs:127-157 calls integrate_matrix once per
// decoded frame).  submit(k) queues upload, integration and hand-over of frame k and returns; collect() waits for
// the oldest frame in flight and returns pointers into its slot of page-locked host memory.  The hand-over
// (adder_frame_out_kernel: events + row-chunk offsets + result header) runs on a second stream, so frame k's
// PCIe transfer overlaps frame k+1's integration.
// ------------------------------------------------------------------------------------------
static bool device_visible_host(const void *p, void **dev) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
    *dev = a.devicePointer;
    return true;
}

static size_t frame_slot_events(const AdderHipCtx *c, float time_spanned) {
    if (c->f_events_per_slot) return c->f_events_per_slot;
    const size_t budget = ((size_t)2 << 30) / sizeof(AdderEvent);
    return std::min(worst_case_events_per_frame(c, time_spanned), budget);
}

static int frame_slot_prepare(AdderHipCtx *c, AdderHipCtx::FrameSlot &fs, size_t need, bool need_host_events) {
    if (!fs.d_frame) {
        HIPCHK(c, dalloc(&fs.d_frame, c->n_units + 16));
        HIPCHK(c, dalloc(&fs.d_offsets, 2));
        // (description and frame table in one block each, like the context's own: one upload per frame)
        uint8_t *dd = nullptr, *hd = nullptr;
        HIPCHK(c, dalloc(&dd, kBatchDescBytes + 64 * sizeof(FrameTab)));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&hd), kBatchDescBytes + 64 * sizeof(FrameTab), hipHostMallocDefault));
        fs.d_batch = reinterpret_cast<BatchArgs *>(dd);
        fs.h_batch = reinterpret_cast<BatchArgs *>(hd);
        fs.d_ftab = reinterpret_cast<FrameTab *>(dd + kBatchDescBytes);
        fs.h_ftab = reinterpret_cast<FrameTab *>(hd + kBatchDescBytes);
        HIPCHK(c, dalloc(&fs.d_counters, 4));
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&fs.h_hdr),
                                sizeof(FrameResult) + ((size_t)c->num_chunks + 1) * sizeof(uint32_t), hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&fs.done, hipEventDisableTiming));
    }
    if (fs.cap < need) {
        if (fs.d_events) HIPCHK(c, hipFree(fs.d_events));
        if (fs.h_events) HIPCHK(c, hipHostFree(fs.h_events));
        fs.d_events = nullptr;
        fs.h_events = nullptr;
        fs.cap = 0;
        HIPCHK(c, dalloc(&fs.d_events, need));
        fs.cap = need;
    }
    if (need_host_events && !fs.h_events)
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&fs.h_events), std::max<size_t>(fs.cap, 1) * sizeof(AdderEvent),
                                hipHostMallocDefault));
    return ADDER_OK;
}

// direct_out: page-locked memory of the caller that takes the events instead of the slot's own (device view), or null
// workgroups of the ring's wire hand-over: enough to keep the link busy, few enough to leave the next frame's kernels
// their CUs (the stores are posted writes over PCIe: a workgroup spends most of its life waiting on them)
static uint32_t wire_scatter_blocks(const AdderHipCtx *c) {
    static const uint32_t env = [] { const char *e = getenv("ADDER_HIP_WIRE_BLOCKS"); return e ? (uint32_t)atoi(e) : 0u; }();
    return env ? env : c->num_cus * 2u;  // (sweep, 1080p e = 0.3: 64-128: 203, 256: 197, 512: 185, 1024: 189 us per frame)
}

struct RingTiming {  // ADDER_HIP_RING_TIMING=1: where adder_hip_frame_submit's host time goes (printed when the process ends)
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t n = 0;
    ~RingTiming() {
        if (n) fprintf(stderr, "[adder_hip] frame_submit host us per call over %llu calls: upload %.1f | in-event pair %.1f | enqueue (description copy + frame kernel) %.1f | frame-event pair %.1f | hand-over launch %.1f | done event %.1f\n",
                       (unsigned long long)n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n);
    }
};
static RingTiming g_ring_timing;
// ADDER_HIP_RING_TIMING=2: the GPU side of the last frames on one clock -- timing events before / behind the upload, around the
// frame kernel and behind the hand-over; adder_hip_destroy prints them relative to the first (us)
struct RingGpuTimeline {
    static constexpr int kFrames = 96, kPer = 5;
    hipEvent_t ev[kFrames][kPer] = {};
    uint64_t n = 0;
    bool made = false;
    void make() {
        if (made) return;
        for (auto &row : ev)
            for (hipEvent_t &e : row) (void)hipEventCreate(&e);
        made = true;
    }
    void report() {
        if (!made || n < 24) return;
        (void)hipDeviceSynchronize();
        const uint64_t first = n > (uint64_t)kFrames ? n - kFrames : 0, from = n - 16;
        hipEvent_t base = ev[from % kFrames][0];
        fprintf(stderr, "[adder_hip] ring timeline, us from frame %llu's upload start: upload start, upload end, kernel start, kernel end, hand-over end\n", (unsigned long long)from);
        (void)first;
        for (uint64_t f = from; f < n; ++f) {
            float t[kPer];
            for (int k = 0; k < kPer; ++k) (void)hipEventElapsedTime(&t[k], base, ev[f % kFrames][k]);
            fprintf(stderr, "  frame %llu: %8.1f %8.1f %8.1f %8.1f %8.1f\n", (unsigned long long)f, t[0] * 1e3f, t[1] * 1e3f, t[2] * 1e3f, t[3] * 1e3f, t[4] * 1e3f);
        }
        n = 0;
    }
};
static RingGpuTimeline g_ring_tl;
static void ring_timeline_report() { g_ring_tl.report(); }
#define RING_T(k) do { if (ring_timing) { const auto t_ = std::chrono::steady_clock::now(); g_ring_timing.acc[k] += std::chrono::duration<double, std::micro>(t_ - rt_).count(); rt_ = t_; } } while (0)
static int frame_submit_impl(AdderHipCtx *c, const uint8_t *frame, size_t row_stride, float time_spanned,
                             AdderEvent *direct_out, size_t direct_cap) {
    if (c->poisoned) return fail(c, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", c->err.c_str());
    if (c->pending) return fail(c, ADDER_E_BAD_PARAMS, "a device batch is pending (call adder_hip_finish)");
    if (c->submitted != c->collected) return fail(c, ADDER_E_BAD_PARAMS, "a raw stream batch is in flight");
    if (!frame) return fail(c, ADDER_E_BAD_PARAMS, "frame is null");
    if (!(time_spanned >= 0.0f)) return fail(c, ADDER_E_BAD_PARAMS, "time_spanned must be >= 0");
    if (c->f_submitted - c->f_collected >= c->f_slots)
        return fail(c, ADDER_E_BAD_PARAMS, "%u frames are in flight: call adder_hip_frame_collect first", c->f_slots);
    if (band_features(c))
        return fail(c, ADDER_E_BAD_PARAMS, "a row band in feature / ROI mode uses the blocking calls (its feature step "
                    "needs the neighbours' halo between the frames)");
    const size_t rowlen = (size_t)c->p.width * c->p.channels;
    if (row_stride == 0) row_stride = rowlen;
    if (row_stride < rowlen) return fail(c, ADDER_E_BAD_PARAMS, "row_stride_bytes smaller than a row");
    HIPCHK(c, hipSetDevice(c->device));
    AdderHipCtx::FrameSlot &fs = c->fslot[c->f_submitted % c->f_slots];
    const size_t need = direct_out ? std::min(direct_cap, worst_case_events_per_frame(c, time_spanned))
                                   : frame_slot_events(c, time_spanned);
    int rc = frame_slot_prepare(c, fs, need, direct_out == nullptr);
    if (rc != ADDER_OK) return rc;
    if (!c->out_s) {
        if (const char *e = getenv("ADDER_HIP_RING_SKIP_STREAMS")) {  // (diagnostic: shifts where the ring's streams land among the runtime's hardware queues)
            for (int k = 0; k < atoi(e); ++k) {
                hipStream_t dummy = nullptr;
                HIPCHK(c, hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking));
                c->dummy_streams.push_back(dummy);
            }
        }
        if (const char *e = getenv("ADDER_HIP_RING_PRIO")) {  // (diagnostic: 1 = out_s, 2 = in_s, 3 = both as high-priority streams)
            int lo_prio = 0, hi_prio = 0;
            HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
            const int m = atoi(e);
            if (m & 1) HIPCHK(c, hipStreamCreateWithPriority(&c->out_s, hipStreamNonBlocking, hi_prio));
            if ((m & 2) && !c->in_s) HIPCHK(c, hipStreamCreateWithPriority(&c->in_s, hipStreamNonBlocking, hi_prio));
            if (m & 4) HIPCHK(c, hipStreamCreateWithPriority(&c->out_s, hipStreamNonBlocking, lo_prio));
            if ((m & 8) && !c->in_s) HIPCHK(c, hipStreamCreateWithPriority(&c->in_s, hipStreamNonBlocking, lo_prio));
        }
        if (!c->out_s) HIPCHK(c, hipStreamCreateWithFlags(&c->out_s, hipStreamNonBlocking));
    }
    if (!c->frame_e) HIPCHK(c, hipEventCreateWithFlags(&c->frame_e, hipEventDisableTiming));
    // The ring's uploads go on a stream of their own, so that the next frame's 2 MB cross the link beside this frame's
    // kernels instead of behind them: 220 -> 84 us per 1080p frame at the default quality (sparse frames: real video).
    // Frames dense with events are bound by their events' way to the host and lose 3 % to the contention (194 -> 200 us
    // at crf 0).  Always, not by the frames' density: a context whose uploads change their HIP stream back and forth stays
    // slow for life (194 -> 238-330 us per dense frame after four uploads on the other stream; `profiles/r04_ring_ab.txt`).
    // ADDER_HIP_RING_DIRECT_WIRE=1 (wire format): the expansion itself stores the records into the slot's page-locked
    // buffer (adder_hip_integrate_wire_device's path, no hand-over pass) -- measured no better for sparse frames and
    // 5 % worse for dense ones (the byte stores at its waves' edges are single PCIe writes): off by default.
    static const bool direct_wire_on = env_flag("ADDER_HIP_RING_DIRECT_WIRE");
    static const bool own_upload_off = env_flag("ADDER_HIP_RING_ONE_STREAM");
    const bool wire_direct = direct_wire_on && c->f_wire && !direct_out && !c->continuous && !feature_path(c);
    const bool own_upload = !direct_out && !own_upload_off;
    hipStream_t up = c->stream;
    if (own_upload) {
        static const bool upload_on_out = env_flag("ADDER_HIP_RING_UPLOAD_ON_OUT");  // (diagnostic: uploads share the hand-over's stream)
        if (!c->in_s && upload_on_out) c->in_s = c->out_s;
        if (!c->in_s) HIPCHK(c, hipStreamCreateWithFlags(&c->in_s, hipStreamNonBlocking));
        if (!c->in_e) HIPCHK(c, hipEventCreateWithFlags(&c->in_e, hipEventDisableTiming));
        up = c->in_s;
    }
    // the ring's stream arrangement for this frame (AdderHipCtx::RingCand): the chosen one, or the candidate being measured
    static const bool ring_tune_off = env_flag("ADDER_HIP_RING_NO_TUNE");
    static const bool post_on_main_env = env_flag("ADDER_HIP_RING_POST_ON_MAIN");
    hipStream_t ring_out = c->out_s;
    bool post_on_main = post_on_main_env;
    if (own_upload && !ring_tune_off && !post_on_main_env) {
        if (c->ring_cands.empty()) {
            // the context's own pair, post-processing on the context's stream, and three more pairs -- a spare stream between
            // two pairs, so that the pairs sit at every offset of the runtime's round robin over its (four) hardware queues
            const int n_cands = 5;
            c->ring_cands.resize(n_cands);
            c->ring_cands[0].out_s = c->out_s; c->ring_cands[0].in_s = c->in_s;
            c->ring_cands[1].out_s = c->out_s; c->ring_cands[1].in_s = c->in_s; c->ring_cands[1].post_on_main = true;
            for (int k = 2; k < n_cands; ++k) {
                hipStream_t t[3] = {nullptr, nullptr, nullptr};
                for (hipStream_t &x : t) {
                    HIPCHK(c, hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
                    c->ring_streams.push_back(x);
                }
                c->ring_cands[k].out_s = t[1];  // (t[0]: the spare)
                c->ring_cands[k].in_s = t[2];
            }
            c->ring_cur = 0;
            c->ring_win_frames = c->ring_win_waits = 0;
            c->ring_win_skip = 6;
        }
        const AdderHipCtx::RingCand &rc_ = c->ring_cands[c->ring_chosen >= 0 ? c->ring_chosen : c->ring_cur];
        up = rc_.in_s;
        ring_out = rc_.out_s;
        post_on_main = rc_.post_on_main;
    }
    static const bool ring_timing = env_flag("ADDER_HIP_RING_TIMING");
    static const bool ring_tl = [] { const char *e = getenv("ADDER_HIP_RING_TIMING"); return e && atoi(e) == 2; }();
    hipEvent_t *tl = nullptr;
    if (ring_tl) {
        g_ring_tl.make();
        tl = g_ring_tl.ev[g_ring_tl.n % RingGpuTimeline::kFrames];
        HIPCHK(c, hipEventRecord(tl[0], up));
    }
    auto rt_ = std::chrono::steady_clock::now();
    if (row_stride == rowlen)
        HIPCHK(c, hipMemcpyAsync(fs.d_frame, frame, c->n_units, hipMemcpyHostToDevice, up));
    else
        HIPCHK(c, hipMemcpy2DAsync(fs.d_frame, rowlen, frame, row_stride, rowlen, c->rows, hipMemcpyHostToDevice, up));
    RING_T(0);
    if (tl) HIPCHK(c, hipEventRecord(tl[1], up));
    if (own_upload) {
        HIPCHK(c, hipEventRecord(c->in_e, up));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->in_e, 0));
    }
    RING_T(1);
    if (tl) HIPCHK(c, hipEventRecord(tl[2], c->stream));
    // the slot's own batch description: the shared one may still be read by the copy engine for the frame before
    struct Swap {
        AdderHipCtx *c;
        BatchArgs *db, *hb;
        FrameTab *dt, *ht;
        size_t cap;
        bool graph, snap;
        uint32_t *counters;
        ~Swap() {
            c->d_batch = db; c->h_batch = hb; c->d_ftab = dt; c->h_ftab = ht; c->ftab_cap = cap;
            c->use_graph = graph; c->no_snapshot = snap; c->d_feat_counters = counters;
        }
    } swap{c, c->d_batch, c->h_batch, c->d_ftab, c->h_ftab, c->ftab_cap, c->use_graph, c->no_snapshot, c->d_feat_counters};
    c->d_feat_counters = fs.d_counters;
    c->d_batch = fs.d_batch;
    c->h_batch = fs.h_batch;
    c->d_ftab = fs.d_ftab;
    c->h_ftab = fs.h_ftab;
    c->ftab_cap = 64;
    // Only the frame kernel runs on the context's stream; scan, expansion and hand-over of this frame go to the ring's
    // second stream as ONE captured launch per slot and kernel choice (the graph holds the slot's own description), beside
    // the next frame's kernel -- which needs this frame's pixel state and nothing else of it.  The per-frame path is bound
    // by the frames' GPU time in a row at the reference's default quality (kernel 29 us + scan / expansion / hand-over
    // 40 us per 1080p frame: 82 us sustained); side by side the period is the longer of the two.
    // (ADDER_HIP_RING_NO_SPLIT=1: everything on one stream, eager, as before.)
    static const bool ring_split_off = env_flag("ADDER_HIP_RING_NO_SPLIT");
    const bool ring_graph = swap.graph && !ring_split_off;
    c->use_graph = false;   // (the context's captured graphs point at the shared description)
    struct SlotTag {
        AdderHipCtx *c;
        ~SlotTag() { c->graph_slot = 0u; c->frame_only = false; }
    } slot_tag{c};
    c->graph_slot = 1u + (uint32_t)(c->f_submitted % c->f_slots);
    // (a ring chunk of scratch per frame slot -- its share of the frame totals must also hold the scan's tile sums, 2 words per
    // tile: planes beyond 16 384 segments with a chunk shortened by the memory budget keep the one-stream form)
    const uint32_t scan_tiles = (c->num_waves + kScanTileWaves - 1u) / kScanTileWaves;
    c->frame_only = !ring_split_off && c->f_slots <= c->ring_chunks && c->chunk >= 2u && c->chunk >= 2u * scan_tiles;
    c->no_snapshot = true;  // frames behind this one are submitted before its outcome is known: no rollback
    fs.out = direct_out ? direct_out : fs.h_events;
    fs.out_cap = need;
    c->wire_batch = wire_direct;
    rc = enqueue_frames(c, fs.d_frame, 1, time_spanned, wire_direct ? fs.out : fs.d_events, need, fs.d_offsets, c->stream);
    c->wire_batch = false;
    if (rc != ADDER_OK) {
        c->poisoned = true;
        return rc;
    }
    RING_T(2);
    if (tl) HIPCHK(c, hipEventRecord(tl[3], c->stream));
    // where the frame's scan / expansion / hand-over go: the ring's second stream (behind an event), or -- post_on_main -- the
    // context's own stream behind the frame kernel (no cross-stream dependency at all)
    hipStream_t post_s = post_on_main ? c->stream : ring_out;
    if (!post_on_main) {
        HIPCHK(c, hipEventRecord(c->frame_e, c->stream));
        HIPCHK(c, hipStreamWaitEvent(ring_out, c->frame_e, 0));
    }
    RING_T(3);
    const bool wire = c->f_wire && !direct_out;
    fs.wire = wire;
    // the hand-over: wire scatter (9 / 11-byte records straight into the slot: 25 % fewer bytes over PCIe, and what the raw
    // sink writes) + frame_out (result header, chunk offsets, the AdderEvents copy)
    const bool split = c->frame_only_done;  // (feature mode and timed launches queued the whole pipeline themselves)
    const uint32_t post_variant = c->last_variant;
    auto hand_over = [&](hipStream_t hs) -> int {
        if (split) {  // the frame's scan (a batch of one frame: it writes the offsets too) and expansion
            HIPCHK(c, adder_launch_scan(fs.d_batch, 0u, 1u, c->num_waves, hs, 1u));
            HIPCHK(c, adder_launch_expand(fs.d_batch, 0u, 1u, c->num_waves, post_variant, 0u, hs));
        }
        if (wire && !wire_direct)
            HIPCHK(c, adder_launch_wire_scatter(reinterpret_cast<const AdderEventPod *>(fs.d_events), fs.d_offsets, 1u,
                                                c->d_side_words + 2, wire_record_bytes(c), reinterpret_cast<uint8_t *>(fs.out),
                                                (uint64_t)fs.out_cap * sizeof(AdderEvent), 0ull, c->status, wire_scatter_blocks(c), hs,
                                                (uint64_t)need));  // (a frame that overflowed its slot: only what the expansion kept is read)
        HIPCHK(c, adder_launch_frame_out(reinterpret_cast<const AdderEventPod *>(wire_direct ? fs.out : fs.d_events), fs.d_offsets, fs.out_cap,
                                         wire ? nullptr : reinterpret_cast<AdderEventPod *>(fs.out), reinterpret_cast<FrameResult *>(fs.h_hdr),
                                         reinterpret_cast<uint32_t *>(fs.h_hdr + sizeof(FrameResult)), c->status,
                                         feature_path(c) ? fs.d_counters : nullptr, c->p.row_begin, c->p.chunk_rows, c->num_chunks, hs,
                                         wire_direct ? wire_record_bytes(c) : 0u));
        return ADDER_OK;
    };
    if (ring_graph && !direct_out) {  // ... as one launch too: captured per slot, again whenever something it baked in has changed
                                      // (the blocking call's destination is the caller's: not worth a capture per buffer)
        const uint64_t key[6] = {(uint64_t)(uintptr_t)fs.out, (uint64_t)fs.out_cap, (uint64_t)need,
                                 (uint64_t)(wire ? 1u : 0u) | (wire_direct ? 2u : 0u) | (feature_path(c) ? 4u : 0u) | (split ? 8u : 0u) |
                                     ((uint64_t)c->p.chunk_rows << 8) | ((uint64_t)post_variant << 32),
                                 (uint64_t)(uintptr_t)fs.d_events, (uint64_t)wire_scatter_blocks(c)};
        if (!fs.out_graph || memcmp(key, fs.out_key, sizeof key) != 0) {
            if (fs.out_graph) c->retired_execs.push_back(fs.out_graph);
            fs.out_graph = nullptr;
            hipGraph_t graph = nullptr;
            HIPCHK(c, hipStreamBeginCapture(c->cap_s, hipStreamCaptureModeThreadLocal));
            const int rc_h = hand_over(c->cap_s);
            hipError_t e = hipStreamEndCapture(c->cap_s, &graph);
            if (rc_h != ADDER_OK) {
                if (graph) (void)hipGraphDestroy(graph);
                c->poisoned = true;
                return rc_h;
            }
            HIPCHK(c, e);
            e = hipGraphInstantiate(&fs.out_graph, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            HIPCHK(c, e);
            memcpy(fs.out_key, key, sizeof key);
        }
        HIPCHK(c, hipGraphLaunch(fs.out_graph, post_s));
    } else {
        rc = hand_over(post_s);
        if (rc != ADDER_OK) {
            c->poisoned = true;
            return rc;
        }
    }
    RING_T(4);
    HIPCHK(c, hipEventRecord(fs.done, post_s));
    RING_T(5);
    if (tl) {
        HIPCHK(c, hipEventRecord(tl[4], post_s));
        g_ring_tl.n += 1;
    }
    if (ring_timing) g_ring_timing.n += 1;
    c->f_submitted += 1;
    return ADDER_OK;
}

// Waiting for a frame slot's `done` event: polled for a while before the thread is parked -- an event wait that parks wakes
// up tens of microseconds late (adder_hip_finish spins on its result flag for the same reason), and at one frame per 60 us
// that is a third of the period.  ADDER_HIP_RING_NO_SPIN=1: park at once.
static hipError_t wait_done_event(hipEvent_t e) {
    static const bool no_spin = env_flag("ADDER_HIP_RING_NO_SPIN");
    if (!no_spin) {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        uint32_t polls = 0;
        for (;;) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            if ((++polls & 15u) == 0u && std::chrono::steady_clock::now() >= t_end) break;
        }
    }
    return hipEventSynchronize(e);
}

static int frame_collect_impl(AdderHipCtx *c, const AdderEvent **events, size_t *n_events, const uint32_t **chunk_offsets) {
    if (c->f_collected == c->f_submitted) return fail(c, ADDER_E_BAD_PARAMS, "no frame in flight");
    HIPCHK(c, hipSetDevice(c->device));
    AdderHipCtx::FrameSlot &fs = c->fslot[c->f_collected % c->f_slots];
    if (c->ring_chosen < 0 && !c->ring_cands.empty()) {
        // the candidate's window: frames per wall-clock second while the caller keeps the ring full (a collect that has to
        // wait means the GPU side is the bound; a caller paced by its source never measures, and needs no choice)
        constexpr uint32_t kWindow = 24;
        const auto t0 = std::chrono::steady_clock::now();
        HIPCHK(c, wait_done_event(fs.done));
        const auto t1 = std::chrono::steady_clock::now();
        if (c->ring_win_skip) {  // (frames queued under the candidate before are still draining)
            if (--c->ring_win_skip == 0u) {
                c->ring_win_t0 = t1;
                c->ring_win_frames = c->ring_win_waits = 0;
            }
        } else {
            c->ring_win_frames += 1;
            if (std::chrono::duration<double, std::micro>(t1 - t0).count() > 3.0) c->ring_win_waits += 1;
            if (c->ring_win_frames == kWindow) {
                if (c->ring_win_waits * 2u >= kWindow) {
                    c->ring_cands[c->ring_cur].us = std::chrono::duration<double, std::micro>(t1 - c->ring_win_t0).count() / kWindow;
                    if (++c->ring_cur == (int)c->ring_cands.size()) {
                        int best = 0;
                        for (int k = 1; k < (int)c->ring_cands.size(); ++k)
                            if (c->ring_cands[k].us < c->ring_cands[best].us) best = k;
                        c->ring_chosen = best;
                        if (getenv("ADDER_HIP_DEBUG_TUNE")) {
                            fprintf(stderr, "[adder_hip] ring stream arrangements (us per frame):");
                            for (const auto &rc_ : c->ring_cands) fprintf(stderr, " %.1f", rc_.us);
                            fprintf(stderr, " -> %d\n", best);
                        }
                    }
                }
                c->ring_win_skip = 6;  // (next candidate, or the same one again when the caller was the bound)
            }
        }
    } else {
        HIPCHK(c, wait_done_event(fs.done));
    }
    c->f_collected += 1;
    const FrameResult *res = reinterpret_cast<const FrameResult *>(fs.h_hdr);
    c->last_new_features = res->new_features;
    c->f_last_produced = res->produced;
    if (events) *events = fs.out;
    if (n_events) *n_events = (size_t)res->produced;
    if (chunk_offsets) *chunk_offsets = reinterpret_cast<const uint32_t *>(fs.h_hdr + sizeof(FrameResult));
    if (res->status == kStatusCapacity) {
        c->poisoned = true;  // later frames were stepped on top of this one: there is nothing to roll back to
        return fail(c, ADDER_E_OUT_CAPACITY, "frame slot too small: the frame produced %llu events, the slot holds %zu "
                    "(adder_hip_frames_configure)", (unsigned long long)res->produced, fs.out_cap);
    }
    return status_to_code(c, res->status);
}

extern "C" int adder_hip_frames_configure(AdderHipCtx *c, uint32_t slots, size_t events_per_slot) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->f_submitted != c->f_collected) return fail(c, ADDER_E_BAD_PARAMS, "frames are in flight");
    if (slots > 4) return fail(c, ADDER_E_BAD_PARAMS, "at most 4 frame slots");
    c->f_slots = slots ? slots : 3u;
    c->f_events_per_slot = events_per_slot;
    return ADDER_OK;
}

extern "C" int adder_hip_frame_submit(AdderHipCtx *c, const uint8_t *frame, size_t row_stride, float time_spanned) {
    if (!c) return ADDER_E_BAD_PARAMS;
    return frame_submit_impl(c, frame, row_stride, time_spanned, nullptr, 0);
}

extern "C" int adder_hip_frame_collect(AdderHipCtx *c, const AdderEvent **events, size_t *n_events,
                                       const uint32_t **chunk_offsets) {
    if (!c) return ADDER_E_BAD_PARAMS;
    // (the slot's own format, as it was submitted: reading packed 9 / 11-byte records as 12-byte events would run past them)
    if (c->f_collected != c->f_submitted && c->fslot[c->f_collected % c->f_slots].wire)
        return fail(c, ADDER_E_BAD_PARAMS, "the oldest frame in flight holds wire records: adder_hip_frame_collect_wire");
    return frame_collect_impl(c, events, n_events, chunk_offsets);
}

extern "C" int adder_hip_frames_set_format(AdderHipCtx *c, int wire_records) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (c->f_submitted != c->f_collected) return fail(c, ADDER_E_BAD_PARAMS, "frames are in flight");
    c->f_wire = wire_records != 0;
    return ADDER_OK;
}
extern "C" int adder_hip_frame_collect_wire(AdderHipCtx *c, const uint8_t **bytes, size_t *n_bytes, size_t *n_events,
                                            const uint32_t **chunk_offsets) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (!c->f_wire) return fail(c, ADDER_E_BAD_PARAMS, "the ring hands out AdderEvents (adder_hip_frames_set_format)");
    if (c->f_collected != c->f_submitted && !c->fslot[c->f_collected % c->f_slots].wire)
        return fail(c, ADDER_E_BAD_PARAMS, "the oldest frame in flight holds AdderEvents: adder_hip_frame_collect");
    const AdderEvent *ev = nullptr;
    size_t n = 0;
    int rc = frame_collect_impl(c, &ev, &n, chunk_offsets);
    if (bytes) *bytes = reinterpret_cast<const uint8_t *>(ev);
    if (n_events) *n_events = n;
    if (n_bytes) *n_bytes = n * wire_record_bytes(c);
    return rc;
}

extern "C" uint32_t adder_hip_frames_in_flight(const AdderHipCtx *c) {
    return c ? (uint32_t)(c->f_submitted - c->f_collected) : 0u;
}

extern "C" int adder_hip_integrate(AdderHipCtx *c, const uint8_t *frame, size_t row_stride, float time_spanned,
                                   AdderEvent *out, size_t out_cap, size_t *n_out, uint32_t *chunk_offsets) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (n_out) *n_out = 0;
    // a buffer that holds the frame's worst case needs no rollback: submit + collect, the events sent straight
    // into `out` when the device can see it (adder_hip_alloc_pinned / hipHostRegister), else through a slot
    if (out && c->f_submitted == c->f_collected && !c->pending && c->submitted == c->collected && !c->poisoned &&
        !band_features(c) && time_spanned >= 0.0f && out_cap >= worst_case_events_per_frame(c, time_spanned)) {
        void *dev = nullptr;
        const bool direct = device_visible_host(out, &dev);
        int rc = frame_submit_impl(c, frame, row_stride, time_spanned, direct ? (AdderEvent *)dev : nullptr, out_cap);
        if (rc != ADDER_OK) return rc;
        const AdderEvent *ev = nullptr;
        const uint32_t *ch = nullptr;
        size_t n = 0;
        rc = frame_collect_impl(c, &ev, &n, &ch);
        if (n_out) *n_out = n;
        if (rc != ADDER_OK) return rc;
        if (!direct && n) memcpy(out, ev, n * sizeof(AdderEvent));
        if (chunk_offsets) memcpy(chunk_offsets, ch, ((size_t)c->num_chunks + 1) * sizeof(uint32_t));
        return ADDER_OK;
    }
    size_t n = 0;
    int rc = adder_hip_integrate_batch(c, frame, 1, 0, row_stride, time_spanned, out, out_cap, &n, nullptr);
    if (n_out) *n_out = n;
    if (rc != ADDER_OK) return rc;
    if (chunk_offsets) {
        // events are in raster order, so chunk boundaries are boundaries in y: a binary search per chunk
        size_t lo = 0;
        for (uint32_t ch = 0; ch < c->num_chunks; ++ch) {
            const uint32_t y0 = c->p.row_begin + ch * c->p.chunk_rows;
            size_t hi = n;
            while (lo < hi) {
                const size_t mid = lo + (hi - lo) / 2;
                if (out[mid].y < y0)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            chunk_offsets[ch] = (uint32_t)lo;
        }
        chunk_offsets[c->num_chunks] = (uint32_t)n;
    }
    return ADDER_OK;
}

// ------------------------------------------------------------------------------------------
// Sparse steps of event-camera sources (adder_sparse.hip): integrate_for_px(px, &mut 0, frame_val, intensity,
// time) per step, in the caller's order; all events into one buffer.
// ------------------------------------------------------------------------------------------
static int sparse_prepare(AdderHipCtx *c, size_t n, hipStream_t s) {
    if (!c->continuous) return fail(c, ADDER_E_BAD_PARAMS, "sparse steps need a Mode::Continuous context (pixel_mode)");
    if (n > 0x7fffffffull) return fail(c, ADDER_E_BAD_PARAMS, "too many steps for one call");
    if (!c->sparse_mode) {
        // c_thresh, its counter and running_t advance per integrate call: from here on they differ between pixels
        if (!c->cth_px) {
            HIPCHK(c, dalloc(&c->cth_px, c->n_pad));
            HIPCHK(c, dalloc(&c->cctr_px, c->n_pad));
        }
        if (!c->rt_px) HIPCHK(c, dalloc(&c->rt_px, c->n_pad));
        HIPCHK(c, hipMemsetAsync(c->cth_px, c->c_thresh, c->n_pad, s));
        HIPCHK(c, hipMemsetAsync(c->cctr_px, c->c_counter, c->n_pad, s));
        uint32_t bits;
        memcpy(&bits, &c->running_t, sizeof bits);
        HIPCHK(c, adder_launch_fill_u32(reinterpret_cast<uint32_t *>(c->rt_px), c->n_pad, bits, s));
        c->sparse_mode = true;
    }
    AdderHipCtx::SparseWork &w = c->sw;
    if (w.cap < n) {
        for (void *p : {(void *)w.steps, (void *)w.keys0, (void *)w.keys1, (void *)w.idx0, (void *)w.idx1, (void *)w.count,
                        (void *)w.offs, (void *)w.stage, w.temp})
            if (p) HIPCHK(c, hipFree(p));
        w = AdderHipCtx::SparseWork{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, w.total, 0, 0};
        const size_t cap = std::max<size_t>(n, 4096);
        HIPCHK(c, dalloc(&w.steps, cap));
        HIPCHK(c, dalloc(&w.keys0, cap));
        HIPCHK(c, dalloc(&w.keys1, cap));
        HIPCHK(c, dalloc(&w.idx0, cap));
        HIPCHK(c, dalloc(&w.idx1, cap));
        HIPCHK(c, dalloc(&w.count, cap));
        HIPCHK(c, dalloc(&w.offs, cap));
        HIPCHK(c, dalloc(&w.stage, cap * (c->max_depth + 3u)));
        w.temp_bytes = adder_sparse_temp_bytes((uint32_t)cap);
        HIPCHK(c, hipMalloc(&w.temp, w.temp_bytes));
        w.cap = cap;
    }
    if (!w.total) HIPCHK(c, dalloc(&w.total, 1));
    return ADDER_OK;
}

static SparseArgs sparse_args(AdderHipCtx *c) {
    SparseArgs a{};
    a.hdr = c->hdr;
    a.lastf = c->lastf;
    a.cn_integ = c->cn_integ;
    a.cn_dt = c->cn_dt;
    a.cn_bdt = c->cn_bdt;
    a.cn_meta = c->cn_meta;
    a.plane_stride = c->n_pad;
    a.cth_px = c->cth_px;
    a.cctr_px = c->cctr_px;
    a.rt_px = c->rt_px;
    a.running = c->running_enabled ? c->running : nullptr;
    a.status = c->status;
    a.c_max = c->p.c_thresh_max;
    a.c_vel = c->p.c_increase_velocity;
    a.max_nodes = c->max_depth + 1u;
    a.stage_events = c->max_depth + 3u;
    a.width = c->p.width;
    a.channels = c->p.channels;
    a.row_begin = c->p.row_begin;
    a.rows = c->rows;
    a.sc = make_consts(c, 0.0f);
    return a;
}

// d_steps / d_out: device memory; synchronises `stream` to hand back the count
extern "C" int adder_hip_integrate_sparse_device(AdderHipCtx *c, const AdderSparseStep *d_steps, size_t n, AdderEvent *d_out,
                                                 size_t out_cap, size_t *n_out, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (n_out) *n_out = 0;
    if (c->poisoned) return fail(c, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", c->err.c_str());
    if (c->pending || c->f_submitted != c->f_collected) return fail(c, ADDER_E_BAD_PARAMS, "work is in flight on this context");
    if ((!d_steps && n) || (!d_out && out_cap)) return fail(c, ADDER_E_BAD_PARAMS, "null pointer");
    if (n == 0) return ADDER_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (c->reset_pending) {
        HIPCHK(c, hipStreamWaitEvent(s, c->reset_e, 0));
        c->reset_pending = false;
    }
    int rc = sparse_prepare(c, n, s);
    if (rc != ADDER_OK) return rc;
    if (c->running_enabled && !c->running) {
        int rc_ = alloc_running(c, s);
        if (rc_ != ADDER_OK) return rc_;
    }
    const SparseArgs a = sparse_args(c);
    AdderHipCtx::SparseWork &w = c->sw;
    HIPCHK(c, adder_sparse_run(&a, reinterpret_cast<const SparseStep *>(d_steps), (uint32_t)n, w.keys0, w.keys1, w.idx0, w.idx1,
                               w.temp, w.temp_bytes, w.stage, w.count, w.offs, reinterpret_cast<AdderEventPod *>(d_out),
                               out_cap, w.total, s));
    unsigned long long total = 0;
    uint32_t st = 0;
    HIPCHK(c, hipMemcpyAsync(&total, w.total, sizeof total, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&st, c->status, sizeof st, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (n_out) *n_out = (size_t)total;
    c->frames_done += 1;
    if (st & kStatusCapacity) {
        c->poisoned = true;  // the pixels have been stepped: there is no undo on this route
        return fail(c, ADDER_E_OUT_CAPACITY, "event buffer too small: the steps produced %llu events (a buffer of "
                    "n * (max_depth + 3) events cannot overflow)", total);
    }
    return status_to_code(c, st);
}

extern "C" int adder_hip_integrate_sparse(AdderHipCtx *c, const AdderSparseStep *steps, size_t n, AdderEvent *out,
                                          size_t out_cap, size_t *n_out) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (n_out) *n_out = 0;
    if (!steps && n) return fail(c, ADDER_E_BAD_PARAMS, "steps is null");
    if (!out && out_cap) return fail(c, ADDER_E_BAD_PARAMS, "out is null");
    if (c->poisoned) return fail(c, ADDER_E_POISONED, "context is poisoned by an earlier failure: %s", c->err.c_str());
    if (c->pending || c->f_submitted != c->f_collected || c->submitted != c->collected)
        return fail(c, ADDER_E_BAD_PARAMS, "work is in flight on this context");
    if (n == 0) return ADDER_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->reset_pending) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->reset_e, 0));
        c->reset_pending = false;
    }
    int rc = sparse_prepare(c, n, c->stream);
    if (rc != ADDER_OK) return rc;
    void *vp = c->d_events;
    if ((rc = ensure(c, &vp, &c->d_events_cap, std::max<size_t>(out_cap, 1) * sizeof(AdderEvent))) != ADDER_OK) return rc;
    c->d_events = (AdderEvent *)vp;
    HIPCHK(c, hipMemcpyAsync(c->sw.steps, steps, n * sizeof(SparseStep), hipMemcpyHostToDevice, c->stream));
    size_t total = 0;
    rc = adder_hip_integrate_sparse_device(c, reinterpret_cast<const AdderSparseStep *>(c->sw.steps), n, c->d_events, out_cap,
                                           &total, c->stream);
    if (n_out) *n_out = total;
    if (rc != ADDER_OK) return rc;
    if (total && out) HIPCHK(c, hipMemcpy(out, c->d_events, total * sizeof(AdderEvent), hipMemcpyDeviceToHost));
    return ADDER_OK;
}

extern "C" void *adder_hip_alloc_pinned(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void adder_hip_free_pinned(void *p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int adder_hip_enable_running_intensities(AdderHipCtx *c, int enable) {
    if (!c) return ADDER_E_BAD_PARAMS;
    c->running_enabled = enable != 0;
    return ADDER_OK;
}

extern "C" int adder_hip_running_intensities(AdderHipCtx *c, uint8_t *dst) {
    if (!c || !dst) return ADDER_E_BAD_PARAMS;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->running) {  // never enabled before a batch: the plane is still all zeros
        memset(dst, 0, c->n_units);
        return ADDER_OK;
    }
    { int rc_ = settle_reset(c); if (rc_ != ADDER_OK) return rc_; }
    HIPCHK(c, hipMemcpy(dst, c->running, c->n_units, hipMemcpyDeviceToHost));
    return ADDER_OK;
}

extern "C" int adder_hip_selftest_division(uint64_t *mismatches) {
    if (!mismatches) return ADDER_E_BAD_PARAMS;
    unsigned long long *d = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), sizeof(*d));
    if (e == hipSuccess) e = hipMemset(d, 0, sizeof(*d));
    if (e == hipSuccess) e = adder_launch_divtest(d, nullptr);
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    if (d) (void)hipFree(d);
    if (e != hipSuccess) {
        g_create_error = std::string("division self-test failed to run: ") + hipGetErrorString(e);
        return ADDER_E_HIP;
    }
    *mismatches = h;
    return ADDER_OK;
}

extern "C" int adder_hip_synth_clip_device(uint8_t *d_dst, int content, uint64_t seed, uint32_t width,
                                           uint32_t height, uint32_t channels, uint32_t row_begin, uint32_t rows,
                                           uint32_t frame_begin, uint32_t num_frames, void *stream) {
    if (!d_dst || content < 0 || content > 2 || !width || !height || !channels) return ADDER_E_BAD_PARAMS;
    hipError_t e = adder_launch_synth(d_dst, content, seed, width, height, channels, row_begin, rows, frame_begin,
                                      num_frames, (hipStream_t)stream);
    if (e != hipSuccess) {
        g_create_error = std::string("synth launch failed: ") + hipGetErrorString(e);
        return ADDER_E_HIP;
    }
    return ADDER_OK;
}

// ---- multi-GPU: merge of the row bands' streams (include/adder_hip.h) ----
extern "C" size_t adder_hip_merge_work_bytes(uint32_t world, uint32_t num_frames) {
    return ((size_t)num_frames + 1 + (size_t)world * num_frames) * sizeof(uint64_t);
}

// merged_base: events of the merged stream that precede this batch (a chunk of a longer stream: d_out is then the place
// of the chunk's first event, d_merged_offsets the entry of its first frame, and the offsets written continue from there)
extern "C" int adder_hip_merge_streams_device_at(AdderHipCtx *c, const AdderEvent *d_stage, const uint64_t *d_rank_offsets,
                                                 uint32_t world, uint32_t num_frames, void *d_work, AdderEvent *d_out,
                                                 size_t out_cap, uint64_t *d_merged_offsets, uint64_t merged_base, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    if (!d_rank_offsets || !d_work || world == 0 || (!d_out && out_cap))
        return fail(c, ADDER_E_BAD_PARAMS, "merge: null pointer or empty world");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (num_frames == 0) {
        if (d_merged_offsets) HIPCHK(c, hipMemcpyAsync(d_merged_offsets, &merged_base, sizeof(uint64_t), hipMemcpyHostToDevice, s));
        return ADDER_OK;
    }
    HIPCHK(c, adder_launch_merge(reinterpret_cast<const AdderEventPod *>(d_stage), d_rank_offsets, world, num_frames,
                                 reinterpret_cast<uint64_t *>(d_work), reinterpret_cast<AdderEventPod *>(d_out), out_cap,
                                 d_merged_offsets, merged_base, c->status, s));
    return ADDER_OK;
}

extern "C" int adder_hip_merge_streams_device(AdderHipCtx *c, const AdderEvent *d_stage, const uint64_t *d_rank_offsets,
                                              uint32_t world, uint32_t num_frames, void *d_work, AdderEvent *d_out,
                                              size_t out_cap, uint64_t *d_merged_offsets, void *stream) {
    return adder_hip_merge_streams_device_at(c, d_stage, d_rank_offsets, world, num_frames, d_work, d_out, out_cap,
                                             d_merged_offsets, 0ull, stream);
}

extern "C" int adder_hip_check_status(AdderHipCtx *c, void *stream) {
    if (!c) return ADDER_E_BAD_PARAMS;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    uint32_t st = 0;
    HIPCHK(c, hipMemcpyAsync(&st, c->status, sizeof st, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return status_to_code(c, st);
}
