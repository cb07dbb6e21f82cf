// adder_gather.cpp -- libadder_rccl.so: the multi-GPU event-stream gather (include/adder_gather.h).
// RCCL collectives over xGMI + the merge kernel of libadder_hip.so; nothing here computes events.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "../../include/adder_gather.h"

static_assert(sizeof(ncclUniqueId) == ADDER_GATHER_UNIQUE_ID_BYTES, "ncclUniqueId size");

struct AdderGather {
    AdderHipCtx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool owns_comm = false;
    int rank = 0, world = 1, device = 0;
    // device scratch, grown on demand
    uint64_t *d_all_offs = nullptr;  // [world][T+1]
    size_t all_offs_cap = 0;         // bytes
    void *d_work = nullptr;          // merge layout
    size_t work_cap = 0;
    AdderEvent *d_stage = nullptr;   // root: the ranks' streams back to back
    size_t stage_cap = 0;            // bytes
    std::vector<uint64_t> h_offs;    // host copy of d_all_offs
    int *d_flag = nullptr;           // the ranks' agreement that nothing failed before the payload exchange
    // records over the wire: this rank's image of a chunk, root's copies of the peers' images, the exchanged sizes
    static constexpr int kRecSlots = 2;  // images of two chunks in turn (one being expanded, one arriving)
    uint8_t *d_rec_img[kRecSlots] = {nullptr, nullptr};
    size_t rec_img_cap[kRecSlots] = {0, 0};
    std::vector<uint8_t *> d_peer_img[kRecSlots];
    std::vector<size_t> peer_img_cap[kRecSlots];
    uint64_t *d_meta = nullptr;      // [world + 1][8]
    hipEvent_t copy_ev = nullptr;    // the image copy on the batch's stream -> the transport's stream
    hipEvent_t rec_ev[kRecSlots] = {nullptr, nullptr};  // a slot's transfer / expansion is through (before it is rewritten)
    uint32_t rec_calls = 0;
    std::string err;
};

static thread_local std::string g_err;

static int gfail(AdderGather *g, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g)
        g->err = buf;
    else
        g_err = buf;
    return code;
}

#define GHIP(g, expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return gfail(g, ADDER_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define GNCCL(g, expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess)                                                                             \
            return gfail(g, ADDER_E_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

static int grow(AdderGather *g, void **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return ADDER_OK;
    void *old = *p;
    *p = nullptr;
    *cap = 0;
    if (old) GHIP(g, hipFree(old));
    GHIP(g, hipMalloc(p, std::max<size_t>(need, 256)));
    *cap = need;
    return ADDER_OK;
}

extern "C" int adder_gather_unique_id(uint8_t id_out[ADDER_GATHER_UNIQUE_ID_BYTES]) {
    if (!id_out) return ADDER_E_BAD_PARAMS;
    ncclUniqueId id;
    GNCCL(nullptr, ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return ADDER_OK;
}

static int create_common(AdderHipCtx *ctx, int rank, int world, AdderGather **out, AdderGather **g_out) {
    if (!out) return gfail(nullptr, ADDER_E_BAD_PARAMS, "out is null");
    *out = nullptr;
    if (!ctx || world < 1 || rank < 0 || rank >= world) return gfail(nullptr, ADDER_E_BAD_PARAMS, "bad ctx / rank / world");
    AdderGather *g = new (std::nothrow) AdderGather();
    if (!g) return gfail(nullptr, ADDER_E_HIP, "out of host memory");
    g->ctx = ctx;
    g->rank = rank;
    g->world = world;
    if (hipGetDevice(&g->device) != hipSuccess) g->device = 0;
    *g_out = g;
    return ADDER_OK;
}

extern "C" int adder_gather_create(AdderHipCtx *ctx, void *nccl_comm, int rank, int world, AdderGather **out) {
    if (!nccl_comm) return gfail(nullptr, ADDER_E_BAD_PARAMS, "nccl_comm is null");
    AdderGather *g = nullptr;
    int rc = create_common(ctx, rank, world, out, &g);
    if (rc != ADDER_OK) return rc;
    g->comm = (ncclComm_t)nccl_comm;
    *out = g;
    return ADDER_OK;
}

extern "C" int adder_gather_create_from_id(AdderHipCtx *ctx, const uint8_t id[ADDER_GATHER_UNIQUE_ID_BYTES], int rank,
                                           int world, AdderGather **out) {
    if (!id) return gfail(nullptr, ADDER_E_BAD_PARAMS, "id is null");
    AdderGather *g = nullptr;
    int rc = create_common(ctx, rank, world, out, &g);
    if (rc != ADDER_OK) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = ncclCommInitRank(&g->comm, world, uid, rank);
    if (r != ncclSuccess) {
        g_err = std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r);
        delete g;
        return ADDER_E_HIP;
    }
    g->owns_comm = true;
    *out = g;
    return ADDER_OK;
}

static void free_record_buffers(AdderGather *g) {
    for (int k = 0; k < AdderGather::kRecSlots; ++k) {
        if (g->d_rec_img[k]) (void)hipFree(g->d_rec_img[k]);
        for (uint8_t *p : g->d_peer_img[k])
            if (p) (void)hipFree(p);
    }
    if (g->d_meta) (void)hipFree(g->d_meta);
    if (g->copy_ev) (void)hipEventDestroy(g->copy_ev);
    for (hipEvent_t e : g->rec_ev)
        if (e) (void)hipEventDestroy(e);
}
extern "C" void adder_gather_destroy(AdderGather *g) {
    if (!g) return;
    free_record_buffers(g);
    if (g->d_all_offs) (void)hipFree(g->d_all_offs);
    if (g->d_work) (void)hipFree(g->d_work);
    if (g->d_stage) (void)hipFree(g->d_stage);
    if (g->d_flag) (void)hipFree(g->d_flag);
    if (g->owns_comm && g->comm) (void)ncclCommDestroy(g->comm);
    delete g;
}

extern "C" const char *adder_gather_last_error(const AdderGather *g) { return g ? g->err.c_str() : g_err.c_str(); }
extern "C" int adder_gather_world(const AdderGather *g) { return g ? g->world : 0; }

// all-gather of the ranks' frame offsets -> g->d_all_offs [world][T+1] and its host copy
static int gather_offsets(AdderGather *g, const uint64_t *d_frame_offsets, uint32_t T, hipStream_t s) {
    const size_t per = (size_t)T + 1;
    void *p = g->d_all_offs;
    int rc = grow(g, &p, &g->all_offs_cap, per * g->world * sizeof(uint64_t));
    g->d_all_offs = (uint64_t *)p;
    if (rc != ADDER_OK) return rc;
    GNCCL(g, ncclAllGather(d_frame_offsets, g->d_all_offs, per, ncclUint64, g->comm, s));
    g->h_offs.resize(per * g->world);
    GHIP(g, hipMemcpyAsync(g->h_offs.data(), g->d_all_offs, per * g->world * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    GHIP(g, hipStreamSynchronize(s));
    return ADDER_OK;
}

extern "C" int adder_gather_layout(AdderGather *g, const uint64_t *d_frame_offsets, uint32_t T,
                                   uint64_t *h_merged_offsets, uint64_t *h_my_base, void *stream) {
    if (!g || !d_frame_offsets) return gfail(g, ADDER_E_BAD_PARAMS, "null argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = gather_offsets(g, d_frame_offsets, T, s);
    if (rc != ADDER_OK) return rc;
    const size_t per = (size_t)T + 1;
    uint64_t run = 0;
    if (h_merged_offsets) h_merged_offsets[0] = 0;
    for (uint32_t f = 0; f < T; ++f) {
        uint64_t before = 0;
        for (int r = 0; r < g->world; ++r) {
            if (r == g->rank && h_my_base) h_my_base[f] = run + before;
            before += g->h_offs[r * per + f + 1] - g->h_offs[r * per + f];
        }
        run += before;
        if (h_merged_offsets) h_merged_offsets[f + 1] = run;
    }
    return ADDER_OK;
}

// One chunk of the streams: frames [0, T) of what every rank passes, appended to the merged stream behind merged_base
// events.  Every failure that only one rank can see (root's staging / scratch allocations) is agreed on with a one-word
// all-reduce BEFORE the payload exchange, so that no rank is left waiting in a send.
extern "C" int adder_gather_events_at(AdderGather *g, const AdderEvent *d_events, const uint64_t *d_frame_offsets,
                                      uint32_t T, int root, AdderEvent *d_merged, size_t merged_cap, uint64_t merged_base,
                                      uint64_t *d_merged_offsets, size_t *n_merged, void *stream) {
    if (n_merged) *n_merged = 0;
    if (!g || !d_frame_offsets) return gfail(g, ADDER_E_BAD_PARAMS, "null argument");
    if (root < 0 || root >= g->world) return gfail(g, ADDER_E_BAD_PARAMS, "bad root %d", root);
    hipStream_t s = (hipStream_t)stream;
    int rc = gather_offsets(g, d_frame_offsets, T, s);
    if (rc != ADDER_OK) return rc;
    const size_t per = (size_t)T + 1;
    std::vector<uint64_t> tot(g->world), base(g->world + 1, 0);
    for (int r = 0; r < g->world; ++r) {
        tot[r] = g->h_offs[r * per + T] - g->h_offs[r * per];  // (offsets need not start at 0: a chunk of a stream)
        base[r + 1] = base[r] + tot[r];
    }
    const uint64_t total = base[g->world];
    const uint64_t first = g->h_offs[(size_t)g->rank * per];
    // ---- everything that can fail locally, then one agreement ----
    int local_rc = ADDER_OK;
    if (tot[g->rank] != 0 && !d_events) local_rc = gfail(g, ADDER_E_BAD_PARAMS, "d_events is null but the chunk holds events");
    if (g->rank == root && local_rc == ADDER_OK) {
        void *p = g->d_stage;
        local_rc = grow(g, &p, &g->stage_cap, (size_t)total * sizeof(AdderEvent));
        g->d_stage = (AdderEvent *)p;
        if (local_rc == ADDER_OK) {
            p = g->d_work;
            local_rc = grow(g, &p, &g->work_cap, adder_hip_merge_work_bytes((uint32_t)g->world, T));
            g->d_work = p;
        }
    }
    if (g->world > 1) {
        if (!g->d_flag) GHIP(g, hipMalloc(reinterpret_cast<void **>(&g->d_flag), sizeof(int)));
        const int bad = local_rc != ADDER_OK ? 1 : 0;
        int any = 0;
        GHIP(g, hipMemcpyAsync(g->d_flag, &bad, sizeof bad, hipMemcpyHostToDevice, s));
        GNCCL(g, ncclAllReduce(g->d_flag, g->d_flag, 1, ncclInt32, ncclMax, g->comm, s));
        GHIP(g, hipMemcpyAsync(&any, g->d_flag, sizeof any, hipMemcpyDeviceToHost, s));
        GHIP(g, hipStreamSynchronize(s));
        if (any) return local_rc != ADDER_OK ? local_rc : gfail(g, ADDER_E_HIP, "another rank failed before the exchange");
    } else if (local_rc != ADDER_OK) {
        return local_rc;
    }
    // payload: every rank -> root, back to back in rank order (ncclGroup of point-to-point transfers
    // over xGMI; root's own stream is a device copy)
    GNCCL(g, ncclGroupStart());
    if (g->rank == root) {
        for (int r = 0; r < g->world; ++r) {
            if (r == root || tot[r] == 0) continue;
            GNCCL(g, ncclRecv(g->d_stage + base[r], (size_t)tot[r] * sizeof(AdderEvent), ncclUint8, r, g->comm, s));
        }
    } else if (tot[g->rank] != 0) {
        GNCCL(g, ncclSend(d_events + first, (size_t)tot[g->rank] * sizeof(AdderEvent), ncclUint8, root, g->comm, s));
    }
    GNCCL(g, ncclGroupEnd());
    if (g->rank != root) {
        GHIP(g, hipStreamSynchronize(s));
        return ADDER_OK;
    }
    if (tot[root])
        GHIP(g, hipMemcpyAsync(g->d_stage + base[root], d_events + first, (size_t)tot[root] * sizeof(AdderEvent),
                               hipMemcpyDeviceToDevice, s));
    if (n_merged) *n_merged = (size_t)total;
    if (merged_base > merged_cap || total > merged_cap - merged_base) {
        GHIP(g, hipStreamSynchronize(s));
        return gfail(g, ADDER_E_OUT_CAPACITY, "merged buffer too small: need %llu events",
                     (unsigned long long)(merged_base + total));
    }
    rc = adder_hip_merge_streams_device_at(g->ctx, g->d_stage, g->d_all_offs, (uint32_t)g->world, T, g->d_work,
                                           d_merged ? d_merged + merged_base : nullptr, merged_cap - merged_base,
                                           d_merged_offsets, merged_base, s);
    if (rc != ADDER_OK) return gfail(g, rc, "merge: %s", adder_hip_last_error(g->ctx));
    rc = adder_hip_check_status(g->ctx, s);
    if (rc != ADDER_OK) return gfail(g, rc, "merge: %s", adder_hip_last_error(g->ctx));
    return ADDER_OK;
}

extern "C" int adder_gather_events(AdderGather *g, const AdderEvent *d_events, const uint64_t *d_frame_offsets,
                                   uint32_t T, int root, AdderEvent *d_merged, size_t merged_cap,
                                   uint64_t *d_merged_offsets, size_t *n_merged, void *stream) {
    return adder_gather_events_at(g, d_events, d_frame_offsets, T, root, d_merged, merged_cap, 0ull, d_merged_offsets,
                                  n_merged, stream);
}

// Records over the wire (include/adder_hip.h): every rank passes the chunk it has just integrated with
// adder_hip_integrate_records_device + adder_hip_finish; the ranks exchange the sizes, the peers send ONE contiguous image
// each (adder_hip_records_to_wire) to root, root expands every band -- its own included -- behind merged_base.  Images of
// two chunks are kept in turn, so that the caller may integrate its next chunk (and call this again on another stream)
// while this one's expansion runs.
extern "C" int adder_gather_records_at(AdderGather *g, const AdderBandRecords *rec, uint64_t n_records, uint64_t n_events,
                                       int root, AdderEvent *d_merged, size_t merged_cap, uint64_t merged_base,
                                       uint64_t *d_merged_offsets, size_t *n_merged, void *stream) {
    if (n_merged) *n_merged = 0;
    if (!g || !rec) return gfail(g, ADDER_E_BAD_PARAMS, "null argument");
    if (root < 0 || root >= g->world) return gfail(g, ADDER_E_BAD_PARAMS, "bad root %d", root);
    hipStream_t s = (hipStream_t)stream;
    const int slot = (int)(g->rec_calls++ % AdderGather::kRecSlots);
    const uint32_t nf = rec->num_frames;
    // ---- the sizes ----
    if (!g->d_meta) GHIP(g, hipMalloc(reinterpret_cast<void **>(&g->d_meta), ((size_t)g->world + 1) * 8 * sizeof(uint64_t)));
    uint64_t mine[8] = {nf, rec->num_segments, rec->record_bytes, n_records, rec->row_begin, rec->rows, n_events, 0};
    std::vector<uint64_t> all((size_t)g->world * 8);
    GHIP(g, hipMemcpyAsync(g->d_meta + (size_t)g->world * 8, mine, sizeof mine, hipMemcpyHostToDevice, s));
    GNCCL(g, ncclAllGather(g->d_meta + (size_t)g->world * 8, g->d_meta, 8, ncclUint64, g->comm, s));
    GHIP(g, hipMemcpyAsync(all.data(), g->d_meta, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    GHIP(g, hipStreamSynchronize(s));
    // ---- everything that can fail locally, then one agreement ----
    int local_rc = ADDER_OK;
    uint64_t total = 0;
    for (int r = 0; r < g->world; ++r) {
        if (all[(size_t)r * 8] != nf || all[(size_t)r * 8 + 2] != rec->record_bytes)
            local_rc = gfail(g, ADDER_E_BAD_PARAMS, "rank %d passed a chunk of another length / record size", r);
        total += all[(size_t)r * 8 + 6];
    }
    const size_t my_bytes = adder_hip_records_wire_bytes(nf, rec->num_segments, rec->record_bytes, n_records);
    if (local_rc == ADDER_OK) {
        void *p = g->d_rec_img[slot];
        local_rc = grow(g, &p, &g->rec_img_cap[slot], my_bytes);
        g->d_rec_img[slot] = (uint8_t *)p;
    }
    if (g->rank == root && local_rc == ADDER_OK) {
        g->d_peer_img[slot].resize(g->world, nullptr);
        g->peer_img_cap[slot].resize(g->world, 0);
        for (int r = 0; r < g->world && local_rc == ADDER_OK; ++r) {
            if (r == root) continue;
            void *p = g->d_peer_img[slot][r];
            local_rc = grow(g, &p, &g->peer_img_cap[slot][r],
                            adder_hip_records_wire_bytes(nf, (uint32_t)all[(size_t)r * 8 + 1], (uint32_t)all[(size_t)r * 8 + 2],
                                                         all[(size_t)r * 8 + 3]));
            g->d_peer_img[slot][r] = (uint8_t *)p;
        }
        if (local_rc == ADDER_OK && (merged_base > merged_cap || total > merged_cap - merged_base))
            local_rc = gfail(g, ADDER_E_OUT_CAPACITY, "merged buffer too small: need %llu events",
                             (unsigned long long)(merged_base + total));
    }
    if (local_rc == ADDER_OK) {
        // the image is copied on the BATCH's stream (before the context's next batch can touch its scratch); `stream`
        // waits for the copy
        hipStream_t bs = (hipStream_t)adder_hip_last_batch_stream(g->ctx);
        if (bs != s && g->rec_ev[slot]) GHIP(g, hipStreamWaitEvent(bs, g->rec_ev[slot], 0));  // the slot's last transfer
        local_rc = adder_hip_records_to_wire(g->ctx, rec, n_records, g->d_rec_img[slot], g->rec_img_cap[slot], bs);
        if (local_rc != ADDER_OK) gfail(g, local_rc, "image: %s", adder_hip_last_error(g->ctx));
        else if (bs != s) {
            if (!g->copy_ev) GHIP(g, hipEventCreateWithFlags(&g->copy_ev, hipEventDisableTiming));
            GHIP(g, hipEventRecord(g->copy_ev, bs));
            GHIP(g, hipStreamWaitEvent(s, g->copy_ev, 0));
        }
    }
    if (g->world > 1) {
        if (!g->d_flag) GHIP(g, hipMalloc(reinterpret_cast<void **>(&g->d_flag), sizeof(int)));
        const int bad = local_rc != ADDER_OK ? 1 : 0;
        int any = 0;
        GHIP(g, hipMemcpyAsync(g->d_flag, &bad, sizeof bad, hipMemcpyHostToDevice, s));
        GNCCL(g, ncclAllReduce(g->d_flag, g->d_flag, 1, ncclInt32, ncclMax, g->comm, s));
        GHIP(g, hipMemcpyAsync(&any, g->d_flag, sizeof any, hipMemcpyDeviceToHost, s));
        GHIP(g, hipStreamSynchronize(s));
        if (any) return local_rc != ADDER_OK ? local_rc : gfail(g, ADDER_E_HIP, "another rank failed before the exchange");
    } else if (local_rc != ADDER_OK) {
        return local_rc;
    }
    // ---- the images: every peer -> root, one point-to-point transfer each (its own xGMI link) ----
    GNCCL(g, ncclGroupStart());
    if (g->rank == root) {
        for (int r = 0; r < g->world; ++r) {
            if (r == root) continue;
            GNCCL(g, ncclRecv(g->d_peer_img[slot][r],
                              adder_hip_records_wire_bytes(nf, (uint32_t)all[(size_t)r * 8 + 1], (uint32_t)all[(size_t)r * 8 + 2],
                                                           all[(size_t)r * 8 + 3]),
                              ncclUint8, r, g->comm, s));
        }
    } else {
        GNCCL(g, ncclSend(g->d_rec_img[slot], my_bytes, ncclUint8, root, g->comm, s));
    }
    GNCCL(g, ncclGroupEnd());
    if (!g->rec_ev[slot]) GHIP(g, hipEventCreateWithFlags(&g->rec_ev[slot], hipEventDisableTiming));
    if (g->rank != root) {
        GHIP(g, hipEventRecord(g->rec_ev[slot], s));
        return ADDER_OK;  // (queued: the image is this object's, the caller's context is free)
    }
    // ---- root: every band's description over its image, then the expansion ----
    std::vector<AdderBandRecords> bands(g->world);
    size_t sec[6];
    adder_hip_records_wire_sections(nf, rec->num_segments, rec->record_bytes, sec);
    const void *ftab = g->d_rec_img[slot] + sec[1];  // root's own frame table of these frames
    for (int r = 0; r < g->world; ++r) {
        const uint8_t *img = r == root ? g->d_rec_img[slot] : g->d_peer_img[slot][r];
        AdderBandRecords &b = bands[r];
        b.num_frames = nf;
        b.num_segments = (uint32_t)all[(size_t)r * 8 + 1];
        b.record_bytes = (uint32_t)all[(size_t)r * 8 + 2];
        b.row_begin = (uint32_t)all[(size_t)r * 8 + 4];
        b.rows = (uint32_t)all[(size_t)r * 8 + 5];
        adder_hip_records_wire_sections(nf, b.num_segments, b.record_bytes, sec);
        b.d_frame_offsets = reinterpret_cast<const uint64_t *>(img + sec[0]);
        b.d_frame_table = ftab;
        b.d_counts = reinterpret_cast<const uint32_t *>(img + sec[2]);
        b.d_prefix = reinterpret_cast<const uint32_t *>(img + sec[3]);
        b.d_runs = reinterpret_cast<const uint32_t *>(img + sec[4]);
        b.d_records = img + sec[5];
    }
    int rc = adder_hip_expand_records_device(g->ctx, bands.data(), (uint32_t)g->world, d_merged, merged_cap, merged_base,
                                             d_merged_offsets, s);
    if (rc != ADDER_OK) return gfail(g, rc, "expansion: %s", adder_hip_last_error(g->ctx));
    GHIP(g, hipEventRecord(g->rec_ev[slot], s));
    if (n_merged) *n_merged = (size_t)total;
    return ADDER_OK;
}
