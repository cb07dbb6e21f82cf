// adder_gather.cpp -- libadder_rccl.so: the multi-GPU event-stream gather (include/adder_gather.h).
// RCCL collectives over xGMI + the merge / expansion / scatter kernels of libadder_hip.so; nothing here computes events.
//
// Every exchange goes through an AdderTransport (all-gather, grouped send / recv, one-word all-reduce): RCCL over the
// caller's ncclComm_t in production, an in-process rendezvous between threads (adder_gather_local_*) where a box has
// one GPU and RCCL refuses two ranks on it -- the protocol code above the transport is the same.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/adder_gather.h"

static_assert(sizeof(ncclUniqueId) == ADDER_GATHER_UNIQUE_ID_BYTES, "ncclUniqueId size");

// ---------------------------------------------------------------------------------------------------------------------
// transports
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct RcclTransport {
    ncclComm_t comm = nullptr;
    bool owns = false;
    std::string err;
};
int rccl_fail(RcclTransport *t, const char *what, ncclResult_t r) {
    t->err = std::string(what) + " failed: " + ncclGetErrorString(r);
    return ADDER_E_HIP;
}
int rccl_all_gather(void *self, const void *send, void *recv, size_t bytes, void *stream) {
    auto *t = static_cast<RcclTransport *>(self);
    ncclResult_t r = ncclAllGather(send, recv, bytes, ncclUint8, t->comm, (hipStream_t)stream);
    return r == ncclSuccess ? ADDER_OK : rccl_fail(t, "ncclAllGather", r);
}
int rccl_all_reduce_max(void *self, int32_t *d_word, void *stream) {
    auto *t = static_cast<RcclTransport *>(self);
    ncclResult_t r = ncclAllReduce(d_word, d_word, 1, ncclInt32, ncclMax, t->comm, (hipStream_t)stream);
    return r == ncclSuccess ? ADDER_OK : rccl_fail(t, "ncclAllReduce", r);
}
int rccl_group_start(void *self) {
    auto *t = static_cast<RcclTransport *>(self);
    ncclResult_t r = ncclGroupStart();
    return r == ncclSuccess ? ADDER_OK : rccl_fail(t, "ncclGroupStart", r);
}
int rccl_group_end(void *self, void *) {
    auto *t = static_cast<RcclTransport *>(self);
    ncclResult_t r = ncclGroupEnd();
    return r == ncclSuccess ? ADDER_OK : rccl_fail(t, "ncclGroupEnd", r);
}
int rccl_send(void *self, const void *buf, size_t bytes, int peer, void *stream) {
    auto *t = static_cast<RcclTransport *>(self);
    ncclResult_t r = ncclSend(buf, bytes, ncclUint8, peer, t->comm, (hipStream_t)stream);
    return r == ncclSuccess ? ADDER_OK : rccl_fail(t, "ncclSend", r);
}
int rccl_recv(void *self, void *buf, size_t bytes, int peer, void *stream) {
    auto *t = static_cast<RcclTransport *>(self);
    ncclResult_t r = ncclRecv(buf, bytes, ncclUint8, peer, t->comm, (hipStream_t)stream);
    return r == ncclSuccess ? ADDER_OK : rccl_fail(t, "ncclRecv", r);
}
const char *rccl_error(void *self) { return static_cast<RcclTransport *>(self)->err.c_str(); }

}  // namespace

// In-process transport: the ranks are THREADS of one process (one context each, any devices -- also all on one).  Every
// operation is a blocking rendezvous: the caller's stream is drained, the ranks meet, the bytes move with synchronous
// device copies, the ranks meet again.  A valid (slow) implementation of the transport's contract, so the protocol code
// above it runs unchanged where RCCL cannot (two ranks on one GPU).  A rank that does not turn up within 30 s fails the
// operation on everybody instead of hanging the process.
struct AdderLocalGroup {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    std::vector<const void *> ag_send;                       // all-gather: the ranks' send buffers
    struct P2P { const void *buf; size_t bytes; int peer; };
    std::vector<std::vector<P2P>> sends;                      // [rank]: what it sends in the group being closed
    std::vector<int32_t> words;                               // all-reduce
};
namespace {
bool local_barrier(AdderLocalGroup *g) {
    std::unique_lock<std::mutex> lk(g->m);
    if (g->broken) return false;
    const uint64_t gen = g->generation;
    if (++g->arrived == g->world) {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
        return true;
    }
    if (!g->cv.wait_for(lk, std::chrono::seconds(30), [&] { return g->generation != gen || g->broken; })) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return !g->broken;
}
struct LocalTransport {
    AdderLocalGroup *grp = nullptr;
    int rank = 0;
    struct Recv { void *buf; size_t bytes; int peer; };
    std::vector<Recv> recvs;
    std::string err;
};
int local_fail(LocalTransport *t, const char *what) {
    t->err = std::string(what) + ": a rank of the local group did not arrive (or failed earlier)";
    return ADDER_E_TIMEOUT;
}
int local_all_gather(void *self, const void *send, void *recv, size_t bytes, void *stream) {
    auto *t = static_cast<LocalTransport *>(self);
    AdderLocalGroup *g = t->grp;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return local_fail(t, "all_gather (stream)");
    { std::lock_guard<std::mutex> lk(g->m); g->ag_send[t->rank] = send; }
    if (!local_barrier(g)) return local_fail(t, "all_gather");
    // (device-to-device hipMemcpy does not wait on the host: the copies go onto the RECEIVER's stream, which orders them
    // before whatever it runs next, and that stream is drained before the senders may touch their buffers again)
    for (int r = 0; r < g->world; ++r)
        if (hipMemcpyAsync(static_cast<uint8_t *>(recv) + (size_t)r * bytes, g->ag_send[r], bytes, hipMemcpyDeviceToDevice,
                           (hipStream_t)stream) != hipSuccess)
            return local_fail(t, "all_gather (copy)");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return local_fail(t, "all_gather (copy)");
    if (!local_barrier(g)) return local_fail(t, "all_gather");
    return ADDER_OK;
}
int local_all_reduce_max(void *self, int32_t *d_word, void *stream) {
    auto *t = static_cast<LocalTransport *>(self);
    AdderLocalGroup *g = t->grp;
    int32_t v = 0;
    if (hipMemcpyAsync(&v, d_word, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return local_fail(t, "all_reduce (stream)");
    { std::lock_guard<std::mutex> lk(g->m); g->words[t->rank] = v; }
    if (!local_barrier(g)) return local_fail(t, "all_reduce");
    int32_t mx = g->words[0];
    for (int r = 1; r < g->world; ++r) mx = std::max(mx, g->words[r]);
    if (!local_barrier(g)) return local_fail(t, "all_reduce");
    if (hipMemcpyAsync(d_word, &mx, sizeof mx, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return local_fail(t, "all_reduce (copy)");
    return ADDER_OK;
}
int local_group_start(void *self) {
    auto *t = static_cast<LocalTransport *>(self);
    t->recvs.clear();
    std::lock_guard<std::mutex> lk(t->grp->m);
    t->grp->sends[t->rank].clear();
    return ADDER_OK;
}
int local_send(void *self, const void *buf, size_t bytes, int peer, void *) {
    auto *t = static_cast<LocalTransport *>(self);
    std::lock_guard<std::mutex> lk(t->grp->m);
    t->grp->sends[t->rank].push_back({buf, bytes, peer});
    return ADDER_OK;
}
int local_recv(void *self, void *buf, size_t bytes, int peer, void *) {
    static_cast<LocalTransport *>(self)->recvs.push_back({buf, bytes, peer});
    return ADDER_OK;
}
int local_group_end(void *self, void *stream) {
    auto *t = static_cast<LocalTransport *>(self);
    AdderLocalGroup *g = t->grp;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return local_fail(t, "group_end (stream)");
    if (!local_barrier(g)) return local_fail(t, "group_end");
    int rc = ADDER_OK;
    std::vector<size_t> next(g->world, 0);  // the k-th recv from a peer matches that peer's k-th send to this rank
    for (const auto &rv : t->recvs) {
        const auto &ss = g->sends[rv.peer];
        size_t &k = next[rv.peer];
        while (k < ss.size() && ss[k].peer != t->rank) ++k;
        if (k >= ss.size() || ss[k].bytes != rv.bytes) {  // what RCCL would answer with a hang
            t->err = "group_end: a receive has no matching send of the same size";
            rc = ADDER_E_BAD_PARAMS;
            break;
        }
        if (hipMemcpyAsync(rv.buf, ss[k].buf, rv.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
            rc = local_fail(t, "group_end (copy)");
            break;
        }
        ++k;
    }
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess && rc == ADDER_OK) rc = local_fail(t, "group_end (copy)");
    if (!local_barrier(g)) return local_fail(t, "group_end");
    return rc;
}
const char *local_error(void *self) { return static_cast<LocalTransport *>(self)->err.c_str(); }
}  // namespace

extern "C" int adder_gather_local_group_create(int world, AdderLocalGroup **out) {
    if (!out || world < 1) return ADDER_E_BAD_PARAMS;
    auto *g = new (std::nothrow) AdderLocalGroup();
    if (!g) return ADDER_E_HIP;
    g->world = world;
    g->ag_send.resize(world, nullptr);
    g->sends.resize(world);
    g->words.resize(world, 0);
    *out = g;
    return ADDER_OK;
}
extern "C" void adder_gather_local_group_destroy(AdderLocalGroup *g) { delete g; }

// ---------------------------------------------------------------------------------------------------------------------
// the gather object
// ---------------------------------------------------------------------------------------------------------------------
struct AdderGather {
    AdderHipCtx *ctx = nullptr;
    AdderTransport tr{};
    RcclTransport *rccl = nullptr;    // owned transport objects (null when the caller supplied the vtable)
    LocalTransport *local = nullptr;
    int rank = 0, world = 1, device = 0;
    // device scratch, grown on demand
    uint64_t *d_all_offs = nullptr;  // [world][T+1]
    size_t all_offs_cap = 0;         // bytes
    void *d_work = nullptr;          // merge layout
    size_t work_cap = 0;
    AdderEvent *d_stage = nullptr;   // root: the ranks' streams back to back
    size_t stage_cap = 0;            // bytes
    std::vector<uint64_t> h_offs;    // host copy of d_all_offs
    int32_t *d_flag = nullptr;       // the ranks' agreement that nothing failed before the payload exchange
    // records over the wire: this rank's image of a chunk, root's copies of the peers' images, the exchanged sizes
    static constexpr int kRecSlots = 3;  // images of three chunks in turn (arriving / being expanded / being written)
    uint8_t *d_rec_img[kRecSlots] = {nullptr, nullptr, nullptr};
    size_t rec_img_cap[kRecSlots] = {0, 0, 0};
    std::vector<uint8_t *> d_peer_img[kRecSlots];
    std::vector<size_t> peer_img_cap[kRecSlots];
    uint64_t *d_meta = nullptr;      // [kRecSlots][world + 1][8]: the gathered rows, then this rank's own
    uint64_t *h_meta = nullptr;      // pinned, same shape
    hipEvent_t meta_ev[kRecSlots] = {nullptr, nullptr, nullptr};  // a slot's gathered sizes have reached h_meta
    hipEvent_t copy_ev = nullptr;    // the image copy on the batch's stream -> the transport's stream
    hipEvent_t rec_ev[kRecSlots] = {nullptr, nullptr, nullptr};  // a slot's transfer / expansion is through (before it is rewritten)
    uint32_t rec_calls = 0;
    // streamed records gather (adder_gather_records_begin / _push / _end)
    struct RecordStream {
        bool open = false, agreed = false, overflow = false, wire = false, failed = false;
        int root = 0;
        int agreed_root = -1;        // the root the receive buffers were agreed for (another root: the agreement runs again)
        std::vector<size_t> worst;   // [world] every band's worst case of one chunk, as agreed (every rank holds the same)
        AdderEvent *d_merged = nullptr;
        size_t merged_cap = 0;
        uint64_t merged_base = 0, merged_pos = 0;
        uint64_t *d_merged_offsets = nullptr;
        uint32_t frame_pos = 0;
        hipStream_t s = nullptr;
        uint32_t pushes = 0;
        int pending = -1;            // slot of the chunk whose sizes are on their way (its payload has not been posted)
        uint64_t sent_bytes = 0;
        double host_us = 0.0;        // host time spent inside push() since begin (diagnostics)
    } rs;
    // sink per rank (adder_gather_host_sink_*)
    struct HostSink {
        bool open = false;
        uint8_t *out = nullptr;      // device-visible base of the image
        uint64_t out_cap = 0, header_bytes = 0;
        uint64_t *d_file_pos = nullptr;  // [0] events in the image so far; device
        uint64_t *d_dest = nullptr;      // [kMaxChunkFrames]
        uint64_t *d_chunk_offs = nullptr;  // this rank's offsets of the chunk, copied so that the caller may reuse its array
        uint32_t dest_cap = 0;
    } hs;
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
// a transport call: its own error text
#define GTR(g, expr)                                                                                       \
    do {                                                                                                   \
        int r_ = (expr);                                                                                   \
        if (r_ != ADDER_OK)                                                                                \
            return gfail(g, r_, "transport: %s (%s:%d)", (g)->tr.error ? (g)->tr.error((g)->tr.self) : "failed", __FILE__, __LINE__); \
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
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return gfail(nullptr, ADDER_E_HIP, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
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
static void use_rccl(AdderGather *g, ncclComm_t comm, bool owns) {
    g->rccl = new RcclTransport();
    g->rccl->comm = comm;
    g->rccl->owns = owns;
    g->tr = AdderTransport{g->rccl, rccl_all_gather, rccl_all_reduce_max, rccl_group_start, rccl_send, rccl_recv,
                           rccl_group_end, rccl_error};
}

extern "C" int adder_gather_create(AdderHipCtx *ctx, void *nccl_comm, int rank, int world, AdderGather **out) {
    if (!nccl_comm) return gfail(nullptr, ADDER_E_BAD_PARAMS, "nccl_comm is null");
    AdderGather *g = nullptr;
    int rc = create_common(ctx, rank, world, out, &g);
    if (rc != ADDER_OK) return rc;
    use_rccl(g, (ncclComm_t)nccl_comm, false);
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
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclCommInitRank(&comm, world, uid, rank);
    if (r != ncclSuccess) {
        g_err = std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r);
        delete g;
        return ADDER_E_HIP;
    }
    use_rccl(g, comm, true);
    *out = g;
    return ADDER_OK;
}

extern "C" int adder_gather_create_with_transport(AdderHipCtx *ctx, const AdderTransport *transport, int rank, int world,
                                                  AdderGather **out) {
    if (!transport || !transport->all_gather || !transport->all_reduce_max || !transport->group_start || !transport->send ||
        !transport->recv || !transport->group_end)
        return gfail(nullptr, ADDER_E_BAD_PARAMS, "transport: null table / entry");
    AdderGather *g = nullptr;
    int rc = create_common(ctx, rank, world, out, &g);
    if (rc != ADDER_OK) return rc;
    g->tr = *transport;
    *out = g;
    return ADDER_OK;
}

extern "C" int adder_gather_create_local(AdderHipCtx *ctx, AdderLocalGroup *group, int rank, AdderGather **out) {
    if (!group) return gfail(nullptr, ADDER_E_BAD_PARAMS, "group is null");
    AdderGather *g = nullptr;
    int rc = create_common(ctx, rank, group->world, out, &g);
    if (rc != ADDER_OK) return rc;
    g->local = new LocalTransport();
    g->local->grp = group;
    g->local->rank = rank;
    g->tr = AdderTransport{g->local, local_all_gather, local_all_reduce_max, local_group_start, local_send, local_recv,
                           local_group_end, local_error};
    *out = g;
    return ADDER_OK;
}

static void free_record_buffers(AdderGather *g) {
    for (int k = 0; k < AdderGather::kRecSlots; ++k) {
        if (g->d_rec_img[k]) (void)hipFree(g->d_rec_img[k]);
        for (uint8_t *p : g->d_peer_img[k])
            if (p) (void)hipFree(p);
        if (g->meta_ev[k]) (void)hipEventDestroy(g->meta_ev[k]);
        if (g->rec_ev[k]) (void)hipEventDestroy(g->rec_ev[k]);
    }
    if (g->d_meta) (void)hipFree(g->d_meta);
    if (g->h_meta) (void)hipHostFree(g->h_meta);
    if (g->copy_ev) (void)hipEventDestroy(g->copy_ev);
}
extern "C" void adder_gather_destroy(AdderGather *g) {
    if (!g) return;
    free_record_buffers(g);
    if (g->d_all_offs) (void)hipFree(g->d_all_offs);
    if (g->d_work) (void)hipFree(g->d_work);
    if (g->d_stage) (void)hipFree(g->d_stage);
    if (g->d_flag) (void)hipFree(g->d_flag);
    for (void *p : {(void *)g->hs.d_file_pos, (void *)g->hs.d_dest, (void *)g->hs.d_chunk_offs})
        if (p) (void)hipFree(p);
    if (g->rccl) {
        if (g->rccl->owns && g->rccl->comm) (void)ncclCommDestroy(g->rccl->comm);
        delete g->rccl;
    }
    delete g->local;
    delete g;
}

extern "C" const char *adder_gather_last_error(const AdderGather *g) { return g ? g->err.c_str() : g_err.c_str(); }
extern "C" int adder_gather_world(const AdderGather *g) { return g ? g->world : 0; }

// all-gather of the ranks' frame offsets -> g->d_all_offs [world][T+1]; with `to_host` also its host copy (waits)
static int gather_offsets(AdderGather *g, const uint64_t *d_frame_offsets, uint32_t T, hipStream_t s, bool to_host = true) {
    const size_t per = (size_t)T + 1;
    void *p = g->d_all_offs;
    int rc = grow(g, &p, &g->all_offs_cap, per * g->world * sizeof(uint64_t));
    g->d_all_offs = (uint64_t *)p;
    if (rc != ADDER_OK) return rc;
    GTR(g, g->tr.all_gather(g->tr.self, d_frame_offsets, g->d_all_offs, per * sizeof(uint64_t), s));
    if (!to_host) return ADDER_OK;
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

// the ranks agree (one word, max) on whether any of them failed locally before a payload exchange: every rank then
// returns an error, none is left waiting in a send
static int agree(AdderGather *g, int local_rc, hipStream_t s) {
    if (g->world > 1) {
        if (!g->d_flag) GHIP(g, hipMalloc(reinterpret_cast<void **>(&g->d_flag), sizeof(int32_t)));
        const int32_t bad = local_rc != ADDER_OK ? 1 : 0;
        int32_t any = 0;
        GHIP(g, hipMemcpyAsync(g->d_flag, &bad, sizeof bad, hipMemcpyHostToDevice, s));
        GTR(g, g->tr.all_reduce_max(g->tr.self, g->d_flag, s));
        GHIP(g, hipMemcpyAsync(&any, g->d_flag, sizeof any, hipMemcpyDeviceToHost, s));
        GHIP(g, hipStreamSynchronize(s));
        if (any) {
            if (local_rc != ADDER_OK) return local_rc;
            return gfail(g, ADDER_E_HIP, "another rank failed before the exchange");
        }
        return ADDER_OK;
    }
    return local_rc;
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
    rc = agree(g, local_rc, s);
    if (rc != ADDER_OK) return rc;
    // payload: every rank -> root, back to back in rank order (a group of point-to-point transfers
    // over xGMI; root's own stream is a device copy)
    GTR(g, g->tr.group_start(g->tr.self));
    if (g->rank == root) {
        for (int r = 0; r < g->world; ++r) {
            if (r == root || tot[r] == 0) continue;
            GTR(g, g->tr.recv(g->tr.self, g->d_stage + base[r], (size_t)tot[r] * sizeof(AdderEvent), r, s));
        }
    } else if (tot[g->rank] != 0) {
        GTR(g, g->tr.send(g->tr.self, d_events + first, (size_t)tot[g->rank] * sizeof(AdderEvent), root, s));
    }
    GTR(g, g->tr.group_end(g->tr.self, s));
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

// ---------------------------------------------------------------------------------------------------------------------
// records over the wire
// ---------------------------------------------------------------------------------------------------------------------
constexpr size_t kMetaWords = 8;  // nf, segments, record bytes, records, row_begin, rows, events, flags (1 = local failure)

static uint64_t *meta_rows(AdderGather *g, uint64_t *base, int slot) { return base + (size_t)slot * (g->world + 1) * kMetaWords; }
static int ensure_meta(AdderGather *g) {
    const size_t words = (size_t)AdderGather::kRecSlots * (g->world + 1) * kMetaWords;
    if (!g->d_meta) GHIP(g, hipMalloc(reinterpret_cast<void **>(&g->d_meta), words * sizeof(uint64_t)));
    if (!g->h_meta) GHIP(g, hipHostMalloc(reinterpret_cast<void **>(&g->h_meta), words * sizeof(uint64_t), hipHostMallocDefault));
    for (hipEvent_t &e : g->meta_ev)
        if (!e) GHIP(g, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t &e : g->rec_ev)
        if (!e) GHIP(g, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!g->copy_ev) GHIP(g, hipEventCreateWithFlags(&g->copy_ev, hipEventDisableTiming));
    return ADDER_OK;
}
static size_t row_wire_bytes(const uint64_t *row) {
    return adder_hip_records_wire_bytes((uint32_t)row[0], (uint32_t)row[1], (uint32_t)row[2], row[3]);
}
// this rank's image of the chunk into slot's buffer, copied on the BATCH's stream (before the context's next batch can
// touch its scratch); `s` waits for the copy.  The slot's previous transfer must be through first (device-side wait).
static int image_to_slot(AdderGather *g, const AdderBandRecords *rec, uint64_t n_records, int slot, size_t cap_wanted,
                         hipStream_t s) {
    void *p = g->d_rec_img[slot];
    int rc = grow(g, &p, &g->rec_img_cap[slot], cap_wanted);
    g->d_rec_img[slot] = (uint8_t *)p;
    if (rc != ADDER_OK) return rc;
    hipStream_t bs = (hipStream_t)adder_hip_last_batch_stream(g->ctx);
    if (bs != s) GHIP(g, hipStreamWaitEvent(bs, g->rec_ev[slot], 0));
    rc = adder_hip_records_to_wire(g->ctx, rec, n_records, g->d_rec_img[slot], g->rec_img_cap[slot], bs);
    if (rc != ADDER_OK) return gfail(g, rc, "image: %s", adder_hip_last_error(g->ctx));
    if (bs != s) {
        GHIP(g, hipEventRecord(g->copy_ev, bs));
        GHIP(g, hipStreamWaitEvent(s, g->copy_ev, 0));
    }
    return ADDER_OK;
}
// the payload of the chunk in `slot` (sizes in `all`, [world][kMetaWords]): every peer -> root, one point-to-point transfer
// each (its own xGMI link); on root the bands' descriptions over the images and the expansion behind merged_base
static int transfer_and_expand(AdderGather *g, int slot, const uint64_t *all, int root, AdderEvent *d_merged, size_t merged_cap,
                               uint64_t merged_base, uint64_t *d_merged_offsets, hipStream_t s, bool wire = false) {
    GTR(g, g->tr.group_start(g->tr.self));
    if (g->rank == root) {
        for (int r = 0; r < g->world; ++r)
            if (r != root) GTR(g, g->tr.recv(g->tr.self, g->d_peer_img[slot][r], row_wire_bytes(all + (size_t)r * kMetaWords), r, s));
    } else {
        GTR(g, g->tr.send(g->tr.self, g->d_rec_img[slot], row_wire_bytes(all + (size_t)g->rank * kMetaWords), root, s));
    }
    GTR(g, g->tr.group_end(g->tr.self, s));
    if (g->rank != root) {
        GHIP(g, hipEventRecord(g->rec_ev[slot], s));
        return ADDER_OK;  // (queued: the image is this object's, the caller's context is free)
    }
    std::vector<AdderBandRecords> bands(g->world);
    size_t sec[6];
    const uint64_t *own = all + (size_t)root * kMetaWords;
    adder_hip_records_wire_sections((uint32_t)own[0], (uint32_t)own[1], (uint32_t)own[2], sec);
    const void *ftab = g->d_rec_img[slot] + sec[1];  // root's own frame table of these frames
    for (int r = 0; r < g->world; ++r) {
        const uint64_t *row = all + (size_t)r * kMetaWords;
        const uint8_t *img = r == root ? g->d_rec_img[slot] : g->d_peer_img[slot][r];
        AdderBandRecords &b = bands[r];
        b.num_frames = (uint32_t)row[0];
        b.num_segments = (uint32_t)row[1];
        b.record_bytes = (uint32_t)row[2];
        b.row_begin = (uint32_t)row[4];
        b.rows = (uint32_t)row[5];
        adder_hip_records_wire_sections(b.num_frames, b.num_segments, b.record_bytes, sec);
        b.d_frame_offsets = reinterpret_cast<const uint64_t *>(img + sec[0]);
        b.d_frame_table = ftab;
        b.d_counts = reinterpret_cast<const uint32_t *>(img + sec[2]);
        b.d_prefix = reinterpret_cast<const uint32_t *>(img + sec[3]);
        b.d_runs = reinterpret_cast<const uint32_t *>(img + sec[4]);
        b.d_records = img + sec[5];
    }
    // (wire: merged_cap counts events of the byte buffer behind d_merged)
    const size_t wrec = wire ? adder_hip_wire_record_bytes(g->ctx) : 0;
    int rc = wire ? adder_hip_expand_records_wire_device(g->ctx, bands.data(), (uint32_t)g->world, reinterpret_cast<uint8_t *>(d_merged),
                                                         merged_cap * wrec, merged_base, d_merged_offsets, s)
                  : adder_hip_expand_records_device(g->ctx, bands.data(), (uint32_t)g->world, d_merged, merged_cap, merged_base,
                                                    d_merged_offsets, s);
    if (rc != ADDER_OK) return gfail(g, rc, "expansion: %s", adder_hip_last_error(g->ctx));
    GHIP(g, hipEventRecord(g->rec_ev[slot], s));
    return ADDER_OK;
}
// root's receive buffers of `slot`, at least `need[r]` bytes each
static int ensure_peer_images(AdderGather *g, int slot, const std::vector<size_t> &need) {
    g->d_peer_img[slot].resize(g->world, nullptr);
    g->peer_img_cap[slot].resize(g->world, 0);
    for (int r = 0; r < g->world; ++r) {
        if (r == g->rank) continue;
        void *p = g->d_peer_img[slot][r];
        int rc = grow(g, &p, &g->peer_img_cap[slot][r], need[r]);
        g->d_peer_img[slot][r] = (uint8_t *)p;
        if (rc != ADDER_OK) return rc;
    }
    return ADDER_OK;
}

// One chunk, complete before the call returns its count (two host waits: the sizes, the agreement).  The streamed form
// below takes the host out of the loop.
extern "C" int adder_gather_records_at(AdderGather *g, const AdderBandRecords *rec, uint64_t n_records, uint64_t n_events,
                                       int root, AdderEvent *d_merged, size_t merged_cap, uint64_t merged_base,
                                       uint64_t *d_merged_offsets, size_t *n_merged, void *stream) {
    if (n_merged) *n_merged = 0;
    if (!g || !rec) return gfail(g, ADDER_E_BAD_PARAMS, "null argument");
    if (root < 0 || root >= g->world) return gfail(g, ADDER_E_BAD_PARAMS, "bad root %d", root);
    if (g->rs.open) return gfail(g, ADDER_E_BAD_PARAMS, "a streamed records gather is open (adder_gather_records_end)");
    hipStream_t s = (hipStream_t)stream;
    { int rc_ = ensure_meta(g); if (rc_ != ADDER_OK) return rc_; }
    const int slot = (int)(g->rec_calls++ % AdderGather::kRecSlots);
    const uint32_t nf = rec->num_frames;
    // ---- the sizes ----
    uint64_t *d_rows = meta_rows(g, g->d_meta, slot), *h_rows = meta_rows(g, g->h_meta, slot);
    uint64_t *mine = h_rows + (size_t)g->world * kMetaWords;
    const uint64_t vals[kMetaWords] = {nf, rec->num_segments, rec->record_bytes, n_records, rec->row_begin, rec->rows, n_events, 0};
    memcpy(mine, vals, sizeof vals);
    GHIP(g, hipMemcpyAsync(d_rows + (size_t)g->world * kMetaWords, mine, sizeof vals, hipMemcpyHostToDevice, s));
    GTR(g, g->tr.all_gather(g->tr.self, d_rows + (size_t)g->world * kMetaWords, d_rows, sizeof vals, s));
    GHIP(g, hipMemcpyAsync(h_rows, d_rows, (size_t)g->world * sizeof vals, hipMemcpyDeviceToHost, s));
    GHIP(g, hipStreamSynchronize(s));
    const uint64_t *all = h_rows;
    // ---- everything that can fail locally, then one agreement ----
    int local_rc = ADDER_OK;
    uint64_t total = 0;
    for (int r = 0; r < g->world; ++r) {
        if (all[(size_t)r * kMetaWords] != nf || all[(size_t)r * kMetaWords + 2] != rec->record_bytes)
            local_rc = gfail(g, ADDER_E_BAD_PARAMS, "rank %d passed a chunk of another length / record size", r);
        total += all[(size_t)r * kMetaWords + 6];
    }
    if (g->rank == root && local_rc == ADDER_OK) {
        std::vector<size_t> need(g->world, 0);
        for (int r = 0; r < g->world; ++r) need[r] = row_wire_bytes(all + (size_t)r * kMetaWords);
        local_rc = ensure_peer_images(g, slot, need);
        if (local_rc == ADDER_OK && (merged_base > merged_cap || total > merged_cap - merged_base))
            local_rc = gfail(g, ADDER_E_OUT_CAPACITY, "merged buffer too small: need %llu events",
                             (unsigned long long)(merged_base + total));
    }
    if (local_rc == ADDER_OK) local_rc = image_to_slot(g, rec, n_records, slot, row_wire_bytes(vals), s);
    int rc = agree(g, local_rc, s);
    if (rc != ADDER_OK) return rc;
    rc = transfer_and_expand(g, slot, all, root, d_merged, merged_cap, merged_base, d_merged_offsets, s);
    if (rc != ADDER_OK) return rc;
    if (g->rank == root && n_merged) *n_merged = (size_t)total;
    return ADDER_OK;
}

// ---- streamed form: begin / push per chunk / end ----
// push(k) copies the chunk's image, posts the all-gather of ITS sizes (device -> pinned host, an event behind it) and
// then posts the payload of chunk k-1, whose sizes were gathered a whole chunk ago: the one host wait of a push is for an
// event that has long fired.  Nothing a rank can fail at is left between "sizes known" and "payload posted": the image
// buffers hold the worst case of a chunk from the first push on (that push, and only that one, agrees on the
// allocations with the blocking all-reduce), a local failure travels in the rank's own row of the sizes, and rows that
// do not fit together are seen by every rank alike.  A merged buffer too small is root's own affair: the expansion drops
// what does not fit and end() reports it.
extern "C" int adder_gather_records_begin(AdderGather *g, int root, AdderEvent *d_merged, size_t merged_cap, uint64_t merged_base,
                                          uint64_t *d_merged_offsets, void *stream) {
    if (!g) return ADDER_E_BAD_PARAMS;
    if (root < 0 || root >= g->world) return gfail(g, ADDER_E_BAD_PARAMS, "bad root %d", root);
    if (g->rs.open) return gfail(g, ADDER_E_BAD_PARAMS, "a streamed records gather is already open");
    if (g->rank == root && (!d_merged_offsets || (!d_merged && merged_cap)))
        return gfail(g, ADDER_E_BAD_PARAMS, "root needs the merged buffers");
    { int rc_ = ensure_meta(g); if (rc_ != ADDER_OK) return rc_; }
    AdderGather::RecordStream &rs = g->rs;
    // the receive buffers were sized and agreed on for ONE root: a clip with another root agrees again (every rank sees
    // the same root argument, so every rank re-runs the blocking round in its first push)
    const bool agreed = rs.agreed && rs.agreed_root == root;
    std::vector<size_t> worst = std::move(rs.worst);
    rs = AdderGather::RecordStream{};
    rs.agreed = agreed;
    rs.agreed_root = agreed ? root : -1;
    if (agreed) rs.worst = std::move(worst);
    rs.open = true;
    rs.root = root;
    rs.d_merged = d_merged;
    rs.merged_cap = merged_cap;
    rs.merged_base = rs.merged_pos = merged_base;
    rs.d_merged_offsets = d_merged_offsets;
    rs.s = (hipStream_t)stream;
    return ADDER_OK;
}
extern "C" int adder_gather_records_begin_wire(AdderGather *g, int root, uint8_t *d_wire, size_t wire_cap_bytes, uint64_t merged_base,
                                               uint64_t *d_merged_offsets, void *stream) {
    if (!g) return ADDER_E_BAD_PARAMS;
    const size_t wrec = adder_hip_wire_record_bytes(g->ctx);
    int rc = adder_gather_records_begin(g, root, reinterpret_cast<AdderEvent *>(d_wire), wire_cap_bytes / wrec, merged_base, d_merged_offsets, stream);
    if (rc == ADDER_OK) g->rs.wire = true;
    return rc;
}

// the chunk in `slot`: its sizes have been gathered -- check them, post its payload, expand (root)
static int records_complete(AdderGather *g, int slot) {
    AdderGather::RecordStream &rs = g->rs;
    GHIP(g, hipEventSynchronize(g->meta_ev[slot]));  // (queued a chunk ago: normally no wait)
    const uint64_t *all = meta_rows(g, g->h_meta, slot);
    uint64_t total = 0;
    for (int r = 0; r < g->world; ++r) {
        const uint64_t *row = all + (size_t)r * kMetaWords;
        // every rank reads the same rows, so every rank takes the same way out: nobody is left in a send
        if (row[0] != all[0] || row[2] != all[2]) return gfail(g, ADDER_E_BAD_PARAMS, "rank %d pushed a chunk of another length / record size", r);
        if (row[7] != 0) return gfail(g, ADDER_E_HIP, "rank %d failed before the exchange", r);
        total += row[6];
    }
    const uint32_t nf = (uint32_t)all[0];
    // (every rank holds the agreed worst cases and reads the same rows: a chunk that outgrows them is refused by ALL of
    // them before anything is posted -- a check on root alone would leave the peers in their sends)
    for (int r = 0; r < g->world; ++r)
        if (r != rs.root && (size_t)r < rs.worst.size() && row_wire_bytes(all + (size_t)r * kMetaWords) > rs.worst[r])
            return gfail(g, ADDER_E_BAD_PARAMS, "rank %d's chunk exceeds the worst case its first chunk announced", r);
    if (g->rank == rs.root) {
        if (rs.merged_pos > rs.merged_cap || total > rs.merged_cap - rs.merged_pos) rs.overflow = true;  // (the kernels clamp)
    } else {
        rs.sent_bytes += row_wire_bytes(all + (size_t)g->rank * kMetaWords);
    }
    int rc = transfer_and_expand(g, slot, all, rs.root, rs.d_merged, rs.merged_cap, rs.merged_pos,
                                 rs.d_merged_offsets ? rs.d_merged_offsets + rs.frame_pos : nullptr, rs.s, rs.wire);
    if (rc != ADDER_OK) return rc;
    rs.merged_pos += total;
    rs.frame_pos += nf;
    return ADDER_OK;
}

extern "C" int adder_gather_records_push(AdderGather *g, const AdderBandRecords *rec, uint64_t n_records, uint64_t n_events) {
    if (!g || !rec) return gfail(g, ADDER_E_BAD_PARAMS, "null argument");
    AdderGather::RecordStream &rs = g->rs;
    if (!rs.open) return gfail(g, ADDER_E_BAD_PARAMS, "no streamed records gather is open (adder_gather_records_begin)");
    // a clip that failed posts nothing more: every rank saw the same rows, so every rank stops at the same chunk
    if (rs.failed) return gfail(g, ADDER_E_BAD_PARAMS, "the streamed gather failed at an earlier chunk (adder_gather_records_end, then begin again)");
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = rs.s;
    const int slot = (int)(g->rec_calls++ % AdderGather::kRecSlots);
    const uint32_t nf = rec->num_frames, nseg = rec->num_segments, rb = rec->record_bytes;
    // the worst case of one chunk of this band: a record per unit and frame (adder_hip_chunk_frames() frames)
    const uint32_t nf_max = std::max<uint32_t>(adder_hip_chunk_frames(g->ctx), nf);
    const uint64_t seg_units = adder_hip_segment_units();
    const size_t worst = adder_hip_records_wire_bytes(nf_max, nseg, rb, (uint64_t)nseg * seg_units * nf_max);
    // (the image first: the caller's context is free for its next batch whatever happens below)
    int local_rc = image_to_slot(g, rec, n_records, slot, worst, s);
    // ---- the payload of the chunk before this one, BEFORE anything new is queued: its rows are checked by every rank
    // alike, and a rank that meets a failed row (or a peer's failure flag) returns without having posted a collective the
    // others would wait for -- the failing rank itself returned from the push that queued its last all-gather ----
    if (rs.pending >= 0) {
        int rc = records_complete(g, rs.pending);
        rs.pending = -1;
        if (rc != ADDER_OK) {
            rs.failed = true;
            return rc;
        }
    }
    // ---- this chunk's sizes: gathered, copied to pinned memory, an event behind them; nobody waits here ----
    uint64_t *d_rows = meta_rows(g, g->d_meta, slot), *h_rows = meta_rows(g, g->h_meta, slot);
    uint64_t *mine = h_rows + (size_t)g->world * kMetaWords;
    const uint64_t vals[kMetaWords] = {nf, nseg, rb, n_records, rec->row_begin, rec->rows, n_events, local_rc != ADDER_OK ? 1ull : 0ull};
    memcpy(mine, vals, sizeof vals);
    GHIP(g, hipMemcpyAsync(d_rows + (size_t)g->world * kMetaWords, mine, sizeof vals, hipMemcpyHostToDevice, s));
    GTR(g, g->tr.all_gather(g->tr.self, d_rows + (size_t)g->world * kMetaWords, d_rows, sizeof vals, s));
    GHIP(g, hipMemcpyAsync(h_rows, d_rows, (size_t)g->world * sizeof vals, hipMemcpyDeviceToHost, s));
    GHIP(g, hipEventRecord(g->meta_ev[slot], s));
    if (!rs.agreed) {
        // the first chunk this object sees (for this root): every rank works the bands' worst cases out of the same rows,
        // root sizes its receive buffers (all slots) for them, and the ranks agree that this worked -- the only blocking
        // agreement of the object's life
        GHIP(g, hipEventSynchronize(g->meta_ev[slot]));
        rs.worst.assign(g->world, 0);
        for (int r = 0; r < g->world; ++r) {
            const uint64_t *row = h_rows + (size_t)r * kMetaWords;
            const uint32_t fm = std::max<uint32_t>(nf_max, (uint32_t)row[0]);
            rs.worst[r] = adder_hip_records_wire_bytes(fm, (uint32_t)row[1], (uint32_t)row[2], row[1] * seg_units * fm);
        }
        int rc0 = ADDER_OK;
        if (g->rank == rs.root)
            for (int k = 0; k < AdderGather::kRecSlots && rc0 == ADDER_OK; ++k) rc0 = ensure_peer_images(g, k, rs.worst);
        int rc = agree(g, rc0, s);
        if (rc != ADDER_OK) {
            rs.failed = true;
            return rc;
        }
        rs.agreed = true;
        rs.agreed_root = rs.root;
    }
    rs.pending = slot;
    rs.pushes += 1;
    rs.host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (local_rc != ADDER_OK) {  // (the others learn of it from this rank's row, at their next push or end)
        rs.failed = true;
        return local_rc;
    }
    return ADDER_OK;
}

extern "C" int adder_gather_records_end(AdderGather *g, size_t *n_merged, uint64_t *bytes_sent) {
    if (n_merged) *n_merged = 0;
    if (bytes_sent) *bytes_sent = 0;
    if (!g) return ADDER_E_BAD_PARAMS;
    AdderGather::RecordStream &rs = g->rs;
    if (!rs.open) return gfail(g, ADDER_E_BAD_PARAMS, "no streamed records gather is open");
    rs.open = false;
    int rc = ADDER_OK;
    if (rs.pending >= 0 && !rs.failed) {
        rc = records_complete(g, rs.pending);
    } else if (rs.failed) {
        rc = gfail(g, ADDER_E_HIP, "the streamed gather failed at an earlier chunk");
    }
    rs.pending = -1;
    GHIP(g, hipStreamSynchronize(rs.s));
    if (rc != ADDER_OK) return rc;
    if (bytes_sent) *bytes_sent = rs.sent_bytes;
    if (g->rank == rs.root) {
        if (n_merged) *n_merged = (size_t)(rs.merged_pos - rs.merged_base);
        rc = adder_hip_expand_status(g->ctx, rs.s);
        if (rc != ADDER_OK) return gfail(g, rc, "expansion: %s", adder_hip_last_error(g->ctx));
        if (rs.overflow)
            return gfail(g, ADDER_E_OUT_CAPACITY, "merged buffer too small: need %llu events", (unsigned long long)rs.merged_pos);
    }
    return ADDER_OK;
}

extern "C" double adder_gather_records_host_us(const AdderGather *g) { return g ? g->rs.host_us : 0.0; }

// ---------------------------------------------------------------------------------------------------------------------
// sink per rank
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int adder_gather_host_sink_open(AdderGather *g, void *image, uint64_t image_bytes, uint64_t header_bytes,
                                           void *stream) {
    if (!g || !image) return gfail(g, ADDER_E_BAD_PARAMS, "null argument");
    if (header_bytes > image_bytes) return gfail(g, ADDER_E_BAD_PARAMS, "the header does not fit the image");
    AdderGather::HostSink &hs = g->hs;
    if (!hs.d_file_pos) GHIP(g, hipMalloc(reinterpret_cast<void **>(&hs.d_file_pos), 2 * sizeof(uint64_t)));
    GHIP(g, hipMemsetAsync(hs.d_file_pos, 0, 2 * sizeof(uint64_t), (hipStream_t)stream));
    hs.out = static_cast<uint8_t *>(image);
    hs.out_cap = image_bytes;
    hs.header_bytes = header_bytes;
    hs.open = true;
    return ADDER_OK;
}

// One chunk: all-gather of the chunk's frame offsets, the layout on the device, the scatter.  Queued on `stream`; the
// host waits for nothing (the file position lives on the device until close).
extern "C" int adder_gather_host_sink_chunk(AdderGather *g, const AdderEvent *d_events, const uint64_t *d_frame_offsets,
                                            uint32_t num_frames, void *stream) {
    if (!g || !d_frame_offsets || num_frames == 0) return gfail(g, ADDER_E_BAD_PARAMS, "null argument / no frames");
    AdderGather::HostSink &hs = g->hs;
    if (!hs.open) return gfail(g, ADDER_E_BAD_PARAMS, "the sink is not open (adder_gather_host_sink_open)");
    hipStream_t s = (hipStream_t)stream;
    if (hs.dest_cap < num_frames + 1) {
        GHIP(g, hipStreamSynchronize(s));
        if (hs.d_dest) GHIP(g, hipFree(hs.d_dest));
        if (hs.d_chunk_offs) GHIP(g, hipFree(hs.d_chunk_offs));
        hs.d_dest = hs.d_chunk_offs = nullptr;
        hs.dest_cap = 0;
        const uint32_t cap = std::max<uint32_t>(num_frames + 1, 65u);
        GHIP(g, hipMalloc(reinterpret_cast<void **>(&hs.d_dest), cap * sizeof(uint64_t)));
        GHIP(g, hipMalloc(reinterpret_cast<void **>(&hs.d_chunk_offs), cap * sizeof(uint64_t)));
        hs.dest_cap = cap;
    }
    // (the rank's offsets are copied first: the scatter reads them after the caller's next batch may have rewritten the array)
    GHIP(g, hipMemcpyAsync(hs.d_chunk_offs, d_frame_offsets, ((size_t)num_frames + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    int rc = gather_offsets(g, hs.d_chunk_offs, num_frames, s, false);
    if (rc != ADDER_OK) return rc;
    rc = adder_hip_sink_layout_device(g->ctx, g->d_all_offs, (uint32_t)g->world, (uint32_t)g->rank, num_frames, hs.d_file_pos,
                                      hs.d_dest, nullptr, s);
    if (rc != ADDER_OK) return gfail(g, rc, "sink layout: %s", adder_hip_last_error(g->ctx));
    rc = adder_hip_wire_scatter_device(g->ctx, d_events, hs.d_chunk_offs, num_frames, hs.d_dest, hs.out, hs.out_cap,
                                       hs.header_bytes, s);
    if (rc != ADDER_OK) return gfail(g, rc, "wire scatter: %s", adder_hip_last_error(g->ctx));
    return ADDER_OK;
}

extern "C" int adder_gather_host_sink_close(AdderGather *g, uint64_t *total_events, void *stream) {
    if (total_events) *total_events = 0;
    if (!g) return ADDER_E_BAD_PARAMS;
    AdderGather::HostSink &hs = g->hs;
    if (!hs.open) return gfail(g, ADDER_E_BAD_PARAMS, "the sink is not open");
    hs.open = false;
    hipStream_t s = (hipStream_t)stream;
    uint64_t tot = 0;
    GHIP(g, hipMemcpyAsync(&tot, hs.d_file_pos, sizeof tot, hipMemcpyDeviceToHost, s));
    GHIP(g, hipStreamSynchronize(s));
    if (total_events) *total_events = tot;
    int rc = adder_hip_expand_status(g->ctx, s);  // (the scatter's status word: bytes past the image, c = None)
    if (rc != ADDER_OK) return gfail(g, rc, "sink: %s", adder_hip_last_error(g->ctx));
    return ADDER_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// the image the ranks' sinks store into: a POSIX shared-memory file mapped into the process and into the device
// ---------------------------------------------------------------------------------------------------------------------
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct AdderHostImage {
    std::string name;
    int fd = -1;
    void *host = nullptr, *dev = nullptr;
    uint64_t bytes = 0;
    bool creator = false, registered = false;
};

extern "C" int adder_host_image_open(const char *name, uint64_t bytes, int create, AdderHostImage **out) {
    if (!out) return gfail(nullptr, ADDER_E_BAD_PARAMS, "out is null");
    *out = nullptr;
    if (!name || name[0] != '/' || bytes == 0) return gfail(nullptr, ADDER_E_BAD_PARAMS, "image: a name starting with '/' and a size");
    auto *im = new (std::nothrow) AdderHostImage();
    if (!im) return gfail(nullptr, ADDER_E_HIP, "out of host memory");
    im->name = name;
    im->bytes = bytes;
    im->creator = create != 0;
    auto bail = [&](const char *what) {
        const int e = errno;
        if (im->host) (void)munmap(im->host, im->bytes);
        if (im->fd >= 0) (void)close(im->fd);
        if (im->creator) (void)shm_unlink(name);
        delete im;
        return gfail(nullptr, ADDER_E_HIP, "image %s: %s failed: %s", name, what, strerror(e));
    };
    im->fd = shm_open(name, create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (im->fd < 0) return bail("shm_open");
    if (create && ftruncate(im->fd, (off_t)bytes) != 0) return bail("ftruncate");
    im->host = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, im->fd, 0);
    if (im->host == MAP_FAILED) {
        im->host = nullptr;
        return bail("mmap");
    }
    hipError_t e = hipHostRegister(im->host, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e == hipSuccess) {
        im->registered = true;
        e = hipHostGetDevicePointer(&im->dev, im->host, 0);
    }
    if (e != hipSuccess) {
        if (im->registered) (void)hipHostUnregister(im->host);
        (void)munmap(im->host, bytes);
        (void)close(im->fd);
        if (im->creator) (void)shm_unlink(name);
        const int rc = gfail(nullptr, ADDER_E_HIP, "image %s: hipHostRegister failed: %s", name, hipGetErrorString(e));
        delete im;
        return rc;
    }
    *out = im;
    return ADDER_OK;
}
extern "C" void *adder_host_image_host_ptr(const AdderHostImage *im) { return im ? im->host : nullptr; }
extern "C" void *adder_host_image_device_ptr(const AdderHostImage *im) { return im ? im->dev : nullptr; }
extern "C" int adder_host_image_close(AdderHostImage *im, int64_t final_bytes, int unlink_file) {
    if (!im) return ADDER_E_BAD_PARAMS;
    int rc = ADDER_OK;
    if (im->registered && hipHostUnregister(im->host) != hipSuccess) rc = ADDER_E_HIP;
    if (im->host) (void)munmap(im->host, im->bytes);
    if (final_bytes >= 0 && im->fd >= 0 && ftruncate(im->fd, (off_t)final_bytes) != 0) rc = ADDER_E_HIP;
    if (im->fd >= 0) (void)close(im->fd);
    if (unlink_file) (void)shm_unlink(im->name.c_str());
    delete im;
    return rc;
}
