// adder_compressed.cpp -- the CPU compressed ADDER sink / source behind include/adder_compressed.h
// (SURVEY 8(f)2).  Plain C++: this stage stays on the host (it is the "unchanged CPU arithmetic-coding
// stage" the GPU path feeds), so nothing here touches HIP.
//
// Same stream as the reference's Encoder::new_compressed + CompressedOutput (adder-codec-core/src/codec/
// compressed/stream.rs:126-329) produces, built differently:
//   * an ADU keeps ONE vector of (pixel, d, t) per 16x16 EventCube in arrival order instead of a Vec per
//     pixel (3 x 16 x 16 Vecs per cube in the reference, event_cube.rs:17-19,53-63 -- 25 million vectors for
//     a 4K RGB plane); when the ADU is compressed a stable counting sort per cube recovers every pixel's list,
//     and the ingest-time rule "a pixel with two or more events drops an event that is not later than its
//     last one" (event_cube.rs:135-151) is applied to the list then (it only looks at the pixel's own list);
//   * the adaptive model's contexts are flat Fenwick arrays of uint32 (totals stay below 2^30,
//     event_adu.rs:95); the unused 65 536-symbol default context (context_switching.rs:20-33) is not built;
//   * finished ADUs are compressed on a small pool of worker threads and appended to the stream in ADU order
//     (the reference spawns one thread per ADU and reorders through a priority queue, stream.rs:78-104).
// Checked byte for byte against the test suite's literal restatement of the reference (tests/test_compressed_product.py).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/adder_compressed.h"

namespace {

constexpr uint32_t kBlock = 16;             // event_structure/mod.rs:8
constexpr int kNoEvent = 256;               // DRESIDUAL_NO_EVENT (compressed/mod.rs:11)
constexpr int kSkipCube = 257;              // DRESIDUAL_SKIP_CUBE
constexpr int kDResidualOffset = 255;       // cabac_contexts.rs:21
constexpr uint32_t kEncodeFull = 15;        // BITSHIFT_ENCODE_FULL
constexpr uint32_t kDEmpty = 255;
constexpr uint64_t kMaxDenominator = 1ull << 30;
constexpr uint32_t kPrecision = 64 - (30 + 1);  // encoder.rs:46-49
constexpr uint64_t kHalf = 1ull << (kPrecision - 1), kQuarter = 1ull << (kPrecision - 2);
constexpr uint64_t kThreeQuarter = kHalf + kQuarter;
constexpr int64_t kTResidualMax = (256 - 2) / 2;  // cabac_contexts.rs:33

// ---------------------------------------------------------------- Weights (fenwick/mod.rs:11-110)
struct Weights {
    std::vector<uint32_t> fen;  // [0] = EOF, [i + 1] = symbol i
    uint64_t total = 0;
    void init(const uint32_t *counts, size_t n) {
        fen.assign(n + 1, 0);
        total = 0;
        for (size_t i = 0; i < n; ++i) add(i + 1, counts[i]);
        add(0, 1);
    }
    void add(size_t index, uint32_t delta) {
        for (size_t i = index; i < fen.size(); i |= i + 1) fen[i] += delta;
        total += delta;
    }
    uint64_t prefix(size_t index) const {  // inclusive
        uint64_t s = 0;
        for (size_t i = index + 1; i > 0; i &= i - 1) s += fen[i - 1];
        return s;
    }
    // symbol index -> fenwick index: Some(i) -> i + 1, None -> 0
    void range(size_t index, uint64_t &lo, uint64_t &hi) const {
        hi = prefix(index);
        lo = index == 0 ? 0 : prefix(index - 1);
    }
    // decoding: the fenwick index whose range holds `value` (0 = EOF)
    size_t find(uint64_t value) const {
        if (value < prefix(0)) return 0;
        size_t low = 0, high = fen.size() - 1;  // symbols
        while (low + 1 < high) {
            const size_t i = (low + high - 1) / 2;
            if (prefix(i + 1) > value)
                high = i + 1;
            else
                low = i + 1;
        }
        return low + 1;
    }
};

struct Model {  // FenwickModel + Contexts (context_switching.rs, cabac_contexts.rs:26-47)
    Weights d, t, eof, bitshift;
    Model() {
        uint32_t dc[513];
        for (int i = 0; i < 513; ++i) {  // d_residual_default_weights (:191-234)
            dc[i] = 1;
            if (i >= 245 && i <= 265) dc[i] = 20;
            else if ((i >= 235 && i <= 275) || (i >= 490 && i <= 510) || i <= 20) dc[i] = 10;
            else if (i == 511) dc[i] = 20;
            else if (i == 512) dc[i] = 10;
        }
        d.init(dc, 513);
        uint32_t tc[256];  // t_residual_default_weights (:153-174)
        for (int i = 0; i < 256; ++i) tc[i] = i < 10 ? 10 : 1;
        t.init(tc, 256);
        const uint32_t one = 1;
        eof.init(&one, 1);
        uint32_t bc[16];
        for (auto &x : bc) x = 1;
        bitshift.init(bc, 16);
    }
};

// ---------------------------------------------------------------- bits (bitstream-io BigEndian)
struct BitWriter {
    std::vector<uint8_t> bytes;
    uint32_t acc = 0, nbits = 0;
    void bit(bool b) {
        acc = (acc << 1) | (b ? 1u : 0u);
        if (++nbits == 8) {
            bytes.push_back((uint8_t)acc);
            acc = 0;
            nbits = 0;
        }
    }
    void align() {
        while (nbits) bit(false);
    }
};
struct BitReader {
    const uint8_t *data;
    size_t size, pos = 0;  // pos in bits
    size_t over = 0;       // bits asked for past the end (the coder's tail legitimately reads a few dozen)
    bool bit() {           // Option<bool>: past the end reads as 0
        const size_t byte = pos >> 3;
        if (byte >= size) {
            ++over;
            return false;
        }
        const bool b = (data[byte] >> (7 - (pos & 7))) & 1;
        ++pos;
        return b;
    }
};

// ---------------------------------------------------------------- range coder (arithmetic-coding-adder-dep)
struct ArithEncoder {
    uint64_t low = 0, high = 1ull << kPrecision;
    uint32_t pending = 0;
    BitWriter *out;
    void emit(bool b) {
        out->bit(b);
        for (; pending; --pending) out->bit(!b);
    }
    void encode(Weights &w, size_t index) {  // encoder.rs:127-141, 234-263
        uint64_t lo, hi;
        w.range(index, lo, hi);
        const uint64_t range = high - low + 1;
        high = low + (range * hi) / w.total - 1;
        low += (range * lo) / w.total;
        while (high < kHalf || low >= kHalf) {
            if (high < kHalf) {
                emit(false);
                high <<= 1;
                low <<= 1;
            } else {
                emit(true);
                low = (low - kHalf) << 1;
                high = (high - kHalf) << 1;
            }
        }
        while (low >= kQuarter && high < kThreeQuarter) {
            ++pending;
            low = (low - kQuarter) << 1;
            high = (high - kQuarter) << 1;
        }
        if (w.total < kMaxDenominator) w.add(index, 1);  // context_switching.rs:78-95
    }
    void flush() {  // encoder.rs:275-284
        ++pending;
        emit(low > kQuarter);
    }
    void bytes(Weights &w, const uint8_t *p, size_t n) {
        for (size_t i = 0; i < n; ++i) encode(w, (size_t)p[i] + 1);
    }
};
struct ArithDecoder {
    uint64_t low = 0, high = 1ull << kPrecision, x = 0;
    bool init = false;
    bool bad = false;  // the EOF symbol was decoded, or the input ran out long ago: the caller stops (the reference's
                       // decoder returns an error at EOF; a corrupt ADU must not keep producing events)
    BitReader *in;
    size_t decode(Weights &w) {  // decoder.rs:120-141, 236-296; returns the fenwick index (0 = EOF)
        if (!init) {
            for (uint32_t i = 0; i < kPrecision; ++i) x = (x << 1) | (in->bit() ? 1u : 0u);
            init = true;
        }
        const uint64_t range = high - low + 1;
        const uint64_t value = ((x - low + 1) * w.total - 1) / range;
        const size_t index = w.find(value);
        uint64_t lo, hi;
        w.range(index, lo, hi);
        high = low + (range * hi) / w.total - 1;
        low += (range * lo) / w.total;
        while (high < kHalf || low >= kHalf) {
            if (high < kHalf) {
                high <<= 1;
                low <<= 1;
                x <<= 1;
            } else {
                low = (low - kHalf) << 1;
                high = (high - kHalf) << 1;
                x = (x - kHalf) << 1;
            }
            if (in->bit()) x += 1;
        }
        while (low >= kQuarter && high < kThreeQuarter) {
            low = (low - kQuarter) << 1;
            high = (high - kQuarter) << 1;
            x = (x - kQuarter) << 1;
            if (in->bit()) x += 1;
        }
        if (w.total < kMaxDenominator) w.add(index, 1);
        if (index == 0 || in->over > 2 * kPrecision + 64) bad = true;
        return index;
    }
    uint8_t byte(Weights &w) { return (uint8_t)(decode(w) - 1); }
};

static void be16(int16_t v, uint8_t *p) {
    p[0] = (uint8_t)((uint16_t)v >> 8);
    p[1] = (uint8_t)v;
}
static void be64(int64_t v, uint8_t *p) {
    for (int i = 0; i < 8; ++i) p[i] = (uint8_t)((uint64_t)v >> (56 - 8 * i));
}

// ---------------------------------------------------------------- the source model
struct Ev {  // EventCoordless (lib.rs:497-504)
    uint32_t d, t;
};

static double event_to_intensity(uint32_t d, uint32_t delta_t, uint32_t dt_ref) {  // cabac_contexts.rs:75-85
    double intensity;
    if (d >= 129)
        intensity = 0.0;
    else {
        const double shift = d == 128 ? 0.0 : std::ldexp(1.0, (int)d);  // D_SHIFT[d] as f64; D_SHIFT[128] == 0
        intensity = delta_t == 0 ? shift : shift / (double)delta_t;
    }
    return intensity * (double)dt_ref;
}

// cabac_contexts.rs:87-150 (the lossy bit shift of an inter-coded t residual)
static void residual_to_bitshift2(int64_t t_prediction, int64_t r, const Ev &event, const Ev &prev, uint32_t dt_ref,
                                  double c_thresh_max, uint32_t &bitshift_out, int64_t &residual_out) {
    const int64_t ar = r < 0 ? -r : r;
    if (ar < kTResidualMax) {
        bitshift_out = 0;
        residual_out = r;
        return;
    }
    const uint32_t actual_dt = event.t > prev.t ? event.t - prev.t : 0u;
    const double actual = event_to_intensity(event.d, actual_dt, dt_ref);
    double recon = actual;
    uint32_t bitshift = 0;
    int64_t tr = ar;
    for (;;) {
        if (tr > kTResidualMax && actual - c_thresh_max < recon && actual + c_thresh_max > recon) {
            tr >>= 1;
            bitshift += 1;
            const uint32_t recon_t = (uint32_t)(t_prediction + tr);
            if (recon_t < prev.t) break;
            recon = event_to_intensity(event.d, recon_t - prev.t, dt_ref);
        } else {
            break;
        }
    }
    bitshift = bitshift ? bitshift - 1 : 0;
    tr = ar >> bitshift;
    if (tr < kTResidualMax) {
        bitshift_out = bitshift;
        residual_out = r < 0 ? -tr : tr;
    } else {
        bitshift_out = kEncodeFull;
        residual_out = r;
    }
}

// event_cube.rs:83-118
static uint32_t generate_t_prediction(size_t idx, int d_residual, uint32_t last_delta_t, const Ev &prev,
                                      uint32_t num_intervals, uint32_t dt_ref, uint32_t start_t) {
    if (idx == 1) return start_t + last_delta_t;
    if (d_residual > 14 || d_residual < -14) d_residual = 0;
    if (prev.d == kDEmpty) d_residual = -1;
    const uint32_t pred = d_residual < 0 ? last_delta_t >> (uint32_t)(-d_residual) : last_delta_t << (uint32_t)d_residual;
    const uint32_t cap = (num_intervals & 0xffu) * dt_ref;
    return std::max(prev.t, prev.t + std::min(pred, cap));
}

struct CubeEvent {
    uint16_t pix;  // c * 256 + y * 16 + x inside the cube
    uint8_t d;
    uint32_t t;
};

// One cube's events in arrival order -> per-pixel lists (pixel order c, y, x; arrival order inside a pixel),
// with the ingest-time drop rule applied.  offsets[p] .. offsets[p + 1] index `lists`.
static void pixel_lists(const std::vector<CubeEvent> &in, uint32_t channels, std::vector<Ev> &lists,
                        std::vector<uint32_t> &offsets) {
    const uint32_t npix = channels * kBlock * kBlock;
    offsets.assign(npix + 1, 0);
    for (const auto &e : in) offsets[e.pix + 1] += 1;
    for (uint32_t p = 0; p < npix; ++p) offsets[p + 1] += offsets[p];
    std::vector<Ev> sorted(in.size());
    {
        std::vector<uint32_t> cur(offsets.begin(), offsets.end() - 1);
        for (const auto &e : in) sorted[cur[e.pix]++] = Ev{e.d, e.t};
    }
    // event_cube.rs:135-151: with two or more events already in the pixel's list, an event whose t is not
    // greater than the last one's is dropped
    lists.clear();
    lists.reserve(sorted.size());
    std::vector<uint32_t> kept(npix + 1, 0);
    for (uint32_t p = 0; p < npix; ++p) {
        const size_t first = lists.size();
        for (uint32_t i = offsets[p]; i < offsets[p + 1]; ++i) {
            const size_t len = lists.size() - first;
            if (len > 1 && sorted[i].t <= lists.back().t) continue;
            lists.push_back(sorted[i]);
        }
        kept[p + 1] = (uint32_t)lists.size();
    }
    offsets.swap(kept);
}

struct AduJob {  // one ADU on its way to a worker
    uint32_t id = 0;
    uint32_t start_t = 0;
    std::vector<std::vector<CubeEvent>> cubes;  // [by * bx], row-major
};

// EventAdu::compress (event_adu.rs:83-117) -> the ADU's bytes
static std::vector<uint8_t> compress_adu(AduJob &job, uint32_t channels, uint32_t dt_ref, uint32_t num_intervals,
                                         uint8_t c_thresh_max) {
    BitWriter bw;
    Model m;
    ArithEncoder enc;
    enc.out = &bw;
    uint8_t buf[8];
    buf[0] = (uint8_t)(job.start_t >> 24);
    buf[1] = (uint8_t)(job.start_t >> 16);
    buf[2] = (uint8_t)(job.start_t >> 8);
    buf[3] = (uint8_t)job.start_t;
    enc.bytes(m.t, buf, 4);
    const size_t ncubes = job.cubes.size();
    std::vector<std::vector<Ev>> lists(ncubes);
    std::vector<std::vector<uint32_t>> offs(ncubes);
    const uint32_t npix = channels * kBlock * kBlock;
    // intra: the first event of every pixel, cubes in row-major order (event_cube.rs:310-413)
    for (size_t cb = 0; cb < ncubes; ++cb) {
        if (job.cubes[cb].empty()) {  // skip_cube
            enc.encode(m.d, (size_t)(kSkipCube + kDResidualOffset) + 1);
            continue;
        }
        pixel_lists(job.cubes[cb], channels, lists[cb], offs[cb]);
        std::vector<CubeEvent>().swap(job.cubes[cb]);
        bool have_init = false;
        Ev init{0, 0};
        for (uint32_t p = 0; p < npix; ++p) {
            if (offs[cb][p] == offs[cb][p + 1]) {
                enc.encode(m.d, (size_t)(kNoEvent + kDResidualOffset) + 1);
                continue;
            }
            Ev &event = lists[cb][offs[cb][p]];
            if (have_init) {
                enc.encode(m.d, (size_t)((int)event.d - (int)init.d + kDResidualOffset) + 1);
            } else {
                enc.encode(m.d, (size_t)((int)event.d + kDResidualOffset) + 1);
                init = Ev{event.d, job.start_t};
                have_init = true;
            }
            const int64_t r = (int64_t)event.t - (int64_t)init.t;
            const bool full = !((r < 0 ? -r : r) < kTResidualMax);  // residual_to_bitshift (:49-73)
            enc.encode(m.bitshift, (size_t)(full ? kEncodeFull : 0u) + 1);
            if (full) {
                be64(r, buf);
                enc.bytes(m.t, buf, 8);
                event.t = (uint32_t)((int64_t)init.t + r);
            } else {
                const int16_t tr = (int16_t)r;
                be16(tr, buf);
                enc.bytes(m.t, buf, 2);
                event.t = (uint32_t)((int64_t)init.t + (int64_t)tr);
            }
            init = event;
        }
    }
    // inter: the later events of every pixel (event_cube.rs:415-516)
    for (size_t cb = 0; cb < ncubes; ++cb) {
        if (lists[cb].empty()) continue;
        for (uint32_t p = 0; p < npix; ++p) {
            const uint32_t a = offs[cb][p], b = offs[cb][p + 1];
            if (a == b) continue;
            uint32_t last_delta_t = 0;
            for (size_t idx = 1; a + idx < b; ++idx) {
                const Ev prev = lists[cb][a + idx - 1];
                Ev &event = lists[cb][a + idx];
                const int d_residual = (int)event.d - (int)prev.d;
                be16((int16_t)d_residual, buf);
                enc.bytes(m.d, buf, 2);
                const uint32_t t_pred =
                    generate_t_prediction(idx, d_residual, last_delta_t, prev, num_intervals, dt_ref, job.start_t);
                const int64_t r = (int64_t)event.t - (int64_t)t_pred;
                uint32_t bitshift;
                int64_t t_residual;
                residual_to_bitshift2((int64_t)t_pred, r, event, prev, dt_ref, (double)c_thresh_max, bitshift, t_residual);
                enc.encode(m.bitshift, (size_t)bitshift + 1);
                if (bitshift == kEncodeFull) {
                    be64(t_residual, buf);
                    enc.bytes(m.t, buf, 8);
                    event.t = (uint32_t)((int64_t)t_pred + t_residual);
                } else {
                    const int16_t tr = (int16_t)t_residual;
                    be16(tr, buf);
                    enc.bytes(m.t, buf, 2);
                    event.t = (uint32_t)((int64_t)t_pred + ((int64_t)tr << bitshift));
                }
                event.t = std::max(event.t, prev.t);
                last_delta_t = event.t - prev.t;
            }
            be16((int16_t)kNoEvent, buf);
            enc.bytes(m.d, buf, 2);
        }
    }
    enc.encode(m.eof, 0);  // eof_context (cabac_contexts.rs:226-238)
    enc.flush();
    bw.align();
    return std::move(bw.bytes);
}

static size_t write_header(uint8_t *h, const AdderCompressedParams &p) {  // header.rs + encoder.rs:170-229
    size_t n = 0;
    memcpy(h, "addec", 5);
    n = 5;
    h[n++] = p.codec_version;
    h[n++] = 98;
    auto put16 = [&](uint16_t v) { h[n++] = (uint8_t)(v >> 8); h[n++] = (uint8_t)v; };
    auto put32 = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) h[n++] = (uint8_t)(v >> s); };
    put16(p.width);
    put16(p.height);
    put32(p.tps);
    put32(p.ref_interval);
    put32(p.delta_t_max);
    h[n++] = p.channels == 1 ? 9 : 11;
    h[n++] = p.channels;
    if (p.codec_version >= 1) put32(p.source_camera);
    if (p.codec_version >= 2) put32(p.time_mode);
    if (p.codec_version >= 3) put32(p.adu_interval);
    return n;
}

}  // namespace

struct AdderCompressedEncoder {
    AdderCompressedParams p{};
    uint32_t by = 0, bx = 0;
    uint32_t start_t = 0;  // EventAdu::start_t
    bool skip_adu = true;
    std::vector<std::vector<CubeEvent>> cubes;  // the ADU being filled
    std::vector<uint8_t> stream;                // header + finished ADUs, in order
    uint32_t next_id = 1, written = 0;
    // workers
    std::vector<std::thread> pool;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<std::unique_ptr<AduJob>> jobs;
    std::vector<std::pair<uint32_t, std::vector<uint8_t>>> done;  // finished, not yet appended (any order)
    bool stopping = false;
    bool closed = false;
    std::string err;
};

static thread_local std::string g_cerr;

static int cfail(AdderCompressedEncoder *e, int code, const char *fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e)
        e->err = buf;
    else
        g_cerr = buf;
    return code;
}

static void worker_loop(AdderCompressedEncoder *e) {
    for (;;) {
        std::unique_ptr<AduJob> job;
        {
            std::unique_lock<std::mutex> lk(e->mu);
            e->cv_job.wait(lk, [&] { return e->stopping || !e->jobs.empty(); });
            if (e->jobs.empty()) return;
            job = std::move(e->jobs.front());
            e->jobs.pop_front();
        }
        std::vector<uint8_t> bytes =
            compress_adu(*job, e->p.channels, e->p.ref_interval, e->p.adu_interval, e->p.c_thresh_max);
        {
            std::lock_guard<std::mutex> lk(e->mu);
            e->done.emplace_back(job->id, std::move(bytes));
        }
        e->cv_done.notify_all();
    }
}

// appends the finished ADUs that are next in order (flush_bytes_queue_worker, stream.rs:78-104); mu held
static void drain_in_order(AdderCompressedEncoder *e) {
    for (bool progress = true; progress;) {
        progress = false;
        for (size_t i = 0; i < e->done.size(); ++i) {
            if (e->done[i].first != e->written + 1) continue;
            const std::vector<uint8_t> &b = e->done[i].second;
            const uint32_t n = (uint32_t)b.size();
            const uint8_t len[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
            e->stream.insert(e->stream.end(), len, len + 4);
            e->stream.insert(e->stream.end(), b.begin(), b.end());
            e->done.erase(e->done.begin() + (long)i);
            e->written += 1;
            progress = true;
            break;
        }
    }
}

static void submit_adu(AdderCompressedEncoder *e) {
    std::unique_ptr<AduJob> job(new AduJob());
    job->id = e->next_id++;
    job->start_t = e->start_t;
    job->cubes.swap(e->cubes);
    e->cubes.assign((size_t)e->by * e->bx, {});
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->jobs.push_back(std::move(job));
        drain_in_order(e);
    }
    e->cv_job.notify_one();
    // clear_compression (event_adu.rs:219-227)
    e->skip_adu = true;
    e->start_t += e->p.adu_interval * e->p.ref_interval;
}

extern "C" void adder_compressed_default_params(AdderCompressedParams *p, uint16_t width, uint16_t height,
                                                uint8_t channels) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->abi_version = ADDER_COMPRESSED_ABI_VERSION;
    p->width = width;
    p->height = height;
    p->channels = channels;
    p->codec_version = 3;               // LATEST_CODEC_VERSION
    p->time_mode = ADDER_TIME_ABSOLUTE_T;
    p->write_header = 1;
    p->tps = 2550;                      // CodecMetadata::default (codec/mod.rs:94-107)
    p->ref_interval = 255;
    p->delta_t_max = 255;
    p->adu_interval = 1;
    p->source_camera = 0;
    p->c_thresh_max = 7;                // Crf::new(None) -> quality 3 (rate_controller.rs)
    p->threads = 0;
}

extern "C" int adder_compressed_encoder_create(const AdderCompressedParams *p, AdderCompressedEncoder **out) {
    if (!out) return cfail(nullptr, ADDER_E_BAD_PARAMS, "out is null");
    *out = nullptr;
    if (!p || p->abi_version != ADDER_COMPRESSED_ABI_VERSION) return cfail(nullptr, ADDER_E_BAD_PARAMS, "bad params / abi_version");
    if (!p->width || !p->height || (p->channels != 1 && p->channels != 3))
        return cfail(nullptr, ADDER_E_BAD_PARAMS, "bad plane");
    if (!p->ref_interval || !p->adu_interval || !p->delta_t_max || p->codec_version > 3)
        return cfail(nullptr, ADDER_E_BAD_PARAMS, "bad ref_interval / adu_interval / delta_t_max / codec_version");
    AdderCompressedEncoder *e = new (std::nothrow) AdderCompressedEncoder();
    if (!e) return cfail(nullptr, ADDER_E_HIP, "out of memory");
    e->p = *p;
    e->by = (p->height + kBlock - 1) / kBlock;
    e->bx = (p->width + kBlock - 1) / kBlock;
    e->cubes.assign((size_t)e->by * e->bx, {});
    if (p->write_header) {
        uint8_t h[64];
        const size_t n = write_header(h, *p);
        e->stream.assign(h, h + n);
    }
    uint32_t nt = p->threads ? p->threads : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    for (uint32_t i = 0; i < nt; ++i) e->pool.emplace_back(worker_loop, e);
    *out = e;
    return ADDER_OK;
}

extern "C" void adder_compressed_encoder_destroy(AdderCompressedEncoder *e) {
    if (!e) return;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->stopping = true;
        e->jobs.clear();
    }
    e->cv_job.notify_all();
    for (auto &t : e->pool) t.join();
    delete e;
}

extern "C" const char *adder_compressed_last_error(const AdderCompressedEncoder *e) {
    return e ? e->err.c_str() : g_cerr.c_str();
}

extern "C" int adder_compressed_encoder_ingest(AdderCompressedEncoder *e, const AdderEvent *events, size_t n) {
    if (!e || (!events && n)) return ADDER_E_BAD_PARAMS;
    if (e->closed) return cfail(e, ADDER_E_BAD_PARAMS, "encoder is closed");
    const uint32_t span = e->p.ref_interval * e->p.adu_interval;
    for (size_t i = 0; i < n; ++i) {
        const AdderEvent &ev = events[i];
        if (ev.x >= e->p.width || ev.y >= e->p.height) return cfail(e, ADDER_E_BAD_PARAMS, "event outside the plane");
        const uint32_t c = ev.c == ADDER_C_NONE ? 0u : ev.c;
        if (c >= e->p.channels) return cfail(e, ADDER_E_BAD_PARAMS, "event channel outside the plane");
        // stream.rs:270-313: an event past the ADU's time range closes it (one ADU per event at most)
        if (ev.t > e->start_t + span) submit_adu(e);
        const uint32_t cb = (ev.y / kBlock) * e->bx + ev.x / kBlock;
        e->cubes[cb].push_back(CubeEvent{(uint16_t)(c * 256u + (ev.y % kBlock) * kBlock + ev.x % kBlock), ev.d, ev.t});
        e->skip_adu = false;
    }
    return ADDER_OK;
}

extern "C" int adder_compressed_encoder_progress(AdderCompressedEncoder *e, uint32_t *adus_written, size_t *n_bytes) {
    if (!e) return ADDER_E_BAD_PARAMS;
    std::lock_guard<std::mutex> lk(e->mu);
    drain_in_order(e);
    if (adus_written) *adus_written = e->written;
    if (n_bytes) *n_bytes = e->stream.size();
    return ADDER_OK;
}

extern "C" int adder_compressed_encoder_close(AdderCompressedEncoder *e, const uint8_t **bytes, size_t *n_bytes) {
    if (!e) return ADDER_E_BAD_PARAMS;
    if (!e->closed) {
        if (!e->skip_adu) submit_adu(e);  // into_writer: the partial last ADU (stream.rs:180-229)
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv_done.wait(lk, [&] {
            drain_in_order(e);
            return e->written + 1 == e->next_id;
        });
        e->closed = true;
    }
    if (bytes) *bytes = e->stream.data();
    if (n_bytes) *n_bytes = e->stream.size();
    return ADDER_OK;
}

// ---------------------------------------------------------------- decoding
extern "C" int adder_compressed_decode(const uint8_t *data, size_t size, int has_header, AdderCompressedParams *params,
                                       AdderEvent *out, size_t out_cap, size_t *n_out) {
    if (n_out) *n_out = 0;
    if (!data || !params) return ADDER_E_BAD_PARAMS;
    size_t pos = 0;
    if (has_header) {
        if (size < 25 || memcmp(data, "addec", 5) != 0) return cfail(nullptr, ADDER_E_BAD_PARAMS, "not a compressed ADDER stream");
        auto rd16 = [&](size_t o) { return (uint16_t)((data[o] << 8) | data[o + 1]); };
        auto rd32 = [&](size_t o) {
            return ((uint32_t)data[o] << 24) | ((uint32_t)data[o + 1] << 16) | ((uint32_t)data[o + 2] << 8) | data[o + 3];
        };
        params->codec_version = data[5];
        params->width = rd16(7);
        params->height = rd16(9);
        params->tps = rd32(11);
        params->ref_interval = rd32(15);
        params->delta_t_max = rd32(19);
        params->channels = data[24];
        pos = 25;
        if (params->codec_version >= 1) { if (size < pos + 4) return ADDER_E_BAD_PARAMS; params->source_camera = rd32(pos); pos += 4; }
        if (params->codec_version >= 2) { if (size < pos + 4) return ADDER_E_BAD_PARAMS; params->time_mode = (uint8_t)rd32(pos); pos += 4; }
        if (params->codec_version >= 3) { if (size < pos + 4) return ADDER_E_BAD_PARAMS; params->adu_interval = rd32(pos); pos += 4; }
    }
    const AdderCompressedParams p = *params;
    if (!p.width || !p.height || (p.channels != 1 && p.channels != 3) || !p.ref_interval || !p.adu_interval)
        return cfail(nullptr, ADDER_E_BAD_PARAMS, "bad stream parameters");
    const uint32_t by = (p.height + kBlock - 1) / kBlock, bx = (p.width + kBlock - 1) / kBlock;
    const uint32_t npix = p.channels * kBlock * kBlock;
    // (parameters from a header or from the caller: one event list per pixel-channel is allocated below)
    if ((uint64_t)by * bx * npix > (1ull << 28)) return cfail(nullptr, ADDER_E_BAD_PARAMS, "plane too large to decode");
    uint32_t start_t = 0;
    bool first_run = true;
    size_t total = 0;
    bool overflow = false;
    std::vector<std::vector<std::vector<Ev>>> px((size_t)by * bx);  // [cube][pixel] lists of one ADU
    while (pos + 4 <= size) {
        const uint32_t nbytes = ((uint32_t)data[pos] << 24) | ((uint32_t)data[pos + 1] << 16) | ((uint32_t)data[pos + 2] << 8) | data[pos + 3];
        pos += 4;
        if (pos + nbytes > size) return cfail(nullptr, ADDER_E_BAD_PARAMS, "truncated ADU");
        if (!first_run) start_t += p.adu_interval * p.ref_interval;  // clear_decompression (event_adu.rs:229-240)
        first_run = false;
        BitReader br{data + pos, nbytes};
        pos += nbytes;
        Model m;
        ArithDecoder dec;
        dec.in = &br;
        for (int i = 0; i < 4; ++i) (void)dec.byte(m.t);  // the ADU's start_t: read, not used (event_adu.rs:131-137)
        // An adaptive model can spend far less than a bit per event, so the bytes of an ADU do not bound its events
        // tightly; this cap only guarantees termination (and bounded memory) on corrupt input.
        const size_t adu_event_cap = std::max<size_t>((size_t)1 << 22, (size_t)nbytes * 4096);
        size_t adu_events = 0;
        std::vector<uint8_t> skipped((size_t)by * bx, 0);
        for (size_t cb = 0; cb < px.size(); ++cb) {  // decompress_intra (event_cube.rs:518-599)
            px[cb].assign(npix, {});
            bool have_init = false;
            Ev init{0, start_t};
            skipped[cb] = 1;
            for (uint32_t q = 0; q < npix; ++q) {
                const int d_residual = (int)dec.decode(m.d) - 1 - kDResidualOffset;
                if (d_residual == kSkipCube) {
                    skipped[cb] = 1;
                    break;
                }
                if (d_residual == kNoEvent) continue;
                uint32_t d;
                if (have_init) {
                    d = (uint32_t)((int)init.d + d_residual) & 0xffu;
                } else {
                    have_init = true;
                    init = Ev{0, start_t};
                    d = (uint32_t)d_residual & 0xffu;
                }
                skipped[cb] = 0;
                const uint32_t bitshift = (uint32_t)dec.byte(m.bitshift);
                int64_t t_residual;
                if (bitshift == kEncodeFull) {
                    uint64_t v = 0;
                    for (int i = 0; i < 8; ++i) v = (v << 8) | dec.byte(m.t);
                    t_residual = (int64_t)v;
                } else {
                    uint16_t v = 0;
                    for (int i = 0; i < 2; ++i) v = (uint16_t)((v << 8) | dec.byte(m.t));
                    t_residual = (int64_t)(int16_t)v << bitshift;
                }
                init.d = (uint32_t)((int)init.d + d_residual) & 0xffu;
                init.t = (uint32_t)((int64_t)init.t + t_residual);
                px[cb][q].push_back(Ev{d, init.t});
                ++adu_events;
                if (dec.bad) return cfail(nullptr, ADDER_E_BAD_PARAMS, "corrupt or truncated ADU (end of the coded data inside a cube)");
            }
            if (dec.bad) return cfail(nullptr, ADDER_E_BAD_PARAMS, "corrupt or truncated ADU (end of the coded data inside a cube)");
        }
        for (size_t cb = 0; cb < px.size(); ++cb) {  // decompress_inter (:601-680)
            if (skipped[cb]) continue;
            for (uint32_t q = 0; q < npix; ++q) {
                auto &list = px[cb][q];
                if (list.empty()) continue;
                uint32_t last_delta_t = 0;
                for (size_t idx = 1;; ++idx) {
                    uint16_t raw = 0;
                    for (int i = 0; i < 2; ++i) raw = (uint16_t)((raw << 8) | dec.byte(m.d));
                    const int d_residual = (int16_t)raw;
                    if (d_residual == kNoEvent) break;
                    const Ev prev = list[idx - 1];
                    const uint32_t d = (uint32_t)((int)prev.d + d_residual) & 0xffu;
                    const uint32_t t_pred = generate_t_prediction(idx, d_residual, last_delta_t, prev, p.adu_interval, p.ref_interval, start_t);
                    const uint32_t bitshift = (uint32_t)dec.byte(m.bitshift);
                    int64_t t_residual;
                    if (bitshift == kEncodeFull) {
                        uint64_t v = 0;
                        for (int i = 0; i < 8; ++i) v = (v << 8) | dec.byte(m.t);
                        t_residual = (int64_t)v;
                    } else {
                        uint16_t v = 0;
                        for (int i = 0; i < 2; ++i) v = (uint16_t)((v << 8) | dec.byte(m.t));
                        t_residual = (int64_t)(int16_t)v << bitshift;
                    }
                    const uint32_t t = std::max((uint32_t)((int64_t)t_pred + t_residual), prev.t);
                    last_delta_t = t - prev.t;
                    list.push_back(Ev{d, t});
                    if (dec.bad || ++adu_events > adu_event_cap)
                        return cfail(nullptr, ADDER_E_BAD_PARAMS, "corrupt or truncated ADU (no end marker for a pixel's events)");
                }
                if (dec.bad) return cfail(nullptr, ADDER_E_BAD_PARAMS, "corrupt or truncated ADU (end of the coded data inside a cube)");
            }
        }
        // digest order (event_adu.rs:195-217, event_cube.rs:165-210): cubes row-major; c, y, x; list order
        for (uint32_t cy = 0; cy < by; ++cy)
            for (uint32_t cx = 0; cx < bx; ++cx) {
                const size_t cb = (size_t)cy * bx + cx;
                if (skipped[cb]) continue;
                for (uint32_t q = 0; q < npix; ++q)
                    for (const Ev &ev : px[cb][q]) {
                        if (out && total < out_cap) {
                            AdderEvent o;
                            o.x = (uint16_t)(cx * kBlock + q % kBlock);
                            o.y = (uint16_t)(cy * kBlock + (q / kBlock) % kBlock);
                            o.c = p.channels == 1 ? ADDER_C_NONE : (uint8_t)(q / (kBlock * kBlock));
                            o.d = (uint8_t)ev.d;
                            o.pad = 0;
                            o.t = ev.t;
                            out[total] = o;
                        } else if (out) {
                            overflow = true;
                        }
                        ++total;
                    }
            }
    }
    if (n_out) *n_out = total;
    return overflow ? ADDER_E_OUT_CAPACITY : ADDER_OK;
}
