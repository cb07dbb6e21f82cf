// adder_framer_kernels.hip -- gfx950 kernels of the instantaneous framer (events -> u8 frames),
// the consumer side of the transcode path (framer/driver.rs:437-626, 984-1133).
//
// The reference walks the events one by one; the only ordering that matters is the order of
// the events OF ONE PIXEL (driver.rs:1062-1070).  A launch takes one SEGMENT of the stream in
// which every pixel-channel's events are contiguous (each per-frame segment the transcoder
// emits has that shape: raster order, per-pixel emission order): a thread per event, the
// first thread of a run of equal coordinates walks the run.  Segments are launched in stream
// order on one stream, so a pixel's events are applied in order.  HBM-bound: 12 B read per
// event + 13 B of tracker state read/written per touched pixel + one byte per (pixel, frame)
// written; no atomics on the data path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "adder_framer_kernels.h"

namespace adder {

// an element of the frame ring: u8 / u16 / u32 by FramerConsts::value_type (host byte order; the pop kernel writes the
// reference's big-endian bincode bytes)
__device__ __forceinline__ void framer_ring_store(uint8_t *ring, size_t idx, uint32_t value_type, uint32_t v) {
    if (value_type == 0u) ring[idx] = (uint8_t)v;
    else if (value_type == 1u) reinterpret_cast<uint16_t *>(ring)[idx] = (uint16_t)v;
    else reinterpret_cast<uint32_t *>(ring)[idx] = v;
}

__device__ __forceinline__ uint32_t unit_of(const FramerArgs &a, uint32_t xy, uint32_t cd, bool &ok) {
    const uint32_t x = xy & 0xffffu, y = xy >> 16;
    uint32_t c = cd & 0xffu;
    c = c == 0xffu ? 0u : c;  // coord.c.unwrap_or(0)
    ok = x < a.width && y >= a.row_begin && y - a.row_begin < a.rows && c < a.channels;
    return ((y - a.row_begin) * a.width + x) * a.channels + c;
}

// Event i of the range [e0, e1) in which every pixel-channel's events are contiguous: the first
// event of a run of equal coordinates walks the run and applies it to the pixel's trackers.
// (xy, cd, t) = event i, (pxy, pcd) = event i - 1 (only read when i > e0) and (nxy, ncd) = event i + 1
// (only read when i + 1 < e1), all loaded by the caller in one round trip: most runs have a single
// event, and the walk must not pay a dependent load to find that out.
__device__ __forceinline__ void framer_apply_run(const uint32_t *__restrict__ ev, uint64_t i, uint64_t e0, uint64_t e1,
                                                 const FramerArgs &a, uint32_t y_lo, uint32_t y_hi, uint32_t xy,
                                                 uint32_t cd, uint32_t t0, uint32_t pxy, uint32_t pcd, uint32_t nxy,
                                                 uint32_t ncd) {
    const uint32_t key_c = (cd & 0xffu) == 0xffu ? 0u : (cd & 0xffu);
    if (i > e0) {  // not the first event of its pixel's run: the run's leader handles it
        const uint32_t pc = (pcd & 0xffu) == 0xffu ? 0u : (pcd & 0xffu);
        if (pxy == xy && pc == key_c) return;
    }
    bool ok;
    const uint32_t u = unit_of(a, xy, cd, ok);
    const uint32_t y = xy >> 16;
    if (!ok || y < y_lo || y >= y_hi) {
        atomicOr(a.status, kFramerStatusMalformed);
        return;
    }
    FramerPx p = a.px[u];  // one 16-byte access
    uint32_t flags = 0u;
    for (uint64_t j = i; j < e1; ++j) {
        uint32_t w1 = cd, t = t0;
        if (j != i) {
            const uint32_t jxy = j == i + 1 ? nxy : ev[3 * j];
            w1 = j == i + 1 ? ncd : ev[3 * j + 1];
            const uint32_t jc = (w1 & 0xffu) == 0xffu ? 0u : (w1 & 0xffu);
            if (jxy != xy || jc != key_c) break;
            t = ev[3 * j + 2];
        }
        int32_t from = 0, to = 0;
        bool overflow = false;
        const bool fills = framer_step(p, (w1 >> 8) & 0xffu, t, a.k, from, to, overflow);
        if (overflow) flags |= kFramerStatusRange;
        if (fills) {
            // frames (from, to]; those already handed out (below frames_written) are skipped (:1074-1075)
            int32_t f = from + 1 > a.frames_written ? from + 1 : a.frames_written;
            for (; f <= to; ++f) {
                if ((uint32_t)(f - a.frames_written) >= a.ring_frames) {
                    flags |= kFramerStatusRing;
                    break;
                }
                framer_ring_store(a.ring, (size_t)((uint32_t)f % a.ring_frames) * a.n_units + u, a.k.value_type, p.lasti);
            }
        }
    }
    a.px[u] = p;
    if (flags) atomicOr(a.status, flags);
}

// events [e0, e1) of one segment; ev = 3 dwords per event {x | y<<16, c | d<<8, t}
__global__ __launch_bounds__(256) void adder_framer_segment_kernel(const uint32_t *__restrict__ ev, uint64_t e0,
                                                                   uint64_t e1, FramerArgs a) {
    const uint64_t i = e0 + (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= e1) return;
    const uint64_t ip = i > e0 ? i - 1 : i, in = i + 1 < e1 ? i + 1 : i;
    framer_apply_run(ev, i, e0, e1, a, 0u, 0xffffffffu, ev[3 * i], ev[3 * i + 1], ev[3 * i + 2], ev[3 * ip],
                     ev[3 * ip + 1], ev[3 * in], ev[3 * in + 1]);
}

// The transcoder's streams: T per-frame segments, each in raster order (= ascending unit index, a unit's events
// adjacent).  Two kernels per batch:
//   adder_framer_slices_kernel  a thread per (frame, tile boundary): where tile * kTileUnits starts in the frame's
//                               segment (binary search on the unit index);
//   adder_framer_tiles_kernel   a WAVE owns a tile of kTileUnits consecutive units for the whole batch.  Their
//       trackers live in LDS, and so does a window of `K` output-frame rows of the tile: every (unit, frame) byte is
//       written exactly once over the stream, but at the time the unit's NEXT event arrives -- scattered single
//       bytes in HBM (measured: 15x write amplification) -- so the fills go into the window and whole rows are
//       loaded when they enter it and stored when they leave it.  A batch of 64 events is applied lane-parallel:
//       a unit's events are adjacent lanes, and what couples them (the unit's clock, last_filled, the last intensity)
//       is a segmented prefix sum / max over those lanes, not a serial walk.
// The trackers of a unit are touched by its owner only and a wave's LDS operations are ordered, so there is no
// barrier and no HBM round trip between a pixel's consecutive events.
#ifndef ADDER_FRAMER_TILE
#define ADDER_FRAMER_TILE 256
#endif
constexpr uint32_t kTileUnits = ADDER_FRAMER_TILE;  // 128 / 256 / 512; 256 measured best at 0.3 events per unit-frame:
                                                    // small enough for ~16 waves of LDS per CU, two batches per frame
using FramerRowVec = std::conditional<kTileUnits == 512, uint2, std::conditional<kTileUnits == 256, uint32_t, uint16_t>::type>::type;  // a tile row = one of these per lane
static_assert(sizeof(FramerRowVec) * 64 == kTileUnits, "tile size");

__device__ __forceinline__ uint32_t framer_unit_key(const FramerArgs &a, uint32_t xy, uint32_t cd) {
    uint32_t c = cd & 0xffu;
    c = c == 0xffu ? 0u : c;
    return (((xy >> 16) - a.row_begin) * a.width + (xy & 0xffffu)) * a.channels + c;
}
__device__ __forceinline__ uint64_t framer_lower_bound_unit(const uint32_t *__restrict__ ev, uint64_t lo, uint64_t hi,
                                                            uint32_t unit, const FramerArgs &a) {
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (framer_unit_key(a, ev[3 * mid], ev[3 * mid + 1]) < unit)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// tile_off[f * (ntiles + 1) + tile]: both ends of every slice come from a search over the WHOLE segment, so the
// slices tile it whatever the events look like; the tiles kernel checks each slice event by event
__global__ __launch_bounds__(256) void adder_framer_slices_kernel(const uint32_t *__restrict__ ev,
                                                                  const uint64_t *__restrict__ seg_offsets, uint32_t T,
                                                                  uint32_t ntiles, uint64_t *__restrict__ tile_off,
                                                                  FramerArgs a) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint64_t)T * (ntiles + 1u)) return;
    const uint32_t f = (uint32_t)(idx / (ntiles + 1u)), tile = (uint32_t)(idx % (ntiles + 1u));
    const uint64_t b0 = seg_offsets[f];
    uint64_t b1 = seg_offsets[f + 1];
    if (b1 < b0) {  // offsets that live on the device are first looked at here
        if (tile == 0u) atomicOr(a.status, kFramerStatusMalformed);
        b1 = b0;
    }
    // (an interpolated start + galloping was tried: slower, events cluster where the picture moves)
    tile_off[idx] = tile == 0u ? b0 : tile == ntiles ? b1 : framer_lower_bound_unit(ev, b0, b1, tile * kTileUnits, a);
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {  // l must be wave-uniform
    const uint32_t sl = __builtin_amdgcn_readfirstlane(l);
    return (uint64_t)__builtin_amdgcn_readlane((uint32_t)v, sl) |
           ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), sl) << 32);
}

// the value of the lane below (lane 0 gets 0): one DPP move, no LDS crossbar round trip
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

struct FramerEv {
    uint32_t xy, cd, t;
};
// lanes of a wave hold events base + lane of [lo, hi)
__device__ __forceinline__ FramerEv framer_fetch(const uint32_t *__restrict__ ev, uint64_t i, uint64_t hi) {
    FramerEv e{};
    if (i < hi) {
        e.xy = ev[3 * i];
        e.cd = ev[3 * i + 1];
        e.t = ev[3 * i + 2];
    }
    return e;
}

// inclusive scan over the lanes [head, lane] of the lane's own run (runs are sets of adjacent lanes)
template <class V, class Op>
__device__ __forceinline__ V framer_seg_scan(V v, uint32_t lane, uint32_t head, Op op) {
    for (uint32_t d = 1; __builtin_amdgcn_ballot_w64(lane >= head + d) != 0ull; d <<= 1) {
        const V o = __shfl_up(v, d, 64);
        if (lane >= head + d) v = op(v, o);
    }
    return v;
}

struct FramerLaneOut {
    bool fills;
    int32_t from, to;  // frames (from, to] take `value`
    uint32_t value;
    uint64_t ts_post;  // the run's trackers after this lane's event
    int32_t lastf_post;
    uint32_t flags;
};

// The common case: no run of the batch is longer than 4 lanes and every clock fits 32 bits.  A lane then gathers
// what its (at most 3) predecessors contribute with three wave shifts per quantity -- independent moves, not the
// dependent rounds of a scan.  npred = number of lanes of the same run below this one.
__device__ __forceinline__ FramerLaneOut framer_batch_step_short(bool active, uint32_t npred, uint32_t d, uint32_t t,
                                                                 const FramerPx &p0, const FramerConsts &k) {
    FramerLaneOut o{};
    const uint32_t R = k.ref_interval;
    auto rnd = [&](uint32_t x) -> uint32_t {  // driver.rs:1093-1107
        if (!k.round_up) return x;
        const uint32_t q = fast_div(x, k.by_ref);
        return x - q * R > 0u ? (q + 1u) * R : x;
    };
    const uint32_t ts0 = (uint32_t)p0.ts;
    const bool p1 = npred >= 1u, p2 = npred >= 2u, p3 = npred >= 3u;
    uint32_t ts_pre, ts_post, te = t, prev_clock;
    bool ignored = false;
    if (k.abs_t) {
        const uint32_t r = active ? rnd(t) : 0u;
        const uint32_t r1 = wave_shr1(r), r2 = wave_shr1(r1), r3 = wave_shr1(r2);
        uint32_t prev = ts0;
        prev = p1 && r1 > prev ? r1 : prev;
        prev = p2 && r2 > prev ? r2 : prev;
        prev = p3 && r3 > prev ? r3 : prev;
        ignored = prev >= t;
        ts_pre = t;
        ts_post = ignored ? prev : r;
        if (k.view_mode != kViewSae) te = t > prev ? t - prev : 0u;
        prev_clock = prev;
    } else {
        const uint32_t inc = !active ? 0u : npred == 0u ? rnd(ts0 + t) - ts0 : rnd(t);
        const uint32_t i1 = wave_shr1(inc), i2 = wave_shr1(i1), i3 = wave_shr1(i2);
        const uint32_t excl = (p1 ? i1 : 0u) + (p2 ? i2 : 0u) + (p3 ? i3 : 0u);
        ts_pre = ts0 + excl + t;
        ts_post = ts0 + excl + inc;
        prev_clock = ts0 + excl;
    }
    const uint32_t qq = fast_div(ts_pre ? ts_pre - 1u : 0u, k.by_tpf);
    const bool overflow = active && !ignored && qq > (uint32_t)kFramerMaxFrame;
    if (overflow) o.flags |= 4u;  // kFramerStatusRange
    const bool valid = active && !ignored && !overflow;
    const int32_t qv = valid ? (int32_t)qq : -0x7fffffff - 1;
    const int32_t q1 = (int32_t)wave_shr1((uint32_t)qv), q2 = (int32_t)wave_shr1((uint32_t)q1),
                  q3 = (int32_t)wave_shr1((uint32_t)q2);
    int32_t lastf_prev = p0.lastf;
    lastf_prev = p1 && q1 > lastf_prev ? q1 : lastf_prev;
    lastf_prev = p2 && q2 > lastf_prev ? q2 : lastf_prev;
    lastf_prev = p3 && q3 > lastf_prev ? q3 : lastf_prev;
    o.fills = valid && (int32_t)qq > lastf_prev;
    o.from = lastf_prev;
    o.to = (int32_t)qq;
    o.lastf_post = qv > lastf_prev ? qv : lastf_prev;
    const bool own = o.fills && d != 255u;
    const uint32_t enc = own ? 0x100u | framer_value_u8(d, te, ts_pre, prev_clock, k) : 0u;
    const uint32_t e1 = wave_shr1(enc), e2 = wave_shr1(e1), e3 = wave_shr1(e2);
    o.value = own ? enc & 0xffu
              : (p1 && (e1 & 0x100u)) ? e1 & 0xffu
              : (p2 && (e2 & 0x100u)) ? e2 & 0xffu
              : (p3 && (e3 & 0x100u)) ? e3 & 0xffu : p0.lasti;
    o.ts_post = ts_post;
    return o;
}

// What framer_step does to a unit's trackers for each event of its run, for all 64 lanes at once.  TS = uint32_t
// when every clock of the batch stays below 2^32 (checked by the caller), else uint64_t.
template <class TS>
__device__ __forceinline__ FramerLaneOut framer_batch_step(bool active, uint32_t lane, uint32_t head, uint32_t d,
                                                           uint32_t t, const FramerPx &p0, const FramerConsts &k) {
    FramerLaneOut o{};
    const TS R = (TS)k.ref_interval;
    auto rnd = [&](TS x) -> TS {  // driver.rs:1093-1107
        if (!k.round_up) return x;
        TS q;
        if (sizeof(TS) == 4) q = (TS)fast_div((uint32_t)x, k.by_ref);
        else q = x / R;
        return x - q * R > 0 ? (q + 1) * R : x;
    };
    const TS ts0 = (TS)p0.ts;
    TS ts_pre, ts_post, prev;
    bool ignored = false;
    uint32_t te = t;
    if (k.abs_t) {
        // the clock is the running maximum of the (rounded) event times: an event from the pixel's past is skipped
        // (:1002-1007) and cannot raise it, because the clock it is compared with is itself a rounded time
        const TS r = active ? rnd((TS)t) : (TS)0;
        const TS m = framer_seg_scan(r, lane, head, [](TS x, TS y) { return x > y ? x : y; });
        const TS m_up = __shfl_up(m, 1, 64);
        prev = lane > head ? (m_up > ts0 ? m_up : ts0) : ts0;
        ignored = prev >= (TS)t;
        ts_pre = (TS)t;
        ts_post = ignored ? prev : r;
        const uint32_t pr = (uint32_t)prev;
        if (k.view_mode != kViewSae) te = t > pr ? t - pr : 0u;
    } else {
        // after its first event of the batch the clock is a multiple of ref_interval (or nothing is rounded), so
        // the later events add their own rounded t: a prefix sum
        const TS inc = !active ? (TS)0 : lane == head ? (TS)(rnd(ts0 + (TS)t) - ts0) : rnd((TS)t);
        const TS incl = framer_seg_scan(inc, lane, head, [](TS x, TS y) { return (TS)(x + y); });
        ts_post = ts0 + incl;
        prev = ts0 + (incl - inc);
        ts_pre = prev + (TS)t;
    }
    const TS rm1 = ts_pre ? ts_pre - 1 : 0;
    TS qq;
    if (sizeof(TS) == 4) qq = (TS)fast_div((uint32_t)rm1, k.by_tpf);
    else qq = rm1 / (TS)k.tpf;
    const bool overflow = active && !ignored && qq > (TS)kFramerMaxFrame;
    if (overflow) o.flags |= 4u;  // kFramerStatusRange
    const bool valid = active && !ignored && !overflow;
    const int32_t qv = valid ? (int32_t)qq : -0x7fffffff - 1;
    const int32_t qm = framer_seg_scan(qv, lane, head, [](int32_t x, int32_t y) { return x > y ? x : y; });
    const int32_t qm_up = __shfl_up(qm, 1, 64);
    const int32_t lastf_prev = lane > head ? (qm_up > p0.lastf ? qm_up : p0.lastf) : p0.lastf;
    o.fills = valid && (int32_t)qq > lastf_prev;
    o.from = lastf_prev;
    o.to = (int32_t)qq;
    o.lastf_post = qm > p0.lastf ? qm : p0.lastf;
    // the intensity a fill takes: this event's, or for a D_EMPTY filler the last one before it (:1017-1019)
    const bool own = o.fills && d != 255u;
    const uint32_t enc = own ? ((lane + 1u) << 8) | framer_value_u8(d, te, (uint32_t)ts_pre, (uint32_t)prev, k) : 0u;
    const uint32_t em = framer_seg_scan(enc, lane, head, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    o.value = em ? (em & 0xffu) : p0.lasti;
    o.ts_post = (uint64_t)ts_post;
    return o;
}

// rows of the LDS window are 8 bytes longer than a tile row: the rows of one unit then fall into different banks
// (a pixel that wakes up fills many consecutive frames at one column)
constexpr uint32_t kWinStride = kTileUnits + 8u;

struct FramerCursor {  // wave-uniform: frame f of the group, its slice [lo, hi), the batch of 64 events at base
    uint32_t f;
    uint64_t lo, hi, base;
};

#ifndef ADDER_FRAMER_WAVES
#define ADDER_FRAMER_WAVES 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ADDER_FRAMER_WAVES, ADDER_FRAMER_WAVES)))
void adder_framer_tiles_kernel(const uint32_t *__restrict__ ev,
                                                                const uint64_t *__restrict__ tile_off, uint32_t T,
                                                                uint32_t ntiles, uint32_t K, FramerArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    FramerPx *const px = reinterpret_cast<FramerPx *>(s_mem);    // [kTileUnits]
    uint8_t *const win = s_mem + kTileUnits * sizeof(FramerPx);  // [K][kWinStride], row F at F & (K - 1)
    const uint32_t lane = threadIdx.x, tile = blockIdx.x;
    const uint32_t w0 = tile * kTileUnits;
    const uint32_t w1 = w0 + kTileUnits < a.n_units ? w0 + kTileUnits : a.n_units;
    const uint32_t nu = w1 - w0;
    const int32_t Ki = (int32_t)K;
    uint32_t flags = 0u;

    int32_t mn = 0x7fffffff;
    for (uint32_t q = lane; q < nu; q += 64u) {
        const FramerPx p = a.px[w0 + q];
        px[q] = p;
        mn = p.lastf < mn ? p.lastf : mn;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t v = __shfl_xor(mn, o, 64);
        mn = v < mn ? v : mn;
    }
    // rows of the tile in the frame ring <-> window rows; a row is 8 bytes per lane when the layout allows it
    const bool wide = nu == kTileUnits && (a.n_units % sizeof(FramerRowVec)) == 0u;
    auto ring_slot = [&](int32_t F) { return (uint32_t)F - fast_div((uint32_t)F, a.by_ring) * a.ring_frames; };
    auto row_ok = [&](int32_t F) { return F >= a.frames_written && (uint32_t)(F - a.frames_written) < a.ring_frames; };
    auto row_ptr = [&](int32_t F) { return a.ring + (size_t)ring_slot(F) * a.n_units + w0; };
    auto win_row = [&](int32_t F) { return win + ((uint32_t)F & (K - 1u)) * kWinStride; };
    auto load_row = [&](int32_t F) {
        if (!row_ok(F)) return;
        uint8_t *dst = win_row(F);
        const uint8_t *src = row_ptr(F);
        if (wide) reinterpret_cast<FramerRowVec *>(dst)[lane] = reinterpret_cast<const FramerRowVec *>(src)[lane];
        else for (uint32_t q = lane; q < nu; q += 64u) dst[q] = src[q];
    };
    auto store_row = [&](int32_t F) {
        if (!row_ok(F)) return;
        const uint8_t *src = win_row(F);
        uint8_t *dst = row_ptr(F);
        if (wide) reinterpret_cast<FramerRowVec *>(dst)[lane] = reinterpret_cast<const FramerRowVec *>(src)[lane];
        else for (uint32_t q = lane; q < nu; q += 64u) dst[q] = src[q];
    };
    // the window starts at the first frame a unit of the tile can still fill
    int32_t wbase = mn + 1 > a.frames_written ? mn + 1 : a.frames_written;
    for (uint32_t r = 0; r < K; ++r) load_row(wbase + (int32_t)r);
    bool above_dirty = false;             // a single-byte store went into a row ABOVE the window
    int32_t ahead_row = -0x7fffffff - 1;  // the row held in `ahead` (fetched before it is needed)
    FramerRowVec ahead{};

    for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
        uint64_t my_lo = 0, my_hi = 0;  // lane l: this tile's slice of frame t0 + l
        if (t0 + lane < T) {
            const uint64_t *row = tile_off + (size_t)(t0 + lane) * (ntiles + 1u) + tile;
            my_lo = row[0];
            my_hi = row[1];
            if (my_hi < my_lo) {
                flags |= kFramerStatusMalformed;
                my_hi = my_lo;
            }
        }
        const uint32_t nf = T - t0 < 64u ? T - t0 : 64u;
        // the batch after `c`: the next 64 events of the frame, or the first ones of the next frame (f == nf: none)
        auto advance = [&](const FramerCursor &c) {
            FramerCursor n = c;
            if (c.f >= nf) return n;
            n.base = c.base + 64u;
            if (n.base >= c.hi) {
                n.f = c.f + 1u;
                if (n.f < nf) {
                    n.lo = readlane64(my_lo, n.f);
                    n.hi = readlane64(my_hi, n.f);
                    n.base = n.lo;
                }
            }
            return n;
        };
        auto fetch = [&](const FramerCursor &c) {
            return c.f < nf ? framer_fetch(ev, c.base + lane, c.hi) : FramerEv{};
        };
        // the next batch is in flight while this one is applied
        FramerCursor c0{0u, readlane64(my_lo, 0), readlane64(my_hi, 0), 0ull};
        c0.base = c0.lo;
        FramerEv cur = fetch(c0);
        uint32_t carry_u = 0u;  // the unit of the event just below this batch (same frame)
        while (c0.f < nf) {
            const FramerCursor c1 = advance(c0);
            const FramerEv e1 = fetch(c1);

            const uint64_t lo = c0.lo, hi = c0.hi;
            const uint64_t i = c0.base + lane;
            const uint64_t act = __builtin_amdgcn_ballot_w64(i < hi);
            uint32_t u_here = 0u;
            if (act != 0ull) {
                const bool active = i < hi;
                bool ok = false;
                const uint32_t u = unit_of(a, cur.xy, cur.cd, ok);
                u_here = u;
                bool bad = active && (!ok || u < w0 || u >= w1);
                // ascending unit order inside the slice: equal units are then adjacent lanes
                const bool first = lane == 0u;
                uint32_t pu = wave_shr1(u);
                if (first) pu = carry_u;
                const bool has_prev = active && i > lo;
                if (has_prev && pu > u) bad = true;
                if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
                    flags |= kFramerStatusMalformed;  // nothing of this batch is applied
                } else {
                    const bool is_head = active && (first || pu != u);  // (a run that began in the batch below
                                                                        // continues from the trackers it left)
                    const uint64_t heads = __builtin_amdgcn_ballot_w64(is_head);
                    // lane index of the run's first lane: the highest head bit at or below this lane
                    const uint64_t below = heads & (lane == 63u ? ~0ull : ((2ull << lane) - 1ull));
                    const uint32_t head = active ? 63u - (uint32_t)__builtin_clzll(below | 1ull) : lane;
                    const bool tail = active && (lane == 63u || ((heads | ~act) >> (lane + 1u)) & 1ull);
                    const uint32_t ul = active ? u - w0 : 0u;
                    const FramerPx p0 = px[ul];
                    const uint32_t d = (cur.cd >> 8) & 0xffu;
                    // 32-bit clocks unless something in the batch is large (wave-uniform choice)
                    const bool big = active && ((p0.ts >> 31) != 0ull || cur.t >= (1u << 23));
                    const bool small = __builtin_amdgcn_ballot_w64(big) == 0ull && a.k.ref_interval < (1u << 23);
                    const bool short_runs = __builtin_amdgcn_ballot_w64(active && lane - head > 3u) == 0ull;
                    FramerLaneOut o;
                    if (small && short_runs)
                        o = framer_batch_step_short(active, lane - head, d, cur.t, p0, a.k);
                    else if (small)
                        o = framer_batch_step<uint32_t>(active, lane, head, d, cur.t, p0, a.k);
                    else
                        o = framer_batch_step<uint64_t>(active, lane, head, d, cur.t, p0, a.k);
                    flags |= o.flags;
                    // frames (from, to]: those already handed out are skipped (:1074-1075), those past the ring fail
                    const int32_t F0 = o.from + 1 > a.frames_written ? o.from + 1 : a.frames_written;
                    int32_t F1 = o.to;
                    if (o.fills && F1 >= a.frames_written && (uint32_t)(F1 - a.frames_written) >= a.ring_frames) {
                        flags |= kFramerStatusRing;
                        F1 = a.frames_written + (int32_t)a.ring_frames - 1;
                    }
                    // Window policy.  A fill lands in the window when wbase <= F < wbase + K, else it is a single
                    // byte store into the ring (slow, but only for the stragglers: a pixel far behind, or one whose
                    // clock has run away, e.g. on D_EMPTY fillers in DeltaT streams).  When fills reach beyond the
                    // window it moves up, never down, so that the bulk of this batch's fills -- their mean end
                    // frame -- sits three quarters up the window; rows leave at the bottom and enter at the top whole.
                    const uint64_t beyond = __builtin_amdgcn_ballot_w64(o.fills && F1 >= wbase + Ki);
                    if (beyond != 0ull) {
                        const uint64_t fm = __builtin_amdgcn_ballot_w64(o.fills);
                        int32_t sum = o.fills ? F1 - wbase : 0;  // (relative: no overflow)
                        const int32_t cap = 1 << 20;
                        sum = sum > cap ? cap : sum;
                        for (int sft = 32; sft > 0; sft >>= 1) sum += __shfl_xor(sum, sft, 64);
                        const int32_t centre = wbase + sum / (int32_t)__builtin_popcountll(fm);
                        const int32_t nb = centre - (int32_t)(3u * K / 4u);
                        if (nb > wbase) {
                            for (int32_t F = wbase; F < nb && F < wbase + Ki; ++F) store_row(F);
                            int32_t F = nb > wbase + Ki ? nb : wbase + Ki;  // first row that enters
                            if (above_dirty) {
                                // single-byte stores of earlier batches into rows that enter now must have landed, and
                                // a row fetched ahead of them is stale (same wave, same L1: ordering is all it takes)
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                                above_dirty = false;
                                ahead_row = -0x7fffffff - 1;
                            }
                            if (wide && F == ahead_row) {  // fetched when the window moved last time: no wait now
                                reinterpret_cast<FramerRowVec *>(win_row(F))[lane] = ahead;
                                ++F;
                            }
                            for (; F < nb + Ki; ++F) load_row(F);
                            wbase = nb;
                            // the row that enters next, if the window keeps moving a frame at a time
                            ahead_row = wbase + Ki;
                            if (wide && row_ok(ahead_row)) ahead = reinterpret_cast<const FramerRowVec *>(row_ptr(ahead_row))[lane];
                            else ahead_row = -0x7fffffff - 1;
                        }
                    }
                    // the fills.  Inside the window: a byte per frame at the unit's column, at most K of them; the
                    // loop runs as long as the longest fill of the batch (a pixel that wakes up covers many frames)
                    bool outside = false;
                    if (o.fills) {
                        const int32_t G0 = F0 > wbase ? F0 : wbase, G1 = F1 < wbase + Ki - 1 ? F1 : wbase + Ki - 1;
                        for (int32_t F = G0; F <= G1; ++F) win[((uint32_t)F & (K - 1u)) * kWinStride + ul] = (uint8_t)o.value;
                        outside = F1 >= wbase + Ki;
                    }
                    // the stragglers: frames below the window (the pixel lagged further than the window is deep) or
                    // above it (a clock that ran away) go into the ring as single bytes -- one straggler at a time, a
                    // frame per lane
                    uint64_t strag = __builtin_amdgcn_ballot_w64(o.fills && (F0 < wbase || F1 >= wbase + Ki));
                    while (strag != 0ull) {
                        const uint32_t src = (uint32_t)__builtin_ctzll(strag);
                        strag &= strag - 1ull;
                        const int32_t bF0 = (int32_t)__builtin_amdgcn_readlane((uint32_t)F0, src);
                        const int32_t bF1 = (int32_t)__builtin_amdgcn_readlane((uint32_t)F1, src);
                        const uint32_t bu = __builtin_amdgcn_readlane(u, src), bv = __builtin_amdgcn_readlane(o.value, src);
                        for (int32_t F = bF0 + (int32_t)lane; F <= bF1; F += 64)
                            if (F < wbase || F >= wbase + Ki) a.ring[(size_t)ring_slot(F) * a.n_units + bu] = (uint8_t)bv;
                    }
                    above_dirty = above_dirty || __builtin_amdgcn_ballot_w64(outside) != 0ull;
                    if (tail) {
                        FramerPx pn;
                        pn.ts = o.ts_post;
                        pn.lastf = o.lastf_post;
                        pn.lasti = o.value;
                        px[ul] = pn;
                    }
                }
            }
            if (c1.f == c0.f)  // the next batch continues this frame: its lane 0 follows this batch's lane 63
                carry_u = __builtin_amdgcn_readlane(u_here, 63);
            cur = e1;
            c0 = c1;
        }
    }
    for (uint32_t r = 0; r < K; ++r) store_row(wbase + (int32_t)r);
    for (uint32_t q = lane; q < nu; q += 64u) a.px[w0 + q] = px[q];
    if (flags) atomicOr(a.status, flags);
}

// min / max of last_filled over the band: frames [frames_written, min + 1) are complete
__global__ __launch_bounds__(256) void adder_framer_minmax_kernel(const FramerPx *__restrict__ px, uint32_t n,
                                                                  int32_t *out /* [2] = {min, max} */) {
    int32_t mn = 0x7fffffff, mx = -0x7fffffff - 1;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const int32_t v = px[i].lastf;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    // one pair of atomics per workgroup (same-address atomics serialise in L2)
    __shared__ int32_t s_mn[4], s_mx[4];
    if ((threadIdx.x & 63u) == 0u) {
        s_mn[threadIdx.x >> 6] = mn;
        s_mx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        for (int w = 1; w < 4; ++w) {
            mn = s_mn[w] < mn ? s_mn[w] : mn;
            mx = s_mx[w] > mx ? s_mx[w] : mx;
        }
        atomicMin(&out[0], mn);
        atomicMax(&out[1], mx);
    }
}

// frames [f0, f0 + nf) of the ring -> out[nf][n_units]; masked: a pixel without a value in the
// frame (last_filled < f) reads as 0 (write_frame_bytes' `None => T::default()`, driver.rs:946-950)
__global__ __launch_bounds__(256) void adder_framer_pop_kernel(const uint8_t *__restrict__ ring,
                                                               const FramerPx *__restrict__ px, uint32_t n_units,
                                                               uint32_t ring_frames, int32_t f0, uint32_t masked,
                                                               uint8_t *__restrict__ out, uint32_t value_type) {
    const uint32_t f = (uint32_t)f0 + blockIdx.y;
    if (value_type != 0u) {  // u16 / u32 elements: big-endian bytes (bincode fixint BE, driver.rs:279,395-398)
        const size_t base = (size_t)(f % ring_frames) * n_units;
        uint8_t *dst = out + ((size_t)blockIdx.y * n_units << value_type);
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_units; i += gridDim.x * 256u) {
            const bool has = !masked || px[i].lastf >= (int32_t)f;
            if (value_type == 1u) {
                const uint32_t v = has ? reinterpret_cast<const uint16_t *>(ring)[base + i] : 0u;
                dst[2u * i] = (uint8_t)(v >> 8);
                dst[2u * i + 1u] = (uint8_t)v;
            } else {
                const uint32_t v = has ? reinterpret_cast<const uint32_t *>(ring)[base + i] : 0u;
                reinterpret_cast<uint32_t *>(dst)[i] = __builtin_bswap32(v);
            }
        }
        return;
    }
    const uint8_t *src = ring + (size_t)(f % ring_frames) * n_units;
    uint8_t *dst = out + (size_t)blockIdx.y * n_units;
    for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u; i < n_units; i += gridDim.x * 1024u) {
        if (i + 4u <= n_units && !masked && ((n_units & 3u) == 0u)) {
            *reinterpret_cast<uint32_t *>(dst + i) = *reinterpret_cast<const uint32_t *>(src + i);
        } else {
            for (uint32_t j = i; j < n_units && j < i + 4u; ++j)
                dst[j] = (!masked || px[j].lastf >= (int32_t)f) ? src[j] : (uint8_t)0;
        }
    }
}

// flush_frame_buffer (driver.rs:632-677): every pixel without a value in frame f0 gets its last
// intensity there and its last_filled advances by ONE (as the reference does)
__global__ __launch_bounds__(256) void adder_framer_flush_kernel(uint8_t *ring, FramerPx *px, uint32_t n_units,
                                                                 uint32_t ring_frames, int32_t f0, uint32_t value_type) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_units) return;
    const int32_t lf = px[i].lastf;
    if (lf < f0) {
        framer_ring_store(ring, (size_t)((uint32_t)f0 % ring_frames) * n_units + i, value_type, px[i].lasti);
        px[i].lastf = lf + 1;
    }
}

__global__ void adder_framer_init_kernel(FramerPx *px, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    FramerPx p;
    p.ts = 0;
    p.lastf = -1;
    p.lasti = 0;
    px[i] = p;
}

}  // namespace adder

using namespace adder;

extern "C" hipError_t adder_framer_launch_segment(const void *ev, uint64_t e0, uint64_t e1, const FramerArgs *args,
                                                  hipStream_t s) {
    if (e1 <= e0) return hipSuccess;
    const FramerArgs a = *args;
    const uint64_t grid = (e1 - e0 + 255u) / 256u;
    hipLaunchKernelGGL(adder_framer_segment_kernel, dim3((uint32_t)grid), dim3(256), 0, s,
                       static_cast<const uint32_t *>(ev), e0, e1, a);
    return hipGetLastError();
}
extern "C" uint32_t adder_framer_num_tiles(uint32_t n_units) { return (n_units + kTileUnits - 1u) / kTileUnits; }

// tile_off: [T][ntiles + 1] uint64; window_rows: a power of two <= 64
extern "C" hipError_t adder_framer_launch_tiles(const void *ev, const uint64_t *d_seg_offsets, uint32_t T,
                                                uint64_t *tile_off, uint32_t window_rows, const FramerArgs *args,
                                                hipStream_t s) {
    if (!T) return hipSuccess;
    const FramerArgs a = *args;
    const uint32_t ntiles = adder_framer_num_tiles(a.n_units);
    const uint64_t n_off = (uint64_t)T * (ntiles + 1u);
    hipLaunchKernelGGL(adder_framer_slices_kernel, dim3((uint32_t)((n_off + 255u) / 256u)), dim3(256), 0, s,
                       static_cast<const uint32_t *>(ev), d_seg_offsets, T, ntiles, tile_off, a);
    const size_t lds = (size_t)kTileUnits * sizeof(FramerPx) + (size_t)window_rows * kWinStride;
    hipLaunchKernelGGL(adder_framer_tiles_kernel, dim3(ntiles), dim3(64), lds, s, static_cast<const uint32_t *>(ev), tile_off,
                       T, ntiles, window_rows, a);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_minmax(const FramerPx *px, uint32_t n, int32_t *out, hipStream_t s) {
    const uint32_t grid = (n + 256u * 16u - 1u) / (256u * 16u);
    hipLaunchKernelGGL(adder_framer_minmax_kernel, dim3(grid < 1024u ? (grid ? grid : 1u) : 1024u), dim3(256), 0, s, px,
                       n, out);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_pop(const uint8_t *ring, const FramerPx *px, uint32_t n_units,
                                              uint32_t ring_frames, int32_t f0, uint32_t nf, uint32_t masked,
                                              uint8_t *out, uint32_t value_type, hipStream_t s) {
    if (!nf) return hipSuccess;
    uint32_t gx = (n_units + 1023u) / 1024u;
    gx = gx > 2048u ? 2048u : gx;
    hipLaunchKernelGGL(adder_framer_pop_kernel, dim3(gx, nf), dim3(256), 0, s, ring, px, n_units, ring_frames, f0,
                       masked, out, value_type);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_flush(uint8_t *ring, FramerPx *px, uint32_t n_units, uint32_t ring_frames,
                                                int32_t f0, uint32_t value_type, hipStream_t s) {
    hipLaunchKernelGGL(adder_framer_flush_kernel, dim3((n_units + 255u) / 256u), dim3(256), 0, s, ring, px, n_units,
                       ring_frames, f0, value_type);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_init(FramerPx *px, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(adder_framer_init_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, px, n);
    return hipGetLastError();
}
