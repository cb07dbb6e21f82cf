// adder_kernels.hip -- CDNA4 (gfx950) kernels of the framed->ADDER integration path.
//
// One input frame goes through a wait-free three-kernel pipeline that does what the
// reference's rayon loop does (adder-codec-rs/src/transcoder/source/video.rs:677-734):
// every pixel-channel runs integrate_for_px (video.rs:1318-1380) and the emitted events
// are gathered in raster order (y, x, c, per-pixel emission order).
//
//   K1 adder_frame_kernel   one lane = 4 consecutive pixel-channels.  Loads the header
//        word, the frame bytes and level 0 of the arena as 16-byte-per-lane vectors
//        (structure-of-arrays state resident in HBM across frames, level-planar: plane
//        k holds every pixel's k-th fired node, so the headline mode never goes past
//        plane 0), runs the lean step (step_fast) once per pixel, parks its <= 3 events
//        in a lane-private LDS stack, stores the state, then compacts the events of the
//        WAVE (ballot-free prefix over cross-lane shuffles, no barrier, no atomics) into
//        the wave's scratch segment and writes the segment's event count.
//   Ks adder_scan_kernel    exclusive prefix over the per-segment counts (one block).
//   K2 adder_expand_kernel  reads the parked events linearly and writes each 12-byte
//        event to its final slot of the ordered stream (coordinates from the unit index).
//
// No kernel waits on another workgroup, so there is no residency requirement, no spin
// loop and nothing that can hang; K2 of frame f overlaps K1 of frame f+1 on a second
// stream.  Pixels whose arena is deeper than one fired level (Normal mode, or
// delta_t_max > time_spanned) get their output range reserved by K1 (plan_count) and
// are stepped by adder_generic_kernel (exec_step: the full arena walk) after the scan.
// Memory-bound integer/f32 work: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adder_kernels.h"
#include "adder_pixel.hpp"

namespace adder {

constexpr uint32_t kWave = 64;
constexpr uint32_t kWavesPerBlock = kBlockThreads / kWave;

__device__ __forceinline__ void raise(uint32_t *status, uint32_t bit) {
    __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x, uint32_t lane) {
#pragma unroll
    for (uint32_t o = 1; o < kWave; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, kWave);
        if (lane >= o) x += y;
    }
    return x;
}

struct DeepGlobal {
    float *integ, *dt, *bdt;
    uint8_t *bd;
    size_t stride;
    size_t u;
    __device__ __forceinline__ void load(uint32_t k, Node &n) const {
        const size_t i = (size_t)k * stride + u;
        n.integ = integ[i];
        n.dt = dt[i];
        n.bdt = bdt[i];
        n.bd = bd[i];
    }
    __device__ __forceinline__ void store(uint32_t k, const Node &n) const {
        const size_t i = (size_t)k * stride + u;
        integ[i] = n.integ;
        dt[i] = n.dt;
        bdt[i] = n.bdt;
        bd[i] = (uint8_t)n.bd;
    }
};

struct __attribute__((aligned(4))) EventWords {
    uint32_t xy, cd, t;
};

struct EmitGlobal {
    EventWords *out;
    uint64_t pos, cap;
    uint32_t xy, c;
    bool dropped;
    __device__ __forceinline__ void operator()(uint32_t d, uint32_t t) {
        if (pos < cap) {
            EventWords w;
            w.xy = xy;
            w.cd = c | (d << 8);
            w.t = t;
            out[pos] = w;
        } else {
            dropped = true;
        }
        ++pos;
    }
};

// ------------------------------------------------------------------------------------------
// K1.  GENERIC = false is used when no pixel can ever be deeper than one fired level
// (Collapse with delta_t_max <= time_spanned): the eligibility test, the slot reservation
// for generic pixels and the worklist are compiled out.
// ------------------------------------------------------------------------------------------
template <bool COLLAPSE, bool ABS_T, bool GENERIC>
__global__ __launch_bounds__(kBlockThreads) void adder_frame_kernel(const BatchArgs *__restrict__ b, uint32_t f) {
    const FrameArgs a = frame_args(b, f);
    __shared__ uint2 s_slots[(kSlotsPerLane + 1) * kBlockThreads];  // [slot][thread] {t, d | px<<8 | k<<10}

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + tid / kWave;  // global wave segment
    const uint32_t u0 = blockIdx.x * kTileUnits + tid * kUnitsPerLane;
    const StepConsts sc = a.sc;

    // ---------------- loads: header, input, level 0 ----------------
    const uint4 hdr4 = *reinterpret_cast<const uint4 *>(a.hdr + u0);
    const uint32_t hdrv[4] = {hdr4.x, hdr4.y, hdr4.z, hdr4.w};
    uint32_t vin[4];
    if (u0 + kUnitsPerLane <= a.n_units) {
        uint32_t w;
        __builtin_memcpy(&w, a.frame + u0, 4);
        vin[0] = w & 0xffu;
        vin[1] = (w >> 8) & 0xffu;
        vin[2] = (w >> 16) & 0xffu;
        vin[3] = w >> 24;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) vin[j] = (u0 + j < a.n_units) ? a.frame[u0 + j] : 0u;
    }
    const bool any_m = (((hdrv[0] | hdrv[1] | hdrv[2] | hdrv[3]) >> 24) & kFlagMMask) != 0u;

    PxState px[4];
    {
        float4 li = make_float4(0.f, 0.f, 0.f, 0.f), ld = li, lb = li, lf = li;
        uint32_t lbd = 0u;
        if (any_m) {
            li = *reinterpret_cast<const float4 *>(a.lv_integ + u0);
            ld = *reinterpret_cast<const float4 *>(a.lv_dt + u0);
            lb = *reinterpret_cast<const float4 *>(a.lv_bdt + u0);
            lbd = *reinterpret_cast<const uint32_t *>(a.lv_bd + u0);
        }
        if (ABS_T) lf = *reinterpret_cast<const float4 *>(a.lastf + u0);
        const float liv[4] = {li.x, li.y, li.z, li.w}, ldv[4] = {ld.x, ld.y, ld.z, ld.w};
        const float lbv[4] = {lb.x, lb.y, lb.z, lb.w}, lfv[4] = {lf.x, lf.y, lf.z, lf.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            px[j].hdr = hdrv[j];
            px[j].n0.integ = liv[j];
            px[j].n0.dt = ldv[j];
            px[j].n0.bdt = lbv[j];
            px[j].n0.bd = (lbd >> (8 * j)) & 0xffu;
            px[j].lastf = lfv[j];
        }
    }

    // ---------------- the step; events parked in the lane's LDS stack ----------------
    // Branch-free parking: all three candidate events are written, the stack pointer only
    // advances past the valid ones (hence kSlotsPerLane + 1 rows).
    uint32_t nl = 0;     // events parked by this lane
    uint32_t cnts = 0;   // per-pixel event counts, 8 bits each
    uint32_t gmask = 0;  // pixels left to the generic kernel
    uint2 *my_slots = s_slots + tid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool active = u0 + j < a.n_units && !(a.ablate & 2u);
        if (!GENERIC || fast_eligible<COLLAPSE>(px[j], vin[j])) {
            FastEvents fe;
            PxState nx = px[j];
            step_fast<COLLAPSE, ABS_T>(nx, vin[j], sc, fe);
            if (active) px[j] = nx;
            const uint32_t mask = active ? fe.mask : 0u;
            const uint32_t tag = (uint32_t)j << 8;
            my_slots[nl * kBlockThreads] = make_uint2(fe.ta, fe.da | tag);
            nl += mask & 1u;
            my_slots[nl * kBlockThreads] = make_uint2(fe.tb, fe.db | tag | (1u << 10));
            nl += (mask >> 1) & 1u;
            // index of the event inside its pixel: after A and B if present
            const uint32_t kc = (mask & 1u) + ((mask >> 1) & 1u);
            my_slots[nl * kBlockThreads] = make_uint2(fe.tc, fe.dc | tag | (kc << 10));
            nl += mask >> 2;
            cnts |= (uint32_t)__popc(mask) << (8 * j);
        } else if (active) {
            cnts |= plan_count(px[j], vin[j], sc) << (8 * j);
            gmask |= 1u << j;
        }
    }
    const uint32_t lane_cnt = (cnts & 0xffu) + ((cnts >> 8) & 0xffu) + ((cnts >> 16) & 0xffu) + (cnts >> 24);

    // ---------------- state back to HBM (generic pixels keep their old state) ----------------
    {
        uint4 h;
        h.x = px[0].hdr;
        h.y = px[1].hdr;
        h.z = px[2].hdr;
        h.w = px[3].hdr;
        *reinterpret_cast<uint4 *>(a.hdr + u0) = h;
        if (((h.x | h.y | h.z | h.w) >> 24) & kFlagMMask) {
            *reinterpret_cast<float4 *>(a.lv_integ + u0) =
                make_float4(px[0].n0.integ, px[1].n0.integ, px[2].n0.integ, px[3].n0.integ);
            *reinterpret_cast<float4 *>(a.lv_dt + u0) =
                make_float4(px[0].n0.dt, px[1].n0.dt, px[2].n0.dt, px[3].n0.dt);
            *reinterpret_cast<float4 *>(a.lv_bdt + u0) =
                make_float4(px[0].n0.bdt, px[1].n0.bdt, px[2].n0.bdt, px[3].n0.bdt);
            *reinterpret_cast<uint32_t *>(a.lv_bd + u0) = (px[0].n0.bd & 0xffu) | ((px[1].n0.bd & 0xffu) << 8) |
                                                          ((px[2].n0.bd & 0xffu) << 16) | (px[3].n0.bd << 24);
        }
        if (ABS_T)
            *reinterpret_cast<float4 *>(a.lastf + u0) =
                make_float4(px[0].lastf, px[1].lastf, px[2].lastf, px[3].lastf);
        if (a.running) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (u0 + j < a.n_units && !((gmask >> j) & 1u) && ((px[j].hdr >> 24) & kFlagMMask))
                    a.running[u0 + j] = (uint8_t)frame_value_u8(px[j].n0.bd, f32_as_u32(px[j].n0.bdt),
                                                                (double)sc.ref_time);
        }
    }

    // ---------------- wave-level ordered compaction into the segment ----------------
    // low half: events of the lane in the final stream; high half: events it parked
    const uint32_t packed = lane_cnt | (nl << 16);
    const uint32_t incl = wave_inclusive_scan(packed, lane);
    if (lane == kWave - 1) a.wtot[gw] = incl;
    const uint32_t excl = incl - packed;
    const uint32_t lane_off = excl & 0xffffu;  // final offset of the lane inside the segment
    // exclusive prefix of the per-pixel counts, 8 bits each (sums stay below 256)
    const uint32_t pre = (cnts << 8) + (cnts << 16) + (cnts << 24);
    uint2 *dst = a.park + (size_t)gw * kParkPerWave + (excl >> 16);
    for (uint32_t i = 0; i < nl; ++i) {
        uint2 sl = my_slots[i * kBlockThreads];
        const uint32_t j = (sl.y >> 8) & 3u;
        const uint32_t off = GENERIC ? lane_off + ((pre >> (8u * j)) & 0xffu) + ((sl.y >> 10) & 3u) : lane_off + i;
        sl.y = (sl.y & 0xffu) | ((lane * kUnitsPerLane + j) << 8) | (off << 16);
        dst[i] = sl;
    }
    if (GENERIC && gmask) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((gmask >> j) & 1u) {
                const uint32_t slot = atomicAdd(a.wl_count, 1u);
                a.worklist[slot] = make_uint2(u0 + j, lane_off + ((pre >> (8 * j)) & 0xffu));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Ks: exclusive prefix of the per-segment event counts; also closes the frame's range in
// frame_offsets and clears the worklist counter for the next frame.  One block.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void adder_scan_kernel(const BatchArgs *__restrict__ b, uint32_t f) {
    const FrameArgs a = frame_args(b, f);
    __shared__ uint32_t s_part[kScanThreads / kWave];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wid = tid / kWave;
    // num_waves is a multiple of 4: each thread owns `per` consecutive uint4 groups
    const uint32_t groups = a.num_waves / 4u;
    const uint32_t per = (groups + kScanThreads - 1) / kScanThreads;
    const uint32_t g0 = tid * per;
    const uint32_t g1 = min(g0 + per, groups);
    const uint4 *src = reinterpret_cast<const uint4 *>(a.wtot);
    uint4 *dst = reinterpret_cast<uint4 *>(a.wpref);
    uint32_t sum = 0;
    for (uint32_t g = g0; g < g1; ++g) {
        const uint4 v = src[g];
        sum += (v.x & 0xffffu) + (v.y & 0xffffu) + (v.z & 0xffffu) + (v.w & 0xffffu);
    }
    const uint32_t incl = wave_inclusive_scan(sum, lane);
    if (lane == kWave - 1) s_part[wid] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kScanThreads / kWave; ++w) {
        const uint32_t t = s_part[w];
        if (w < wid) base += t;
        total += t;
    }
    uint32_t run = base + incl - sum;
    for (uint32_t g = g0; g < g1; ++g) {
        const uint4 v = src[g];
        uint4 o;
        o.x = run;
        o.y = o.x + (v.x & 0xffffu);
        o.z = o.y + (v.y & 0xffffu);
        o.w = o.z + (v.z & 0xffffu);
        run = o.w + (v.w & 0xffffu);
        dst[g] = o;
    }
    if (tid == 0) a.frame_offsets[a.frame_idx + 1] = a.frame_offsets[a.frame_idx] + total;
}

// ------------------------------------------------------------------------------------------
// K2: parked events -> final 12-byte events of the ordered stream.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void adder_expand_kernel(const BatchArgs *__restrict__ b, uint32_t f) {
    const FrameArgs a = frame_args(b, f);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
    if (gw >= a.num_waves) return;
    const uint32_t parked = a.wtot[gw] >> 16;
    if (parked == 0u) return;
    const uint64_t base = a.frame_offsets[a.frame_idx] + a.wpref[gw];
    const uint2 *src = a.park + (size_t)gw * kParkPerWave;
    EventWords *out = reinterpret_cast<EventWords *>(a.out);
    bool dropped = false;
    for (uint32_t i = lane; i < parked; i += kWave) {
        const uint2 sl = src[i];
        const uint32_t u = gw * kWaveUnits + ((sl.y >> 8) & 0xffu);
        const uint32_t y = u / a.rowlen;
        const uint32_t rem = u - y * a.rowlen;
        uint32_t x = rem, c = 0xffu;
        if (a.channels != 1u) {
            x = rem / a.channels;
            c = rem - x * a.channels;
        }
        const uint64_t pos = base + (sl.y >> 16);
        if (pos < a.out_cap) {
            EventWords w;
            w.xy = x | ((y + a.row_begin) << 16);
            w.cd = c | ((sl.y & 0xffu) << 8);
            w.t = sl.x;
            out[pos] = w;
        } else {
            dropped = true;
        }
    }
    if (dropped) raise(a.status, kStatusCapacity);
}

// ------------------------------------------------------------------------------------------
// The full arena walk for the pixels K1 listed (deeper than one fired level).  Runs after
// the scan kernel (needs wpref) and clears the worklist counter when done.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void adder_generic_kernel(const BatchArgs *__restrict__ b, uint32_t f) {
    const FrameArgs a = frame_args(b, f);
    const uint32_t n = *a.wl_count;
    const StepConsts sc = a.sc;
    const uint64_t frame_base = a.frame_offsets[a.frame_idx];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint2 e = a.worklist[i];
        const uint32_t u = e.x;
        PxState px;
        px.hdr = a.hdr[u];
        px.n0.integ = px.n0.dt = px.n0.bdt = 0.0f;
        px.n0.bd = 0u;
        if ((px.hdr >> 24) & kFlagMMask) {
            px.n0.integ = a.lv_integ[u];
            px.n0.dt = a.lv_dt[u];
            px.n0.bdt = a.lv_bdt[u];
            px.n0.bd = a.lv_bd[u];
        }
        px.lastf = sc.abs_t ? a.lastf[u] : 0.0f;
        const uint32_t v = a.frame[u];

        const uint32_t y = u / a.rowlen;
        const uint32_t rem = u - y * a.rowlen;
        const uint32_t x = rem / a.channels;
        const uint32_t c = rem - x * a.channels;
        EmitGlobal em;
        em.out = reinterpret_cast<EventWords *>(a.out);
        em.pos = frame_base + a.wpref[u / kWaveUnits] + e.y;
        em.cap = a.out_cap;
        em.dropped = false;
        em.xy = x | ((y + a.row_begin) << 16);
        em.c = a.channels == 1u ? 0xffu : c;
        DeepGlobal deep{a.lv_integ, a.lv_dt, a.lv_bdt, a.lv_bd, a.plane_stride, u};
        const bool depth_ok = exec_step(px, v, sc, deep, em);
        if (em.dropped) raise(a.status, kStatusCapacity);
        if (!depth_ok) raise(a.status, kStatusDepth);

        a.hdr[u] = px.hdr;
        if ((px.hdr >> 24) & kFlagMMask) {
            a.lv_integ[u] = px.n0.integ;
            a.lv_dt[u] = px.n0.dt;
            a.lv_bdt[u] = px.n0.bdt;
            a.lv_bd[u] = (uint8_t)px.n0.bd;
            if (a.running)
                a.running[u] = (uint8_t)frame_value_u8(px.n0.bd, f32_as_u32(px.n0.bdt), (double)sc.ref_time);
        }
        if (sc.abs_t) a.lastf[u] = px.lastf;
    }
}

__global__ void adder_clear_u32_kernel(uint32_t *p) { *p = 0u; }

// update_crf / update_quality_manual per-pixel reset (video.rs:1247-1250,1283-1286)
__global__ void adder_reset_c_thresh_kernel(uint32_t *hdr, size_t n, uint32_t baseline) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hdr[i] = (hdr[i] & 0xff0000ffu) | (baseline << 8);
}

__global__ void adder_fill_u32_kernel(uint32_t *p, size_t n, uint32_t v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// offsets[c] = first event with y >= row_begin + c*chunk_rows (events are y-sorted)
__global__ void adder_chunk_offsets_kernel(const AdderEventPod *ev, uint32_t n, uint32_t row_begin,
                                           uint32_t chunk_rows, uint32_t num_chunks, uint32_t *offsets) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > num_chunks) return;
    if (c == num_chunks) {
        offsets[c] = n;
        return;
    }
    const uint32_t y0 = row_begin + c * chunk_rows;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ev[mid].y < y0)
            lo = mid + 1;
        else
            hi = mid;
    }
    offsets[c] = lo;
}

// ---- deterministic synthetic content ----
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void adder_synth_kernel(uint8_t *dst, int content, uint64_t seed, uint32_t W, uint32_t H,
                                   uint32_t C, uint32_t y0, uint32_t rows, uint32_t k0, uint64_t total) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t per_frame = (uint64_t)rows * W * C;
    const uint64_t kk = i / per_frame;
    uint64_t r = i - kk * per_frame;
    const uint64_t yy = r / ((uint64_t)W * C);
    r -= yy * (uint64_t)W * C;
    const uint64_t x = r / C;
    const uint64_t c = r - x * C;
    const uint64_t k = k0 + kk, y = y0 + yy;
    uint32_t v;
    if (content == 0) {
        v = (uint32_t)(splitmix64(seed ^ ((0ull << 42) ^ (y << 28) ^ (x << 8) ^ c)) & 255u);
    } else if (content == 1) {
        v = (uint32_t)(splitmix64(seed ^ ((k << 42) ^ (y << 28) ^ (x << 8) ^ c)) & 255u);
    } else {
        const uint32_t bg = (uint32_t)((x * 255u / W + y * 127u / H) & 255u);
        v = bg;
        const int64_t bx = (((int64_t)x - 4 * (int64_t)k) % (int64_t)W + W) % W;
        const int64_t by = (((int64_t)y - 2 * (int64_t)k) % (int64_t)H + H) % H;
        if (bx < (int64_t)(W / 8) && by < (int64_t)(H / 8)) v = 255u - bg;
        const uint64_t h = splitmix64(seed ^ ((k << 42) ^ (y << 28) ^ (x << 8) ^ c));
        if (h % 8ull == 0ull) {
            int vv = (int)v + (int)((h >> 8) % 3ull) - 1;
            v = (uint32_t)(vv < 0 ? 0 : (vv > 255 ? 255 : vv));
        }
    }
    dst[i] = (uint8_t)v;
}

}  // namespace adder

// ------------------------- launch wrappers (called from adder_hip_api.cpp) -------------------------
using namespace adder;

typedef void (*FrameKernelFn)(const BatchArgs *, uint32_t);
static FrameKernelFn pick_frame_kernel(uint32_t variant) {
    const bool collapse = variant & 1u, abs_t = variant & 2u, generic = variant & 4u;
    if (collapse) {
        if (generic) return abs_t ? adder_frame_kernel<true, true, true> : adder_frame_kernel<true, false, true>;
        return abs_t ? adder_frame_kernel<true, true, false> : adder_frame_kernel<true, false, false>;
    }
    return abs_t ? adder_frame_kernel<false, true, true> : adder_frame_kernel<false, false, true>;
}

extern "C" hipError_t adder_launch_frame(const BatchArgs *b, uint32_t f, uint32_t variant, uint32_t num_waves,
                                         hipStream_t stream) {
    const uint32_t grid = (num_waves + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(pick_frame_kernel(variant), dim3(grid), dim3(kBlockThreads), 0, stream, b, f);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_scan(const BatchArgs *b, uint32_t f, hipStream_t stream) {
    hipLaunchKernelGGL(adder_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, b, f);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_expand(const BatchArgs *b, uint32_t f, uint32_t num_waves, hipStream_t stream) {
    const uint32_t grid = (num_waves + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(adder_expand_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, b, f);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_generic(const BatchArgs *b, uint32_t f, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(adder_generic_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, b, f);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_clear_u32(uint32_t *p, hipStream_t stream) {
    hipLaunchKernelGGL(adder_clear_u32_kernel, dim3(1), dim3(1), 0, stream, p);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_reset_c_thresh(uint32_t *hdr, size_t n, uint32_t baseline, hipStream_t stream) {
    const uint32_t bs = 256;
    hipLaunchKernelGGL(adder_reset_c_thresh_kernel, dim3((uint32_t)((n + bs - 1) / bs)), dim3(bs), 0, stream,
                       hdr, n, baseline);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_fill_u32(uint32_t *p, size_t n, uint32_t v, hipStream_t stream) {
    const uint32_t bs = 256;
    hipLaunchKernelGGL(adder_fill_u32_kernel, dim3((uint32_t)((n + bs - 1) / bs)), dim3(bs), 0, stream, p, n, v);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_chunk_offsets(const AdderEventPod *ev, uint32_t n, uint32_t row_begin,
                                                 uint32_t chunk_rows, uint32_t num_chunks, uint32_t *offsets,
                                                 hipStream_t stream) {
    const uint32_t bs = 256;
    hipLaunchKernelGGL(adder_chunk_offsets_kernel, dim3((num_chunks + 1 + bs - 1) / bs), dim3(bs), 0, stream, ev,
                       n, row_begin, chunk_rows, num_chunks, offsets);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_synth(uint8_t *dst, int content, uint64_t seed, uint32_t W, uint32_t H,
                                         uint32_t C, uint32_t y0, uint32_t rows, uint32_t k0, uint32_t nframes,
                                         hipStream_t stream) {
    const uint64_t total = (uint64_t)nframes * rows * W * C;
    if (total == 0) return hipSuccess;
    const uint32_t bs = 256;
    const uint64_t blocks = (total + bs - 1) / bs;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(adder_synth_kernel, dim3((uint32_t)blocks), dim3(bs), 0, stream, dst, content, seed, W, H,
                       C, y0, rows, k0, total);
    return hipGetLastError();
}
