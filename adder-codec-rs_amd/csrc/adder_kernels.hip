// adder_kernels.hip -- CDNA4 (gfx950) kernels of the framed->ADDER integration path.
//
// One launch of adder_frame_kernel does for one input frame what the reference's
// rayon loop does (adder-codec-rs/src/transcoder/source/video.rs:677-734): every
// pixel-channel runs integrate_for_px (video.rs:1318-1380) and the emitted events
// are gathered in raster order (y, x, c, per-pixel emission order).
//
// Mapping to the hardware (memory-bound, no MFMA):
//   * structure-of-arrays pixel state resident in HBM across frames (fields: see
//     adder_pixel.hpp).  A lane owns 4 consecutive pixel-channels, so every state access
//     is a 16-byte-per-lane coalesced vector load/store and the frame row is read as one
//     dword per lane.  Level-planar arena storage: plane k holds every pixel's k-th fired
//     node, so only the planes a wave actually needs are touched (the headline mode never
//     goes past plane 0).
//   * the lean step (step_fast) runs once per pixel, its <= 3 events are parked in a
//     lane-private LDS stack (conflict-free [slot][thread] layout);
//   * ordered stream compaction in the same pass: per-lane counts -> wave prefix
//     (cross-lane shuffles) -> block prefix in LDS -> tile prefix from a TWO-LEVEL scan over
//     8-byte {valid,count} descriptors (per tile, and per group of 32 tiles) that are
//     read/written with relaxed agent-scope atomics -- the data is the flag, so no fences,
//     and every tile needs exactly two dependent hops instead of a serial look-back walk;
//     then each lane copies its parked events to their final slots;
//   * a persistent grid (<= resident capacity, verified by a census launch) strides over
//     the 1024-unit tiles, so a wait can never be on a block that is not running; every
//     wait is bounded and reports ADDER_E_TIMEOUT instead of hanging.
//   * pixels whose arena is deeper than one fired level (Normal mode, delta_t_max >
//     time_spanned) are listed in a worklist with their reserved output range and stepped
//     by adder_generic_kernel right after (exec_step: the full arena walk).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adder_kernels.h"
#include "adder_pixel.hpp"

namespace adder {

constexpr uint32_t kWave = 64;
constexpr uint32_t kWavesPerBlock = kBlockThreads / kWave;
constexpr uint64_t kValid = 1ull << 32;

__device__ __forceinline__ uint64_t desc_load(uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
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
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
#pragma unroll
    for (uint32_t o = kWave / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
    return x;
}

// Waits until every lane's descriptor is valid and returns the wave-wide sum of the
// counts; lanes with use == false contribute 0 and do not load.  false on time-out.
__device__ __forceinline__ bool poll_sum(uint64_t *p, bool use, uint32_t spin_limit, uint32_t *sum) {
    uint32_t spins = 0;
    for (;;) {
        const uint64_t d = use ? desc_load(p) : kValid;
        if (__ballot((uint32_t)(d >> 32) == 0u) == 0ull) {
            *sum = wave_sum(use ? (uint32_t)d : 0u);
            return true;
        }
        if (++spins > spin_limit) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// Exclusive prefix of `tile` (events of all earlier tiles of this frame).  One full wave.
// Level 1: the tile's predecessors inside its group of kGroupTiles.  Level 2: the sums of
// all earlier groups (published by whichever block handles a group's last tile).  The
// caller keeps (known_groups, known_sum) across its tiles so later rounds only read the
// groups completed since.
__device__ __forceinline__ uint32_t resolve_prefix(const FrameArgs &a, uint32_t tile, uint32_t block_total,
                                                   uint32_t lane, uint32_t &known_groups,
                                                   uint32_t &known_sum) {
    const uint32_t g = tile / kGroupTiles;
    const uint32_t r = tile - g * kGroupTiles;
    const uint32_t gfirst = g * kGroupTiles;
    const uint32_t in_group_tiles = min(kGroupTiles, a.num_tiles - gfirst);
    bool ok = true;
    uint32_t in_group = 0;
    ok &= poll_sum(a.agg_cur + gfirst + lane, lane < r, a.spin_limit, &in_group);
    if (r == in_group_tiles - 1u && lane == 0)
        desc_store(a.gsum_cur + g, kValid | (uint64_t)(in_group + block_total));
    uint32_t gs = known_sum;
    for (uint32_t g0 = known_groups; g0 < g; g0 += kWave) {
        uint32_t part = 0;
        ok &= poll_sum(a.gsum_cur + g0 + lane, g0 + lane < g, a.spin_limit, &part);
        gs += part;
    }
    if (!ok && lane == 0) raise(a.status, kStatusTimeout);
    known_groups = g;
    known_sum = gs;
    return gs + in_group;
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

// GENERIC = false is used when no pixel can ever be deeper than one fired level
// (Collapse with delta_t_max <= time_spanned): the eligibility test, the slot reservation
// for generic pixels and the worklist are compiled out.
template <bool COLLAPSE, bool ABS_T, bool GENERIC>
__global__ __launch_bounds__(kBlockThreads) void adder_frame_kernel(FrameArgs a) {
    __shared__ uint2 s_slots[kSlotsPerLane * kBlockThreads];  // [slot][thread] {t, d | px << 8}
    __shared__ uint32_t s_wave_tot[kWavesPerBlock];
    __shared__ uint32_t s_tile_base;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wid = tid / kWave;

    if (a.census) {
        // residency census: every block of the grid must be running at the same time
        if (tid == 0) {
            __hip_atomic_fetch_add(a.census, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0;
            while (__hip_atomic_load(a.census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                if (++spins > a.spin_limit) {
                    raise(a.status, kStatusTimeout);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        return;
    }

    const StepConsts sc = a.sc;
    const uint64_t frame_base = a.frame_offsets[a.frame_idx];
    uint32_t known_groups = 0, known_sum = 0;

    for (uint32_t tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
        const uint32_t u0 = tile * kTileUnits + tid * kUnitsPerLane;

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
        uint32_t nl = 0;     // events parked by this lane
        uint32_t cnts = 0;   // per-pixel event counts, 8 bits each
        uint32_t gmask = 0;  // pixels left to the generic kernel
        uint2 *my_slots = s_slots + tid;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (u0 + j < a.n_units && !(a.ablate & 2u)) {
                if (!GENERIC || fast_eligible<COLLAPSE>(px[j], vin[j])) {
                    FastEvents fe;
                    step_fast<COLLAPSE, ABS_T>(px[j], vin[j], sc, fe);
                    const uint32_t tag = (uint32_t)j << 8;
                    if (fe.mask & 1u) {
                        my_slots[nl * kBlockThreads] = make_uint2(fe.ta, fe.da | tag);
                        ++nl;
                    }
                    if (fe.mask & 2u) {
                        my_slots[nl * kBlockThreads] = make_uint2(fe.tb, fe.db | tag);
                        ++nl;
                    }
                    if (fe.mask & 4u) {
                        my_slots[nl * kBlockThreads] = make_uint2(fe.tc, fe.dc | tag);
                        ++nl;
                    }
                    cnts |= (uint32_t)__popc(fe.mask) << (8 * j);
                } else {
                    cnts |= plan_count(px[j], vin[j], sc) << (8 * j);
                    gmask |= 1u << j;
                }
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

        // ---------------- ordered prefix: lane -> wave -> block -> tile ----------------
        const uint32_t incl = wave_inclusive_scan(lane_cnt, lane);
        if (lane == kWave - 1) s_wave_tot[wid] = incl;
        __syncthreads();
        uint32_t wave_off = 0, block_total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWavesPerBlock; ++w) {
            const uint32_t t = s_wave_tot[w];
            if (w < wid) wave_off += t;
            block_total += t;
        }
        const uint32_t lane_off = wave_off + incl - lane_cnt;

        if (wid == 0) {
            if (lane == 0) {
                desc_store(a.agg_cur + tile, kValid | block_total);
                a.agg_next[tile] = 0ull;  // ready for the next frame's launch
                if (tile % kGroupTiles == 0u) a.gsum_next[tile / kGroupTiles] = 0ull;
                if (GENERIC && tile == 0u) *a.wl_count_next = 0u;
            }
            uint32_t excl = 0;
            if (!(a.ablate & 1u)) excl = resolve_prefix(a, tile, block_total, lane, known_groups, known_sum);
            if (lane == 0) {
                s_tile_base = excl;
                if (tile == a.num_tiles - 1)
                    a.frame_offsets[a.frame_idx + 1] = frame_base + excl + block_total;
            }
        }
        __syncthreads();

        // ---------------- parked events -> their final slots ----------------
        if (lane_cnt | gmask) {
            const uint32_t rel0 = s_tile_base + lane_off;
            uint64_t pos = frame_base + rel0;
            uint32_t rel = rel0;
            EventWords *out = reinterpret_cast<EventWords *>(a.out);
            bool dropped = false;
            // coordinates of the lane's first unit; later units advance with carry
            uint32_t y = u0 / a.rowlen;
            uint32_t rem = u0 - y * a.rowlen;
            uint32_t x, c;
            if (a.channels == 1u) {
                x = rem;
                c = 0u;
            } else {
                x = rem / a.channels;
                c = rem - x * a.channels;
            }
            uint32_t si = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t cj = (cnts >> (8 * j)) & 0xffu;
                if (GENERIC && ((gmask >> j) & 1u)) {
                    const uint32_t slot = atomicAdd(a.wl_count_cur, 1u);
                    a.worklist[slot] = make_uint2(u0 + j, rel);
                } else {
                    const uint32_t xy = x | ((y + a.row_begin) << 16);
                    const uint32_t cc = a.channels == 1u ? 0xffu : c;
#pragma unroll
                    for (uint32_t k = 0; k < 3; ++k) {
                        if (k < cj) {
                            const uint2 sl = my_slots[(si + k) * kBlockThreads];
                            if (pos + k < a.out_cap) {
                                EventWords w;
                                w.xy = xy;
                                w.cd = cc | ((sl.y & 0xffu) << 8);
                                w.t = sl.x;
                                out[pos + k] = w;
                            } else {
                                dropped = true;
                            }
                        }
                    }
                    si += cj;
                }
                pos += cj;
                rel += cj;
                if (++c >= a.channels) {
                    c = 0u;
                    if (++x >= a.width) {
                        x = 0u;
                        ++y;
                    }
                }
            }
            if (dropped) raise(a.status, kStatusCapacity);
        }
        __syncthreads();  // s_slots / s_wave_tot / s_tile_base are reused by the next tile
    }
}

// The full arena walk for the pixels the frame kernel listed (deeper than one fired level).
__global__ __launch_bounds__(kBlockThreads) void adder_generic_kernel(FrameArgs a) {
    const uint32_t n = *a.wl_count_cur;
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
        em.pos = frame_base + e.y;
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

typedef void (*FrameKernelFn)(FrameArgs);
static FrameKernelFn pick_frame_kernel(const FrameArgs *a) {
    if (a->sc.collapse) {
        if (a->generic)
            return a->sc.abs_t ? adder_frame_kernel<true, true, true> : adder_frame_kernel<true, false, true>;
        return a->sc.abs_t ? adder_frame_kernel<true, true, false> : adder_frame_kernel<true, false, false>;
    }
    return a->sc.abs_t ? adder_frame_kernel<false, true, true> : adder_frame_kernel<false, false, true>;
}

extern "C" hipError_t adder_launch_frame(const FrameArgs *args, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(pick_frame_kernel(args), dim3(grid), dim3(kBlockThreads), 0, stream, *args);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_generic(const FrameArgs *args, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(adder_generic_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, *args);
    return hipGetLastError();
}

extern "C" hipError_t adder_frame_kernel_occupancy(const FrameArgs *args, int *blocks_per_cu) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, pick_frame_kernel(args), kBlockThreads, 0);
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
