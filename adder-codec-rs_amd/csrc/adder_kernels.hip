// adder_kernels.hip -- CDNA4 (gfx950) kernels of the framed->ADDER integration path.
//
// One input frame goes through a wait-free three-kernel pipeline that does what the
// reference's rayon loop does (adder-codec-rs/src/transcoder/source/video.rs:677-734):
// every pixel-channel runs integrate_for_px (video.rs:1318-1380) and the emitted events
// are gathered in raster order (y, x, c, per-pixel emission order).
//
//   K1 adder_frame_kernel   one lane = 2 consecutive pixel-channels, one wave = one
//        128-unit segment.  Loads the header word, the frame bytes and level 0 of the arena
//        as 8-byte-per-lane vectors (structure-of-arrays state resident in HBM across
//        frames, level-planar: plane k holds every pixel's k-th fired node, so the headline
//        mode never goes past plane 0) and steps up to 16 consecutive frames with the state
//        in registers.  Per frame: the lean step (step_fast) once per pixel, its <= 3 events
//        held in registers, a DPP prefix scan over the WAVE (no barrier, no atomics, no
//        LDS), the events written compacted into the wave's scratch segment, the segment's
//        event count to wtot.
//   Ks adder_scan_kernel    exclusive prefix over the per-segment counts (one block per
//        frame) + adder_offsets_kernel (the frame_offsets chain); run once per CHUNK of frames.
//   K2 expand_block         reads the parked events linearly and writes each 12-byte event
//        to its final slot of the ordered stream (coordinates from the unit index).  Runs
//        as extra workgroups inside K1's grid (the previous chunk's frames: memory-bound
//        work sharing the SIMDs with K1's VALU-bound step) and as adder_expand_kernel for
//        the last chunk of a batch.
//
// No kernel waits on another workgroup, so there is no residency requirement, no spin
// loop and nothing that can hang.  Pixels whose arena is deeper than one fired level
// (Normal mode, or delta_t_max > time_spanned) take the full arena walk (exec_step)
// inside the GENERIC instantiations of K1.
// HBM/VALU-bound integer/f32 work: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adder_kernels.h"
#include "adder_pixel.hpp"

namespace adder {

constexpr uint32_t kWave = 64;
// compact parked records (non-generic DeltaT variants)
constexpr uint32_t kParkWideT = 0x1ffffu;                 // t field marker: the value is in the overflow list
constexpr uint32_t kParkOverflowBase = kParkPerWave;      // dword index of the overflow list in a segment's scratch
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

// exec_step's events of one generic unit, parked with their final in-segment offset
struct EmitPark {
    uint2 *dst;     // next free parked slot of this lane
    uint32_t tag;   // unit_in_wave << 8
    uint32_t off;   // final offset of the unit's next event inside the segment
    __device__ __forceinline__ void operator()(uint32_t d, uint32_t t) {
        *dst++ = make_uint2(t, d | tag | (off << 16));
        ++off;
    }
};

// ------------------------------------------------------------------------------------------
// K1 building blocks.  GENERIC = false is used when no pixel can ever be deeper than one fired
// level (Collapse with delta_t_max <= time_spanned): the eligibility test and the full arena
// walk are compiled out.
// ------------------------------------------------------------------------------------------
// kUnitsPerLane consecutive values as one vector access
template <class T, int N>
struct VecOf;
template <> struct VecOf<uint32_t, 4> { using type = uint4; };
template <> struct VecOf<uint32_t, 2> { using type = uint2; };
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<float, 2> { using type = float2; };
template <> struct VecOf<uint8_t, 4> { using type = uint32_t; };
template <> struct VecOf<uint8_t, 2> { using type = uint16_t; };

// Global-address-space accesses as (wave-uniform base, 32-bit byte offset of the lane): the
// pointers of the argument block are generic in the IR, which would make every access a FLAT
// instruction with a 64-bit VALU address; with these the base stays in SGPRs and the lane
// supplies one 32-bit offset (global_load/store ... saddr).  Offsets stay below 4 GiB: one state
// plane holds n_pad * 4 bytes (adder_hip_create bounds n_pad), a parked segment a few KiB.
#define ADDER_GLOBAL __attribute__((address_space(1)))
template <int BYTES> struct RawOf;
template <> struct RawOf<1> { using type = uint8_t; };
template <> struct RawOf<2> { using type = uint16_t; };
template <> struct RawOf<4> { using type = uint32_t; };
template <> struct RawOf<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct RawOf<12> { typedef uint32_t type __attribute__((ext_vector_type(3))); };
template <> struct RawOf<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <class V>
__device__ __forceinline__ V gload(const void *base, uint32_t byte_off) {
    using R = typename RawOf<sizeof(V)>::type;
    const R r = *reinterpret_cast<const ADDER_GLOBAL R *>((const ADDER_GLOBAL char *)base + byte_off);
    V v;
    __builtin_memcpy(&v, &r, sizeof(V));
    return v;
}
template <class V>
__device__ __forceinline__ void gstore(void *base, uint32_t byte_off, V v) {
    using R = typename RawOf<sizeof(V)>::type;
    R r;
    __builtin_memcpy(&r, &v, sizeof(V));
    *reinterpret_cast<ADDER_GLOBAL R *>((ADDER_GLOBAL char *)base + byte_off) = r;
}
// a pointer that is the same in every lane, forced into SGPRs
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    const uint64_t x = (uint64_t)p;
    // (the builtin returns int: without the casts the low half would be sign-extended)
    return (T *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x >> 32)) << 32) |
                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)x));
}

template <class T>
__device__ __forceinline__ void load_vec(const T *plane, uint32_t u0, T (&v)[kUnitsPerLane]) {
    using V = typename VecOf<T, kUnitsPerLane>::type;
    const V x = gload<V>(plane, u0 * (uint32_t)sizeof(T));
    __builtin_memcpy(v, &x, sizeof(V));
}
template <class T>
__device__ __forceinline__ void store_vec(T *plane, uint32_t u0, const T (&v)[kUnitsPerLane]) {
    using V = typename VecOf<T, kUnitsPerLane>::type;
    V x;
    __builtin_memcpy(&x, v, sizeof(V));
    gstore<V>(plane, u0 * (uint32_t)sizeof(T), x);
}

// the lane's kUnitsPerLane input bytes of one frame, packed little-endian
__device__ __forceinline__ uint32_t load_input(const uint8_t *frame, uint32_t u0, uint32_t n_units) {
    uint32_t w = 0u;
    if (u0 + kUnitsPerLane <= n_units) {
        w = gload<typename VecOf<uint8_t, kUnitsPerLane>::type>(frame, u0);
    } else {
#pragma unroll
        for (uint32_t j = 0; j < kUnitsPerLane; ++j)
            if (u0 + j < n_units) w |= (uint32_t)gload<uint8_t>(frame, u0 + j) << (8 * j);
    }
    return w;
}

// inclusive prefix sum across the wave with DPP row shifts / broadcasts (no LDS traffic)
__device__ __forceinline__ uint32_t wave_inclusive_scan_dpp(uint32_t x) {
    // row_shr:1,2,4,8 within rows of 16, then row_bcast:15 and row_bcast:31
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2,3
    return x;
}

// One segment (64 lanes x kUnitsPerLane units) through `nb` consecutive frames starting at
// a.frame_idx.  The pixel state lives in registers for the whole run (temporal blocking): it
// is read from HBM once and written back once; per frame only the input bytes are loaded
// (one frame ahead) and the segment's events are compacted into that frame's scratch slot.
// nb > 1 is only used when no pixel can need the generic kernel.
// A segment's state as loaded (one memory round trip, nothing consumed yet)
struct RawSegment {
    uint32_t hdrv[kUnitsPerLane];
    float liv[kUnitsPerLane], ldv[kUnitsPerLane], lbv[kUnitsPerLane], lfv[kUnitsPerLane];
    uint8_t bdv[kUnitsPerLane];
    uint32_t vin_w;
};

// SPECULATE: level 0 is fetched together with the header word instead of after it (the
// depth-1 kernels: one round trip instead of two; nearly every unit has a fired level)
template <bool ABS_T, bool SPECULATE>
__device__ __forceinline__ void load_raw(const FrameArgs &a, uint32_t u0, bool full, RawSegment &r) {
    constexpr uint32_t N = kUnitsPerLane;
    load_vec(a.hdr, u0, r.hdrv);
    r.vin_w = load_input(a.frame, u0, full ? 0xffffffffu : a.n_units);
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        r.liv[j] = r.ldv[j] = r.lbv[j] = r.lfv[j] = 0.0f;
        r.bdv[j] = 0;
    }
    bool any = true;
    if (!SPECULATE) {
        uint32_t hor = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) hor |= r.hdrv[j];
        any = (hor >> 24) & kFlagMMask;
    }
    if (any) {
        load_vec(a.lv_integ, u0, r.liv);
        load_vec(a.lv_dt, u0, r.ldv);
        load_vec(a.lv_bdt, u0, r.lbv);
        load_vec(a.lv_bd, u0, r.bdv);
    }
    if (ABS_T) load_vec(a.lastf, u0, r.lfv);
}

template <bool COLLAPSE, bool ABS_T, bool GENERIC>
__device__ __forceinline__ void run_segment(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb,
                                            uint32_t u0, uint32_t gw, uint32_t lane, const RawSegment &raw) {
    constexpr uint32_t N = kUnitsPerLane;
    constexpr bool PARK4 = !GENERIC && !ABS_T;  // compact parked records (see below and expand_block)
    // whole wave inside the band: the common case takes the unguarded vector input load
    const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
    FastPx px[N];
    uint32_t vin_w = raw.vin_w;
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        PxState st;
        st.hdr = raw.hdrv[j];
        st.n0.integ = raw.liv[j];
        st.n0.dt = raw.ldv[j];
        st.n0.bdt = raw.lbv[j];
        st.n0.bd = raw.bdv[j];
        st.lastf = raw.lfv[j];
        px[j] = unpack_px(st);
        if (GENERIC) px[j].has0 = (raw.hdrv[j] >> 24) & kFlagMMask;  // keep the full m for the generic test
    }
    StepConsts sc = a.sc;
    uint32_t gmask = 0;  // pixels left to the generic kernel (nb == 1 only)
    // running_t of the launch's frames (nb <= kMaxFramesPerLaunch <= 64): one load, then a lane read per frame
    const uint32_t rt_vec = (lane < kMaxFramesPerLaunch && lane < nb) ? __float_as_uint(b->running_t[a.frame_idx + lane]) : 0u;

    // wave-uniform bases of the per-frame accesses, in SGPRs
    const uint32_t sgw = __builtin_amdgcn_readfirstlane(gw);
    const uint8_t *const frames_u = uniform_ptr(b->frames);
    uint2 *const park_ring_u = uniform_ptr(b->park_ring);
    uint32_t *const wtot_ring_u = uniform_ptr(b->wtot_ring);
    const uint32_t park_stride_u = __builtin_amdgcn_readfirstlane(b->park_stride);
    const uint32_t slots_u = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t num_waves_u = __builtin_amdgcn_readfirstlane(a.num_waves);
    const uint32_t n_units_u = __builtin_amdgcn_readfirstlane(a.n_units);
    uint32_t slot = __builtin_amdgcn_readfirstlane(a.frame_idx % slots_u);

    for (uint32_t i = 0; i < nb; ++i, slot = (slot + 1u == slots_u) ? 0u : slot + 1u) {
        const uint32_t f = a.frame_idx + i;
        uint32_t next_w = 0u;
        if (i + 1 < nb)
            next_w = load_input(frames_u + (size_t)(f + 1) * n_units_u, u0, full ? 0xffffffffu : n_units_u);
        sc.running_t = __uint_as_float(__builtin_amdgcn_readlane(rt_vec, i));
        sc.running_t_u32 = __builtin_amdgcn_readfirstlane(f32_as_u32(sc.running_t));

        // ---------------- the step ----------------
        uint32_t nl = 0;    // events parked by this lane's fast units
        uint32_t ngen = 0;  // events its generic units will park
        uint32_t cnts = 0;  // per-pixel event counts, 8 bits each
        gmask = 0;
        FastEvents fe[N];   // <= 3 events per fast unit, kept in registers until the scan is done
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) fe[j].mask = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const uint32_t v = (vin_w >> (8 * j)) & 0xffu;
            // units past the band's end are padding: their state may be stepped freely, only
            // their events must be suppressed
            const bool active = full || u0 + j < a.n_units;
            bool fast = true;
            PxState st;
            if (GENERIC) {
                st.hdr = (pack_hdr(px[j]) & 0x00ffffffu) | ((px[j].has0 | (px[j].popped << 5)) << 24);
                st.n0 = px[j].n0;
                st.lastf = px[j].lastf;
                fast = fast_eligible<COLLAPSE>(st, v);
            }
            if (fast) {
                step_fast<COLLAPSE, ABS_T>(px[j], v, sc, fe[j]);
                fe[j].mask = active ? fe[j].mask : 0u;
                const uint32_t c = (uint32_t)__popc(fe[j].mask);
                cnts |= c << (8 * j);
                nl += c;
            } else if (active) {
                const uint32_t planned = plan_count(st, v, sc);
                cnts |= planned << (8 * j);
                gmask |= 1u << j;
                ngen += planned;
            }
        }
        uint32_t lane_cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) lane_cnt += (cnts >> (8 * j)) & 0xffu;

        // ---------------- wave-level ordered compaction into the frame's segment ----------------
        // low half: events of the lane in the final stream; high half: events it parks (the
        // fast ones held in registers, then those of its generic units)
        const uint32_t packed = lane_cnt | ((nl + ngen) << 16);
        const uint32_t incl = wave_inclusive_scan_dpp(packed);
        const size_t seg_idx = (size_t)slot * num_waves_u + sgw;  // uniform
        if (lane == kWave - 1) gstore<uint32_t>(wtot_ring_u + seg_idx, 0u, incl);
        const uint32_t excl = incl - packed;
        const uint32_t lane_off = excl & 0xffffu;  // final offset of the lane inside the segment
        // exclusive prefix of the per-pixel counts, 8 bits each (sums stay below 256)
        const uint32_t pre = (cnts << 8) + (cnts << 16) + (cnts << 24);
        uint2 *const seg = park_ring_u + seg_idx * park_stride_u;  // uniform
        if (PARK4) {
            // compact 4-byte records {t (17 bits) | d << 17 | unit << 25}; the parked position IS the
            // final in-segment offset here (every event of these variants is a fast one).  d = 255
            // (D_EMPTY) carries no t: it is the frame's running_t, which the expansion knows.  A t that
            // does not fit (>= 2^17 - 1 ticks: a run of more than 500 frames) is written as the marker
            // kParkWideT and its value goes to the segment's overflow list (second half of the
            // segment's scratch), at the rank of the record among the segment's marked records.
            const uint32_t poff4 = (excl >> 16) * 4u;
            uint32_t e = 0, wide = 0;  // wide: bit per event slot (3 per unit) whose t needs the overflow list
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) {
                const uint32_t m = fe[j].mask;
                const uint32_t tag = (lane * N + j) << 25;
                const uint32_t ts[3] = {fe[j].ta, fe[j].tb, fe[j].tc};
                const uint32_t ds[3] = {fe[j].da, fe[j].db, fe[j].dc};
#pragma unroll
                for (uint32_t q = 0; q < 3u; ++q) {
                    const bool on = (m >> q) & 1u;
                    const bool w = on && ds[q] != kDEmpty && ts[q] >= kParkWideT;
                    const uint32_t tt = ds[q] == kDEmpty ? 0u : (w ? kParkWideT : ts[q]);
                    if (on) gstore<uint32_t>(seg, poff4 + 4u * e, tt | (ds[q] << 17) | tag);
                    wide |= w ? 1u << (3u * j + q) : 0u;
                    e += on ? 1u : 0u;
                }
            }
            if (__builtin_amdgcn_ballot_w64(wide != 0u) != 0ull) {  // rare
                const uint32_t nw = (uint32_t)__popc(wide);
                uint32_t rank = wave_inclusive_scan_dpp(nw) - nw;  // marked records of lower lanes
#pragma unroll
                for (uint32_t j = 0; j < N; ++j) {
                    const uint32_t ts[3] = {fe[j].ta, fe[j].tb, fe[j].tc};
#pragma unroll
                    for (uint32_t q = 0; q < 3u; ++q)
                        if ((wide >> (3u * j + q)) & 1u) {
                            gstore<uint32_t>(seg, (kParkOverflowBase + rank) * 4u, ts[q]);
                            rank += 1u;
                        }
                }
            }
        } else {
        const uint32_t poff = (excl >> 16) * (uint32_t)sizeof(uint2);  // the lane's first parked slot
        // the lane's fast events -> its range of the segment, each with {t, d | unit << 8 |
        // final in-segment offset << 16}
        {
            uint32_t e = 0;
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) {
                const uint32_t m = fe[j].mask;
                const uint32_t tag = (lane * N + j) << 8;
                uint32_t off = GENERIC ? lane_off + ((pre >> (8u * j)) & 0xffu) : lane_off + e;
                if (m & 1u) gstore(seg, poff + 8u * e, make_uint2(fe[j].ta, fe[j].da | tag | (off << 16)));
                e += m & 1u;
                off += m & 1u;
                if (m & 2u) gstore(seg, poff + 8u * e, make_uint2(fe[j].tb, fe[j].db | tag | (off << 16)));
                e += (m >> 1) & 1u;
                off += (m >> 1) & 1u;
                if (m & 4u) gstore(seg, poff + 8u * e, make_uint2(fe[j].tc, fe[j].dc | tag | (off << 16)));
                e += m >> 2;
            }
        }
        }
        if (GENERIC && gmask) {
            // units deeper than one fired level: the full arena walk (exec_step), levels >= 1
            // straight from / to the level planes; their events are parked behind the lane's
            // fast ones (every parked event carries its final offset, so the order is free)
            uint2 *gdst = seg + (excl >> 16) + nl;
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) {
                if (!((gmask >> j) & 1u)) continue;
                const uint32_t v = (vin_w >> (8 * j)) & 0xffu;
                PxState st;
                st.hdr = pack_hdr(px[j]);
                st.n0 = px[j].n0;
                st.lastf = px[j].lastf;
                DeepGlobal deep{a.lv_integ, a.lv_dt, a.lv_bdt, a.lv_bd, a.plane_stride, (size_t)u0 + j};
                EmitPark em{gdst, (lane * N + j) << 8, lane_off + ((pre >> (8 * j)) & 0xffu)};
                if (!exec_step(st, v, sc, deep, em)) raise(a.status, kStatusDepth);
                gdst = em.dst;
                px[j] = unpack_px(st);
                px[j].has0 = (st.hdr >> 24) & kFlagMMask;  // keep the full m
            }
        }
        vin_w = next_w;
    }

    // ---------------- state back to HBM ----------------
    {
        uint32_t hdrv[N];
        float liv[N], ldv[N], lbv[N], lfv[N];
        uint8_t bdv[N];
        uint32_t hor = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            hdrv[j] = GENERIC ? ((pack_hdr(px[j]) & 0x00ffffffu) | ((px[j].has0 | (px[j].popped << 5)) << 24))
                              : pack_hdr(px[j]);
            hor |= hdrv[j];
            liv[j] = px[j].n0.integ;
            ldv[j] = px[j].n0.dt;
            lbv[j] = px[j].n0.bdt;
            bdv[j] = (uint8_t)px[j].n0.bd;
            lfv[j] = px[j].lastf;
        }
        store_vec(a.hdr, u0, hdrv);
        if ((hor >> 24) & kFlagMMask) {
            store_vec(a.lv_integ, u0, liv);
            store_vec(a.lv_dt, u0, ldv);
            store_vec(a.lv_bdt, u0, lbv);
            store_vec(a.lv_bd, u0, bdv);
        }
        if (ABS_T) store_vec(a.lastf, u0, lfv);
        if (a.running) {  // side plane (the host keeps nb == 1 while it is enabled)
#pragma unroll
            for (uint32_t j = 0; j < N; ++j)
                if (u0 + j < a.n_units && ((hdrv[j] >> 24) & kFlagMMask))
                    a.running[u0 + j] = (uint8_t)frame_value_u8(px[j].n0.bd, f32_as_u32(px[j].n0.bdt),
                                                                (double)sc.ref_time);
        }
    }
}

__device__ __forceinline__ void expand_block(const BatchArgs *__restrict__ b, uint32_t f, uint32_t xblock);

// K1.  The grid holds the workgroups that step frames [f, f + nb) (a wave per segment) and,
// interleaved with them in dispatch order (the scarcer kind spread evenly through the other,
// see adder_launch_frame), the workgroups that expand frames
// [exp_f0, exp_f0 + exp_nf) of the PREVIOUS chunk (already scanned).  The expansion is
// memory-bound and the step VALU-bound, so sharing the SIMDs overlaps them; nothing in one
// role waits for the other.
template <bool COLLAPSE, bool ABS_T, bool GENERIC>
__global__ __launch_bounds__(kBlockThreads, GENERIC ? 4 : kFrameKernelWavesPerSimd) void adder_frame_kernel(
    const BatchArgs *__restrict__ b, uint32_t f, uint32_t nb, uint32_t exp_f0, uint32_t exp_blocks_per_frame,
    uint32_t grp_steps, uint32_t grp_exps, uint32_t groups, uint32_t rem_is_step) {
    // dispatch order: `groups` groups of (grp_steps step workgroups, grp_exps expansion workgroups),
    // then whatever is left of either kind
    uint32_t bid = blockIdx.x;
    if (exp_blocks_per_frame != 0u) {
        const uint32_t span = grp_steps + grp_exps;
        const uint32_t inter = groups * span;
        bool is_step;
        uint32_t idx;
        if (bid < inter) {
            const uint32_t g = bid / span, k = bid - g * span;
            is_step = k < grp_steps;
            idx = is_step ? g * grp_steps + k : g * grp_exps + (k - grp_steps);
        } else {  // only one kind has a remainder (see adder_launch_frame)
            is_step = rem_is_step != 0u;
            idx = (is_step ? groups * grp_steps : groups * grp_exps) + (bid - inter);
        }
        if (!is_step) {
            const uint32_t ef = idx / exp_blocks_per_frame;
            expand_block(b, exp_f0 + ef, idx - ef * exp_blocks_per_frame);
            return;
        }
        bid = idx;
    }
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t gw = bid * kWavesPerBlock + tid / kWave;  // the wave's segment
    const uint32_t u0 = gw * kWaveUnits + lane * kUnitsPerLane;
    const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
    RawSegment raw;
    load_raw<ABS_T, false>(a, u0, full, raw);
    run_segment<COLLAPSE, ABS_T, GENERIC>(b, a, nb, u0, gw, lane, raw);
}

// K1 at temporal depth 1 (the per-frame `consume` contract; HBM-bound): a wave takes TWO
// consecutive segments and issues the loads of both before it steps the first, so the second
// segment's memory round trip hides under the first one's step (with one segment per wave all
// resident waves load, then all compute, then all store, in lock step).
template <bool COLLAPSE, bool ABS_T>
__global__ __launch_bounds__(kBlockThreads, 6) void adder_frame1_kernel(const BatchArgs *__restrict__ b, uint32_t f) {
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t gw0 = (blockIdx.x * kWavesPerBlock + tid / kWave) * 2u;
    if (gw0 >= a.num_waves) return;
    RawSegment raw[2];
#pragma unroll
    for (uint32_t s = 0; s < 2u; ++s) {
        const uint32_t gw = gw0 + s;
        const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
        load_raw<ABS_T, true>(a, gw * kWaveUnits + lane * kUnitsPerLane, full, raw[s]);
    }
#pragma unroll
    for (uint32_t s = 0; s < 2u; ++s) {
        const uint32_t gw = gw0 + s;
        run_segment<COLLAPSE, ABS_T, false>(b, a, 1u, gw * kWaveUnits + lane * kUnitsPerLane, gw, lane, raw[s]);
    }
}

// ------------------------------------------------------------------------------------------
// Scan: exclusive prefix of the per-segment event counts of one frame per block
// (blockIdx.x = frame inside the chunk) and the frame's total.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void adder_scan_kernel(const BatchArgs *__restrict__ b, uint32_t f0) {
    const FrameArgs a = frame_args(b, f0 + blockIdx.x);
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
    if (tid == 0) *a.ftot = total;
}

// frame_offsets[f+1] = frame_offsets[f] + events(f) for the frames of the chunk, in order.
__global__ void adder_offsets_kernel(const BatchArgs *__restrict__ b, uint32_t f0, uint32_t nf) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t *offs = b->base.frame_offsets;
    uint64_t run = offs[f0];
    for (uint32_t i = 0; i < nf; ++i) {
        run += b->ftot_ring[(f0 + i) % b->slots];
        offs[f0 + i + 1] = run;
    }
}

// ------------------------------------------------------------------------------------------
// K2: parked events -> final 12-byte events of the ordered stream (blockIdx.y = frame
// inside the chunk).  A segment parks only a few dozen events, so a wave per segment would
// be nothing but start-up latency (kernel arguments -> metadata -> parked slots -> store).
// Each wave therefore takes kExpandSegs consecutive segments and issues ALL their loads --
// metadata and, speculatively, the first 64 parked slots of every segment -- before it
// consumes any of them: one memory round trip per wave.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kExpandSegs = ADDER_EXPAND_SEGS;
// One workgroup's share of a frame's expansion: 4 waves x kExpandSegs segments.
__device__ __forceinline__ void expand_block(const BatchArgs *__restrict__ b, uint32_t f, uint32_t xblock) {
    // only the frame-independent part of the arguments is needed here (no running_t fetch)
    const uint32_t slots = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t slot = __builtin_amdgcn_readfirstlane(f % slots);
    const uint32_t num_waves = __builtin_amdgcn_readfirstlane(b->base.num_waves);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t seg0 = __builtin_amdgcn_readfirstlane((xblock * kWavesPerBlock + threadIdx.x / kWave) * kExpandSegs);
    if (seg0 >= num_waves) return;
    const uint32_t park_stride = __builtin_amdgcn_readfirstlane(b->park_stride);
    // wave-uniform bases (SGPRs); the lanes add 32-bit byte offsets
    const uint2 *park = uniform_ptr(b->park_ring) + ((size_t)slot * num_waves + seg0) * park_stride;
    const uint32_t *wtot = uniform_ptr(b->wtot_ring) + (size_t)slot * num_waves + seg0;
    const uint32_t *wpref = uniform_ptr(b->wpref_ring) + (size_t)slot * num_waves + seg0;
    const uint32_t rowlen = __builtin_amdgcn_readfirstlane(b->base.rowlen);
    const uint32_t channels = __builtin_amdgcn_readfirstlane(b->base.channels);
    const uint32_t row_begin = __builtin_amdgcn_readfirstlane(b->base.row_begin);
    const uint64_t out_cap = b->base.out_cap;

    const bool park4 = __builtin_amdgcn_readfirstlane(b->base.park4) != 0u;  // compact 4-byte records (K1, PARK4)
    const uint32_t rt_u32 = park4 ? __builtin_amdgcn_readfirstlane(f32_as_u32(b->running_t[f])) : 0u;  // t of D_EMPTY

    // num_waves is a multiple of kExpandSegs (n_pad is padded accordingly)
    uint2 first[kExpandSegs];
#pragma unroll
    for (uint32_t q = 0; q < kExpandSegs; ++q) {
        if (park4)
            first[q] = make_uint2(gload<uint32_t>(park + (size_t)q * park_stride, lane * 4u), 0u);
        else
            first[q] = gload<uint2>(park + (size_t)q * park_stride, lane * (uint32_t)sizeof(uint2));
    }
    uint32_t my_tot = 0u, my_pref = 0u;
    if (lane < kExpandSegs) {
        my_tot = gload<uint32_t>(wtot, lane * 4u);
        my_pref = gload<uint32_t>(wpref, lane * 4u);
    }
    const uint64_t frame_base = b->base.frame_offsets[f];
    AdderEventPod *const out = uniform_ptr(b->base.out);
    bool dropped = false;
    // (row, offset in row) of the first unit of segment seg0: ONE wave-uniform division; the
    // following segments advance it with scalar add/compare instead of dividing again
    uint32_t y0 = __builtin_amdgcn_readfirstlane((seg0 * kWaveUnits) / rowlen);
    uint32_t rem0 = seg0 * kWaveUnits - y0 * rowlen;
    const bool one_wrap = rowlen >= kWaveUnits;
#pragma unroll
    for (uint32_t q = 0; q < kExpandSegs; ++q) {
        const uint32_t parked = __builtin_amdgcn_readlane(my_tot, q) >> 16;
        // the segment's first event in the stream (uniform); the lanes address relative to it
        const uint64_t base = frame_base + __builtin_amdgcn_readlane(my_pref, q);
        const uint64_t room64 = base < out_cap ? out_cap - base : 0ull;
        const uint32_t room = room64 > 0xffffffffull ? 0xffffffffu : (uint32_t)room64;  // events that still fit
        EventWords *const seg_out = reinterpret_cast<EventWords *>(out) + base;
        const uint2 *const seg_park = park + (size_t)q * park_stride;
        uint32_t wide_seen = 0u;  // marked records in earlier rounds of this segment (uniform)
        for (uint32_t i0 = 0; i0 < parked; i0 += kWave) {  // uniform trip count
            const uint32_t i = i0 + lane;
            const bool on = i < parked;
            uint32_t ev_t, ev_d, unit, pos;
            if (park4) {
                const uint32_t rec = !on ? 0u : (i0 == 0u ? first[q].x : gload<uint32_t>(seg_park, i * 4u));
                ev_t = rec & kParkWideT;
                ev_d = (rec >> 17) & 0xffu;
                unit = rec >> 25;
                pos = i;
                const bool marked = on && ev_d != kDEmpty && ev_t == kParkWideT;
                const uint64_t mb = __builtin_amdgcn_ballot_w64(marked);
                if (mb != 0ull) {  // rare: fetch the real t from the segment's overflow list
                    const uint32_t rank = wide_seen + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
                    if (marked) ev_t = gload<uint32_t>(seg_park, (kParkOverflowBase + rank) * 4u);
                    wide_seen += (uint32_t)__popcll(mb);
                }
                ev_t = ev_d == kDEmpty ? rt_u32 : ev_t;
            } else {
                const uint2 sl = !on ? make_uint2(0u, 0u)
                                     : (i0 == 0u ? first[q] : gload<uint2>(seg_park, i * (uint32_t)sizeof(uint2)));
                ev_t = sl.x;
                ev_d = sl.y & 0xffu;
                unit = (sl.y >> 8) & 0xffu;
                pos = sl.y >> 16;  // final offset inside the segment
            }
            if (!on) continue;
            uint32_t rem = rem0 + unit;
            uint32_t y = y0;
            if (one_wrap) {
                const bool wrap = rem >= rowlen;
                rem -= wrap ? rowlen : 0u;
                y += wrap ? 1u : 0u;
            } else {
                const uint32_t dq = rem / rowlen;
                rem -= dq * rowlen;
                y += dq;
            }
            uint32_t x = rem, c = 0xffu;
            if (channels == 3u) {
                x = (uint32_t)(((uint64_t)rem * 0xAAAAAAABull) >> 33);  // rem / 3
                c = rem - 3u * x;
            }
            if (pos < room) {
                EventWords w;
                w.xy = x | ((y + row_begin) << 16);
                w.cd = c | (ev_d << 8);
                w.t = ev_t;
                gstore(seg_out, pos * (uint32_t)sizeof(EventWords), w);
            } else {
                dropped = true;
            }
        }
        // next segment starts kWaveUnits units later (uniform, scalar)
        rem0 += kWaveUnits;
        while (rem0 >= rowlen) {
            rem0 -= rowlen;
            ++y0;
        }
    }
    if (dropped) raise(b->base.status, kStatusCapacity);
}

__global__ __launch_bounds__(kBlockThreads) void adder_expand_kernel(const BatchArgs *__restrict__ b, uint32_t f0) {
    expand_block(b, f0 + blockIdx.y, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// K3: the raw sink's event serialisation (RawOutput::ingest_event, raw/stream.rs:101-120 =
// bincode fixint big-endian) on the device: 12-byte AdderEvents -> 9-byte `EventSingle`
// {x u16, y u16, d u8, t u32} records on a 1-channel plane, 11-byte `Event` {x, y, 0x01, c, d, t}
// records otherwise.  The host then writes the bytes as they are (and D2H moves 9 instead of
// 12 bytes per event).  A workgroup converts kWireEvents events: coalesced dword loads into
// LDS, per-event repack in LDS, coalesced dword stores of the record bytes (kWireEvents is a
// multiple of 4, so every workgroup's output starts dword-aligned).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kWireEvents = 1024;
__global__ __launch_bounds__(256) void adder_wire_kernel(const uint32_t *__restrict__ ev, uint64_t n, uint32_t rec,
                                                         uint8_t *__restrict__ out, uint32_t *status) {
    __shared__ uint32_t s_w[kWireEvents * 3];
    const uint64_t e0 = (uint64_t)blockIdx.x * kWireEvents;
    const uint32_t cnt = (uint32_t)min((uint64_t)kWireEvents, n - e0);
    const uint32_t tid = threadIdx.x;
    const uint32_t *src = ev + e0 * 3u;
    for (uint32_t i = tid; i < cnt * 3u; i += 256u) s_w[i] = src[i];
    __syncthreads();
    bool bad = false;
    for (uint32_t e = tid; e < cnt; e += 256u) {
        const uint32_t xy = s_w[3 * e], cd = s_w[3 * e + 1], t = s_w[3 * e + 2];
        const uint32_t x = xy & 0xffffu, y = xy >> 16, c = cd & 0xffu, d = (cd >> 8) & 0xffu;
        // record bytes, little-endian packed into dwords: byte k of the record = bits 8k.. of (w0,w1,w2)
        const uint32_t w0 = (x >> 8) | ((x & 0xffu) << 8) | ((y >> 8) << 16) | ((y & 0xffu) << 24);
        uint32_t w1, w2;
        if (rec == 9u) {  // d, t3, t2, t1 | t0
            w1 = d | ((t >> 24) << 8) | (((t >> 16) & 0xffu) << 16) | (((t >> 8) & 0xffu) << 24);
            w2 = t & 0xffu;
        } else {  // 0x01, c, d, t3 | t2, t1, t0
            bad |= c == 0xffu;  // Option::None inside a multi-channel plane is a 10-byte record: not produced by the integrator
            w1 = 1u | (c << 8) | (d << 16) | ((t >> 24) << 24);
            w2 = ((t >> 16) & 0xffu) | (((t >> 8) & 0xffu) << 8) | ((t & 0xffu) << 16);
        }
        s_w[3 * e] = w0;
        s_w[3 * e + 1] = w1;
        s_w[3 * e + 2] = w2;
    }
    __syncthreads();
    const uint32_t bytes = cnt * rec;
    uint8_t *dst = out + e0 * rec;
    const uint32_t magic = rec == 9u ? 0x38e38e39u : 0xba2e8ba3u;  // floor(B / rec) = mulhi(B, magic) >> (1 | 3)
    const uint32_t sh = rec == 9u ? 1u : 3u;
    for (uint32_t k = tid; k * 4u < bytes; k += 256u) {
        uint32_t o = 0u;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t B = 4u * k + j;
            const uint32_t e = __umulhi(B, magic) >> sh;
            const uint32_t r = B - e * rec;
            const uint32_t byte = B < bytes ? (s_w[3u * e + (r >> 2)] >> (8u * (r & 3u))) & 0xffu : 0u;
            o |= byte << (8u * j);
        }
        if (4u * k + 4u <= bytes) {
            reinterpret_cast<uint32_t *>(dst)[k] = o;
        } else {  // the stream's last, partial dword
            for (uint32_t j = 0; 4u * k + j < bytes; ++j) dst[4u * k + j] = (uint8_t)(o >> (8u * j));
        }
    }
    if (bad) raise(status, kStatusWire);
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

// Exhaustive check of fdiv_small against the IEEE division on its whole domain
// (a = blockIdx.x * 256 + threadIdx.x + 1 in [1, 2^24], b = blockIdx.y + 1 in [1, 255]).
__global__ __launch_bounds__(256) void adder_divtest_kernel(unsigned long long *bad) {
    const float a = (float)(blockIdx.x * 256u + threadIdx.x + 1u), b = (float)(blockIdx.y + 1u);
    if (f32_as_u32(fdiv_small(a, b)) != f32_as_u32(fdiv(a, b))) atomicAdd(bad, 1ull);
}

typedef void (*FrameKernelFn)(const BatchArgs *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                              uint32_t);
static FrameKernelFn pick_frame_kernel(uint32_t variant) {
    const bool collapse = variant & 1u, abs_t = variant & 2u, generic = variant & 4u;
    if (collapse) {
        if (generic) return abs_t ? adder_frame_kernel<true, true, true> : adder_frame_kernel<true, false, true>;
        return abs_t ? adder_frame_kernel<true, true, false> : adder_frame_kernel<true, false, false>;
    }
    return abs_t ? adder_frame_kernel<false, true, true> : adder_frame_kernel<false, false, true>;
}

extern "C" hipError_t adder_launch_divtest(unsigned long long *d_bad, hipStream_t stream) {
    hipLaunchKernelGGL(adder_divtest_kernel, dim3((1u << 24) / 256u, 255), dim3(256), 0, stream, d_bad);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_frame(const BatchArgs *b, uint32_t f, uint32_t nb, uint32_t variant,
                                         uint32_t num_waves, uint32_t exp_f0, uint32_t exp_nf, hipStream_t stream) {
    if (nb == 1u && !(variant & 4u) && exp_nf == 0u && (num_waves & 1u) == 0u) {
        const uint32_t grid = (num_waves / 2u + kWavesPerBlock - 1) / kWavesPerBlock;
        switch (variant & 3u) {
            case 0: hipLaunchKernelGGL((adder_frame1_kernel<false, false>), dim3(grid), dim3(kBlockThreads), 0, stream, b, f); break;
            case 1: hipLaunchKernelGGL((adder_frame1_kernel<true, false>), dim3(grid), dim3(kBlockThreads), 0, stream, b, f); break;
            case 2: hipLaunchKernelGGL((adder_frame1_kernel<false, true>), dim3(grid), dim3(kBlockThreads), 0, stream, b, f); break;
            default: hipLaunchKernelGGL((adder_frame1_kernel<true, true>), dim3(grid), dim3(kBlockThreads), 0, stream, b, f); break;
        }
        return hipGetLastError();
    }
    const uint32_t S = (num_waves + kWavesPerBlock - 1) / kWavesPerBlock;  // step workgroups
    const uint32_t per_block = kWavesPerBlock * kExpandSegs;
    const uint32_t exp_bpf = exp_nf ? (num_waves + per_block - 1) / per_block : 0u;
    const uint32_t E = exp_bpf * exp_nf;  // expansion workgroups
    // spread the scarcer kind evenly through the dispatch order
    uint32_t grp_steps = 1, grp_exps = 1, groups = 0, rem_is_step = 1;
    if (E >= S) {
        grp_exps = E / S;
        groups = S;
        rem_is_step = 0;
    } else if (E) {
        grp_steps = S / E;
        groups = E;
    }
    hipLaunchKernelGGL(pick_frame_kernel(variant), dim3(S + E), dim3(kBlockThreads), 0, stream, b, f, nb, exp_f0, exp_bpf,
                       grp_steps, grp_exps, groups, rem_is_step);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_wire(const AdderEventPod *ev, uint64_t n, uint32_t rec, uint8_t *out, uint32_t *status,
                                        hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t grid = (n + kWireEvents - 1) / kWireEvents;
    hipLaunchKernelGGL(adder_wire_kernel, dim3((uint32_t)grid), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t *>(ev), n, rec, out, status);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_scan(const BatchArgs *b, uint32_t f0, uint32_t nf, hipStream_t stream) {
    hipLaunchKernelGGL(adder_scan_kernel, dim3(nf), dim3(kScanThreads), 0, stream, b, f0);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_offsets(const BatchArgs *b, uint32_t f0, uint32_t nf, hipStream_t stream) {
    hipLaunchKernelGGL(adder_offsets_kernel, dim3(1), dim3(64), 0, stream, b, f0, nf);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_expand(const BatchArgs *b, uint32_t f0, uint32_t nf, uint32_t num_waves,
                                          hipStream_t stream) {
    const uint32_t per_block = kWavesPerBlock * kExpandSegs;  // segments per block
    const uint32_t grid = (num_waves + per_block - 1) / per_block;
    hipLaunchKernelGGL(adder_expand_kernel, dim3(grid, nf), dim3(kBlockThreads), 0, stream, b, f0);
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
