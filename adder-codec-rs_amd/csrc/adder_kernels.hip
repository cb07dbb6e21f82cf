// adder_kernels.hip -- CDNA4 (gfx950) kernels of the framed->ADDER integration path.
//
// One input frame goes through a wait-free three-kernel pipeline that does what the
// reference's rayon loop does (adder-codec-rs/src/transcoder/source/video.rs:677-734):
// every pixel-channel runs integrate_for_px (video.rs:1318-1380) and the emitted events
// are gathered in raster order (y, x, c, per-pixel emission order).
//
//   K1 adder_lean_kernel / adder_frame_kernel   one lane = kUnitsPerLane consecutive
//        pixel-channels, one wave = one segment.  Loads level 0 of the arena (16 bytes per
//        unit, structure-of-arrays planes resident in HBM across frames) and the frame bytes as
//        coalesced vectors and steps up to 32 consecutive frames with the state in registers.
//        Per frame and unit the step leaves at most ONE 12-byte record (the lean variants: the raw
//        material of its <= 3 events) -- the records are compacted per wave with ballot + mbcnt
//        (no barrier, no atomics, no LDS) into the frame's scratch segment, the segment's event and
//        record counts go to wtot.
//   Ks adder_scan_kernel    exclusive prefix over the per-segment counts (one block per
//        frame) + adder_offsets_kernel (the frame_offsets chain); run once per CHUNK of frames.
//   K2 adder_expand_kernel  reads the parked records linearly, decodes them and writes each 12-byte
//        event to its final slot of the ordered stream (coordinates from the unit index).  Its own
//        kernel, on a second stream beside the frame kernel of the next chunk; both are launched with
//        a few workgroups per CU that walk their work, so both are resident on every CU.
//
// No kernel waits on another workgroup, so there is no residency requirement, no spin
// loop and nothing that can hang.  Pixels whose arena is deeper than one fired level
// (Normal mode, or delta_t_max > time_spanned) run the phased generic step (gen_root / gen_emit /
// gen_walk / gen_pop) of adder_frame_kernel, with levels 1..4 of the arena held in LDS for the launch.
// HBM/VALU-bound integer/f32 work: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "adder_kernels.h"
#include "adder_pixel.hpp"
#include "adder_kernel_util.hpp"

namespace adder {

// levels k >= 1 of one unit, straight from / to the deep planes (index k - 1)
struct DeepGlobal {
    float *integ, *dt, *bdt;
    uint8_t *bd;
    size_t stride;
    size_t u;
    __device__ __forceinline__ void load(uint32_t k, Node &n) const {
        const size_t i = (size_t)(k - 1u) * stride + u;
        n.integ = integ[i];
        n.dt = dt[i];
        n.bdt = bdt[i];
        n.bd = bd[i];
    }
    __device__ __forceinline__ void store(uint32_t k, const Node &n) const {
        const size_t i = (size_t)(k - 1u) * stride + u;
        integ[i] = n.integ;
        dt[i] = n.dt;
        bdt[i] = n.bdt;
        bd[i] = (uint8_t)n.bd;
    }
};

// BatchArgs::snap_dv_*: the unit's live levels into the batch's undo copy (first launch of the batch only)
__device__ __forceinline__ bool snap_deep_wanted(const BatchArgs *__restrict__ b, const FrameArgs &a) {
    return __builtin_amdgcn_readfirstlane(a.frame_idx == 0u && b->snap_dv_integ != nullptr);
}
__device__ __forceinline__ void snap_deep_levels(const BatchArgs *__restrict__ b, const FrameArgs &a, size_t u, uint32_t m) {
    const DeepGlobal src{a.dv_integ, a.dv_dt, a.dv_bdt, a.dv_bd, a.plane_stride, u};
    const DeepGlobal dst{b->snap_dv_integ, b->snap_dv_dt, b->snap_dv_bdt, b->snap_dv_bd, a.plane_stride, u};
    for (uint32_t k = 1; k < m; ++k) {
        Node n;
        src.load(k, n);
        dst.store(k, n);
    }
}

struct __attribute__((aligned(4))) EventWords {
    uint32_t xy, cd, t;
};

// gen_emit's events of one unit, parked in final order with their final in-segment offset
struct EmitPark {
    uint2 *seg;     // the segment's run of this frame (wave-uniform)
    uint32_t tag;   // unit_in_wave << 8
    uint32_t off;   // final offset of the unit's next event inside the segment
    __device__ __forceinline__ void operator()(uint32_t d, uint32_t t);
};

// kUnitsPerLane consecutive values as one vector access
template <class T, int N>
struct VecOf;
template <> struct VecOf<uint32_t, 4> { using type = uint4; };
template <> struct VecOf<uint32_t, 2> { using type = uint2; };
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<float, 2> { using type = float2; };
template <> struct VecOf<uint8_t, 4> { using type = uint32_t; };
template <> struct VecOf<uint8_t, 2> { using type = uint16_t; };

__device__ __forceinline__ void EmitPark::operator()(uint32_t d, uint32_t t) {
    gstore<uint2>(seg, off * 8u, make_uint2(t, d | tag | (off << 16)));  // (uniform base + the lane's 32-bit offset)
    ++off;
}
// The bounded Collapse kernel's records: {t, code | unit << 9 | final offset << 16} with the 9-bit d code of
// adder_pixel.hpp (cb_d_from_code): the top nine bits of the threshold word, so one v_alignbit builds the word.
struct EmitCb {
    uint2 *seg;       // the segment's run of this frame (wave-uniform)
    uint32_t tagoff;  // unit_in_wave | offset << 7
    uint32_t boff;    // offset * 8: where the next record goes
    __device__ __forceinline__ void put(uint32_t w1, uint32_t t) {
        gstore<uint2>(seg, boff, make_uint2(t, w1));
        tagoff += 1u << 7;
        boff += kGenRecBytes;
    }
    __device__ __forceinline__ void ev(uint32_t thr_bits, uint32_t t) { put(__builtin_amdgcn_alignbit(tagoff, thr_bits, 23), t); }
    __device__ __forceinline__ void filler(uint32_t t) { put((tagoff << 9) | kCbCodeEmpty, t); }
};
template <bool NT = false, class T>
__device__ __forceinline__ void load_vec(const T *plane, uint32_t u0, T (&v)[kUnitsPerLane]) {
    using V = typename VecOf<T, kUnitsPerLane>::type;
    const V x = NT ? gload_nt<V>(plane, u0 * (uint32_t)sizeof(T)) : gload<V>(plane, u0 * (uint32_t)sizeof(T));
    __builtin_memcpy(v, &x, sizeof(V));
}
template <bool NT = false, class T>
__device__ __forceinline__ void store_vec(T *plane, uint32_t u0, const T (&v)[kUnitsPerLane]) {
    using V = typename VecOf<T, kUnitsPerLane>::type;
    V x;
    __builtin_memcpy(&x, v, sizeof(V));
    if (NT) gstore_nt<V>(plane, u0 * (uint32_t)sizeof(T), x);
    else gstore<V>(plane, u0 * (uint32_t)sizeof(T), x);
}

// the lane's kUnitsPerLane input bytes of one frame, packed little-endian
__device__ __forceinline__ uint32_t load_input(const uint8_t *frame, uint32_t u0, uint32_t n_units) {
    uint32_t w = 0u;
    if (u0 + kUnitsPerLane <= n_units) {
        using InV = typename VecOf<uint8_t, kUnitsPerLane>::type;
        w = ADDER_NT_INPUT ? gload_nt<InV>(frame, u0) : gload<InV>(frame, u0);
    } else {
#pragma unroll
        for (uint32_t j = 0; j < kUnitsPerLane; ++j)
            if (u0 + j < n_units) w |= (uint32_t)gload<uint8_t>(frame, u0 + j) << (8 * j);
    }
    return w;
}

// packed 16-bit operations of the quiet groups' statistics (two units of a lane per instruction)
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) {  // max(a - b, 0) per half
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {  // a * b + c per half
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// min over each row of 16 lanes, left in the row's last lane (DPP row shifts; lanes without a source keep their own value)
__device__ __forceinline__ uint32_t row16_min_to_last(uint32_t x) {
    uint32_t y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x111, 0xf, 0xf, false); x = y < x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x112, 0xf, 0xf, false); x = y < x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x114, 0xf, 0xf, false); x = y < x ? y : x;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x118, 0xf, 0xf, false); x = y < x ? y : x;
    return x;
}

// ------------------------------------------------------------------------------------------
// K1, lean variants (Collapse with delta_t_max <= time_spanned: BASELINE configs 2-4).
// One segment (64 lanes x kUnitsPerLane units) through `nb` consecutive frames starting at
// a.frame_idx.  The pixel state lives in registers for the whole run (temporal blocking): it
// is read from HBM once and written back once; per frame only the input bytes are loaded
// (one frame ahead) and the segment's records are compacted into that frame's scratch slot.
// ------------------------------------------------------------------------------------------
struct LeanRaw {  // a segment's state as loaded (one memory round trip, nothing consumed yet)
    uint32_t hdrv[kUnitsPerLane];
    float iv[kUnitsPerLane], dv[kUnitsPerLane], bv[kUnitsPerLane], lfv[kUnitsPerLane];
    uint32_t vin_w;
};

template <bool ABS_T, bool NT = false>
__device__ __forceinline__ void lean_load(const FrameArgs &a, uint32_t u0, bool full, LeanRaw &r) {
    load_vec<NT>(a.hdr, u0, r.hdrv);
    r.vin_w = load_input(a.frame, u0, full ? 0xffffffffu : a.n_units);
    // level 0 is fetched whether or not the unit has one (m == 0 leaves it unread by the step)
    load_vec<NT>(a.integ0, u0, r.iv);
    load_vec<NT>(a.dt0, u0, r.dv);
    load_vec<NT>(a.bdt0, u0, r.bv);
    if (ABS_T) {
        load_vec<NT>(a.lastf, u0, r.lfv);
    } else {
#pragma unroll
        for (uint32_t j = 0; j < kUnitsPerLane; ++j) r.lfv[j] = 0.0f;
    }
}

// a unit's record into its scratch slot: 12 bytes {ta, tc, w} (AbsoluteT) or 8 bytes {ta, w8} (DeltaT)
template <bool ABS_T, bool NT>
__device__ __forceinline__ void lean_store_rec(void *seg, uint32_t byte_off, const LeanRec &r) {
    struct R12 { uint32_t a, b, c; };
    if (ABS_T) {
        const R12 v{r.ta, r.tc, r.w};
        if (NT) gstore_nt(seg, byte_off, v);
        else gstore(seg, byte_off, v);
    } else {
        const uint2 v = make_uint2(r.ta, r.w8);
        if (NT) gstore_nt(seg, byte_off, v);
        else gstore(seg, byte_off, v);
    }
}

// FULL: the whole wave lies inside the band (every wave but possibly the band's last ones): no
// per-unit bounds handling inside the frame loop.
#ifndef ADDER_LEAN_QUIET_PATH
#define ADDER_LEAN_QUIET_PATH 1
#endif
#ifndef ADDER_LEAN_QUIET_GROUPS
#define ADDER_LEAN_QUIET_GROUPS 1  // (0: A/B build without the group form of the quiet loop)
#endif
// LOG: the records are appended to the segment's log of the chunk (BatchArgs::log_cap != 0: dense, frame after frame,
// a frame's run found through wofs) instead of one fixed slot per frame (park_layout).
template <bool ABS_T, bool FULL, uint32_t NB_MAX, bool LOG = false>
__device__ __forceinline__ void lean_frames(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb,
                                            uint32_t u0, uint32_t gw, uint32_t lane, const LeanRaw &raw,
                                            uint8_t *lds_in) {
    constexpr uint32_t N = kUnitsPerLane;
    using L = WaveLanes;
    LeanPxT<L> px[N];
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) px[j] = lean_unpack<L>(raw.hdrv[j], raw.iv[j], raw.dv[j], raw.bv[j], raw.lfv[j]);
    StepConsts sc = a.sc;
    const float T = sc.time_spanned;

    // wave-uniform bases of the per-frame accesses, in SGPRs
    const uint32_t sgw = __builtin_amdgcn_readfirstlane(gw);
    const uint32_t slots_u = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t num_waves_u = __builtin_amdgcn_readfirstlane(a.num_waves);
    const uint32_t n_units_u = __builtin_amdgcn_readfirstlane(a.n_units);
    const uint32_t f0 = __builtin_amdgcn_readfirstlane(a.frame_idx);
    const uint32_t slot0 = __builtin_amdgcn_readfirstlane(f0 % slots_u);
    const uint32_t park_bytes_u = __builtin_amdgcn_readfirstlane(b->park_bytes);
    // the launch's rows of the per-frame table (nb <= kMaxFramesPerLaunch <= 64): one vector load, then a
    // lane read per frame -- no memory wait inside the frame loop
    uint32_t tab_cth = 0u, tab_rt = 0u;
    if (lane < nb) {
        const uint2 e = gload<uint2>(uniform_ptr(b->ftab), (f0 + lane) * (uint32_t)sizeof(FrameTab));
        tab_rt = e.x;
        tab_cth = e.y;
    }
    // the segment's scratch of the launch's first frame and the next frame's input bytes: 64-bit
    // uniform pointers advanced by adds (the ring wraps at `slots`)
    // (a launch never crosses a chunk boundary, so the segment's frames are contiguous: park_offset)
    const uint32_t chunk_u = __builtin_amdgcn_readfirstlane(b->chunk);
    const ParkLayout lay = park_layout_u(b);
    const uint32_t frame_stride_u = lay.frame_stride;
    uint8_t *seg = uniform_ptr(b->park_ring) + (LOG ? (size_t)0 : park_offset(slot0, sgw, chunk_u, num_waves_u, park_bytes_u, lay));
    // the segment's frame slots may be rotated (ParkLayout): the walk wraps where the chunk's slots end
    uint32_t ridx = __builtin_amdgcn_readfirstlane((slot0 % chunk_u + (sgw >> lay.rot_shift)) & lay.rot_mask);
    const uint32_t wrap_bytes = chunk_u * frame_stride_u;
    // LOG: the segment's region of the chunk holds log_cap = 128 * chunk records (at most one per unit and frame: it
    // cannot overflow); the cursor survives from one launch of a chunk to the next in wcur
    uint32_t log_cur0 = 0u;
    uint32_t *wcur_p = nullptr;
    if (LOG) {
        const uint32_t cap = __builtin_amdgcn_readfirstlane(b->log_cap);
        const uint32_t cir = slot0 / chunk_u;
        const size_t seg_idx = (size_t)cir * num_waves_u + sgw;
        wcur_p = uniform_ptr(b->wcur) + seg_idx;
        if (slot0 != cir * chunk_u) log_cur0 = __builtin_amdgcn_readfirstlane(*wcur_p);  // not the chunk's first launch
        seg += (seg_idx * cap + log_cur0) * lean_rec_bytes(ABS_T);
    }
    // The input bytes of ALL the launch's frames are requested up front and parked in the wave's slice of
    // LDS: the record stores of the frame loop sit in divergent regions, so the compiler cannot count
    // them, and every global load waited for inside the loop would cost a full `s_waitcnt vmcnt(0)` --
    // i.e. the acknowledgement of the previous frame's stores.  LDS reads wait on lgkmcnt instead, so the
    // loop only ever issues stores to memory.
    using InT = typename VecOf<uint8_t, N>::type;
    InT *const in_lds = reinterpret_cast<InT *>(lds_in) + lane;  // [frame][lane]
    if constexpr (NB_MAX > 1u) {
        // (frames past the launch's last one re-read that one: no control flow between the loads)
        const uint8_t *const fr0 = uniform_ptr(b->frames) + (size_t)f0 * n_units_u;
#if ADDER_LDS_DIRECT_INPUT
        // Straight into LDS (global_load_lds_dwordx4: no staging registers -- 63 of them held the kernel at 111
        // VGPRs): lane i fetches 16 bytes of frame 8 g + i / 8, and the instruction parks lane i's bytes at
        // M0 + 16 i, which is exactly the [frame][128 bytes] layout the loop reads.  Needs whole, 16-byte aligned
        // segments; anything else takes the register path eight frames at a time.
        static_assert(kWaveUnits == 128u && NB_MAX % 8u == 0u, "eight frames of one segment per instruction");
        const bool direct = FULL && __builtin_amdgcn_readfirstlane(((n_units_u | (uint32_t)(uintptr_t)fr0) & 15u) == 0u);
        if (direct) {
            const uint8_t *const seg_in = fr0 + (size_t)sgw * kWaveUnits + (lane & 7u) * 16u;
#pragma unroll
            for (uint32_t g = 0; g < NB_MAX / 8u; ++g) {
                uint32_t k = g * 8u + (lane >> 3);
                k = k < nb ? k : nb - 1u;
                __builtin_amdgcn_global_load_lds((const ADDER_GLOBAL void *)(seg_in + (size_t)k * n_units_u),
                                                 (__attribute__((address_space(3))) void *)(lds_in + g * 1024u), 16, 0,
                                                 ADDER_NT_INPUT ? 2 : 0);
            }
        } else
#endif
        {
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < NB_MAX; k0 += 8u) {
                uint32_t vin8[8];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) {
                    const uint32_t k = k0 + q;
                    const uint32_t kk = k < nb ? k : nb - 1u;  // uniform
                    vin8[q] = load_input(fr0 + (size_t)kk * n_units_u, u0, FULL ? 0xffffffffu : n_units_u);
                }
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) in_lds[(k0 + q) * kWave] = (InT)vin8[q];
            }
        }
        // vmcnt(0): everything requested so far has landed -- the LDS writes of the direct loads, the frame table --
        // so that nothing inside the frame loop ever waits on memory (a wait there would also wait for the record stores)
        __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    uint32_t vin_w = raw.vin_w;
    uint32_t wt = 0u;  // lane i: {events | records << 16} of the launch's i-th frame
    uint64_t active[N];  // units inside the band (the rest of the wave's segment is padding)
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) active[j] = FULL ? ~0ull : L::from(u0 + j < n_units_u);

    uint32_t i = 0u;
#if ADDER_LEAN_QUIET_PATH
    // ---------------- quiet frames first (lean_quiet / lean_step_quiet, adder_pixel.hpp) ----------------
    // A wave ALL of whose units hold their root and are popped (or black) stays in this loop for as long as every unit also
    // passes its contrast test: the roots accumulate or fire, nothing leaves, no record is parked -- static content, lossy
    // content away from what moves.  The first frame that is not quiet, and every later one of the launch, goes through the
    // general loop below, which carries no test (adder_cb_kernel has the same loop; a test inside the general loop cost it
    // 9-13 %).  Blocked launches only.
    if constexpr (NB_MAX > 1u) {
        uint64_t ok = ~0ull;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) ok &= px[j].has0 & (px[j].popped | L::from(px[j].integ == 0.0f));
        if (ok == ~0ull) {  // uniform
#if ADDER_LEAN_QUIET_GROUPS
            // the smallest c_thresh of every group of kQuietGroup frames (lane 16 g + 15: group g; frames past the launch do not count)
            const uint32_t tab_cmin = row16_min_to_last(lane < nb ? tab_cth : 0xffu);
#endif
            for (; i < nb;) {
#if ADDER_LEAN_QUIET_GROUPS
                // ---------------- a whole group of frames at once (quiet_group_apply, adder_pixel.hpp; adder_cb_kernel's form) ----------------
                if ((i % kQuietGroup) == 0u) {
                    static_assert(N == 2u, "two units per lane in one packed register");
                    const uint32_t gn = nb - i < kQuietGroup ? nb - i : kQuietGroup;  // (uniform)
                    const uint32_t cth_min = __builtin_amdgcn_readlane(tab_cmin, i + kQuietGroup - 1u);
                    const uint32_t need2 = quiet_group_need(px[0].integ, px[0].thr) | (quiet_group_need(px[1].integ, px[1].thr) << 16);
                    uint32_t mn = 0x00ff00ffu, mx = 0u, P = 0u, cnt = 0u, pm = 0u;
                    for (uint32_t k = 0; k < gn; ++k) {
                        const uint32_t pk = __builtin_amdgcn_perm(0u, (uint32_t)in_lds[(i + k) * kWave], 0x0c010c00u);  // {v0, v1} as halves
                        mn = pk_min_u16(mn, pk);
                        mx = pk_max_u16(mx, pk);
                        P = pk_add_u16(P, pk);
                        const uint32_t below = pk_min_u16(pk_sub_sat_u16(need2, P), 0x00010001u);  // prefix sum < need
                        cnt = pk_add_u16(cnt, below);
                        pm = pk_mad_u16(pk, below, pm);
                    }
                    LeanPxT<L> t[N];
                    uint32_t verdict[N];
                    bool any_no = false, any_slow = false;
#pragma unroll
                    for (uint32_t j = 0; j < N; ++j) {
                        QuietGroupStats g;
                        g.mn = (mn >> (16u * j)) & 0xffffu;
                        g.mx = (mx >> (16u * j)) & 0xffffu;
                        g.sum = (P >> (16u * j)) & 0xffffu;
                        g.cnt = (cnt >> (16u * j)) & 0xffffu;
                        g.pm = (pm >> (16u * j)) & 0xffffu;
                        const uint32_t row = g.cnt < gn ? g.cnt : gn - 1u;
                        g.vc = ((uint32_t)in_lds[(i + row) * kWave] >> (8u * j)) & 0xffu;
                        t[j] = px[j];
                        verdict[j] = lean_group_apply<L>(t[j], g, gn, cth_min, T);
                        any_no = any_no || verdict[j] == kQuietNo;
                        any_slow = any_slow || verdict[j] == kQuietSlow;
                    }
                    if (__builtin_amdgcn_ballot_w64(any_no) == 0ull) {  // uniform: every unit of the wave is quiet in every frame
#pragma unroll
                        for (uint32_t j = 0; j < N; ++j) {  // (only the root's four floats move; the masks are the wave's)
                            const bool done = verdict[j] == kQuietDone;
                            px[j].integ = done ? t[j].integ : px[j].integ;
                            px[j].dt = done ? t[j].dt : px[j].dt;
                            px[j].bdt = done ? t[j].bdt : px[j].bdt;
                            px[j].thr = done ? t[j].thr : px[j].thr;
                        }
                        if (__builtin_amdgcn_ballot_w64(any_slow) != 0ull) {  // (rare: second firings, black roots that wake up)
                            for (uint32_t k = 0; k < gn; ++k) {
                                const uint32_t vin_s = (uint32_t)in_lds[(i + k) * kWave];
#pragma unroll
                                for (uint32_t j = 0; j < N; ++j) {
                                    LeanPxT<L> u = px[j];
                                    lean_step_quiet<L, true>(u, (vin_s >> (8 * j)) & 0xffu, T);
                                    const bool slow = verdict[j] == kQuietSlow;
                                    px[j].integ = slow ? u.integ : px[j].integ;
                                    px[j].dt = slow ? u.dt : px[j].dt;
                                    px[j].bdt = slow ? u.bdt : px[j].bdt;
                                    px[j].thr = slow ? u.thr : px[j].thr;
                                }
                            }
                        }
                        i += gn;  // (wt of these frames stays 0: no events, no records)
                        continue;
                    }
                }
#endif
                const uint32_t vin_q = (uint32_t)in_lds[i * kWave];
                const uint32_t cth_q = __builtin_amdgcn_readlane(tab_cth, i);
                uint64_t quiet = ~0ull, keeps = ~0ull;
#pragma unroll
                for (uint32_t j = 0; j < N; ++j) {
                    const uint32_t v = (vin_q >> (8 * j)) & 0xffu;
                    quiet &= lean_quiet<L>(px[j], v, cth_q);
                    keeps &= lean_quiet_keeps<L>(px[j], v);
                }
                if (quiet != ~0ull) break;  // frame i is the general loop's
                if (keeps == ~0ull) {
#pragma unroll
                    for (uint32_t j = 0; j < N; ++j) lean_step_quiet<L, false>(px[j], (vin_q >> (8 * j)) & 0xffu, T);
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < N; ++j) lean_step_quiet<L, true>(px[j], (vin_q >> (8 * j)) & 0xffu, T);
                }
                // (wt of this frame stays 0: no events, no records; the slot walk below starts at frame i)
                ++i;
            }
            vin_w = i < nb ? (uint32_t)in_lds[i * kWave] : 0u;
            if (!LOG) {
                for (uint32_t k = 0; k < i; ++k) {  // uniform: the frames skipped above own their slots all the same
                    seg += frame_stride_u;
                    if (++ridx == chunk_u) {
                        ridx = 0u;
                        seg -= wrap_bytes;
                    }
                }
            }
        }
    }
#endif
    for (; i < nb; ++i) {
        uint32_t next_w = 0u;
        if (NB_MAX > 1u && i + 1u < nb) next_w = (uint32_t)in_lds[(i + 1u) * kWave];  // one frame ahead
        const uint32_t cth = __builtin_amdgcn_readlane(tab_cth, i);
        if (ABS_T) sc.running_t = __uint_as_float(__builtin_amdgcn_readlane(tab_rt, i));

        // ---------------- the step: <= one record per unit ----------------
        LeanRec rec[N];
        uint64_t mrec[N];
        uint32_t nev = 0u, nrec = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const uint32_t v = (vin_w >> (8 * j)) & 0xffu;
            const uint32_t tag = (lane * N + j) << kLeanUnitShift;
            LeanFlagsT<L> fl = lean_step<ABS_T, L>(px[j], v, cth, T, sc, tag, rec[j]);
            if (!FULL) {
                // units past the band's end are padding: their state may be stepped freely, only
                // their events must be suppressed
                fl.a &= active[j];
                fl.b &= active[j];
                fl.c &= active[j];
            }
            mrec[j] = fl.a | fl.c;
            nrec += (uint32_t)__popcll(mrec[j]);
            nev += (uint32_t)__popcll(fl.a) + (uint32_t)__popcll(fl.b) + (uint32_t)__popcll(fl.c);
        }
        // ---------------- wave-level ordered compaction into the frame's segment ----------------
        // records of lower lanes (mbcnt over the record masks), then the lane's own earlier units
        uint32_t pos = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j)
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mrec[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mrec[j], pos));
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const bool has = L::lane(mrec[j]);
            if (has) lean_store_rec<ABS_T, ADDER_NT_RECSTORE != 0 && (NB_MAX > 1u)>(seg, pos * lean_rec_bytes(ABS_T), rec[j]);
            pos += has ? 1u : 0u;
        }
        wt = lane == i ? (nev | (nrec << 16)) : wt;
        vin_w = next_w;
        if (LOG) {
            seg += nrec * lean_rec_bytes(ABS_T);  // the next frame's run follows this one's
        } else {
            seg += frame_stride_u;
            if (++ridx == chunk_u) {  // uniform; never taken without rotation (a launch stays inside its chunk)
                ridx = 0u;
                seg -= wrap_bytes;
            }
        }
    }

    // ---------------- per-frame segment totals ----------------
    uint32_t wofs = 0u;
    if (LOG) {  // where each frame's run starts inside the region: an exclusive scan of the record counts over the frames
        const uint32_t nr = wt >> 16;
        const uint32_t incl = wave_inclusive_scan_dpp(nr);
        wofs = log_cur0 + incl - nr;
        if (lane == kWave - 1u) *wcur_p = log_cur0 + incl;
    }
    if (lane < nb) {
        uint32_t s = slot0 + lane;
        s = s >= slots_u ? s - slots_u : s;
        gstore<uint32_t>(uniform_ptr(b->wtot_ring), (s * num_waves_u + sgw) * 4u, wt);
        if (LOG) gstore<uint32_t>(uniform_ptr(b->wofs_ring), (s * num_waves_u + sgw) * 4u, wofs);
    }

    // ---------------- state back to HBM ----------------
    {
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N], lfv[N];
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            hdrv[j] = lean_hdr(px[j]);
            iv[j] = px[j].integ;
            dv[j] = px[j].dt;
            bv[j] = px[j].bdt;
            lfv[j] = px[j].lastf;
        }
        constexpr bool NTS = ADDER_NT_STATE != 0 && NB_MAX > 1u;  // blocked launches: the state comes back a chunk later
        store_vec<NTS>(a.hdr, u0, hdrv);
        store_vec<NTS>(a.integ0, u0, iv);
        store_vec<NTS>(a.dt0, u0, dv);
        store_vec<NTS>(a.bdt0, u0, bv);
        if (ABS_T) store_vec<NTS>(a.lastf, u0, lfv);
        if (a.running) {  // side plane (the host keeps nb == 1 while it is enabled)
#pragma unroll
            for (uint32_t j = 0; j < N; ++j)
                if (L::lane(active[j] & px[j].has0))
                    a.running[u0 + j] = (uint8_t)frame_value_u8(lean_bd_from_thr(f32_to_bits(px[j].thr)),
                                                                f32_as_u32(px[j].bdt), (double)sc.ref_time);
        }
    }
}

// NB_MAX: the most frames a launch of this kernel steps (1: no LDS is touched, lds_in may be null)
template <bool ABS_T, uint32_t NB_MAX, bool LOG = false>
__device__ __forceinline__ void lean_run_segment(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb,
                                                 uint32_t u0, uint32_t gw, uint32_t lane, const LeanRaw &raw,
                                                 uint8_t *lds_in) {
    const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
    if (full)
        lean_frames<ABS_T, true, NB_MAX, LOG>(b, a, nb, u0, gw, lane, raw, lds_in);
    else
        lean_frames<ABS_T, false, NB_MAX, LOG>(b, a, nb, u0, gw, lane, raw, lds_in);
}

template <bool ABS_T, bool LOG>
__global__ __launch_bounds__(kBlockThreads, kLeanWavesPerSimd) void adder_lean_kernel(
    const BatchArgs *__restrict__ b, uint32_t f, uint32_t nb) {
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kWavesPerBlock][kMaxFramesPerLaunch * kWaveUnits];
    timeline_mark(b, 0u, f, false);
    // The grid may be smaller than the number of segments (a few workgroups per CU that walk the segments): this
    // kernel is bound by instruction issue and the expansion of the chunk before by memory, so the two are meant
    // to be resident side by side -- two grids that each fill the chip would run one after the other.  Nothing
    // below is shared between waves, so the walk needs no barrier.
    for (uint32_t gw = blockIdx.x * kWavesPerBlock + tid / kWave; gw < a.num_waves; gw += gridDim.x * kWavesPerBlock) {
        const uint32_t u0 = gw * kWaveUnits + lane * kUnitsPerLane;
        const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
        LeanRaw raw;
        lean_load<ABS_T, ADDER_NT_STATE != 0>(a, u0, full, raw);
        lean_run_segment<ABS_T, kMaxFramesPerLaunch, LOG>(b, a, nb, u0, gw, lane, raw, s_in[tid / kWave]);
    }
    timeline_mark(b, 0u, f, true);
}

// ------------------------------------------------------------------------------------------
// K1, lean runs -- `adder_lr_kernel`: the lean regime at crf 0 in DeltaT (the headline: BASELINE configs 2-4).  Under
// the constant-run conditions (adder_pixel.hpp) a unit is {base_val, rho, popped}: the step is compares and a counter
// (lr_step), the parked record carries (base_val, rho) of a flushed root and the expansion works event A out
// (lr_decode8) -- the divisions moved from a sixth of the frame kernel's lanes to the expansion's dense ones.  Loads
// the header and delta_t planes only (rho = delta_t / T), stores the four level-0 planes in their resident form
// (lr_pack), so any other kernel may run next.  Same slots, scan and offsets as adder_lean_kernel.
// ------------------------------------------------------------------------------------------
#ifndef ADDER_LR_WAVES_PER_SIMD
#define ADDER_LR_WAVES_PER_SIMD 8
#endif
#ifndef ADDER_LR_IN_FRAMES
#define ADDER_LR_IN_FRAMES 32
#endif
#ifndef ADDER_LR_QUIET_GROUPS
#define ADDER_LR_QUIET_GROUPS 1  // (0: A/B build without the group form of the quiet path)
#endif
// input frames in the wave's LDS slice (two groups of half as many, one being stepped, one on its way): the step needs ~46
// registers, so the slice decides the occupancy -- 32 frames (4 KB per wave, 16 KB per workgroup) leave room for 8 waves
// per SIMD
constexpr uint32_t kLrInFrames = ADDER_LR_IN_FRAMES;
template <bool FULL, bool ABS_T>
__device__ __forceinline__ void lr_frames(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb, uint32_t u0,
                                          uint32_t gw, uint32_t lane, uint8_t *lds_in, uint8_t *lds_base, bool lazy) {
    constexpr uint32_t N = kUnitsPerLane;
    constexpr uint32_t NB_MAX = kMaxFramesPerLaunch;
    using L = WaveLanes;
    const float T = a.sc.time_spanned;
    LrPxT<L> px[N];
    const uint32_t sgw = __builtin_amdgcn_readfirstlane(gw);
    const uint32_t slots_u = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t num_waves_u = __builtin_amdgcn_readfirstlane(a.num_waves);
    const uint32_t n_units_u = __builtin_amdgcn_readfirstlane(a.n_units);
    const uint32_t f0 = __builtin_amdgcn_readfirstlane(a.frame_idx);
    const uint32_t slot0 = __builtin_amdgcn_readfirstlane(f0 % slots_u);
    const uint32_t park_bytes_u = __builtin_amdgcn_readfirstlane(b->park_bytes);
    const uint32_t chunk_u = __builtin_amdgcn_readfirstlane(b->chunk);
    const ParkLayout lay = park_layout_u(b);
    const uint32_t frame_stride_u = lay.frame_stride;
    uint8_t *seg = uniform_ptr(b->park_ring) + park_offset(slot0, sgw, chunk_u, num_waves_u, park_bytes_u, lay);
    // rotated frame slots (ParkLayout) wrap once per chunk at most: the launch's frame index at which that happens
    const uint32_t ridx0 = __builtin_amdgcn_readfirstlane((slot0 % chunk_u + (sgw >> lay.rot_shift)) & lay.rot_mask);
    const uint32_t wrap_at = chunk_u - 1u - ridx0;  // (after this frame of the launch the walk returns to the chunk's first slot)
    const uint32_t wrap_bytes = chunk_u * frame_stride_u;
    // The launch's input bytes go through the wave's LDS slice in groups of kLrGroup frames, ONE GROUP AHEAD: the slice
    // holds two groups, the loads of group g + 1 are issued when group g starts and waited for when it ends -- a wave never
    // sits out a memory round trip inside the loop (the waves of a CU start together and met those waits together).  The
    // first two groups are issued before the state's planes are even asked for.
    using InT = typename VecOf<uint8_t, N>::type;
    InT *const in_lds = reinterpret_cast<InT *>(lds_in) + lane;  // [frame % kLrInFrames][lane]
    const uint8_t *const fr0 = uniform_ptr(b->frames) + (size_t)f0 * n_units_u;
    constexpr uint32_t kLrGroup = kLrInFrames / 2u;
    static_assert(kWaveUnits == 128u && kLrGroup % 8u == 0u && NB_MAX % kLrInFrames == 0u, "eight frames of one segment per instruction");
    const bool direct = ADDER_LDS_DIRECT_INPUT != 0 && FULL &&
                        __builtin_amdgcn_readfirstlane(((n_units_u | (uint32_t)(uintptr_t)fr0) & 15u) == 0u);
    auto stage_issue = [&](uint32_t k0) {  // frames [k0, k0 + kLrGroup) of the launch -> their half of the slice
        const uint32_t half = (k0 / kLrGroup) & 1u;
        if (direct) {
            const uint8_t *const seg_in = fr0 + (size_t)sgw * kWaveUnits + (lane & 7u) * 16u;
#pragma unroll
            for (uint32_t g = 0; g < kLrGroup / 8u; ++g) {
                uint32_t k = k0 + g * 8u + (lane >> 3);
                k = k < nb ? k : nb - 1u;
                __builtin_amdgcn_global_load_lds((const ADDER_GLOBAL void *)(seg_in + (size_t)k * n_units_u),
                                                 (__attribute__((address_space(3))) void *)(lds_in + (half * (kLrGroup / 8u) + g) * 1024u), 16, 0,
                                                 ADDER_NT_INPUT ? 2 : 0);
            }
        } else {  // (ragged or unaligned planes: through registers, waited for at once)
#pragma unroll 1
            for (uint32_t q0 = 0; q0 < kLrGroup; q0 += 8u) {
                uint32_t vin8[8];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) {
                    const uint32_t k = k0 + q0 + q;
                    const uint32_t kk = k < nb ? k : nb - 1u;
                    vin8[q] = load_input(fr0 + (size_t)kk * n_units_u, u0, FULL ? 0xffffffffu : n_units_u);
                }
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) in_lds[(half * kLrGroup + q0 + q) * kWave] = (InT)vin8[q];
            }
        }
    };
    // group g starts: its bytes have landed (vmcnt(0): nothing else inside the frame loop waits on memory), group g + 1 leaves
    auto stage = [&](uint32_t i) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        if (i != 0u && i + kLrGroup < nb) stage_issue(i + kLrGroup);
    };
    stage_issue(0u);
    if (kLrGroup < nb) stage_issue(kLrGroup);
    uint32_t lq[N];  // AbsoluteT: last_fired_t / T (lr_step_lq)
    const uint32_t frame0 = ABS_T ? __builtin_amdgcn_readfirstlane((uint32_t)fdiv(a.sc.running_t, T)) : 0u;
    constexpr uint32_t REC = ABS_T ? 12u : 8u;
    {
        uint32_t hdrv[N];
        float dv[N], lfv[N];
        load_vec<ADDER_NT_STATE != 0>(a.hdr, u0, hdrv);
        load_vec<ADDER_NT_STATE != 0>(a.dt0, u0, dv);
        if (ABS_T) load_vec<ADDER_NT_STATE != 0>(a.lastf, u0, lfv);
        bool all_ok = true;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            bool ok;
            px[j] = lr_unpack<L>(hdrv[j], dv[j], T, ok);
            lq[j] = ABS_T ? (uint32_t)fdiv(lfv[j], T) : 0u;
            all_ok = all_ok && (ok || (!FULL && u0 + j >= n_units_u));
        }
        if (!all_ok) raise(a.status, kStatusLeanRuns);
    }
    // The loop is bound by instruction issue: a SIMD issues at most one vector and one scalar instruction per four
    // cycles (from different waves), so a frame costs about max(vector, scalar instructions) x 4 cycles per wave and
    // SIMD.  Four versions -- 62 scalar + 45 vector instructions per frame, 38 + 53, 54 + 31, 43 + 48 -- took 110, 107, 110
    // and 103 us per 64 frames: whichever side was lightened, the other one bound.  So BOTH sides are kept short: the masks
    // are scalar (WaveLanes) and popped_dtm is no mask of its own (LrPxT); the record word is one byte permute of the
    // previous and the current input words plus the unit -- which events it stands for is worked out by the expansion
    // (lr_decode8); frames go in pairs (the loop's own scalar work and the select into wt halve).
    uint32_t wt = 0u;  // lane i: {events | records << 16} of the launch's i-th frame
    uint64_t active[N], nz_old[N];
    uint32_t prev_w = 0u;  // the units' base_vals as an input word
    uint32_t sel[N];       // v_perm_b32 selectors in vector registers (as scalar constants they are set again per use)
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        active[j] = FULL ? ~0ull : L::from(u0 + j < n_units_u);
        nz_old[j] = L::from(px[j].base != 0u);
        prev_w |= px[j].base << (8u * j);
        // bytes {0, prev_w[j], vin_w[j], 0}: selectors 0-3 = bytes of the second operand, 4-7 of the first, 12 = 0
        asm volatile("v_mov_b32 %0, %1" : "=v"(sel[j]) : "s"(0x0c00000cu | (j << 8) | ((4u + j) << 16)));
    }
    auto frame = [&](uint32_t i, uint32_t &k_lane) -> uint32_t {  // -> events | records << 16 of the segment's frame i (uniform)
        const uint32_t vin_w = (uint32_t)in_lds[(i % kLrInFrames) * kWave];
        uint32_t w0[N], w1[N], w8[N];
        uint64_t mrec[N];
        uint32_t nev = 0u, nrec = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const uint32_t pair = __builtin_amdgcn_perm(vin_w, prev_w, sel[j]);
            LeanFlagsT<L> fl = lr_step<L>(px[j], (vin_w >> (8 * j)) & 0xffu, pair, lane * N + j, nz_old[j], w0[j], w8[j]);
            w1[j] = 0u;
            if (ABS_T) lr_step_lq<L>(lq[j], fl, frame0 + i, w1[j]);
            if (!FULL) {  // padding units: stepped freely, no events
                fl.a &= active[j];
                fl.b &= active[j];
                fl.c &= active[j];
            }
            mrec[j] = fl.a | fl.c;
            {   // k_lane += a + b + c (the carry-in of v_addc_co_u32 is a lane mask)
                uint64_t co;
                asm volatile("v_addc_co_u32 %0, %1, %0, 0, %2" : "+v"(k_lane), "=s"(co) : "s"(fl.a));
                asm volatile("v_addc_co_u32 %0, %1, %0, 0, %2" : "+v"(k_lane), "=s"(co) : "s"(fl.b));
                asm volatile("v_addc_co_u32 %0, %1, %0, 0, %2" : "+v"(k_lane), "=s"(co) : "s"(fl.c));
            }
        }
        prev_w = vin_w;
        uint32_t pos = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j)
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mrec[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mrec[j], pos));
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const bool has = L::lane(mrec[j]);
            if (has) {
                if (ABS_T) {
                    struct R12 { uint32_t a, b, c; };
                    gstore(seg, pos * REC, R12{w0[j], w1[j], w8[j]});
                } else {
                    gstore(seg, pos * REC, make_uint2(w0[j], w8[j]));
                }
            }
            pos += has ? 1u : 0u;
        }
        seg += frame_stride_u;
        if (__builtin_expect(i == wrap_at, 0)) seg -= wrap_bytes;  // (a branch, not selects: taken once per chunk at most)
        return (uint32_t)__builtin_amdgcn_readlane((int)pos, kWave - 1) << 16;  // (the records parked: the last lane's offset)
    };
    static_assert(kLrGroup % 2u == 0u && N <= 4u, "pairs of frames never straddle a staging group; an input word holds the units' bytes");
    // QUIET GROUPS (lr_quiet_run): a staged group ALL of whose bytes equal the units' base_vals -- static content -- is decided
    // at once: no flush, no event, no record, every run grows by the group.  The units' base_vals go to LDS as one more
    // frame row and every lane compares its 16-byte pieces of the group's rows with its piece of that row (kLrGroup / 8 + 1
    // LDS reads, an OR of XORs, one ballot -- against 16 x 80 instructions of stepping).  The test is made only where the
    // frame before left no record in the segment (and at the launch's start), so waves of busy content never pay for it.
    static_assert(kLrGroup == kQuietGroup, "the group the sim's lr block decides at once");
    auto group_quiet = [&](uint32_t i) -> bool {  // uniform
        const InT bw = (InT)prev_w;
        __builtin_memcpy(lds_base + lane * (uint32_t)sizeof(InT), &bw, sizeof(InT));
        const uint8_t *const g = lds_in + ((i / kLrGroup) & 1u) * (kLrGroup * kWaveUnits) + lane * 16u;
        uint4 r;
        __builtin_memcpy(&r, (const uint8_t *)__builtin_assume_aligned(lds_base + (lane & 7u) * 16u, 16), 16);
        uint32_t diff = 0u;
#pragma unroll
        for (uint32_t q = 0; q < kLrGroup / 8u; ++q) {
            uint4 x;
            __builtin_memcpy(&x, (const uint8_t *)__builtin_assume_aligned(g + q * 1024u, 16), 16);
            diff |= (x.x ^ r.x) | (x.y ^ r.y) | (x.z ^ r.z) | (x.w ^ r.w);
        }
        return __builtin_amdgcn_ballot_w64(diff != 0u) == 0ull;
    };
    // The frame loop below is bound by instruction issue and sensitive to what is live across it (a first version with the
    // group test inside it cost the busy headline 18 %: 93 -> 110 us per 60 frames): the quiet groups are a loop of their own
    // IN FRONT of it, and the frame loop runs one staged group per trip of the outer loop, so that a wave whose content
    // calms down goes back to the group test at the next group (only after a frame that parked no record).
    uint32_t i = 0u;
    uint32_t last_recs = 0u;  // records the segment parked in the last frame stepped (uniform)
    while (i < nb) {
        stage(i);  // (i is a multiple of kLrGroup here)
        if (ADDER_LR_QUIET_GROUPS && last_recs == 0u && group_quiet(i)) {
            const uint32_t gn = nb - i < kLrGroup ? nb - i : kLrGroup;  // (a short last group: its other rows repeat the last frame)
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) lr_quiet_run<L>(px[j], gn);
            seg += gn * frame_stride_u;
            if (wrap_at - i < gn) seg -= wrap_bytes;  // (unsigned: wrap_at lies in [i, i + gn))
            i += gn;  // (wt of these frames stays 0)
            continue;
        }
        // (REENTER: one staged group per trip; otherwise the rest of the launch, staging as it goes)
        const uint32_t i_first = i;
        const uint32_t i_end = 1 ? (i + kLrGroup < nb ? i + kLrGroup : nb) : nb;
#pragma clang loop unroll(disable)
        for (; i + 2u <= i_end; i += 2u) {
            if (!1 && (i % kLrGroup) == 0u && i != i_first) stage(i);
            uint32_t k0 = 0u, k1 = 0u;
            uint32_t t0 = frame(i, k0);
            uint32_t t1 = frame(i + 1u, k1);
            if (1) last_recs = t1 >> 16;
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan_dpp(k0 | (k1 << 16)), kWave - 1);
            t0 |= tot & 0xffffu;
            t1 |= tot >> 16;
            wt = lane == i ? t0 : lane == i + 1u ? t1 : wt;
        }
        if (i < i_end) {  // (the launch's last frame, odd launches only)
            if (!1 && (i % kLrGroup) == 0u && i != i_first) stage(i);
            uint32_t k0 = 0u;
            uint32_t t0 = frame(i, k0);
            t0 |= (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan_dpp(k0), kWave - 1);
            wt = lane == i ? t0 : wt;
            i += 1u;
        }
    }
    if (lane < nb) {
        uint32_t s = slot0 + lane;
        s = s >= slots_u ? s - slots_u : s;
        gstore<uint32_t>(uniform_ptr(b->wtot_ring), (s * num_waves_u + sgw) * 4u, wt);
    }
    report_run_max(b, px[0].rho > px[1].rho ? px[0].rho : px[1].rho, lane);
    constexpr bool NTS_ = ADDER_NT_STATE != 0;
    if (lazy) {  // another launch of this batch follows: only what lr_unpack reads (header, delta_t, last_fired_t)
        uint32_t hdrv[N];
        float dv[N], lfv[N];
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            hdrv[j] = hdr_make(px[j].base, 0u, px[j].rho != 0u ? 1u : 0u, px[j].base != 0u);
            dv[j] = px[j].base != 0u ? fmul((float)px[j].rho, T) : 0.0f;
            lfv[j] = ABS_T ? fmul((float)lq[j], T) : 0.0f;
        }
        store_vec<NTS_>(a.hdr, u0, hdrv);
        store_vec<NTS_>(a.dt0, u0, dv);
        if (ABS_T) store_vec<NTS_>(a.lastf, u0, lfv);
        return;
    }
    {   // state back to HBM in its resident form
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N];
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) hdrv[j] = lr_pack<L>(px[j], T, iv[j], dv[j], bv[j]);
        constexpr bool NTS = ADDER_NT_STATE != 0;
        store_vec<NTS>(a.hdr, u0, hdrv);
        store_vec<NTS>(a.integ0, u0, iv);
        store_vec<NTS>(a.dt0, u0, dv);
        store_vec<NTS>(a.bdt0, u0, bv);
        if (ABS_T) {
            float lfv[N];
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) lfv[j] = fmul((float)lq[j], T);
            store_vec<NTS>(a.lastf, u0, lfv);
        }
    }
}

template <bool ABS_T>
__global__ __launch_bounds__(kBlockThreads, ADDER_LR_WAVES_PER_SIMD) void adder_lr_kernel(const BatchArgs *__restrict__ b,
                                                                                         uint32_t f, uint32_t nb, uint32_t lazy) {
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kWavesPerBlock][kLrInFrames * kWaveUnits];
    __shared__ __attribute__((aligned(16))) uint8_t s_base[kWavesPerBlock][kWaveUnits];  // the units' base_vals as a frame row (quiet groups)
    timeline_mark(b, 0u, f, false);
    chain_zero(b, f, nb);  // (adder_scan_kernel CHAIN)
    for (uint32_t gw = blockIdx.x * kWavesPerBlock + tid / kWave; gw < a.num_waves; gw += gridDim.x * kWavesPerBlock) {
        const uint32_t u0 = gw * kWaveUnits + lane * kUnitsPerLane;
        const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
        if (full) lr_frames<true, ABS_T>(b, a, nb, u0, gw, lane, s_in[tid / kWave], s_base[tid / kWave], lazy != 0u);
        else lr_frames<false, ABS_T>(b, a, nb, u0, gw, lane, s_in[tid / kWave], s_base[tid / kWave], lazy != 0u);
    }
    timeline_mark(b, 0u, f, true);
}

// Lean K1 at temporal depth 1 (the per-frame `consume` contract; HBM-bound): a wave takes kLean1Segs
// consecutive segments and issues the loads of all of them before it steps the first, so the later
// segments' memory round trips hide under the first one's step (with one segment per wave all
// resident waves load, then all compute, then all store, in lock step).
#ifndef ADDER_LEAN1_SEGS
#define ADDER_LEAN1_SEGS 4
#endif
#ifndef ADDER_LEAN1_WAVES
#define ADDER_LEAN1_WAVES 4
#endif
constexpr uint32_t kLean1Segs = ADDER_LEAN1_SEGS;
template <bool ABS_T>
__global__ __launch_bounds__(kBlockThreads, ADDER_LEAN1_WAVES) void adder_lean1_kernel(const BatchArgs *__restrict__ b,
                                                                                      uint32_t f) {
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t gw0 = (blockIdx.x * kWavesPerBlock + tid / kWave) * kLean1Segs;
    if (gw0 >= a.num_waves) return;
    LeanRaw raw[kLean1Segs];
#pragma unroll
    for (uint32_t s = 0; s < kLean1Segs; ++s) {
        const uint32_t gw = gw0 + s;
        const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
        lean_load<ABS_T>(a, gw * kWaveUnits + lane * kUnitsPerLane, full, raw[s]);
    }
#pragma unroll
    for (uint32_t s = 0; s < kLean1Segs; ++s) {
        const uint32_t gw = gw0 + s;
        lean_run_segment<ABS_T, 1u>(b, a, 1u, gw * kWaveUnits + lane * kUnitsPerLane, gw, lane, raw[s], nullptr);
    }
}

// Lean K1 at temporal depth 1 with 16-byte accesses: one lane = 4 consecutive units, so every state access of
// the wave is a global_load/store_dwordx4 over 1 KiB of one plane (8-byte accesses reach 0.54-0.70 of the
// 16-byte rate on this memory system) and the input is one dword per lane.  A wave then covers a PAIR of
// segments (lanes 0-31 / 32-63): the records are compacted per half into the two segments' scratch slots, so
// scan and expansion see exactly what the 2-unit kernels leave.  kLean1wPairs pairs per wave, all loads first.
#if ADDER_UNITS_PER_LANE == 2
#ifndef ADDER_LEAN1W_PAIRS
#define ADDER_LEAN1W_PAIRS 2
#endif
#ifndef ADDER_LEAN1W_WAVES
#define ADDER_LEAN1W_WAVES 4
#endif
constexpr uint32_t kLean1wPairs = ADDER_LEAN1W_PAIRS;
constexpr uint32_t kWideUnits = 4;
struct WideRaw {
    uint4 hdr;
    float4 iv, dv, bv, lfv;
    uint32_t vin;
};

// All of a pair's loads, no control flow: the input dword of a lane that straddles the band's end is read
// from the band's last four bytes and shifted down (the bytes past the end belong to padding units, whose
// events are suppressed; the launcher keeps bands below four units on the 2-unit kernel).
template <bool ABS_T>
__device__ __forceinline__ void wide_load_state(const Lean1wArgs &st, uint32_t u0, WideRaw &r) {
    r.hdr = gload<uint4>(st.hdr, u0 * 4u);
    r.iv = gload<float4>(st.integ0, u0 * 4u);
    r.dv = gload<float4>(st.dt0, u0 * 4u);
    r.bv = gload<float4>(st.bdt0, u0 * 4u);
    if (ABS_T) r.lfv = gload<float4>(st.lastf, u0 * 4u);
    else r.lfv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
__device__ __forceinline__ void wide_load_input(const uint8_t *frame, uint32_t n_units, uint32_t u0, WideRaw &r) {
    const uint32_t ua = min(u0, n_units - kWideUnits);
    r.vin = (ADDER_NT_INPUT ? gload_nt<uint32_t>(frame, ua) : gload<uint32_t>(frame, ua)) >> (8u * min(u0 - ua, 3u));
}

template <bool ABS_T, bool FULL>
__device__ __forceinline__ void wide_step_pair(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t gw0,
                                               uint32_t lane, const WideRaw &raw) {
    constexpr uint32_t N = kWideUnits;
    using L = WaveLanes;
    const uint32_t u0 = gw0 * kWaveUnits + lane * N;
    const uint32_t hdrv[N] = {raw.hdr.x, raw.hdr.y, raw.hdr.z, raw.hdr.w};
    const float iv[N] = {raw.iv.x, raw.iv.y, raw.iv.z, raw.iv.w}, dv[N] = {raw.dv.x, raw.dv.y, raw.dv.z, raw.dv.w};
    const float bv[N] = {raw.bv.x, raw.bv.y, raw.bv.z, raw.bv.w}, lv[N] = {raw.lfv.x, raw.lfv.y, raw.lfv.z, raw.lfv.w};
    LeanPxT<L> px[N];
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) px[j] = lean_unpack<L>(hdrv[j], iv[j], dv[j], bv[j], lv[j]);
    const StepConsts sc = a.sc;  // (running_t and cth are this frame's: frame_args)
    const float T = sc.time_spanned;
    const uint32_t cth = __builtin_amdgcn_readfirstlane(sc.cth);
    const uint32_t n_units_u = __builtin_amdgcn_readfirstlane(a.n_units);
    const bool upper = lane >= 32u;

    LeanRec rec[N];
    uint64_t mrec[N], active[N];
    uint32_t nev_lo = 0u, nev_hi = 0u, nrec_lo = 0u, nrec_hi = 0u;
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        active[j] = FULL ? ~0ull : L::from(u0 + j < n_units_u);
        const uint32_t v = (raw.vin >> (8 * j)) & 0xffu;
        const uint32_t tag = ((lane & 31u) * N + j) << kLeanUnitShift;
        LeanFlagsT<L> fl = lean_step<ABS_T, L>(px[j], v, cth, T, sc, tag, rec[j]);
        if (!FULL) {
            fl.a &= active[j];
            fl.b &= active[j];
            fl.c &= active[j];
        }
        mrec[j] = fl.a | fl.c;
        nrec_lo += (uint32_t)__popc((uint32_t)mrec[j]);
        nrec_hi += (uint32_t)__popc((uint32_t)(mrec[j] >> 32));
        nev_lo += (uint32_t)__popc((uint32_t)fl.a) + (uint32_t)__popc((uint32_t)fl.b) + (uint32_t)__popc((uint32_t)fl.c);
        nev_hi += (uint32_t)__popc((uint32_t)(fl.a >> 32)) + (uint32_t)__popc((uint32_t)(fl.b >> 32)) +
                  (uint32_t)__popc((uint32_t)(fl.c >> 32));
    }
    // ordered compaction per half: records of the lower lanes of the lane's own half, then its earlier units
    uint32_t pos = 0u;
#pragma unroll
    for (uint32_t j = 0; j < N; ++j)
        pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mrec[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mrec[j], pos));
    pos -= upper ? nrec_lo : 0u;  // (for lanes >= 32 mbcnt_lo counted the whole lower half)
    const uint32_t slot = __builtin_amdgcn_readfirstlane(a.frame_idx % b->slots);
    const uint32_t chunk_u = __builtin_amdgcn_readfirstlane(b->chunk);
    const uint32_t park_bytes_u = __builtin_amdgcn_readfirstlane(b->park_bytes);
    const ParkLayout lay = park_layout_u(b);  // (a pair never straddles a group: gw0 is even, groups hold 2^k >= 2 ... or 1)
    const uint32_t seg_stride_u = __builtin_amdgcn_readfirstlane(
        (uint32_t)(park_offset(slot, gw0 + 1u, chunk_u, a.num_waves, park_bytes_u, lay) -
                   park_offset(slot, gw0, chunk_u, a.num_waves, park_bytes_u, lay)));
    uint8_t *const seg = uniform_ptr(b->park_ring) +
                         park_offset(slot, __builtin_amdgcn_readfirstlane(gw0), chunk_u,
                                     __builtin_amdgcn_readfirstlane(a.num_waves), park_bytes_u, lay);
    uint32_t off = (upper ? seg_stride_u : 0u) + pos * lean_rec_bytes(ABS_T);  // (the upper half is segment gw0 + 1)
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        const bool has = L::lane(mrec[j]);
        if (has) lean_store_rec<ABS_T, false>(seg, off, rec[j]);
        off += has ? lean_rec_bytes(ABS_T) : 0u;
    }
    if (lane == 0u)
        gstore<uint2>(uniform_ptr(a.wtot), __builtin_amdgcn_readfirstlane(gw0) * 4u,
                      make_uint2(nev_lo | (nrec_lo << 16), nev_hi | (nrec_hi << 16)));

    // ---------------- state back to HBM ----------------
    uint32_t ho[N];
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) ho[j] = lean_hdr(px[j]);
    gstore<uint4>(a.hdr, u0 * 4u, make_uint4(ho[0], ho[1], ho[2], ho[3]));
    gstore<float4>(a.integ0, u0 * 4u, make_float4(px[0].integ, px[1].integ, px[2].integ, px[3].integ));
    gstore<float4>(a.dt0, u0 * 4u, make_float4(px[0].dt, px[1].dt, px[2].dt, px[3].dt));
    gstore<float4>(a.bdt0, u0 * 4u, make_float4(px[0].bdt, px[1].bdt, px[2].bdt, px[3].bdt));
    if (ABS_T) gstore<float4>(a.lastf, u0 * 4u, make_float4(px[0].lastf, px[1].lastf, px[2].lastf, px[3].lastf));
    if (a.running) {
#pragma unroll
        for (uint32_t j = 0; j < N; ++j)
            if (L::lane(active[j] & px[j].has0))
                a.running[u0 + j] = (uint8_t)frame_value_u8(lean_bd_from_thr(f32_to_bits(px[j].thr)),
                                                            f32_as_u32(px[j].bdt), (double)sc.ref_time);
    }
}

#define ADDER_CONSTANT __attribute__((address_space(4)))
// The state planes and the band's size come as kernel arguments (constant for the life of the context, so a
// captured graph may hold them): the state loads go out one scalar round trip after the wave starts instead of
// three (kernel arguments -> batch description -> its fields).
template <bool ABS_T>
__global__ __launch_bounds__(kBlockThreads, ADDER_LEAN1W_WAVES) void adder_lean1w_kernel(const BatchArgs *__restrict__ b,
                                                                                        uint32_t f, Lean1wArgs st) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t n_units = st.n_units;
    const uint32_t num_pairs = st.num_waves / 2u;
    const uint32_t gp0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kWavesPerBlock + tid / kWave) * kLean1wPairs);
    if (gp0 >= num_pairs) return;
    WideRaw raw[kLean1wPairs];
#pragma unroll
    for (uint32_t s = 0; s < kLean1wPairs; ++s) {
        const uint32_t gw = 2u * min(gp0 + s, num_pairs - 1u);  // (a wave past the end re-reads the last pair)
        wide_load_state<ABS_T>(st, gw * kWaveUnits + lane * kWideUnits, raw[s]);
    }
    // the frame's row of the table as a scalar load (uploaded with the batch description before the launch)
    using TabRaw = typename RawOf<sizeof(FrameTab)>::type;
    const TabRaw tab = *reinterpret_cast<const ADDER_CONSTANT TabRaw *>((const ADDER_CONSTANT char *)(uint64_t)b->ftab +
                                                                        (size_t)f * sizeof(FrameTab));
    FrameArgs a = frame_args(b, f);
    a.sc.running_t = __uint_as_float(tab[0]);
    a.sc.running_t_u32 = f32_as_u32(a.sc.running_t);
    a.sc.cth = tab[1];
    const uint8_t *const frame = uniform_ptr(b->frames) + (size_t)f * n_units;
#pragma unroll
    for (uint32_t s = 0; s < kLean1wPairs; ++s) {
        const uint32_t gw = 2u * min(gp0 + s, num_pairs - 1u);
        wide_load_input(frame, n_units, gw * kWaveUnits + lane * kWideUnits, raw[s]);
    }
    // All pairs' loads first.  Measured on top of this (each one slower or equal, A/B inside one run): a
    // scheduling barrier that holds the vector loads until the scalar prologue has landed (+0.3 ... 0.6 us of
    // 14), one that keeps every load ahead of the first consumer (+-0), waiting for the next pair's loads
    // before this pair's stores so that no wait falls behind uncounted stores (+0.4), the second pair's loads
    // issued when the first pair's have landed (+-0); 1 / 3 / 4 pairs per wave, 3 / 5 waves per SIMD.
#pragma unroll
    for (uint32_t s = 0; s < kLean1wPairs; ++s) {
        if (gp0 + s >= num_pairs) break;
        const uint32_t gw = 2u * (gp0 + s);
        const bool full = gw * kWaveUnits + 2u * kWaveUnits <= n_units;
        if (full) wide_step_pair<ABS_T, true>(b, a, gw, lane, raw[s]);
        else wide_step_pair<ABS_T, false>(b, a, gw, lane, raw[s]);
    }
}
#endif  // ADDER_UNITS_PER_LANE == 2

// ------------------------------------------------------------------------------------------
// K1, generic variants (Normal mode, or delta_t_max > time_spanned): any arena depth.  Every unit runs the
// four phases of the generic step (adder_pixel.hpp): gen_root on the register-resident level 0 (branch-
// free; it also yields the unit's event count, so a DPP prefix scan over the wave can place the events
// before they are produced), gen_emit (parks 8-byte records {t, d | unit << 8 | final offset << 16} in
// final order), gen_walk over the deeper levels and gen_pop.  Levels 1..kGenLdsLevels of the wave's units
// live in LDS for the whole launch (temporal blocking: fetched from the deep planes once, written back
// once); deeper ones -- long static runs in Normal mode -- are accessed in the deep planes directly.
// ------------------------------------------------------------------------------------------
#ifndef ADDER_GEN_LDS_LEVELS
#define ADDER_GEN_LDS_LEVELS 4
#endif
constexpr uint32_t kGenLdsLevels = ADDER_GEN_LDS_LEVELS;

struct DeepHybrid {
    uint4 *lds;     // the unit's level-1 slot in the wave's LDS slice; level k at lds[(k - 1) * kWaveUnits]
    DeepGlobal g;   // levels beyond kGenLdsLevels
    __device__ __forceinline__ void load(uint32_t k, Node &n) const {
        if (k <= kGenLdsLevels) {
            const uint4 r = lds[(k - 1u) * kWaveUnits];
            n.integ = __uint_as_float(r.x);
            n.dt = __uint_as_float(r.y);
            n.bdt = __uint_as_float(r.z);
            n.bd = r.w;
        } else {
            g.load(k, n);
        }
    }
    __device__ __forceinline__ void store(uint32_t k, const Node &n) const {
        if (k <= kGenLdsLevels)
            lds[(k - 1u) * kWaveUnits] = make_uint4(__float_as_uint(n.integ), __float_as_uint(n.dt), __float_as_uint(n.bdt), n.bd);
        else
            g.store(k, n);
    }
};

// The record log of one segment for the frames of one launch (BatchArgs::log_cap): wave-uniform, in SGPRs.
struct SegLog {
    uint2 *region;      // the segment's region of the launch's chunk
    uint32_t cur, cap;  // records appended so far / the region's capacity
    uint32_t *wcur_p;   // where the cursor lives between the launches of a chunk
    __device__ __forceinline__ void open(const BatchArgs *__restrict__ b, uint32_t f0, uint32_t sgw, uint32_t num_waves_u) {
        const uint32_t chunk_u = __builtin_amdgcn_readfirstlane(b->chunk);
        const uint32_t slots_u = __builtin_amdgcn_readfirstlane(b->slots);
        cap = __builtin_amdgcn_readfirstlane(b->log_cap);
        const uint32_t slot0 = f0 % slots_u;
        const uint32_t cir = slot0 / chunk_u;  // (a launch never crosses a chunk boundary)
        const size_t seg_idx = (size_t)cir * num_waves_u + sgw;
        region = reinterpret_cast<uint2 *>(uniform_ptr(b->park_ring)) + seg_idx * cap;
        wcur_p = uniform_ptr(b->wcur) + seg_idx;
        cur = 0u;
        if (slot0 != cir * chunk_u) cur = __builtin_amdgcn_readfirstlane(*wcur_p);  // not the chunk's first launch
    }
    // room for n records of the next frame: where they start, or null (the bound was violated: reported, nothing stored)
    __device__ __forceinline__ uint2 *append(uint32_t n, uint32_t &start, uint32_t *status) {
        start = cur;
        if (cur + n > cap) {
            raise(status, kStatusScratch);
            return nullptr;
        }
        uint2 *p = region + cur;
        cur += n;
        return p;
    }
    __device__ __forceinline__ void close(uint32_t lane) {
        if (lane == 0u) *wcur_p = cur;
    }
};

template <bool COLLAPSE, bool ABS_T>
__device__ __forceinline__ void gen_run_segment(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb,
                                                uint32_t u0, uint32_t gw, uint32_t lane, uint4 *lds_levels) {
    constexpr uint32_t N = kUnitsPerLane;
    const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
    PxState px[N];
    uint32_t vin_w;
    {
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N], lfv[N];
        load_vec(a.hdr, u0, hdrv);
        vin_w = load_input(a.frame, u0, full ? 0xffffffffu : a.n_units);
        load_vec(a.integ0, u0, iv);
        load_vec(a.dt0, u0, dv);
        load_vec(a.bdt0, u0, bv);
        if (ABS_T) load_vec(a.lastf, u0, lfv);
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) px[j] = px_unpack(hdrv[j], iv[j], dv[j], bv[j], ABS_T ? lfv[j] : 0.0f);
    }
    StepConsts sc = a.sc;
    // c_thresh / c_increase_counter per unit once feature-driven rate control or an ROI has made them differ
    const bool perpx = a.cth_px != nullptr;  // uniform
    uint8_t cthv[N], cctrv[N];
    if (perpx) {
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            cthv[j] = a.cth_px[u0 + j];
            cctrv[j] = a.cctr_px[u0 + j];
        }
    }
    // the unit's LDS slot: [level][j][lane] (consecutive lanes -> consecutive 16-byte slots)
    DeepHybrid deep[N];
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        deep[j].lds = lds_levels + j * kWave + lane;
        deep[j].g = DeepGlobal{a.dv_integ, a.dv_dt, a.dv_bdt, a.dv_bd, a.plane_stride, (size_t)u0 + j};
        for (uint32_t k = 1; k <= kGenLdsLevels && k < px[j].m; ++k) {  // levels 1.. of the launch's first frame
            Node nk;
            deep[j].g.load(k, nk);
            deep[j].store(k, nk);
        }
    }
    if (snap_deep_wanted(b, a)) {
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) snap_deep_levels(b, a, (size_t)u0 + j, px[j].m);
    }

    const uint32_t sgw = __builtin_amdgcn_readfirstlane(gw);
    const uint8_t *const frames_u = uniform_ptr(b->frames);
    const FrameTab *const ftab_u = uniform_ptr(b->ftab);
    uint32_t *const wtot_ring_u = uniform_ptr(b->wtot_ring);
    const uint32_t slots_u = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t num_waves_u = __builtin_amdgcn_readfirstlane(a.num_waves);
    const uint32_t n_units_u = __builtin_amdgcn_readfirstlane(a.n_units);
    uint32_t slot = __builtin_amdgcn_readfirstlane(a.frame_idx % slots_u);
    bool depth_error = false;
    uint32_t *const wofs_ring_u = uniform_ptr(b->wofs_ring);
    SegLog log;
    log.open(b, __builtin_amdgcn_readfirstlane(a.frame_idx), sgw, num_waves_u);

    for (uint32_t i = 0; i < nb; ++i, slot = (slot + 1u == slots_u) ? 0u : slot + 1u) {
        const uint32_t f = a.frame_idx + i;
        uint32_t next_w = 0u;
        if (i + 1 < nb)
            next_w = load_input(frames_u + (size_t)(f + 1) * n_units_u, u0, full ? 0xffffffffu : n_units_u);
        const uint2 ft = gload<uint2>(ftab_u, f * (uint32_t)sizeof(FrameTab));
        sc.running_t = __uint_as_float(__builtin_amdgcn_readfirstlane(ft.x));
        sc.running_t_u32 = __builtin_amdgcn_readfirstlane(f32_as_u32(sc.running_t));
        sc.cth = __builtin_amdgcn_readfirstlane(ft.y);

        // ---------------- level 0 of every unit + the event counts ----------------
        GenPlan plan[N];
        uint32_t lane_cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const uint32_t v = (vin_w >> (8 * j)) & 0xffu;
            if (perpx) {
                sc.cth = cthv[j];
                c_thresh_advance(cthv[j], cctrv[j], (uint8_t)a.c_max, (uint8_t)a.c_vel, sc.time_spanned, sc.ref_time);
            }
            gen_root<COLLAPSE>(px[j], v, sc, plan[j]);
            // units past the band's end are padding: their state may be stepped freely, only
            // their events must be suppressed
            if (!(full || u0 + j < a.n_units)) {
                plan[j].count = 0u;
                plan[j].flush = false;
                plan[j].need_pop = false;
            }
            lane_cnt += plan[j].count;
        }
        // ---------------- wave-level ordered compaction into the frame's segment ----------------
        const uint32_t incl = wave_inclusive_scan_dpp(lane_cnt);
        const size_t seg_idx = (size_t)slot * num_waves_u + sgw;  // uniform
        // events of the segment (low half) = parked records (high half): one record per event, in final order,
        // appended to the segment's log of the chunk
        uint32_t run_start;
        uint2 *const seg = log.append(__builtin_amdgcn_readlane(incl, kWave - 1), run_start, a.status);  // uniform
        if (lane == kWave - 1) {
            gstore<uint32_t>(wtot_ring_u + seg_idx, 0u, incl | (incl << 16));
            gstore<uint32_t>(wofs_ring_u + seg_idx, 0u, run_start);
        }
        uint32_t off = incl - lane_cnt;  // final offset of the lane's first event inside the segment
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            if (plan[j].count != 0u && seg) {
                EmitPark em{seg, (lane * N + j) << 8, off};
                gen_emit<ABS_T>(px[j], plan[j], sc, deep[j], em);
            }
            off += plan[j].count;
        }
        // ---------------- the deeper levels ----------------
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const uint32_t v = (vin_w >> (8 * j)) & 0xffu;
            if (!gen_walk(px[j], v, plan[j], sc, deep[j])) depth_error = true;
            gen_pop(px[j], plan[j], deep[j]);
        }
        vin_w = next_w;
    }
    if (depth_error) raise(a.status, kStatusDepth);
    log.close(lane);

    // ---------------- state back to HBM ----------------
    {
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N], lfv[N];
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            hdrv[j] = px_hdr(px[j]);
            iv[j] = px[j].n0.integ;
            dv[j] = px[j].n0.dt;
            bv[j] = px[j].n0.bdt;
            lfv[j] = px[j].lastf;
            for (uint32_t k = 1; k <= kGenLdsLevels && k < px[j].m; ++k) {
                Node nk;
                deep[j].load(k, nk);
                deep[j].g.store(k, nk);
            }
        }
        store_vec(a.hdr, u0, hdrv);
        store_vec(a.integ0, u0, iv);
        store_vec(a.dt0, u0, dv);
        store_vec(a.bdt0, u0, bv);
        if (ABS_T) store_vec(a.lastf, u0, lfv);
        if (perpx) {
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) {
                a.cth_px[u0 + j] = cthv[j];
                a.cctr_px[u0 + j] = cctrv[j];
            }
        }
        if (a.running) {  // side plane (the host keeps nb == 1 while it is enabled)
#pragma unroll
            for (uint32_t j = 0; j < N; ++j)
                if (u0 + j < a.n_units && px[j].m != 0u)
                    a.running[u0 + j] = (uint8_t)frame_value_u8(px[j].n0.bd, f32_as_u32(px[j].n0.bdt),
                                                                (double)sc.ref_time);
        }
    }
}

template <bool COLLAPSE, bool ABS_T>
__global__ __launch_bounds__(kBlockThreads, 4) void adder_frame_kernel(const BatchArgs *__restrict__ b, uint32_t f,
                                                                      uint32_t nb) {
    __shared__ __attribute__((aligned(16))) uint4 s_levels[kWavesPerBlock][kGenLdsLevels * kWaveUnits];
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    // (a capped grid walks the segments, like the lean kernel; the wave's LDS slice is its own, no barrier needed)
    for (uint32_t gw = blockIdx.x * kWavesPerBlock + tid / kWave; gw < a.num_waves; gw += gridDim.x * kWavesPerBlock) {
        const uint32_t u0 = gw * kWaveUnits + lane * kUnitsPerLane;
        gen_run_segment<COLLAPSE, ABS_T>(b, a, nb, u0, gw, lane, s_levels[tid / kWave]);
    }
}

// ------------------------------------------------------------------------------------------
// K1, bounded Collapse step (Collapse with delta_t_max > time_spanned: the reference's DEFAULT mode and BASELINE
// config 5) -- adder_pixel.hpp cb_step / cb_emit / cb_pop.  Temporally blocked like the lean kernel: the root of
// every unit lives in registers and levels 1..4 in the wave's LDS slice for the whole launch, in PREFIX COORDINATES
// (a level that is visited without firing needs no update, and "which level fires" is four compares on one
// 16-byte LDS read); deeper levels -- delta_t_max far beyond 30 frames -- sit in the deep planes in the same form.
// Events leave as 8-byte records with their final in-segment offset (the generic kernel's format), appended to the
// segment's log of the chunk.  The input bytes come through LDS in groups of kCbInFrames frames.
// ------------------------------------------------------------------------------------------
#ifndef ADDER_CB_IN_FRAMES
#define ADDER_CB_IN_FRAMES 16
#endif
#ifndef ADDER_CB_WAVES_PER_SIMD
#define ADDER_CB_WAVES_PER_SIMD 4
#endif
#ifndef ADDER_CB_WAVE_LANES
#define ADDER_CB_WAVE_LANES 0
#endif
constexpr uint32_t kCbInFrames = ADDER_CB_IN_FRAMES;
#define ADDER_LDS __attribute__((address_space(3)))
using CbLevelsDev = CbLevelsT<ADDER_LDS float *>;
struct __attribute__((aligned(16))) CbWaveLds {
    float F[kWaveUnits * kCbFastLevels];  // [unit slot][level - 1]
    float Q[kWaveUnits * kCbFastLevels];
    float BT[kWaveUnits * kCbFastLevels * 2];  // {best_delta_t, threshold 2^d} pairs
    uint8_t in[kCbInFrames * kWaveUnits]; // [frame of the group][unit]
};

// the next group of input frames of one segment -> the wave's LDS slice (frames [k0, k0 + kCbInFrames) of the launch);
// cb_stage_issue only asks for them (vmcnt(0) before the first read), cb_stage_input waits as well
template <bool FULL>
__device__ __forceinline__ void cb_stage_issue(const uint8_t *fr0, uint32_t n_units_u, uint32_t sgw, uint32_t u0,
                                               uint32_t lane, uint32_t k0, uint32_t nb, uint8_t *lds_in, bool direct) {
    constexpr uint32_t N = kUnitsPerLane;
    using InT = typename VecOf<uint8_t, N>::type;
    static_assert(kWaveUnits == 128u && kCbInFrames % 8u == 0u, "eight frames of one segment per instruction");
    if (direct) {  // global_load_lds_dwordx4: lane i fetches 16 bytes of frame 8 g + i / 8, parked at M0 + 16 i
        const uint8_t *const seg_in = fr0 + (size_t)sgw * kWaveUnits + (lane & 7u) * 16u;
#pragma unroll
        for (uint32_t g = 0; g < kCbInFrames / 8u; ++g) {
            uint32_t k = k0 + g * 8u + (lane >> 3);
            k = k < nb ? k : nb - 1u;
            __builtin_amdgcn_global_load_lds((const ADDER_GLOBAL void *)(seg_in + (size_t)k * n_units_u),
                                             (__attribute__((address_space(3))) void *)(lds_in + g * 1024u), 16, 0,
                                             ADDER_NT_INPUT ? 2 : 0);
        }
    } else {  // ragged or unaligned segments: through registers, a frame at a time (rare: not unrolled)
        InT *const in_lds = reinterpret_cast<InT *>(lds_in) + lane;
#pragma unroll 1
        for (uint32_t q = 0; q < kCbInFrames; ++q) {
            const uint32_t k = k0 + q;
            const uint32_t kk = k < nb ? k : nb - 1u;
            in_lds[q * kWave] = (InT)load_input(fr0 + (size_t)kk * n_units_u, u0, FULL ? 0xffffffffu : n_units_u);
        }
    }
}
template <bool FULL>
__device__ __forceinline__ void cb_stage_input(const uint8_t *fr0, uint32_t n_units_u, uint32_t sgw, uint32_t u0,
                                               uint32_t lane, uint32_t k0, uint32_t nb, uint8_t *lds_in, bool direct) {
    cb_stage_issue<FULL>(fr0, n_units_u, sgw, u0, lane, k0, nb, lds_in, direct);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the group has landed (and so have the records stored so far)
}

#ifndef ADDER_CB_QUIET_PATH
#define ADDER_CB_QUIET_PATH 1
#endif
#ifndef ADDER_CB_QUIET_GROUPS
#define ADDER_CB_QUIET_GROUPS 1  // (0: A/B build without the group form of the quiet path)
#endif
template <bool ABS_T, bool FULL>
__device__ __forceinline__ void cb_run_segment(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb,
                                               uint32_t u0, uint32_t gw, uint32_t lane, CbWaveLds &w) {
    constexpr uint32_t N = kUnitsPerLane;
    // (booleans as wave masks -- the lean kernel's trick -- measured SLOWER here: with two units' worth of masks live
    // across the step the scalar registers spill into VGPR lanes, 258 instead of 238 VALU instructions per wave-frame)
#if ADDER_CB_WAVE_LANES
    using L = WaveLanes;
#else
    using L = ScalarLanes;
#endif
#ifdef ADDER_CB_PROFILE  // (diagnostic build: cycles per section of the wave, into the timeline buffer's spare slots; tools/probes/cb_profile.py)
    const unsigned long long cbp_wave0 = __builtin_readcyclecounter();
    unsigned long long cbp_acc[4] = {0ull, 0ull, 0ull, 0ull};  // general-frame cycles, general frames, quiet-section cycles, quiet sections
#endif
    CbPxT<L> px[N];
    uint32_t snap_m[N];  // fired levels as the header has them (the undo copy takes all of them)
    {
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N], lfv[N];
        load_vec<ADDER_NT_STATE != 0>(a.hdr, u0, hdrv);
        load_vec<ADDER_NT_STATE != 0>(a.integ0, u0, iv);
        load_vec<ADDER_NT_STATE != 0>(a.dt0, u0, dv);
        load_vec<ADDER_NT_STATE != 0>(a.bdt0, u0, bv);
        if (ABS_T) load_vec<ADDER_NT_STATE != 0>(a.lastf, u0, lfv);
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            px[j] = cb_unpack<L>(hdrv[j], iv[j], dv[j], bv[j], ABS_T ? lfv[j] : 0.0f);
            snap_m[j] = px[j].m;
            if (L::lane(px[j].popped) && px[j].m > 1u) px[j].m = 1u;  // a popped arena keeps only its root
        }
    }
    StepConsts sc = a.sc;
    const float T = sc.time_spanned;
    // levels >= 1: resident form -> prefix coordinates, once per launch
    CbLevelsDev lv[N];
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) {
        const uint32_t slot = j * kWave + lane;  // consecutive lanes -> consecutive 16-byte slots
        lv[j] = CbLevelsDev{(ADDER_LDS float *)(w.F + slot * kCbFastLevels), (ADDER_LDS float *)(w.Q + slot * kCbFastLevels),
                            (ADDER_LDS float *)(w.BT + slot * kCbFastLevels * 2u),
                            a.dv_integ, a.dv_dt, a.dv_bdt, a.dv_bd, a.plane_stride, (size_t)u0 + j};
        const DeepGlobal g{a.dv_integ, a.dv_dt, a.dv_bdt, a.dv_bd, a.plane_stride, (size_t)u0 + j};
        for (uint32_t k = 1; k < px[j].m; ++k) {
            Node nk;
            g.load(k, nk);
            lv[j].store(k, cb_level_from_node(px[j], nk));
        }
    }
    if (snap_deep_wanted(b, a)) {  // (the levels as they are in the planes: also those a popped arena no longer needs)
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) snap_deep_levels(b, a, (size_t)u0 + j, snap_m[j]);
    }

    const uint32_t sgw = __builtin_amdgcn_readfirstlane(gw);
    const uint32_t slots_u = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t num_waves_u = __builtin_amdgcn_readfirstlane(a.num_waves);
    const uint32_t n_units_u = __builtin_amdgcn_readfirstlane(a.n_units);
    const uint32_t f0 = __builtin_amdgcn_readfirstlane(a.frame_idx);
    const uint32_t slot0 = __builtin_amdgcn_readfirstlane(f0 % slots_u);
    // the launch's rows of the per-frame table: one vector load, a lane read per frame
    uint32_t tab_cth = 0u, tab_rt = 0u;
    if (lane < nb) {
        const uint2 e = gload<uint2>(uniform_ptr(b->ftab), (f0 + lane) * (uint32_t)sizeof(FrameTab));
        tab_rt = e.x;
        tab_cth = e.y;
    }
    SegLog log;
    log.open(b, f0, sgw, num_waves_u);
    const uint8_t *const fr0 = uniform_ptr(b->frames) + (size_t)f0 * n_units_u;
    const bool direct = ADDER_LDS_DIRECT_INPUT != 0 && FULL &&
                        __builtin_amdgcn_readfirstlane(((n_units_u | (uint32_t)(uintptr_t)fr0) & 15u) == 0u);
    using InT = typename VecOf<uint8_t, N>::type;
    const InT *const in_lds = reinterpret_cast<const InT *>(w.in) + lane;
    // the smallest c_thresh of every input group (lane 16 g + 15: group g; frames past the launch do not count)
    const uint32_t tab_cmin = row16_min_to_last(lane < nb ? tab_cth : 0xffu);
    uint32_t wt = 0u, wo = 0u;  // lane i: {events | records << 16} and the run's start of the launch's i-th frame
    typename L::Mask depth_error = L::from(false);

    // everything loaded so far (state, frame table, log cursor) has landed BEFORE the loop: otherwise the compiler must
    // assume at the loop header that one of those registers is still in flight and waits vmcnt(0) in EVERY frame --
    // i.e. for the acknowledgement of the previous frame's record stores
    __builtin_amdgcn_s_waitcnt(0x0f70);
    uint32_t i = 0u;
    while (i < nb) {  // quiet frames, then general frames up to the next input group, then the same again
#ifdef ADDER_CB_PROFILE
        const unsigned long long cbp_q0 = __builtin_readcyclecounter();
        const uint32_t cbp_i0 = i;
#endif
#if ADDER_CB_QUIET_PATH
    // ---------------- quiet frames (cb_quiet / cb_step_quiet, adder_pixel.hpp) ----------------
    // A wave ALL of whose units start the launch popped down to their root (or black: a d = 128 root) stays in this loop
    // for as long as every unit also passes its contrast test: only the roots integrate, nothing leaves -- the steady state
    // of static content and of lossy content away from what moves.  The first frame that is not quiet goes through the
    // general loop below, which knows nothing of this one and hands back at the next input group (every 16 frames: a wave
    // that calms down mid-launch -- a fresh clip pops everywhere at frame 30 -- is noticed there): a test inside the general loop,
    // before the step or riding on its first half, cost the busy crf-0 scene 9-13 % of this kernel (and making it only
    // after a calm frame, or with a back-off, cost more still).
    {
        bool lane_ok = true;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) lane_ok = lane_ok && px[j].m == 1u && (L::lane(px[j].popped) || px[j].thr0 == 0.0f);
        if (__builtin_amdgcn_ballot_w64(!lane_ok) == 0ull) {  // uniform
            // cb_quiet as wave masks: what does not change while the wave stays in this loop (m == 1; popped; a black
            // root's zero threshold) is a scalar-register pair per unit, a frame adds two compares per unit
            uint64_t unpopped[N];
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) unpopped[j] = __builtin_amdgcn_ballot_w64(!L::lane(px[j].popped));
            // Input staging of THIS loop is double-buffered: no unit of such a wave holds a level (m == 1 everywhere), so the
            // wave's fast-level slots are idle and take the NEXT group's bytes while this group is worked on (the quiet groups
            // below get through a group in a few hundred instructions: the wait for its bytes would be most of the time).
            // The general loop knows nothing of it: whoever leaves this loop waits for what is in flight (nothing may land in
            // the level slots once levels live there again) and puts the current group into w.in if it is not there.
            static_assert(sizeof(w.F) >= sizeof(w.in), "the level slots hold a group of input frames");
            uint8_t *const buf_b = reinterpret_cast<uint8_t *>(w.F);
            bool pref = false;   // the next group's bytes are on their way into the other buffer (uniform)
            uint32_t cur = 0u;   // where the current group is: 0 = w.in, 1 = the level slots (uniform)
            const InT *cur_in = in_lds;
            for (; i < nb;) {
                if ((i % kCbInFrames) == 0u) {
                    if (pref) cur ^= 1u;
                    else {
                        cur = 0u;
                        cb_stage_issue<FULL>(fr0, n_units_u, sgw, u0, lane, i, nb, w.in, direct);
                    }
                    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the group has landed
                    pref = 1 != 0 && i + kCbInFrames < nb;
                    if (pref) cb_stage_issue<FULL>(fr0, n_units_u, sgw, u0, lane, i + kCbInFrames, nb, cur ? w.in : buf_b, direct);
                    cur_in = reinterpret_cast<const InT *>(cur ? buf_b : w.in) + lane;
#if ADDER_CB_QUIET_GROUPS
                    // ---------------- the whole group at once (quiet_group_apply, adder_pixel.hpp) ----------------
                    // min / max / sum of every unit's bytes over the group, the number of frames before the root fires and
                    // what it has accumulated by then -- packed 16-bit operations, both units of a lane per instruction --
                    // then ONE contrast test against the group's smallest c_thresh and ONE firing test per unit.  A group
                    // in which some unit is not quiet is stepped frame by frame below, as before.
                    static_assert(N == 2u && kCbInFrames == kQuietGroup, "two units per lane in one packed register");
                    {
                        const uint32_t gn = nb - i < kCbInFrames ? nb - i : kCbInFrames;  // (uniform; a launch's last group may be short)
                        const uint32_t cth_min = __builtin_amdgcn_readlane(tab_cmin, i + kCbInFrames - 1u);
                        const uint32_t need2 = quiet_group_need(px[0].S, px[0].thr0) | (quiet_group_need(px[1].S, px[1].thr0) << 16);
                        uint32_t mn = 0x00ff00ffu, mx = 0u, P = 0u, cnt = 0u, pm = 0u;
                        for (uint32_t k = 0; k < gn; ++k) {
                            const uint32_t pk = __builtin_amdgcn_perm(0u, (uint32_t)cur_in[k * kWave], 0x0c010c00u);  // {v0, v1} as halves
                            mn = pk_min_u16(mn, pk);
                            mx = pk_max_u16(mx, pk);
                            P = pk_add_u16(P, pk);
                            const uint32_t below = pk_min_u16(pk_sub_sat_u16(need2, P), 0x00010001u);  // prefix sum < need
                            cnt = pk_add_u16(cnt, below);
                            pm = pk_mad_u16(pk, below, pm);
                        }
                        CbPxT<L> t[N];
                        uint32_t verdict[N];
                        bool any_no = false, any_slow = false;
#pragma unroll
                        for (uint32_t j = 0; j < N; ++j) {
                            QuietGroupStats g;
                            g.mn = (mn >> (16u * j)) & 0xffffu;
                            g.mx = (mx >> (16u * j)) & 0xffffu;
                            g.sum = (P >> (16u * j)) & 0xffffu;
                            g.cnt = (cnt >> (16u * j)) & 0xffffu;
                            g.pm = (pm >> (16u * j)) & 0xffffu;
                            const uint32_t row = g.cnt < gn ? g.cnt : gn - 1u;
                            g.vc = ((uint32_t)cur_in[row * kWave] >> (8u * j)) & 0xffu;
                            t[j] = px[j];
                            verdict[j] = cb_group_apply<L>(t[j], g, gn, cth_min, T);
                            any_no = any_no || verdict[j] == kQuietNo;
                            any_slow = any_slow || verdict[j] == kQuietSlow;
                        }
                        if (__builtin_amdgcn_ballot_w64(any_no) == 0ull) {  // uniform: every unit of the wave is quiet in every frame
#pragma unroll
                            for (uint32_t j = 0; j < N; ++j)
                                if (verdict[j] == kQuietDone) px[j] = t[j];
                            if (__builtin_amdgcn_ballot_w64(any_slow) != 0ull) {  // (rare: second firings, black roots that wake up)
                                for (uint32_t k = 0; k < gn; ++k) {
                                    const uint32_t vin_s = (uint32_t)cur_in[k * kWave];
#pragma unroll
                                    for (uint32_t j = 0; j < N; ++j)
                                        if (verdict[j] == kQuietSlow) cb_step_quiet<L, true>(px[j], (vin_s >> (8 * j)) & 0xffu, T);
                                }
                            }
                            i += gn;  // (wt / wo of these frames stay 0: no events, no records)
                            continue;
                        }
                    }
#endif
                }
                const uint32_t vin_q = (uint32_t)cur_in[(i % kCbInFrames) * kWave];
                const uint32_t cth_q = __builtin_amdgcn_readlane(tab_cth, i);
                uint64_t not_quiet = 0ull, fires = 0ull;
#pragma unroll
                for (uint32_t j = 0; j < N; ++j) {
                    const uint32_t v = (vin_q >> (8 * j)) & 0xffu;
                    // (an unpopped unit is in here as a black one: it stays quiet only on a zero)
                    not_quiet |= __builtin_amdgcn_ballot_w64(contrast_exceeded(v, px[j].base, cth_q)) |
                                 (unpopped[j] & __builtin_amdgcn_ballot_w64(v != 0u));
                    fires |= __builtin_amdgcn_ballot_w64(cb_quiet_fires<L>(px[j], v));
                }
                if (not_quiet != 0ull) break;  // frame i: the general loop's
                if (fires != 0ull) {
#pragma unroll
                    for (uint32_t j = 0; j < N; ++j) cb_step_quiet<L, true>(px[j], (vin_q >> (8 * j)) & 0xffu, T);
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < N; ++j) cb_step_quiet<L, false>(px[j], (vin_q >> (8 * j)) & 0xffu, T);
                }
                // (wt / wo of this frame stay 0: no events, no records)
                ++i;
            }
            if (i < nb) {  // frame i goes to the general loop, which stages into w.in at a group's first frame
                if (pref) __builtin_amdgcn_s_waitcnt(0x0f70);  // (what was on its way into the level slots has landed: they may hold levels again)
                if (cur != 0u && (i % kCbInFrames) != 0u)
                    cb_stage_input<FULL>(fr0, n_units_u, sgw, u0, lane, i - (i % kCbInFrames), nb, w.in, direct);
            }
        }
    }
#endif
    // (general frames up to the end of the input group -- the quiet test above then gets another look; the bound is the
    // loop's own, the frames carry no test)
    const uint32_t i_end = ADDER_CB_QUIET_PATH ? ((i / kCbInFrames + 1u) * kCbInFrames < nb ? (i / kCbInFrames + 1u) * kCbInFrames : nb) : nb;
#ifdef ADDER_CB_PROFILE
    const unsigned long long cbp_g0 = __builtin_readcyclecounter();
    cbp_acc[2] += cbp_g0 - cbp_q0;
    cbp_acc[3] += i - cbp_i0;
    const uint32_t cbp_i1 = i;
#endif
    for (; i < i_end; ++i) {
        // (a frame handed over by the quiet loop mid-group finds its group staged)
        if ((i % kCbInFrames) == 0u) cb_stage_input<FULL>(fr0, n_units_u, sgw, u0, lane, i, nb, w.in, direct);
        // (reading the next frame's bytes one frame ahead was measured: no difference)
        const uint32_t vin_w = (uint32_t)in_lds[(i % kCbInFrames) * kWave];
        sc.cth = __builtin_amdgcn_readlane(tab_cth, i);
        sc.running_t = __uint_as_float(__builtin_amdgcn_readlane(tab_rt, i));
        sc.running_t_u32 = f32_as_u32(sc.running_t);

        // ---------------- the step of every unit + the event counts ----------------
        CbPlanT<L> plan[N];
        uint32_t lane_cnt = 0u;
        // one unit after the other, each through its whole step: the step's booleans of one unit are dead before the next
        // unit's are made (with the booleans as wave masks that halves the scalar registers the step holds at once)
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            CbMidT<L> mid1;
            cb_step_a<L>(px[j], lv[j], (vin_w >> (8 * j)) & 0xffu, sc, plan[j], mid1);
            if (__builtin_amdgcn_ballot_w64(L::lane(mid1.walk)) != 0ull) {  // uniform
                cb_step_reads<L>(lv[j], mid1);
                cb_step_b<L, CbLevelsDev, true>(px[j], lv[j], T, sc, plan[j], mid1);
            } else {
                cb_step_b<L, CbLevelsDev, false>(px[j], lv[j], T, sc, plan[j], mid1);
            }
            depth_error = L::or_(depth_error, plan[j].depth_error);
            if (!FULL) plan[j].count = u0 + j < n_units_u ? plan[j].count : 0u;
            lane_cnt += plan[j].count;
        }
        // ---------------- wave-level ordered compaction into the segment's log ----------------
        const uint32_t incl = wave_inclusive_scan_dpp(lane_cnt);
        const uint32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
        uint32_t run_start;
        uint2 *const seg = log.append(total, run_start, a.status);  // uniform
        wt = lane == i ? (total | (total << 16)) : wt;
        wo = lane == i ? run_start : wo;
        uint32_t off = incl - lane_cnt;  // final offset of the lane's first event inside the segment
        if (total != 0u)  // (uniform: most frames of static or lossy content leave a segment without a single event)
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            if (plan[j].count != 0u && seg) {
                EmitCb em{seg, (lane * N + j) | (off << 7), off * kGenRecBytes};
                cb_emit<ABS_T, L>(px[j], plan[j], sc, lv[j], em);
            }
            off += plan[j].count;
            cb_pop<L>(px[j], plan[j], lv[j]);
        }
    }
#ifdef ADDER_CB_PROFILE
    cbp_acc[0] += __builtin_readcyclecounter() - cbp_g0;
    cbp_acc[1] += i - cbp_i1;
#endif
    }
#ifdef ADDER_CB_PROFILE
    const unsigned long long cbp_loop_end = __builtin_readcyclecounter();
#endif
    if (L::lane(depth_error)) raise(a.status, kStatusDepth);
    log.close(lane);
    if (lane < nb) {
        uint32_t s = slot0 + lane;
        s = s >= slots_u ? s - slots_u : s;
        gstore<uint32_t>(uniform_ptr(b->wtot_ring), (s * num_waves_u + sgw) * 4u, wt);
        gstore<uint32_t>(uniform_ptr(b->wofs_ring), (s * num_waves_u + sgw) * 4u, wo);
    }

    // ---------------- state back to HBM: levels in their resident form ----------------
    {
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N], lfv[N];
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const DeepGlobal g{a.dv_integ, a.dv_dt, a.dv_bdt, a.dv_bd, a.plane_stride, (size_t)u0 + j};
            for (uint32_t k = 1; k < px[j].m; ++k) g.store(k, cb_node_from_level(px[j], lv[j].load(k)));
            hdrv[j] = cb_hdr(px[j]);
            iv[j] = px[j].S;
            dv[j] = px[j].dt0;
            bv[j] = px[j].bdt0;
            lfv[j] = px[j].lastf;
        }
        constexpr bool NTS = ADDER_NT_STATE != 0;
        store_vec<NTS>(a.hdr, u0, hdrv);
        store_vec<NTS>(a.integ0, u0, iv);
        store_vec<NTS>(a.dt0, u0, dv);
        store_vec<NTS>(a.bdt0, u0, bv);
        if (ABS_T) store_vec<NTS>(a.lastf, u0, lfv);
        if (a.running) {  // side plane (the host keeps nb == 1 while it is enabled)
#pragma unroll
            for (uint32_t j = 0; j < N; ++j)
                if (u0 + j < n_units_u && px[j].m != 0u)
                    a.running[u0 + j] = (uint8_t)frame_value_u8(lean_bd_from_thr(f32_to_bits(px[j].thr0)),
                                                                f32_as_u32(px[j].bdt0), (double)sc.ref_time);
        }
    }
#ifdef ADDER_CB_PROFILE
    if (b->timeline && lane == 0u) {
        const unsigned long long now = __builtin_readcyclecounter();
        unsigned long long *const t = b->timeline + (3u * kTimelineChunks + 32u) * 2u + 1u;
        for (uint32_t q = 0; q < 4u; ++q) atomicAdd(t + q * 2u, cbp_acc[q]);
        atomicAdd(t + 4u * 2u, now - cbp_wave0);
        atomicAdd(t + 5u * 2u, 1ull);
        atomicMax(t + 6u * 2u, now - cbp_wave0);
        atomicMax(t + 7u * 2u, cbp_acc[0]);       // the wave with the most general-path cycles
        atomicMax(t + 8u * 2u, cbp_acc[1]);       // ... and the most general frames
        atomicAdd(t + 9u * 2u, cbp_loop_end - cbp_wave0 - cbp_acc[0] - cbp_acc[2]);  // prologue
        atomicAdd(t + 10u * 2u, now - cbp_loop_end);  // epilogue
        if (cbp_acc[1] >= 48ull) { atomicAdd(t + 11u * 2u, now - cbp_wave0); atomicAdd(t + 12u * 2u, 1ull); atomicAdd(t + 13u * 2u, cbp_acc[0]); }  // busy waves
    }
#endif
}

template <bool ABS_T>
__global__ __launch_bounds__(kBlockThreads, ADDER_CB_WAVES_PER_SIMD) void adder_cb_kernel(const BatchArgs *__restrict__ b,
                                                                                         uint32_t f, uint32_t nb) {
    __shared__ CbWaveLds s_w[kWavesPerBlock];
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    timeline_mark(b, 0u, f, false);
    chain_zero(b, f, nb);  // (adder_scan_kernel CHAIN)
    // (a capped grid walks the segments, like the lean kernel; the wave's LDS slice is its own, no barrier needed)
    for (uint32_t gw = blockIdx.x * kWavesPerBlock + tid / kWave; gw < a.num_waves; gw += gridDim.x * kWavesPerBlock) {
        const uint32_t u0 = gw * kWaveUnits + lane * kUnitsPerLane;
        const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
        if (full) cb_run_segment<ABS_T, true>(b, a, nb, u0, gw, lane, s_w[tid / kWave]);
        else cb_run_segment<ABS_T, false>(b, a, nb, u0, gw, lane, s_w[tid / kWave]);
    }
    timeline_mark(b, 0u, f, true);
}

// ------------------------------------------------------------------------------------------
// K1, constant runs -- `adder_cr_kernel<ABS_T>`: the bounded Collapse regime while c_thresh is 0 in every frame (crf 0).
// There a run integrates ONE intensity and the arena is a function of (intensity, frames since the flush)
// (adder_pixel.hpp, CONSTANT-RUN STEP): only the roots are loaded, stepped and stored; the levels behind a root are
// worked out in closed form when a flush emits them or pop_top promotes one, and written back to the deep planes in their
// resident form at the end of the launch, so that any other step may run next.  No level storage in LDS, no walk: the
// wave's LDS slice only parks the launch's input bytes (all of them up front, like the lean kernel).  Records, logs and
// expansion are the bounded Collapse kernel's (EmitCb / SegLog / expansion format 0 with the 9-bit d code).
// ------------------------------------------------------------------------------------------
#ifndef ADDER_CR_WAVES_PER_SIMD
#define ADDER_CR_WAVES_PER_SIMD 5
#endif
#ifndef ADDER_CR_WAVE_LANES
#define ADDER_CR_WAVE_LANES 1
#endif
template <bool ABS_T, bool FULL>
__device__ __forceinline__ void cr_run_segment(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb,
                                               uint32_t u0, uint32_t gw, uint32_t lane, uint8_t *lds_in) {
    constexpr uint32_t N = kUnitsPerLane;
#if ADDER_CR_WAVE_LANES
    using L = WaveLanes;
#else
    using L = ScalarLanes;
#endif
    StepConsts sc = a.sc;
    const float T = sc.time_spanned;
    CrPxT<L> px[N];
    uint32_t snap_m[N];
    {
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N], lfv[N];
        load_vec<ADDER_NT_STATE != 0>(a.hdr, u0, hdrv);
        load_vec<ADDER_NT_STATE != 0>(a.integ0, u0, iv);
        load_vec<ADDER_NT_STATE != 0>(a.dt0, u0, dv);
        load_vec<ADDER_NT_STATE != 0>(a.bdt0, u0, bv);
        if (ABS_T) load_vec<ADDER_NT_STATE != 0>(a.lastf, u0, lfv);
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            px[j] = cr_unpack<L>(hdrv[j], iv[j], dv[j], bv[j], ABS_T ? lfv[j] : 0.0f, T);
            snap_m[j] = hdr_m(hdrv[j]);
        }
    }
    if (snap_deep_wanted(b, a)) {  // the undo copy takes the levels as the planes have them
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) snap_deep_levels(b, a, (size_t)u0 + j, snap_m[j]);
    }
    const uint32_t sgw = __builtin_amdgcn_readfirstlane(gw);
    const uint32_t slots_u = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t num_waves_u = __builtin_amdgcn_readfirstlane(a.num_waves);
    const uint32_t n_units_u = __builtin_amdgcn_readfirstlane(a.n_units);
    const uint32_t f0 = __builtin_amdgcn_readfirstlane(a.frame_idx);
    const uint32_t slot0 = __builtin_amdgcn_readfirstlane(f0 % slots_u);
    uint32_t tab_rt = 0u;
    if (lane < nb) tab_rt = gload<uint2>(uniform_ptr(b->ftab), (f0 + lane) * (uint32_t)sizeof(FrameTab)).x;
    SegLog log;
    log.open(b, f0, sgw, num_waves_u);
    // the launch's input bytes, all of them, into the wave's LDS slice (lean_frames has the reasons)
    using InT = typename VecOf<uint8_t, N>::type;
    InT *const in_lds = reinterpret_cast<InT *>(lds_in) + lane;  // [frame][lane]
    {
        constexpr uint32_t NB_MAX = kMaxFramesPerLaunch;
        const uint8_t *const fr0 = uniform_ptr(b->frames) + (size_t)f0 * n_units_u;
        static_assert(kWaveUnits == 128u && NB_MAX % 8u == 0u, "eight frames of one segment per instruction");
        const bool direct = ADDER_LDS_DIRECT_INPUT != 0 && FULL &&
                            __builtin_amdgcn_readfirstlane(((n_units_u | (uint32_t)(uintptr_t)fr0) & 15u) == 0u);
        if (direct) {
            const uint8_t *const seg_in = fr0 + (size_t)sgw * kWaveUnits + (lane & 7u) * 16u;
#pragma unroll
            for (uint32_t g = 0; g < NB_MAX / 8u; ++g) {
                uint32_t k = g * 8u + (lane >> 3);
                k = k < nb ? k : nb - 1u;
                __builtin_amdgcn_global_load_lds((const ADDER_GLOBAL void *)(seg_in + (size_t)k * n_units_u),
                                                 (__attribute__((address_space(3))) void *)(lds_in + g * 1024u), 16, 0,
                                                 ADDER_NT_INPUT ? 2 : 0);
            }
        } else {
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < NB_MAX; k0 += 8u) {
                uint32_t vin8[8];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) {
                    const uint32_t k = k0 + q;
                    const uint32_t kk = k < nb ? k : nb - 1u;
                    vin8[q] = load_input(fr0 + (size_t)kk * n_units_u, u0, FULL ? 0xffffffffu : n_units_u);
                }
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) in_lds[(k0 + q) * kWave] = (InT)vin8[q];
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): nothing inside the frame loop waits on memory
    }
    uint32_t wt = 0u, wo = 0u;  // lane i: {events | records << 16} and the run's start of the launch's i-th frame
    for (uint32_t i = 0; i < nb; ++i) {
        const uint32_t vin_w = (uint32_t)in_lds[i * kWave];
        sc.running_t = __uint_as_float(__builtin_amdgcn_readlane(tab_rt, i));
        sc.running_t_u32 = f32_as_u32(sc.running_t);
        CrPlanT<L> plan[N];
        uint32_t lane_cnt = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            cr_step<L>(px[j], (vin_w >> (8 * j)) & 0xffu, T, sc, plan[j]);
            if (!FULL) plan[j].count = u0 + j < n_units_u ? plan[j].count : 0u;  // padding units: stepped freely, no events
            lane_cnt += plan[j].count;
        }
        const uint32_t incl = wave_inclusive_scan_dpp(lane_cnt);
        const uint32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
        uint32_t run_start;
        uint2 *const seg = log.append(total, run_start, a.status);  // uniform
        wt = lane == i ? (total | (total << 16)) : wt;
        wo = lane == i ? run_start : wo;
        uint32_t off = incl - lane_cnt;
        if (total != 0u) {  // (uniform)
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) {
                if (plan[j].count != 0u && seg) {
                    EmitCb em{seg, (lane * N + j) | (off << 7), off * kGenRecBytes};
                    cr_emit<ABS_T, L>(px[j], plan[j], T, sc, em);
                }
                off += plan[j].count;
                cr_pop<L>(px[j], plan[j], T);
            }
        }
    }
    log.close(lane);
    if (lane < nb) {
        uint32_t s = slot0 + lane;
        s = s >= slots_u ? s - slots_u : s;
        gstore<uint32_t>(uniform_ptr(b->wtot_ring), (s * num_waves_u + sgw) * 4u, wt);
        gstore<uint32_t>(uniform_ptr(b->wofs_ring), (s * num_waves_u + sgw) * 4u, wo);
    }
    // ---------------- state back to HBM: the roots, and the levels worked out into their resident form ----------------
    {
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N], lfv[N];
        bool too_deep = false;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            struct Store {
                DeepGlobal g;
                uint32_t max_depth;
                bool over;
                __device__ __forceinline__ void operator()(uint32_t k, const Node &n) {
                    if (k < max_depth) g.store(k, n);
                    else over = true;
                }
            } st{DeepGlobal{a.dv_integ, a.dv_dt, a.dv_bdt, a.dv_bd, a.plane_stride, (size_t)u0 + j}, sc.max_depth, false};
            const uint32_t m = cr_materialize<L>(px[j], T, st);
            too_deep = too_deep || (st.over && (FULL || u0 + j < n_units_u));
            hdrv[j] = hdr_make(px[j].base, lean_bd_from_thr(f32_to_bits(px[j].thr0)), m < kHdrMMask ? m : kHdrMMask,
                               L::lane(px[j].popped));
            iv[j] = px[j].S;
            dv[j] = px[j].dt0;
            bv[j] = px[j].bdt0;
            lfv[j] = px[j].lastf;
        }
        if (too_deep) raise(a.status, kStatusDepth);
        constexpr bool NTS = ADDER_NT_STATE != 0;
        store_vec<NTS>(a.hdr, u0, hdrv);
        store_vec<NTS>(a.integ0, u0, iv);
        store_vec<NTS>(a.dt0, u0, dv);
        store_vec<NTS>(a.bdt0, u0, bv);
        if (ABS_T) store_vec<NTS>(a.lastf, u0, lfv);
        if (a.running) {  // side plane (the host keeps nb == 1 while it is enabled)
#pragma unroll
            for (uint32_t j = 0; j < N; ++j)
                if (u0 + j < n_units_u && L::lane(px[j].has))
                    a.running[u0 + j] = (uint8_t)frame_value_u8(lean_bd_from_thr(f32_to_bits(px[j].thr0)),
                                                                f32_as_u32(px[j].bdt0), (double)sc.ref_time);
        }
    }
}

template <bool ABS_T>
__global__ __launch_bounds__(kBlockThreads, ADDER_CR_WAVES_PER_SIMD) void adder_cr_kernel(const BatchArgs *__restrict__ b,
                                                                                         uint32_t f, uint32_t nb) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kWavesPerBlock][kMaxFramesPerLaunch * kWaveUnits];
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    timeline_mark(b, 0u, f, false);
    chain_zero(b, f, nb);  // (adder_scan_kernel CHAIN)
    for (uint32_t gw = blockIdx.x * kWavesPerBlock + tid / kWave; gw < a.num_waves; gw += gridDim.x * kWavesPerBlock) {
        const uint32_t u0 = gw * kWaveUnits + lane * kUnitsPerLane;
        const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
        if (full) cr_run_segment<ABS_T, true>(b, a, nb, u0, gw, lane, s_in[tid / kWave]);
        else cr_run_segment<ABS_T, false>(b, a, nb, u0, gw, lane, s_in[tid / kWave]);
    }
    timeline_mark(b, 0u, f, true);
}

// ------------------------------------------------------------------------------------------
// K1, run records -- `adder_rr_kernel<ABS_T>`: the bounded Collapse regime under the constant-run conditions with the
// whole step in integers (adder_pixel.hpp, RUN RECORDS): a unit is {base_val, n, popped} and, in AbsoluteT,
// last_fired_t / T.  The step is a handful of compares and a counter; a flush / collapsed flush / pop_top parks ONE record
// of 8 (DeltaT) or 12 bytes, whatever the number of events it stands for (a byte table of chain lengths gives the count),
// and the expansion works the events out.  Loads the header and delta_t (and last_fired_t) planes, stores the level-0
// planes and the levels in their resident form.  Records go to a fixed slot per (segment, frame) laid out like the lean
// records' (at most one per unit and frame; groups of 16 segments contiguous per frame: what one expansion wave reads),
// inside the generic log ring; the expansion is format 4.
// ------------------------------------------------------------------------------------------
#ifndef ADDER_RR_WAVES_PER_SIMD
#define ADDER_RR_WAVES_PER_SIMD 6
#endif
constexpr uint32_t kRrInFrames = 32;
template <bool ABS_T, bool FULL>
__device__ __forceinline__ void rr_run_segment(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb, uint32_t u0,
                                               uint32_t gw, uint32_t lane, uint8_t *lds_in, const uint8_t *lds_tab, bool lazy) {
    constexpr uint32_t N = kUnitsPerLane;
    constexpr uint32_t NB_MAX = kMaxFramesPerLaunch;
    constexpr uint32_t REC = ABS_T ? 12u : 8u;
    using L = WaveLanes;
    const float T = a.sc.time_spanned;
    RrPxT<L> px[N];
    {
        uint32_t hdrv[N];
        float dv[N], lfv[N];
        load_vec<ADDER_NT_STATE != 0>(a.hdr, u0, hdrv);
        load_vec<ADDER_NT_STATE != 0>(a.dt0, u0, dv);
        if (ABS_T) load_vec<ADDER_NT_STATE != 0>(a.lastf, u0, lfv);
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) px[j] = rr_unpack<L>(hdrv[j], dv[j], ABS_T ? lfv[j] : 0.0f, T, ABS_T);
        if (snap_deep_wanted(b, a)) {  // the undo copy takes the levels as the planes have them
#pragma unroll
            for (uint32_t j = 0; j < N; ++j) snap_deep_levels(b, a, (size_t)u0 + j, hdr_m(hdrv[j]));
        }
    }
    const uint32_t sgw = __builtin_amdgcn_readfirstlane(gw);
    const uint32_t slots_u = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t num_waves_u = __builtin_amdgcn_readfirstlane(a.num_waves);
    const uint32_t n_units_u = __builtin_amdgcn_readfirstlane(a.n_units);
    const uint32_t f0 = __builtin_amdgcn_readfirstlane(a.frame_idx);
    const uint32_t slot0 = __builtin_amdgcn_readfirstlane(f0 % slots_u);
    const uint32_t chunk_u = __builtin_amdgcn_readfirstlane(b->chunk);
    // frames since the reset before this launch, and the frames after which the root is popped (:394-396)
    const uint32_t frame0 = __builtin_amdgcn_readfirstlane((uint32_t)fdiv(a.sc.running_t, T));
    const bool collapse_u = __builtin_amdgcn_readfirstlane((uint32_t)a.sc.collapse) != 0u;  // (Mode Normal runs the same step: rr_step)
    const uint32_t Tu = __builtin_amdgcn_readfirstlane((uint32_t)T);
    const uint32_t n_pop = __builtin_amdgcn_readfirstlane(((uint32_t)a.sc.dtm_f + Tu - 1u) / Tu);
    // the segment's slot of the launch's first frame (lr_frames: the same ring layout and walk)
    const uint32_t park_bytes_u = __builtin_amdgcn_readfirstlane(b->park_bytes);
    const ParkLayout lay = park_layout_u(b);
    const uint32_t frame_stride_u = lay.frame_stride;
    uint8_t *seg = uniform_ptr(b->park_ring) + park_offset(slot0, sgw, chunk_u, num_waves_u, park_bytes_u, lay);
    const uint32_t ridx0 = __builtin_amdgcn_readfirstlane((slot0 % chunk_u + (sgw >> lay.rot_shift)) & lay.rot_mask);
    const uint32_t wrap_at = chunk_u - 1u - ridx0;  // (after this frame of the launch the walk returns to the chunk's first slot)
    const uint32_t wrap_bytes = chunk_u * frame_stride_u;
    // the launch's input bytes into the wave's LDS slice, kRrInFrames frames at a time (lr_frames has the reasons)
    using InT = typename VecOf<uint8_t, N>::type;
    InT *const in_lds = reinterpret_cast<InT *>(lds_in) + lane;  // [frame of the group][lane]
    const uint8_t *const fr0 = uniform_ptr(b->frames) + (size_t)f0 * n_units_u;
    static_assert(kWaveUnits == 128u && kRrInFrames % 8u == 0u && NB_MAX % kRrInFrames == 0u, "eight frames of one segment per instruction");
    const bool direct = ADDER_LDS_DIRECT_INPUT != 0 && FULL &&
                        __builtin_amdgcn_readfirstlane(((n_units_u | (uint32_t)(uintptr_t)fr0) & 15u) == 0u);
    auto stage = [&](uint32_t k0) {  // frames [k0, k0 + kRrInFrames) of the launch
        if (direct) {
            const uint8_t *const seg_in = fr0 + (size_t)sgw * kWaveUnits + (lane & 7u) * 16u;
#pragma unroll
            for (uint32_t g = 0; g < kRrInFrames / 8u; ++g) {
                uint32_t k = k0 + g * 8u + (lane >> 3);
                k = k < nb ? k : nb - 1u;
                __builtin_amdgcn_global_load_lds((const ADDER_GLOBAL void *)(seg_in + (size_t)k * n_units_u),
                                                 (__attribute__((address_space(3))) void *)(lds_in + g * 1024u), 16, 0,
                                                 ADDER_NT_INPUT ? 2 : 0);
            }
        } else {
#pragma unroll 1
            for (uint32_t q0 = 0; q0 < kRrInFrames; q0 += 8u) {
                uint32_t vin8[8];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) {
                    const uint32_t k = k0 + q0 + q;
                    const uint32_t kk = k < nb ? k : nb - 1u;
                    vin8[q] = load_input(fr0 + (size_t)kk * n_units_u, u0, FULL ? 0xffffffffu : n_units_u);
                }
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) in_lds[(q0 + q) * kWave] = (InT)vin8[q];
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): nothing else inside the frame loop waits on memory
    };
    const auto chain_tab = [&](uint32_t I, uint32_t r) -> uint32_t { return lds_tab[I * kRrTabRows + r]; };
    uint32_t wt = 0u;  // lane i: {events | records << 16} of the launch's i-th frame
    bool active[N];
#pragma unroll
    for (uint32_t j = 0; j < N; ++j) active[j] = FULL || u0 + j < n_units_u;
    for (uint32_t i = 0; i < nb; ++i) {
        if ((i % kRrInFrames) == 0u) stage(i);
        const uint32_t vin_w = (uint32_t)in_lds[(i % kRrInFrames) * kWave];
        uint32_t w0[N], w1[N], w2[N], cnt[N];
        uint64_t mrec[N];
        uint32_t lane_cnt = 0u, nrec = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            rr_step<ABS_T, L>(px[j], (vin_w >> (8 * j)) & 0xffu, frame0 + i, n_pop, T, chain_tab, (lane * N + j) << kRrUnitShift,
                              w0[j], w1[j], w2[j], cnt[j], collapse_u);
            if (!FULL) cnt[j] = active[j] ? cnt[j] : 0u;  // padding units: stepped freely, no events
            mrec[j] = L::from(cnt[j] != 0u);
            nrec += (uint32_t)__popcll(mrec[j]);
            lane_cnt += cnt[j];
        }
        const uint32_t nev = __builtin_amdgcn_readlane(wave_inclusive_scan_dpp(lane_cnt), kWave - 1);
        uint32_t pos = 0u;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j)
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mrec[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mrec[j], pos));
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const bool has = L::lane(mrec[j]);
            if (has) {
                if (ABS_T) {
                    struct R12 { uint32_t a, b, c; };
                    gstore(seg, pos * REC, R12{w0[j], w1[j], w2[j]});
                } else {
                    gstore(seg, pos * REC, make_uint2(w0[j], w2[j]));
                }
            }
            pos += has ? 1u : 0u;
        }
        wt = lane == i ? (nev | (nrec << 16)) : wt;
        seg += frame_stride_u;
        if (__builtin_expect(i == wrap_at, 0)) seg -= wrap_bytes;
    }
    if (lane < nb) {
        uint32_t s = slot0 + lane;
        s = s >= slots_u ? s - slots_u : s;
        gstore<uint32_t>(uniform_ptr(b->wtot_ring), (s * num_waves_u + sgw) * 4u, wt);
    }
    report_run_max(b, px[0].n > px[1].n ? px[0].n : px[1].n, lane);
    if (lazy) {  // another launch of this batch follows: only what rr_unpack reads (header, delta_t, last_fired_t)
        uint32_t hdrv[N];
        float dv[N], lfv[N];
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            hdrv[j] = hdr_make(px[j].base, 0u, px[j].n != 0u ? 1u : 0u, L::lane(px[j].popped));
            dv[j] = px[j].base != 0u ? fmul((float)px[j].n, T) : 0.0f;
            lfv[j] = ABS_T ? fmul((float)px[j].lq, T) : 0.0f;
        }
        constexpr bool NTS_ = ADDER_NT_STATE != 0;
        store_vec<NTS_>(a.hdr, u0, hdrv);
        store_vec<NTS_>(a.dt0, u0, dv);
        if (ABS_T) store_vec<NTS_>(a.lastf, u0, lfv);
        return;
    }
    {   // state back to HBM in its resident form
        uint32_t hdrv[N];
        float iv[N], dv[N], bv[N], lfv[N];
        bool too_deep = false;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            struct Store {
                DeepGlobal g;
                uint32_t max_depth;
                bool over;
                __device__ __forceinline__ void operator()(uint32_t k, const Node &n) {
                    if (k < max_depth) g.store(k, n);
                    else over = true;
                }
            } st{DeepGlobal{a.dv_integ, a.dv_dt, a.dv_bdt, a.dv_bd, a.plane_stride, (size_t)u0 + j}, a.sc.max_depth, false};
            hdrv[j] = rr_pack<L>(px[j], T, iv[j], dv[j], bv[j], lfv[j], st, collapse_u);
            too_deep = too_deep || (st.over && active[j]);
        }
        if (too_deep) raise(a.status, kStatusDepth);
        constexpr bool NTS = ADDER_NT_STATE != 0;
        store_vec<NTS>(a.hdr, u0, hdrv);
        store_vec<NTS>(a.integ0, u0, iv);
        store_vec<NTS>(a.dt0, u0, dv);
        store_vec<NTS>(a.bdt0, u0, bv);
        if (ABS_T) store_vec<NTS>(a.lastf, u0, lfv);
        if (a.running) {  // side plane (the host keeps nb == 1 while it is enabled)
#pragma unroll
            for (uint32_t j = 0; j < N; ++j)
                if (u0 + j < n_units_u && hdr_m(hdrv[j]) != 0u)
                    a.running[u0 + j] = (uint8_t)frame_value_u8((hdrv[j] >> kHdrBdShift) & 0xffu, f32_as_u32(bv[j]), (double)a.sc.ref_time);
        }
    }
}

template <bool ABS_T>
__global__ __launch_bounds__(kBlockThreads, ADDER_RR_WAVES_PER_SIMD) void adder_rr_kernel(const BatchArgs *__restrict__ b,
                                                                                         uint32_t f, uint32_t nb, uint32_t lazy) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kWavesPerBlock][kRrInFrames * kWaveUnits];
    __shared__ __attribute__((aligned(16))) uint8_t s_tab[256 * kRrTabRows];
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    timeline_mark(b, 0u, f, false);
    chain_zero(b, f, nb);  // (adder_scan_kernel CHAIN)
    {   // the table of chain lengths (8 KB, the same for every launch: it comes out of L2)
        const uint4 *const src = reinterpret_cast<const uint4 *>(b->rr_tab);
        uint4 *const dst = reinterpret_cast<uint4 *>(s_tab);
        for (uint32_t k = tid; k < 256u * kRrTabRows / 16u; k += kBlockThreads) dst[k] = src[k];
        __syncthreads();
    }
    for (uint32_t gw = blockIdx.x * kWavesPerBlock + tid / kWave; gw < a.num_waves; gw += gridDim.x * kWavesPerBlock) {
        const uint32_t u0 = gw * kWaveUnits + lane * kUnitsPerLane;
        const bool full = __builtin_amdgcn_readfirstlane(gw * kWaveUnits + kWaveUnits <= a.n_units);
        if (full) rr_run_segment<ABS_T, true>(b, a, nb, u0, gw, lane, s_in[tid / kWave], s_tab, lazy != 0u);
        else rr_run_segment<ABS_T, false>(b, a, nb, u0, gw, lane, s_in[tid / kWave], s_tab, lazy != 0u);
    }
    timeline_mark(b, 0u, f, true);
}

// ------------------------------------------------------------------------------------------
// K1, Mode::Continuous: the general arena step (adder_pixel.hpp cont_step), a unit's nodes
// straight from / to the node planes.  A first, correct version -- the sources that use this mode are sparse
// event cameras; nothing here is tuned.  A unit's events cannot be counted before they are produced, so they
// are STAGED: 8-byte {t, d} records at a fixed slot range per unit plus a per-unit count; the expansion
// (format 2) orders them.  Segment scratch: [kWaveUnits] u8 counts, then [kWaveUnits][stage_events] records.
// ------------------------------------------------------------------------------------------
struct ContGlobal {
    float *integ, *dt, *bdt;
    uint32_t *meta;
    size_t stride, u;
    __device__ __forceinline__ ANode load(uint32_t k) const {
        const size_t i = (size_t)k * stride + u;
        ANode n;
        n.integ = integ[i];
        n.dt = dt[i];
        n.bdt = bdt[i];
        anode_set_meta(n, meta[i]);
        return n;
    }
    __device__ __forceinline__ void store(uint32_t k, const ANode &n) const {
        const size_t i = (size_t)k * stride + u;
        integ[i] = n.integ;
        dt[i] = n.dt;
        bdt[i] = n.bdt;
        meta[i] = anode_meta(n);
    }
};
struct EmitStage {
    uint2 *dst;
    uint32_t n, cap;
    __device__ __forceinline__ void operator()(uint32_t d, uint32_t t) {
        if (n < cap) dst[n] = make_uint2(t, d);
        ++n;
    }
};

template <bool ABS_T>
__global__ __launch_bounds__(kBlockThreads) void adder_cont_kernel(const BatchArgs *__restrict__ b, uint32_t f0,
                                                                   uint32_t nb) {
    constexpr uint32_t N = kUnitsPerLane;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t gw = blockIdx.x * kWavesPerBlock + tid / kWave;
    const uint32_t u0 = gw * kWaveUnits + lane * N;
    for (uint32_t i = 0; i < nb; ++i) {
        const FrameArgs a = frame_args(b, f0 + i);
        uint8_t *const seg = b->park_ring + park_offset((f0 + i) % b->slots, gw, b->chunk, a.num_waves, b->park_bytes, b->park_layout);
        uint32_t lane_cnt = 0;
        bool bad = false;
#pragma unroll
        for (uint32_t j = 0; j < N; ++j) {
            const uint32_t u = u0 + j;
            uint32_t cnt = 0;
            if (u < a.n_units) {
                APx s = apx_unpack(a.hdr[u], a.lastf[u]);
                ContGlobal acc{a.cn_integ, a.cn_dt, a.cn_bdt, a.cn_meta, a.plane_stride, u};
                EmitStage em{reinterpret_cast<uint2 *>(seg + kWaveUnits) + (size_t)(lane * N + j) * a.stage_events, 0u,
                             a.stage_events};
                const uint32_t v = a.frame[u];
                const bool ok = cont_step<ABS_T>(s, acc, v, (float)v, a.sc.time_spanned, a.sc, a.max_nodes, em);
                bad = bad || !ok || em.n > em.cap;
                cnt = em.n < em.cap ? em.n : em.cap;
                a.hdr[u] = apx_hdr(s);
                a.lastf[u] = s.lastf;
                if (a.running) {  // side plane (video.rs:713-730): the root's best event, if it has one
                    const ANode r = acc.load(0);
                    if (r.has_best) a.running[u] = (uint8_t)frame_value_u8(r.bd, f32_as_u32(r.bdt), (double)a.sc.ref_time);
                }
            }
            seg[lane * N + j] = (uint8_t)cnt;
            lane_cnt += cnt;
        }
        const uint32_t incl = wave_inclusive_scan_dpp(lane_cnt);
        if (lane == kWave - 1) a.wtot[gw] = incl | 0xffff0000u;  // records: staged (the expansion reads the counts)
        if (bad) raise(a.status, kStatusDepth);
    }
}

// ------------------------------------------------------------------------------------------
// Scan: exclusive prefix of the per-segment event counts of one frame per block
// (blockIdx.x = frame inside the chunk) and the frame's total.
// ------------------------------------------------------------------------------------------
// 256 threads: in the pipelined graph this kernel starts beside the resident frame kernel of the next chunk (5
// waves per SIMD), and a 1024-thread block needs 4 free wave slots on EVERY SIMD of one CU: it queued for 32 us
// where it runs 8 (round 6, with the scan + expansion branch the step's critical path: 1024 / 512 / 256 / 128 threads =
// 0.915 / 0.908 / 0.900 / 0.92 ms per headline step).  A thread owns `per` consecutive uint4 groups; up to kScanRegGroups of them are fetched in one
// go and kept in registers for both passes.
constexpr uint32_t kScanThreads = ADDER_SCAN_THREADS;
#ifndef ADDER_SCAN_REG_GROUPS
#define ADDER_SCAN_REG_GROUPS 16
#endif
constexpr uint32_t kScanRegGroups = ADDER_SCAN_REG_GROUPS;
__device__ __forceinline__ uint32_t scan_lo4(const uint4 &v) {
    return (v.x & 0xffffu) + (v.y & 0xffffu) + (v.z & 0xffffu) + (v.w & 0xffffu);
}
__device__ __forceinline__ uint32_t scan_hi4(const uint4 &v) { return (v.x >> 16) + (v.y >> 16) + (v.z >> 16) + (v.w >> 16); }
__device__ __forceinline__ uint4 scan_prefix4(const uint4 &v, uint32_t &run) {
    uint4 o;
    o.x = run;
    o.y = o.x + (v.x & 0xffffu);
    o.z = o.y + (v.y & 0xffffu);
    o.w = o.z + (v.z & 0xffffu);
    run = o.w + (v.w & 0xffffu);
    return o;
}
// Planes of more than kScanTileWaves segments (4K and up) are scanned in TILES of that many, a workgroup each: one block
// per frame walked 95 uncoalesced groups per thread on a 4K RGB plane (194 400 segments) and took 134 us per 48-frame chunk
// with 48 workgroups on the chip.  adder_scan_tiles_kernel first leaves every tile's {events, records}; the scan proper
// then adds up the tiles in front of its own (<= a few dozen words) and scans its tile from registers.  The tile sums live
// behind the two halves of ftot_ring: [slot][tile][2].
__device__ __forceinline__ uint32_t scan_tiles_of(uint32_t num_waves) { return (num_waves + kScanTileWaves - 1u) / kScanTileWaves; }
__global__ __launch_bounds__(kScanThreads) void adder_scan_tiles_kernel(const BatchArgs *__restrict__ b, uint32_t f0) {
    const uint32_t ntiles = scan_tiles_of(b->base.num_waves);
    const uint32_t fr = blockIdx.x / ntiles, tile = blockIdx.x - fr * ntiles;
    const FrameArgs a = frame_args(b, f0 + fr);
    __shared__ uint32_t s_ev[kScanThreads / kWave], s_rc[kScanThreads / kWave];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    const uint32_t groups = a.num_waves / 4u, tg = kScanTileWaves / 4u;
    const uint32_t t0 = tile * tg, t1 = min(t0 + tg, groups);
    const uint4 *src = reinterpret_cast<const uint4 *>(a.wtot);
    uint32_t ev = 0u, rc = 0u;
    for (uint32_t g = t0 + tid; g < t1; g += kScanThreads) {  // (only sums: coalesced)
        const uint4 w = src[g];
        ev += scan_lo4(w);
        rc += scan_hi4(w);
    }
#pragma unroll
    for (uint32_t o = kWave / 2; o > 0; o >>= 1) {
        ev += __shfl_down(ev, o, kWave);
        rc += __shfl_down(rc, o, kWave);
    }
    if (lane == 0) {
        s_ev[wid] = ev;
        s_rc[wid] = rc;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t e = 0u, r = 0u;
        for (uint32_t w = 0; w < kScanThreads / kWave; ++w) {
            e += s_ev[w];
            r += s_rc[w];
        }
        uint32_t *const ts = b->ftot_ring + 2u * b->slots + (((f0 + fr) % b->slots) * ntiles + tile) * 2u;
        ts[0] = e;
        ts[1] = r;
    }
}
// rec_prefix: also the exclusive prefix of the segments' RECORD counts, into the frame's wofs row (batches that hand their
// records out, adder_hip_integrate_records_device: where a segment's run goes in the packed buffer).
__global__ __launch_bounds__(kScanThreads) void adder_scan_kernel(const BatchArgs *__restrict__ b, uint32_t f0, uint32_t whole_batch,
                                                                  uint32_t rec_prefix) {
    const uint32_t ntiles = scan_tiles_of(b->base.num_waves);  // (1 for every plane up to 2 M units: the whole frame in one block)
    const uint32_t fr = ntiles == 1u ? blockIdx.x : blockIdx.x / ntiles, tile = blockIdx.x - fr * ntiles;
    const FrameArgs a = frame_args(b, f0 + fr);
    timeline_mark(b, 1u, f0, false);
    __shared__ uint32_t s_part[kScanThreads / kWave];
    __shared__ uint32_t s_recs[kScanThreads / kWave];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wid = tid / kWave;
    // num_waves is a multiple of 4: each thread owns `per` consecutive uint4 groups of the block's tile
    const uint32_t groups = a.num_waves / 4u;
    const uint32_t t0 = tile * (kScanTileWaves / 4u), t1 = min(t0 + kScanTileWaves / 4u, groups);
    const uint32_t per = (t1 - t0 + kScanThreads - 1) / kScanThreads;
    const uint32_t g0 = t0 + tid * per;
    const uint32_t g1 = min(g0 + per, t1);
    const uint4 *src = reinterpret_cast<const uint4 *>(a.wtot);
    uint4 *dst = reinterpret_cast<uint4 *>(a.wpref);
    // the tiles in front of this one (their sums are there: adder_scan_tiles_kernel ran first) and, for the block of the
    // last tile, the frame's records
    uint32_t tile_base = 0u, tile_recs = 0u, tile_rbase = 0u;
    if (ntiles > 1u) {
        const uint32_t *const ts = b->ftot_ring + 2u * b->slots + ((f0 + fr) % b->slots) * ntiles * 2u;
        for (uint32_t t = 0; t < ntiles; ++t) {  // (uniform: scalar loads)
            if (t < tile) tile_base += ts[2u * t];
            if (t < tile) tile_rbase += ts[2u * t + 1u];
            if (t != tile) tile_recs += ts[2u * t + 1u];
        }
    }
    const bool in_regs = per <= kScanRegGroups;  // uniform
    uint4 v[kScanRegGroups];
    uint32_t sum = 0, recs = 0;
    if (in_regs) {
#pragma unroll
        for (uint32_t k = 0; k < kScanRegGroups; ++k) v[k] = g0 + k < g1 ? src[g0 + k] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (uint32_t k = 0; k < kScanRegGroups; ++k) {
            sum += scan_lo4(v[k]);
            recs += scan_hi4(v[k]);
        }
    } else {
        for (uint32_t g = g0; g < g1; ++g) {
            const uint4 w = src[g];
            sum += scan_lo4(w);
            recs += scan_hi4(w);
        }
    }
    // parked records of the frame (bench.py's byte accounting; the records gather); the offsets kernel adds the frames up
    const uint32_t incl_r = wave_inclusive_scan(recs, lane);
    if (lane == kWave - 1) s_recs[wid] = incl_r;
    const uint32_t incl = wave_inclusive_scan(sum, lane);
    if (lane == kWave - 1) s_part[wid] = incl;
    __syncthreads();
    uint32_t base = tile_base, total = tile_base, rec_total = tile_recs, rbase = tile_rbase;
#pragma unroll
    for (uint32_t w = 0; w < kScanThreads / kWave; ++w) {
        const uint32_t t = s_part[w], r = s_recs[w];
        if (w < wid) base += t;
        if (w < wid) rbase += r;
        total += t;
        rec_total += r;
    }
    if (rec_prefix) {  // (uniform) the segments' record prefix of this frame -> its wofs row
        uint4 *const dst_r = reinterpret_cast<uint4 *>(b->wofs_ring + (size_t)((f0 + fr) % b->slots) * a.num_waves);
        uint32_t run_r = rbase + incl_r - recs;
        for (uint32_t g = g0; g < g1; ++g) {
            const uint4 w = src[g];  // (read again: indexing the register copy by a loop counter would put it in scratch)
            uint4 o;
            o.x = run_r;
            o.y = o.x + (w.x >> 16);
            o.z = o.y + (w.y >> 16);
            o.w = o.z + (w.z >> 16);
            run_r = o.w + (w.w >> 16);
            dst_r[g] = o;
        }
    }
    uint32_t run = base + incl - sum;
    if (in_regs) {
#pragma unroll
        for (uint32_t k = 0; k < kScanRegGroups; ++k) {
            const uint4 o = scan_prefix4(v[k], run);
            if (g0 + k < g1) dst[g0 + k] = o;
        }
    } else {
        for (uint32_t g = g0; g < g1; ++g) dst[g] = scan_prefix4(src[g], run);
    }
    if (tid == 0 && tile + 1u == ntiles) {  // (the last tile's block knows the frame's totals)
        *a.ftot = total;
        a.ftot[b->slots] = rec_total;  // second half of the ring: records per frame
        if (whole_batch & 1u) {  // a batch of ONE frame (the per-frame calls and the ring): the offsets kernel's work, a launch less
            b->base.frame_offsets[0] = 0ull;
            b->base.frame_offsets[1] = total;
            if (b->rec_total) *b->rec_total = rec_total;
        }
    }
    // CHAIN (whole_batch & 2; the packed lean-runs batches): the frame_offsets chain without the offsets kernel's launch.
    // frame_offsets[f0 + j + 1] = frame_offsets[f0] + the totals of frames f0 .. f0 + j: this frame's block ADDS its total
    // to every later entry of the chunk (one atomic instruction, a lane per entry; the chunk's first block also adds the
    // chain's value in front of the chunk -- final since the previous chunk's scan ended).  The entries were zeroed by the
    // frame kernel that ran these frames (adder_lp_kernel); sums commute, nothing waits, the expansion starts a kernel later.
    if ((whole_batch & 2u) && wid == 0u && tile + 1u == ntiles) {
        const uint32_t nf = gridDim.x / ntiles;
        unsigned long long *const offs = reinterpret_cast<unsigned long long *>(b->base.frame_offsets);
        unsigned long long add = total;
        if (fr == 0u && f0 != 0u) add += offs[f0];
        if (lane >= fr && lane < nf) (void)__hip_atomic_fetch_add(&offs[f0 + lane + 1u], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0u && b->rec_total)
            (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(b->rec_total), (unsigned long long)rec_total, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    }
    timeline_mark(b, 1u, f0, true);
}

// frame_offsets[f+1] = frame_offsets[f] + events(f) for the frames of the chunk, in order.
// The batch's first chunk also starts the chain (frame_offsets[0] = 0) and the record count, so the host queues no
// memset in front of a batch.
// One wave: lane i fetches frame f0 + i's totals (one memory round trip for the whole chunk instead of a serial
// chain of nf of them: 8 -> 3 us per chunk), a wave prefix sum chains them.  nf <= kMaxChunk <= 64.
__global__ __launch_bounds__(kWave) void adder_offsets_kernel(const BatchArgs *__restrict__ b, uint32_t f0, uint32_t nf) {
    if (blockIdx.x != 0) return;
    timeline_mark(b, 2u, f0, false);
    const uint32_t lane = threadIdx.x;
    uint64_t *offs = b->base.frame_offsets;
    const uint32_t slots = b->slots;
    uint64_t run = f0 == 0u ? 0ull : offs[f0];
    uint64_t recs = (f0 == 0u || !b->rec_total) ? 0ull : *b->rec_total;
    if (lane == 0u && f0 == 0u) offs[0] = 0ull;
    for (uint32_t i0 = 0; i0 < nf; i0 += kWave) {  // (one trip: a chunk holds <= kMaxChunk <= 64 frames)
        const uint32_t i = i0 + lane;
        uint64_t ev = 0ull, rc = 0ull;
        if (i < nf) {
            const uint32_t slot = (f0 + i) % slots;
            ev = b->ftot_ring[slot];
            rc = b->ftot_ring[slots + slot];
        }
#pragma unroll
        for (uint32_t o = 1; o < kWave; o <<= 1) {  // inclusive prefix sums over the lanes
            const uint64_t e = __shfl_up(ev, o, kWave), r = __shfl_up(rc, o, kWave);
            if (lane >= o) {
                ev += e;
                rc += r;
            }
        }
        if (i < nf) offs[f0 + i + 1u] = run + ev;
        run += __shfl(ev, kWave - 1, kWave);
        recs += __shfl(rc, kWave - 1, kWave);
    }
    if (b->rec_total && lane == 0u) *b->rec_total = recs;
    timeline_mark(b, 2u, f0, true);
}

// Last node of a batch: what the host needs to know about it, written straight into page-locked host memory (the
// host then only waits for the batch's last event: no device-to-host copies).
__global__ void adder_publish_kernel(const BatchArgs *__restrict__ b, uint32_t num_frames, BatchResult *h) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    h->total_events = b->base.frame_offsets[num_frames];
    h->records = b->rec_total ? *b->rec_total : 0ull;
    h->status = *b->base.status;
    h->max_run = 0u;
    if (b->run_max) {  // (read and cleared here: the next batch starts from zero without a host-side memset in front of it)
        h->max_run = *b->run_max;
        *b->run_max = 0u;
    }
    __threadfence_system();
    h->valid = 1u;
}

// ------------------------------------------------------------------------------------------
// K2: parked records -> final 12-byte events of the ordered stream.  A segment parks only a few
// dozen records, so a wave per segment would be nothing but start-up latency.  Each wave takes
// kExpandSegs consecutive segments of one frame: their events are CONTIGUOUS in the stream (the scan
// gave segment s the events [wpref[s], wpref[s+1])), so the wave decodes its records into an LDS
// staging buffer in stream order and writes the buffer out with full-width, 16-byte aligned,
// coalesced stores (a lane writing its own 12-byte events straight to HBM made the kernel
// store-issue bound: three sparse dwordx3 stores per 64 records).
// Lean batches park one LeanRec per unit with events: the wave decodes 64 records at a time
// (lean_decode: <= 3 events each) and a DPP scan of the per-record event counts gives each
// event its place; generic batches park one 8-byte record per event with its final offset.
// ------------------------------------------------------------------------------------------
#ifndef ADDER_XBUF_EVENTS
#define ADDER_XBUF_EVENTS 448
#endif
constexpr uint32_t kXbufEvents = ADDER_XBUF_EVENTS;   // staging capacity of one wave, in events (>= 192)
constexpr uint32_t kXbufDwords = kXbufEvents * 3 + 4; // + the 16-byte phase of the destination

struct UnitCoord {  // (row, offset in row) of a segment's first unit + the plane geometry (uniform)
    uint32_t y0, rem0, rowlen, channels, row_begin;
    bool one_wrap;
};
__device__ __forceinline__ uint32_t coord_xy_c(const UnitCoord &uc, uint32_t unit, uint32_t &c) {
    uint32_t rem = uc.rem0 + unit;
    uint32_t y = uc.y0;
    if (uc.one_wrap) {
        const bool wrap = rem >= uc.rowlen;
        rem -= wrap ? uc.rowlen : 0u;
        y += wrap ? 1u : 0u;
    } else {
        const uint32_t dq = rem / uc.rowlen;
        rem -= dq * uc.rowlen;
        y += dq;
    }
    uint32_t x = rem;
    c = 0xffu;
    if (uc.channels == 3u) {
        x = (uint32_t)(((uint64_t)rem * 0xAAAAAAABull) >> 33);  // rem / 3
        c = rem - 3u * x;
    }
    return x | ((y + uc.row_begin) << 16);
}

// The wave's staging buffer -> the stream.  xb[phase .. phase + 3 n) holds n events whose first dword
// goes to dword `gd0` of the output (phase == gd0 & 3, so 16-byte blocks of the buffer are 16-byte
// blocks of the destination).  Uniform arguments; returns nothing, the caller resets its fill.
__device__ __forceinline__ void xbuf_flush_dwords(const uint32_t *xb, uint32_t phase, uint32_t nd, uint32_t *out_dw,
                                                  uint64_t gd0, uint32_t lane);
__device__ __forceinline__ void xbuf_flush(const uint32_t *xb, uint32_t phase, uint32_t n, uint32_t *out_dw,
                                           uint64_t gd0, uint32_t lane) {
    xbuf_flush_dwords(xb, phase, n * 3u, out_dw, gd0, lane);
}
// xb[phase .. phase + nd) -> dwords [gd0, gd0 + nd) of the output, phase == gd0 & 3
__device__ __forceinline__ void xbuf_flush_dwords(const uint32_t *xb, uint32_t phase, uint32_t nd, uint32_t *out_dw,
                                                  uint64_t gd0, uint32_t lane) {
    uint32_t head = (4u - phase) & 3u;
    head = head < nd ? head : nd;
    uint32_t *const dst = out_dw + gd0;  // uniform 64-bit base; the lanes add 32-bit offsets
    if (lane < head) gstore_ev<uint32_t>(dst, lane * 4u, xb[phase + lane]);
    const uint32_t body = (nd - head) >> 2;  // whole 16-byte blocks
    const uint32_t b0 = phase + head;        // a multiple of 4
    for (uint32_t k = lane; k < body; k += kWave) {
        const uint4 v = *reinterpret_cast<const uint4 *>(xb + b0 + 4u * k);
        gstore_ev<uint4>(dst, (head + 4u * k) * 4u, v);
    }
    const uint32_t tail = (nd - head) & 3u;
    if (lane < tail) gstore_ev<uint32_t>(dst, (head + 4u * body + lane) * 4u, xb[b0 + 4u * body + lane]);
}

// The same straight into the raw sink's records (RawOutput::ingest_event, raw/stream.rs:101-120: bincode fixint big-endian;
// 9 bytes {x u16, y u16, d u8, t u32} on a 1-channel plane, 11 bytes {x, y, 0x01, c, d, t} otherwise): the stream's
// largest buffer shrinks by a quarter and no second pass serialises it.  Event k of the stream lies at byte k * rec, so
// FOUR events are rec whole dwords: a lane takes four staged events whose stream index is a multiple of four (three
// 16-byte LDS reads: the caller staged the events so that these are aligned), funnels their 36 / 44 record bytes into 9 /
// 11 dwords and stores them; the <= 3 events before the first such group and after the last one share their dwords with
// the neighbouring waves' events and go out byte by byte.
struct WireWords {
    uint32_t w0, w1, w2;  // the record's bytes, little-endian packed (w2: its last 1 / 3 bytes)
};
__device__ __forceinline__ WireWords wire_words(uint32_t xy, uint32_t cd, uint32_t t, uint32_t rec) {
    WireWords r;
    r.w0 = __builtin_amdgcn_perm(0u, xy, 0x02030001u);  // x_hi x_lo y_hi y_lo
    const uint32_t tb = __builtin_amdgcn_perm(0u, t, 0x00010203u);  // t3 t2 t1 t0
    const uint32_t d = (cd >> 8) & 0xffu;
    if (rec == 9u) {
        r.w1 = d | (tb << 8);
        r.w2 = tb >> 24;
    } else {
        r.w1 = 1u | ((cd & 0xffffu) << 8) | (tb << 24);
        r.w2 = tb >> 8;
    }
    return r;
}
__device__ __forceinline__ void xbuf_flush_wire(const uint32_t *xb, uint32_t pad, uint32_t n, uint8_t *out, uint64_t g0,
                                                uint32_t rec, uint32_t lane) {
    uint32_t head = (4u - ((uint32_t)g0 & 3u)) & 3u;
    head = head < n ? head : n;
    const uint32_t body = (n - head) >> 2, tail = (n - head) & 3u;
    uint8_t *const dst = out + g0 * rec;  // uniform 64-bit base; the lanes add 32-bit offsets
    if (lane < head + tail) {  // byte by byte: the events that share dwords with another wave's
        const uint32_t k = lane < head ? lane : n - tail + (lane - head);
        const WireWords r = wire_words(xb[pad + 3u * k], xb[pad + 3u * k + 1u], xb[pad + 3u * k + 2u], rec);
        uint8_t *const p = dst + k * rec;
        // (global stores may sit at any byte address: two dwords and the rest, instead of nine to eleven byte stores --
        // the stores of a flush are counted in instructions, not bytes)
        __builtin_memcpy(p, &r.w0, 4);
        __builtin_memcpy(p + 4, &r.w1, 4);
        if (rec == 9u) {
            p[8] = (uint8_t)r.w2;
        } else {
            const uint16_t lo = (uint16_t)r.w2;
            __builtin_memcpy(p + 8, &lo, 2);
            p[10] = (uint8_t)(r.w2 >> 16);
        }
    }
    if (body == 0u) return;
    // the body: four events -> rec dwords, written back over the staged events (every lane has read its twelve dwords
    // before any lane writes, and the compacted run stays below what later passes still have to read), at the 16-byte
    // phase of the run's destination -- then the plain flush: 16-byte stores, a kilobyte per instruction
    const uint32_t b0 = pad + 3u * head;  // a multiple of 4 (the caller's pad)
    const uint64_t gd0 = (g0 + head) * rec >> 2;  // the run's first dword in the output
    const uint32_t q0 = (uint32_t)gd0 & 3u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t *const xw = const_cast<uint32_t *>(xb);
    for (uint32_t m0 = 0; m0 < body; m0 += kWave) {  // uniform trip count
        const uint32_t m = m0 + lane;
        if (m < body) {
            const uint4 q0v = *reinterpret_cast<const uint4 *>(xb + b0 + 12u * m);
            const uint4 q1 = *reinterpret_cast<const uint4 *>(xb + b0 + 12u * m + 4u);
            const uint4 q2 = *reinterpret_cast<const uint4 *>(xb + b0 + 12u * m + 8u);
            const WireWords r0 = wire_words(q0v.x, q0v.y, q0v.z, rec), r1 = wire_words(q0v.w, q1.x, q1.y, rec);
            const WireWords r2 = wire_words(q1.z, q1.w, q2.x, rec), r3 = wire_words(q2.y, q2.z, q2.w, rec);
            uint32_t *const o = xw + q0 + rec * m;  // rec dwords per group of four
            if (rec == 9u) {
                o[0] = r0.w0;
                o[1] = r0.w1;
                o[2] = (r0.w2 & 0xffu) | (r1.w0 << 8);
                o[3] = (r1.w0 >> 24) | (r1.w1 << 8);
                o[4] = (r1.w1 >> 24) | ((r1.w2 & 0xffu) << 8) | (r2.w0 << 16);
                o[5] = (r2.w0 >> 16) | (r2.w1 << 16);
                o[6] = (r2.w1 >> 16) | ((r2.w2 & 0xffu) << 16) | (r3.w0 << 24);
                o[7] = (r3.w0 >> 8) | (r3.w1 << 24);
                o[8] = (r3.w1 >> 8) | (r3.w2 << 24);
            } else {
                o[0] = r0.w0;
                o[1] = r0.w1;
                o[2] = (r0.w2 & 0xffffffu) | (r1.w0 << 24);
                o[3] = (r1.w0 >> 8) | (r1.w1 << 24);
                o[4] = (r1.w1 >> 8) | (r1.w2 << 24);
                o[5] = ((r1.w2 >> 8) & 0xffffu) | (r2.w0 << 16);
                o[6] = (r2.w0 >> 16) | (r2.w1 << 16);
                o[7] = (r2.w1 >> 16) | (r2.w2 << 16);
                o[8] = ((r2.w2 >> 16) & 0xffu) | (r3.w0 << 8);
                o[9] = (r3.w0 >> 24) | (r3.w1 << 8);
                o[10] = (r3.w1 >> 24) | (r3.w2 << 8);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next pass reads beyond what this one wrote; order the LDS traffic anyway)
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    xbuf_flush_dwords(xb, q0, body * rec, reinterpret_cast<uint32_t *>(out), gd0, lane);
}
// where the first staged event goes in the LDS buffer (dwords): events -> the 16-byte phase of its destination; wire
// records -> so that the first event whose stream index is a multiple of four lies on a 16-byte boundary
__device__ __forceinline__ uint32_t xbuf_phase(uint64_t gpos, uint32_t wire_rec) {
    if (wire_rec == 0u) return (uint32_t)(gpos * 3u) & 3u;
    const uint32_t head = (4u - ((uint32_t)gpos & 3u)) & 3u;
    return (4u - ((3u * head) & 3u)) & 3u;
}

// Stages the <= 3 events of one decoded lean record at dword w of the buffer.
__device__ __forceinline__ void stage_lean(uint32_t *xb, uint32_t w, const LeanEvents &e, uint32_t xy, uint32_t c) {
    if (e.a) {
        xb[w] = xy;
        xb[w + 1u] = c | (e.da << 8);
        xb[w + 2u] = e.ta;
        w += 3u;
    }
    if (e.b) {
        xb[w] = xy;
        xb[w + 1u] = c | (kDEmpty << 8);
        xb[w + 2u] = e.tb;
        w += 3u;
    }
    if (e.c) {
        xb[w] = xy;
        xb[w + 1u] = c | (e.dc << 8);
        xb[w + 2u] = e.tc;
    }
}

// One workgroup's share of a frame's expansion: 4 waves x kExpandSegs segments.
// FORMAT of the parked records: 0 = generic (8 bytes per event with its final offset), 1 = lean (one 16-byte
// LeanRec per unit), 2 = staged (Mode::Continuous: per-unit counts + 8-byte {t, d} records at fixed slots)
// a parked lean record as {ta, tc, w, -} (AbsoluteT, 12 bytes) or {ta, w8, -, -} (DeltaT, 8 bytes)
template <bool ABS_T>
__device__ __forceinline__ uint4 lean_load_rec(const void *base, uint32_t byte_off) {
    if (ABS_T) {
        struct R12 { uint32_t a, b, c; };
        const R12 r = gload_rec<R12>(base, byte_off);
        return make_uint4(r.a, r.b, r.c, 0u);
    }
    const uint2 r = gload_rec<uint2>(base, byte_off);
    return make_uint4(r.x, r.y, 0u, 0u);
}

// record `idx` of a segment's run at `base` (+ byte offset `off`)
template <int FORMAT, bool ABS_T>
__device__ __forceinline__ uint4 load_rec_at(const void *base, uint32_t off, uint32_t idx) {
    return lean_load_rec<ABS_T>(base, off + idx * lean_rec_bytes(ABS_T));
}
template <int FORMAT, bool ABS_T, bool WIRE = false>
__device__ __forceinline__ void expand_block(const BatchArgs *__restrict__ b, uint32_t f, uint32_t xblock) {
    constexpr bool LEAN = FORMAT == 1 || FORMAT == 3 || FORMAT == 4 || FORMAT == 5 || FORMAT == 6;  // (3: lean records in per-segment logs, variant bit 64; 4: run records there)
    constexpr bool RR = FORMAT == 4;
    constexpr bool LR = FORMAT == 5 || FORMAT == 6;  // lean-runs records (adder_lr_kernel) in fixed slots (5) or found through a run table (6:
                                      // the bands' packed records on root); formats of their own, so that
                                      // the decoders do not meet in one instantiation (their results would merge through registers)
    // (adder_lp_kernel's 4-byte records have an expansion of their own: adder_lpx_kernel, adder_lp_kernels.hip)
    // staging capacity of one wave, in events: run-record rounds hold up to 64 x (depth + 1) events and like room
    constexpr uint32_t XE = RR ? 640u : kXbufEvents;
    __shared__ __attribute__((aligned(16))) uint32_t s_xbuf[kWavesPerBlock][XE * 3u + 4u];
    // lean runs: event C by input byte (lr_build_tab), 1 KB per workgroup out of L2 -- a division less per record
    __shared__ uint32_t s_tab_c[LR && !ABS_T ? 256u : 1u];
    if (LR && !ABS_T) {
        // (the table's words are asked for first and parked after the test below: the two loads share one round trip -- a test
        // in front of the table load cost busy content a round trip per workgroup, 165 -> 177 us per chunk)
        const uint32_t tab_word = gload<uint32_t>(uniform_ptr(b->lr_tab), (256u * kLrTabRuns + threadIdx.x) * 4u);
        // a frame without a single event -- static content -- has nothing to expand: the workgroup is done before the table,
        // the barrier and the counts' round trip (the scan's frame total; the bands' descriptions of the records gather
        // carry none)
        const uint32_t *const ft = b->ftot_ring;
        if (ft != nullptr && __builtin_amdgcn_readfirstlane(ft[f % b->slots]) == 0u) return;
        s_tab_c[threadIdx.x] = tab_word;
        __syncthreads();
    }
    static_assert(kExpandSegs % 2u == 0u, "segments are expanded in pairs");
    // only the frame-independent part of the arguments is needed here
    const uint32_t slots = __builtin_amdgcn_readfirstlane(b->slots);
    const uint32_t slot = __builtin_amdgcn_readfirstlane(f % slots);
    const uint32_t num_waves = __builtin_amdgcn_readfirstlane(b->base.num_waves);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t seg0 = __builtin_amdgcn_readfirstlane((xblock * kWavesPerBlock + wid) * kExpandSegs);
    if (seg0 >= num_waves) return;
    uint32_t *const xb = s_xbuf[wid];
    const uint32_t park_bytes = __builtin_amdgcn_readfirstlane(b->park_bytes);
    // wave-uniform bases (SGPRs); the lanes add 32-bit byte offsets
    const uint32_t chunk_frames = __builtin_amdgcn_readfirstlane(b->chunk);
    // (the wave's kExpandSegs segments lie in one group -- seg0 is a multiple of kExpandSegs, a group holds 1 or a
    // multiple of kExpandSegs segments -- so consecutive ones are a constant stride apart)
    const ParkLayout lay = park_layout_u(b);
    // lean records of blocked batches lie in per-segment logs like the per-event ones (log_cap records per segment and
    // chunk, a frame's run at wofs); batches launched one frame at a time keep a fixed slot per frame
    constexpr bool lean_log = FORMAT == 3 || FORMAT == 6;
    const uint32_t lean_log_cap = lean_log ? __builtin_amdgcn_readfirstlane(b->log_cap) : 0u;
    const uint32_t seg_stride = FORMAT == 0 ? 0u : lean_log ? lean_log_cap * lean_rec_bytes(ABS_T) : __builtin_amdgcn_readfirstlane(
        (uint32_t)(park_offset(slot, seg0 + 1u, chunk_frames, num_waves, park_bytes, lay) -
                   park_offset(slot, seg0, chunk_frames, num_waves, park_bytes, lay)));
    const uint8_t *park = uniform_ptr(b->park_ring) +
                          (FORMAT == 0 ? (size_t)0
                           : lean_log  ? ((size_t)(slot / chunk_frames) * num_waves + seg0) * seg_stride
                                       : park_offset(slot, seg0, chunk_frames, num_waves, park_bytes, lay));
    const uint32_t *wtot = uniform_ptr(b->wtot_ring) + (size_t)slot * num_waves + seg0;
    const uint32_t *wpref = uniform_ptr(b->wpref_ring) + (size_t)slot * num_waves + seg0;
    UnitCoord uc;
    uc.rowlen = __builtin_amdgcn_readfirstlane(b->base.rowlen);
    uc.channels = __builtin_amdgcn_readfirstlane(b->base.channels);
    uc.row_begin = __builtin_amdgcn_readfirstlane(b->base.row_begin);
    uc.one_wrap = uc.rowlen >= 2u * kWaveUnits;  // a pair of segments crosses at most one row end
    const uint64_t out_cap = b->base.out_cap;
    const uint32_t rt_u32 = __builtin_amdgcn_readfirstlane(f32_as_u32(b->ftab[f].running_t));  // t of D_EMPTY (lean)
    const float time_spanned_u = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b->base.sc.time_spanned)));
    const uint32_t frame_idx_u = (LR && ABS_T) ? __builtin_amdgcn_readfirstlane((uint32_t)fdiv(b->ftab[f].running_t, time_spanned_u)) : 0u;  // (lr_decode12)
    constexpr bool lean_runs = LR;

    // num_waves is a multiple of kExpandSegs (n_pad is padded accordingly).  Two round trips: the
    // segments' counts first, then exactly the records they hold (a speculative fetch of 64 records per
    // segment moved 3.4x the bytes: a segment parks ~20 records of 16 bytes)
    uint32_t my_tot = 0u;
    if (lane < kExpandSegs) my_tot = gload<uint32_t>(wtot, lane * 4u);
    // (a wave's life is a chain of memory round trips -- counts, records, stores: everything that does not depend on the
    // counts is asked for in the same trip as the counts)
    const uint32_t pref0 = gload<uint32_t>(wpref, 0u);  // events of the frame before segment seg0
    const uint64_t fo = b->base.frame_offsets[f];
    uint32_t lean_ofs = 0u;  // lean logs: lane q < kExpandSegs holds the byte offset of segment seg0 + q's run in its region
    if (lean_log && lane < kExpandSegs)
        lean_ofs = gload<uint32_t>(uniform_ptr(b->wofs_ring) + (size_t)slot * num_waves + seg0, lane * 4u) * lean_rec_bytes(ABS_T);
    // quiet content: most waves find sixteen empty segments -- they are done here, before the records and the set-up
    if (__builtin_amdgcn_ballot_w64((my_tot & 0xffffu) != 0u) == 0ull) return;
    // Lean: segments go in PAIRS, lanes 0-31 on the even one and lanes 32-63 on the odd one (a segment
    // parks ~20 records: one 64-lane round per segment would leave two thirds of the lanes idle); a pair
    // with a segment of more than 32 records takes the one-segment-at-a-time path below.
    const uint32_t half = lane >> 5, hl = lane & 31u;
    uint4 first[RR ? 1u : LEAN ? kExpandSegs / 2u : kExpandSegs];  // (lean: {ta, tc, w, -}; run records are fetched pair by pair)
    const uint8_t *log0 = nullptr;  // per-event records: the log region of segment seg0, the regions' stride, and
    uint32_t log_stride = 0u;       // where each segment's run of this frame starts inside its region (bytes)
    uint32_t log_run[FORMAT == 0 ? kExpandSegs : 1u];
    if (LEAN) {
        auto fetch_pair = [&](uint32_t p) -> uint4 {
            const uint32_t pa = __builtin_amdgcn_readlane(my_tot, 2 * p) >> 16;
            const uint32_t pb = __builtin_amdgcn_readlane(my_tot, 2 * p + 1) >> 16;
            const uint32_t oa = lean_log ? __builtin_amdgcn_readlane(lean_ofs, 2 * p) : 0u;
            const uint32_t ob = lean_log ? __builtin_amdgcn_readlane(lean_ofs, 2 * p + 1) : 0u;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (hl < (half ? pb : pa)) {
                v = load_rec_at<FORMAT, ABS_T>(park + (size_t)(2 * p) * seg_stride, (half ? seg_stride + ob : oa), hl);
            }
            return v;
        };
        if constexpr (RR) {
            first[0] = fetch_pair(0u);
        } else {
#pragma unroll
            for (uint32_t p = 0; p < kExpandSegs / 2u; ++p) first[p] = fetch_pair(p);
        }
    } else if (FORMAT == 0) {
        // per-event records: a segment's run of this frame starts at wofs inside the segment's log of the chunk
        uint32_t my_ofs = 0u;
        if (lane < kExpandSegs)
            my_ofs = gload<uint32_t>(uniform_ptr(b->wofs_ring) + (size_t)slot * num_waves + seg0, lane * 4u);
        const uint32_t log_cap = __builtin_amdgcn_readfirstlane(b->log_cap);
        log0 = uniform_ptr(b->park_ring) + ((size_t)(slot / chunk_frames) * num_waves + seg0) * log_cap * kGenRecBytes;
        log_stride = log_cap * kGenRecBytes;
#pragma unroll
        for (uint32_t q = 0; q < kExpandSegs; ++q) {
            const uint32_t parked = __builtin_amdgcn_readlane(my_tot, q) >> 16;
            log_run[q] = __builtin_amdgcn_readlane(my_ofs, q) * kGenRecBytes;
            first[q] = make_uint4(0u, 0u, 0u, 0u);
            if (lane < parked) {
                const uint2 v = gload_rec<uint2>(log0 + (size_t)q * log_stride, log_run[q] + lane * kGenRecBytes);
                first[q] = make_uint4(v.x, v.y, 0u, 0u);
            }
        }
    }
    uint32_t *const out_dw = reinterpret_cast<uint32_t *>(uniform_ptr(b->base.out));
    // the wave's events occupy [gpos, gpos + sum of its segments' event counts) of the stream
    // (frame_offsets is written by the offsets kernel: a vector load; its value goes to scalar registers by hand, or all
    // of the flush arithmetic below -- capacity check, phase, destination -- runs on the vector ALU in every lane)
    uint64_t gpos = (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(fo >> 32)) << 32) |
                     __builtin_amdgcn_readfirstlane((uint32_t)fo)) + __builtin_amdgcn_readfirstlane(pref0);
    uint32_t fill = 0u;                        // events staged (uniform)
    // WIRE: the raw sink's records instead of AdderEvents (an instantiation of its own: both flushes in one body cost
    // the unrolled segment loop its registers)
    const uint32_t wire_rec = WIRE ? __builtin_amdgcn_readfirstlane(b->base.wire_rec) : 0u;
    uint32_t phase = xbuf_phase(gpos, wire_rec);  // where the staging buffer's first event goes (dwords)
    bool dropped = false;
    // staged events -> stream; events past the caller's capacity are dropped and reported
    auto flush = [&]() {
        const uint64_t room64 = gpos < out_cap ? out_cap - gpos : 0ull;
        const uint32_t n = (uint64_t)fill <= room64 ? fill : (uint32_t)room64;
        dropped = dropped || n != fill;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n) {
            if constexpr (WIRE) xbuf_flush_wire(xb, phase, n, reinterpret_cast<uint8_t *>(out_dw), gpos, wire_rec, lane);
            else xbuf_flush(xb, phase, n, out_dw, gpos * 3u, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        gpos += fill;
        phase = xbuf_phase(gpos, wire_rec);
        fill = 0u;
    };
    // 64 lean records (or none) of ONE segment whose first unit is at (uc.y0, uc.rem0 + unit_shift)
    auto lean_round = [&](const uint4 &rw, uint32_t unit_shift) {
        if (fill + 3u * kWave > XE) flush();  // room for this round's worst case
        LeanRec r;
        r.ta = rw.x;
        r.tc = rw.y;
        r.w = rw.z;
        r.w8 = rw.y;
        // (an all-zero record decodes to no events)
        // (DeltaT batches of the lean-runs kernel park {rho, ..base_val..}: event A is worked out here -- uniform choice)
        LeanEvents e;
        if constexpr (LR && ABS_T) e = lr_decode12(rw.x, rw.y, rw.z, time_spanned_u, rt_u32, frame_idx_u);
        else if constexpr (LR) e = lr_decode8_tab(rw.x, rw.y, time_spanned_u, rt_u32, nullptr, s_tab_c);
        else if constexpr (ABS_T) e = lean_decode(r, true, rt_u32);
        else e = lean_decode8(rw.x, rw.y, time_spanned_u, rt_u32);
        const uint32_t n = (e.a ? 1u : 0u) + (e.b ? 1u : 0u) + (e.c ? 1u : 0u);
        const uint32_t incl = wave_inclusive_scan_dpp(n);
        const uint32_t ev0 = fill + incl - n;
        const uint32_t w = phase + ev0 + (ev0 << 1);  // the record's first dword in the buffer (x3 without a 64-bit multiply-add)
        fill += __builtin_amdgcn_readlane(incl, kWave - 1);
        uint32_t c;
        const uint32_t unit = lean_runs ? (ABS_T ? rw.z : rw.y) & 0x7fu : ABS_T ? (r.w >> kLeanUnitShift) & 0x3ffu : (rw.y >> kLean8UnitShift) & 0x7fu;
        const uint32_t xy = coord_xy_c(uc, unit + unit_shift, c);
        stage_lean(xb, w, e, xy, c);
    };
    // 64 run records (or none) of ONE segment: a scan of the records' counts places the round's events, every lane
    // walks its record's chain (one cr_node per event: the node's event, its last firing takes the walk to the next
    // level).  One lane per EVENT instead (an owner map in LDS, the record's words over ds_bpermute, k hops + the node per
    // lane) measured 204 us per 64-frame chunk against this loop's 186: the rounds hold ~58 events, so the second pass
    // of 64 lanes runs nearly empty, and the LDS round trips sit in the wave's critical path.
    auto rr_round = [&](const uint4 &rw, uint32_t unit_shift) {
        const uint32_t w2 = ABS_T ? rw.z : rw.y;
        const uint32_t cnt = w2 >> kRrCountShift;  // (an all-zero record: no events)
        const uint32_t incl = wave_inclusive_scan_dpp(cnt);
        const uint32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
        const uint32_t first_ev = incl - cnt;
        const uint32_t kind = w2 & 3u, Iu = (w2 >> kRrBaseShift) & 0xffu;
        const float I = (float)Iu;
        uint32_t c;
        const uint32_t xy = coord_xy_c(uc, ((w2 >> kRrUnitShift) & 0x7fu) + unit_shift, c);
        for (uint32_t e0 = 0; e0 < total; e0 += XE) {  // (uniform; one pass unless the round outgrows the buffer)
            const uint32_t piece = total - e0 < XE ? total - e0 : XE;
            if (fill + piece > XE) flush();
            const uint32_t n_run = rw.x & kRrRunMask;
            uint32_t r = n_run;
            for (uint32_t k = 0; k < cnt; ++k) {
                uint32_t d = kDEmpty, t = rt_u32;  // (the filler of a collapsed flush, :258-264)
                if (!(kind == kRrCollapsed && k != 0u)) {
                    // (kRrFlushPop's last event: the new run's root (v, 1) -- the chain has ended at r = 0 by then)
                    const bool popped_root = kind == kRrFlushPop && k + 1u == cnt;
                    const float Ik = popped_root ? (float)(rw.x >> kRrPopInShift) : I;
                    const uint32_t rk = popped_root ? 1u : r;
                    float bdt = time_spanned_u;  // a black root: (D_ZERO, time_spanned)
                    uint32_t j = 1u;
                    d = kDZero;
                    if (Iu != 0u || popped_root) {
                        const CrNode nd = cr_node(Ik, rk, time_spanned_u);
                        bdt = nd.bdt;
                        j = nd.j;
                        d = lean_bd_from_thr(f32_to_bits(nd.thr));
                    }
                    // event k's own last_fired_t: the firings above it telescope to n - r
                    t = f32_as_u32(ABS_T ? fadd(bdt, fmul((float)(rw.y + (n_run - r)), time_spanned_u)) : bdt);
                    r -= popped_root ? 0u : j;
                }
                const uint32_t pos = first_ev + k;
                if (pos >= e0 && pos < e0 + piece) {
                    const uint32_t w = phase + (fill + pos - e0) * 3u;
                    xb[w] = xy;
                    xb[w + 1u] = c | (d << 8);
                    xb[w + 2u] = t;
                }
            }
            fill += piece;
        }
    };
    auto record_round = [&](const uint4 &rw, uint32_t unit_shift) {
        if constexpr (RR) rr_round(rw, unit_shift);  // adder_rr_kernel's records: 1 .. depth + 1 events each
        else lean_round(rw, unit_shift);
    };
    // (row, offset in row) of the first unit of segment seg0: ONE wave-uniform division; the
    // following segments advance it with scalar add/compare instead of dividing again
    uc.y0 = __builtin_amdgcn_readfirstlane((seg0 * kWaveUnits) / uc.rowlen);
    uc.rem0 = seg0 * kWaveUnits - uc.y0 * uc.rowlen;
    auto next_segment = [&]() {  // uniform, scalar
        uc.rem0 += kWaveUnits;
        while (uc.rem0 >= uc.rowlen) {
            uc.rem0 -= uc.rowlen;
            ++uc.y0;
        }
    };
    if constexpr (RR) {
        // run records: a round is a loop over every lane's chain -- the pairs are not unrolled; the next pair's records are
        // fetched before this pair's round
        auto fetch_pair_rr = [&](uint32_t p) -> uint4 {
            const uint32_t pa = __builtin_amdgcn_readlane(my_tot, 2 * p) >> 16;
            const uint32_t pb = __builtin_amdgcn_readlane(my_tot, 2 * p + 1) >> 16;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (hl < (half ? pb : pa))
                v = lean_load_rec<ABS_T>(park + (size_t)(2 * p) * seg_stride, (half ? seg_stride : 0u) + hl * lean_rec_bytes(ABS_T));
            return v;
        };
        uint4 cur_rec = first[0];
#pragma unroll 1
        for (uint32_t p = 0; p < kExpandSegs / 2u; ++p) {
            const uint32_t pa = __builtin_amdgcn_readlane(my_tot, 2 * p) >> 16;
            const uint32_t pb = __builtin_amdgcn_readlane(my_tot, 2 * p + 1) >> 16;
            uint4 next_rec = make_uint4(0u, 0u, 0u, 0u);
            if (p + 1u < kExpandSegs / 2u) next_rec = fetch_pair_rr(p + 1u);
            if (pa <= 32u && pb <= 32u) {
                if (pa + pb != 0u) rr_round(cur_rec, half * kWaveUnits);
                next_segment();
                next_segment();
            } else {
                const uint32_t cnt2[2] = {pa, pb};
                for (uint32_t h = 0; h < 2u; ++h) {
                    const uint8_t *const seg_park = park + (size_t)(2 * p + h) * seg_stride;
                    for (uint32_t i0 = 0; i0 < cnt2[h]; i0 += kWave) {  // uniform trip count
                        uint4 rw = make_uint4(0u, 0u, 0u, 0u);
                        if (i0 + lane < cnt2[h]) rw = lean_load_rec<ABS_T>(seg_park, (i0 + lane) * lean_rec_bytes(ABS_T));
                        rr_round(rw, 0u);
                    }
                    next_segment();
                }
            }
            cur_rec = next_rec;
        }
    } else if (LEAN) {
#pragma unroll
        for (uint32_t p = 0; p < kExpandSegs / 2u; ++p) {
            const uint32_t pa = __builtin_amdgcn_readlane(my_tot, 2 * p) >> 16;
            const uint32_t pb = __builtin_amdgcn_readlane(my_tot, 2 * p + 1) >> 16;
            if (pa <= 32u && pb <= 32u) {
                if (pa + pb != 0u) record_round(first[p], half * kWaveUnits);
                next_segment();
                next_segment();
            } else {
                const uint32_t cnt[2] = {pa, pb};
#pragma unroll
                for (uint32_t h = 0; h < 2u; ++h) {
                    const uint8_t *const seg_park = park + (size_t)(2 * p + h) * seg_stride +
                                                    (lean_log ? __builtin_amdgcn_readlane(lean_ofs, 2 * p + h) : 0u);
                    for (uint32_t i0 = 0; i0 < cnt[h]; i0 += kWave) {  // uniform trip count
                        uint4 rw = make_uint4(0u, 0u, 0u, 0u);
                        if (i0 + lane < cnt[h]) {
                            rw = load_rec_at<FORMAT, ABS_T>(seg_park, 0u, i0 + lane);
                        }
                        record_round(rw, 0u);
                    }
                    next_segment();
                }
            }
        }
    } else if (FORMAT == 2) {
        // staged: a lane owns kUnitsPerLane units of the segment; a DPP scan of its event counts places them.
        // Written straight to the stream (no LDS staging: this mode is not tuned).
        const uint32_t stage_events = __builtin_amdgcn_readfirstlane(b->base.stage_events);
#pragma unroll 1
        for (uint32_t q = 0; q < kExpandSegs; ++q) {
            const uint32_t seg_events = __builtin_amdgcn_readlane(my_tot, q) & 0xffffu;
            const uint8_t *const seg_park = park + (size_t)q * seg_stride;
            if (seg_events != 0u) {
                uint32_t cnt[kUnitsPerLane], lane_cnt = 0;
#pragma unroll
                for (uint32_t j = 0; j < kUnitsPerLane; ++j) {
                    cnt[j] = gload<uint8_t>(seg_park, lane * kUnitsPerLane + j);
                    lane_cnt += cnt[j];
                }
                uint32_t pos = wave_inclusive_scan_dpp(lane_cnt) - lane_cnt;
                const uint64_t room64 = gpos < out_cap ? out_cap - gpos : 0ull;
                EventWords *const seg_out = reinterpret_cast<EventWords *>(out_dw) + gpos;
#pragma unroll
                for (uint32_t j = 0; j < kUnitsPerLane; ++j) {
                    uint32_t c;
                    const uint32_t xy = coord_xy_c(uc, lane * kUnitsPerLane + j, c);
                    for (uint32_t e = 0; e < cnt[j]; ++e, ++pos) {
                        const uint2 r = gload<uint2>(seg_park, kWaveUnits + ((lane * kUnitsPerLane + j) * stage_events + e) * 8u);
                        if ((uint64_t)pos < room64) {
                            EventWords w{xy, c | ((r.y & 0xffu) << 8), r.x};
                            gstore(seg_out, pos * (uint32_t)sizeof(EventWords), w);
                        } else {
                            dropped = true;
                        }
                    }
                }
                gpos += seg_events;
            }
            next_segment();
        }
    } else {
#pragma unroll
        for (uint32_t q = 0; q < kExpandSegs; ++q) {
            const uint32_t tot = __builtin_amdgcn_readlane(my_tot, q);
            const uint32_t parked = tot >> 16;
            const uint8_t *const seg_park = log0 + (size_t)q * log_stride + log_run[FORMAT == 0 ? q : 0u];
            // every record carries its event's offset inside the segment; a segment's events are staged
            // in pieces of the buffer's size (the segment total is known: tot & 0xffff)
            const uint32_t seg_events = tot & 0xffffu;
            for (uint32_t e0 = 0; e0 < seg_events; e0 += XE) {
                const uint32_t piece = seg_events - e0 < XE ? seg_events - e0 : XE;
                if (fill + piece > XE) flush();
                for (uint32_t i0 = 0; i0 < parked; i0 += kWave) {
                    const uint32_t i = i0 + lane;
                    if (i >= parked) continue;
                    uint2 sl;
                    if (i0 == 0u)
                        sl = make_uint2(first[q].x, first[q].y);
                    else
                        sl = gload_rec<uint2>(seg_park, i * kGenRecBytes);
                    const uint32_t pos = sl.y >> 16;  // final offset inside the segment
                    if (pos < e0 || pos >= e0 + piece) continue;
                    uint32_t c;
                    // (FORMAT 0 with ABS_T set: the bounded Collapse kernel's records, d as a 9-bit code -- EmitCb)
                    const uint32_t unit = ABS_T ? (sl.y >> 9) & 0x7fu : (sl.y >> 8) & 0xffu;
                    const uint32_t d = ABS_T ? cb_d_from_code(sl.y & 0x1ffu) : sl.y & 0xffu;
                    const uint32_t xy = coord_xy_c(uc, unit, c);
                    const uint32_t w = phase + (fill + pos - e0) * 3u;
                    xb[w] = xy;
                    xb[w + 1u] = c | (d << 8);
                    xb[w + 2u] = sl.x;
                }
                fill += piece;
            }
            next_segment();
        }
    }
    flush();
    if (dropped) raise(b->base.status, kStatusCapacity);
}

template <int FORMAT, bool ABS_T, bool WIRE = false>
__global__ __launch_bounds__(kBlockThreads) void adder_expand_kernel(const BatchArgs *__restrict__ b, uint32_t f0,
                                                                     uint32_t xblocks, uint32_t nf, uint32_t items) {
    // (frame, group of segments) pairs, frame-major; a workgroup takes `items` consecutive ones (a frame without events costs
    // its workgroups' dispatch and nothing else: fewer, longer workgroups), a grid smaller than their number walks them
    timeline_mark(b, 3u, f0, false);
    const uint32_t total = xblocks * nf;
    for (uint32_t w0 = blockIdx.x * items; w0 < total; w0 += gridDim.x * items) {
        const uint32_t w1 = w0 + items < total ? w0 + items : total;
        for (uint32_t w = w0; w < w1; ++w) {
            expand_block<FORMAT, ABS_T, WIRE>(b, f0 + w / xblocks, w % xblocks);
            __syncthreads();  // (the next item refills the workgroup's table)
        }
    }
    timeline_mark(b, 3u, f0, true);
}

// Records over the wire, root: ALL bands of a chunk in one launch (adder_hip_expand_records_device).  Work items are
// (frame, band, block of the band) frame-major, so the merged stream is written front to back as by the whole-plane
// expansion; the bands' BatchArgs lie `stride` bytes apart (eight launches of an eighth of the plane each took 337 us per
// chunk where the whole plane takes 150).
struct BandBlocks {
    uint32_t n_bands;
    uint32_t cum[kMaxBands + 1];  // blocks of the bands before band r (cum[n_bands] = all of them)
};
template <int FORMAT, bool ABS_T, bool WIRE>
__global__ __launch_bounds__(kBlockThreads) void adder_expand_bands_kernel(const uint8_t *__restrict__ descs, uint32_t stride,
                                                                          BandBlocks bb, uint32_t nf) {
    const uint32_t per_frame = bb.cum[bb.n_bands];
    const uint32_t w = blockIdx.x;
    if (w >= per_frame * nf) return;
    const uint32_t f = w / per_frame, rem = w - f * per_frame;
    uint32_t r = 0;
    while (r + 1u < bb.n_bands && rem >= bb.cum[r + 1u]) ++r;  // uniform
    expand_block<FORMAT, ABS_T, WIRE>(reinterpret_cast<const BatchArgs *>(descs + (size_t)r * stride), f, rem - bb.cum[r]);
}
// runs: the bands shipped lean-RUNS records (format 6) instead of lean records (format 3); wire: the merged output is the raw
// sink's 9 / 11-byte records (BatchArgs::base.wire_rec of the descriptions) instead of AdderEvents
extern "C" hipError_t adder_launch_expand_bands(const uint8_t *descs, uint32_t stride, uint32_t n_bands,
                                                const uint32_t *num_waves, uint32_t nf, uint32_t abs_t, hipStream_t stream,
                                                uint32_t runs, uint32_t wire) {
    if (n_bands == 0 || n_bands > kMaxBands) return hipErrorInvalidValue;
    BandBlocks bb;
    bb.n_bands = n_bands;
    bb.cum[0] = 0;
    const uint32_t per_block = kWavesPerBlock * kExpandSegs;
    for (uint32_t r = 0; r < n_bands; ++r) bb.cum[r + 1] = bb.cum[r] + (num_waves[r] + per_block - 1) / per_block;
    for (uint32_t r = n_bands + 1; r <= kMaxBands; ++r) bb.cum[r] = bb.cum[n_bands];
    const dim3 grid(bb.cum[n_bands] * nf);
#define ADDER_XB(F, A, W) hipLaunchKernelGGL((adder_expand_bands_kernel<F, A, W>), grid, dim3(kBlockThreads), 0, stream, descs, stride, bb, nf)
    if (runs) {
        if (abs_t) { if (wire) ADDER_XB(6, true, true); else ADDER_XB(6, true, false); }
        else { if (wire) ADDER_XB(6, false, true); else ADDER_XB(6, false, false); }
    } else {
        if (abs_t) { if (wire) ADDER_XB(3, true, true); else ADDER_XB(3, true, false); }
        else { if (wire) ADDER_XB(3, false, true); else ADDER_XB(3, false, false); }
    }
#undef ADDER_XB
    return hipGetLastError();
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

// ------------------------------------------------------------------------------------------
// Sink per rank (the alternative to the funnel: "each GPU D2H's its own segment and the host
// concatenates -- 8 PCIe links vs one"; consumer: video.rs:736-740 -> encoder.rs:233-273, raw/stream.rs:101-120).
// Every rank serialises ITS band's events of a chunk of frames to wire records and stores them straight to their final
// byte positions of the one .adder image, which lives in host memory every rank has mapped (a POSIX shm file = the
// file itself): no gather over xGMI, no staging copy, no host round trip per chunk.
//   adder_sink_layout_kernel   from the ranks' all-gathered frame offsets of the chunk: where this rank's segment of
//                              every frame starts in the file (event index), and the file position after the chunk.
//   adder_wire_scatter_kernel  12-byte AdderEvents -> 9 / 11-byte records (adder_wire_kernel's repack) at
//                              header + dest[f] * rec; a frame's segment starts at any byte phase, so every workgroup
//                              writes leading / trailing bytes singly and the dwords in between whole.
// ------------------------------------------------------------------------------------------
__global__ void adder_sink_layout_kernel(const uint64_t *__restrict__ all_offs, uint32_t world, uint32_t rank, uint32_t nf,
                                         uint64_t *file_pos, uint64_t *__restrict__ dest, uint64_t *__restrict__ merged_offs) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const size_t per = (size_t)nf + 1;
    uint64_t run = *file_pos;
    if (merged_offs) merged_offs[0] = run;
    for (uint32_t f = 0; f < nf; ++f) {
        uint64_t before = 0, tot = 0;
        for (uint32_t r = 0; r < world; ++r) {
            const uint64_t c = all_offs[r * per + f + 1] - all_offs[r * per + f];
            before += r < rank ? c : 0ull;
            tot += c;
        }
        dest[f] = run + before;
        run += tot;
        if (merged_offs) merged_offs[f + 1] = run;
    }
    *file_pos = run;
}

__global__ __launch_bounds__(256) void adder_wire_scatter_kernel(const uint32_t *__restrict__ ev, const uint64_t *__restrict__ offs,
                                                                 uint32_t nf, const uint64_t *__restrict__ dest, uint32_t rec,
                                                                 uint8_t *__restrict__ out, uint64_t out_cap, uint64_t header,
                                                                 uint32_t *status, uint64_t src_cap) {
    __shared__ uint32_t s_w[kWireEvents * 3];
    const uint32_t tid = threadIdx.x;
    const uint32_t magic = rec == 9u ? 0x38e38e39u : 0xba2e8ba3u;  // floor(B / rec) = mulhi(B, magic) >> (1 | 3)
    const uint32_t sh = rec == 9u ? 1u : 3u;
    bool bad = false, over = false;
    for (uint32_t f = 0; f < nf; ++f) {
        const uint64_t fb = offs[f];
        uint64_t fcnt = offs[f + 1] - fb;
        // (the offsets count what the frames PRODUCED; the buffer holds what fitted -- src_cap events: the rest was dropped
        // by the expansion and must not be read)
        const uint64_t held = fb < src_cap ? src_cap - fb : 0ull;
        if (fcnt > held) {
            fcnt = held;
            over = true;
        }
        const uint64_t nblocks = (fcnt + kWireEvents - 1) / kWireEvents;
        for (uint64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {  // uniform per workgroup
            const uint64_t e0 = blk * kWireEvents;
            const uint32_t cnt = (uint32_t)min((uint64_t)kWireEvents, fcnt - e0);
            const uint32_t *src = ev + (fb + e0) * 3u;
            __syncthreads();  // (the previous block's reads of s_w)
            for (uint32_t i = tid; i < cnt * 3u; i += 256u) s_w[i] = src[i];
            __syncthreads();
            for (uint32_t e = tid; e < cnt; e += 256u) {
                const uint32_t xy = s_w[3 * e], cd = s_w[3 * e + 1], t = s_w[3 * e + 2];
                const uint32_t x = xy & 0xffffu, y = xy >> 16, c = cd & 0xffu, d = (cd >> 8) & 0xffu;
                const uint32_t w0 = (x >> 8) | ((x & 0xffu) << 8) | ((y >> 8) << 16) | ((y & 0xffu) << 24);
                uint32_t w1, w2;
                if (rec == 9u) {
                    w1 = d | ((t >> 24) << 8) | (((t >> 16) & 0xffu) << 16) | (((t >> 8) & 0xffu) << 24);
                    w2 = t & 0xffu;
                } else {
                    bad |= c == 0xffu;
                    w1 = 1u | (c << 8) | (d << 16) | ((t >> 24) << 24);
                    w2 = ((t >> 16) & 0xffu) | (((t >> 8) & 0xffu) << 8) | ((t & 0xffu) << 16);
                }
                s_w[3 * e] = w0;
                s_w[3 * e + 1] = w1;
                s_w[3 * e + 2] = w2;
            }
            __syncthreads();
            const uint32_t bytes = cnt * rec;
            const uint64_t a0 = header + (dest[f] + e0) * rec;  // the block's first byte in the image
            if (a0 + bytes > out_cap) {
                over = true;
                continue;
            }
            uint8_t *const dst = out + a0;
            auto seg_byte = [&](uint32_t B) -> uint32_t {
                const uint32_t e = __umulhi(B, magic) >> sh;
                const uint32_t r = B - e * rec;
                return (s_w[3u * e + (r >> 2)] >> (8u * (r & 3u))) & 0xffu;
            };
            uint32_t head = (uint32_t)((4u - (a0 & 3u)) & 3u);
            head = head < bytes ? head : bytes;
            if (tid < head) dst[tid] = (uint8_t)seg_byte(tid);
            const uint32_t nd = (bytes - head) >> 2;
            uint32_t *const dst_dw = reinterpret_cast<uint32_t *>(dst + head);
            for (uint32_t k = tid; k < nd; k += 256u) {
                const uint32_t B = head + 4u * k;
                dst_dw[k] = seg_byte(B) | (seg_byte(B + 1u) << 8) | (seg_byte(B + 2u) << 16) | (seg_byte(B + 3u) << 24);
            }
            const uint32_t tail = (bytes - head) & 3u;
            if (tid < tail) dst[head + 4u * nd + tid] = (uint8_t)seg_byte(head + 4u * nd + tid);
        }
    }
    if (bad) raise(status, kStatusWire);
    if (over) raise(status, kStatusCapacity);
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

// ------------------------------------------------------------------------------------------
// Feature-driven rate control (video.rs:883-1112 handle_features, :866-882 handle_roi), run once
// per frame after the frame's events are in place and before the next frame is stepped.  A lane per event:
// an event is looked at if it is the last of its pixel's run (`e1.coord != e2.coord` over the row chunk's
// CIRCULAR windows), on channel 0 / None and not a D_EMPTY filler; FAST 9_16 on the running intensities decides
// whether its pixel joins or leaves the feature set.  A pixel is looked at once per frame at most, so the
// membership plane needs no atomics.  Around each NEW feature the wave then sets c_thresh = min(baseline, 2)
// over the clamped square of radius feature_c_radius, all channels; every writer stores the same value.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void feature_fill_rect(uint8_t *cth, uint32_t width, uint32_t channels, uint32_t x0,
                                                  uint32_t y0, uint32_t x1, uint32_t y1, uint32_t low, uint32_t lane) {
    const uint32_t span = (x1 - x0 + 1u) * channels;
    for (uint32_t y = y0; y <= y1; ++y) {
        uint8_t *row = cth + ((size_t)y * width + x0) * channels;
        for (uint32_t i = lane; i < span; i += kWave) row[i] = (uint8_t)low;
    }
}

// the clamped square of radius r around plane pixel (fx, fy), cut to the context's rows [row_begin, row_begin + rows)
__device__ __forceinline__ void feature_reset_around(const FrameArgs &a, const FeatureArgs &fa, uint32_t rows, uint32_t fx,
                                                     uint32_t fy, uint32_t lane) {
    const uint32_t x0 = fx > fa.radius ? fx - fa.radius : 0u, x1 = min(fx + fa.radius, a.width - 1u);
    const uint32_t gy0 = max(fy > fa.radius ? fy - fa.radius : 0u, a.row_begin);
    const uint32_t gy1 = min(min(fy + fa.radius, fa.plane_h - 1u), a.row_begin + rows - 1u);
    if (gy0 > gy1) return;
    feature_fill_rect(a.cth_px, a.width, a.channels, x0, gy0 - a.row_begin, x1, gy1 - a.row_begin, fa.low, lane);
}

__global__ __launch_bounds__(kBlockThreads) void adder_feature_kernel(const BatchArgs *__restrict__ b, uint32_t f,
                                                                      const FeatureArgs fa) {
    const FrameArgs &a = b->base;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave, nwaves = gridDim.x * kWavesPerBlock;
    const uint32_t rows = a.n_units / a.rowlen;
    if (fa.detect) {
        const uint64_t begin = a.frame_offsets[f];
        uint64_t end = a.frame_offsets[f + 1];
        if (end > a.out_cap) end = a.out_cap;  // the overflow is on record in the status word
        const AdderEventPod *const ev = a.out;
        // the running intensities in PLANE coordinates: row y of the plane at img + y * rowlen.  A band context only
        // holds rows [row_begin - 3, row_begin + rows + 3) of it (its own and the neighbours' halo); the corner test
        // reads rows y - 3 .. y + 3 of a pixel that is not within 3 pixels of the PLANE's border (cv.rs:57).
        const uint8_t *const img = a.running - (size_t)a.row_begin * a.rowlen;
        for (uint64_t base = begin + (uint64_t)wave * kWave; base < end; base += (uint64_t)nwaves * kWave) {
            const uint64_t i = base + lane;
            bool is_new = false;
            uint32_t x = 0, gy = 0;
            if (i < end) {
                x = ev[i].x;
                gy = ev[i].y;
                const bool look = feature_looked_at(ev, begin, end, i, a.row_begin, fa.chunk_rows);
                if (look) {
                    uint8_t *member = fa.fset + (size_t)(gy - a.row_begin) * a.width + x;
                    if (fast9_is_feature(img, a.width, fa.plane_h, a.channels, x, gy)) {
                        is_new = *member == 0u;
                        *member = 1u;
                    } else {
                        *member = 0u;
                    }
                }
            }
            uint64_t m = __ballot(is_new);
            if (m != 0ull) {
                uint32_t list_base = 0u;
                if (lane == 0) {
                    atomicAdd(fa.counters, (uint32_t)__popcll(m));
                    if (fa.new_xy) list_base = atomicAdd(fa.counters + 1, (uint32_t)__popcll(m));
                }
                if (fa.new_xy) {
                    list_base = __builtin_amdgcn_readfirstlane(list_base);
                    const uint32_t slot = list_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (is_new && slot < fa.new_cap) fa.new_xy[slot] = x | (gy << 16);
                }
            }
            if (fa.radius != 0u) {
                while (m != 0ull) {
                    const uint32_t src = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1ull;
                    feature_reset_around(a, fa, rows, __builtin_amdgcn_readlane(x, src), __builtin_amdgcn_readlane(gy, src), lane);
                }
            }
        }
    }
    if (fa.roi_on) {  // plane coordinates -> rows of this context
        const uint32_t y_lo = max(fa.ry0, a.row_begin), y_hi = min(fa.ry1, a.row_begin + rows - 1u);
        const uint32_t x1 = min(fa.rx1, a.width - 1u);
        if (fa.rx0 <= x1)
            for (uint32_t y = y_lo + wave; y <= y_hi && y_lo <= y_hi; y += nwaves)
                feature_fill_rect(a.cth_px, a.width, a.channels, fa.rx0, y - a.row_begin, x1, y - a.row_begin, fa.low, lane);
    }
}

// features found by the other row bands this frame: their reset squares reach into this band's rows
__global__ __launch_bounds__(kBlockThreads) void adder_feature_apply_kernel(const BatchArgs *__restrict__ b, const FeatureArgs fa,
                                                                            const uint32_t *__restrict__ xy, uint32_t n) {
    const FrameArgs &a = b->base;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave, nwaves = gridDim.x * kWavesPerBlock;
    const uint32_t rows = a.n_units / a.rowlen;
    if (fa.radius == 0u) return;
    for (uint32_t i = wave; i < n; i += nwaves) {
        const uint32_t w = xy[i];
        feature_reset_around(a, fa, rows, w & 0xffffu, w >> 16, lane);
    }
}

// ------------------------------------------------------------------------------------------
// Per-frame `consume` contract (framed.rs:127-157; adder_hip_frame_submit): hands one frame's events to the
// host without the host knowing their number in advance.  Blocks [0, copy_blocks) stream the events from the
// device buffer into page-locked host memory (16-byte stores over PCIe), the others compute the row chunks'
// offsets (Vec<Vec<Event>> structure, video.rs:677-691) and the result header.  d_offsets[0..1] = the frame's
// range in d_ev.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adder_frame_out_kernel(const AdderEventPod *__restrict__ d_ev,
                                                              const uint64_t *__restrict__ d_offsets, uint64_t cap,
                                                              AdderEventPod *__restrict__ h_ev, FrameResult *h_res,
                                                              uint32_t *__restrict__ h_chunks,
                                                              const uint32_t *__restrict__ status,
                                                              const uint32_t *__restrict__ counters, uint32_t row_begin,
                                                              uint32_t chunk_rows, uint32_t num_chunks,
                                                              uint32_t copy_blocks, uint32_t wire_rec) {
    // wire_rec != 0: d_ev is the frame's run of 9 / 11-byte records (the expansion wrote them, possibly into page-locked
    // host memory); the chunk search reads y out of the records
    const uint64_t begin = d_offsets[0];
    const uint64_t produced = d_offsets[1] - begin;
    const uint64_t n = produced < cap ? produced : cap;
    const AdderEventPod *const ev = d_ev + begin;
    if (blockIdx.x < copy_blocks) {
        const uint64_t dwords = n * 3ull, quads = dwords >> 2;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *src = reinterpret_cast<const u32x4 *>(ev);  // begin == 0 in every caller: 16-byte aligned
        u32x4 *dst = reinterpret_cast<u32x4 *>(h_ev);
        for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < quads; i += (uint64_t)copy_blocks * 256u)
            __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        if (blockIdx.x == 0 && threadIdx.x < (dwords & 3ull)) {
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(ev);
            reinterpret_cast<uint32_t *>(h_ev)[quads * 4 + threadIdx.x] = s32[quads * 4 + threadIdx.x];
        }
        return;
    }
    const uint32_t c = (blockIdx.x - copy_blocks) * 256u + threadIdx.x;
    if (c == 0) {
        h_res->produced = produced;
        h_res->status = *status | (produced > cap ? kStatusCapacity : 0u);
        h_res->new_features = counters ? counters[0] : 0u;
    }
    if (c > num_chunks) return;
    if (c == num_chunks) {
        h_chunks[c] = (uint32_t)n;
        return;
    }
    const uint32_t y0 = row_begin + c * chunk_rows;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        uint32_t y;
        if (wire_rec) {
            const uint8_t *const p = reinterpret_cast<const uint8_t *>(d_ev) + (begin + mid) * wire_rec;
            y = ((uint32_t)p[2] << 8) | p[3];
        } else {
            y = ev[mid].y;
        }
        if (y < y0) lo = mid + 1;
        else hi = mid;
    }
    h_chunks[c] = (uint32_t)lo;
}

// ------------------------------------------------------------------------------------------
// Multi-GPU: merge of the row bands' event streams (the reference's split is
// video.rs:677-691).  Rank r's stream is frame-major with offsets offs[r][0..T]; the merged stream
// is frame-major with, inside a frame, rank 0's events first, then rank 1's, ... = raster order.
// `stage` holds the ranks' streams back to back (rank r at stage_base[r] = sum of the totals of
// the lower ranks).  Two launches: the layout (one block), then the copy.
// work layout (uint64): [0, T] merged frame offsets, then dst[r][f] for r < world, f < T.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adder_merge_layout_kernel(const uint64_t *__restrict__ offs, uint32_t world,
                                                                 uint32_t T, uint64_t *__restrict__ work,
                                                                 uint64_t *__restrict__ merged_offsets,
                                                                 uint64_t merged_base) {
    // frame totals -> exclusive prefix over frames (serial per 256-frame tile; T is a few hundred)
    __shared__ uint64_t s_tile[256];
    __shared__ uint64_t s_carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        s_carry = 0;
        work[0] = 0;
        if (merged_offsets) merged_offsets[0] = merged_base;  // (a chunk of a longer stream: the events before it)
    }
    __syncthreads();
    for (uint32_t f0 = 0; f0 < T; f0 += 256u) {
        const uint32_t f = f0 + tid;
        uint64_t tot = 0;
        if (f < T)
            for (uint32_t r = 0; r < world; ++r) tot += offs[(size_t)r * (T + 1) + f + 1] - offs[(size_t)r * (T + 1) + f];
        s_tile[tid] = tot;
        __syncthreads();
        if (tid == 0) {
            uint64_t run = s_carry;
            for (uint32_t k = 0; k < 256u; ++k) {
                const uint64_t t = s_tile[k];
                s_tile[k] = run;  // exclusive
                run += t;
            }
            s_carry = run;
        }
        __syncthreads();
        if (f < T) {
            const uint64_t base = s_tile[tid];
            uint64_t before = 0;
            for (uint32_t r = 0; r < world; ++r) {
                work[(T + 1) + (size_t)r * T + f] = base + before;
                before += offs[(size_t)r * (T + 1) + f + 1] - offs[(size_t)r * (T + 1) + f];
            }
            work[f + 1] = base + before;
            if (merged_offsets) merged_offsets[f + 1] = merged_base + base + before;
        }
        __syncthreads();
    }
}

// blockIdx.y = r * T + f: copies rank r's events of frame f (12-byte records, dword granularity).
__global__ __launch_bounds__(256) void adder_merge_copy_kernel(const uint32_t *__restrict__ stage,
                                                               const uint64_t *__restrict__ offs, uint32_t world,
                                                               uint32_t T, const uint64_t *__restrict__ work,
                                                               uint32_t *__restrict__ out, uint64_t out_cap,
                                                               uint32_t *status) {
    // (rank, frame) pairs: blockIdx.y walks them (world * T may exceed the 65 535 rows a grid can have)
    for (uint32_t rf = blockIdx.y; rf < world * T; rf += gridDim.y) {
        const uint32_t r = rf / T, f = rf - r * T;
        uint64_t stage_base = 0;
        for (uint32_t k = 0; k < r; ++k) stage_base += offs[(size_t)k * (T + 1) + T] - offs[(size_t)k * (T + 1)];
        const uint64_t first = offs[(size_t)r * (T + 1)];  // (a rank's offsets need not start at 0: a chunk of its stream)
        const uint64_t a = offs[(size_t)r * (T + 1) + f] - first, e = offs[(size_t)r * (T + 1) + f + 1] - first;
        const uint64_t dst = work[(T + 1) + (size_t)r * T + f];
        uint64_t n = e - a;
        if (dst + n > out_cap) {
            if (threadIdx.x == 0 && blockIdx.x == 0) raise(status, kStatusCapacity);
            n = dst < out_cap ? out_cap - dst : 0;
        }
        const uint32_t *src = stage + (stage_base + a) * 3u;
        uint32_t *d = out + dst * 3u;
        const uint64_t nd = n * 3u;
        for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nd; i += (uint64_t)gridDim.x * 256u) d[i] = src[i];
    }
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

extern "C" hipError_t adder_launch_divtest(unsigned long long *d_bad, hipStream_t stream) {
    hipLaunchKernelGGL(adder_divtest_kernel, dim3((1u << 24) / 256u, 255), dim3(256), 0, stream, d_bad);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_frame(const BatchArgs *b, uint32_t f, uint32_t nb, uint32_t variant,
                                         uint32_t num_waves, uint32_t grid_cap, hipStream_t stream,
                                         const Lean1wArgs *wide) {
    const bool collapse = variant & 1u, abs_t = variant & 2u, generic = variant & 4u;
    const uint32_t S = (num_waves + kWavesPerBlock - 1) / kWavesPerBlock;  // step workgroups
    if (variant & 8u) {  // Mode::Continuous
        if (abs_t) hipLaunchKernelGGL((adder_cont_kernel<true>), dim3(S), dim3(kBlockThreads), 0, stream, b, f, nb);
        else hipLaunchKernelGGL((adder_cont_kernel<false>), dim3(S), dim3(kBlockThreads), 0, stream, b, f, nb);
        return hipGetLastError();
    }
    if (variant & 512u) {  // run records (the bounded Collapse regime at c_thresh 0, integer state)
        const uint32_t SG = grid_cap && grid_cap < S ? grid_cap : S;
        const uint32_t lazy = (variant & 2048u) ? 1u : 0u;  // (more launches of this batch follow: adder_hip_api.cpp lazy_state_bit)
        if (abs_t) hipLaunchKernelGGL((adder_rr_kernel<true>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb, lazy);
        else hipLaunchKernelGGL((adder_rr_kernel<false>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb, lazy);
        return hipGetLastError();
    }
    if (variant & 128u) {  // constant runs (the bounded Collapse regime at c_thresh 0)
        const uint32_t SG = grid_cap && grid_cap < S ? grid_cap : S;
        if (abs_t) hipLaunchKernelGGL((adder_cr_kernel<true>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb);
        else hipLaunchKernelGGL((adder_cr_kernel<false>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb);
        return hipGetLastError();
    }
    if (variant & 32u) {  // bounded Collapse step
        const uint32_t SG = grid_cap && grid_cap < S ? grid_cap : S;
        if (abs_t) hipLaunchKernelGGL((adder_cb_kernel<true>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb);
        else hipLaunchKernelGGL((adder_cb_kernel<false>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb);
        return hipGetLastError();
    }
    if (generic) {
        const uint32_t SG = grid_cap && grid_cap < S ? grid_cap : S;
        if (collapse) {
            if (abs_t) hipLaunchKernelGGL((adder_frame_kernel<true, true>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb);
            else hipLaunchKernelGGL((adder_frame_kernel<true, false>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb);
        } else {
            if (abs_t) hipLaunchKernelGGL((adder_frame_kernel<false, true>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb);
            else hipLaunchKernelGGL((adder_frame_kernel<false, false>), dim3(SG), dim3(kBlockThreads), 0, stream, b, f, nb);
        }
        return hipGetLastError();
    }
    if (!collapse) return hipErrorInvalidValue;  // the lean step is Collapse-only
    if (variant & 4096u)  // lean runs, packed bytes (adder_lp_kernels.hip): DeltaT batches of events or wire records
        return adder_launch_lp(b, f, nb, (variant & 2048u) ? 1u : 0u, num_waves, grid_cap, stream);
    if (variant & 256u) {  // lean runs (DeltaT, constant runs): every launch of the batch, whatever its length
        const uint32_t SR = grid_cap && grid_cap < S ? grid_cap : S;
        const uint32_t lazy = (variant & 2048u) ? 1u : 0u;  // (more launches of this batch follow: adder_hip_api.cpp lazy_state_bit)
        if (abs_t) hipLaunchKernelGGL((adder_lr_kernel<true>), dim3(SR), dim3(kBlockThreads), 0, stream, b, f, nb, lazy);
        else hipLaunchKernelGGL((adder_lr_kernel<false>), dim3(SR), dim3(kBlockThreads), 0, stream, b, f, nb, lazy);
        return hipGetLastError();
    }
#if ADDER_UNITS_PER_LANE == 2 && ADDER_LEAN1_WIDE
    const bool lean_log = variant & 64u;  // blocked batches: records in per-segment logs (every launch, also a chunk's 1-frame tail)
    if (nb == 1u && !lean_log && num_waves % 2u == 0u && (variant & 16u) && wide) {
        const uint32_t waves = (num_waves / 2u + kLean1wPairs - 1u) / kLean1wPairs;
        const uint32_t grid = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
        if (abs_t) hipLaunchKernelGGL((adder_lean1w_kernel<true>), dim3(grid), dim3(kBlockThreads), 0, stream, b, f, *wide);
        else hipLaunchKernelGGL((adder_lean1w_kernel<false>), dim3(grid), dim3(kBlockThreads), 0, stream, b, f, *wide);
        return hipGetLastError();
    }
#endif
    if (nb == 1u && !lean_log && num_waves % kLean1Segs == 0u) {
        const uint32_t grid = (num_waves / kLean1Segs + kWavesPerBlock - 1) / kWavesPerBlock;
        if (abs_t) hipLaunchKernelGGL((adder_lean1_kernel<true>), dim3(grid), dim3(kBlockThreads), 0, stream, b, f);
        else hipLaunchKernelGGL((adder_lean1_kernel<false>), dim3(grid), dim3(kBlockThreads), 0, stream, b, f);
        return hipGetLastError();
    }
    const uint32_t SL = grid_cap && grid_cap < S ? grid_cap : S;  // (only the blocked lean kernel walks: see its comment)
    if (lean_log) {
        if (abs_t) hipLaunchKernelGGL((adder_lean_kernel<true, true>), dim3(SL), dim3(kBlockThreads), 0, stream, b, f, nb);
        else hipLaunchKernelGGL((adder_lean_kernel<false, true>), dim3(SL), dim3(kBlockThreads), 0, stream, b, f, nb);
    } else {
        if (abs_t) hipLaunchKernelGGL((adder_lean_kernel<true, false>), dim3(SL), dim3(kBlockThreads), 0, stream, b, f, nb);
        else hipLaunchKernelGGL((adder_lean_kernel<false, false>), dim3(SL), dim3(kBlockThreads), 0, stream, b, f, nb);
    }
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

extern "C" hipError_t adder_launch_sink_layout(const uint64_t *all_offs, uint32_t world, uint32_t rank, uint32_t nf,
                                               uint64_t *file_pos, uint64_t *dest, uint64_t *merged_offs, hipStream_t stream) {
    hipLaunchKernelGGL(adder_sink_layout_kernel, dim3(1), dim3(64), 0, stream, all_offs, world, rank, nf, file_pos, dest,
                       merged_offs);
    return hipGetLastError();
}
extern "C" hipError_t adder_launch_wire_scatter(const AdderEventPod *ev, const uint64_t *offs, uint32_t nf, const uint64_t *dest,
                                                uint32_t rec, uint8_t *out, uint64_t out_cap, uint64_t header, uint32_t *status,
                                                uint32_t grid, hipStream_t stream, uint64_t src_cap_events) {
    if (nf == 0) return hipSuccess;
    hipLaunchKernelGGL(adder_wire_scatter_kernel, dim3(grid ? grid : 1u), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t *>(ev), offs, nf, dest, rec, out, out_cap, header, status, src_cap_events);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_scan(const BatchArgs *b, uint32_t f0, uint32_t nf, uint32_t num_waves, hipStream_t stream, uint32_t whole_batch,
                                        uint32_t rec_prefix, uint32_t chain) {
    const uint32_t ntiles = (num_waves + kScanTileWaves - 1u) / kScanTileWaves;  // (as the kernels work it out from the batch)
    if (ntiles > 1u) hipLaunchKernelGGL(adder_scan_tiles_kernel, dim3(nf * ntiles), dim3(kScanThreads), 0, stream, b, f0);
    hipLaunchKernelGGL(adder_scan_kernel, dim3(nf * ntiles), dim3(kScanThreads), 0, stream, b, f0,
                       ((whole_batch && f0 == 0u && nf == 1u) ? 1u : 0u) | (chain ? 2u : 0u), rec_prefix);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_publish(const BatchArgs *b, uint32_t num_frames, BatchResult *h, hipStream_t stream) {
    hipLaunchKernelGGL(adder_publish_kernel, dim3(1), dim3(64), 0, stream, b, num_frames, h);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_offsets(const BatchArgs *b, uint32_t f0, uint32_t nf, hipStream_t stream) {
    hipLaunchKernelGGL(adder_offsets_kernel, dim3(1), dim3(64), 0, stream, b, f0, nf);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_expand(const BatchArgs *b, uint32_t f0, uint32_t nf, uint32_t num_waves,
                                          uint32_t variant, uint32_t grid_cap, hipStream_t stream, const BatchArgs *host_b) {
    const uint32_t per_block = kWavesPerBlock * kExpandSegs;  // segments per block
    const uint32_t xblocks = (num_waves + per_block - 1) / per_block;
    constexpr uint32_t items = 1u;  // (2 / 4 / 8 work items per workgroup measured slower on every content: DESIGN appendix)
    uint32_t total = xblocks * nf;
    if (grid_cap && grid_cap < total) total = grid_cap;
    const dim3 grid(total);
    const bool abs_t = variant & 2u, generic = variant & 4u, continuous = variant & 8u, wire = variant & 1024u;
#define ADDER_X(F, A, W) hipLaunchKernelGGL((adder_expand_kernel<F, A, W>), grid, dim3(kBlockThreads), 0, stream, b, f0, xblocks, nf, items)
#define ADDER_XW(F, A) do { if (wire) ADDER_X(F, A, true); else ADDER_X(F, A, false); } while (0)
    if (continuous) ADDER_X(2, false, false);  // (its events are stored as they are decoded: no wire form)
    else if (variant & 512u) {  // run records in per-segment logs
        if (abs_t) ADDER_XW(4, true);
        else ADDER_XW(4, false);
    } else if (generic && (variant & 32u)) ADDER_XW(0, true);
    else if (generic) ADDER_XW(0, false);
    else if ((variant & 64u) && (variant & 256u)) {  // lean-runs records found through a run table (a band's packed records on root)
        if (abs_t) ADDER_XW(6, true);
        else ADDER_XW(6, false);
    } else if (variant & 64u) {
        if (abs_t) ADDER_XW(3, true);
        else ADDER_XW(3, false);
    } else if (variant & 4096u) {  // adder_lp_kernel's records (DeltaT): the expansion of adder_lp_kernels.hip
        if (host_b == nullptr || f0 % host_b->chunk != 0u || nf > host_b->chunk) return hipErrorInvalidValue;  // (a launch covers frames of one chunk)
        return adder_launch_lpx(b, host_b, f0, nf, wire ? ((variant & 8192u) ? 11u : 9u) : 12u, stream);
    } else if (variant & 256u) {
        if (abs_t) ADDER_XW(5, true);
        else ADDER_XW(5, false);
    } else if (abs_t) ADDER_XW(1, true);
    else ADDER_XW(1, false);
#undef ADDER_XW
#undef ADDER_X
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Records over the wire (multi-GPU gather): a band ships its PARKED RECORDS (0.35x the bytes of its events) and root
// expands them.  adder_log_pack_*: the lean record logs of one batch of <= chunk frames (it starts at slot 0) -> one
// packed buffer, every segment's used prefix behind the one before it; wofs rows then index the packed buffer.
// adder_band_layout_kernel (root): where each band's events of each frame go in the merged stream.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void adder_log_bases_kernel(const uint32_t *__restrict__ wcur, uint32_t num_waves,
                                                              uint32_t *__restrict__ pbase, uint64_t *__restrict__ total) {
    __shared__ uint32_t s_part[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    uint32_t run = 0u;
    for (uint32_t s0 = 0; s0 < num_waves; s0 += 1024u) {  // uniform trips
        const uint32_t sgm = s0 + tid;
        const uint32_t v = sgm < num_waves ? wcur[sgm] : 0u;
        const uint32_t incl = wave_inclusive_scan(v, lane);
        if (lane == 63u) s_part[wid] = incl;
        __syncthreads();
        uint32_t base = run, tot = 0u;
        for (uint32_t w = 0; w < 16u; ++w) {
            const uint32_t t = s_part[w];
            if (w < wid) base += t;
            tot += t;
        }
        if (sgm < num_waves) pbase[sgm] = base + incl - v;
        run += tot;
        __syncthreads();
    }
    if (tid == 0) *total = run;
}
// one wave per segment: its used log prefix -> packed + pbase[seg]; its column of the wofs rows += pbase[seg]
__global__ __launch_bounds__(kBlockThreads) void adder_log_pack_kernel(const uint8_t *__restrict__ logs, uint32_t log_cap,
                                                                      uint32_t rec_bytes, const uint32_t *__restrict__ wcur,
                                                                      const uint32_t *__restrict__ pbase, uint32_t num_waves,
                                                                      uint32_t nf, uint32_t *__restrict__ wofs_rows,
                                                                      uint8_t *__restrict__ packed, uint64_t packed_cap_bytes,
                                                                      uint32_t *status) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t sgm = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
    if (sgm >= num_waves) return;
    const uint32_t n = wcur[sgm], base = pbase[sgm];
    const uint32_t dwords = n * (rec_bytes / 4u);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(logs + (size_t)sgm * log_cap * rec_bytes);
    uint32_t *dst = reinterpret_cast<uint32_t *>(packed + (size_t)base * rec_bytes);
    if ((uint64_t)(base + n) * rec_bytes > packed_cap_bytes) {
        if (lane == 0u) raise(status, kStatusScratch);
    } else {
        for (uint32_t k = lane; k < dwords; k += kWave) dst[k] = src[k];
    }
    for (uint32_t f = lane; f < nf; f += kWave) wofs_rows[(size_t)f * num_waves + sgm] += base;
}
// Records in FIXED SLOTS (the lean-runs kernel's, park_offset) -> one packed buffer, frame-major: a wave takes the sixteen
// segments one expansion wave will read of a frame; a segment's run goes to the frame's base (the records of the frames
// before it) + the scan's record prefix of the segment (its wofs row, rec_prefix), and the row then holds the run's
// ABSOLUTE start -- the run table root's expansion indexes the packed buffer with (AdderBandRecords::d_runs).
__global__ __launch_bounds__(kBlockThreads) void adder_slot_pack_kernel(const BatchArgs *__restrict__ b, uint32_t nf, uint32_t rec_bytes,
                                                                       uint8_t *__restrict__ packed, uint64_t packed_cap_bytes,
                                                                       uint32_t *status) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t num_waves = b->base.num_waves;
    const uint32_t groups = (num_waves + 15u) / 16u;
    const uint32_t item = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
    if (item >= groups * nf) return;
    const uint32_t f = item / groups, seg0 = (item - f * groups) * 16u;
    const uint32_t slots = b->slots;
    // records of the batch's frames before f (the batch started at slot 0: frame f is slot f)
    uint32_t before = lane < f ? b->ftot_ring[slots + lane] : 0u;
#pragma unroll
    for (uint32_t o = kWave / 2; o > 0; o >>= 1) before += __shfl_xor(before, o, kWave);
    const ParkLayout lay = b->park_layout;
    const uint32_t *const wtot = b->wtot_ring + (size_t)f * num_waves;
    uint32_t *const wofs = b->wofs_ring + (size_t)f * num_waves;
    bool over = false;
    for (uint32_t q = 0; q < 16u && seg0 + q < num_waves; ++q) {  // uniform
        const uint32_t sgm = seg0 + q;
        const uint32_t n = wtot[sgm] >> 16;
        const uint32_t start = before + wofs[sgm];
        const uint32_t ndw = n * (rec_bytes / 4u);
        if ((uint64_t)(start + n) * rec_bytes > packed_cap_bytes) {
            over = true;
        } else {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(b->park_ring + park_offset(f, sgm, b->chunk, num_waves, b->park_bytes, lay));
            uint32_t *dst = reinterpret_cast<uint32_t *>(packed + (size_t)start * rec_bytes);
            for (uint32_t k = lane; k < ndw; k += kWave) dst[k] = src[k];
        }
        __builtin_amdgcn_wave_barrier();  // (every lane has read the prefix before lane 0 replaces it)
        if (lane == 0u) wofs[sgm] = start;
    }
    if (over && lane == 0u) raise(status, kStatusScratch);
}
extern "C" hipError_t adder_launch_slot_pack(const BatchArgs *b, uint32_t nf, uint32_t num_waves, uint32_t rec_bytes, uint8_t *packed,
                                             uint64_t packed_cap_bytes, uint32_t *status, hipStream_t stream) {
    const uint32_t items = ((num_waves + 15u) / 16u) * nf;
    hipLaunchKernelGGL(adder_slot_pack_kernel, dim3((items + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlockThreads), 0, stream, b, nf,
                       rec_bytes, packed, packed_cap_bytes, status);
    return hipGetLastError();
}
// One block.  offs[r] = band r's local frame offsets (nf + 1 entries, starting anywhere); merged_offsets[0 .. nf] (the
// caller's, [0] = merged_base on entry... written here) and dest[r * nf + f] = where band r's events of frame f start.
__global__ __launch_bounds__(kWave) void adder_band_layout_kernel(const uint64_t *const *__restrict__ offs, uint32_t n_bands,
                                                                 uint32_t nf, uint64_t merged_base,
                                                                 uint64_t *__restrict__ merged_offsets,
                                                                 uint64_t *__restrict__ dest) {
    const uint32_t lane = threadIdx.x;
    uint64_t run = merged_base;
    for (uint32_t f0 = 0; f0 < nf; f0 += kWave) {  // (one trip: nf <= 64)
        const uint32_t f = f0 + lane;
        uint64_t tot = 0ull;
        if (f < nf)
            for (uint32_t r = 0; r < n_bands; ++r) tot += offs[r][f + 1u] - offs[r][f];
        uint64_t incl = tot;
#pragma unroll
        for (uint32_t o = 1; o < kWave; o <<= 1) {
            const uint64_t e = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += e;
        }
        if (f < nf) {
            uint64_t pos = run + incl - tot;
            if (f == 0u) merged_offsets[0] = pos;
            for (uint32_t r = 0; r < n_bands; ++r) {
                dest[(size_t)r * nf + f] = pos;
                pos += offs[r][f + 1u] - offs[r][f];
            }
            merged_offsets[f + 1u] = run + incl;
        }
        run += __shfl(incl, kWave - 1, kWave);
    }
}
extern "C" hipError_t adder_launch_log_pack(const uint8_t *logs, uint32_t log_cap, uint32_t rec_bytes, const uint32_t *wcur,
                                            uint32_t *pbase, uint32_t num_waves, uint32_t nf, uint32_t *wofs_rows,
                                            uint8_t *packed, uint64_t packed_cap_bytes, uint64_t *d_total, uint32_t *status,
                                            hipStream_t stream) {
    hipLaunchKernelGGL(adder_log_bases_kernel, dim3(1), dim3(1024), 0, stream, wcur, num_waves, pbase, d_total);
    const uint32_t grid = (num_waves + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(adder_log_pack_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, logs, log_cap, rec_bytes, wcur, pbase,
                       num_waves, nf, wofs_rows, packed, packed_cap_bytes, status);
    return hipGetLastError();
}
extern "C" hipError_t adder_launch_band_layout(const uint64_t *const *offs, uint32_t n_bands, uint32_t nf, uint64_t merged_base,
                                               uint64_t *merged_offsets, uint64_t *dest, hipStream_t stream) {
    hipLaunchKernelGGL(adder_band_layout_kernel, dim3(1), dim3(kWave), 0, stream, offs, n_bands, nf, merged_base,
                       merged_offsets, dest);
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

extern "C" hipError_t adder_launch_frame_out(const AdderEventPod *d_ev, const uint64_t *d_offsets, uint64_t cap,
                                             AdderEventPod *h_ev, FrameResult *h_res, uint32_t *h_chunks,
                                             const uint32_t *status, const uint32_t *counters, uint32_t row_begin,
                                             uint32_t chunk_rows, uint32_t num_chunks, hipStream_t stream, uint32_t wire_rec) {
    // a slice of the chip keeps a x16 link busy (null: the wire scatter did the hand-over); ADDER_HIP_OUT_BLOCKS for A/Bs
    static const uint32_t want_blocks = [] { const char *e = getenv("ADDER_HIP_OUT_BLOCKS"); return e ? (uint32_t)atoi(e) : 128u; }();
    const uint32_t copy_blocks = h_ev ? (want_blocks ? want_blocks : 128u) : 0;
    const uint32_t chunk_blocks = (num_chunks + 1 + 255) / 256;
    hipLaunchKernelGGL(adder_frame_out_kernel, dim3(copy_blocks + chunk_blocks), dim3(256), 0, stream, d_ev, d_offsets,
                       cap, h_ev, h_res, h_chunks, status, counters, row_begin, chunk_rows, num_chunks, copy_blocks, wire_rec);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_features(const BatchArgs *b, uint32_t f, const FeatureArgs *fa, hipStream_t stream) {
    hipLaunchKernelGGL(adder_feature_kernel, dim3(512), dim3(kBlockThreads), 0, stream, b, f, *fa);
    return hipGetLastError();
}

extern "C" hipError_t adder_launch_feature_apply(const BatchArgs *b, const FeatureArgs *fa, const uint32_t *xy, uint32_t n,
                                                 hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(adder_feature_apply_kernel, dim3(64), dim3(kBlockThreads), 0, stream, b, *fa, xy, n);
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

extern "C" hipError_t adder_launch_merge(const adder::AdderEventPod *stage, const uint64_t *offs, uint32_t world, uint32_t T,
                                         uint64_t *work, adder::AdderEventPod *out, uint64_t out_cap,
                                         uint64_t *merged_offsets, uint64_t merged_base, uint32_t *status, hipStream_t stream) {
    if (world == 0 || T == 0) return hipSuccess;
    hipLaunchKernelGGL(adder_merge_layout_kernel, dim3(1), dim3(256), 0, stream, offs, world, T, work, merged_offsets,
                       merged_base);
    const uint64_t pairs = (uint64_t)world * T;
    hipLaunchKernelGGL(adder_merge_copy_kernel, dim3(64, (uint32_t)(pairs < 65535u ? pairs : 65535u)), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t *>(stage), offs, world, T, work,
                       reinterpret_cast<uint32_t *>(out), out_cap, status);
    return hipGetLastError();
}
