// adder_lp_kernels.hip -- the headline regime's frame kernel in PACKED BYTES (gfx950).
//
// adder_lp_kernel is adder_lr_kernel's step (adder_pixel.hpp: LEAN RUNS -- Collapse, delta_t_max <= time_spanned, c_thresh 0
// throughout, one integer time_spanned, DeltaT; video.rs:1318-1380 over event_pixel_tree.rs:213-287, 317-413 in that regime)
// with FOUR units per lane and the lane's four input bytes of a frame never taken apart (adder_pixel.hpp: LEAN RUNS,
// PACKED): one wave steps a PAIR of segments (256 units), the flush / event masks are bit 7 of the unit's byte, the frame's
// events are v_bcnt of three words, rho is the distance to the frame of the unit's last flush -- a unit that does not
// change costs nothing of its own.  Per frame and wave: one DPP scan of {records | events << 16} places the records and
// leaves both segments' totals (lanes 31 and 63, parked in LDS until the launch ends), the four byte positions put
// their records into the wave's LDS run under the flush mask, one coalesced store writes the run out.  The pair's records are CONTIGUOUS in the pair's two slots (segment 2p's then
// segment 2p + 1's: the expansion's format 7 reads one run per pair), FOUR bytes each: unit (8 bits) | base_val << 8 |
// input << 16 | min(rho', 255) << 24, a run longer than that in an escape word at the far end of the pair's slots
// (adder_pixel.hpp lp_park4).  Same resident planes, scan, offsets and ring as every other frame kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "adder_kernel_util.hpp"
#include "adder_kernels.h"
#include "adder_pixel.hpp"

namespace adder {

#ifndef ADDER_LP_GROUP
#define ADDER_LP_GROUP 8  // frames per staging group (two groups in LDS: 2 x 8 x 256 bytes per wave)
#endif
#ifndef ADDER_LP_WAVES_PER_SIMD
#define ADDER_LP_WAVES_PER_SIMD 8
#endif
constexpr uint32_t kLpGroup = ADDER_LP_GROUP;
constexpr uint32_t kLpInFrames = 2u * kLpGroup;
constexpr uint32_t kLpUnits = 4u;                    // units per lane
constexpr uint32_t kLpPairUnits = kWave * kLpUnits;  // 256: a pair of segments
static_assert(kWaveUnits == 128u && kLpPairUnits == 2u * kWaveUnits, "a wave steps two 128-unit segments");
static_assert(kLpGroup % 4u == 0u && kMaxFramesPerLaunch % kLpInFrames == 0u, "four frames of a pair per load instruction");

__device__ __forceinline__ uint32_t lp_bcnt(uint32_t x, uint32_t acc) { return (uint32_t)__builtin_popcount(x) + acc; }

template <bool FULL>
__device__ __forceinline__ void lp_frames(const BatchArgs *__restrict__ b, const FrameArgs &a, uint32_t nb, uint32_t pw,
                                          uint32_t lane, uint8_t *lds_in, uint32_t *lds_tot, uint32_t *rec_lds, bool lazy) {
    const float T = a.sc.time_spanned;
    const uint32_t spw = __builtin_amdgcn_readfirstlane(pw);
    const uint32_t sgw = spw * 2u;  // the pair's first segment
    const uint32_t u0 = spw * kLpPairUnits + lane * kLpUnits;
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
    const uint32_t wrap_at = chunk_u - 1u - ridx0;
    const uint32_t wrap_bytes = chunk_u * frame_stride_u;
    // the launch's input bytes go through the wave's LDS slice in groups of kLpGroup frames, one group ahead (adder_lr_kernel)
    uint32_t *const in_lds = reinterpret_cast<uint32_t *>(lds_in) + lane;  // [frame % kLpInFrames][lane]
    const uint8_t *const fr0 = uniform_ptr(b->frames) + (size_t)f0 * n_units_u;
    const bool direct = FULL && __builtin_amdgcn_readfirstlane(((n_units_u | (uint32_t)(uintptr_t)fr0) & 15u) == 0u);
    auto stage_issue = [&](uint32_t k0) {  // frames [k0, k0 + kLpGroup) of the launch -> their half of the slice
        const uint32_t half = (k0 / kLpGroup) & 1u;
        if (direct) {
            const uint8_t *const seg_in = fr0 + (size_t)spw * kLpPairUnits + (lane & 15u) * 16u;
#pragma unroll
            for (uint32_t g = 0; g < kLpGroup / 4u; ++g) {
                uint32_t k = k0 + g * 4u + (lane >> 4);
                k = k < nb ? k : nb - 1u;
                __builtin_amdgcn_global_load_lds((const ADDER_GLOBAL void *)(seg_in + (size_t)k * n_units_u),
                                                 (__attribute__((address_space(3))) void *)(lds_in + (half * (kLpGroup / 4u) + g) * 1024u), 16, 0,
                                                 ADDER_NT_INPUT ? 2 : 0);
            }
        } else {  // (ragged or unaligned planes: through registers; bytes beyond the plane read as zero)
#pragma unroll 1
            for (uint32_t q = 0; q < kLpGroup; ++q) {
                const uint32_t k = k0 + q;
                const uint8_t *const fr = fr0 + (size_t)(k < nb ? k : nb - 1u) * n_units_u;
                uint32_t w = 0u;
                if (u0 + kLpUnits <= n_units_u && ((n_units_u | (uint32_t)(uintptr_t)fr0) & 3u) == 0u) {
                    w = gload<uint32_t>(fr, u0);
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < kLpUnits; ++j)
                        if (u0 + j < n_units_u) w |= (uint32_t)gload<uint8_t>(fr, u0 + j) << (8u * j);
                }
                in_lds[(half * kLpGroup + q) * kWave] = w;
            }
        }
    };
    auto stage = [&](uint32_t i) {  // group i starts: its bytes have landed, the next group leaves
        __builtin_amdgcn_s_waitcnt(0x0f70);
        if (i != 0u && i + kLpGroup < nb) stage_issue(i + kLpGroup);
    };
    stage_issue(0u);
    if (kLpGroup < nb) stage_issue(kLpGroup);
    reinterpret_cast<uint2 *>(lds_tot)[lane] = make_uint2(0u, 0u);  // (frames without a flush leave their totals at zero)

    LpWord s;
    {
        const uint4 hv = gload_nt<uint4>(a.hdr, u0 * 4u);
        const float4 dv = gload_nt<float4>(a.dt0, u0 * 4u);
        const uint32_t hdrv[4] = {hv.x, hv.y, hv.z, hv.w};
        const float dtv[4] = {dv.x, dv.y, dv.z, dv.w};
        LrPx p[4];
        bool all_ok = true;
#pragma unroll
        for (uint32_t j = 0; j < kLpUnits; ++j) {
            bool ok;
            p[j] = lr_unpack<ScalarLanes>(hdrv[j], dtv[j], T, ok);
            all_ok = all_ok && (ok || (!FULL && u0 + j >= n_units_u));
        }
        if (!all_ok) raise(a.status, kStatusLeanRuns);
        lp_init(s, p);
    }
    uint32_t sel[kLpUnits];  // v_perm_b32 selectors: bytes {0, prev[j], vin[j], 0}
#pragma unroll
    for (uint32_t j = 0; j < kLpUnits; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(sel[j]) : "s"(0x0c00000cu | (j << 8) | ((4u + j) << 16)));
    const uint32_t unit0 = lane * kLpUnits;
    uint32_t *const tot_lds = lds_tot + (lane >> 5);
    const bool tot_lane = (lane & 31u) == 31u;

    // ESC: some unit of the pair carries a run in that can reach rho' = 255 within this launch (uniform; never on content that
    // keeps changing) -- its frames also look for escaping records and park their full rho' (adder_pixel.hpp lp_park4)
    auto frame = [&](uint32_t i, auto esc_tag) {
        constexpr bool ESC = decltype(esc_tag)::value;
        const uint32_t vin = in_lds[(i % kLpInFrames) * kWave];
        const uint32_t x = vin ^ s.prev;
        if (__builtin_amdgcn_ballot_w64(x != 0u) != 0ull) {
            const uint32_t base_w = s.prev;
            const LpMasks m = lp_step(s, vin);
            const uint32_t nrec = lp_bcnt(m.h, 0u);
            const uint32_t nev = lp_bcnt(m.c, lp_bcnt(m.b, lp_bcnt(m.a, 0u)));
            const uint32_t sw = nrec | (nev << 16);
            const uint32_t incl = wave_inclusive_scan_dpp(sw);
            if (tot_lane) tot_lds[i * 2u] = incl;
            // the records go through the wave's LDS run and leave as ONE contiguous store per 64 of them: four sparse
            // 8-byte stores per frame (one per byte position) cost the kernel 40 of its 100 us -- the memory pipeline takes
            // a store instruction at a time, whatever its lanes hold
            uint32_t pos = (incl - sw) & 0xffffu;
            uint32_t rho[kLpUnits];
            uint32_t n_esc = 0u, epos = 0u;
#pragma unroll
            for (uint32_t j = 0; j < kLpUnits; ++j) rho[j] = i - s.start[j];
            if (ESC) {
                uint32_t ne = 0u;
#pragma unroll
                for (uint32_t j = 0; j < kLpUnits; ++j) ne += ((m.h & (0x80u << (8u * j))) && rho[j] >= kLpRhoEsc) ? 1u : 0u;
                const uint32_t ei = wave_inclusive_scan_dpp(ne);
                epos = ei - ne;
                n_esc = (uint32_t)__builtin_amdgcn_readlane((int)ei, kWave - 1);
            }
#pragma unroll
            for (uint32_t j = 0; j < kLpUnits; ++j) {
                if (m.h & (0x80u << (8u * j))) {
                    const uint32_t w8 = __builtin_amdgcn_perm(vin, base_w, sel[j]) | unit0 | j;
                    uint32_t r8 = rho[j];
                    if (ESC) {
                        if (r8 >= kLpRhoEsc) {
                            rec_lds[kLpPairUnits + epos] = r8;
                            epos += 1u;
                            r8 = kLpRhoEsc;
                        }
                    }
                    rec_lds[pos] = w8 | (r8 << kLpRhoShift);
                    pos += 1u;
                    s.start[j] = i;
                }
            }
            const uint32_t n_rec = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1) & 0xffffu;
#if !defined(ADDER_DBG_LP_NOSTORE)  // (diagnostic A/B build: everything but the record stores)
            for (uint32_t k0 = 0; k0 < n_rec; k0 += kWave) {  // uniform trip count: one trip unless a quarter of the units flush
                const uint32_t k = k0 + lane;
                if (k < n_rec) gstore(seg, k * 4u, rec_lds[k]);
            }
            if (ESC) {
                for (uint32_t k0 = 0; k0 < n_esc; k0 += kWave) {  // (escape k: 4 (k + 1) bytes below the end of the pair's two slots)
                    const uint32_t k = k0 + lane;
                    if (k < n_esc) gstore(seg, 2u * park_bytes_u - 4u * (k + 1u), rec_lds[kLpPairUnits + k]);
                }
            }
#endif
        } else {
            lp_quiet(s);
        }
        seg += frame_stride_u;
        if (__builtin_expect(i == wrap_at, 0)) seg -= wrap_bytes;
    };

    // can a run carried into this launch reach rho' = 255 before the launch ends?  (start is -rho' at the launch's first frame)
    uint32_t carried = 0u;
#pragma unroll
    for (uint32_t j = 0; j < kLpUnits; ++j) carried = (0u - s.start[j]) > carried ? 0u - s.start[j] : carried;
    const bool esc = __builtin_amdgcn_ballot_w64(carried + nb > kLpRhoEsc) != 0ull;
    uint32_t i = 0u;
    while (i < nb) {
        stage(i);  // (i is a multiple of kLpGroup here)
        const uint32_t i_end = i + kLpGroup < nb ? i + kLpGroup : nb;
        if (esc) {
#pragma clang loop unroll(disable)
            for (; i < i_end; ++i) frame(i, std::true_type{});
        } else {
#pragma clang loop unroll(disable)
            for (; i < i_end; ++i) frame(i, std::false_type{});
        }
    }

    // the frames' totals: lane f holds frame f's {records | events << 16} up to lane 31 and up to lane 63
    if (lane < nb) {
        const uint2 t = reinterpret_cast<const uint2 *>(lds_tot)[lane];
        const uint32_t ta = t.x, tb = t.y - t.x;
        uint32_t sl = slot0 + lane;
        sl = sl >= slots_u ? sl - slots_u : sl;
        // wtot = events | records << 16 of each segment
        gstore<uint2>(uniform_ptr(b->wtot_ring), (sl * num_waves_u + sgw) * 4u,
                      make_uint2((ta >> 16) | (ta << 16), (tb >> 16) | (tb << 16)));
    }
    LrPx q[kLpUnits];
    uint32_t rmax = 0u;
#pragma unroll
    for (uint32_t j = 0; j < kLpUnits; ++j) {
        q[j] = lp_final(s, j, nb);
        rmax = q[j].rho > rmax ? q[j].rho : rmax;
    }
    report_run_max(b, rmax, lane);
    constexpr bool NTS = ADDER_NT_STATE != 0;
    auto put4 = [&](void *plane, const uint32_t (&v)[4]) {
        const uint4 x = make_uint4(v[0], v[1], v[2], v[3]);
        if (NTS) gstore_nt<uint4>(plane, u0 * 4u, x);
        else gstore<uint4>(plane, u0 * 4u, x);
    };
    if (lazy) {  // another launch of this batch follows: only what lr_unpack reads (header, delta_t)
        uint32_t hdrv[4], dv[4];
#pragma unroll
        for (uint32_t j = 0; j < kLpUnits; ++j) {
            hdrv[j] = hdr_make(q[j].base, 0u, q[j].rho != 0u ? 1u : 0u, q[j].base != 0u);
            dv[j] = __float_as_uint(q[j].base != 0u ? fmul((float)q[j].rho, T) : 0.0f);
        }
        put4(a.hdr, hdrv);
        put4(a.dt0, dv);
        return;
    }
    uint32_t hdrv[4], iv[4], dv[4], bv[4];
#pragma unroll
    for (uint32_t j = 0; j < kLpUnits; ++j) {
        float fi, fd, fb;
        hdrv[j] = lr_pack<ScalarLanes>(q[j], T, fi, fd, fb);
        iv[j] = __float_as_uint(fi);
        dv[j] = __float_as_uint(fd);
        bv[j] = __float_as_uint(fb);
    }
    put4(a.hdr, hdrv);
    put4(a.integ0, iv);
    put4(a.dt0, dv);
    put4(a.bdt0, bv);
}

__global__ __launch_bounds__(kBlockThreads, ADDER_LP_WAVES_PER_SIMD) void adder_lp_kernel(const BatchArgs *__restrict__ b,
                                                                                         uint32_t f, uint32_t nb, uint32_t lazy) {
    const FrameArgs a = frame_args(b, f);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kWavesPerBlock][kLpInFrames * kLpPairUnits];
    __shared__ __attribute__((aligned(8))) uint32_t s_tot[kWavesPerBlock][2u * kMaxFramesPerLaunch];
    __shared__ uint32_t s_rec[kWavesPerBlock][2u * kLpPairUnits];  // a frame's records of the pair in order, then its escape words
    timeline_mark(b, 0u, f, false);
    const uint32_t num_pairs = a.num_waves / 2u;  // (num_waves is a multiple of kExpandSegs)
    for (uint32_t pw = blockIdx.x * kWavesPerBlock + tid / kWave; pw < num_pairs; pw += gridDim.x * kWavesPerBlock) {
        const bool full = __builtin_amdgcn_readfirstlane(pw * kLpPairUnits + kLpPairUnits <= a.n_units);
        if (full) lp_frames<true>(b, a, nb, pw, lane, s_in[tid / kWave], s_tot[tid / kWave], s_rec[tid / kWave], lazy != 0u);
        else lp_frames<false>(b, a, nb, pw, lane, s_in[tid / kWave], s_tot[tid / kWave], s_rec[tid / kWave], lazy != 0u);
    }
    timeline_mark(b, 0u, f, true);
}

// ------------------------------------------------------------------------------------------
// K2 for adder_lp_kernel's records (format 7): adder_lpx_kernel<REC>.  A wave takes kLpxPairs consecutive pairs (16
// segments, 2048 units) of ONE frame -- their events are contiguous in the stream.
//   1. the pairs' runs of 4-byte records (one coalesced load per pair, all in flight at once) are unparked into ONE dense run
//      of {rho' | pair << 28, word} in LDS: the decode rounds below then run on 64 records each, whatever the pairs held
//      (a round per pair left 40 of 64 lanes idle);
//   2. a round decodes 64 records (lr_decode8_tab: event A worked out from (base_val, rho), C from the 256-word table), a
//      DPP scan of the records' event counts places the events, and every event is written into the staging buffer in its
//      FINAL bytes -- REC = 9 / 11: the raw sink's record (RawOutput::ingest_event, raw/stream.rs:101-120: bincode fixint
//      big-endian {x u16, y u16, [0x01, c,] d u8, t u32}), byte-aligned LDS stores (gfx950 takes them); REC = 12: the AdderEvent;
//   3. the staging buffer sits at the 16-byte phase of its destination, so a flush is 16-byte LDS reads -> 16-byte global
//      stores, a kilobyte per instruction, and single bytes for the <= 15 + 15 bytes the wave shares with its neighbours' blocks.
// No conversion pass, no per-pair rounds, one copy of every loop: a tenth of adder_expand_kernel<5>'s code.
// ------------------------------------------------------------------------------------------
#ifndef ADDER_LPX_REC_CAP
#define ADDER_LPX_REC_CAP 384  // records unparked per batch of pairs (>= 256: one pair's worst case)
#endif
#ifndef ADDER_LPX_STAGE_EVENTS
#define ADDER_LPX_STAGE_EVENTS 512  // events the staging buffer holds (>= 192 + what a flush should carry)
#endif
constexpr uint32_t kLpxPairs = kExpandSegs / 2u;
constexpr uint32_t kLpxRecCap = ADDER_LPX_REC_CAP;
constexpr uint32_t kLpxStageEvents = ADDER_LPX_STAGE_EVENTS;
static_assert(kLpxPairs == 8u && kLpxRecCap >= kLpPairUnits && kLpxStageEvents >= 4u * kWave, "sizes the loops below assume");

template <uint32_t REC>
__device__ __forceinline__ void lpx_put(uint8_t *p, uint32_t xy, uint32_t xyw, uint32_t c, uint32_t d, uint32_t t) {
    if constexpr (REC == 12u) {
        const uint32_t w[3] = {xy, c | (d << 8), t};
        __builtin_memcpy(p, w, 12);
    } else {
        const uint32_t tb = __builtin_amdgcn_perm(0u, t, 0x00010203u);  // t3 t2 t1 t0
        if constexpr (REC == 9u) {
            const uint32_t w[2] = {xyw, d | (tb << 8)};
            __builtin_memcpy(p, w, 8);
            p[8] = (uint8_t)(tb >> 24);
        } else {
            const uint32_t w[2] = {xyw, 1u | (c << 8) | (d << 16) | (tb << 24)};
            __builtin_memcpy(p, w, 8);
            const uint16_t m = (uint16_t)(tb >> 8);
            __builtin_memcpy(p + 8, &m, 2);
            p[10] = (uint8_t)(tb >> 24);
        }
    }
}

template <uint32_t REC>
__device__ __forceinline__ void lpx_wave(const BatchArgs *__restrict__ b, uint32_t f, uint32_t slot, uint32_t cir, uint32_t seg0,
                                         uint32_t lane, uint8_t *stage, uint2 *rec_lds, const uint32_t *tab_c) {
    const uint32_t num_waves = __builtin_amdgcn_readfirstlane(b->base.num_waves);
    const uint32_t park_bytes = __builtin_amdgcn_readfirstlane(b->park_bytes);
    const uint32_t chunk_frames = __builtin_amdgcn_readfirstlane(b->chunk);
    const ParkLayout lay = park_layout_u(b);
    const uint32_t pair_stride = 2u * lay.seg_stride;  // (the wave's 16 segments lie in one group: a constant stride apart)
    // park_offset() with the launch's chunk-in-ring and frame slot given (the host knows both: no division here)
    const uint32_t fi = (slot - cir * chunk_frames + (seg0 >> lay.rot_shift)) & lay.rot_mask;
    const uint32_t group = seg0 >> lay.group_shift;
    const uint8_t *const park = uniform_ptr(b->park_ring) + (size_t)cir * num_waves * chunk_frames * park_bytes +
                                (size_t)group * lay.group_stride + (size_t)fi * lay.frame_stride +
                                (size_t)(seg0 - (group << lay.group_shift)) * lay.seg_stride;
    const uint32_t *const wtot = uniform_ptr(b->wtot_ring) + (size_t)slot * num_waves + seg0;
    const uint32_t *const wpref = uniform_ptr(b->wpref_ring) + (size_t)slot * num_waves + seg0;
    // one round trip: the segments' totals, the events in front of them, the frame's place in the stream
    uint32_t my_tot = 0u;
    if (lane < kExpandSegs) my_tot = gload<uint32_t>(wtot, lane * 4u);
    const uint32_t pref0 = gload<uint32_t>(wpref, 0u);
    const uint64_t fo = b->base.frame_offsets[f];
    if (__builtin_amdgcn_ballot_w64((my_tot & 0xffffu) != 0u) == 0ull) return;  // quiet content: sixteen empty segments
    // records of pair p = of segments 2p and 2p + 1 (lanes 2p, 2p + 1 hold them: a quad permute adds the neighbour's)
    const uint32_t recs = my_tot >> 16;
    const uint32_t pair_recs = recs + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)recs, 0xb1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    uint32_t np[kLpxPairs];
#pragma unroll
    for (uint32_t p = 0; p < kLpxPairs; ++p) np[p] = (uint32_t)__builtin_amdgcn_readlane((int)pair_recs, 2 * p);
    // every pair's first 64 records, asked for at once
    uint32_t first[kLpxPairs];
#pragma unroll
    for (uint32_t p = 0; p < kLpxPairs; ++p) {
        first[p] = 0u;
        if (lane < np[p]) first[p] = gload_rec<uint32_t>(park + (size_t)p * pair_stride, lane * 4u);
    }
    // geometry: (row, offset in row) of the wave's first unit; a unit of the wave lies at most 2047 behind it
    const uint32_t rowlen = __builtin_amdgcn_readfirstlane(b->base.rowlen);
    const uint32_t channels = __builtin_amdgcn_readfirstlane(b->base.channels);
    const uint32_t row_begin = __builtin_amdgcn_readfirstlane(b->base.row_begin);
    const float inv_row = 1.0f / (float)rowlen;
    uint32_t y0;
    {   // seg0 * 128 / rowlen: a float estimate (units stay below 2^26), fixed either way
        const uint32_t u = seg0 * kWaveUnits;
        uint32_t q = (uint32_t)((float)u * inv_row);
        q -= (q != 0u && q * rowlen > u) ? 1u : 0u;
        q -= (q != 0u && q * rowlen > u) ? 1u : 0u;
        q += (q + 1u) * rowlen <= u ? 1u : 0u;
        q += (q + 1u) * rowlen <= u ? 1u : 0u;
        y0 = __builtin_amdgcn_readfirstlane(q);
    }
    const uint32_t rem0 = seg0 * kWaveUnits - y0 * rowlen;
    const uint32_t wraps = rowlen >= kLpxPairs * kLpPairUnits ? 1u : rowlen >= kLpxPairs * kLpPairUnits / 2u ? 2u : 0u;  // (0: divide)
    const uint32_t rt_u32 = __builtin_amdgcn_readfirstlane(f32_as_u32(b->ftab[f].running_t));  // t of D_EMPTY
    const float T = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b->base.sc.time_spanned)));
    const uint64_t out_cap = b->base.out_cap;
    uint8_t *const out = reinterpret_cast<uint8_t *>(uniform_ptr(b->base.out));
    uint64_t gpos = (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(fo >> 32)) << 32) |
                     __builtin_amdgcn_readfirstlane((uint32_t)fo)) + __builtin_amdgcn_readfirstlane(pref0);
    uint32_t fill = 0u;  // events staged (uniform)
    uint32_t phase = (uint32_t)((uintptr_t)out + gpos * REC) & 15u;  // the staging buffer's first byte within its 16-byte block
    bool dropped = false;
    constexpr uint32_t CAPE = kLpxStageEvents;

    auto flush = [&]() {  // staged events -> stream; events past the caller's capacity are dropped and reported
        const uint64_t room64 = gpos < out_cap ? out_cap - gpos : 0ull;
        const uint32_t n = (uint64_t)fill <= room64 ? fill : (uint32_t)room64;
        dropped = dropped || n != fill;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n) {
            uint8_t *const dst = out + gpos * REC - phase;  // 16-byte aligned: LDS byte L of the buffer goes to dst + L
            const uint32_t end = phase + n * REC;
            const uint32_t kb = phase ? 1u : 0u, ke = end >> 4;
            for (uint32_t k = kb + lane; k < ke; k += kWave) {
                uint4 v;
                __builtin_memcpy(&v, (const uint8_t *)__builtin_assume_aligned(stage + 16u * k, 16), 16);
                gstore_ev<uint4>(dst, 16u * k, v);
            }
            // the bytes in front of the first whole block and behind the last one: shared with the neighbouring waves' blocks
            const uint32_t head_n = phase ? (end < 16u ? end : 16u) - phase : 0u;
            const uint32_t tail_lo = (end > 16u || phase == 0u) ? (ke << 4) : end;
            const uint32_t tail_n = end - tail_lo;
            if (lane < 32u) {
                const uint32_t t = lane & 15u;
                const bool is_tail = lane >= 16u;
                if (t < (is_tail ? tail_n : head_n)) {
                    const uint32_t L = (is_tail ? tail_lo : phase) + t;
                    gstore<uint8_t>(dst, L, stage[L]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        gpos += fill;
        phase = (uint32_t)((uintptr_t)out + gpos * REC) & 15u;
        fill = 0u;
    };

    uint32_t pstart = 0u;
    while (pstart < kLpxPairs) {  // batches of pairs whose records fit the LDS run (one batch unless the content is dense)
        uint32_t pe = pstart, R = 0u;
#pragma unroll
        for (uint32_t p = 0; p < kLpxPairs; ++p)
            if (p >= pstart && p == pe && R + np[p] <= kLpxRecCap) {
                R += np[p];
                pe = p + 1u;
            }
        // 1. unpark the batch's runs into one dense run of {rho' | pair << 28, word}.  (The record's rho' byte is taken apart
        // here; an escaping record -- rho' = 255: rare -- is patched with its full rho' behind the plain copy.)
        uint32_t rb = 0u;
        uint64_t esc_any = 0ull;
#pragma unroll
        for (uint32_t p = 0; p < kLpxPairs; ++p) {
            if (p >= pstart && p < pe && np[p] != 0u) {  // uniform
                const uint32_t w4 = first[p];
                if (lane < np[p]) rec_lds[rb + lane] = make_uint2((w4 >> kLpRhoShift) | (p << 28), w4);
                esc_any |= __builtin_amdgcn_ballot_w64(lp_escapes(w4));
                if (__builtin_expect(np[p] > kWave, 0)) {  // (more than a quarter of the pair's units flushed)
                    const uint8_t *const pp = park + (size_t)p * pair_stride;
                    for (uint32_t l0 = kWave; l0 < np[p]; l0 += kWave) {
                        const uint32_t idx = l0 + lane;
                        const uint32_t w = idx < np[p] ? gload_rec<uint32_t>(pp, idx * 4u) : 0u;
                        if (idx < np[p]) rec_lds[rb + idx] = make_uint2((w >> kLpRhoShift) | (p << 28), w);
                        esc_any |= __builtin_amdgcn_ballot_w64(lp_escapes(w));
                    }
                }
                rb += np[p];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__builtin_expect(esc_any != 0ull, 0)) {
            // escapes: the k-th escaping record of a pair's run keeps its full rho' 4 (k + 1) bytes below the end of the pair's slots
            uint32_t qb = 0u;
#pragma unroll 1
            for (uint32_t p = pstart; p < pe; ++p) {
                const uint32_t n_p = (uint32_t)__builtin_amdgcn_readlane((int)pair_recs, 2 * p);
                const uint8_t *const pp = park + (size_t)p * pair_stride;
                uint32_t esc_before = 0u;
                for (uint32_t l0 = 0; l0 < n_p; l0 += kWave) {
                    const uint32_t idx = l0 + lane;
                    uint2 r = make_uint2(0u, 0u);
                    if (idx < n_p) r = rec_lds[qb + idx];
                    const bool e = idx < n_p && lp_escapes(r.y);
                    const uint64_t em = __builtin_amdgcn_ballot_w64(e);
                    if (em != 0ull) {
                        const uint32_t rank = esc_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                        if (e) rec_lds[qb + idx] = make_uint2(gload_rec<uint32_t>(pp, pair_stride - 4u * (rank + 1u)) | (p << 28), r.y);
                        esc_before += (uint32_t)__popcll(em);
                    }
                }
                qb += n_p;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // 2. dense rounds
#pragma clang loop unroll(disable)
        for (uint32_t r0 = 0; r0 < R; r0 += kWave) {
            if (fill + 3u * kWave > CAPE) flush();
            const bool valid = r0 + lane < R;
            uint2 rec = make_uint2(0u, 0u);
            if (valid) rec = rec_lds[r0 + lane];
            const uint32_t w8 = rec.y;
            const LeanEvents e = lr_decode8_tab(lp_rho(rec.x & 0x0fffffffu, w8), w8, T, rt_u32, nullptr, tab_c);  // (a zero record: no events)
            const uint32_t n = (e.a ? 1u : 0u) + (e.b ? 1u : 0u) + (e.c ? 1u : 0u);
            const uint32_t incl = wave_inclusive_scan_dpp(n);
            // coordinates: the unit counted from the wave's first
            uint32_t rem = rem0 + ((rec.x >> 28) << 8) + (w8 & 0xffu);
            uint32_t y = y0 + row_begin;
            if (wraps != 0u) {
                const bool w1 = rem >= rowlen;
                rem -= w1 ? rowlen : 0u;
                y += w1 ? 1u : 0u;
                if (wraps == 2u) {
                    const bool w2 = rem >= rowlen;
                    rem -= w2 ? rowlen : 0u;
                    y += w2 ? 1u : 0u;
                }
            } else {  // narrow planes: a quotient estimate, one step either way
                uint32_t q = (uint32_t)((float)rem * inv_row);
                q -= q * rowlen > rem ? 1u : 0u;
                q += (q + 1u) * rowlen <= rem ? 1u : 0u;
                rem -= q * rowlen;
                y += q;
            }
            uint32_t x = rem, c = 0xffu;
            if (channels == 3u) {
                x = (uint32_t)(((uint64_t)rem * 0xAAAAAAABull) >> 33);  // rem / 3
                c = rem - 3u * x;
            }
            const uint32_t xy = x | (y << 16);
            const uint32_t xyw = __builtin_amdgcn_perm(0u, xy, 0x02030001u);  // x_hi x_lo y_hi y_lo
            uint8_t *p = stage + phase + (fill + incl - n) * REC;
            fill += (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
            if (e.a) {
                lpx_put<REC>(p, xy, xyw, c, e.da, e.ta);
                p += REC;
            }
            if (e.b) {
                lpx_put<REC>(p, xy, xyw, c, kDEmpty, e.tb);
                p += REC;
            }
            if (e.c) lpx_put<REC>(p, xy, xyw, c, e.dc, e.tc);
        }
        pstart = pe;
    }
    flush();
    if (dropped) raise(b->base.status, kStatusCapacity);
}

// grid: (blocks of 64 segments, frames of the launch); slot0 / cir: the first frame's slot of the scratch ring and its chunk
// in the ring (a launch covers frames of ONE chunk, consecutive slots)
template <uint32_t REC>
__global__ __launch_bounds__(kBlockThreads) void adder_lpx_kernel(const BatchArgs *__restrict__ b, uint32_t f0, uint32_t slot0,
                                                                  uint32_t cir) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWavesPerBlock][kLpxStageEvents * REC + 16u];
    __shared__ __attribute__((aligned(8))) uint2 s_rec[kWavesPerBlock][kLpxRecCap];
    __shared__ uint32_t s_tab_c[256];  // event C by input byte (lr_build_tab)
    timeline_mark(b, 3u, f0, false);
    const uint32_t f = f0 + blockIdx.y, slot = slot0 + blockIdx.y, xblock = blockIdx.x;
    // (the table's words are asked for first and parked after the test: the two loads share one round trip)
    const uint32_t tab_word = gload<uint32_t>(uniform_ptr(b->lr_tab), (256u * kLrTabRuns + threadIdx.x) * 4u);
    const uint32_t *const ft = b->ftot_ring;  // a frame without a single event has nothing to expand
    if (ft != nullptr && __builtin_amdgcn_readfirstlane(ft[slot]) == 0u) return;
    s_tab_c[threadIdx.x] = tab_word;
    __syncthreads();
    const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t seg0 = __builtin_amdgcn_readfirstlane((xblock * kWavesPerBlock + wid) * kExpandSegs);
    if (seg0 < __builtin_amdgcn_readfirstlane(b->base.num_waves))
        lpx_wave<REC>(b, f, slot, cir, seg0, threadIdx.x & (kWave - 1u), s_stage[wid], s_rec[wid], s_tab_c);
    timeline_mark(b, 3u, f0, true);
}

}  // namespace adder

using namespace adder;

extern "C" hipError_t adder_launch_lp(const BatchArgs *b, uint32_t f, uint32_t nb, uint32_t lazy, uint32_t num_waves,
                                      uint32_t grid_cap, hipStream_t stream) {
    const uint32_t pairs = num_waves / 2u;
    uint32_t grid = (pairs + kWavesPerBlock - 1u) / kWavesPerBlock;
    if (grid_cap && grid_cap < grid) grid = grid_cap;
    hipLaunchKernelGGL(adder_lp_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, b, f, nb, lazy);
    return hipGetLastError();
}


extern "C" hipError_t adder_launch_lpx(const BatchArgs *b, uint32_t f0, uint32_t nf, uint32_t num_waves, uint32_t rec,
                                       uint32_t slot0, uint32_t cir, hipStream_t stream) {
    const uint32_t per_block = kWavesPerBlock * kExpandSegs;  // segments per block
    const dim3 grid((num_waves + per_block - 1u) / per_block, nf);
    if (rec == 9u) hipLaunchKernelGGL((adder_lpx_kernel<9u>), grid, dim3(kBlockThreads), 0, stream, b, f0, slot0, cir);
    else if (rec == 11u) hipLaunchKernelGGL((adder_lpx_kernel<11u>), grid, dim3(kBlockThreads), 0, stream, b, f0, slot0, cir);
    else hipLaunchKernelGGL((adder_lpx_kernel<12u>), grid, dim3(kBlockThreads), 0, stream, b, f0, slot0, cir);
    return hipGetLastError();
}
