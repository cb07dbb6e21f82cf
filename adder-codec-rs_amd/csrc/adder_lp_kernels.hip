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
// segment 2p + 1's: adder_lpx_kernel reads one run per pair), FOUR bytes each: unit (8 bits) | base_val << 8 |
// input << 16 | min(rho', 255) << 24, a run longer than that in an escape word at the far end of the pair's slots
// (adder_pixel.hpp lp_park4).  Same resident planes, scan, offsets and ring as every other frame kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
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
static_assert(kLpGroup % 4u == 0u && kMaxFramesPerLaunch % kLpInFrames == 0u && kLpGroup >= 4u, "four frames of a pair per load instruction");

__device__ __forceinline__ uint32_t lp_bcnt(uint32_t x, uint32_t acc) { return (uint32_t)__builtin_popcount(x) + acc; }

// One byte position of a frame (units 4 lane + J): the lanes whose unit flushed (bit 7 of byte J of h) put their record word
// {unit | base_val << 8 | input << 16 | (i - start) << 24} at LDS address `addr`, step the address and restart the unit's run.
// By hand: half of the kernel's vector instructions are these four blocks, and the compiler's form of one is eleven
// instructions (two copies of the new start, a temporary for the address) where seven do.  rho' = i - start < 255 here (lp_frames).
#define ADDER_LP_SLOT(J)                                                                                                      \
    asm volatile("v_cmp_ne_u32_sdwa vcc, %[h], %[z] src0_sel:BYTE_" #J " src1_sel:DWORD\n\t"                                   \
                 "s_and_saveexec_b64 %[sv], vcc\n\t"                                                                          \
                 "v_sub_u32_sdwa %[t], %[i], %[st] dst_sel:BYTE_3 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"    \
                 "v_perm_b32 %[p], %[vin], %[bw], %[sel]\n\t"                                                                 \
                 "v_or3_b32 %[t], %[t], %[p], %[un]\n\t"                                                                      \
                 "ds_write_b32 %[a], %[t]\n\t"                                                                                \
                 "v_add_u32 %[a], 4, %[a]\n\t"                                                                                \
                 "v_mov_b32 %[st], %[i]\n\t"                                                                                  \
                 "s_mov_b64 exec, %[sv]"                                                                                       \
                 : [a] "+v"(addr), [st] "+v"(s.start[J]), [t] "=&v"(slot_t), [p] "=&v"(slot_p), [sv] "=&s"(slot_sv)             \
                 : [h] "v"(m.h), [z] "v"(zero_v), [i] "s"(i), [vin] "v"(vin), [bw] "v"(base_w), [sel] "v"(sel[J]), [un] "v"(unitj[J]) \
                 : "vcc", "scc", "memory")  /* (s_and_saveexec writes SCC) */

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
    // (A COUNTED wait here -- vmcnt(n), n = the record stores issued behind the group's loads, so that a wave does not sit out its
    // last stores' acknowledgements -- measured 84 against 86 us per launch: inside the noise, and it leans on loads and stores
    // retiring in issue order.  The plain wait stays.)
    uint32_t since = 0u;  // (record stores issued in this group: statistics only)
    auto stage = [&](uint32_t i) {  // group i starts: its bytes have landed, the next group leaves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (i != 0u && i + kLpGroup < nb) stage_issue(i + kLpGroup);
        since = 0u;
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
    uint32_t unitj[kLpUnits];
#pragma unroll
    for (uint32_t j = 0; j < kLpUnits; ++j) asm volatile("v_or_b32 %0, %1, %2" : "=v"(unitj[j]) : "v"(unit0), "s"(j));  // (kept in registers)
    uint32_t zero_v;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    const uint32_t rec_lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)rec_lds;
    uint32_t *const tot_lds = lds_tot + (lane >> 5);
    const bool tot_lane = (lane & 31u) == 31u;

    // ESC: some unit of the pair carries a run in that can reach rho' = 255 within this launch (uniform; never on content that
    // keeps changing) -- its frames also look for escaping records and park their full rho' (adder_pixel.hpp lp_park4)
    // The frame's records leave ONE FRAME LATER (pend_n of them, to pend_seg): the LDS run is read back at the top of the next
    // frame and stored behind its packed step, so no frame waits for its own LDS round trip.
    uint32_t act_w = 0u;  // bit 7 of byte j: unit 4 lane + j lies inside the plane
#pragma unroll
    for (uint32_t j = 0; j < kLpUnits; ++j) act_w |= (FULL || u0 + j < n_units_u) ? (0x80u << (8u * j)) : 0u;
    uint32_t pend_n = 0u;
    uint8_t *pend_seg = seg;
    const uint32_t lane4 = lane * 4u;
    auto frame = [&](uint32_t i, const uint32_t *in_row, auto esc_tag) {  // in_row: the frame's row of the LDS slice
        constexpr bool ESC = decltype(esc_tag)::value;
        const uint32_t vin = *in_row;
        const uint32_t pend_w = rec_lds[lane];  // (every lane, every frame: no branch around the LDS read; lanes >= pend_n read what is not stored)
        const uint32_t x = vin ^ s.prev;
        const bool busy = __builtin_amdgcn_ballot_w64(x != 0u) != 0ull;
        LpMasks m{0u, 0u, 0u, 0u};
        const uint32_t base_w = s.prev;
        if (busy) m = lp_step(s, vin);
        else lp_quiet(s);
        if (!FULL) {  // padding units (beyond the plane) are stepped freely, like every frame kernel steps them -- and leave nothing
            m.h &= act_w;
            m.a &= act_w;
            m.b &= act_w;
            m.c &= act_w;
        }
#if !defined(ADDER_DBG_LP_NOSTORE)
        if (pend_n != 0u) {
            if (lane < pend_n) asm volatile("global_store_dword %0, %1, %2" : : "v"(lane4), "v"(pend_w), "s"(pend_seg) : "memory");
            since += 1u;
            pend_n = 0u;
        }
#else
        pend_n = 0u;
#endif
        if (busy) {
            const uint32_t nrec = lp_bcnt(m.h, 0u);
            const uint32_t nev = lp_bcnt(m.c, lp_bcnt(m.b, lp_bcnt(m.a, 0u)));
            // {events | 4 x records << 16}: the scan's upper half is the byte offset of the lane's first record in the LDS run
            const uint32_t sw = nev | (nrec << 18);
            const uint32_t incl = wave_inclusive_scan_dpp(sw);
            if (tot_lane) tot_lds[i * 2u] = incl;
            // the records go through the wave's LDS run and leave as ONE contiguous store per 64 of them: four sparse
            // 8-byte stores per frame (one per byte position) cost the kernel 40 of its 100 us -- the memory pipeline takes
            // a store instruction at a time, whatever its lanes hold
            uint32_t pos = (incl - sw) >> 18;
            uint32_t n_esc = 0u, epos = 0u;
            if constexpr (!ESC) {
                uint32_t addr = rec_lds_addr + ((incl - sw) >> 16), slot_t, slot_p;
                uint64_t slot_sv;
                ADDER_LP_SLOT(0);
                ADDER_LP_SLOT(1);
                ADDER_LP_SLOT(2);
                ADDER_LP_SLOT(3);
            }
            uint32_t rho[kLpUnits];
            if (ESC) {
#pragma unroll
                for (uint32_t j = 0; j < kLpUnits; ++j) rho[j] = i - s.start[j];
                uint32_t ne = 0u;
#pragma unroll
                for (uint32_t j = 0; j < kLpUnits; ++j) ne += ((m.h & (0x80u << (8u * j))) && rho[j] >= kLpRhoEsc) ? 1u : 0u;
                const uint32_t ei = wave_inclusive_scan_dpp(ne);
                epos = ei - ne;
                n_esc = (uint32_t)__builtin_amdgcn_readlane((int)ei, kWave - 1);
            }
            if (ESC) {
#pragma unroll
                for (uint32_t j = 0; j < kLpUnits; ++j) {
                    if (m.h & (0x80u << (8u * j))) {
                        const uint32_t w8 = __builtin_amdgcn_perm(vin, base_w, sel[j]) | unit0 | j;
                        uint32_t r8 = rho[j];
                        if (r8 >= kLpRhoEsc) {
                            rec_lds[kLpPairUnits + epos] = r8;
                            epos += 1u;
                            r8 = kLpRhoEsc;
                        }
                        rec_lds[pos] = w8 | (r8 << kLpRhoShift);
                        pos += 1u;
                        s.start[j] = i;
                    }
                }
            }
            const uint32_t n_rec = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1) >> 18;
            if (!ESC && n_rec <= kWave) {  // (uniform) the usual case: the store waits for the next frame
                pend_n = n_rec;
                pend_seg = seg;
            } else {
#if !defined(ADDER_DBG_LP_NOSTORE)  // (diagnostic A/B build: everything but the record stores)
                for (uint32_t k0 = 0; k0 < n_rec; k0 += kWave) {  // more than a quarter of the units flushed, or escapes: at once
                    const uint32_t k = k0 + lane;
                    if (k < n_rec) gstore(seg, k * 4u, rec_lds[k]);
                    since += 1u;
                }
                if (ESC) {
                    for (uint32_t k0 = 0; k0 < n_esc; k0 += kWave) {  // (escape k: 4 (k + 1) bytes below the end of the pair's two slots)
                        const uint32_t k = k0 + lane;
                        if (k < n_esc) gstore(seg, 2u * park_bytes_u - 4u * (k + 1u), rec_lds[kLpPairUnits + k]);
                        since += 1u;
                    }
                }
#endif
            }
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
            for (; i < i_end; ++i) frame(i, in_lds + (i % kLpInFrames) * kWave, std::true_type{});
#if !defined(ADDER_LP_NO_UNROLL_GROUP)
        } else if (i + kLpGroup == i_end) {
            // a whole group, unrolled (a loop of convergent operations is only unrolled at a constant trip count): no registers
            // rotated between frames, the rows' LDS addresses as instruction offsets
            const uint32_t *const row0 = in_lds + (i % kLpInFrames) * kWave;
#pragma unroll
            for (uint32_t q = 0; q < kLpGroup; ++q) frame(i + q, row0 + q * kWave, std::false_type{});
            i += kLpGroup;
#endif
        } else {
#pragma clang loop unroll(disable)
            for (; i < i_end; ++i) frame(i, in_lds + (i % kLpInFrames) * kWave, std::false_type{});
        }
    }

#if !defined(ADDER_DBG_LP_NOSTORE)
    if (pend_n != 0u && lane < pend_n) gstore(pend_seg, lane * 4u, rec_lds[lane]);  // the last frame's records
#endif
    // the frames' totals: lane f holds frame f's {records | events << 16} up to lane 31 and up to lane 63
    if (lane < nb) {
        const uint2 t = reinterpret_cast<const uint2 *>(lds_tot)[lane];
        const uint32_t ta = t.x, tb = t.y - t.x;  // {events | 4 x records << 16} of segment 2p and of segment 2p + 1
        uint32_t sl = slot0 + lane;
        sl = sl >= slots_u ? sl - slots_u : sl;
        // wtot = events | records << 16 of each segment
        gstore<uint2>(uniform_ptr(b->wtot_ring), (sl * num_waves_u + sgw) * 4u,
                      make_uint2((ta & 0xffffu) | ((ta >> 18) << 16), (tb & 0xffffu) | ((tb >> 18) << 16)));
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
    chain_zero(b, f, nb);  // (the scan chains the frame offsets of these batches: adder_scan_kernel CHAIN)
    const uint32_t num_pairs = a.num_waves / 2u;  // (num_waves is a multiple of kExpandSegs)
    for (uint32_t pw = blockIdx.x * kWavesPerBlock + tid / kWave; pw < num_pairs; pw += gridDim.x * kWavesPerBlock) {
        const bool full = __builtin_amdgcn_readfirstlane(pw * kLpPairUnits + kLpPairUnits <= a.n_units);
        if (full) lp_frames<true>(b, a, nb, pw, lane, s_in[tid / kWave], s_tot[tid / kWave], s_rec[tid / kWave], lazy != 0u);
        else lp_frames<false>(b, a, nb, pw, lane, s_in[tid / kWave], s_tot[tid / kWave], s_rec[tid / kWave], lazy != 0u);
    }
    timeline_mark(b, 0u, f, true);
}

// ------------------------------------------------------------------------------------------
// K2 for adder_lp_kernel's records: adder_lpx_kernel<REC>.  A wave takes kLpxPairs consecutive pairs (16
// segments, 2048 units) of ONE frame -- their events are contiguous in the stream.
//   1. the pairs' runs of 4-byte records (one coalesced load per pair, all in flight at once) are unparked into ONE dense run
//      of {rho' | pair << 28, word} in LDS: the decode rounds below then run on 64 records each, whatever the pairs held
//      (a round per pair left 40 of 64 lanes idle);
//   2. a round decodes 64 records (lr_decode8: events A and C worked out from (base_val, rho) and the input byte), a
//      DPP scan of the records' event counts places the events, and every event is written into the staging buffer in its
//      FINAL bytes -- REC = 9 / 11: the raw sink's record (RawOutput::ingest_event, raw/stream.rs:101-120: bincode fixint
//      big-endian {x u16, y u16, [0x01, c,] d u8, t u32}) at whatever byte it falls on; REC = 12: the AdderEvent.  (An LDS
//      store that is not naturally aligned costs fifteen to twenty-three aligned ones -- tools/ubench/lds_writes.hip --, so a
//      9-byte record goes in as nine byte stores, an 11-byte one as eleven; three
//      aligned dwords per event + a packing pass of byte permutes measured 154-159 us per launch against 138-147.)
//   3. the staging buffer sits at the 16-byte phase of its destination, so a flush is 16-byte LDS reads -> 16-byte global
//      stores, a kilobyte per instruction, and single bytes for the <= 15 + 15 bytes the wave shares with its neighbours' blocks.
// One copy of every loop: a sixth of adder_expand_kernel<5>'s code.
// ------------------------------------------------------------------------------------------
#ifndef ADDER_LPX_REC_CAP
#define ADDER_LPX_REC_CAP 384  // records unparked per batch of pairs (>= 256: one pair's worst case)
#endif
#ifndef ADDER_LPX_STAGE_EVENTS
#define ADDER_LPX_STAGE_EVENTS 400  // events the staging buffer holds (>= 192 + what a flush should carry; a multiple of 16).
// 400: a wave's LDS is 6.7 KB for 9-byte records -- six waves per SIMD (512: five; 117.4 / 113.5 against 118.8 / 115.1 us per
// launch), five for 11- and 12-byte ones (512: four).  Eight waves per workgroup, at four per SIMD: 139 us.
#endif
constexpr uint32_t kLpxPairs = kExpandSegs / 2u;
constexpr uint32_t kLpxRecCap = ADDER_LPX_REC_CAP;
#ifndef ADDER_LPX_WAVES
#define ADDER_LPX_WAVES 4
#endif
constexpr uint32_t kLpxWaves = ADDER_LPX_WAVES;  // waves (= items) per workgroup: they share nothing
constexpr uint32_t kLpxStageEvents = ADDER_LPX_STAGE_EVENTS;
static_assert(kLpxPairs == 8u && kLpxRecCap >= kLpPairUnits && kLpxRecCap % kWave == 0u && kLpxStageEvents >= 6u * kWave && kLpxStageEvents % 16u == 0u, "sizes the loops below assume");

// What adder_lpx_kernel takes by value: what stays the same for every batch of a context's current scratch ring.
struct LpxArgs {
    const uint8_t *park_chunk;             // the launch's chunk of the scratch ring
    const uint32_t *wtot, *wpref, *ftot;   // the rows of the launch's first frame slot
    const FrameTab *ftab;
    uint32_t *status;
    uint32_t num_waves, fi0;               // fi0: the first frame's slot inside its chunk
    uint32_t group_shift, group_stride, frame_stride, seg_stride, rot_shift, rot_mask;  // ParkLayout
    uint32_t rowlen, channels, row_begin, wraps;
    float inv_row;
};
#define ADDER_LDS __attribute__((address_space(3)))
typedef uint32_t lpx_u32x2 __attribute__((ext_vector_type(2)));  // (plain vectors: HIP's uint2 / uint4 classes do not live in LDS address space)
typedef uint32_t lpx_u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t __attribute__((aligned(1))) lpx_u64_any;  // (an 8-byte LDS store at any byte address: gfx950 takes it)
typedef uint16_t __attribute__((aligned(1))) lpx_u16_any;

// One event into the staging buffer at byte offset `off` (32-bit LDS addressing).  REC = 12: the AdderEvent {x | y << 16,
// c | d << 8, t}; REC = 9 / 11: the raw sink's record in its FINAL bytes {x_hi x_lo y_hi y_lo, [01 c] d, t3 t2 t1 t0}.
template <uint32_t REC>
__device__ __forceinline__ void lpx_put(ADDER_LDS uint8_t *stage, uint32_t off, uint32_t xy, uint32_t xyw, uint32_t c, uint32_t d, uint32_t t) {
    if constexpr (REC == 12u) {
        ADDER_LDS uint32_t *const w = reinterpret_cast<ADDER_LDS uint32_t *>(stage + off);
        w[0] = xy;
        w[1] = c | (d << 8);
        w[2] = t;
    } else {
        const uint32_t tb = __builtin_amdgcn_perm(0u, t, 0x00010203u);  // t3 t2 t1 t0
        if constexpr (REC == 9u) {
#if !defined(ADDER_LPX_WIDE_STAGE)  // nine byte stores -- none misaligned: 9 x 26 ns per wave-instruction per CU -- instead of an 8-byte store at
            // any address + a byte (400-600 + 26 ns, tools/ubench/lds_writes.hip): 1 % of the step, eager or overlapped (-DADDER_LPX_WIDE_STAGE: the old form)
            const uint32_t addr = (uint32_t)(uintptr_t)(stage + off);
            asm volatile("ds_write_b8 %0, %1\n\t"
                         "ds_write_b8_d16_hi %0, %1 offset:2\n\t"
                         "ds_write_b8 %0, %2 offset:1\n\t"
                         "ds_write_b8_d16_hi %0, %2 offset:3\n\t"
                         "ds_write_b8 %0, %3 offset:4\n\t"
                         "ds_write_b8 %0, %4 offset:5\n\t"
                         "ds_write_b8_d16_hi %0, %4 offset:7\n\t"
                         "ds_write_b8 %0, %5 offset:6\n\t"
                         "ds_write_b8_d16_hi %0, %5 offset:8"
                         :
                         : "v"(addr), "v"(xyw), "v"(xyw >> 8), "v"(d), "v"(tb), "v"(tb >> 8)
                         : "memory");
#else
            *reinterpret_cast<ADDER_LDS lpx_u64_any *>(stage + off) = (uint64_t)xyw | ((uint64_t)(d | (tb << 8)) << 32);
            stage[off + 8u] = (uint8_t)(tb >> 24);
#endif
        } else {
#if !defined(ADDER_LPX_WIDE_STAGE)  // (eleven byte stores, like the nine above)
            const uint32_t addr = (uint32_t)(uintptr_t)(stage + off), w1 = 1u | (c << 8) | (d << 16);
            asm volatile("ds_write_b8 %0, %1\n\t"
                         "ds_write_b8_d16_hi %0, %1 offset:2\n\t"
                         "ds_write_b8 %0, %2 offset:1\n\t"
                         "ds_write_b8_d16_hi %0, %2 offset:3\n\t"
                         "ds_write_b8 %0, %3 offset:4\n\t"
                         "ds_write_b8_d16_hi %0, %3 offset:6\n\t"
                         "ds_write_b8 %0, %4 offset:5\n\t"
                         "ds_write_b8 %0, %5 offset:7\n\t"
                         "ds_write_b8_d16_hi %0, %5 offset:9\n\t"
                         "ds_write_b8 %0, %6 offset:8\n\t"
                         "ds_write_b8_d16_hi %0, %6 offset:10"
                         :
                         : "v"(addr), "v"(xyw), "v"(xyw >> 8), "v"(w1), "v"(w1 >> 8), "v"(tb), "v"(tb >> 8)
                         : "memory");
#else
            *reinterpret_cast<ADDER_LDS lpx_u64_any *>(stage + off) = (uint64_t)xyw | ((uint64_t)(1u | (c << 8) | (d << 16) | (tb << 24)) << 32);
            *reinterpret_cast<ADDER_LDS lpx_u16_any *>(stage + off + 8u) = (uint16_t)(tb >> 8);
            stage[off + 10u] = (uint8_t)(tb >> 24);
#endif
        }
    }
}

// One wave's item: kLpxPairs pairs (16 segments) of frame f, the launch's frame `fy`.
template <uint32_t REC>
__device__ __forceinline__ void lpx_wave(const BatchArgs *__restrict__ b, const LpxArgs &x, uint32_t f, uint32_t fy, uint32_t seg0,
                                         uint32_t lane, ADDER_LDS uint8_t *stage, ADDER_LDS lpx_u32x2 *rec_lds) {
    const uint32_t pair_stride = 2u * x.seg_stride;  // (16 segments of one group: a constant stride apart)
    // park_offset() inside the launch's chunk of the ring
    const uint32_t fi = (x.fi0 + fy + (seg0 >> x.rot_shift)) & x.rot_mask;
    const uint32_t group = seg0 >> x.group_shift;
    const uint8_t *const park = x.park_chunk + (size_t)group * x.group_stride + (size_t)fi * x.frame_stride +
                                (size_t)(seg0 - (group << x.group_shift)) * x.seg_stride;
    // one round trip: the segments' totals, the events in front of them, the frame's place in the stream -- and (below) the
    // records themselves, which do not wait for the totals
    // Every vector load of the item is ISSUED HERE, before anything is waited for, by every lane (a load under a lane mask is
    // a branch the compiler waits at) and as volatile asm: written as ordinary loads the compiler sinks them below the
    // early exits, behind the scalar loads those wait for -- a chain of six round trips where one will do.  lpx_loads_done()
    // is their s_waitcnt.
    const size_t row = (size_t)fy * x.num_waves + seg0;
    uint32_t my_tot, pref0, first[kLpxPairs];
    const uint32_t tot_off = (lane < kExpandSegs ? lane : kExpandSegs - 1u) * 4u, zero_off = 0u, rec_off = lane * 4u;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(my_tot) : "v"(tot_off), "s"(x.wtot + row) : "memory");
    asm volatile("global_load_dword %0, %1, %2" : "=v"(pref0) : "v"(zero_off), "s"(x.wpref + row) : "memory");
#pragma unroll
    for (uint32_t p = 0; p < kLpxPairs; ++p)  // every pair's first 64 record words (whatever the pair holds: the slots are there)
        asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(first[p]) : "v"(rec_off), "s"(park + (size_t)p * pair_stride) : "memory");
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 fo2;  // the frame's place in the stream (the batch's own table: its address comes out of the description first)
    const uint32_t fo_off = f * 8u;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(fo2) : "v"(fo_off), "s"(b->base.frame_offsets) : "memory");
    const float T = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b->base.sc.time_spanned)));
    const uint64_t out_cap = b->base.out_cap;
    uint8_t *const out = reinterpret_cast<uint8_t *>(uniform_ptr(b->base.out));
    const uint32_t rt_u32 = __builtin_amdgcn_readfirstlane(f32_as_u32(x.ftab[f].running_t));  // t of D_EMPTY
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(fo2), "+v"(my_tot), "+v"(pref0), "+v"(first[0]), "+v"(first[1]), "+v"(first[2]), "+v"(first[3]), "+v"(first[4]), "+v"(first[5]),
                   "+v"(first[6]), "+v"(first[7])
                 :
                 : "memory");
    my_tot = lane < kExpandSegs ? my_tot : 0u;
    if (__builtin_amdgcn_ballot_w64((my_tot & 0xffffu) != 0u) == 0ull) return;  // quiet content: sixteen empty segments
    // records of pair p = of segments 2p and 2p + 1 (lanes 2p, 2p + 1 hold them: a quad permute adds the neighbour's)
    const uint32_t recs = my_tot >> 16;
    const uint32_t pair_recs = recs + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)recs, 0xb1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    uint32_t np[kLpxPairs];
#pragma unroll
    for (uint32_t p = 0; p < kLpxPairs; ++p) np[p] = (uint32_t)__builtin_amdgcn_readlane((int)pair_recs, 2 * p);
    // geometry: (row, offset in row) of the wave's first unit; a unit of the wave lies at most 2047 behind it
    const uint32_t rowlen = x.rowlen;
    const bool rgb = x.channels == 3u;
    const float inv_row = x.inv_row;
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
    const uint32_t yb = y0 + x.row_begin;
    const uint32_t wraps = x.wraps;
    uint64_t gpos = (((uint64_t)__builtin_amdgcn_readfirstlane(fo2.y) << 32) | __builtin_amdgcn_readfirstlane(fo2.x)) +
                    __builtin_amdgcn_readfirstlane(pref0);
    uint32_t fill = 0u;  // events staged (uniform)
    // the staging buffer's first byte sits at the 16-byte phase of its destination: LDS blocks are destination blocks
    uint32_t phase = (uint32_t)((uintptr_t)out + gpos * REC) & 15u;
    bool dropped = false;
    constexpr uint32_t CAPE = kLpxStageEvents;

    auto flush = [&]() {  // staged events -> stream; events past the caller's capacity are dropped and reported
        const uint64_t room64 = gpos < out_cap ? out_cap - gpos : 0ull;
        const uint32_t n = (uint64_t)fill <= room64 ? fill : (uint32_t)room64;
        dropped = dropped || n != fill;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (n) {
            uint8_t *const dst = uniform_ptr(out + gpos * REC - phase);  // 16-byte aligned: LDS byte L of the buffer goes to dst + L
            const uint32_t end = phase + n * REC;
            const uint32_t kb = phase ? 1u : 0u, ke = end >> 4;
            for (uint32_t k = kb + lane; k < ke; k += kWave) {  // 16-byte LDS reads -> 16-byte stores, a kilobyte per instruction
                const lpx_u32x4 lv = *reinterpret_cast<ADDER_LDS const lpx_u32x4 *>(stage + 16u * k);
                const uint4 v = make_uint4(lv.x, lv.y, lv.z, lv.w);
#if !defined(ADDER_DBG_LPX) || ADDER_DBG_LPX < 1  // (diagnostic A/B builds: 1 no event stores, 2 no staging either, 3 no rounds at all)
                gstore_ev<uint4>(dst, 16u * k, v);
#else
                if (v.x == 0x12345u && v.w == 0x54321u) gstore_ev<uint4>(dst, 16u * k, v);
#endif
            }
            // the bytes in front of the first whole block and behind the last one: shared with the neighbouring waves' blocks
            const uint32_t head_n = phase ? (end < 16u ? end : 16u) - phase : 0u;
            const uint32_t tail_lo = (end > 16u || phase == 0u) ? (ke << 4) : end;
            const uint32_t tail_n = end - tail_lo;
            const uint32_t t = lane & 15u;
            const bool is_tail = lane >= 16u;
            if (lane < 32u && t < (is_tail ? tail_n : head_n)) {
                const uint32_t L = (is_tail ? tail_lo : phase) + t;
                gstore<uint8_t>(dst, L, stage[L]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        gpos += fill;
        phase = (uint32_t)((uintptr_t)out + gpos * REC) & 15u;
        fill = 0u;
    };

    // (the rare paths below work their pairs' addresses out again from an opaque copy of the base: reusing the eight pointers
    // of the loads above kept sixteen scalar registers alive through the whole item -- and spilled as many)
    const uint8_t *park_rare = park;
    asm volatile("" : "+s"(park_rare));
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
                // (an opaque copy of the count: its lane masks are loop invariants otherwise, all eight pairs' hoisted in
                // front of this loop -- 48 scalar registers, most of them spilled to lanes and read back here)
                uint32_t n_p = np[p];
                asm volatile("" : "+s"(n_p));
                const uint32_t w4 = first[p];
                if (lane < n_p) rec_lds[rb + lane] = lpx_u32x2{(w4 >> kLpRhoShift) | (p << 28), w4};
                esc_any |= __builtin_amdgcn_ballot_w64(lane < n_p && lp_escapes(w4));
                if (__builtin_expect(n_p > kWave, 0)) {  // (more than a quarter of the pair's units flushed)
                    const uint8_t *pr = park_rare;
                    asm volatile("" : "+s"(pr));  // (worked out here, not hoisted)
                    const uint8_t *const pp = pr + (size_t)p * pair_stride;
                    for (uint32_t l0 = kWave; l0 < n_p; l0 += kWave) {
                        const uint32_t idx = l0 + lane;
                        const uint32_t w = idx < n_p ? gload_rec<uint32_t>(pp, idx * 4u) : 0u;
                        if (idx < n_p) rec_lds[rb + idx] = lpx_u32x2{(w >> kLpRhoShift) | (p << 28), w};
                        esc_any |= __builtin_amdgcn_ballot_w64(lp_escapes(w));
                    }
                }
                rb += n_p;
            }
        }
        // (the last round's idle lanes read zero records -- no events -- instead of being masked out of the read)
        if ((R & (kWave - 1u)) != 0u && lane >= (R & (kWave - 1u))) rec_lds[(R & ~(kWave - 1u)) + lane] = lpx_u32x2{0u, 0u};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__builtin_expect(esc_any != 0ull, 0)) {
            // escapes: the k-th escaping record of a pair's run keeps its full rho' 4 (k + 1) bytes below the end of the pair's slots
            uint32_t qb = 0u;
#pragma unroll 1
            for (uint32_t p = pstart; p < pe; ++p) {
                const uint32_t n_p = (uint32_t)__builtin_amdgcn_readlane((int)pair_recs, 2 * p);
                const uint8_t *const pp = park_rare + (size_t)p * pair_stride;
                uint32_t esc_before = 0u;
                for (uint32_t l0 = 0; l0 < n_p; l0 += kWave) {
                    const uint32_t idx = l0 + lane;
                    lpx_u32x2 r = {0u, 0u};
                    if (idx < n_p) r = rec_lds[qb + idx];
                    const bool e = idx < n_p && lp_escapes(r.y);
                    const uint64_t em = __builtin_amdgcn_ballot_w64(e);
                    if (em != 0ull) {
                        const uint32_t rank = esc_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                        if (e) rec_lds[qb + idx] = lpx_u32x2{gload_rec<uint32_t>(pp, pair_stride - 4u * (rank + 1u)) | (p << 28), r.y};
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
        struct Round {
            LeanEvents e;
            uint32_t n, incl, xy, xyw, c;
        };
        auto decode = [&](uint32_t idx) {  // 64 records -> their events, the events' places among the round's, the coordinates
            Round r;
            const lpx_u32x2 rec = rec_lds[idx];
            const uint32_t w8 = rec.y;
            r.e = lr_decode8(lp_rho(rec.x & 0x0fffffffu, w8), w8, T, rt_u32);  // (a zero record: no events)
            r.n = (r.e.a ? 1u : 0u) + (r.e.b ? 1u : 0u) + (r.e.c ? 1u : 0u);
            r.incl = wave_inclusive_scan_dpp(r.n);
            // coordinates: the unit counted from the wave's first
            uint32_t rem = rem0 + ((rec.x >> 28) << 8) + (w8 & 0xffu);
            uint32_t y = yb;
            if (wraps != 0u) {  // at most two rows further on: min(rem, rem - rowlen) is rem - rowlen exactly when that does not wrap
                y += (rem >= rowlen ? 1u : 0u) + (rem >= 2u * rowlen ? 1u : 0u);
                rem = min(rem, rem - rowlen);
                rem = min(rem, rem - rowlen);
            } else {  // narrow planes: a quotient estimate, one step either way
                uint32_t q = (uint32_t)((float)rem * inv_row);
                q -= q * rowlen > rem ? 1u : 0u;
                q += (q + 1u) * rowlen <= rem ? 1u : 0u;
                rem -= q * rowlen;
                y += q;
            }
            uint32_t x = rem;
            r.c = 0xffu;
            if (rgb) {
                x = (uint32_t)(((uint64_t)rem * 0xAAAAAAABull) >> 33);  // rem / 3
                r.c = rem - 3u * x;
            }
            r.xy = x | (y << 16);
            r.xyw = __builtin_amdgcn_perm(0u, r.xy, 0x02030001u);  // x_hi x_lo y_hi y_lo
            return r;
        };
        auto put = [&](const Round &r, uint32_t off) {
#if defined(ADDER_DBG_LPX) && ADDER_DBG_LPX >= 2
            if (r.e.a && r.e.ta == 0x7654321u && r.e.tc == 0x1234567u && r.xyw == 0x33u) lpx_put<REC>(stage, off, r.xy, r.xyw, r.c, r.e.da, r.e.ta + r.e.tc);
#else
            if (r.e.a) {
                lpx_put<REC>(stage, off, r.xy, r.xyw, r.c, r.e.da, r.e.ta);
                off += REC;
            }
            if (r.e.b) {
                lpx_put<REC>(stage, off, r.xy, r.xyw, r.c, kDEmpty, r.e.tb);
                off += REC;
            }
            if (r.e.c) lpx_put<REC>(stage, off, r.xy, r.xyw, r.c, r.e.dc, r.e.tc);
#endif
        };
        uint32_t r0 = 0u;
#if defined(ADDER_DBG_LPX) && ADDER_DBG_LPX >= 3
        const uint32_t R_run = R == 0x7777u ? 64u : 0u;
#else
        const uint32_t R_run = R;
#endif
#if defined(ADDER_LPX_ROUNDS2)
        // two rounds at a time while there are two: their chains (an LDS read, three divisions, a DPP scan, the LDS stores) are
        // independent up to the second's place in the stream -- a wave spent 39 % of its life in s_waitcnt with one at a time
#pragma clang loop unroll(disable)
        for (; r0 + kWave < R_run; r0 += 2u * kWave) {
            if (fill + 6u * kWave > CAPE) flush();
            const Round ra = decode(r0 + lane), rb2 = decode(r0 + kWave + lane);
            const uint32_t ta = (uint32_t)__builtin_amdgcn_readlane((int)ra.incl, kWave - 1);
            const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)rb2.incl, kWave - 1);
            const uint32_t off_a = phase + (fill + ra.incl - ra.n) * REC, off_b = phase + (fill + ta + rb2.incl - rb2.n) * REC;
            fill += ta + tb;
            put(ra, off_a);
            put(rb2, off_b);
        }
#endif
#pragma clang loop unroll(disable)
        for (; r0 < R_run; r0 += kWave) {
            if (fill + 3u * kWave > CAPE) flush();
            const Round r = decode(r0 + lane);
            const uint32_t off = phase + (fill + r.incl - r.n) * REC;
            fill += (uint32_t)__builtin_amdgcn_readlane((int)r.incl, kWave - 1);
            put(r, off);
        }
        pstart = pe;
    }
    flush();
    if (dropped) raise(x.status, kStatusCapacity);
}

// grid: (blocks of 64 segments, frames of the launch).  What does not change from batch to batch of a context -- the scratch
// ring, the plane's geometry -- arrives as kernel arguments (LpxArgs: scalar registers when the wave starts); the batch's own
// (output buffer, capacity, frame offsets, time step) come from the device-resident description like everywhere else, so a
// captured graph still serves any batch.  (With everything read from the description a wave spent 54 % of its life in
// s_waitcnt: 18 dependent scalar loads.)  (Workgroups that walk several items with the next
// items' loads in flight were tried: 147-172 us per launch against this form's 138-144 -- the uniform state of three items
// in flight spilled the scalar registers.)
template <uint32_t REC>
__global__ __launch_bounds__(kLpxWaves * kWave) void adder_lpx_kernel(const BatchArgs *__restrict__ b, const LpxArgs x, uint32_t f0) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kLpxWaves][kLpxStageEvents * REC + 16u];
    __shared__ __attribute__((aligned(8))) uint2 s_rec[kLpxWaves][kLpxRecCap];
    // (the waves of a workgroup share nothing: no table in LDS -- event C's one division is worked out like event A's --, no barrier)
    const uint32_t fy = blockIdx.y, xblock = blockIdx.x;
    // a frame without a single event has nothing to expand: static content is mostly such frames, and their workgroups are gone
    // before a vector load is issued (one scalar round trip; busy frames pay it once more -- 1 % of a wave's life)
    if (x.ftot[fy] == 0u) return;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t seg0 = __builtin_amdgcn_readfirstlane((xblock * kLpxWaves + wid) * kExpandSegs);
    if (seg0 < x.num_waves)
        lpx_wave<REC>(b, x, f0 + fy, fy, seg0, threadIdx.x & (kWave - 1u), (ADDER_LDS uint8_t *)s_stage[wid], (ADDER_LDS lpx_u32x2 *)s_rec[wid]);
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

extern "C" hipError_t adder_launch_lpx(const BatchArgs *b, const BatchArgs *hb, uint32_t f0, uint32_t nf, uint32_t rec, hipStream_t stream) {
    // hb: the host's copy of the description *b -- the kernel arguments are worked out from its context-constant part
    const uint32_t num_waves = hb->base.num_waves, slot0 = f0 % hb->slots, cir = slot0 / hb->chunk;
    LpxArgs x;
    x.park_chunk = hb->park_ring + (size_t)cir * num_waves * hb->chunk * hb->park_bytes;
    x.wtot = hb->wtot_ring + (size_t)slot0 * num_waves;
    x.wpref = hb->wpref_ring + (size_t)slot0 * num_waves;
    x.ftot = hb->ftot_ring + slot0;
    x.ftab = hb->ftab;
    x.status = hb->base.status;
    x.num_waves = num_waves;
    x.fi0 = slot0 - cir * hb->chunk;
    x.group_shift = hb->park_layout.group_shift;
    x.group_stride = hb->park_layout.group_stride;
    x.frame_stride = hb->park_layout.frame_stride;
    x.seg_stride = hb->park_layout.seg_stride;
    x.rot_shift = hb->park_layout.rot_shift;
    x.rot_mask = hb->park_layout.rot_mask;
    x.rowlen = hb->base.rowlen;
    x.channels = hb->base.channels;
    x.row_begin = hb->base.row_begin;
    x.wraps = x.rowlen >= kLpxPairs * kLpPairUnits ? 1u : x.rowlen >= kLpxPairs * kLpPairUnits / 2u ? 2u : 0u;  // (0: divide)
    x.inv_row = 1.0f / (float)x.rowlen;
    const uint32_t per_block = kLpxWaves * kExpandSegs;  // segments per block
    const dim3 grid((num_waves + per_block - 1u) / per_block, nf), block(kLpxWaves * kWave);
    if (rec == 9u) hipLaunchKernelGGL((adder_lpx_kernel<9u>), grid, block, 0, stream, b, x, f0);
    else if (rec == 11u) hipLaunchKernelGGL((adder_lpx_kernel<11u>), grid, block, 0, stream, b, x, f0);
    else hipLaunchKernelGGL((adder_lpx_kernel<12u>), grid, block, 0, stream, b, x, f0);
    return hipGetLastError();
}
