// adder_lp_kernels.hip -- the headline regime's frame kernel in PACKED BYTES (gfx950).
//
// adder_lp_kernel is adder_lr_kernel's step (adder_pixel.hpp: LEAN RUNS -- Collapse, delta_t_max <= time_spanned, c_thresh 0
// throughout, one integer time_spanned, DeltaT; video.rs:1318-1380 over event_pixel_tree.rs:213-287, 317-413 in that regime)
// with FOUR units per lane and the lane's four input bytes of a frame never taken apart (adder_pixel.hpp: LEAN RUNS,
// PACKED): one wave steps a PAIR of segments (256 units), the flush / event masks are bit 7 of the unit's byte, the frame's
// events are v_bcnt of three words, rho is the distance to the frame of the unit's last flush -- a unit that does not
// change costs nothing of its own.  Per frame and wave: one DPP scan of {records | events << 16} places the records and
// leaves both segments' totals (lanes 31 and 63, parked in LDS until the launch ends), the four byte positions store
// their records under the flush mask.  The pair's records are CONTIGUOUS in the pair's two slots (segment 2p's then
// segment 2p + 1's: the expansion's format 7 reads one run per pair), 8 bytes each: {rho', unit (8 bits) | base_val << 8 |
// input << 16}.  Same resident planes, scan, offsets and ring as every other frame kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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
                                          uint32_t lane, uint8_t *lds_in, uint32_t *lds_tot, bool lazy) {
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

    auto frame = [&](uint32_t i) {
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
            uint32_t off = ((incl - sw) & 0xffffu) * 8u;
#pragma unroll
            for (uint32_t j = 0; j < kLpUnits; ++j) {
                if (m.h & (0x80u << (8u * j))) {
                    const uint32_t w0 = i - s.start[j];
                    const uint32_t w8 = __builtin_amdgcn_perm(vin, base_w, sel[j]) | unit0 | j;
                    gstore(seg, off, make_uint2(w0, w8));
                    off += 8u;
                    s.start[j] = i;
                }
            }
        } else {
            lp_quiet(s);
        }
        seg += frame_stride_u;
        if (__builtin_expect(i == wrap_at, 0)) seg -= wrap_bytes;
    };

    uint32_t i = 0u;
    while (i < nb) {
        stage(i);  // (i is a multiple of kLpGroup here)
        const uint32_t i_end = i + kLpGroup < nb ? i + kLpGroup : nb;
#pragma clang loop unroll(disable)
        for (; i < i_end; ++i) frame(i);
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
    timeline_mark(b, 0u, f, false);
    const uint32_t num_pairs = a.num_waves / 2u;  // (num_waves is a multiple of kExpandSegs)
    for (uint32_t pw = blockIdx.x * kWavesPerBlock + tid / kWave; pw < num_pairs; pw += gridDim.x * kWavesPerBlock) {
        const bool full = __builtin_amdgcn_readfirstlane(pw * kLpPairUnits + kLpPairUnits <= a.n_units);
        if (full) lp_frames<true>(b, a, nb, pw, lane, s_in[tid / kWave], s_tot[tid / kWave], lazy != 0u);
        else lp_frames<false>(b, a, nb, pw, lane, s_in[tid / kWave], s_tot[tid / kWave], lazy != 0u);
    }
    timeline_mark(b, 0u, f, true);
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
