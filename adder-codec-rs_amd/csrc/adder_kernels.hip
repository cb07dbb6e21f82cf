// adder_kernels.hip -- CDNA4 (gfx950) kernels of the framed->ADDER integration path.
//
// One launch of adder_frame_kernel does for one input frame what the reference's
// rayon loop does (adder-codec-rs/src/transcoder/source/video.rs:677-734): every
// pixel-channel runs integrate_for_px (video.rs:1318-1380) and the emitted events
// are gathered in raster order (y, x, c, per-pixel emission order).
//
// Mapping to the hardware (memory-bound, no MFMA):
//   * structure-of-arrays pixel state resident in HBM across frames (see
//     adder_pixel.hpp for the fields); a lane owns 4 consecutive pixel-channels so
//     every state access is a 16-byte-per-lane coalesced vector load/store and the
//     frame row is read as one dword per lane;
//   * level-planar arena storage: plane k holds every pixel's k-th fired node, so
//     only the planes a wave actually needs are touched;
//   * ordered stream compaction in ONE pass: per-lane event counts (phase A of the
//     step) -> wave prefix (cross-lane shuffles) -> block prefix in LDS -> tile
//     prefix by decoupled look-back over 8-byte {status,value} descriptors that are
//     read/written with relaxed agent-scope atomics (the data is the flag, so no
//     fences), then phase B emits each event straight to its final slot;
//   * a persistent grid (<= resident capacity, verified by a census launch) strides
//     over 1024-unit tiles, so the look-back can never wait on a block that is not
//     running; every wait is bounded and reports ADDER_E_TIMEOUT instead of hanging.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adder_kernels.h"
#include "adder_pixel.hpp"

namespace adder {

constexpr uint32_t kWave = 64;
constexpr uint32_t kWavesPerBlock = kBlockThreads / kWave;

constexpr uint64_t kDescAggregate = 1ull << 32;
constexpr uint64_t kDescPrefix = 2ull << 32;

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

// Decoupled look-back (single-pass chained scan): returns the number of events
// emitted by all tiles before `tile` in this frame.  Executed by one full wave.
__device__ __forceinline__ uint32_t lookback(uint64_t *desc, uint32_t tile, uint32_t lane,
                                             uint32_t spin_limit, uint32_t *status) {
    uint32_t excl = 0;
    int32_t pos = (int32_t)tile - 1;
    uint32_t spins = 0;
    for (;;) {
        const int32_t idx = pos - (int32_t)lane;
        const uint64_t d = idx >= 0 ? desc_load(desc + idx) : kDescPrefix;  // before tile 0: prefix 0
        const uint32_t st = (uint32_t)(d >> 32);
        const uint64_t invalid = __ballot(st == 0u);
        const uint64_t pmask = __ballot(st == 2u);
        const uint32_t first_p = pmask ? (uint32_t)__builtin_ctzll(pmask) : 63u;
        const uint64_t need = first_p >= 63u ? ~0ull : ((2ull << first_p) - 1ull);
        if (invalid & need) {
            if (++spins > spin_limit) {
                if (lane == 0) raise(status, kStatusTimeout);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        excl += wave_sum(((need >> lane) & 1ull) ? (uint32_t)d : 0u);
        if (pmask) break;
        pos -= (int32_t)kWave;
    }
    return excl;
}

struct DeepGlobal {
    float *integ, *dt, *bdt;
    uint16_t *dbd;
    size_t stride;
    size_t u;
    __device__ __forceinline__ void load(uint32_t k, Node &n) const {
        const size_t i = (size_t)k * stride + u;
        n.integ = integ[i];
        n.dt = dt[i];
        n.bdt = bdt[i];
        const uint32_t w = dbd[i];
        n.d = w & 0xffu;
        n.bd = w >> 8;
    }
    __device__ __forceinline__ void store(uint32_t k, const Node &n) const {
        const size_t i = (size_t)k * stride + u;
        integ[i] = n.integ;
        dt[i] = n.dt;
        bdt[i] = n.bdt;
        dbd[i] = (uint16_t)(n.d | (n.bd << 8));
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

__global__ __launch_bounds__(kBlockThreads) void adder_frame_kernel(FrameArgs a) {
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

    for (uint32_t tile = blockIdx.x; tile < a.num_tiles; tile += gridDim.x) {
        const uint32_t u0 = tile * kTileUnits + tid * kUnitsPerLane;

        // ---------------- loads: header, input, resident state ----------------
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
        const uint32_t fl_or = (hdrv[0] | hdrv[1] | hdrv[2] | hdrv[3]) >> 24;
        const bool any_m = (fl_or & kFlagMMask) != 0u;
        const bool any_live = (fl_or & kFlagTailLive) != 0u;

        PxState px[4];
        float4 li = make_float4(0.f, 0.f, 0.f, 0.f), ld = li, lb = li, ti = li, tt = li, lf = li;
        uint2 ldbd = make_uint2(0u, 0u);
        uint32_t tdw = 0u;
        if (any_m) {
            li = *reinterpret_cast<const float4 *>(a.lv_integ + u0);
            ld = *reinterpret_cast<const float4 *>(a.lv_dt + u0);
            lb = *reinterpret_cast<const float4 *>(a.lv_bdt + u0);
            ldbd = *reinterpret_cast<const uint2 *>(a.lv_dbd + u0);
        }
        if (any_live) {
            ti = *reinterpret_cast<const float4 *>(a.tinteg + u0);
            tt = *reinterpret_cast<const float4 *>(a.tdt + u0);
            tdw = *reinterpret_cast<const uint32_t *>(a.td + u0);
        }
        if (sc.abs_t) lf = *reinterpret_cast<const float4 *>(a.lastf + u0);
        {
            const float liv[4] = {li.x, li.y, li.z, li.w}, ldv[4] = {ld.x, ld.y, ld.z, ld.w};
            const float lbv[4] = {lb.x, lb.y, lb.z, lb.w}, tiv[4] = {ti.x, ti.y, ti.z, ti.w};
            const float ttv[4] = {tt.x, tt.y, tt.z, tt.w}, lfv[4] = {lf.x, lf.y, lf.z, lf.w};
            const uint32_t dbdv[4] = {ldbd.x & 0xffffu, ldbd.x >> 16, ldbd.y & 0xffffu, ldbd.y >> 16};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                px[j].hdr = hdrv[j];
                px[j].n0.integ = liv[j];
                px[j].n0.dt = ldv[j];
                px[j].n0.bdt = lbv[j];
                px[j].n0.d = dbdv[j] & 0xffu;
                px[j].n0.bd = dbdv[j] >> 8;
                px[j].tinteg = tiv[j];
                px[j].tdt = ttv[j];
                px[j].td = (tdw >> (8 * j)) & 0xffu;
                px[j].lastf = lfv[j];
            }
        }

        // ---------------- phase A: counts and the ordered prefix ----------------
        uint32_t cnt[4];
        uint32_t lane_cnt = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cnt[j] = (u0 + j < a.n_units) ? plan_count(px[j], vin[j], sc) : 0u;
            lane_cnt += cnt[j];
        }
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
            // publish this tile's aggregate, then resolve its exclusive prefix
            if (lane == 0) {
                desc_store(a.desc_cur + tile, (tile == 0 ? kDescPrefix : kDescAggregate) | block_total);
                a.desc_next[tile] = 0ull;  // ready for the next frame's launch
            }
            uint32_t excl = 0;
            if (tile != 0) {
                excl = lookback(a.desc_cur, tile, lane, a.spin_limit, a.status);
                if (lane == 0) desc_store(a.desc_cur + tile, kDescPrefix | (excl + block_total));
            }
            if (lane == 0) {
                s_tile_base = excl;
                if (tile == a.num_tiles - 1)
                    a.frame_offsets[a.frame_idx + 1] = frame_base + excl + block_total;
            }
        }
        __syncthreads();
        const uint64_t out_pos0 = frame_base + s_tile_base + lane_off;

        // ---------------- phase B: the step itself, events to their final slots ----------------
        EmitGlobal em;
        em.out = reinterpret_cast<EventWords *>(a.out);
        em.pos = out_pos0;
        em.cap = a.out_cap;
        em.dropped = false;
        bool depth_ok = true;
        {
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
            DeepGlobal deep{a.lv_integ, a.lv_dt, a.lv_bdt, a.lv_dbd, a.plane_stride, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (u0 + j < a.n_units) {
                    em.xy = x | ((y + a.row_begin) << 16);
                    em.c = a.channels == 1u ? 0xffu : c;
                    deep.u = u0 + j;
                    depth_ok &= exec_step(px[j], vin[j], sc, deep, em);
                    if (a.running) {
                        if ((px[j].hdr >> 24) & kFlagMMask)
                            a.running[u0 + j] = (uint8_t)frame_value_u8(
                                px[j].n0.bd, f32_as_u32(px[j].n0.bdt), (double)sc.ref_time);
                    }
                }
                if (++c >= a.channels) {
                    c = 0u;
                    if (++x >= a.width) {
                        x = 0u;
                        ++y;
                    }
                }
            }
        }
        if (em.dropped) raise(a.status, kStatusCapacity);
        if (!depth_ok) raise(a.status, kStatusDepth);

        // ---------------- stores ----------------
        {
            uint4 h;
            h.x = px[0].hdr;
            h.y = px[1].hdr;
            h.z = px[2].hdr;
            h.w = px[3].hdr;
            *reinterpret_cast<uint4 *>(a.hdr + u0) = h;
            const uint32_t nfl = (h.x | h.y | h.z | h.w) >> 24;
            if (nfl & kFlagMMask) {
                *reinterpret_cast<float4 *>(a.lv_integ + u0) =
                    make_float4(px[0].n0.integ, px[1].n0.integ, px[2].n0.integ, px[3].n0.integ);
                *reinterpret_cast<float4 *>(a.lv_dt + u0) =
                    make_float4(px[0].n0.dt, px[1].n0.dt, px[2].n0.dt, px[3].n0.dt);
                *reinterpret_cast<float4 *>(a.lv_bdt + u0) =
                    make_float4(px[0].n0.bdt, px[1].n0.bdt, px[2].n0.bdt, px[3].n0.bdt);
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (px[j].n0.d & 0xffu) | ((px[j].n0.bd & 0xffu) << 8);
                *reinterpret_cast<uint2 *>(a.lv_dbd + u0) = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
            }
            if (nfl & kFlagTailLive) {
                *reinterpret_cast<float4 *>(a.tinteg + u0) =
                    make_float4(px[0].tinteg, px[1].tinteg, px[2].tinteg, px[3].tinteg);
                *reinterpret_cast<float4 *>(a.tdt + u0) =
                    make_float4(px[0].tdt, px[1].tdt, px[2].tdt, px[3].tdt);
                *reinterpret_cast<uint32_t *>(a.td + u0) = (px[0].td & 0xffu) | ((px[1].td & 0xffu) << 8) |
                                                           ((px[2].td & 0xffu) << 16) | (px[3].td << 24);
            }
            if (sc.abs_t)
                *reinterpret_cast<float4 *>(a.lastf + u0) =
                    make_float4(px[0].lastf, px[1].lastf, px[2].lastf, px[3].lastf);
        }
        __syncthreads();  // s_wave_tot / s_tile_base are reused by the next tile
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

extern "C" hipError_t adder_launch_frame(const FrameArgs *args, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(adder_frame_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, *args);
    return hipGetLastError();
}

extern "C" hipError_t adder_frame_kernel_occupancy(int *blocks_per_cu) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, adder_frame_kernel, kBlockThreads, 0);
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
