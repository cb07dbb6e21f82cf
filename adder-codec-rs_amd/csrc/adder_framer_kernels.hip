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

#include "adder_framer_kernels.h"

namespace adder {

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
                a.ring[(size_t)((uint32_t)f % a.ring_frames) * a.n_units + u] = (uint8_t)p.lasti;
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

// The transcoder's streams: T per-frame segments, each in raster order.  A workgroup owns
// `rows_per_block` rows of the band for the WHOLE batch: it finds its rows' slice of every
// frame segment (events are sorted by y: two binary searches per frame, done for all frames
// in parallel by different threads), then walks the frames in order with a workgroup barrier
// in between -- the pixel trackers of its rows are private to it, so the barrier is all the
// ordering the per-pixel event order needs, and one launch replaces T launches.
__device__ __forceinline__ uint64_t framer_lower_bound_y(const uint32_t *__restrict__ ev, uint64_t lo, uint64_t hi,
                                                         uint32_t y) {
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if ((ev[3 * mid] >> 16) < y)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
constexpr uint32_t kFramerRowsThreads = 512;
__global__ __launch_bounds__(kFramerRowsThreads) void adder_framer_rows_kernel(const uint32_t *__restrict__ ev,
                                                                               const uint64_t *__restrict__ seg_offsets,
                                                                               uint32_t T, uint32_t rows_per_block,
                                                                               FramerArgs a) {
    __shared__ uint64_t s_lo[kFramerRowsMaxFrames], s_hi[kFramerRowsMaxFrames];
    const uint32_t y0 = a.row_begin + blockIdx.x * rows_per_block;
    const uint32_t y_end = a.row_begin + a.rows;
    const uint32_t y1 = y0 + rows_per_block < y_end ? y0 + rows_per_block : y_end;
    for (uint32_t t = threadIdx.x; t < T; t += kFramerRowsThreads) {
        const uint64_t b0 = seg_offsets[t], b1 = seg_offsets[t + 1];
        const uint64_t lo = framer_lower_bound_y(ev, b0, b1, y0);
        s_lo[t] = lo;
        s_hi[t] = framer_lower_bound_y(ev, lo, b1, y1);
    }
    __syncthreads();
    // the thread's first event of a frame is fetched while the previous frame is still being
    // applied (events do not depend on the trackers): one memory round trip less per frame
    uint32_t xy = 0, cd = 0, t0 = 0, pxy = 0, pcd = 0, nxy = 0, ncd = 0;
    auto fetch = [&](uint32_t f) {
        const uint64_t lo = s_lo[f], i = lo + threadIdx.x;
        if (i < s_hi[f]) {
            const uint64_t ip = i > lo ? i - 1 : i, in = i + 1 < s_hi[f] ? i + 1 : i;
            nxy = ev[3 * in];
            ncd = ev[3 * in + 1];
            xy = ev[3 * i];
            cd = ev[3 * i + 1];
            t0 = ev[3 * i + 2];
            pxy = ev[3 * ip];
            pcd = ev[3 * ip + 1];
        }
    };
    if (T) fetch(0);
    for (uint32_t f = 0; f < T; ++f) {
        const uint64_t lo = s_lo[f], hi = s_hi[f];
        const uint32_t cxy = xy, ccd = cd, ct = t0, cpxy = pxy, cpcd = pcd, cnxy = nxy, cncd = ncd;
        if (f + 1 < T) fetch(f + 1);
        uint64_t i = lo + threadIdx.x;
        if (i < hi) framer_apply_run(ev, i, lo, hi, a, y0, y1, cxy, ccd, ct, cpxy, cpcd, cnxy, cncd);
        for (i += kFramerRowsThreads; i < hi; i += kFramerRowsThreads) {
            const uint64_t in = i + 1 < hi ? i + 1 : i;
            framer_apply_run(ev, i, lo, hi, a, y0, y1, ev[3 * i], ev[3 * i + 1], ev[3 * i + 2], ev[3 * (i - 1)],
                             ev[3 * (i - 1) + 1], ev[3 * in], ev[3 * in + 1]);
        }
        __syncthreads();  // frame f's tracker updates are visible before frame f + 1 reads them
    }
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
                                                               uint8_t *__restrict__ out) {
    const uint32_t f = (uint32_t)f0 + blockIdx.y;
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
                                                                 uint32_t ring_frames, int32_t f0) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_units) return;
    const int32_t lf = px[i].lastf;
    if (lf < f0) {
        ring[(size_t)((uint32_t)f0 % ring_frames) * n_units + i] = (uint8_t)px[i].lasti;
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
extern "C" hipError_t adder_framer_launch_rows(const void *ev, const uint64_t *d_seg_offsets, uint32_t T,
                                               const FramerArgs *args, hipStream_t s) {
    if (!T) return hipSuccess;
    const FramerArgs a = *args;
    // all workgroups resident at once (4 of 512 threads per CU): every frame step of a workgroup is a
    // chain of memory round trips, so the launch takes T steps however few rows a workgroup owns
    uint32_t rpb = (a.rows + 1023u) / 1024u;
    if (rpb < 1u) rpb = 1u;
    const uint32_t grid = (a.rows + rpb - 1u) / rpb;
    hipLaunchKernelGGL(adder_framer_rows_kernel, dim3(grid), dim3(kFramerRowsThreads), 0, s,
                       static_cast<const uint32_t *>(ev), d_seg_offsets, T, rpb, a);
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
                                              uint8_t *out, hipStream_t s) {
    if (!nf) return hipSuccess;
    uint32_t gx = (n_units + 1023u) / 1024u;
    gx = gx > 2048u ? 2048u : gx;
    hipLaunchKernelGGL(adder_framer_pop_kernel, dim3(gx, nf), dim3(256), 0, s, ring, px, n_units, ring_frames, f0,
                       masked, out);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_flush(uint8_t *ring, FramerPx *px, uint32_t n_units, uint32_t ring_frames,
                                                int32_t f0, hipStream_t s) {
    hipLaunchKernelGGL(adder_framer_flush_kernel, dim3((n_units + 255u) / 256u), dim3(256), 0, s, ring, px, n_units,
                       ring_frames, f0);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_init(FramerPx *px, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(adder_framer_init_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, px, n);
    return hipGetLastError();
}
