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

// events [e0, e1) of one segment; ev = 3 dwords per event {x | y<<16, c | d<<8, t}
__global__ __launch_bounds__(256) void adder_framer_segment_kernel(const uint32_t *__restrict__ ev, uint64_t e0,
                                                                   uint64_t e1, FramerArgs a) {
    const uint64_t i = e0 + (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= e1) return;
    const uint32_t xy = ev[3 * i], cd = ev[3 * i + 1];
    const uint32_t key_c = (cd & 0xffu) == 0xffu ? 0u : (cd & 0xffu);
    if (i > e0) {  // not the first event of its pixel's run: the run's leader handles it
        const uint32_t pxy = ev[3 * (i - 1)], pcd = ev[3 * (i - 1) + 1];
        const uint32_t pc = (pcd & 0xffu) == 0xffu ? 0u : (pcd & 0xffu);
        if (pxy == xy && pc == key_c) return;
    }
    bool ok;
    const uint32_t u = unit_of(a, xy, cd, ok);
    if (!ok) {
        atomicOr(a.status, kFramerStatusMalformed);
        return;
    }
    FramerPx p;
    p.ts = a.ts[u];
    p.lastf = a.lastf[u];
    p.lasti = a.lasti[u];
    uint32_t flags = 0u;
    for (uint64_t j = i; j < e1; ++j) {
        uint32_t w1 = cd, t;
        if (j != i) {
            const uint32_t jxy = ev[3 * j];
            w1 = ev[3 * j + 1];
            const uint32_t jc = (w1 & 0xffu) == 0xffu ? 0u : (w1 & 0xffu);
            if (jxy != xy || jc != key_c) break;
        }
        t = ev[3 * j + 2];
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
    a.ts[u] = p.ts;
    a.lastf[u] = p.lastf;
    a.lasti[u] = (uint8_t)p.lasti;
    if (flags) atomicOr(a.status, flags);
}

// min / max of last_filled over the band: frames [frames_written, min + 1) are complete
__global__ __launch_bounds__(256) void adder_framer_minmax_kernel(const int32_t *__restrict__ lastf, uint32_t n,
                                                                  int32_t *out /* [2] = {min, max} */) {
    int32_t mn = 0x7fffffff, mx = -0x7fffffff - 1;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const int32_t v = lastf[i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63u) == 0u) {
        atomicMin(&out[0], mn);
        atomicMax(&out[1], mx);
    }
}

// frames [f0, f0 + nf) of the ring -> out[nf][n_units]; masked: a pixel without a value in the
// frame (last_filled < f) reads as 0 (write_frame_bytes' `None => T::default()`, driver.rs:946-950)
__global__ __launch_bounds__(256) void adder_framer_pop_kernel(const uint8_t *__restrict__ ring,
                                                               const int32_t *__restrict__ lastf, uint32_t n_units,
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
                dst[j] = (!masked || lastf[j] >= (int32_t)f) ? src[j] : (uint8_t)0;
        }
    }
}

// flush_frame_buffer (driver.rs:632-677): every pixel without a value in frame f0 gets its last
// intensity there and its last_filled advances by ONE (as the reference does)
__global__ __launch_bounds__(256) void adder_framer_flush_kernel(uint8_t *ring, int32_t *lastf,
                                                                 const uint8_t *__restrict__ lasti, uint32_t n_units,
                                                                 uint32_t ring_frames, int32_t f0) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_units) return;
    const int32_t lf = lastf[i];
    if (lf < f0) {
        ring[(size_t)((uint32_t)f0 % ring_frames) * n_units + i] = lasti[i];
        lastf[i] = lf + 1;
    }
}

__global__ void adder_framer_init_kernel(uint64_t *ts, int32_t *lastf, uint8_t *lasti, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    ts[i] = 0;
    lastf[i] = -1;
    lasti[i] = 0;
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
extern "C" hipError_t adder_framer_launch_minmax(const int32_t *lastf, uint32_t n, int32_t *out, hipStream_t s) {
    const uint32_t grid = (n + 256u * 16u - 1u) / (256u * 16u);
    hipLaunchKernelGGL(adder_framer_minmax_kernel, dim3(grid < 1024u ? (grid ? grid : 1u) : 1024u), dim3(256), 0, s, lastf,
                       n, out);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_pop(const uint8_t *ring, const int32_t *lastf, uint32_t n_units,
                                              uint32_t ring_frames, int32_t f0, uint32_t nf, uint32_t masked,
                                              uint8_t *out, hipStream_t s) {
    if (!nf) return hipSuccess;
    uint32_t gx = (n_units + 1023u) / 1024u;
    gx = gx > 2048u ? 2048u : gx;
    hipLaunchKernelGGL(adder_framer_pop_kernel, dim3(gx, nf), dim3(256), 0, s, ring, lastf, n_units, ring_frames, f0,
                       masked, out);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_flush(uint8_t *ring, int32_t *lastf, const uint8_t *lasti, uint32_t n_units,
                                                uint32_t ring_frames, int32_t f0, hipStream_t s) {
    hipLaunchKernelGGL(adder_framer_flush_kernel, dim3((n_units + 255u) / 256u), dim3(256), 0, s, ring, lastf, lasti,
                       n_units, ring_frames, f0);
    return hipGetLastError();
}
extern "C" hipError_t adder_framer_launch_init(uint64_t *ts, int32_t *lastf, uint8_t *lasti, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(adder_framer_init_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, ts, lastf, lasti, n);
    return hipGetLastError();
}
