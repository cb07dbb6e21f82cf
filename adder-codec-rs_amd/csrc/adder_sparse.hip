// adder_sparse.hip -- sparse steps of event-camera sources on the device (the callers of
// integrate_for_px in prophesee.rs:170-258, 330-372 and davis.rs).
//
// The host turns a camera's events into STEPS {pixel, frame_val, intensity, time} in the camera's order; each step
// is integrate_for_px(px, &mut 0, frame_val, intensity, time) on a Mode::Continuous arena, and all events go to one
// buffer in step order.  Pixels are independent, a pixel's steps are not, and a pixel turns up anywhere in the
// list.  So:
//   1. key = unit index, value = step index; a stable radix sort brings a unit's steps together, in order;
//   2. a thread per RUN of equal units walks it with cont_step (adder_pixel.hpp), c_thresh / its counter /
//      running_t per unit (they advance per integrate call, and calls per pixel differ here), and stages each
//      step's events at a fixed slot range of the step + a count;
//   3. an exclusive scan of the counts in STEP order places every step's events;
//   4. a thread per step copies them out with the step's coordinates.
// hipCUB (rocPRIM) does the sort and the scan.  Nothing here is on the framed hot path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "adder_kernels.h"
#include "adder_pixel.hpp"

namespace adder {

struct SparseNodes {  // a unit's arena nodes in the node planes
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
struct SparseEmit {
    uint2 *dst;
    uint32_t n, cap;
    __device__ __forceinline__ void operator()(uint32_t d, uint32_t t) {
        if (n < cap) dst[n] = make_uint2(t, d);
        ++n;
    }
};

__device__ __forceinline__ bool sparse_unit(const SparseArgs &a, const SparseStep &s, uint32_t &u) {
    const uint32_t c = s.c == 0xffu ? 0u : s.c;
    const bool ok = s.x < a.width && s.y >= a.row_begin && s.y - a.row_begin < a.rows && c < a.channels;
    u = ((s.y - a.row_begin) * a.width + s.x) * a.channels + c;
    return ok;
}

__global__ __launch_bounds__(256) void adder_sparse_keys_kernel(const SparseStep *__restrict__ steps, uint32_t n,
                                                                SparseArgs a, uint32_t *__restrict__ keys,
                                                                uint32_t *__restrict__ idx) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t u;
    if (!sparse_unit(a, steps[i], u)) {
        atomicOr(a.status, kStatusSparse);
        u = 0xffffffffu;  // sorted to the end, never walked
    }
    keys[i] = u;
    idx[i] = i;
}

template <bool ABS_T>
__global__ __launch_bounds__(256) void adder_sparse_run_kernel(const SparseStep *__restrict__ steps, uint32_t n,
                                                               SparseArgs a, const uint32_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ idx,
                                                               uint2 *__restrict__ stage, uint32_t *__restrict__ count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = keys[i];
    if (u == 0xffffffffu) {
        count[idx[i]] = 0u;
        return;
    }
    if (i > 0u && keys[i - 1] == u) return;  // not the first step of its unit's run
    APx s = apx_unpack(a.hdr[u], a.lastf[u]);
    SparseNodes acc{a.cn_integ, a.cn_dt, a.cn_bdt, a.cn_meta, a.plane_stride, u};
    uint8_t cth = a.cth_px[u], cctr = a.cctr_px[u];
    float rt = a.rt_px[u];
    StepConsts sc = a.sc;
    bool bad = false, side_set = false;
    uint32_t side = 0u;
    for (uint32_t j = i; j < n && keys[j] == u; ++j) {
        const uint32_t k = idx[j];
        const SparseStep st = steps[k];
        // (the `let mut base_val = 0` in front of every call, prophesee.rs:206,244,334, is an OUT parameter:
        // integrate_for_px overwrites it with px.base_val before the contrast test, video.rs:1336 -- the test sees
        // the pixel's persisted base_val)
        sc.time_spanned = st.time;
        sc.running_t = rt;
        sc.running_t_u32 = f32_as_u32(rt);
        sc.cth = cth;
        SparseEmit em{stage + (size_t)k * a.stage_events, 0u, a.stage_events};
        const bool ok = cont_step<ABS_T>(s, acc, st.frame_val, st.intensity, st.time, sc, a.max_nodes, em, st.pad);
        bad = bad || !ok || em.n > em.cap;
        count[k] = em.n < em.cap ? em.n : em.cap;
        if (!(st.pad & (kSparseTestOnly | kSparseFlush))) {  // (both advance inside PixelArena::integrate)
            rt += st.time;  // `self.running_t += time` (event_pixel_tree.rs:336)
            c_thresh_advance(cth, cctr, (uint8_t)a.c_max, (uint8_t)a.c_vel, st.time, sc.ref_time);
        }
        // side plane (prophesee.rs:259-283): sampled once per CAMERA event, after its last integrate_for_px call --
        // a step flagged ADDER_SPARSE_NO_SIDE (the first of a camera event's two, or an end_events step) is not sampled
        if (a.running && !(st.pad & 1u)) {
            const ANode r = acc.load(0);
            if (r.has_best) {
                side = frame_value_u8(r.bd, f32_as_u32(r.bdt), (double)sc.ref_time);
                side_set = true;
            }
        }
    }
    a.hdr[u] = apx_hdr(s);
    a.lastf[u] = s.lastf;
    a.cth_px[u] = cth;
    a.cctr_px[u] = cctr;
    a.rt_px[u] = rt;
    if (side_set) a.running[u] = (uint8_t)side;
    if (bad) atomicOr(a.status, kStatusDepth);
}

// offs = exclusive prefix of count (step order), offs[n] = total
__global__ __launch_bounds__(256) void adder_sparse_emit_kernel(const SparseStep *__restrict__ steps, uint32_t n,
                                                                SparseArgs a, const uint2 *__restrict__ stage,
                                                                const uint32_t *__restrict__ count,
                                                                const uint32_t *__restrict__ offs,
                                                                AdderEventPod *__restrict__ out, uint64_t out_cap,
                                                                unsigned long long *__restrict__ total_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = count[i], o = offs[i];
    if (i == n - 1u) {
        *total_out = (unsigned long long)o + c;
        if ((uint64_t)o + c > out_cap) atomicOr(a.status, kStatusCapacity);
    }
    const SparseStep st = steps[i];
    for (uint32_t e = 0; e < c; ++e) {
        if ((uint64_t)o + e >= out_cap) break;
        const uint2 r = stage[(size_t)i * a.stage_events + e];
        AdderEventPod ev;
        ev.x = st.x;
        ev.y = st.y;
        ev.c = st.c;
        ev.d = (uint8_t)r.y;
        ev.pad = 0;
        ev.t = r.x;
        out[(size_t)o + e] = ev;
    }
}

}  // namespace adder

using namespace adder;

extern "C" size_t adder_sparse_temp_bytes(uint32_t n) {
    size_t a = 0, b = 0;
    hipcub::DoubleBuffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, k, v, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
    return (a > b ? a : b) + 256;
}

// keys / idx: two buffers of n uint32 each; stage: n * stage_events uint2; count, offs: n uint32
extern "C" hipError_t adder_sparse_run(const SparseArgs *args, const SparseStep *d_steps, uint32_t n, uint32_t *keys0,
                                       uint32_t *keys1, uint32_t *idx0, uint32_t *idx1, void *d_temp, size_t temp_bytes,
                                       uint2 *stage, uint32_t *count, uint32_t *offs, AdderEventPod *d_out,
                                       uint64_t out_cap, unsigned long long *d_total, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const SparseArgs a = *args;
    const uint32_t grid = (n + 255u) / 256u;
    hipLaunchKernelGGL(adder_sparse_keys_kernel, dim3(grid), dim3(256), 0, stream, d_steps, n, a, keys0, idx0);
    hipcub::DoubleBuffer<uint32_t> k(keys0, keys1), v(idx0, idx1);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, k, v, (int)n, 0, 32, stream);
    if (e != hipSuccess) return e;
    if (a.sc.abs_t)
        hipLaunchKernelGGL((adder_sparse_run_kernel<true>), dim3(grid), dim3(256), 0, stream, d_steps, n, a, k.Current(),
                           v.Current(), stage, count);
    else
        hipLaunchKernelGGL((adder_sparse_run_kernel<false>), dim3(grid), dim3(256), 0, stream, d_steps, n, a, k.Current(),
                           v.Current(), stage, count);
    e = hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, count, offs, (int)n, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(adder_sparse_emit_kernel, dim3(grid), dim3(256), 0, stream, d_steps, n, a, stage, count, offs, d_out,
                       out_cap, d_total);
    return hipGetLastError();
}
