// adder_kernel_util.hpp -- device helpers shared by the kernel files (adder_kernels.hip, adder_lp_kernels.hip):
// wave-uniform addressing, cache-policy stores, the DPP scan, the batch's ring layout.  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adder_kernels.h"
#include "adder_pixel.hpp"

namespace adder {

constexpr uint32_t kWave = 64;
constexpr uint32_t kWavesPerBlock = kBlockThreads / kWave;

// diagnostics: a workgroup's start / end into the batch's timeline (no-ops when the batch has none)
__device__ __forceinline__ void timeline_mark(const BatchArgs *b, uint32_t kind, uint32_t f, bool end) {
    if (!b->timeline || threadIdx.x != 0) return;
    const uint32_t chunk = f / b->chunk;
    if (chunk >= kTimelineChunks) return;
    unsigned long long *slot = b->timeline + ((size_t)kind * kTimelineChunks + chunk) * 2u + (end ? 1u : 0u);
    const unsigned long long t = wall_clock64();
    if (end) atomicMax(slot, t);
    else atomicMin(slot, t);
}

// the segment's longest run into BatchArgs::run_max, once it is long enough to matter (kRunReportMin frames: below that the
// wave touches no memory) and only when it beats what the batch's other waves have reported (static content grows every
// unit's run alike: a handful of atomics per launch, not one per wave)
__device__ __forceinline__ void report_run_max(const BatchArgs *__restrict__ b, uint32_t lane_max, uint32_t lane) {
    if (__builtin_amdgcn_ballot_w64(lane_max >= kRunReportMin) == 0ull) return;  // uniform
    uint32_t *const rm = b->run_max;
    if (rm == nullptr) return;
    uint32_t m = lane_max;
#pragma unroll
    for (uint32_t d = 32u; d != 0u; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)m, (int)d);
        m = o > m ? o : m;
    }
    if (lane == 0u && m > __hip_atomic_load(rm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(rm, m);
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

// the batch's ring layout in SGPRs
__device__ __forceinline__ ParkLayout park_layout_u(const BatchArgs *__restrict__ b) {
    ParkLayout l;
    l.group_shift = __builtin_amdgcn_readfirstlane(b->park_layout.group_shift);
    l.group_stride = __builtin_amdgcn_readfirstlane(b->park_layout.group_stride);
    l.frame_stride = __builtin_amdgcn_readfirstlane(b->park_layout.frame_stride);
    l.seg_stride = __builtin_amdgcn_readfirstlane(b->park_layout.seg_stride);
    l.rot_shift = __builtin_amdgcn_readfirstlane(b->park_layout.rot_shift);
    l.rot_mask = __builtin_amdgcn_readfirstlane(b->park_layout.rot_mask);
    return l;
}

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
// The same with the non-temporal hint (`nt`: the line is marked for early eviction): for bytes that are
// touched once -- the event stream on its way out, parked records on their way back in.
template <class V>
__device__ __forceinline__ V gload_nt(const void *base, uint32_t byte_off) {
    using R = typename RawOf<sizeof(V)>::type;
    const R r = __builtin_nontemporal_load(reinterpret_cast<const ADDER_GLOBAL R *>((const ADDER_GLOBAL char *)base + byte_off));
    V v;
    __builtin_memcpy(&v, &r, sizeof(V));
    return v;
}
template <class V>
__device__ __forceinline__ void gstore_nt(void *base, uint32_t byte_off, V v) {
    using R = typename RawOf<sizeof(V)>::type;
    R r;
    __builtin_memcpy(&r, &v, sizeof(V));
    __builtin_nontemporal_store(r, reinterpret_cast<ADDER_GLOBAL R *>((ADDER_GLOBAL char *)base + byte_off));
}
#ifndef ADDER_NT_EVENTS
#define ADDER_NT_EVENTS 1
#endif
#ifndef ADDER_NT_RECLOAD
#define ADDER_NT_RECLOAD 1
#endif
#ifndef ADDER_NT_INPUT
#define ADDER_NT_INPUT 1
#endif
#ifndef ADDER_LDS_DIRECT_INPUT
#define ADDER_LDS_DIRECT_INPUT 1
#endif
#ifndef ADDER_NT_STATE
#define ADDER_NT_STATE 1
#endif
#ifndef ADDER_NT_RECSTORE
#define ADDER_NT_RECSTORE 0
#endif

template <class V>
__device__ __forceinline__ void gstore_ev(void *base, uint32_t byte_off, V v) {
#if defined(ADDER_EV_POLICY_ID) && defined(__HIP_DEVICE_COMPILE__)  // A/B builds: the cache-policy bits of the event stream's 16-byte stores
#if ADDER_EV_POLICY_ID == 1
#define ADDER_EV_POLICY "sc1"
#elif ADDER_EV_POLICY_ID == 2
#define ADDER_EV_POLICY "sc0 sc1"
#elif ADDER_EV_POLICY_ID == 3
#define ADDER_EV_POLICY "nt sc1"
#elif ADDER_EV_POLICY_ID == 4
#define ADDER_EV_POLICY "sc0 sc1 nt"
#elif ADDER_EV_POLICY_ID == 5
#define ADDER_EV_POLICY "sc0"
#else
#define ADDER_EV_POLICY "sc0 nt"
#endif
    if constexpr (sizeof(V) == 16) {
        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
        u4v r;
        __builtin_memcpy(&r, &v, 16);
        ADDER_GLOBAL char *const a64 = (ADDER_GLOBAL char *)base + byte_off;
        asm volatile("global_store_dwordx4 %0, %1, off " ADDER_EV_POLICY : : "v"(a64), "v"(r) : "memory");
        return;
    }
#endif
    if (ADDER_NT_EVENTS) gstore_nt<V>(base, byte_off, v);
    else gstore<V>(base, byte_off, v);
}
template <class V>
__device__ __forceinline__ V gload_rec(const void *base, uint32_t byte_off) {
    if (ADDER_NT_RECLOAD) return gload_nt<V>(base, byte_off);
    return gload<V>(base, byte_off);
}
// a pointer that is the same in every lane, forced into SGPRs
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    const uint64_t x = (uint64_t)p;
    // (the builtin returns int: without the casts the low half would be sign-extended)
    return (T *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x >> 32)) << 32) |
                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)x));
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


// CHAIN (adder_scan_kernel): a frame kernel whose batches chain their frame offsets in the scan zeroes the entries of its
// launch's frames first -- the scan's blocks ADD the chain to them.  Atomics: no line of the table sits dirty in this XCD's L2
// while another XCD adds to it.  nb <= 64 frames per launch.
__device__ __forceinline__ void chain_zero(const BatchArgs *__restrict__ b, uint32_t f, uint32_t nb) {
    if (blockIdx.x != 0u || threadIdx.x >= kWave) return;
    unsigned long long *const offs = reinterpret_cast<unsigned long long *>(b->base.frame_offsets);
    if (threadIdx.x < nb) (void)__hip_atomic_exchange(&offs[f + threadIdx.x + 1u], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f == 0u && threadIdx.x == kWave - 1u) {
        (void)__hip_atomic_exchange(&offs[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b->rec_total)
            (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(b->rec_total), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace adder
