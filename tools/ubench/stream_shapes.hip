// stream_shapes.hip -- what the memory system of one MI355X gives a "read 16 B of state + 1 B of input per unit,
// write 16 B of state" kernel of the size of one 1080p frame (2 073 600 units, 68 MB), as a function of how the
// state is laid out and accessed.  Calibration for the one-frame-per-launch kernel (adder_lean1w_kernel): the
// compute is a token add, so what is measured is the access shape alone.
//   copy      : float4 grid copy of the same number of bytes (reference point for this size)
//   soa8      : four planes, 8-byte accesses, 2 units per lane           (adder_lean1_kernel's shape)
//   soa16     : four planes, 16-byte accesses, 4 units per lane          (adder_lean1w_kernel's shape)
//   aos16     : one array of 16-byte unit records, unit = j * 64 + lane   (every wave access is 1 KiB contiguous)
//   aos16s    : the same array, 4 consecutive units per lane              (64-byte stride between lanes)
// P = tiles per wave (all loads first), W = waves per SIMD asked for.
// Build: hipcc --offload-arch=gfx950 -O3 stream_shapes.hip -o stream_shapes ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_copy(const u4 *__restrict__ src, u4 *__restrict__ dst, uint32_t n16) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i] + 1u;
}

template <int P, int W>
__global__ __launch_bounds__(256, W) void k_soa8(uint32_t *p0, uint32_t *p1, uint32_t *p2, uint32_t *p3, const uint8_t *in,
                                                 uint32_t n_tiles) {  // tile = 128 units
    const uint32_t lane = threadIdx.x & 63u, w = (blockIdx.x * 4u + threadIdx.x / 64u) * P;
    if (w >= n_tiles) return;
    u2 a[P], b[P], c[P], d[P];
    uint32_t v[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const uint32_t u = (min(w + s, n_tiles - 1u)) * 128u + lane * 2u;
        a[s] = *(const u2 *)(p0 + u);
        v[s] = *(const uint16_t *)(in + u);
        b[s] = *(const u2 *)(p1 + u);
        c[s] = *(const u2 *)(p2 + u);
        d[s] = *(const u2 *)(p3 + u);
    }
#pragma unroll
    for (int s = 0; s < P; ++s) {
        if (w + s >= n_tiles) break;
        const uint32_t u = (w + s) * 128u + lane * 2u;
        *(u2 *)(p0 + u) = a[s] + v[s];
        *(u2 *)(p1 + u) = b[s] + 1u;
        *(u2 *)(p2 + u) = c[s] + 1u;
        *(u2 *)(p3 + u) = d[s] + 1u;
    }
}

// WORK rounds of 32 dependent VALU instructions on a tile's 16 loaded words (8 rounds ~ the lean step's 250 per pair)
__device__ __forceinline__ void fake_work(u4 &a, u4 &b, u4 &c, u4 &d, int rounds) {
    for (int r = 0; r < rounds; ++r) {
        a = (a + b) ^ 0x9e3779b9u;
        b = (b + c) ^ 0x7f4a7c15u;
        c = (c + d) ^ 0x85ebca6bu;
        d = (d + a) ^ 0xc2b2ae35u;
    }
}

template <int P, int W, int WORK>
__global__ __launch_bounds__(256, W) void k_soa16(uint32_t *p0, uint32_t *p1, uint32_t *p2, uint32_t *p3, const uint8_t *in,
                                                  uint32_t n_tiles) {  // tile = 256 units
    const uint32_t lane = threadIdx.x & 63u, w = (blockIdx.x * 4u + threadIdx.x / 64u) * P;
    if (w >= n_tiles) return;
    u4 a[P], b[P], c[P], d[P];
    uint32_t v[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const uint32_t u = (min(w + s, n_tiles - 1u)) * 256u + lane * 4u;
        a[s] = *(const u4 *)(p0 + u);
        v[s] = *(const uint32_t *)(in + u);
        b[s] = *(const u4 *)(p1 + u);
        c[s] = *(const u4 *)(p2 + u);
        d[s] = *(const u4 *)(p3 + u);
    }
#pragma unroll
    for (int s = 0; s < P; ++s) {
        if (w + s >= n_tiles) break;
        const uint32_t u = (w + s) * 256u + lane * 4u;
        fake_work(a[s], b[s], c[s], d[s], WORK);
        *(u4 *)(p0 + u) = a[s] + v[s];
        *(u4 *)(p1 + u) = b[s] + 1u;
        *(u4 *)(p2 + u) = c[s] + 1u;
        *(u4 *)(p3 + u) = d[s] + 1u;
    }
}

// STRIDED = false: unit = tile * 256 + j * 64 + lane; true: unit = tile * 256 + lane * 4 + j
template <int P, int W, bool STRIDED>
__global__ __launch_bounds__(256, W) void k_aos16(u4 *rec, const uint8_t *in, uint32_t n_tiles) {
    const uint32_t lane = threadIdx.x & 63u, w = (blockIdx.x * 4u + threadIdx.x / 64u) * P;
    if (w >= n_tiles) return;
    u4 r[P][4];
    uint32_t v[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const uint32_t t = min(w + s, n_tiles - 1u) * 256u;
        v[s] = *(const uint32_t *)(in + t + lane * 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[s][j] = rec[STRIDED ? t + lane * 4u + j : t + j * 64u + lane];
    }
#pragma unroll
    for (int s = 0; s < P; ++s) {
        if (w + s >= n_tiles) break;
        const uint32_t t = (w + s) * 256u;
#pragma unroll
        for (int j = 0; j < 4; ++j) rec[STRIDED ? t + lane * 4u + j : t + j * 64u + lane] = r[s][j] + v[s];
    }
}

template <class F>
static float time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int k = 0; k < 5; ++k) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best * 1000.f / reps;
}

int main() {
    const uint32_t units = 1920u * 1080u;                 // one 1080p gray frame
    const uint32_t n_pad = (units + 2047u) / 2048u * 2048u;
    uint32_t *planes;
    uint8_t *in;
    CHECK(hipMalloc(&planes, (size_t)n_pad * 16u + 4096u));
    CHECK(hipMalloc(&in, n_pad + 4096u));
    CHECK(hipMemset(planes, 0, (size_t)n_pad * 16u));
    CHECK(hipMemset(in, 1, n_pad));
    uint32_t *p0 = planes, *p1 = planes + n_pad, *p2 = planes + 2 * (size_t)n_pad, *p3 = planes + 3 * (size_t)n_pad;
    u4 *copy_dst;
    CHECK(hipMalloc(&copy_dst, (size_t)n_pad * 16u));
    const double bytes = (double)n_pad * 33.0;
    const int reps = 200;
#define REPORT(name, us) printf("%-22s %8.2f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, us, bytes / (us)*1e-3, bytes / (us)*1e-3 / 8000.0)
    {
        const uint32_t n16 = n_pad;  // n_pad float4 = 16 B per unit read + 16 B written
        for (uint32_t blocks : {2048u, 4096u, 8192u, n16 / 256u}) {
            const float us = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const u4 *)planes, copy_dst, n16); }, reps);
            char nm[64];
            snprintf(nm, sizeof nm, "copy blocks=%u", blocks);
            printf("%-22s %8.2f us  %7.1f GB/s (32 B per unit)\n", nm, us, (double)n_pad * 32.0 / us * 1e-3);
        }
    }
#define RUN_SOA8(P, W)                                                                                                        \
    {                                                                                                                         \
        const uint32_t tiles = n_pad / 128u, waves = (tiles + P - 1) / P, blocks = (waves + 3) / 4;                           \
        const float us = time_us([&] { hipLaunchKernelGGL((k_soa8<P, W>), dim3(blocks), dim3(256), 0, 0, p0, p1, p2, p3, in, tiles); }, reps); \
        REPORT("soa8  P=" #P " W=" #W, us);                                                                                   \
    }
#define RUN_SOA16(P, W, K)                                                                                                     \
    {                                                                                                                         \
        const uint32_t tiles = n_pad / 256u, waves = (tiles + P - 1) / P, blocks = (waves + 3) / 4;                           \
        const float us = time_us([&] { hipLaunchKernelGGL((k_soa16<P, W, K>), dim3(blocks), dim3(256), 0, 0, p0, p1, p2, p3, in, tiles); }, reps); \
        REPORT("soa16 P=" #P " W=" #W " work=" #K, us);                                                                                  \
    }
#define RUN_AOS16(P, W, S)                                                                                                    \
    {                                                                                                                         \
        const uint32_t tiles = n_pad / 256u, waves = (tiles + P - 1) / P, blocks = (waves + 3) / 4;                           \
        const float us = time_us([&] { hipLaunchKernelGGL((k_aos16<P, W, S>), dim3(blocks), dim3(256), 0, 0, (u4 *)planes, in, tiles); }, reps); \
        REPORT(S ? "aos16s P=" #P " W=" #W : "aos16 P=" #P " W=" #W, us);                                                     \
    }
    RUN_SOA8(4, 4)
    RUN_SOA8(2, 8)
    RUN_SOA16(1, 4, 0)
    RUN_SOA16(2, 4, 0)
    RUN_SOA16(1, 8, 0)
    RUN_SOA16(4, 2, 0)
    RUN_SOA16(1, 4, 8)
    RUN_SOA16(2, 4, 8)
    RUN_SOA16(1, 8, 8)
    RUN_SOA16(2, 8, 8)
    RUN_SOA16(4, 2, 8)
    RUN_SOA16(4, 4, 8)
    RUN_SOA16(2, 4, 16)
    RUN_SOA16(1, 8, 16)
    RUN_SOA16(2, 8, 16)
    RUN_AOS16(1, 4, false)
    RUN_AOS16(2, 4, false)
    RUN_AOS16(1, 8, false)
    RUN_AOS16(2, 2, false)
    RUN_AOS16(1, 4, true)
    RUN_AOS16(2, 4, true)
    return 0;
}
