// write_stream.hip -- the ceiling of a pure WRITE stream on one MI355X, in the shapes the expansion kernel could use:
// 477 MB (the events of 64 1080p frames at 0.31 events per pixel-frame) written once with 16-byte lane stores.
//   grid   : grid-stride float4 fill (every wave instruction writes 1 KiB, consecutive waves consecutive KiBs)
//   chunk C: each wave owns a contiguous chunk of C bytes (the expansion: a wave's 16 segments = ~7.7 KB) and writes it
//            1 KiB per instruction; chunks handed out in wave order
// each with plain stores and with the non-temporal hint.  Also `chunkmis`: chunks that start at a 12-byte phase
// (head dwords, 16-byte body, tail dwords -- the expansion's xbuf_flush).
// Build: hipcc --offload-arch=gfx950 -O3 write_stream.hip -o write_stream ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st16(u4 *p, u4 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_grid(u4 *dst, size_t n16) {
    const u4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) st16<NT>(dst + i, v);
}
// one wave per chunk of `c16` 16-byte blocks; blocks of 4 waves
template <bool NT>
__global__ __launch_bounds__(256) void k_chunk(u4 *dst, size_t n16, uint32_t c16) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t w = (size_t)blockIdx.x * 4u + threadIdx.x / 64u;
    const size_t b0 = w * c16;
    if (b0 >= n16) return;
    const u4 v = {lane, (uint32_t)w, 3u, 4u};
    for (uint32_t k = lane; k < c16 && b0 + k < n16; k += 64u) st16<NT>(dst + b0 + k, v);
}
// the same with every chunk starting 4 or 8 bytes off a 16-byte boundary: dword head, 16-byte body, dword tail
template <bool NT>
__global__ __launch_bounds__(256) void k_chunkmis(uint32_t *dst, size_t ndw, uint32_t cdw) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t w = (size_t)blockIdx.x * 4u + threadIdx.x / 64u;
    const size_t d0 = w * cdw;  // cdw is a multiple of 3, not of 4
    if (d0 + cdw > ndw) return;
    const uint32_t head = (4u - (uint32_t)(d0 & 3u)) & 3u;
    if (lane < head) dst[d0 + lane] = lane;
    const uint32_t body = (cdw - head) >> 2;
    u4 *const b = reinterpret_cast<u4 *>(dst + d0 + head);
    const u4 v = {lane, (uint32_t)w, 3u, 4u};
    for (uint32_t k = lane; k < body; k += 64u) st16<NT>(b + k, v);
    const uint32_t tail = (cdw - head) & 3u;
    if (lane < tail) dst[d0 + head + 4u * body + lane] = lane;
}

template <class F>
static float timeit(F launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1000.0f;
}

int main() {
    const size_t bytes = (size_t)477 << 20;
    const size_t n16 = bytes / 16;
    u4 *dst; CHECK(hipMalloc(&dst, bytes + 4096));
    // a second, 2 GiB buffer written between the timed kernels is not needed: every kernel writes 477 MB > the 256 MiB cache
    const int reps = 20;
    auto rep = [&](const char *name, float us) { printf("%-34s %8.1f us  %6.2f TB/s\n", name, us, bytes / us * 1e-6); };
    for (uint32_t g : {2048u, 8192u, 65536u}) {
        char nm[64];
        snprintf(nm, 64, "grid %u plain", g); rep(nm, timeit([&] { hipLaunchKernelGGL(k_grid<false>, dim3(g), dim3(256), 0, 0, dst, n16); }, reps));
        snprintf(nm, 64, "grid %u nt", g);    rep(nm, timeit([&] { hipLaunchKernelGGL(k_grid<true>, dim3(g), dim3(256), 0, 0, dst, n16); }, reps));
    }
    for (uint32_t cb : {1024u, 4096u, 7680u, 16384u, 65536u, 262144u}) {
        const uint32_t c16 = cb / 16;
        const uint32_t waves = (uint32_t)((n16 + c16 - 1) / c16);
        char nm[64];
        snprintf(nm, 64, "chunk %u B plain", cb); rep(nm, timeit([&] { hipLaunchKernelGGL(k_chunk<false>, dim3((waves + 3) / 4), dim3(256), 0, 0, dst, n16, c16); }, reps));
        snprintf(nm, 64, "chunk %u B nt", cb);    rep(nm, timeit([&] { hipLaunchKernelGGL(k_chunk<true>, dim3((waves + 3) / 4), dim3(256), 0, 0, dst, n16, c16); }, reps));
    }
    for (uint32_t ev : {640u, 1280u}) {
        const uint32_t cdw = ev * 3u + 3u;  // (+3: the phase walks through 0..3)
        const size_t ndw = bytes / 4;
        const uint32_t waves = (uint32_t)(ndw / cdw);
        char nm[64];
        snprintf(nm, 64, "chunkmis %u ev plain", ev); rep(nm, timeit([&] { hipLaunchKernelGGL(k_chunkmis<false>, dim3((waves + 3) / 4), dim3(256), 0, 0, (uint32_t *)dst, ndw, cdw); }, reps));
        snprintf(nm, 64, "chunkmis %u ev nt", ev);    rep(nm, timeit([&] { hipLaunchKernelGGL(k_chunkmis<true>, dim3((waves + 3) / 4), dim3(256), 0, 0, (uint32_t *)dst, ndw, cdw); }, reps));
    }
    return 0;
}
