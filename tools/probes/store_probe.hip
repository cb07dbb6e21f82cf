// Store bandwidth on gfx950 for the expansion's pattern: every wave writes one contiguous run of R bytes with 16-byte
// stores per lane (1 KB per instruction), runs back to back in launch order; against hipMemsetAsync.
//   hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void fill(uint4 *out, size_t run16 /* uint4 per wave run */, int nt) {
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
    uint4 *p = out + wave * run16;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t k = threadIdx.x % 64; k < run16; k += 64) {
        if (nt) __builtin_nontemporal_store(v.x, &p[k].x), __builtin_nontemporal_store(v.y, &p[k].y), __builtin_nontemporal_store(v.z, &p[k].z), __builtin_nontemporal_store(v.w, &p[k].w);
        else p[k] = v;
    }
}
__global__ __launch_bounds__(256) void copy(const uint4 *in, uint4 *out, size_t n) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) out[k] = in[k];
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    uint4 *d, *s;
    hipMalloc(&d, bytes);
    hipMalloc(&s, bytes);
    hipMemset(s, 1, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto timeit = [&](const char *name, auto fn, double moved) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            hipEventRecord(a);
            fn();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, moved / best / 1e6);
    };
    timeit("hipMemsetAsync 2 GiB", [&] { hipMemsetAsync(d, 0, bytes, 0); }, (double)bytes);
    for (size_t run : {1024, 4096, 6144, 16384, 65536}) {
        const size_t waves = bytes / run;
        char name[64];
        snprintf(name, sizeof name, "fill, %zu-byte run per wave", run);
        timeit(name, [&] { fill<<<(unsigned)(waves / 4), 256>>>(d, run / 16, 0); }, (double)bytes);
    }
    timeit("fill nt, 6144-byte run per wave", [&] { fill<<<(unsigned)(bytes / 6144 / 4), 256>>>(d, 6144 / 16, 1); }, (double)(bytes / 6144 / 4) * 4 * 6144);
    timeit("copy 2 GiB -> 2 GiB (read + write bytes)", [&] { copy<<<256 * 16, 256>>>(s, d, bytes / 16); }, 2.0 * bytes);
    return 0;
}
