// lds_writes.hip -- what one wave64 LDS store instruction costs a gfx950 CU, by size and alignment: the expansion stages every
// event in its final 9 / 11 / 12 bytes, so a 9-byte record is an 8-byte store at an odd address + a byte.
// Build: hipcc --offload-arch=gfx950 -O2 lds_writes.hip -o lds_writes ; run on the GPU box.  Prints shader cycles per
// instruction per CU (16 waves per CU resident, every wave in the same loop).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int KIND>
__global__ __launch_bounds__(256) void ub(uint64_t *out, int iters, uint32_t pitch) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[4][8192];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t addr = (uint32_t)(uintptr_t)(buf[w]) + lane * pitch;  // (LDS byte address)
    uint32_t a = threadIdx.x, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u;
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u2 v2 = {a, b};
    u3 v3 = {a, b, c};
    u4 v4 = {a, b, c, d};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(REP16("ds_write_b64 %0, %1\n") : : "v"(addr), "v"(v2) : "memory");
        if (KIND == 1) asm volatile(REP16("ds_write_b8 %0, %1 offset:8\n") : : "v"(addr), "v"(a) : "memory");
        if (KIND == 2) asm volatile(REP16("ds_write_b32 %0, %1\n") : : "v"(addr), "v"(a) : "memory");
        if (KIND == 3) asm volatile(REP16("ds_write_b96 %0, %1\n") : : "v"(addr), "v"(v3) : "memory");
        if (KIND == 4) asm volatile(REP16("ds_write_b128 %0, %1\n") : : "v"(addr), "v"(v4) : "memory");
        if (KIND == 5) asm volatile(REP16("ds_write2_b32 %0, %1, %2 offset1:1\n") : : "v"(addr), "v"(a), "v"(b) : "memory");
        if (KIND == 6) asm volatile(REP16("ds_write_b16 %0, %1 offset:8\n") : : "v"(addr), "v"(a) : "memory");
        if (KIND == 7) asm volatile(REP16("ds_read_b128 %0, %1\n") : "=v"(v4) : "v"(addr) : "memory");
        if (KIND == 8) asm volatile(REP16("ds_read_b64 %0, %1\n") : "=v"(v2) : "v"(addr) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + v2.x + v4.x == 0x12345678u) out[0] = buf[w][lane];
}

template <int KIND>
static void run(const char *name, uint32_t pitch, uint64_t *d_out, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL(ub<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 10, pitch);
    hipLaunchKernelGGL(ub<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, pitch);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    // s_memtime counts at 100 MHz; convert with the shader clock measured elsewhere (~2.4 GHz): report ns per instruction per CU
    // 16 waves per CU (4 blocks of 4 waves), each issuing iters * 16 instructions
    const double ns = avg * 10.0;
    printf("%-28s pitch %3u : %7.2f ns per wave-instruction per CU (%.1f cycles at 2.4 GHz)\n", name, pitch, ns / (iters * 16.0 * 16.0),
           ns / (iters * 16.0 * 16.0) * 2.4);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4;  // 4 blocks of 4 waves per CU
    uint64_t *d_out;
    hipMalloc(&d_out, blocks * 8);
    for (uint32_t pitch : {8u, 9u, 12u, 16u, 18u, 27u}) run<0>("ds_write_b64", pitch, d_out, blocks);
    for (uint32_t pitch : {9u, 12u, 27u}) run<1>("ds_write_b8", pitch, d_out, blocks);
    for (uint32_t pitch : {4u, 9u, 12u}) run<2>("ds_write_b32", pitch, d_out, blocks);
    for (uint32_t pitch : {12u, 16u, 36u}) run<3>("ds_write_b96", pitch, d_out, blocks);
    for (uint32_t pitch : {16u}) run<4>("ds_write_b128", pitch, d_out, blocks);
    for (uint32_t pitch : {8u, 12u, 9u}) run<5>("ds_write2_b32", pitch, d_out, blocks);
    for (uint32_t pitch : {9u, 11u}) run<6>("ds_write_b16", pitch, d_out, blocks);
    for (uint32_t pitch : {16u}) run<7>("ds_read_b128", pitch, d_out, blocks);
    for (uint32_t pitch : {8u, 9u}) run<8>("ds_read_b64", pitch, d_out, blocks);
    return 0;
}
