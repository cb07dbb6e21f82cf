// Issue rates on gfx950: how many cycles a CU spends per SALU / VALU instruction when every SIMD holds W waves of the same
// straight-line loop (tools/probes: measurements behind DESIGN.md's issue model, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 issue_probe.hip -o issue_probe && ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t *out, int iters) {
    uint32_t v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 16 SALU (64-bit mask logic is what the kernels do; 32-bit adds here: same unit)
            REP16(asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_and_b32 %2, %2, %3\n s_or_b32 %3, %3, %0" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));)
        } else if (MODE == 1) {  // 16 x 4 VALU
            REP16(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_or_b32 %3, %3, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
        } else if (MODE == 2) {  // both, interleaved
            REP16(asm volatile("s_add_u32 %0, %0, %1\n v_add_u32 %4, %4, %5\n s_xor_b32 %1, %1, %2\n v_xor_b32 %5, %5, %6\n s_and_b32 %2, %2, %3\n v_and_b32 %6, %6, %7\n s_or_b32 %3, %3, %0\n v_or_b32 %7, %7, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
        } else if (MODE == 3) {  // 64-bit scalar mask logic
            uint64_t m0 = s0, m1 = s1;
            REP16(asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_andn2_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0" : "+s"(m0), "+s"(m1));)
            s0 += (uint32_t)m0 + (uint32_t)m1;
        } else if (MODE == 4) {  // v_cmp writing an SGPR pair + v_cndmask reading it
            uint64_t m0;
            REP16(asm volatile("v_cmp_ne_u32 %4, %0, %1\n v_cndmask_b32 %2, %2, %3, %4\n v_cmp_lt_u32 %4, %1, %2\n v_cndmask_b32 %0, %0, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=s"(m0));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3;
}
int main() {
    uint32_t *d;
    const int cus = 256;
    hipMalloc(&d, (size_t)cus * 8 * 256 * 4 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 2000;
    const char *names[] = {"SALU 32-bit", "VALU", "SALU+VALU interleaved", "SALU 64-bit masks", "v_cmp->sgpr + v_cndmask"};
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int grid = cus * waves_per_simd;  // 256-thread blocks: 4 waves, one per SIMD
        for (int mode = 0; mode < 5; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                switch (mode) {
                    case 0: probe<0><<<grid, 256>>>(d, iters); break;
                    case 1: probe<1><<<grid, 256>>>(d, iters); break;
                    case 2: probe<2><<<grid, 256>>>(d, iters); break;
                    case 3: probe<3><<<grid, 256>>>(d, iters); break;
                    case 4: probe<4><<<grid, 256>>>(d, iters); break;
                }
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const double per_kind = (double)iters * 64.0;  // instructions of one kind per wave (mode 2: 64 of each)
            // cycles (2.4 GHz) the CU spends per instruction of one wave-set: time / (waves per SIMD * instructions)
            const double cyc = best * 1e-3 * 2.4e9 / (waves_per_simd * per_kind);
            printf("waves/SIMD %d  %-26s %8.3f ms  %6.2f cycles per instruction per SIMD-wave\n", waves_per_simd, names[mode], best, cyc);
        }
    }
    return 0;
}
