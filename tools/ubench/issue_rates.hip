// issue_rates.hip -- gfx950 issue-rate calibration for the K1 design: how many shader cycles one
// wave64 instruction of each kind costs a SIMD, alone and mixed with SALU work, at 1..8 waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 issue_rates.hip -o issue_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int KIND>
__global__ __launch_bounds__(256) void ub(uint64_t *out, int iters) {
    uint32_t a = threadIdx.x, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u;
    float fa = (float)a + 1.0f, fb = 1.0001f, fc = 0.5f, fd = 2.0f;
    uint32_t s0 = blockIdx.x, s1 = 3, s2 = 5, s3 = 7;
    s0 = __builtin_amdgcn_readfirstlane(s0);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0)
            asm volatile(REP8("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 1)
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n")
                         : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));
        if (KIND == 2)
            asm volatile(REP8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (KIND == 3)
            asm volatile(REP8("v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 s[20:21], %1, %2\n v_cmp_gt_u32 s[22:23], %2, %3\n v_cmp_gt_u32 s[24:25], %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");
        if (KIND == 4)
            asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
                         : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));
        if (KIND == 5)
            asm volatile(REP8("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        if (KIND == 6)  // 32 VALU + 32 SALU interleaved
            asm volatile(REP8("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_add_u32 %1, %1, %2\n s_add_u32 %5, %5, %6\n v_add_u32 %2, %2, %3\n s_add_u32 %6, %6, %7\n v_add_u32 %3, %3, %0\n s_add_u32 %7, %7, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        if (KIND == 7)  // 32 VALU + 16 SALU
            asm volatile(REP8("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n s_add_u32 %6, %6, %7\n v_add_u32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        if (KIND == 8)
            asm volatile(REP8("v_cvt_f32_ubyte1 %0, %1\n v_sad_u32 %1, %1, %2, %3\n v_bfe_u32 %2, %3, 8, 8\n v_mbcnt_lo_u32_b32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 9)
            asm volatile(REP8("v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_add_f32 %0, %0, %1\n v_pk_fma_f32 %1, %1, %0, %0\n")
                         : "+v"(*(double *)&fa), "+v"(*(double *)&fc));
        if (KIND == 10)
            asm volatile(REP8("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %0 row_bcast:15 row_mask:0xa\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 11)  // s_and_b64 style mask logic
            asm volatile(REP8("s_and_b64 s[20:21], s[20:21], vcc\n s_or_b64 s[22:23], s[22:23], s[20:21]\n s_andn2_b64 s[24:25], s[24:25], s[22:23]\n s_bcnt1_i32_b64 %0, s[24:25]\n")
                         : "+s"(s0) : : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");
        if (KIND == 12)  // 32 VALU + 32 v_cvt_u32_f32 (mixed normal ALU + conversion)
            asm volatile(REP8("v_cvt_u32_f32 %0, %4\n v_cvt_f32_u32 %4, %1\n v_cvt_u32_f32 %2, %5\n v_cvt_f32_u32 %5, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(fa), "+v"(fb));
        if (KIND == 13)  // v_cmp writing sgpr + s_and on it + v_cndmask reading it (dependent mask chain)
            asm volatile(REP8("v_cmp_gt_u32 s[20:21], %0, %1\n s_and_b64 s[22:23], s[20:21], s[24:25]\n v_cndmask_b32 %2, %2, %3, s[22:23]\n v_add_u32 %0, %0, %2\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");

        if (KIND == 14)
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");
        if (KIND == 15)
            asm volatile(REP8("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_gt_u32 vcc, %2, %3\n v_cndmask_b32 %3, %3, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (KIND == 16)
            asm volatile(REP8("v_cmp_gt_u32 s[20:21], %0, %1\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cmp_gt_u32 s[22:23], %2, %3\n v_cndmask_b32_e64 %3, %3, %0, s[22:23]\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21", "s22", "s23");
        if (KIND == 17)
            asm volatile("v_cmp_gt_u32 vcc, %0, %1\n" REP8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (KIND == 18)
            asm volatile(REP8("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (KIND == 19)
            asm volatile(REP8("v_bfe_u32 %0, %1, 8, 8\n v_and_or_b32 %1, %2, %3, %0\n v_lshl_or_b32 %2, %3, 8, %1\n v_or3_b32 %3, %0, %1, %2\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 20)
            asm volatile(REP8("v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %1, %2\n v_cvt_f32_ubyte2 %2, %3\n v_cvt_f32_ubyte3 %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 21)
            asm volatile(REP8("v_sad_u32 %0, %0, %1, %2\n v_sad_u32 %1, %1, %2, %3\n v_sad_u32 %2, %2, %3, %0\n v_sad_u32 %3, %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 22)
            asm volatile(REP8("v_mbcnt_lo_u32_b32 %0, s20, %1\n v_mbcnt_hi_u32_b32 %1, s21, %0\n v_mbcnt_lo_u32_b32 %2, s20, %3\n v_mbcnt_hi_u32_b32 %3, s21, %2\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");
        if (KIND == 23)
            asm volatile(REP8("v_and_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_lshlrev_b32 %2, 3, %3\n v_xor_b32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 24)
            asm volatile(REP8("v_cmp_ge_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_eq_f32 s[24:25], %2, %3\n v_cmp_gt_f32 vcc, %3, %0\n")
                         : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");
        if (KIND == 25)
            asm volatile(REP8("v_add_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_add_f32 %3, %3, %0\n")
                         : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));
        if (KIND == 26)
            asm volatile(REP8("v_cndmask_b32 %0, 0, %1, vcc\n v_cndmask_b32 %1, 0, %2, vcc\n v_cndmask_b32 %2, 0, %3, vcc\n v_cndmask_b32 %3, 0, %0, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (KIND == 27)
            asm volatile(REP8("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %1, s20, 5\n v_readfirstlane_b32 s21, %2\n v_mov_b32 %3, s21\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");
        if (KIND == 28)
            asm volatile(REP8("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 29)
            asm volatile(REP8("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 30)
            asm volatile(REP8("v_mul_u32_u24 %0, %0, %1\n v_mad_u32_u24 %1, %1, %2, %3\n v_mul_u32_u24 %2, %2, %3\n v_mad_u32_u24 %3, %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 31)
            asm volatile(REP8("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n")
                         : "+v"(*(uint64_t *)&fa), "+v"(*(uint64_t *)&fc), "+v"(c), "+v"(d) : : "vcc");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint32_t r = a + b + c + d + (uint32_t)fa + (uint32_t)fb + (uint32_t)fc + (uint32_t)fd + s0 + s1 + s2 + s3;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r;
    }
}

typedef void (*Fn)(uint64_t *, int);
static const char *names[] = {"v_add_u32", "v_fma_f32", "v_cndmask", "v_cmp->sgpr", "v_rcp_f32", "s_add_u32 only",
                              "32 valu + 32 salu", "32 valu + 16 salu", "cvt_ubyte/sad/bfe/mbcnt", "v_pk f32",
                              "v_add dpp", "s mask logic only", "v_cvt u32<->f32", "cmp->s_and->cndmask->add chain", "v_cndmask e64 sgpr mask", "v_cmp vcc + v_cndmask vcc pairs", "v_cmp sgpr + v_cndmask e64 pairs", "1 v_cmp vcc + 32 v_cndmask vcc", "v_addc_co vcc chain", "bfe/and_or/lshl_or/or3", "v_cvt_f32_ubyteN", "v_sad_u32", "v_mbcnt lo/hi", "and/or/shl/xor", "v_cmp f32 -> sgpr", "add/mul/sub f32", "v_cndmask 0,v,vcc", "readlane/writelane/readfirstlane/mov", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24 / v_mad_u32_u24", "v_mad_u64_u32"};
static const int instr_per_iter[] = {32, 32, 32, 32, 32, 32, 64, 48, 32, 32, 32, 32, 32, 32, 32, 32, 32, 33, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32};

int main() {
    Fn fns[] = {ub<0>, ub<1>, ub<2>, ub<3>, ub<4>, ub<5>, ub<6>, ub<7>, ub<8>, ub<9>, ub<10>, ub<11>, ub<12>, ub<13>, ub<14>, ub<15>, ub<16>, ub<17>, ub<18>, ub<19>, ub<20>, ub<21>, ub<22>, ub<23>, ub<24>, ub<25>, ub<26>, ub<27>, ub<28>, ub<29>, ub<30>, ub<31>};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    uint64_t *d;
    hipMalloc(&d, sizeof(uint64_t) * 2 * cus * 8);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 32; ++k) {
        for (int w : {4, 8}) {
            const int grid = cus * w;  // 256-thread blocks: one wave per SIMD each
            hipLaunchKernelGGL(fns[k], dim3(grid), dim3(256), 0, 0, d, 64);  // warm
            hipEventRecord(e0);
            hipLaunchKernelGGL(fns[k], dim3(grid), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> h(2 * grid);
            hipMemcpy(h.data(), d, sizeof(uint64_t) * 2 * grid, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < grid; ++i) avg += (double)h[2 * i];
            avg /= grid;
            // per SIMD: w waves x iters x instr_per_iter wave-instructions in `avg` memtime ticks
            const double per = avg / ((double)iters * instr_per_iter[k] * w);
            printf("%-34s waves/SIMD %d : %8.3f ms, memtime ticks/wave %10.0f, ticks per wave-instr per SIMD %.3f, ns per wave-instr per SIMD %.4f\n",
                   names[k], w, ms, avg, per, ms * 1e6 / ((double)iters * instr_per_iter[k] * w));
        }
    }
    return 0;
}
