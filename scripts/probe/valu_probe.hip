// probe 3: issue cost of VALU instruction forms on gfx950 (8 waves / SIMD, 6 independent accumulators per wave):
// which operand kinds / encodings make an fp32 instruction slower than the plain VGPR-only VOP2 form
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP6(a) a(0) a(1) a(2) a(3) a(4) a(5)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s1, float s2, int iters) {
    const float t = (float)threadIdx.x;
    float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5;
    float m = 1.0f + 1e-7f * t;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)      asm volatile("v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3\n v_mul_f32 %4, %6, %4\n v_mul_f32 %5, %6, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m));
        else if (MODE == 1) asm volatile("v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3\n v_mul_f32 %4, %6, %4\n v_mul_f32 %5, %6, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "s"(s1));
        else if (MODE == 2) asm volatile("v_fma_f32 %0, %6, %0, %0\n v_fma_f32 %1, %6, %1, %1\n v_fma_f32 %2, %6, %2, %2\n v_fma_f32 %3, %6, %3, %3\n v_fma_f32 %4, %6, %4, %4\n v_fma_f32 %5, %6, %5, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m));
        else if (MODE == 3) asm volatile("v_fma_f32 %0, %6, %0, %0\n v_fma_f32 %1, %6, %1, %1\n v_fma_f32 %2, %6, %2, %2\n v_fma_f32 %3, %6, %3, %3\n v_fma_f32 %4, %6, %4, %4\n v_fma_f32 %5, %6, %5, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "s"(s1));
        else if (MODE == 4) asm volatile("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n v_fmac_f32 %4, %6, %7\n v_fmac_f32 %5, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m), "v"(t));
        else if (MODE == 5) asm volatile("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n v_fmac_f32 %4, %6, %7\n v_fmac_f32 %5, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "s"(s1), "v"(t));
        else if (MODE == 6) asm volatile("v_min_f32 %0, %6, %0\n v_min_f32 %1, %6, %1\n v_min_f32 %2, %6, %2\n v_min_f32 %3, %6, %3\n v_min_f32 %4, %6, %4\n v_min_f32 %5, %6, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "s"(s1));
        else if (MODE == 7) asm volatile("v_min3_f32 %0, %6, %0, %1\n v_min3_f32 %1, %6, %1, %2\n v_min3_f32 %2, %6, %2, %3\n v_min3_f32 %3, %6, %3, %4\n v_min3_f32 %4, %6, %4, %5\n v_min3_f32 %5, %6, %5, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m));
        else if (MODE == 8) asm volatile("v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
        else if (MODE == 9) asm volatile("v_cvt_pk_u8_f32 %0, %6, 1, %0\n v_cvt_pk_u8_f32 %1, %6, 1, %1\n v_cvt_pk_u8_f32 %2, %6, 1, %2\n v_cvt_pk_u8_f32 %3, %6, 1, %3\n v_cvt_pk_u8_f32 %4, %6, 1, %4\n v_cvt_pk_u8_f32 %5, %6, 1, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5) : "v"(m));
        else if (MODE == 10) asm volatile("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n v_bfe_u32 %4, %4, 8, 8\n v_bfe_u32 %5, %5, 8, 8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5));
        else if (MODE == 11) asm volatile("v_lshl_add_u32 %0, %0, 2, %6\n v_lshl_add_u32 %1, %1, 2, %6\n v_lshl_add_u32 %2, %2, 2, %6\n v_lshl_add_u32 %3, %3, 2, %6\n v_lshl_add_u32 %4, %4, 2, %6\n v_lshl_add_u32 %5, %5, 2, %6" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5) : "v"(threadIdx.x));
        else if (MODE == 12) asm volatile("v_cvt_f32_ubyte1 %0, %6\n v_cvt_f32_ubyte2 %1, %6\n v_cvt_f32_ubyte3 %2, %6\n v_cvt_f32_ubyte0 %3, %6\n v_cvt_f32_ubyte1 %4, %6\n v_cvt_f32_ubyte2 %5, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(u0));
        else if (MODE == 13) asm volatile("v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %6, vcc\n v_cndmask_b32 %4, %4, %6, vcc\n v_cndmask_b32 %5, %5, %6, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m) : "vcc");
        else if (MODE == 14) asm volatile("v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n v_add_u32 %4, %4, %6\n v_add_u32 %5, %5, %6" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5) : "v"(threadIdx.x));
        else if (MODE == 15) asm volatile("v_mul_f32 %0, 0x3f8ccccd, %0\n v_mul_f32 %1, 0x3f8ccccd, %1\n v_mul_f32 %2, 0x3f8ccccd, %2\n v_mul_f32 %3, 0x3f8ccccd, %3\n v_mul_f32 %4, 0x3f8ccccd, %4\n v_mul_f32 %5, 0x3f8ccccd, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
        else if (MODE == 16) asm volatile("v_fma_f32 %0, %6, %0, %7\n v_fma_f32 %1, %6, %1, %7\n v_fma_f32 %2, %6, %2, %7\n v_fma_f32 %3, %6, %3, %7\n v_fma_f32 %4, %6, %4, %7\n v_fma_f32 %5, %6, %5, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m), "v"(t));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + (float)(u0 + u1 + u2 + u3 + u4 + u5);
}
template <int MODE> float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f, 16);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const int iters = 20000, blocks = 256 * 8;
    float* d; (void)hipMalloc(&d, blocks * 256 * 4);
    const double winst = (double)blocks * 4 * iters * 6;
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 * 1024.0 / winst; };
    const char* names[] = {"v_mul_f32 vgpr", "v_mul_f32 sgpr src0", "v_fma_f32 vgpr (acc twice)", "v_fma_f32 sgpr", "v_fmac_f32 vgpr", "v_fmac_f32 sgpr src0", "v_min_f32 sgpr src0", "v_min3_f32 vgpr",
                           "v_min_f32_dpp", "v_cvt_pk_u8_f32", "v_bfe_u32", "v_lshl_add_u32", "v_cvt_f32_ubyteN", "v_cndmask_b32 vcc", "v_add_u32", "v_mul_f32 literal", "v_fma_f32 3 distinct vgprs"};
    float t[17];
    t[0] = run<0>(d, blocks, iters); t[1] = run<1>(d, blocks, iters); t[2] = run<2>(d, blocks, iters); t[3] = run<3>(d, blocks, iters); t[4] = run<4>(d, blocks, iters);
    t[5] = run<5>(d, blocks, iters); t[6] = run<6>(d, blocks, iters); t[7] = run<7>(d, blocks, iters); t[8] = run<8>(d, blocks, iters); t[9] = run<9>(d, blocks, iters);
    t[10] = run<10>(d, blocks, iters); t[11] = run<11>(d, blocks, iters); t[12] = run<12>(d, blocks, iters); t[13] = run<13>(d, blocks, iters); t[14] = run<14>(d, blocks, iters);
    t[15] = run<15>(d, blocks, iters); t[16] = run<16>(d, blocks, iters);
    for (int i = 0; i < 17; ++i) printf("%-28s %.2f cyc/inst/SIMD (2.4 GHz)\n", names[i], cyc(t[i]));
    return 0;
}
