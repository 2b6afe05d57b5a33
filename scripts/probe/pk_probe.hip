// probe: issue rate of v_pk_fma_f32 (SGPR-pair operand) against v_fma_f32 on gfx950, for the BVH slab test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, f2 sa, f2 sb, f2 sc, int iters) {
    const float t = (float)threadIdx.x;
    f2 a0 = {t, t + 1}, a1 = {t + 2, t + 3}, a2 = {t + 4, t + 5}, a3 = {t + 6, t + 7}, a4 = {t + 8, t + 9}, a5 = {t + 10, t + 11};
    f2 m = {1.0f + 1e-7f * t, 1.0f - 1e-7f * t};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 12 v_fma_f32, scalar operand
            asm volatile("v_fma_f32 %0, %6, %8, %0\n v_fma_f32 %1, %7, %9, %1\n v_fma_f32 %2, %6, %8, %2\n v_fma_f32 %3, %7, %9, %3\n v_fma_f32 %4, %6, %8, %4\n v_fma_f32 %5, %7, %9, %5\n"
                         : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y) : "s"(sa.x), "s"(sa.y), "v"(m.x), "v"(m.y));
            asm volatile("v_fma_f32 %0, %6, %8, %0\n v_fma_f32 %1, %7, %9, %1\n v_fma_f32 %2, %6, %8, %2\n v_fma_f32 %3, %7, %9, %3\n v_fma_f32 %4, %6, %8, %4\n v_fma_f32 %5, %7, %9, %5\n"
                         : "+v"(a3.x), "+v"(a3.y), "+v"(a4.x), "+v"(a4.y), "+v"(a5.x), "+v"(a5.y) : "s"(sb.x), "s"(sb.y), "v"(m.x), "v"(m.y));
        } else if (MODE == 1) {   // 6 v_pk_fma_f32, scalar pair operand
            asm volatile("v_pk_fma_f32 %0, %6, %9, %0\n v_pk_fma_f32 %1, %7, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %6, %9, %3\n v_pk_fma_f32 %4, %7, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "s"(sa), "s"(sb), "s"(sc), "v"(m));
        } else if (MODE == 2) {   // 6 v_pk_fma_f32, all vector operands
            asm volatile("v_pk_fma_f32 %0, %6, %7, %0\n v_pk_fma_f32 %1, %6, %7, %1\n v_pk_fma_f32 %2, %6, %7, %2\n v_pk_fma_f32 %3, %6, %7, %3\n v_pk_fma_f32 %4, %6, %7, %4\n v_pk_fma_f32 %5, %6, %7, %5\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m), "v"(m));
        } else {   // 12 min/max3 mix
            asm volatile("v_max3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_max_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n v_max3_f32 %4, %4, %5, %0\n v_min3_f32 %5, %5, %0, %1\n"
                         : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y));
            asm volatile("v_max3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_max_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n v_max3_f32 %4, %4, %5, %0\n v_min3_f32 %5, %5, %0, %1\n"
                         : "+v"(a3.x), "+v"(a3.y), "+v"(a4.x), "+v"(a4.y), "+v"(a5.x), "+v"(a5.y));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y + a4.x + a4.y + a5.x + a5.y;
}
template <int MODE> float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f2 sa = {1.0f, 0.999f}, sb = {1.001f, 0.998f}, sc = {0.5f, 0.25f};
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sa, sb, sc, 16);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sa, sb, sc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const int blocks = 256 * 8, iters = 20000;
    float* d; hipMalloc(&d, blocks * 256 * 4);
    const double winst = (double)blocks * 4 * iters;   // wave-iterations
    const float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters), t3 = run<3>(d, blocks, iters);
    // cycles per wave-instruction per SIMD at 2.4 GHz: time * 2.4e9 * (256 CUs * 4 SIMDs) / instructions
    auto cyc = [&](float ms, int per_iter) { return ms * 1e-3 * 2.4e9 * 1024.0 / (winst * per_iter); };
    printf("12 v_fma_f32 (sgpr)     : %.3f ms  %.2f cyc/inst/SIMD\n", t0, cyc(t0, 12));
    printf(" 6 v_pk_fma_f32 (sgpr)  : %.3f ms  %.2f cyc/inst/SIMD\n", t1, cyc(t1, 6));
    printf(" 6 v_pk_fma_f32 (vgpr)  : %.3f ms  %.2f cyc/inst/SIMD\n", t2, cyc(t2, 6));
    printf("12 min/max/min3/max3    : %.3f ms  %.2f cyc/inst/SIMD\n", t3, cyc(t3, 12));
    return 0;
}
