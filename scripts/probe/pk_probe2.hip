// probe 2: v_pk_mul_f32 / v_pk_add_f32 issue rate, op_sel broadcast forms, and dependent-chain latency of packed vs plain fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    const float t = (float)threadIdx.x;
    f2 a0 = {t, t + 1}, a1 = {t + 2, t + 3}, a2 = {t + 4, t + 5}, a3 = {t + 6, t + 7}, a4 = {t + 8, t + 9}, a5 = {t + 10, t + 11};
    f2 m = {1.0f + 1e-7f * t, 1.0f - 1e-7f * t};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 6 independent v_pk_mul / v_pk_add
            asm volatile("v_pk_mul_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n v_pk_mul_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %6\n v_pk_mul_f32 %4, %4, %6\n v_pk_add_f32 %5, %5, %6\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m));
        } else if (MODE == 1) {   // 6 independent pk_fma with op_sel broadcast + neg modifiers
            asm volatile("v_pk_fma_f32 %0, %6, %0, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %6, %1, %1 op_sel:[1,0,0]\n v_pk_fma_f32 %2, %6, %2, %2 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                         "v_pk_fma_f32 %3, %6, %3, %3 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %4, %6, %4, %4 op_sel:[1,0,0]\n v_pk_fma_f32 %5, %6, %5, %5 op_sel_hi:[0,1,1]\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m));
        } else if (MODE == 2) {   // dependent chain: 6 pk_fma on one accumulator
            asm volatile("v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %0, %1, %0, %0\n v_pk_fma_f32 %0, %1, %0, %0\n"
                         : "+v"(a0) : "v"(m));
        } else if (MODE == 3) {   // dependent chain: 6 v_fma on one accumulator
            asm volatile("v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n"
                         : "+v"(a0.x) : "v"(m.x));
        } else {   // 6 independent plain v_mul / v_add
            asm volatile("v_mul_f32 %0, %0, %6\n v_add_f32 %1, %1, %6\n v_mul_f32 %2, %2, %6\n v_add_f32 %3, %3, %6\n v_mul_f32 %4, %4, %6\n v_add_f32 %5, %5, %6\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x) : "v"(m.x));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y + a4.x + a4.y + a5.x + a5.y;
}
template <int MODE> float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const int iters = 20000;
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int wps = 8; wps >= 1; wps /= 2) {   // waves per SIMD: blocks of 4 waves, 256 CUs
        const int blocks = 256 * wps;
        const double winst = (double)blocks * 4 * iters * 6;
        auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 * 1024.0 / winst; };
        const float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters), t3 = run<3>(d, blocks, iters), t4 = run<4>(d, blocks, iters);
        printf("%d waves/SIMD  cyc/inst/SIMD (at 2.4 GHz): pk_mul/add indep %.2f | pk_fma op_sel indep %.2f | pk_fma dependent %.2f | v_fma dependent %.2f | v_mul/add indep %.2f\n",
               wps, cyc(t0), cyc(t1), cyc(t2), cyc(t3), cyc(t4));
    }
    return 0;
}
