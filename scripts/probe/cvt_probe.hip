// probe: rounding / saturation of v_cvt_pk_u8_f32 on gfx950 (used by the sweep kernel's message packing)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* x, unsigned* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1, 0xAABBCCDDu); }
int main() {
    const int n = 260 * 64 + 8;
    float* hx = new float[n]; unsigned* hy = new unsigned[n];
    for (int i = 0; i < 260 * 64; ++i) hx[i] = -2.0f + (float)i / 64.0f;
    hx[n - 8] = 1e9f; hx[n - 7] = -1e9f; hx[n - 6] = NAN; hx[n - 5] = INFINITY; hx[n - 4] = 254.5f; hx[n - 3] = 255.5f; hx[n - 2] = 0.49999997f; hx[n - 1] = 255.49998f;
    float* dx; unsigned* dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n); hipMemcpy(hy, dy, n * 4, hipMemcpyDeviceToHost);
    int bad_rne = 0, bad_trunc = 0, bad_pack = 0;
    for (int i = 0; i < n; ++i) {
        const float x = hx[i]; const unsigned got = (hy[i] >> 8) & 0xFF;
        if ((hy[i] & 0xFFFF00FFu) != 0xAABB00DDu) ++bad_pack;
        if (x != x) { printf("nan -> %u\n", got); continue; }
        const float c = fminf(fmaxf(x, 0.0f), 255.0f);
        const unsigned rne = (unsigned)lrintf(c), tr = (unsigned)c;
        if (got != rne) { if (bad_rne < 6) printf("x=%.6f got %u rne %u trunc %u\n", x, got, rne, tr); ++bad_rne; }
        if (got != tr) ++bad_trunc;
    }
    printf("n=%d mismatches vs saturating RNE: %d, vs saturating truncation: %d, pack errors: %d\n", n, bad_rne, bad_trunc, bad_pack);
    return 0;
}
