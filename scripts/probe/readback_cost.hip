// What does a 4-byte read-back cost between two kernels?  (a) hipMemcpyAsync into pageable memory + hipStreamSynchronize (read_u32 of k_dc.hip),
// (b) the same into pinned memory, (c) a one-thread kernel that stores the word to pinned memory + release, the host spinning on a sequence
// number (the solver's report ring).  Each measured as the time from the end of kernel A to the start of a dependent kernel B launched by the host
// after it has the value: wall clock of (A, read-back, B) minus the same without the read-back.
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/readback_cost scripts/probe/readback_cost.hip && /tmp/readback_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work_kernel(uint32_t* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { p[i] = p[i] * 3u + 1u; acc += p[i]; }
    if (acc == 0xDEADBEEFu) out[1] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (uint32_t)n;
}
__global__ void report_kernel(const uint32_t* src, uint32_t* dst, uint32_t* seq, uint32_t s) { *dst = *src; __threadfence_system(); __hip_atomic_store(seq, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
int main() {
    const size_t n = 8u << 20;   // ~30 us kernels
    uint32_t *a, *d, *hp, *dhp; uint32_t pageable[4];
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&d, 64)); CK(hipMemset(a, 1, n * 4));
    CK(hipHostMalloc((void**)&hp, 64, hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&dhp, hp, 0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t seq = 0;
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<double> us;
        for (int t = 0; t < 30; ++t) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 10; ++r) {
                hipLaunchKernelGGL(work_kernel, dim3(2048), dim3(256), 0, s, a, n, d);
                uint32_t v = 0;
                if (mode == 1) { CK(hipMemcpyAsync(pageable, d, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); v = pageable[0]; }
                if (mode == 2) { CK(hipMemcpyAsync(hp + 8, d, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); v = hp[8]; }
                if (mode == 3) { ++seq; hipLaunchKernelGGL(report_kernel, dim3(1), dim3(1), 0, s, d, dhp, dhp + 1, seq);
                                 while (__atomic_load_n((volatile uint32_t*)(hp + 1), __ATOMIC_ACQUIRE) != seq) { } v = hp[0]; }
                hipLaunchKernelGGL(work_kernel, dim3(2048), dim3(256), 0, s, a, v ? n : n, d);
            }
            CK(hipStreamSynchronize(s));
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10.0);
        }
        std::sort(us.begin(), us.end());
        const char* name[] = {"A, B back to back (no read-back)", "hipMemcpyAsync to pageable + hipStreamSynchronize", "hipMemcpyAsync to pinned + hipStreamSynchronize", "report kernel + host spin on a pinned sequence number"};
        printf("%-58s %8.1f us per (A, read-back, B)\n", name[mode], us[us.size() / 2]);
    }
    return 0;
}
