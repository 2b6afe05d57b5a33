// Does a tiny kernel between two big ones cost more when it WRITES PINNED HOST MEMORY?  (the solver's step kernel reports through a
// pinned ring: rocprofv3 shows ~6 us in front of it and ~6 us behind it, while the sweep's own four launches run back to back)
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/host_arg_gap scripts/probe/host_arg_gap.hip && /tmp/host_arg_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void big_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] * 3u + 1u;
}
// mode 0: device memory only; 1: one store to the pinned word, no fence; 2: store + system-scope fence + release store (the step kernel's report)
__global__ void __launch_bounds__(1024) tiny_kernel(uint32_t* dev_word, uint32_t* host_word, uint32_t* host_seq, int mode, const uint4* part) {
    __shared__ uint32_t sh[16];
    uint32_t acc = 0;
#pragma unroll 8
    for (uint32_t b = threadIdx.x; b < 8192u; b += 1024u) { const uint4 v = part[b]; acc += v.x + v.z; }   // the step kernel sums 8192 energy pairs
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x) return;
    dev_word[1] = sh[0] + sh[5];
    const uint32_t v = dev_word[0] + 1u; dev_word[0] = v;
    if (mode >= 1) host_word[0] = v;
    if (mode >= 2) { __threadfence_system(); __hip_atomic_store(host_seq, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main() {
    const size_t n = 32u << 20;   // 128 MB in, 128 MB out per full-size launch
    uint32_t *a, *b, *dw, *hw, *dhw;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&dw, 256)); CK(hipMemset(a, 1, n * 4)); CK(hipMemset(dw, 0, 256));
    CK(hipHostMalloc((void**)&hw, 256, hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&dhw, hw, 0));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 40;
    // cases: no tiny kernel at all; tiny with device args only (host pointers passed as device pointers); tiny that is HANDED host pointers but does not touch them;
    // tiny that stores to host memory; tiny that stores + fences
    struct Case { const char* name; int tiny; int host_args; int mode; } cases[] = {
        {"4 big launches, no tiny kernel", 0, 0, 0}, {"+ tiny kernel, device pointers only", 1, 0, 0}, {"+ tiny kernel, pinned pointers among its arguments (unused)", 1, 1, 0},
        {"+ tiny kernel storing to pinned memory", 1, 1, 1}, {"+ tiny kernel: store, system fence, release store (the step kernel)", 1, 1, 2}};
    for (int graph = 0; graph < 2; ++graph)
        for (const Case& c : cases) {
            auto body = [&]() {
                for (int r = 0; r < 4; ++r)   // one "sweep": four dependent big launches
                    for (int p = 0; p < 4; ++p) hipLaunchKernelGGL(big_kernel, dim3(2048), dim3(256), 0, s, p & 1 ? b : a, p & 1 ? a : b, n);
                };
            (void)body;
            auto sweep = [&]() {
                for (int p = 0; p < 4; ++p) hipLaunchKernelGGL(big_kernel, dim3(2048), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, n >> (p == 3 ? 3 : p == 2 ? 1 : 0));   // like the four colour phases: two large, one medium, one small
                if (c.tiny) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(1024), 0, s, dw, c.host_args ? dhw : dw + 8, c.host_args ? dhw + 8 : dw + 16, c.mode, (const uint4*)a);
            };
            hipGraphExec_t ge = nullptr;
            if (graph) {
                hipGraph_t g; CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < 4; ++k) sweep();
                CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
            }
            std::vector<float> ms;
            for (int t = 0; t < 5; ++t) {
                CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
                for (int r = 0; r < REP / 4; ++r) { if (graph) CK(hipGraphLaunch(ge, s)); else for (int k = 0; k < 4; ++k) sweep(); }
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m * 1e3f / REP);
            }
            std::sort(ms.begin(), ms.end());
            printf("%-6s %-78s %8.1f us per sweep\n", graph ? "graph" : "direct", c.name, ms[2]);
            if (ge) CK(hipGraphExecDestroy(ge));
        }
    return 0;
}
