#include <hip/hip_runtime.h>
// pattern of the proposed sweep: stage node it+1's words with LDS-direct loads while the tile of node `it` is read and written in LDS
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ g, const float* __restrict__ a, float* out, int n_iter) {
    __shared__ uint4 s_stage[2][256];
    __shared__ float s_tile[256 + 8];
    const int t = threadIdx.x;
    float acc = 0.f;
    auto issue = [&](int it, int buf) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)it * 256 + t), (__attribute__((address_space(3))) void*)(&s_stage[buf][t & ~63]), 16, 0, 0);
    };
    issue(0, 0);
    for (int it = 0; it < n_iter; ++it) {
        issue(it + 1, (it + 1) & 1);                 // next node's words on their way
        const uint4 w = s_stage[it & 1][t];          // this node's words (loaded one iteration ago)
        s_tile[t] = a[it * 256 + t] + __uint_as_float(w.x);   // the per-node LDS tile: write ...
        acc += s_tile[(t * 7 + w.y) & 255];          // ... and gather
    }
    out[blockIdx.x * 256 + t] = acc;
}
