// scan.hip -- device-wide exclusive prefix sum of u32 (reduce / scan / downsweep).
// Integer sums: result independent of the launch geometry.  Used for the
// pass-rank table, the CSR column pointers (calculate_data_costs.cpp:291-298
// writes columns in face order) and the MRF message offsets.
#include "ctx.h"

namespace mvs {

namespace {
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns the block total in *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        const uint32_t s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ block_sums, size_t n) {
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) s += in[base + k];
    uint32_t tot;
    (void)block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_down_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ block_prefix,
                                 size_t n, uint32_t* __restrict__ d_total) {
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < n) ? in[base + k] : 0u; s += v[k]; }
    uint32_t tot;
    uint32_t run = block_excl_scan(s, &tot) + (block_prefix ? block_prefix[blockIdx.x] : 0u);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (d_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) *d_total = run;
}
}  // namespace

static void scan_rec(mvs_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t* d_total, uint32_t* tmp, size_t tmp_cap) {
    const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb <= 1) {
        hipLaunchKernelGGL(scan_down_kernel, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, in, out, (const uint32_t*)nullptr, n, d_total);
        MVS_LAUNCH_CHECK();
        return;
    }
    if (nb + 1 > tmp_cap) throw HipError("scan: temp buffer too small");
    uint32_t* sums = tmp;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, sums, n);
    MVS_LAUNCH_CHECK();
    scan_rec(ctx, sums, sums, nb, nullptr, tmp + nb, tmp_cap - nb);  // in-place scan of the block sums
    hipLaunchKernelGGL(scan_down_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, out, (const uint32_t*)sums, n, d_total);
    MVS_LAUNCH_CHECK();
}

namespace {
__global__ void __launch_bounds__(256) sum64_kernel(const uint32_t* __restrict__ in, size_t n, unsigned long long* __restrict__ out) {
    unsigned long long s = 0;
    for (size_t k = (size_t)blockIdx.x * 256u + threadIdx.x; k < n; k += (size_t)gridDim.x * 256u) s += in[k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ unsigned long long ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
}  // namespace

// 64-bit total of a u32 array (blocking).  The scans above are 32 bits wide: callers whose totals can pass 2^32
// (pairs of a huge scene, message elements) take this exact total first and refuse instead of wrapping.
uint64_t sum_u32(mvs_ctx* ctx, const uint32_t* in, size_t n) {
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
    ctx->scan_tmp.ensure(2 * 1024 + 8);
    unsigned long long* part = reinterpret_cast<unsigned long long*>(ctx->scan_tmp.p);
    hipLaunchKernelGGL(sum64_kernel, dim3(blocks), dim3(256), 0, ctx->stream, in, n, part);
    MVS_LAUNCH_CHECK();
    std::vector<unsigned long long> h(blocks);
    MVS_HIP(hipMemcpyAsync(h.data(), part, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t t = 0; for (auto v : h) t += v;
    return t;
}

void exclusive_scan_u32(mvs_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t* d_total) {
    if (n == 0) {
        if (d_total) MVS_HIP(hipMemsetAsync(d_total, 0, sizeof(uint32_t), ctx->stream));
        return;
    }
    size_t need = 0;
    for (size_t m = (n + SCAN_TILE - 1) / SCAN_TILE; m > 1; m = (m + SCAN_TILE - 1) / SCAN_TILE) need += m + 1;
    need += 8;
    ctx->scan_tmp.ensure(need);
    scan_rec(ctx, in, out, n, d_total, ctx->scan_tmp.p, ctx->scan_tmp.cap);
}

}  // namespace mvs
