// k_region.hip -- region moves: a two-level step after the ICM polish (option mvs_mrf_params.region_rounds, off by default;
// in the spirit of mapMAP's multilevel contraction of same-label regions, view_selection.cpp:103-115 use_multilevel; the
// algorithm is DEFINED in oracle/oracle.cpp mrf_region_round and restated here operation for operation on integers).
//
// A REGION = connected component of equally labelled faces over the model's edges, named by its smallest face.  It may take
// the label l of a neighbouring region if every one of its faces has l among its candidates; the energy changes by
//   sum_i (D_i(l) - D_i(l_i))  -  #edges to neighbours labelled l         (32.32 fixed point: exact, order independent).
// Per region the best candidate (largest gain, ties to the smaller label); a region moves iff its gain is positive and beats
// the gains of all neighbouring regions (ties to the smaller region id): an independent set, the energy drops by the sum.
//
// Kernels: lock-free union-find (smaller id wins) -> cut-edge flags -> scan -> (region, label) keys -> radix sort -> unique
// with counts -> per-face accumulation into the candidates of its region (aggregated per wave before any atomic; integer
// atomics only) -> best candidate per region (atomicMax on the gain, atomicMin on the label among the maxima) -> lose flags
// over the cut edges -> apply.
#include "ctx.h"
#include <rocprim/rocprim.hpp>

namespace mvs {
void resolve_best(mvs_ctx* ctx);

namespace {

__device__ inline uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t uf_find(uint32_t* parent, uint32_t x) {
    uint32_t curr = ld_agent(parent + x);
    if (curr != x) {
        uint32_t prev = x, next;
        while (curr > (next = ld_agent(parent + curr))) { st_agent(parent + prev, next); prev = curr; curr = next; }
    }
    return curr;
}
__device__ __forceinline__ unsigned long long fix32(float d) { return (unsigned long long)((double)d * 4294967296.0); }

__global__ void rg_init_kernel(uint32_t* __restrict__ parent, unsigned long long* __restrict__ gain, unsigned long long* __restrict__ cur,
                               uint32_t* __restrict__ size, uint32_t* __restrict__ bestl, uint32_t* __restrict__ lose, uint32_t* __restrict__ cfirst, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { parent[i] = i; gain[i] = 0ull; cur[i] = 0ull; size[i] = 0u; bestl[i] = 0xFFFFFFFFu; lose[i] = 0u; cfirst[i] = 0xFFFFFFFFu; }
}
// link i with every equally labelled neighbour j < i over a model edge (both labels non-zero)
__global__ void rg_hook_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const uint32_t* __restrict__ lab, uint32_t* __restrict__ parent, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t L = lab[i];
    if (L == 0u) return;
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        const uint32_t j = adj[e];
        if (j >= i || lab[j] != L) continue;
        uint32_t u = uf_find(parent, i), v = uf_find(parent, j);
        while (u != v) {
            if (u < v) { const uint32_t t = u; u = v; v = t; }
            const uint32_t old = atomicCAS(parent + u, u, v);
            if (old == u) break;
            u = old;
        }
    }
}
__global__ void rg_flatten_kernel(uint32_t* __restrict__ parent, uint32_t* __restrict__ root, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = i, p;
    while ((p = ld_agent(parent + r)) != r) r = p;
    root[i] = r;
}
// flag[e] = 1 for directed model edges (i <- j) whose ends carry different labels; flag[E] = 0
__global__ void rg_cut_flag_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const uint32_t* __restrict__ lab, uint32_t n, uint32_t E, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) flag[E] = 0u;
    if (i >= n) return;
    const uint32_t L = lab[i];
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) { const uint32_t lj = lab[adj[e]]; flag[e] = (L != 0u && lj != 0u && lj != L) ? 1u : 0u; }
}
__global__ void rg_cut_emit_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const uint32_t* __restrict__ lab, const uint32_t* __restrict__ root,
                                   const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint32_t n,
                                   unsigned long long* __restrict__ key, uint2* __restrict__ cut) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        if (!flag[e]) continue;
        const uint32_t j = adj[e];
        key[pos[e]] = ((unsigned long long)root[i] << 16) | lab[j];
        cut[pos[e]] = make_uint2(i, j);
    }
}
__global__ void rg_first_kernel(const unsigned long long* __restrict__ key, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= n) flag[k] = (k < n && (k == 0 || key[k] != key[k - 1])) ? 1u : 0u;
}
// unique candidates: ck[c] = key, cstart[c] = first position of its run (cstart[nC] = n): count = cstart[c + 1] - cstart[c]
__global__ void rg_unique_kernel(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint32_t n, uint32_t nC,
                                 unsigned long long* __restrict__ ck, uint32_t* __restrict__ cstart, uint32_t* __restrict__ cfirst) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && flag[k]) {
        ck[pos[k]] = key[k]; cstart[pos[k]] = k;
        // first candidate of its region (keys are sorted by region, then label): where the region's faces start reading
        if (k == 0 || (key[k - 1] >> 16) != (key[k] >> 16)) cfirst[(uint32_t)(key[k] >> 16)] = pos[k];
    }
    if (k == 0) cstart[nC] = n;
}
// per face: its region's size and current unary sum, and for every candidate label of its region whether the face has it and
// at what cost -- aggregated over the lanes of a wave that share the region before anything touches memory
__global__ void __launch_bounds__(256) rg_accumulate_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                            const uint32_t* __restrict__ sel, const uint32_t* __restrict__ root, uint32_t n,
                                                            const unsigned long long* __restrict__ ck, const uint32_t* __restrict__ cfirst, uint32_t nC,
                                                            uint32_t* __restrict__ size, unsigned long long* __restrict__ cur,
                                                            uint32_t* __restrict__ have, unsigned long long* __restrict__ sum) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < n;
    const uint32_t R = act ? root[i] : 0xFFFFFFFFu;
    const uint32_t p0 = act ? col_ptr[i] : 0u, K = act ? col_ptr[i + 1] - p0 : 0u;
    const unsigned long long mine = act ? (K ? fix32(cost[p0 + sel[i]]) : fix32(1.0f)) : 0ull;
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(act);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t R0 = (uint32_t)__shfl((int)R, leader, 64);
        const bool in = act && R == R0;
        const unsigned long long grp = __ballot(in);
        todo &= ~grp;
        // size and current unary sum of the region
        unsigned long long v = in ? mine : 0ull;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == leader) { atomicAdd(&size[R0], (uint32_t)__popcll(grp)); atomicAdd(&cur[R0], v); }
        // candidates of the region: the run of keys with region field R0, from cfirst[R0]; 64 keys per round trip (lane k
        // holds key k of the chunk, broadcast with a shuffle), wave-uniform loop
        const uint32_t lo = cfirst[R0];
        if (lo == 0xFFFFFFFFu) continue;
        for (uint32_t base = lo; base < nC; base += 64) {
            const unsigned long long mykey = (base + lane < nC) ? ck[base + lane] : ~0ull;
            bool more = true;
            for (int k = 0; k < 64; ++k) {
                const unsigned long long key = __shfl(mykey, k, 64);
                if ((uint32_t)(key >> 16) != R0) { more = false; break; }     // wave-uniform (~0 never matches: regions are < 2^32 - 1)
                const uint32_t c = base + (uint32_t)k;
                const uint32_t want = (uint32_t)(key & 0xFFFFull) - 1u;       // view id of the candidate label
                unsigned long long f = 0ull; uint32_t h = 0u;
                if (in) {
                    uint32_t a = 0, b = K;
                    while (a < b) { const uint32_t m = (a + b) >> 1; if ((uint32_t)view_id[p0 + m] < want) a = m + 1; else b = m; }
                    if (a < K && (uint32_t)view_id[p0 + a] == want) { h = 1u; f = fix32(cost[p0 + a]); }
                }
                for (int o = 32; o > 0; o >>= 1) { f += __shfl_xor(f, o, 64); h += __shfl_xor(h, o, 64); }
                if (lane == leader && h) { atomicAdd(&have[c], h); atomicAdd(&sum[c], f); }
            }
            if (!more) break;
        }
    }
}
// gain of candidate c = (cur[R] - sum[c]) + (count << 32), if every face of R has the label; the region keeps the maximum
__global__ void rg_gain_kernel(const unsigned long long* __restrict__ ck, const uint32_t* __restrict__ cstart, uint32_t nC, const uint32_t* __restrict__ size,
                               const unsigned long long* __restrict__ cur, const uint32_t* __restrict__ have, const unsigned long long* __restrict__ sum,
                               long long* __restrict__ cgain, unsigned long long* __restrict__ gain) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nC) return;
    const uint32_t R = (uint32_t)(ck[c] >> 16);
    long long g = 0;
    if (have[c] == size[R]) g = (long long)(cur[R] - sum[c]) + ((long long)(cstart[c + 1] - cstart[c]) << 32);
    cgain[c] = g;
    if (g > 0) atomicMax(&gain[R], (unsigned long long)g);
}
__global__ void rg_label_kernel(const unsigned long long* __restrict__ ck, uint32_t nC, const long long* __restrict__ cgain, const unsigned long long* __restrict__ gain,
                                uint32_t* __restrict__ bestl) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nC) return;
    const uint32_t R = (uint32_t)(ck[c] >> 16);
    if (cgain[c] > 0 && (unsigned long long)cgain[c] == gain[R]) atomicMin(&bestl[R], (uint32_t)(ck[c] & 0xFFFFull));
}
// A region is NAMED by its smallest face in the caller's numbering (the oracle's): when the library keeps the nodes in its own order
// (orig != null) the union-find root is the smallest POSITION, and the name -- needed only to break ties between equal gains --
// is the minimum of the caller's ids over the region's faces.
__global__ void rg_name_kernel(const uint32_t* __restrict__ root, const uint32_t* __restrict__ orig, uint32_t n, uint32_t* __restrict__ name) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMin(&name[root[i]], orig[i]);
}
__global__ void rg_lose_kernel(const uint2* __restrict__ cut, uint32_t n_cut, const uint32_t* __restrict__ root, const unsigned long long* __restrict__ gain,
                               const uint32_t* __restrict__ name /* null: a region's name is its root */, uint32_t* __restrict__ lose) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cut) return;
    const uint32_t R = root[cut[k].x], S = root[cut[k].y];
    const unsigned long long gR = gain[R], gS = gain[S];
    if (gS > gR || (gS == gR && (name ? name[S] < name[R] : S < R))) lose[R] = 1u;      // racing stores of the same value
}
__global__ void rg_apply_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost, const uint32_t* __restrict__ root,
                                const unsigned long long* __restrict__ gain, const uint32_t* __restrict__ bestl, const uint32_t* __restrict__ lose, uint32_t n,
                                uint32_t* __restrict__ sel, uint32_t* __restrict__ lab, float* __restrict__ selcost, uint32_t* __restrict__ moved) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t R = root[i];
    if (gain[R] == 0ull || lose[R]) return;
    const uint32_t l = bestl[R], want = l - 1u, p0 = col_ptr[i], K = col_ptr[i + 1] - p0;
    uint32_t a = 0, b = K;
    while (a < b) { const uint32_t m = (a + b) >> 1; if ((uint32_t)view_id[p0 + m] < want) a = m + 1; else b = m; }
    sel[i] = a; lab[i] = l; selcost[i] = cost[p0 + a];
    if (i == R) atomicAdd(moved, 1u);
}

uint32_t read_u32(mvs_ctx* ctx, const uint32_t* d) {
    uint32_t h = 0;
    MVS_HIP(hipMemcpyAsync(&h, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    return h;
}

}  // namespace

// One round of region moves on the best labeling (whole graph; exact unaries must be in place: mrf_exact_costs).
// Returns the number of regions that moved.
uint32_t mrf_region_round(mvs_ctx* ctx) {
    resolve_best(ctx);
    hipStream_t s = ctx->stream;
    const uint32_t F = ctx->csr_faces, E = ctx->m_n_adj;
    if (F == 0) return 0;
    Prof pr(ctx, "mrf_region");
    const unsigned nb = (F + 255) / 256;
    ctx->rg_parent.ensure((size_t)F + 2); ctx->rg_root.ensure((size_t)F + 2); ctx->rg_gain.ensure((size_t)F + 2); ctx->rg_cur.ensure((size_t)F + 2);
    ctx->rg_size.ensure((size_t)F + 2); ctx->rg_bestl.ensure((size_t)F + 2); ctx->rg_lose.ensure((size_t)F + 2); ctx->rg_cfirst.ensure((size_t)F + 2);
    ctx->rg_flag.ensure((size_t)E + 2); ctx->rg_pos.ensure((size_t)E + 2);
    hipLaunchKernelGGL(rg_init_kernel, dim3(nb), dim3(256), 0, s, ctx->rg_parent.p, ctx->rg_gain.p, ctx->rg_cur.p, ctx->rg_size.p, ctx->rg_bestl.p, ctx->rg_lose.p, ctx->rg_cfirst.p, F); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(rg_hook_kernel, dim3(nb), dim3(256), 0, s, ctx->r_adj_ptr, ctx->r_adj, ctx->b_lab, ctx->rg_parent.p, F); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(rg_flatten_kernel, dim3(nb), dim3(256), 0, s, ctx->rg_parent.p, ctx->rg_root.p, F); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(rg_cut_flag_kernel, dim3(nb), dim3(256), 0, s, ctx->r_adj_ptr, ctx->r_adj, ctx->b_lab, F, E, ctx->rg_flag.p); MVS_LAUNCH_CHECK();
    exclusive_scan_u32(ctx, ctx->rg_flag.p, ctx->rg_pos.p, (size_t)E + 1, nullptr);
    const uint32_t n_cut = read_u32(ctx, ctx->rg_pos.p + E);
    if (n_cut == 0) return 0;
    ctx->rg_key.ensure((size_t)n_cut + 2); ctx->rg_key2.ensure((size_t)n_cut + 2); ctx->rg_cut.ensure((size_t)n_cut + 2);
    hipLaunchKernelGGL(rg_cut_emit_kernel, dim3(nb), dim3(256), 0, s, ctx->r_adj_ptr, ctx->r_adj, ctx->b_lab, ctx->rg_root.p, ctx->rg_flag.p, ctx->rg_pos.p, F, ctx->rg_key.p, ctx->rg_cut.p); MVS_LAUNCH_CHECK();
    size_t tmp = 0;
    MVS_HIP(rocprim::radix_sort_keys(nullptr, tmp, ctx->rg_key.p, ctx->rg_key2.p, n_cut, 0, 48, s));
    ctx->sort_tmp.ensure(tmp + 16);
    MVS_HIP(rocprim::radix_sort_keys(ctx->sort_tmp.p, tmp, ctx->rg_key.p, ctx->rg_key2.p, n_cut, 0, 48, s));
    // unique candidates
    ctx->rg_flag.ensure((size_t)n_cut + 2); ctx->rg_pos.ensure((size_t)n_cut + 2);
    hipLaunchKernelGGL(rg_first_kernel, dim3((n_cut + 256) / 256), dim3(256), 0, s, ctx->rg_key2.p, n_cut, ctx->rg_flag.p); MVS_LAUNCH_CHECK();
    exclusive_scan_u32(ctx, ctx->rg_flag.p, ctx->rg_pos.p, (size_t)n_cut + 1, nullptr);
    const uint32_t nC = read_u32(ctx, ctx->rg_pos.p + n_cut);
    ctx->rg_ck.ensure((size_t)nC + 2); ctx->rg_cstart.ensure((size_t)nC + 2); ctx->rg_have.ensure((size_t)nC + 2); ctx->rg_sum.ensure((size_t)nC + 2); ctx->rg_cgain.ensure((size_t)nC + 2);
    hipLaunchKernelGGL(rg_unique_kernel, dim3((n_cut + 255) / 256), dim3(256), 0, s, ctx->rg_key2.p, ctx->rg_flag.p, ctx->rg_pos.p, n_cut, nC, ctx->rg_ck.p, ctx->rg_cstart.p, ctx->rg_cfirst.p); MVS_LAUNCH_CHECK();
    MVS_HIP(hipMemsetAsync(ctx->rg_have.p, 0, ((size_t)nC + 1) * sizeof(uint32_t), s));
    MVS_HIP(hipMemsetAsync(ctx->rg_sum.p, 0, ((size_t)nC + 1) * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(rg_accumulate_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->b_sel, ctx->rg_root.p, F, ctx->rg_ck.p, ctx->rg_cfirst.p, nC,
                       ctx->rg_size.p, ctx->rg_cur.p, ctx->rg_have.p, ctx->rg_sum.p); MVS_LAUNCH_CHECK();
    const unsigned cb = (nC + 255) / 256;
    hipLaunchKernelGGL(rg_gain_kernel, dim3(cb), dim3(256), 0, s, ctx->rg_ck.p, ctx->rg_cstart.p, nC, ctx->rg_size.p, ctx->rg_cur.p, ctx->rg_have.p, ctx->rg_sum.p, ctx->rg_cgain.p, ctx->rg_gain.p); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(rg_label_kernel, dim3(cb), dim3(256), 0, s, ctx->rg_ck.p, nC, ctx->rg_cgain.p, ctx->rg_gain.p, ctx->rg_bestl.p); MVS_LAUNCH_CHECK();
    const uint32_t* name = nullptr;
    if (ctx->t_perm) {
        ctx->rg_name.ensure((size_t)F + 2);
        MVS_HIP(hipMemsetAsync(ctx->rg_name.p, 0xFF, (size_t)F * sizeof(uint32_t), s));
        hipLaunchKernelGGL(rg_name_kernel, dim3(nb), dim3(256), 0, s, ctx->rg_root.p, ctx->t_perm, F, ctx->rg_name.p); MVS_LAUNCH_CHECK();
        name = ctx->rg_name.p;
    }
    hipLaunchKernelGGL(rg_lose_kernel, dim3((n_cut + 255) / 256), dim3(256), 0, s, ctx->rg_cut.p, n_cut, ctx->rg_root.p, ctx->rg_gain.p, name, ctx->rg_lose.p); MVS_LAUNCH_CHECK();
    ctx->m_moved.ensure(8 + 2 * 64);
    MVS_HIP(hipMemsetAsync(ctx->m_moved.p + 7, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(rg_apply_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->rg_root.p, ctx->rg_gain.p, ctx->rg_bestl.p, ctx->rg_lose.p, F,
                       ctx->b_sel, ctx->b_lab, ctx->b_cost, ctx->m_moved.p + 7); MVS_LAUNCH_CHECK();
    ctx->icm_dirty_valid = false;   // labels changed behind the ICM's active list
    return read_u32(ctx, ctx->m_moved.p + 7);
}

}  // namespace mvs
