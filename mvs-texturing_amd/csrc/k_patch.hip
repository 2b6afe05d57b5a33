// k_patch.hip -- row f3 of SURVEY.md section 8(f): the step immediately after the view-selection path,
//   UniGraph::get_subgraphs (libs/tex/uni_graph.cpp:21-55), called once per label by generate_texture_patches
//   (libs/tex/generate_texture_patches.cpp:469-475): the connected components of equally labelled faces.
// The reference scans all faces once per label (O(V * F)) and runs a sequential BFS per component.  Here all labels
// are handled at once:
//   1. components by lock-free union-find with "smaller id wins" links -> a component's root is its smallest face,
//      which is exactly the face the reference's ascending scan starts the component's BFS from;
//   2. components ordered by (label, root): label L's subgraphs are a contiguous run, in the reference's order;
//   3. every component's BFS queue is reproduced by one workgroup, directly in the output segment: the queue order
//      of the sequential BFS is "by position of the parent in the queue, then by adjacency-list slot", which the
//      block restores chunk by chunk with an atomicMin claim (earliest queue position wins) and a prefix sum.
// Integer work only; the output equals the reference's vectors element for element.
#include "ctx.h"
#include <rocprim/rocprim.hpp>

namespace mvs {

mvs_status api_fail(mvs_status st, const std::string& msg);

namespace {

uint32_t read_u32(mvs_ctx* ctx, const uint32_t* d) {
    uint32_t h = 0;
    MVS_HIP(hipMemcpyAsync(&h, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    return h;
}

__device__ inline uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// representative of x with path halving; parent[] only ever moves towards smaller ids, so racing writers are benign
__device__ inline uint32_t uf_find(uint32_t* parent, uint32_t x) {
    uint32_t curr = ld_agent(parent + x);
    if (curr != x) {
        uint32_t prev = x, next;
        while (curr > (next = ld_agent(parent + curr))) { st_agent(parent + prev, next); prev = curr; curr = next; }
    }
    return curr;
}

__global__ void cc_init_kernel(uint32_t* __restrict__ parent, uint32_t* __restrict__ state, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { parent[i] = i; state[i] = 0xFFFFFFFFu; }
}
__global__ void cc_check_labels_kernel(const uint32_t* __restrict__ labels, uint32_t n, uint32_t n_labels, uint32_t* __restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && labels[i] >= n_labels) *bad = 1u;   // rare: plain racy store of the same value
}
// link i with every equally labelled neighbour j < i (each undirected edge once)
__global__ void cc_hook_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const uint32_t* __restrict__ labels,
                               uint32_t* __restrict__ parent, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t L = labels[i];
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        const uint32_t j = adj[e];
        if (j >= i || labels[j] != L) continue;
        uint32_t u = uf_find(parent, i), v = uf_find(parent, j);
        while (u != v) {
            if (u < v) { const uint32_t t = u; u = v; v = t; }           // u > v: hang u below v, if u is still a root
            const uint32_t old = atomicCAS(parent + u, u, v);
            if (old == u) break;
            u = old;                                                    // u got a (smaller) parent meanwhile: retry from there
        }
    }
}
__global__ void cc_flatten_kernel(uint32_t* __restrict__ parent, uint32_t* __restrict__ root, uint32_t* __restrict__ is_root, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = i, p;
    while ((p = ld_agent(parent + r)) != r) r = p;
    root[i] = r; is_root[i] = (r == i) ? 1u : 0u;
    if (i == n - 1) is_root[n] = 0u;
}
__global__ void cc_compact_roots_kernel(const uint32_t* __restrict__ is_root, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ labels,
                                        uint32_t n, uint32_t* __restrict__ roots, uint32_t* __restrict__ root_labels) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && is_root[i]) { roots[pos[i]] = i; root_labels[pos[i]] = labels[i]; }
}
__global__ void cc_rank_kernel(const uint32_t* __restrict__ roots_sorted, uint32_t n_comp, uint32_t* __restrict__ rank_of) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_comp) rank_of[roots_sorted[c]] = c;
}
// label_ptr[L] = first component (in (label, root) order) whose label is >= L
__global__ void cc_label_ptr_kernel(const uint32_t* __restrict__ labels_sorted, uint32_t n_comp, uint32_t n_labels, uint32_t* __restrict__ label_ptr) {
    const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L > n_labels) return;
    uint32_t lo = 0, hi = n_comp;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (labels_sorted[mid] < L) lo = mid + 1; else hi = mid; }
    label_ptr[L] = lo;
}
__global__ void cc_face_key_kernel(const uint32_t* __restrict__ root, const uint32_t* __restrict__ rank_of, uint32_t n, uint32_t* __restrict__ key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = rank_of[root[i]];
}
// every component owns >= 1 face and the keys are dense, so the run starts of the sorted keys are comp_ptr
__global__ void cc_comp_ptr_kernel(const uint32_t* __restrict__ sorted, uint32_t n, uint32_t n_comp, uint32_t* __restrict__ comp_ptr) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && (p == 0 || sorted[p] != sorted[p - 1])) comp_ptr[sorted[p]] = p;
    if (p == 0) comp_ptr[n_comp] = n;
}

// One workgroup reproduces the sequential BFS queue (uni_graph.cpp:31-50) of one component in out[p0, p1).
// state[v]: 0xFFFFFFFF untouched; 0 root; else 1 + queue position of the node that pushed v.  While the chunk
// [head, head + BS) of the queue is expanded, "state < head + 1" means "pushed before this chunk" (= used[v]).
template <int BS>
__global__ void __launch_bounds__(BS) bfs_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const uint32_t* __restrict__ labels,
                                                 const uint32_t* __restrict__ roots_sorted, const uint32_t* __restrict__ comp_ptr,
                                                 uint32_t* __restrict__ state, uint32_t* __restrict__ out, uint32_t n_comp) {
    __shared__ uint32_t s_wave[BS / 64];
    __shared__ uint32_t s_total;
    const uint32_t c = blockIdx.x;
    if (c >= n_comp) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t p0 = comp_ptr[c], p1 = comp_ptr[c + 1], root = roots_sorted[c];
    if (tid == 0) { st_agent(out + p0, root); st_agent(state + root, 0u); }
    if (p1 - p0 == 1) return;                                            // single face: done (uniform)
    const uint32_t L = labels[root];
    __syncthreads();
    uint32_t head = p0, tail = p0 + 1;
    while (head < tail) {
        const uint32_t chunk = min((uint32_t)BS, tail - head), q = head + tid;
        const bool active = tid < chunk;
        uint32_t e0 = 0, e1 = 0;
        if (active) { const uint32_t u = ld_agent(out + q); e0 = adj_ptr[u]; e1 = adj_ptr[u + 1]; }
        // claim: the earliest queue position adjacent to an unused node pushes it (uni_graph.cpp:44-47)
        for (uint32_t e = e0; e < e1; ++e) {
            const uint32_t v = adj[e];
            if (labels[v] == L && ld_agent(state + v) >= head + 1u) atomicMin(state + v, q + 1u);
        }
        __syncthreads();
        // count the pushes of this node, in adjacency-list order
        uint32_t n = 0;
        for (uint32_t e = e0; e < e1; ++e) {
            const uint32_t v = adj[e];
            if (labels[v] != L || ld_agent(state + v) != q + 1u) continue;
            bool dup = false;
            for (uint32_t e2 = e0; e2 < e; ++e2) dup = dup || adj[e2] == v;
            if (!dup) ++n;
        }
        // exclusive prefix sum over the block
        uint32_t incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if ((int)(tid & 63u) >= o) incl += t; }
        if ((tid & 63u) == 63u) s_wave[tid >> 6] = incl;
        __syncthreads();
        if (tid == 0) { uint32_t acc = 0; for (int w = 0; w < BS / 64; ++w) { const uint32_t t = s_wave[w]; s_wave[w] = acc; acc += t; } s_total = acc; }
        __syncthreads();
        uint32_t k = tail + s_wave[tid >> 6] + incl - n;
        const uint32_t total = s_total;
        for (uint32_t e = e0; e < e1 && n; ++e) {
            const uint32_t v = adj[e];
            if (labels[v] != L || ld_agent(state + v) != q + 1u) continue;
            bool dup = false;
            for (uint32_t e2 = e0; e2 < e; ++e2) dup = dup || adj[e2] == v;
            if (!dup) st_agent(out + k++, v);
        }
        __syncthreads();
        head += chunk; tail += total;
    }
}

}  // namespace

// Components of equal labels; results in ctx->p_label_ptr [n_labels + 1], p_comp_ptr [C + 1], p_comp_faces [F]. Returns C.
uint32_t get_subgraphs(mvs_ctx* ctx, const uint32_t* d_adj_ptr, const uint32_t* d_adj, const uint32_t* d_labels, uint32_t F, uint32_t n_labels) {
    hipStream_t s = ctx->stream;
    ctx->p_label_ptr.ensure((size_t)n_labels + 2); ctx->p_comp_faces.ensure((size_t)F + 1);
    if (F == 0) {
        MVS_HIP(hipMemsetAsync(ctx->p_label_ptr.p, 0, ((size_t)n_labels + 1) * sizeof(uint32_t), s));
        ctx->p_comp_ptr.ensure(2); MVS_HIP(hipMemsetAsync(ctx->p_comp_ptr.p, 0, sizeof(uint32_t), s));
        return 0;
    }
    const unsigned nb = (F + 255) / 256;
    ctx->p_parent.ensure((size_t)F + 1); ctx->p_root.ensure((size_t)F + 1); ctx->p_state.ensure((size_t)F + 1);
    ctx->p_flag.ensure((size_t)F + 2); ctx->p_pos.ensure((size_t)F + 2);
    ctx->m_moved.ensure(8);
    uint32_t* bad = ctx->m_moved.p + 7;
    MVS_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(cc_check_labels_kernel, dim3(nb), dim3(256), 0, s, d_labels, F, n_labels, bad); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(cc_init_kernel, dim3(nb), dim3(256), 0, s, ctx->p_parent.p, ctx->p_state.p, F); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(cc_hook_kernel, dim3(nb), dim3(256), 0, s, d_adj_ptr, d_adj, d_labels, ctx->p_parent.p, F); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(cc_flatten_kernel, dim3(nb), dim3(256), 0, s, ctx->p_parent.p, ctx->p_root.p, ctx->p_flag.p, F); MVS_LAUNCH_CHECK();
    exclusive_scan_u32(ctx, ctx->p_flag.p, ctx->p_pos.p, (size_t)F + 1, nullptr);
    const uint32_t C = read_u32(ctx, ctx->p_pos.p + F);
    if (read_u32(ctx, bad)) throw StatusError(MVS_ERR_INVALID, "get_subgraphs: a label is >= n_labels");
    // components in (label, root) order: roots ascending, then a stable sort by label
    ctx->p_roots.ensure((size_t)C + 1); ctx->p_roots2.ensure((size_t)C + 1); ctx->p_rlab.ensure((size_t)C + 1); ctx->p_rlab2.ensure((size_t)C + 1);
    hipLaunchKernelGGL(cc_compact_roots_kernel, dim3(nb), dim3(256), 0, s, ctx->p_flag.p, ctx->p_pos.p, d_labels, F, ctx->p_roots.p, ctx->p_rlab.p); MVS_LAUNCH_CHECK();
    int lbits = 1; while ((1ull << lbits) < (unsigned long long)n_labels && lbits < 32) ++lbits;
    size_t tmp_bytes = 0;
    MVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ctx->p_rlab.p, ctx->p_rlab2.p, ctx->p_roots.p, ctx->p_roots2.p, C, 0, lbits, s));
    ctx->sort_tmp.ensure(tmp_bytes + 16);
    MVS_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->p_rlab.p, ctx->p_rlab2.p, ctx->p_roots.p, ctx->p_roots2.p, C, 0, lbits, s));
    const unsigned cb = (C + 255) / 256;
    uint32_t* rank_of = ctx->p_parent.p;   // the union-find forest is no longer needed
    hipLaunchKernelGGL(cc_rank_kernel, dim3(cb), dim3(256), 0, s, ctx->p_roots2.p, C, rank_of); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(cc_label_ptr_kernel, dim3((n_labels + 256) / 256), dim3(256), 0, s, ctx->p_rlab2.p, C, n_labels, ctx->p_label_ptr.p); MVS_LAUNCH_CHECK();
    // component sizes: sort the faces' component ranks, run starts = comp_ptr
    uint32_t* key = ctx->p_flag.p; uint32_t* key_sorted = ctx->p_pos.p;
    hipLaunchKernelGGL(cc_face_key_kernel, dim3(nb), dim3(256), 0, s, ctx->p_root.p, rank_of, F, key); MVS_LAUNCH_CHECK();
    int cbits = 1; while ((1ull << cbits) < (unsigned long long)C && cbits < 32) ++cbits;
    MVS_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, key, key_sorted, F, 0, cbits, s));
    ctx->sort_tmp.ensure(tmp_bytes + 16);
    MVS_HIP(rocprim::radix_sort_keys(ctx->sort_tmp.p, tmp_bytes, key, key_sorted, F, 0, cbits, s));
    ctx->p_comp_ptr.ensure((size_t)C + 2);
    hipLaunchKernelGGL(cc_comp_ptr_kernel, dim3(nb), dim3(256), 0, s, key_sorted, F, C, ctx->p_comp_ptr.p); MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(bfs_kernel<256>, dim3(C), dim3(256), 0, s, d_adj_ptr, d_adj, d_labels, ctx->p_roots2.p, ctx->p_comp_ptr.p, ctx->p_state.p, ctx->p_comp_faces.p, C);
    MVS_LAUNCH_CHECK();
    return C;
}

}  // namespace mvs

using namespace mvs;

extern "C" {

mvs_status mvs_ctx_get_subgraphs(mvs_ctx* ctx, uint32_t n_faces, const uint32_t* adj_ptr, const uint32_t* adj, int adj_on_device,
                                 const uint32_t* labels, int labels_on_device, uint32_t n_labels, mvs_subgraphs* out, int out_on_device) {
    if (!ctx || !out || (n_faces && (!adj_ptr || !adj || !labels))) return api_fail(MVS_ERR_INVALID, "null argument");
    try {
        MVS_HIP(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        const uint32_t* d_adj_ptr = adj_ptr; const uint32_t* d_adj = adj; const uint32_t* d_labels = labels;
        if (!adj_on_device && n_faces) {
            const size_t E = adj_ptr[n_faces];
            ctx->p_adj_ptr.ensure((size_t)n_faces + 2); ctx->p_adj.ensure(E + 1);
            MVS_HIP(hipMemcpyAsync(ctx->p_adj_ptr.p, adj_ptr, ((size_t)n_faces + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            if (E) MVS_HIP(hipMemcpyAsync(ctx->p_adj.p, adj, E * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            d_adj_ptr = ctx->p_adj_ptr.p; d_adj = ctx->p_adj.p;
        }
        if (!labels_on_device && n_faces) {
            ctx->p_labels.ensure((size_t)n_faces + 1);
            MVS_HIP(hipMemcpyAsync(ctx->p_labels.p, labels, (size_t)n_faces * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            d_labels = ctx->p_labels.p;
        }
        if (n_faces && (!adj_on_device || !labels_on_device)) MVS_HIP(hipStreamSynchronize(s));   // host buffers are borrowed for the call only
        uint32_t C;
        { Prof pr(ctx, "get_subgraphs"); C = get_subgraphs(ctx, d_adj_ptr, d_adj, d_labels, n_faces, n_labels); }
        out->n_faces = n_faces; out->n_labels = n_labels; out->n_components = C;
        if (out_on_device) {
            out->label_ptr = ctx->p_label_ptr.p; out->comp_ptr = ctx->p_comp_ptr.p; out->comp_faces = ctx->p_comp_faces.p;
        } else {
            out->label_ptr = (uint32_t*)malloc(((size_t)n_labels + 1) * sizeof(uint32_t));
            out->comp_ptr = (uint32_t*)malloc(((size_t)C + 1) * sizeof(uint32_t));
            out->comp_faces = (uint32_t*)malloc(((size_t)n_faces + 1) * sizeof(uint32_t));
            if (!out->label_ptr || !out->comp_ptr || !out->comp_faces) { mvs_subgraphs_free(out); throw StatusError(MVS_ERR_INVALID, "out of host memory"); }
            MVS_HIP(hipMemcpyAsync(out->label_ptr, ctx->p_label_ptr.p, ((size_t)n_labels + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            MVS_HIP(hipMemcpyAsync(out->comp_ptr, ctx->p_comp_ptr.p, ((size_t)C + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            if (n_faces) MVS_HIP(hipMemcpyAsync(out->comp_faces, ctx->p_comp_faces.p, (size_t)n_faces * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            MVS_HIP(hipStreamSynchronize(s));
        }
    } catch (const StatusError& e) { return api_fail(e.st, e.what()); }
      catch (const HipError& e) { return api_fail(MVS_ERR_HIP, e.what()); }
      catch (const std::exception& e) { return api_fail(MVS_ERR_HIP, e.what()); }
    return MVS_OK;
}

mvs_status mvs_get_subgraphs(uint32_t n_faces, const uint32_t* adj_ptr, const uint32_t* adj, const uint32_t* labels, uint32_t n_labels, mvs_subgraphs* out) {
    if (!out || (n_faces && (!adj_ptr || !adj || !labels))) return api_fail(MVS_ERR_INVALID, "null argument");
    mvs_ctx* ctx = nullptr;
    mvs_status st = mvs_ctx_create(mvs::default_device(), &ctx);
    if (st != MVS_OK) return st;
    st = mvs_ctx_get_subgraphs(ctx, n_faces, adj_ptr, adj, 0, labels, 0, n_labels, out, 0);
    mvs_ctx_destroy(ctx);
    return st;
}

void mvs_subgraphs_free(mvs_subgraphs* sg) {
    if (!sg) return;
    free(sg->label_ptr); free(sg->comp_ptr); free(sg->comp_faces);
    sg->label_ptr = sg->comp_ptr = sg->comp_faces = nullptr;
}

}  // extern "C"
