// k_mesh.hip -- row f1 of SURVEY.md section 8(f): the two stages immediately before the view-selection path,
//   tex::prepare_mesh          (libs/tex/prepare_mesh.cpp:14-70)   redundant-face removal + face normals
//   tex::build_adjacency_graph (libs/tex/build_adjacency_graph.cpp:16-53) face adjacency in UniGraph list order
// Both are integer / sorting work (HBM bound); results are exact (pinned to the reference's own two files through oracle/_ref,
// tests/test_reference_pins.py, including meshes with repeated-vertex faces).
#include "ctx.h"
#include <rocprim/rocprim.hpp>

namespace mvs {

void build_vertex_faces(mvs_ctx* ctx, const uint32_t* d_faces, uint32_t F, uint32_t NV);   // k_bvh.hip
mvs_status api_fail(mvs_status st, const std::string& msg);

namespace {

constexpr int MAX_NEIGHBOURS = 48;   // distinct neighbours of one face (3 for a manifold mesh)

__global__ void edge_key_kernel(const uint32_t* __restrict__ faces, uint32_t n_faces, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
                                uint32_t* __restrict__ n_repeated) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n_faces) return;
    const uint32_t f = i / 3, k = i - 3 * f;
    const uint32_t a = faces[3 * (size_t)f + k], b = faces[3 * (size_t)f + (k + 1) % 3];   // edges (v1,v2), (v2,v3), (v3,v1): build_adjacency_graph.cpp:31-34
    keys[i] = (unsigned long long)min(a, b) << 32 | max(a, b);
    vals[i] = i;
    if (a == b) atomicAdd(n_repeated, 1u);    // a face with a repeated vertex: the mesh takes adjacency_general_kernel
}
__global__ void invert_kernel(const uint32_t* __restrict__ vals, uint32_t n, uint32_t* __restrict__ pos) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) pos[vals[p]] = p;
}

// Neighbour list of face f in the order UniGraph::add_edge (uni_graph.h:86-93) produces when
// build_adjacency_graph walks the faces in ascending order:
//   [neighbours g < f, ascending (pushed when g was processed)] ++ [neighbours g > f in edge order, first occurrence]
// Faces sharing an edge are taken in ascending id (the stable sort keeps (face, edge) order inside a key run).
template <bool WRITE>
__global__ void adjacency_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ pos,
                                 uint32_t n_faces, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ adj_ptr, uint32_t* __restrict__ adj,
                                 uint32_t* __restrict__ overflow) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    const uint32_t n_rec = 3 * n_faces;
    uint32_t small[MAX_NEIGHBOURS], large[MAX_NEIGHBOURS];
    int ns = 0, nl = 0;
    bool over = false;
    for (int k = 0; k < 3; ++k) {
        const uint32_t p = pos[3 * f + k];
        const unsigned long long key = keys[p];
        uint32_t s = p;
        while (s > 0 && keys[s - 1] == key) --s;
        for (uint32_t q = s; q < n_rec && keys[q] == key; ++q) {
            const uint32_t g = vals[q] / 3;
            if (g == f) continue;                                  /* :41 avoid self referencing */
            if (g < f) {                                           // sorted unique insert
                int at = 0; bool dup = false;
                while (at < ns && small[at] < g) ++at;
                if (at < ns && small[at] == g) dup = true;
                if (!dup) {
                    if (ns >= MAX_NEIGHBOURS) { over = true; continue; }
                    for (int m = ns; m > at; --m) small[m] = small[m - 1];
                    small[at] = g; ++ns;
                }
            } else {                                               // first occurrence order
                bool dup = false;
                for (int m = 0; m < nl; ++m) dup = dup || large[m] == g;
                if (!dup) { if (nl >= MAX_NEIGHBOURS) { over = true; continue; } large[nl++] = g; }
            }
        }
    }
    if (over) atomicAdd(overflow, 1u);
    if (!WRITE) { cnt[f] = (uint32_t)(ns + nl); if (f == 0) cnt[n_faces] = 0; }
    else {
        uint32_t o = adj_ptr[f];
        for (int m = 0; m < ns; ++m) adj[o++] = small[m];
        for (int m = 0; m < nl; ++m) adj[o++] = large[m];
    }
}

// The same lists for a mesh that has faces with a REPEATED vertex (a, a, b).  The reference asks MeshInfo for the faces of
// every "edge" of a face (build_adjacency_graph.cpp:31-34); get_faces_for_edge(x, y) is the intersection of the two vertices'
// face lists, so for the edge (a, a) it returns EVERY face at a.  Queries are then no longer symmetric (the repeated-vertex
// face finds its neighbours, they do not find it), and add_edge order gives, for face f:
//   [g < f whose queries find f, ascending] ++ [what f's own queries find, in query order, not yet listed]
//   ++ [g > f whose queries find f but which f's queries did not find, ascending]
// One thread per face over the union of the faces at its vertices (vertex -> faces CSR); a rare path, kept simple.
constexpr int MAX_CANDIDATES = 128;
template <bool WRITE>
__global__ void adjacency_general_kernel(const uint32_t* __restrict__ faces, uint32_t n_faces, const uint32_t* __restrict__ vf_ptr, const uint32_t* __restrict__ vf,
                                         uint32_t* __restrict__ cnt, const uint32_t* __restrict__ adj_ptr, uint32_t* __restrict__ adj, uint32_t* __restrict__ overflow) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    const uint32_t fv[3] = {faces[3 * (size_t)f], faces[3 * (size_t)f + 1], faces[3 * (size_t)f + 2]};
    uint32_t cand[MAX_CANDIDATES];
    int nc = 0;
    bool over = false;
    for (int j = 0; j < 3; ++j) {
        if ((j > 0 && fv[j] == fv[0]) || (j > 1 && fv[j] == fv[1])) continue;
        for (uint32_t p = vf_ptr[fv[j]]; p < vf_ptr[fv[j] + 1]; ++p) {
            const uint32_t g = vf[p];
            if (g == f) continue;                                  /* :41 avoid self referencing */
            int at = 0;
            while (at < nc && cand[at] < g) ++at;
            if (at < nc && cand[at] == g) continue;
            if (nc >= MAX_CANDIDATES) { over = true; continue; }
            for (int m = nc; m > at; --m) cand[m] = cand[m - 1];
            cand[at] = g; ++nc;
        }
    }
    auto in_f = [&](uint32_t v) { return v == fv[0] || v == fv[1] || v == fv[2]; };
    // does one of g's three queries return f?
    auto finds_f = [&](uint32_t g) {
        const uint32_t gv[3] = {faces[3 * (size_t)g], faces[3 * (size_t)g + 1], faces[3 * (size_t)g + 2]};
        bool r = false;
        for (int k = 0; k < 3; ++k) { const uint32_t x = gv[k], y = gv[(k + 1) % 3]; r = r || (in_f(x) && in_f(y)); }
        return r;
    };
    // does f's query (x, y) return g?
    auto query_has = [&](uint32_t g, uint32_t x, uint32_t y) {
        const uint32_t g0 = faces[3 * (size_t)g], g1 = faces[3 * (size_t)g + 1], g2 = faces[3 * (size_t)g + 2];
        return (g0 == x || g1 == x || g2 == x) && (g0 == y || g1 == y || g2 == y);
    };
    uint32_t out[MAX_NEIGHBOURS];
    int no = 0;
    auto listed = [&](uint32_t g) { bool r = false; for (int m = 0; m < no; ++m) r = r || out[m] == g; return r; };
    auto push = [&](uint32_t g) { if (no >= MAX_NEIGHBOURS) over = true; else out[no++] = g; };
    for (int c = 0; c < nc; ++c) if (cand[c] < f && finds_f(cand[c])) push(cand[c]);
    for (int k = 0; k < 3; ++k) {
        const uint32_t x = fv[k], y = fv[(k + 1) % 3];
        for (int c = 0; c < nc; ++c) if (query_has(cand[c], x, y) && !listed(cand[c])) push(cand[c]);
    }
    for (int c = 0; c < nc; ++c) if (cand[c] > f && finds_f(cand[c]) && !listed(cand[c])) push(cand[c]);
    if (over) atomicAdd(overflow, 1u);
    if (!WRITE) { cnt[f] = (uint32_t)no; if (f == 0) cnt[n_faces] = 0; }
    else { uint32_t o = adj_ptr[f]; for (int m = 0; m < no; ++m) adj[o++] = out[m]; }
}

// remove_redundant_faces (prepare_mesh.cpp:14-55): keep[f] = 0 iff a face g > f touching a vertex of f has all its vertices in f
__global__ void redundant_kernel(const uint32_t* __restrict__ faces, uint32_t n_faces, const uint32_t* __restrict__ vf_ptr, const uint32_t* __restrict__ vf,
                                 uint32_t* __restrict__ keep) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f > n_faces) return;
    if (f == n_faces) { keep[f] = 0; return; }
    const uint32_t v0 = faces[3 * (size_t)f], v1 = faces[3 * (size_t)f + 1], v2 = faces[3 * (size_t)f + 2];
    const uint32_t fv[3] = {v0, v1, v2};
    bool redundant = false;
    for (int j = 0; j < 3 && !redundant; ++j)
        for (uint32_t p = vf_ptr[fv[j]]; p < vf_ptr[fv[j] + 1] && !redundant; ++p) {
            const uint32_t g = vf[p];
            if (g <= f) continue;                                  /* :29 remove only the redundant face with smaller id */
            bool identical = true;
            for (int l = 0; l < 3; ++l) { const uint32_t v = faces[3 * (size_t)g + l]; identical = identical && (v == v0 || v == v1 || v == v2); }
            redundant = identical;
        }
    keep[f] = redundant ? 0u : 1u;
}

// compaction in face order + face normals = normalised (b - a) x (c - a), zero vector if degenerate (ensure_normals, prepare_mesh.cpp:64)
__global__ void compact_faces_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, uint32_t n_faces, const uint32_t* __restrict__ keep,
                                     const uint32_t* __restrict__ dst, uint32_t* __restrict__ faces_out, float* __restrict__ normals_out) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_faces || !keep[f]) return;
    const size_t o = dst[f];
    const uint32_t i0 = faces[3 * (size_t)f], i1 = faces[3 * (size_t)f + 1], i2 = faces[3 * (size_t)f + 2];
    faces_out[3 * o] = i0; faces_out[3 * o + 1] = i1; faces_out[3 * o + 2] = i2;
    const V3 a = {verts[3 * (size_t)i0], verts[3 * (size_t)i0 + 1], verts[3 * (size_t)i0 + 2]};
    const V3 b = {verts[3 * (size_t)i1], verts[3 * (size_t)i1 + 1], verts[3 * (size_t)i1 + 2]};
    const V3 c = {verts[3 * (size_t)i2], verts[3 * (size_t)i2 + 1], verts[3 * (size_t)i2 + 2]};
    V3 n = cross(b - a, c - a);
    const float len = norm(n);
    if (len != 0.0f) n = n / len;
    normals_out[3 * o] = n.x; normals_out[3 * o + 1] = n.y; normals_out[3 * o + 2] = n.z;
}

uint32_t read_u32(mvs_ctx* ctx, const uint32_t* d) {
    uint32_t h = 0;
    MVS_HIP(hipMemcpyAsync(&h, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    return h;
}

}  // namespace

// adjacency of `d_faces` into ctx->g_adj_ptr / ctx->g_adj (device); returns the number of list entries
uint64_t build_adjacency(mvs_ctx* ctx, const uint32_t* d_faces, uint32_t F, uint32_t NV) {
    hipStream_t s = ctx->stream;
    ctx->g_adj_ptr.ensure((size_t)F + 2);
    if (F == 0) { MVS_HIP(hipMemsetAsync(ctx->g_adj_ptr.p, 0, 2 * sizeof(uint32_t), s)); ctx->g_adj.ensure(4); return 0; }
    const uint32_t n = 3 * F;
    ctx->g_keys.ensure(n); ctx->g_keys2.ensure(n); ctx->g_vals.ensure(n); ctx->g_vals2.ensure(n); ctx->g_pos.ensure(n); ctx->g_cnt.ensure((size_t)F + 2);
    ctx->m_moved.ensure(8);
    uint32_t* overflow = ctx->m_moved.p + 6;     // [0] overflow, [1] edges (a, a)
    MVS_HIP(hipMemsetAsync(overflow, 0, 2 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(edge_key_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_faces, F, ctx->g_keys.p, ctx->g_vals.p, overflow + 1);
    MVS_LAUNCH_CHECK();
    if (read_u32(ctx, overflow + 1)) {           // faces with a repeated vertex: asymmetric edge queries, general kernel
        build_vertex_faces(ctx, d_faces, F, NV);
        hipLaunchKernelGGL(adjacency_general_kernel<false>, dim3((F + 127) / 128), dim3(128), 0, s, d_faces, F, (const uint32_t*)ctx->vf_ptr.p,
                           (const uint32_t*)ctx->vf.p, ctx->g_cnt.p, (const uint32_t*)nullptr, (uint32_t*)nullptr, overflow);
        MVS_LAUNCH_CHECK();
        exclusive_scan_u32(ctx, ctx->g_cnt.p, ctx->g_adj_ptr.p, (size_t)F + 1, nullptr);
        const uint32_t total = read_u32(ctx, ctx->g_adj_ptr.p + F);
        if (read_u32(ctx, overflow)) throw StatusError(MVS_ERR_UNSUPPORTED, "a face has more than 48 distinct neighbours (or a repeated vertex with more than 128 faces)");
        ctx->g_adj.ensure((size_t)total + 1);
        hipLaunchKernelGGL(adjacency_general_kernel<true>, dim3((F + 127) / 128), dim3(128), 0, s, d_faces, F, (const uint32_t*)ctx->vf_ptr.p,
                           (const uint32_t*)ctx->vf.p, ctx->g_cnt.p, (const uint32_t*)ctx->g_adj_ptr.p, ctx->g_adj.p, overflow);
        MVS_LAUNCH_CHECK();
        return total;
    }
    int vbits = 1; while ((1ull << vbits) < (unsigned long long)NV + 1 && vbits < 32) ++vbits;
    size_t tmp_bytes = 0;
    MVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ctx->g_keys.p, ctx->g_keys2.p, ctx->g_vals.p, ctx->g_vals2.p, n, 0, 32 + vbits, s));
    ctx->sort_tmp.ensure(tmp_bytes + 16);
    MVS_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->g_keys.p, ctx->g_keys2.p, ctx->g_vals.p, ctx->g_vals2.p, n, 0, 32 + vbits, s));
    hipLaunchKernelGGL(invert_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ctx->g_vals2.p, n, ctx->g_pos.p);
    MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(adjacency_kernel<false>, dim3((F + 127) / 128), dim3(128), 0, s, ctx->g_keys2.p, ctx->g_vals2.p, ctx->g_pos.p, F, ctx->g_cnt.p,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, overflow);
    MVS_LAUNCH_CHECK();
    exclusive_scan_u32(ctx, ctx->g_cnt.p, ctx->g_adj_ptr.p, (size_t)F + 1, nullptr);
    const uint32_t total = read_u32(ctx, ctx->g_adj_ptr.p + F);
    if (read_u32(ctx, overflow)) throw StatusError(MVS_ERR_UNSUPPORTED, "a face has more than 48 distinct neighbours");
    ctx->g_adj.ensure((size_t)total + 1);
    hipLaunchKernelGGL(adjacency_kernel<true>, dim3((F + 127) / 128), dim3(128), 0, s, ctx->g_keys2.p, ctx->g_vals2.p, ctx->g_pos.p, F, ctx->g_cnt.p,
                       (const uint32_t*)ctx->g_adj_ptr.p, ctx->g_adj.p, overflow);
    MVS_LAUNCH_CHECK();
    return total;
}

// prepare_mesh of (d_verts, d_faces): kept faces + their normals into ctx->g_faces / ctx->g_normals; returns the number kept
uint32_t prepare_mesh(mvs_ctx* ctx, const float* d_verts, const uint32_t* d_faces, uint32_t F, uint32_t NV) {
    hipStream_t s = ctx->stream;
    ctx->g_faces.ensure(3 * (size_t)F + 4); ctx->g_normals.ensure(3 * (size_t)F + 4);
    if (F == 0) return 0;
    build_vertex_faces(ctx, d_faces, F, NV);
    ctx->g_cnt.ensure((size_t)F + 2); ctx->g_pos.ensure((size_t)F + 2);
    hipLaunchKernelGGL(redundant_kernel, dim3((F + 256) / 256), dim3(256), 0, s, d_faces, F, ctx->vf_ptr.p, ctx->vf.p, ctx->g_cnt.p);
    MVS_LAUNCH_CHECK();
    exclusive_scan_u32(ctx, ctx->g_cnt.p, ctx->g_pos.p, (size_t)F + 1, nullptr);
    const uint32_t kept = read_u32(ctx, ctx->g_pos.p + F);
    hipLaunchKernelGGL(compact_faces_kernel, dim3((F + 255) / 256), dim3(256), 0, s, d_verts, d_faces, F, ctx->g_cnt.p, ctx->g_pos.p, ctx->g_faces.p, ctx->g_normals.p);
    MVS_LAUNCH_CHECK();
    return kept;
}

}  // namespace mvs

using namespace mvs;

#define MVS_API_BEGIN try { MVS_HIP(hipSetDevice(ctx->device));
#define MVS_API_END                                                               \
    } catch (const StatusError& e) { return api_fail(e.st, e.what()); }           \
      catch (const HipError& e) { return api_fail(MVS_ERR_HIP, e.what()); }       \
      catch (const std::exception& e) { return api_fail(MVS_ERR_HIP, e.what()); } \
    return MVS_OK;

extern "C" {

mvs_status mvs_ctx_build_adjacency(mvs_ctx* ctx, uint32_t** adj_ptr_device, uint32_t** adj_device, uint64_t* n_entries) {
    if (!ctx) return api_fail(MVS_ERR_INVALID, "ctx is null");
    if (!ctx->d_faces && ctx->n_faces) return api_fail(MVS_ERR_STATE, "no mesh resident");
    MVS_API_BEGIN
    Prof pr(ctx, "build_adjacency");
    const uint64_t n = build_adjacency(ctx, ctx->d_faces, ctx->n_faces, ctx->n_verts);
    pr.end();
    ctx->g_adj_entries = n; ctx->have_adj = true;
    if (adj_ptr_device) *adj_ptr_device = ctx->g_adj_ptr.p;
    if (adj_device) *adj_device = ctx->g_adj.p;
    if (n_entries) *n_entries = n;
    MVS_API_END
}

mvs_status mvs_build_adjacency_graph(uint32_t n_verts, uint32_t n_faces, const uint32_t* faces, uint32_t* adj_ptr_out, uint32_t** adj_out, uint64_t* n_entries) {
    if ((!faces && n_faces) || !adj_ptr_out || !adj_out) return api_fail(MVS_ERR_INVALID, "null argument");
    mvs_ctx* ctx = nullptr;
    mvs_status st = mvs_ctx_create(mvs::default_device(), &ctx);
    if (st != MVS_OK) return st;
    try {
        DBuf<uint32_t> d_faces; d_faces.ensure(3 * (size_t)n_faces + 4);
        if (n_faces) MVS_HIP(hipMemcpy(d_faces.p, faces, 3 * (size_t)n_faces * sizeof(uint32_t), hipMemcpyHostToDevice));
        const uint64_t n = build_adjacency(ctx, d_faces.p, n_faces, n_verts);
        MVS_HIP(hipMemcpyAsync(adj_ptr_out, ctx->g_adj_ptr.p, ((size_t)n_faces + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        *adj_out = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
        if (n) MVS_HIP(hipMemcpyAsync(*adj_out, ctx->g_adj.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        if (n_entries) *n_entries = n;
    } catch (const StatusError& e) { mvs_ctx_destroy(ctx); return api_fail(e.st, e.what()); }
      catch (const std::exception& e) { mvs_ctx_destroy(ctx); return api_fail(MVS_ERR_HIP, e.what()); }
    mvs_ctx_destroy(ctx);
    return MVS_OK;
}

mvs_status mvs_prepare_mesh(uint32_t n_verts, const float* verts, uint32_t n_faces, const uint32_t* faces, uint32_t* faces_out, float* normals_out, uint32_t* n_kept) {
    if ((!verts && n_verts) || (!faces && n_faces) || !faces_out || !normals_out || !n_kept) return api_fail(MVS_ERR_INVALID, "null argument");
    mvs_ctx* ctx = nullptr;
    mvs_status st = mvs_ctx_create(mvs::default_device(), &ctx);
    if (st != MVS_OK) return st;
    try {
        DBuf<uint32_t> d_faces; DBuf<float> d_verts; d_faces.ensure(3 * (size_t)n_faces + 4); d_verts.ensure(3 * (size_t)n_verts + 4);
        if (n_faces) MVS_HIP(hipMemcpy(d_faces.p, faces, 3 * (size_t)n_faces * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (n_verts) MVS_HIP(hipMemcpy(d_verts.p, verts, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice));
        const uint32_t kept = prepare_mesh(ctx, d_verts.p, d_faces.p, n_faces, n_verts);
        if (kept) {
            MVS_HIP(hipMemcpyAsync(faces_out, ctx->g_faces.p, 3 * (size_t)kept * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            MVS_HIP(hipMemcpyAsync(normals_out, ctx->g_normals.p, 3 * (size_t)kept * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        }
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        *n_kept = kept;
    } catch (const StatusError& e) { mvs_ctx_destroy(ctx); return api_fail(e.st, e.what()); }
      catch (const std::exception& e) { mvs_ctx_destroy(ctx); return api_fail(MVS_ERR_HIP, e.what()); }
    mvs_ctx_destroy(ctx);
    return MVS_OK;
}

}  // extern "C"
