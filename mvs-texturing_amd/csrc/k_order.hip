// k_order.hip -- the boundary between the caller's face numbering and the library's own layout (ctx.h "the library's own
// mesh layout"; DESIGN.md "Face order and partition").
//
// The reference hands the path its faces in mesh-file order (calculate_data_costs.cpp:136-138, texrecon.cpp:73-92) and
// nothing here may depend on that order being coherent.  Internally every per-face array -- bit matrices, the cost table,
// the solver's nodes, the parts of the sharded path -- is indexed by the position of the face on a Hilbert curve
// (k_bvh.hip build_scene_order).  This file holds what translates at the ABI:
//   * the cost table back into the caller's order (mvs_ctx_costs_download / mvs_data_costs / tex::calculate_data_costs);
//   * the adjacency lists (UniGraph, build_adjacency_graph.cpp:16-53) into the table's order, LIST ORDER KEPT -- the solver
//     sums messages in list order, and the colouring is keyed on the caller's ids (k_mrf.hip), so labels do not depend on
//     the layout;
//   * mvs_partition_faces / mvs_ctx_partition_faces: the order itself and its cut into `world` equal contiguous parts -- the
//     partition of the sharded path (METIS is not available).
#include "ctx.h"

namespace mvs {
void build_scene_order(mvs_ctx* ctx);
bool scene_order_commit(mvs_ctx* ctx);
mvs_status api_fail(mvs_status st, const std::string& msg);

namespace {

// cnt[f] = length of the column of the caller's face f (cnt[F] = 0)
__global__ void order_count_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ pos, uint32_t F, uint32_t* __restrict__ cnt) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f > F) return;
    uint32_t c = 0;
    if (f < F) { const uint32_t p = pos[f]; c = col_ptr[p + 1] - col_ptr[p]; }
    cnt[f] = c;
}
// 16 lanes per face: the column of the caller's face f (position pos[f] of the table) to its place in the caller's order.  Faces
// that follow each other in the caller's numbering write one contiguous stream; the reads are whole columns.
__global__ void __launch_bounds__(256) order_copy_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                         const float* __restrict__ q /* may be null */, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ out_ptr,
                                                         uint32_t F, uint16_t* __restrict__ out_view, float* __restrict__ out_cost, float* __restrict__ out_q) {
    const uint32_t f = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, gl = threadIdx.x & 15;
    if (f >= F) return;
    const uint32_t p = pos[f], s0 = col_ptr[p], K = col_ptr[p + 1] - s0, d0 = out_ptr[f];
    for (uint32_t t = gl; t < K; t += 16) {
        out_view[d0 + t] = view_id[s0 + t]; out_cost[d0 + t] = cost[s0 + t];
        if (q) out_q[d0 + t] = q[s0 + t];
    }
}
// deg[p] = list length of the caller's face perm[p] (deg[F] = 0)
__global__ void adj_deg_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ perm, uint32_t F, uint32_t* __restrict__ deg) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > F) return;
    uint32_t d = 0;
    if (p < F) { const uint32_t o = perm[p]; d = adj_ptr[o + 1] - adj_ptr[o]; }
    deg[p] = d;
}
__global__ void adj_fill_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pos,
                                const uint32_t* __restrict__ new_ptr, uint32_t F, uint32_t* __restrict__ new_adj) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= F) return;
    const uint32_t o = perm[p], e0 = adj_ptr[o], n = adj_ptr[o + 1] - e0, d0 = new_ptr[p];
    for (uint32_t k = 0; k < n; ++k) { const uint32_t j = adj[e0 + k]; new_adj[d0 + k] = j < F ? pos[j] : j; }   // list order kept
}
__global__ void iota_u32_kernel(uint32_t* __restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

}  // namespace

// The active table in the caller's order -> ctx->u_ptr / u_view / u_cost (/ u_q); cached until the table changes.  A table that
// already is in the caller's order (t_perm == null) is not copied: the function then returns false and the caller reads r_*.
bool table_to_caller_order(mvs_ctx* ctx, bool with_quality) {
    if (!ctx->t_perm || !ctx->t_pos) return false;   // (no inverse: the table of a face range, whose columns stay in position order)
    const bool want_q = with_quality && ctx->csr_q_valid;
    if (ctx->u_valid && (!want_q || ctx->u_q_valid)) return true;
    hipStream_t s = ctx->stream;
    const uint32_t F = ctx->csr_faces; const size_t nnz = ctx->csr_nnz;
    Prof pr(ctx, "order_table_out");
    ctx->u_cnt.ensure((size_t)F + 2); ctx->u_ptr.ensure((size_t)F + 2); ctx->u_view.ensure(nnz + 1); ctx->u_cost.ensure(nnz + 1);
    if (want_q) ctx->u_q.ensure(nnz + 1);
    hipLaunchKernelGGL(order_count_kernel, dim3((F + 256) / 256), dim3(256), 0, s, ctx->r_ptr, ctx->t_pos, F, ctx->u_cnt.p); MVS_LAUNCH_CHECK();
    exclusive_scan_u32(ctx, ctx->u_cnt.p, ctx->u_ptr.p, (size_t)F + 1, nullptr);
    if (F) {
        hipLaunchKernelGGL(order_copy_kernel, dim3((unsigned)(((size_t)F * 16 + 255) / 256)), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_cost,
                           want_q ? (const float*)ctx->csr_q.p : (const float*)nullptr, ctx->t_pos, (const uint32_t*)ctx->u_ptr.p, F, ctx->u_view.p, ctx->u_cost.p, ctx->u_q.p);
        MVS_LAUNCH_CHECK();
    }
    ctx->u_valid = true; ctx->u_q_valid = want_q;
    return true;
}

// adjacency lists of F faces given in the caller's numbering (DEVICE arrays, E entries) -> out_ptr / out_adj in the order `perm`
// (position -> caller's id; `pos` its inverse), list order kept
void renumber_adjacency(mvs_ctx* ctx, uint32_t F, const uint32_t* perm, const uint32_t* pos, const uint32_t* d_adj_ptr, const uint32_t* d_adj, size_t E,
                        DBuf<uint32_t>& out_ptr, DBuf<uint32_t>& out_adj) {
    hipStream_t s = ctx->stream;
    Prof pr(ctx, "order_adjacency");
    out_ptr.ensure((size_t)F + 2); out_adj.ensure(E + 1); ctx->m_tmp_c.ensure((size_t)F + 2);
    hipLaunchKernelGGL(adj_deg_kernel, dim3((F + 256) / 256), dim3(256), 0, s, d_adj_ptr, perm, F, ctx->m_tmp_c.p); MVS_LAUNCH_CHECK();
    exclusive_scan_u32(ctx, ctx->m_tmp_c.p, out_ptr.p, (size_t)F + 1, nullptr);
    if (F) { hipLaunchKernelGGL(adj_fill_kernel, dim3((F + 255) / 256), dim3(256), 0, s, d_adj_ptr, d_adj, perm, pos, (const uint32_t*)out_ptr.p, F, out_adj.p); MVS_LAUNCH_CHECK(); }
}
// ... of the active table -> ctx->m_adj_ptr / m_adj
void adjacency_to_table_order(mvs_ctx* ctx, const uint32_t* d_adj_ptr, const uint32_t* d_adj, size_t E) {
    renumber_adjacency(ctx, ctx->csr_faces, ctx->t_perm, ctx->t_pos, d_adj_ptr, d_adj, E, ctx->m_adj_ptr, ctx->m_adj);
}

}  // namespace mvs

using namespace mvs;

#define MVS_API_BEGIN try {
#define MVS_API_END                                                               \
    } catch (const StatusError& e) { return api_fail(e.st, e.what()); }           \
      catch (const HipError& e) { return api_fail(MVS_ERR_HIP, e.what()); }       \
      catch (const std::exception& e) { return api_fail(MVS_ERR_HIP, e.what()); } \
    return MVS_OK;

static void equal_cut(uint32_t F, int world, uint32_t* part_begin) {
    for (int q = 0; q <= world; ++q) part_begin[q] = (uint32_t)(((uint64_t)F * (uint64_t)q) / (uint64_t)world);
}

extern "C" {

/* the layout of the resident mesh: perm_device[p] = the caller's id of the face at position p; part_begin[0 .. world] (host) = its cut
 * into `world` contiguous parts of (nearly) equal size */
mvs_status mvs_ctx_partition_faces(mvs_ctx* ctx, int world, uint32_t* perm_device, uint32_t* part_begin) {
    if (!ctx || world < 1 || (!perm_device && !part_begin)) return api_fail(MVS_ERR_INVALID, "bad argument");
    if (!ctx->d_verts || !ctx->d_faces) return api_fail(MVS_ERR_STATE, "no mesh resident");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    const uint32_t F = ctx->n_faces;
    if (perm_device && F) {
        Prof pr(ctx, "partition");
        build_scene_order(ctx); (void)scene_order_commit(ctx);
        if (ctx->mesh_ordered) MVS_HIP(hipMemcpyAsync(perm_device, ctx->f_perm.p, (size_t)F * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        else { hipLaunchKernelGGL(iota_u32_kernel, dim3((F + 255) / 256), dim3(256), 0, ctx->stream, perm_device, F); MVS_LAUNCH_CHECK(); }   // option "face_order" = 0: the caller's own order
        pr.end();
        MVS_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (part_begin) equal_cut(F, world, part_begin);
    MVS_API_END
}

/* one-shot form on host arrays (normals are not needed: mesh->face_normals may be null) */
mvs_status mvs_partition_faces(const mvs_mesh* mesh, int world, uint32_t* perm_out, uint32_t* part_begin_out) {
    if (!mesh || world < 1 || !perm_out || !part_begin_out || !mesh->verts || !mesh->faces) return api_fail(MVS_ERR_INVALID, "bad argument");
    mvs_ctx* ctx = nullptr;
    mvs_status st = mvs_ctx_create(default_device(), &ctx);
    if (st != MVS_OK) return st;
    try {
        const size_t NV = mesh->n_verts, F = mesh->n_faces;
        ctx->own_verts.ensure(3 * NV + 4); ctx->own_faces.ensure(3 * F + 4); ctx->own_normals.ensure(3 * F + 4);
        MVS_HIP(hipMemcpyAsync(ctx->own_verts.p, mesh->verts, 3 * NV * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        MVS_HIP(hipMemcpyAsync(ctx->own_faces.p, mesh->faces, 3 * F * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        MVS_HIP(hipMemsetAsync(ctx->own_normals.p, 0, 3 * F * sizeof(float), ctx->stream));
        ctx->d_verts = ctx->own_verts.p; ctx->d_faces = ctx->own_faces.p; ctx->d_normals = ctx->own_normals.p;
        ctx->n_verts = mesh->n_verts; ctx->n_faces = mesh->n_faces;
        DBuf<uint32_t> perm; perm.ensure(F + 1);
        st = mvs_ctx_partition_faces(ctx, world, perm.p, part_begin_out);
        if (st == MVS_OK && F) { MVS_HIP(hipMemcpyAsync(perm_out, perm.p, F * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream)); MVS_HIP(hipStreamSynchronize(ctx->stream)); }
    } catch (const StatusError& e) { st = api_fail(e.st, e.what()); }
      catch (const std::exception& e) { st = api_fail(MVS_ERR_HIP, e.what()); }
    mvs_ctx_destroy(ctx);
    return st;
}

/* order of the ACTIVE cost table: *ordered = 1 and perm_device[p] (caller-owned DEVICE array of n_faces words, may be NULL) = the
 * caller's face id of column p when the table lives in the library's own order; *ordered = 0 (nothing written): the caller's order.
 * The table of a face range (mvs_scene_set_face_range) has end - begin columns, in position order: ordered = 1 and that many words. */
mvs_status mvs_ctx_table_order(mvs_ctx* ctx, uint32_t* perm_device, int* ordered) {
    if (!ctx || !ordered) return api_fail(MVS_ERR_INVALID, "null argument");
    *ordered = ctx->t_perm ? 1 : 0;
    if (!ctx->t_perm || !perm_device) return MVS_OK;
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    MVS_HIP(hipMemcpyAsync(perm_device, ctx->t_perm, (size_t)ctx->csr_faces * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    MVS_API_END
}

}  // extern "C"
