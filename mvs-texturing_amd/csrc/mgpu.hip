// mgpu.hip -- libmvs_blocks.so: the building blocks of include/mvs_viewsel_blocks.h (per-phase sweeps of a node range, gather /
// scatter of halo elements, the data-cost reduce hooks).  NOT in the product library: the product's sharded driver is csrc/shard.hip.
// The collectives themselves are issued by whoever drives these blocks (the test harness tests/tools/multigpu.py) on its own
// device buffers; these entry points only move data between those buffers and the solver's arrays and run the per-range kernels.
#include "ctx.h"
#include "../../include/mvs_viewsel_blocks.h"
#include <vector>

namespace mvs {
void mrf_setup(mvs_ctx* ctx, const mvs_mrf_params* params);
void mrf_sweep(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_energy(mvs_ctx* ctx, bool best, uint32_t nb0, uint32_t ne0, bool reduce = true);
void mrf_keep_best(mvs_ctx* ctx);
void mrf_sweep_phase(mvs_ctx* ctx, uint32_t phase, uint32_t nb0, uint32_t ne0, int part = MRF_PART_ALL);
void mrf_step(mvs_ctx* ctx, const unsigned long long* energy, const unsigned long long* const* peer_tab = nullptr, uint32_t n_peer = 0, uint32_t peer_off = 0);
void mrf_poll(mvs_ctx* ctx, uint32_t step, mvs_mrf_progress* out);
void mrf_icm_gain(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_icm_apply(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_labels(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0, uint32_t* d_labels, uint32_t out[2], bool caller_order = false);
void resolve_best(mvs_ctx* ctx);
void set_adjacency(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj, int on_device, bool table_order);
mvs_status api_fail(mvs_status st, const std::string& msg);

namespace {
// st != null: the array is the CURRENT decode buffer, i.e. offset st->w * buf_stride (the step kernel flips w on the device)
__global__ void gather_kernel(const uint32_t* __restrict__ src, const mvs_mrf_progress* __restrict__ st, uint32_t buf_stride, const uint32_t* __restrict__ idx, uint64_t n, uint32_t* __restrict__ dst) {
    if (st) src += (size_t)st->w * buf_stride;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[k] = src[idx[k]];
}
__global__ void scatter_kernel(uint32_t* __restrict__ dst, const mvs_mrf_progress* __restrict__ st, uint32_t buf_stride, const uint32_t* __restrict__ idx, uint64_t n, const uint32_t* __restrict__ src) {
    if (st) dst += (size_t)st->w * buf_stride;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[idx[k]] = src[k];
}
// combined addressing: index < 2^31 -> array a (messages), else array b[index & 0x7FFFFFFF] (labels):
// one gather / scatter per sweep moves cut-edge messages AND boundary labels
// message elements are 8-bit codes: moved zero-extended in 4-byte exchange words
__global__ void gather_msg_kernel(const uint8_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n, uint32_t* __restrict__ dst) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[k] = src[idx[k]];
}
__global__ void scatter_msg_kernel(uint8_t* __restrict__ dst, const uint32_t* __restrict__ idx, uint64_t n, const uint32_t* __restrict__ src) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[idx[k]] = (uint8_t)src[k];
}
__global__ void gather2_kernel(const uint8_t* __restrict__ a, const uint32_t* __restrict__ b, const mvs_mrf_progress* __restrict__ st, uint32_t buf_stride, const uint32_t* __restrict__ idx, uint64_t n, uint32_t* __restrict__ dst) {
    b += (size_t)st->w * buf_stride;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = idx[k];
        dst[k] = (i & 0x80000000u) ? b[i & 0x7FFFFFFFu] : a[i];
    }
}
__global__ void scatter2_kernel(uint8_t* __restrict__ a, uint32_t* __restrict__ b, const mvs_mrf_progress* __restrict__ st, uint32_t buf_stride, const uint32_t* __restrict__ idx, uint64_t n, const uint32_t* __restrict__ src) {
    b += (size_t)st->w * buf_stride;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = idx[k];
        if (i & 0x80000000u) b[i & 0x7FFFFFFFu] = src[k]; else a[i] = (uint8_t)src[k];
    }
}
__global__ void counts_kernel(const uint32_t* __restrict__ col_ptr, uint32_t n, uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) counts[i] = col_ptr[i + 1] - col_ptr[i];
}
}  // namespace

static uint8_t* mrf_msg(mvs_ctx* ctx) { return ctx->m_msg_a.p; }  // one buffer, updated in place
// LAB = the current decode buffer (resolved on the device through the solver state), BEST_LAB = the best labeling's buffer
static uint32_t* mrf_array(mvs_ctx* ctx, int which, const mvs_mrf_progress** st) {
    *st = nullptr;
    switch (which) {
        case MVS_MRF_LAB: *st = ctx->m_state.p; return ctx->m_lab.p;
        case MVS_MRF_GAIN: return (uint32_t*)ctx->m_gain.p;
        case MVS_MRF_BEST_LAB: resolve_best(ctx); return ctx->b_lab;
    }
    throw StatusError(MVS_ERR_INVALID, "bad array selector");
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

mvs_status mvs_ctx_dc_get_max(mvs_ctx* ctx, float* dst) {
    if (!ctx || !dst) return api_fail(MVS_ERR_INVALID, "null argument");
    if (ctx->dc_phase < 1) return api_fail(MVS_ERR_STATE, "dc_phase1 first");
    MVS_API_BEGIN
    MVS_HIP(hipMemcpyAsync(dst, ctx->max_q.p, sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_API_END
}
mvs_status mvs_ctx_dc_set_max(mvs_ctx* ctx, const float* src) {
    if (!ctx || !src) return api_fail(MVS_ERR_INVALID, "null argument");
    if (ctx->dc_phase < 1) return api_fail(MVS_ERR_STATE, "dc_phase1 first");
    MVS_API_BEGIN
    MVS_HIP(hipMemcpyAsync(ctx->max_q.p, src, sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_API_END
}
mvs_status mvs_ctx_dc_get_histogram(mvs_ctx* ctx, uint32_t* dst) {
    if (!ctx || !dst) return api_fail(MVS_ERR_INVALID, "null argument");
    if (ctx->dc_phase < 2) return api_fail(MVS_ERR_STATE, "dc_phase2 first");
    MVS_API_BEGIN
    MVS_HIP(hipMemcpyAsync(dst, ctx->hist.p, MVS_HIST_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_API_END
}
mvs_status mvs_ctx_dc_set_histogram(mvs_ctx* ctx, const uint32_t* src) {
    if (!ctx || !src) return api_fail(MVS_ERR_INVALID, "null argument");
    if (ctx->dc_phase < 2) return api_fail(MVS_ERR_STATE, "dc_phase2 first");
    MVS_API_BEGIN
    MVS_HIP(hipMemcpyAsync(ctx->hist.p, src, MVS_HIST_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_API_END
}

mvs_status mvs_ctx_costs_export(mvs_ctx* ctx, uint32_t* counts, uint16_t* view_id, float* cost) {
    if (!ctx || !counts) return api_fail(MVS_ERR_INVALID, "null argument");
    if (!ctx->have_costs) return api_fail(MVS_ERR_STATE, "no data costs on the device");
    MVS_API_BEGIN
    const uint32_t F = ctx->csr_faces;
    if (F) { hipLaunchKernelGGL(counts_kernel, dim3((F + 255) / 256), dim3(256), 0, ctx->stream, ctx->r_ptr, F, counts); MVS_LAUNCH_CHECK(); }
    if (ctx->csr_nnz && view_id) MVS_HIP(hipMemcpyAsync(view_id, ctx->r_view, ctx->csr_nnz * sizeof(uint16_t), hipMemcpyDeviceToDevice, ctx->stream));
    if (ctx->csr_nnz && cost) MVS_HIP(hipMemcpyAsync(cost, ctx->r_cost, ctx->csr_nnz * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_API_END
}

mvs_status mvs_ctx_mrf_setup(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj, int adj_on_device, const mvs_mrf_params* params) {
    if (!ctx || !adj_ptr || !adj) return api_fail(MVS_ERR_INVALID, "null argument");
    if (!ctx->have_costs) return api_fail(MVS_ERR_STATE, "mrf setup needs data costs");
    MVS_API_BEGIN
    mvs_mrf_params P; if (params) P = *params; else mvs_mrf_default_params(&P);
    set_adjacency(ctx, adj_ptr, adj, adj_on_device, false);
    mrf_setup(ctx, &P);
    MVS_API_END
}

mvs_status mvs_ctx_mrf_sweep(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    if (!ctx || nb0 > ne0 || ne0 > ctx->csr_faces) return api_fail(MVS_ERR_INVALID, "bad node range");
    MVS_API_BEGIN
    Prof pr(ctx, "mrf_sweep");
    mrf_sweep(ctx, nb0, ne0);
    MVS_API_END
}

mvs_status mvs_ctx_mrf_sweep_phase(mvs_ctx* ctx, uint32_t phase, uint32_t nb0, uint32_t ne0) {
    if (!ctx || nb0 > ne0 || ne0 > ctx->csr_faces || phase >= ctx->m_colours) return api_fail(MVS_ERR_INVALID, "bad phase or node range");
    MVS_API_BEGIN
    Prof pr(ctx, "mrf_sweep");
    mrf_sweep_phase(ctx, phase, nb0, ne0);
    MVS_API_END
}
/* the set-up with BOUNDARY MARKS (marks_device[i] != 0: node i goes to the boundary zone of its colour class, see k_mrf.hip "zones") and one
 * zone of a colour phase: what csrc/shard.hip does per rank, for drivers / measuring scripts one level below it */
mvs_status mvs_ctx_mrf_setup_marked(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj, int adj_on_device, const mvs_mrf_params* params, const uint8_t* marks_device) {
    if (!ctx || !adj_ptr || !adj) return api_fail(MVS_ERR_INVALID, "null argument");
    if (!ctx->have_costs) return api_fail(MVS_ERR_STATE, "mrf setup needs data costs");
    MVS_API_BEGIN
    mvs_mrf_params P; if (params) P = *params; else mvs_mrf_default_params(&P);
    set_adjacency(ctx, adj_ptr, adj, adj_on_device, false);
    struct Marks { mvs_ctx* c; ~Marks() { c->m_bnd = nullptr; } } marks{ctx};
    ctx->m_bnd = marks_device;
    mrf_setup(ctx, &P);
    MVS_API_END
}
mvs_status mvs_ctx_mrf_sweep_phase_part(mvs_ctx* ctx, uint32_t phase, uint32_t nb0, uint32_t ne0, int part) {
    if (!ctx || nb0 > ne0 || ne0 > ctx->csr_faces || phase >= ctx->m_colours || part < 0 || part > 2) return api_fail(MVS_ERR_INVALID, "bad phase, node range or part");
    MVS_API_BEGIN
    Prof pr(ctx, part == MRF_PART_BOUNDARY ? "mrf_sweep_boundary" : "mrf_sweep");
    mrf_sweep_phase(ctx, phase, nb0, ne0, part);
    MVS_API_END
}
mvs_status mvs_ctx_mrf_layout(mvs_ctx* ctx, uint32_t* in_off_host, uint64_t n_edges) {
    if (!ctx || (n_edges && !in_off_host)) return api_fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    // MrfEdge = {in_off, out_off, kj}: strided copy of the first member
    if (n_edges) MVS_HIP(hipMemcpy2DAsync(in_off_host, sizeof(uint32_t), ctx->m_edge.p, sizeof(MrfEdge), sizeof(uint32_t), n_edges, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    MVS_API_END
}

mvs_status mvs_ctx_mrf_gather(mvs_ctx* ctx, int which, const uint32_t* idx, uint64_t n, void* dst) {
    if (!ctx || (n && (!idx || !dst))) return api_fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    if (n && which == MVS_MRF_MSG_LAB) {
        if (ctx->m_total >= 0x80000000ull) throw StatusError(MVS_ERR_UNSUPPORTED, "combined addressing needs < 2^31 message words");
        hipLaunchKernelGGL(gather2_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, mrf_msg(ctx), ctx->m_lab.p, ctx->m_state.p, ctx->m_stride, idx, n, (uint32_t*)dst);
        MVS_LAUNCH_CHECK();
    } else if (n && which == MVS_MRF_MSG) {
        hipLaunchKernelGGL(gather_msg_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, mrf_msg(ctx), idx, n, (uint32_t*)dst);
        MVS_LAUNCH_CHECK();
    } else if (n) {
        const mvs_mrf_progress* st; const uint32_t* arr = mrf_array(ctx, which, &st);
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, arr, st, ctx->m_stride, idx, n, (uint32_t*)dst);
        MVS_LAUNCH_CHECK();
    }
    MVS_API_END
}
mvs_status mvs_ctx_mrf_scatter(mvs_ctx* ctx, int which, const uint32_t* idx, uint64_t n, const void* src) {
    if (!ctx || (n && (!idx || !src))) return api_fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    if (n) { ctx->icm_dirty_valid = false; ctx->exact_valid = false; }   // labels changed behind the ICM active set
    if (n && which == MVS_MRF_MSG_LAB) {
        hipLaunchKernelGGL(scatter2_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, mrf_msg(ctx), ctx->m_lab.p, ctx->m_state.p, ctx->m_stride, idx, n, (const uint32_t*)src);
        MVS_LAUNCH_CHECK();
    } else if (n && which == MVS_MRF_MSG) {
        hipLaunchKernelGGL(scatter_msg_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, mrf_msg(ctx), idx, n, (const uint32_t*)src);
        MVS_LAUNCH_CHECK();
    } else if (n) {
        const mvs_mrf_progress* st; uint32_t* arr = mrf_array(ctx, which, &st);
        hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, arr, st, ctx->m_stride, idx, n, (const uint32_t*)src);
        MVS_LAUNCH_CHECK();
    }
    MVS_API_END
}

mvs_status mvs_ctx_mrf_energy(mvs_ctx* ctx, int which_sel, uint32_t nb0, uint32_t ne0, uint64_t* dst) {
    if (!ctx || !dst || nb0 > ne0 || ne0 > ctx->csr_faces) return api_fail(MVS_ERR_INVALID, "bad argument");
    if (which_sel != MVS_MRF_LAB && which_sel != MVS_MRF_BEST_LAB) return api_fail(MVS_ERR_INVALID, "energy: LAB or BEST_LAB");
    MVS_API_BEGIN
    mrf_energy(ctx, which_sel == MVS_MRF_BEST_LAB, nb0, ne0);
    MVS_HIP(hipMemcpyAsync(dst, ctx->m_energy.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_API_END
}

mvs_status mvs_ctx_mrf_keep_best(mvs_ctx* ctx) {
    if (!ctx) return api_fail(MVS_ERR_INVALID, "ctx is null");
    MVS_API_BEGIN
    mrf_keep_best(ctx);
    MVS_API_END
}

mvs_status mvs_ctx_mrf_step(mvs_ctx* ctx, const uint64_t* energy_device) {
    if (!ctx) return api_fail(MVS_ERR_INVALID, "ctx is null");
    MVS_API_BEGIN
    mrf_step(ctx, (const unsigned long long*)energy_device);
    MVS_API_END
}
mvs_status mvs_ctx_mrf_poll(mvs_ctx* ctx, uint32_t step, mvs_mrf_progress* out) {
    if (!ctx || !out) return api_fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    mrf_poll(ctx, step, out);
    MVS_API_END
}

mvs_status mvs_ctx_mrf_icm_gain(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    if (!ctx || nb0 > ne0 || ne0 > ctx->csr_faces) return api_fail(MVS_ERR_INVALID, "bad node range");
    MVS_API_BEGIN
    mrf_icm_gain(ctx, nb0, ne0);
    MVS_API_END
}
mvs_status mvs_ctx_mrf_icm_apply(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0, uint32_t* moved) {
    if (!ctx || !moved || nb0 > ne0 || ne0 > ctx->csr_faces) return api_fail(MVS_ERR_INVALID, "bad argument");
    MVS_API_BEGIN
    mrf_icm_apply(ctx, nb0, ne0);  // in place: winners form an independent set
    MVS_HIP(hipMemcpyAsync(moved, ctx->m_moved.p, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_API_END
}

mvs_status mvs_ctx_mrf_labels(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0, uint32_t* labels, uint32_t* unseen_out) {
    if (!ctx || !labels || nb0 > ne0 || ne0 > ctx->csr_faces) return api_fail(MVS_ERR_INVALID, "bad argument");
    MVS_API_BEGIN
    uint32_t bu[2];
    mrf_labels(ctx, nb0, ne0, labels, bu);
    if (unseen_out) *unseen_out = bu[1];
    if (bu[0]) throw StatusError(MVS_ERR_LABELING, "Incorrect labeling");
    MVS_API_END
}

}  // extern "C"
