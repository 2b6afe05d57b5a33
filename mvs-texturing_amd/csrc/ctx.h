// ctx.h -- context object, device buffers and launch helpers (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mvs_viewsel.h"
#include "dmath.h"

namespace mvs {

struct HipError : std::runtime_error {
    explicit HipError(const std::string& m) : std::runtime_error(m) {}
};
struct StatusError : std::runtime_error {
    mvs_status st;
    StatusError(mvs_status s, const std::string& m) : std::runtime_error(m), st(s) {}
};

#define MVS_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            throw mvs::HipError(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" +     \
                                __FILE__ + ":" + std::to_string(__LINE__) + ")");              \
    } while (0)

#define MVS_LAUNCH_CHECK() MVS_HIP(hipGetLastError())

// Growable device buffer; capacity persists across calls so that a steady-state
// step performs no hipMalloc.
template <class T>
struct DBuf {
    T* p = nullptr;
    size_t cap = 0;
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) MVS_HIP(hipFree(p));
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 64;
        MVS_HIP(hipMalloc((void**)&p, want * sizeof(T)));
        cap = want;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    ~DBuf() { release(); }
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
};

// Growable PINNED host buffer (device-to-host staging of the streamed table download, api.hip)
template <class T>
struct HBuf {
    T* p = nullptr;
    size_t cap = 0;
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) MVS_HIP(hipHostFree(p));
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 64;
        MVS_HIP(hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
    }
    ~HBuf() { if (p) (void)hipHostFree(p); }
    HBuf() = default;
    HBuf(const HBuf&) = delete;
    HBuf& operator=(const HBuf&) = delete;
};

// ---- implicit 4-ary BVH over Hilbert-sorted triangles (k_bvh.hip) ----
struct alignas(128) Node4 {
    float b[4][6];            // child c: lo.x lo.y lo.z hi.x hi.y hi.z -- child-major, so that (lo.x, lo.y) (lo.z, hi.x) (hi.y, hi.z) are
                              // the aligned SGPR pairs of three v_pk_fma_f32 per child; absent children are far-away points
    uint32_t nchild;
    uint32_t pad_[7];
    __host__ __device__ float lo(int a, int c) const { return b[c][a]; }
    __host__ __device__ float hi(int a, int c) const { return b[c][3 + a]; }
};
static_assert(sizeof(Node4) == 128, "Node4 must be one 128-byte line");

struct BvhDev {
    const Node4* nodes;       // all levels, level L at nodes + level_off[L]
    const float4* tris;       // 4 float4 per triangle: {a, lo.x} {e1 = b - a, lo.y} {e2 = c - a, lo.z} {hi, 0}  (curve order, zero padded)
    uint32_t level_off[16];
    uint32_t level_cnt[16];
    int32_t top;              // index of the root level (level_cnt[top] == 1)
    uint32_t n_leaves;
};

// Stage profiler: hipEvent pairs recorded on the context's stream (enabled by
// mvs_set_option("profile", 1)); read back with mvs_ctx_get_profile.
struct ProfSpan { std::string name; hipEvent_t a, b; bool owns_a; };   // owns_a = false: `a` is the previous span's `b` (ProfChain)

struct MrfEdge {      // per directed edge e = (i <- j) in adjacency-CSR order
    uint32_t in_off;  // offset of the message INTO i over e (K_i floats, aligned with i's labels)
    uint32_t out_off; // offset of the message i sends over e, i.e. in_off of the reverse edge (K_j floats)
    uint32_t kj;      // K_j if the edge is valid (both columns non-empty), else 0
};

// Fast-path (degree <= 3, every column <= 255 labels) per-node descriptor: 48 bytes = 3 x 16-byte loads, stored in
// (colour, id) order.  Offsets of message runs are multiples of 4 elements, which frees their low bits for flags.
struct alignas(16) NodeDesc {
    uint32_t rec;         // first 32-bit word of the node's RECORD in m_rec: ceil4(k) label words {cost code << 16 | view id},
                          // then, for every out-edge whose two label lists differ, ceil4(kj) one-byte re-alignment map entries
    uint32_t id;          // the node (face) this descriptor belongs to
    uint32_t in_off[3];   // incoming message offsets; bit 0: that neighbour has a LOWER colour (its label of this sweep is final when this node is swept)
    uint32_t out_off[3];  // outgoing message offsets (= in_off of the reverse edges); bit 0: identical label lists (map = identity, not stored)
    uint32_t kk;          // k | kj[0] << 8 | kj[1] << 16 | kj[2] << 24   (kj = 0: edge not in the model)
    uint32_t nbr[3];      // neighbour node ids (0xFFFFFFFF = none)
};
static_assert(sizeof(NodeDesc) == 48, "NodeDesc must be 48 bytes");

}  // namespace mvs

struct mvs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool verbose = false;
    bool profile = false;
    std::vector<mvs::ProfSpan> prof_spans; std::vector<hipEvent_t> prof_pool;
    bool prep_fused = true;  // image prep: luminance + Sobel in one pass through LDS (false: the two-pass kernels; identical output)
    bool stats = false;      // fill the cull-reason counters of mvs_dc_stats (diagnostics; costs atomics)
    bool count_rays = false;
    bool dc_overlap_prep = true;     // dc_phase1: image preparation on a second stream beside the face order + BVH build
    uint32_t* h_kd_flags = nullptr; int kd_pending = 0; bool kd_disabled = false;   // k_kdorder.hip: per-level overflow words (pinned), levels awaiting scene_order_commit, "this mesh keeps the curve order"
    uint32_t* h_rb = nullptr; uint32_t* d_rb = nullptr; uint32_t rb_seq = 0;   // read_words() / read_block(): 64 data words, the sequence number, 4 KB of staging -- pinned
    hipStream_t aux_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the image preparation runs beside the face order + BVH build (k_dc.hip dc_phase1)
    uint32_t bvh_upper_min_faces = 1000000;   // meshes below this many faces keep the Hilbert order above the LDS window (k_bvh.hip build_scene_order)
    uint32_t bvh_window = 262144;    // upper levels of the face order: exact top-down median cuts inside aligned windows of this many positions of the Hilbert order (k_kdorder.hip); 0 = the whole mesh, 1 = none
    mvs::DBuf<float> kd_c[2][3]; mvs::DBuf<uint32_t> kd_id[2], kd_hist, kd_cursor, kd_tie, kd_pivot, kd_box;   // k_kdorder.hip work buffers
    bool bvh_caller_order = false;   // experiment hook (with face_order = 0): the implicit BVH is built over the caller's face order as it is (tree-quality probes)
    int ray_xcd = 1;         // XCD-aware block order in the packet ray kernel
    int info_wave_area = 32;   // footprints (sampled ones) above this many pixels go to the wave-per-footprint kernel (k_dc.hip wave_info_kernel); 0 = every footprint serial = bit-exact with the reference's fp64 scan order
    int info_wave_area_words = 384;   // the same threshold where info_kernel walks a footprint four pixels per load as integers ("info_words": gradient term, no outlier removal): one lane
                                      // keeps up with the lane group up to a few hundred pixels (real-like scene, threshold 32 / 128 / 256 / 512 / 1024 / 2048: dc_face_info 1.38 / 1.13 / 1.04 / 1.04 / 1.17 / 1.39 ms)
    bool info_words = true;    // small footprints of the gradient term: integer word walk + certificate in info_kernel (option "info_words"; 0: serial fp64 walk)
    int info_cert_shift = 0;   // test hook: widens the exactness certificate of wave_info_kernel by this many bits (forces its serial fallback)
    uint32_t dc_stats_deferred = 0;
    int max_labels = 0;      // > 0: label-space compression after the data costs (k_dc.hip prune_write_kernel); 0 = the reference's model
    float cos_limit = 0.0f;  // see dmath.h cull_pair

    // ---- scene ----
    uint32_t n_verts = 0, n_faces = 0, n_views = 0;
    uint32_t face_begin = 0, face_end = 0;
    const float* d_verts = nullptr; const uint32_t* d_faces = nullptr; const float* d_normals = nullptr;
    mvs::DBuf<float> own_verts, own_normals; mvs::DBuf<uint32_t> own_faces;
    std::vector<mvs::ViewParams> h_views;
    mvs::DBuf<mvs::ViewParams> d_views;
    std::vector<mvs::DBuf<uint8_t>*> own_rgb;
    mvs::DBuf<uint8_t> gmi_all;      // all views' gradient-magnitude planes
    mvs::DBuf<uint8_t> lum_all;      // luminance planes (vectorised prep path)
    mvs::DBuf<uint32_t> mask_all;    // all views' bit-packed validity masks
    mvs::DBuf<uint32_t> mask_zero, mask_tmp;
    mvs::DBuf<uint32_t> msum_all;    // all views' mask tile summaries (dmath.h ViewParams::msum)
    std::vector<size_t> gmi_off, mask_off, msum_off;
    mvs::DBuf<size_t> view_off;
    bool mesh_dirty = true, views_dirty = true;

    // ---- the library's own mesh layout (k_bvh.hip build_scene_order; DESIGN.md "Face order") ----
    // The caller's mesh arrives in file order (calculate_data_costs.cpp:136-138); nothing about it is assumed.  Every data-cost pass
    // first lays the mesh out along a Hilbert curve: vertices by their position, faces by their centroid (refined by median splits
    // inside 512-face windows: the BVH's leaf order).  i_verts / i_faces / i_normals are that copy -- vertex s of the copy is the
    // s-th vertex of the curve, face p the p-th face -- and everything downstream (culls, rays, footprints, the cost table, the
    // solver's nodes, the parts of the sharded path) is indexed by those positions.  f_perm[p] = the caller's id of face p, f_pos its
    // inverse; results cross the ABI in the caller's numbering.  Option "face_order" = 0 keeps the caller's face numbering (a caller
    // that laid the faces out itself; the building-block API of mgpu.hip); vertices are renumbered either way (invisible outside).
    int face_order = 1;
    mvs::DBuf<float> i_verts, i_normals; mvs::DBuf<uint32_t> i_faces, f_perm, f_pos, f_tmp;
    const float* iv = nullptr; const uint32_t* ifc = nullptr; const float* inr = nullptr;   // the arrays the kernels read (i_* or the caller's)
    bool mesh_ordered = false;        // f_perm / f_pos describe the resident mesh (set by build_scene_order when face_order != 0)
    bool order_pinned = false;        // a shard was made of this mesh (mvs_shard_create): its parts, its renumbered adjacency and its halo plan ARE this
                                      // layout, so the data-cost passes of the sharded path keep it instead of deriving it again (mvs_scene_set_mesh un-pins)
    // order of the ACTIVE cost table (r_ptr ...): t_perm[p] = caller's id of column p (null: the table is in the caller's order)
    const uint32_t* t_perm = nullptr; const uint32_t* t_pos = nullptr;
    mvs::DBuf<uint32_t> u_ptr, u_cnt; mvs::DBuf<uint16_t> u_view; mvs::DBuf<float> u_cost, u_q;   // the table in the caller's order (built on demand)
    bool u_valid = false, u_q_valid = false;
    mvs::DBuf<unsigned long long> fp_acc;   // device-side fingerprint of the table handed out (api.hip mvs_data_costs_stream)
    mvs::HBuf<uint32_t> stage_ptr; mvs::HBuf<uint16_t> stage_view[2]; mvs::HBuf<float> stage_cost[2];   // pinned staging of the chunked download

    // ---- BVH + incidence ----
    mvs::DBuf<mvs::Node4> bvh_nodes; mvs::DBuf<float4> bvh_tris;
    mvs::DBuf<uint32_t> sort_k, sort_k2, sort_v, sort_v2; mvs::DBuf<char> sort_tmp;
    mvs::DBuf<float> lvl_box_a, lvl_box_b; mvs::DBuf<float> scene_box;
    mvs::BvhDev bvh{};
    mvs::DBuf<uint32_t> vf_ptr, vf_cursor, vf;
    mvs::DBuf<uint32_t> vperm, vpos;   // vertices in Hilbert order and the inverse map (caller's vertex ids; used to build i_verts / i_faces only)
    const uint32_t* tri_order = nullptr;   // BVH triangle slot -> face position (null: identity, the faces ARE in curve order)

    // ---- data costs work buffers ----
    mvs::DBuf<unsigned long long> pass_bits, need_bits, occl_bits, surv_bits;
    mvs::DBuf<uint32_t> pass_base;      // exclusive scan of popc(pass words)
    mvs::DBuf<uint32_t> pass_face;      // [view chunk][face] the chunk's pass bits of a face (face-major copy read by need_kernel)
    mvs::DBuf<unsigned long long> defer_bits; mvs::DBuf<uint32_t> defer_base; mvs::DBuf<uint2> defer_list; mvs::DBuf<uint32_t> rewalk_list;   // large footprints left to the wave-per-footprint kernel
    mvs::DBuf<float> pq;                // quality per passing pair
    mvs::DBuf<float> pcol;              // 3 floats per passing pair (outlier removal only)
    mvs::DBuf<uint32_t> face_cnt, scan_tmp;
    mvs::DBuf<unsigned long long> counters;  // see k_dc.hip
    mvs::DBuf<float> max_q; mvs::DBuf<uint32_t> hist; mvs::DBuf<float> pctl;
    // pre-outlier CSR (only when outlier removal is on)
    mvs::DBuf<uint32_t> pre_ptr; mvs::DBuf<uint16_t> pre_view; mvs::DBuf<float> pre_q; mvs::DBuf<float> pre_col;
    mvs::DBuf<uint8_t> pre_inl;
    // result CSR
    mvs::DBuf<uint32_t> csr_ptr; mvs::DBuf<uint16_t> csr_view; mvs::DBuf<float> csr_cost; mvs::DBuf<float> csr_q;
    uint32_t csr_faces = 0, csr_views = 0; uint64_t csr_nnz = 0;
    const uint32_t* r_ptr = nullptr; const uint16_t* r_view = nullptr; const float* r_cost = nullptr;  // active CSR
    bool csr_q_valid = false;   // csr_q holds the qualities of the active CSR (set by the data-cost stage)
    bool have_costs = false;
    mvs_settings dc_settings{}; mvs_dc_stats dc_stats{}; int dc_phase = 0;

    // ---- row f1: mesh preparation + adjacency graph (k_mesh.hip) ----
    mvs::DBuf<unsigned long long> g_keys, g_keys2; mvs::DBuf<uint32_t> g_vals, g_vals2, g_pos, g_cnt, g_adj_ptr, g_adj, g_faces; mvs::DBuf<float> g_normals;
    uint64_t g_adj_entries = 0; bool have_adj = false;

    // ---- row f3: patch components (k_patch.hip) ----
    mvs::DBuf<uint32_t> p_label_ptr, p_comp_ptr, p_comp_faces, p_parent, p_root, p_state, p_flag, p_pos, p_roots, p_roots2, p_rlab, p_rlab2, p_adj_ptr, p_adj, p_labels;

    // ---- region moves (k_region.hip) ----
    mvs::DBuf<uint32_t> rg_parent, rg_root, rg_size, rg_bestl, rg_lose, rg_flag, rg_pos, rg_cstart, rg_have, rg_cfirst, rg_name;
    mvs::DBuf<unsigned long long> rg_gain, rg_cur, rg_key, rg_key2, rg_ck, rg_sum; mvs::DBuf<long long> rg_cgain; mvs::DBuf<uint2> rg_cut;

    // ---- MRF ----
    mvs::DBuf<uint32_t> m_adj_ptr, m_adj, a_stage_ptr, a_stage; const uint32_t* r_adj_ptr = nullptr; const uint32_t* r_adj = nullptr;   // a_stage*: host lists on their way into the table's order
    uint32_t r_adj_edges = 0; bool r_adj_edges_known = false;   // length of r_adj where set_adjacency learned it (host lists, renumbered lists)
    mvs::DBuf<mvs::NodeDesc> m_desc; mvs::DBuf<uint8_t> m_ident; mvs::DBuf<uint32_t> m_rec; uint64_t m_rec_words = 0; bool m_fast = false; int mrf_blocks_per_cu = 0 /* 0 = resident count from the occupancy API */, mrf_xcd = 1, mrf_late_old = 1, mrf_run_pad = 4;
    mvs::DBuf<mvs::MrfEdge> m_edge; mvs::DBuf<uint32_t> m_size, m_rev /* reverse directed edge of every in-edge */; mvs::DBuf<uint16_t> m_map;
    mvs::DBuf<uint8_t> m_msg_a;    // messages as 8-bit fixed point over [0, 1/rho], updated in place (one colour class at a time)
    // decode of a sweep = position in the column (sel), label (view + 1) and the unary of that label as the sweeps see it.
    // TWO buffers of F + 1 entries each: the sweeps write buffer m_state->w, the best labeling so far is buffer
    // m_state->best_w; "keep the best" is a flip of those two indices by the step kernel, never a copy.
    mvs::DBuf<uint32_t> m_sel, m_sel2, m_cand; mvs::DBuf<float> m_gain;
    mvs::DBuf<uint32_t> m_lab; mvs::DBuf<float> m_cost;
    uint32_t m_stride = 0;      // F + 1: offset of the second buffer
    bool exact_valid = false; uint32_t exact_nb = 0, exact_ne = 0;   // b_sel / b_cost of nodes [exact_nb, exact_ne) are derived from b_lab (k_mrf.hip mrf_exact_costs)
    uint32_t* b_sel = nullptr; uint32_t* b_lab = nullptr; float* b_cost = nullptr; bool best_resolved = false;   // the best buffer, once the host knows which one it is (resolve_best)
    mvs::DBuf<unsigned long long> m_energy; mvs::DBuf<uint32_t> m_moved; mvs::DBuf<uint32_t> m_alist; bool icm_dirty_valid = false;   // ICM active set: nodes whose gain the next pass re-evaluates
    uint32_t m_n_adj = 0;      // directed edges of the adjacency given to mrf_setup
    uint32_t m_energy_blocks = 0;   // per-block partial pairs the last mrf_energy left behind m_energy.p + 4
    bool m_energy_from_sweep = false;   // ... or the last sweep's own kernels did (fast path: the energy is accumulated while sweeping)
    uint64_t m_total = 0; uint32_t m_kmax = 0, m_degmax = 0;
    // colour-phased schedule: colours of the adjacency graph and per-node classes (k_mrf.hip mrf_node_class); nodes in SCHEDULE order =
    // sorted by sub-class key (fast nodes by (colour, class, id), then generic nodes by (colour, id)); m_sub_begin[key] = first position
    // of a key >= `key` (host copy of m_sub); positions [0, m_n_fast) are the fast nodes
    mvs::DBuf<uint32_t> m_colour, m_perm, m_tmp_a, m_tmp_b, m_tmp_c, m_sub; mvs::DBuf<uint8_t> m_cls; uint32_t m_colours = 0, m_n_fast = 0; std::vector<uint32_t> m_sub_begin;
    const uint32_t* m_colour_in = nullptr;   // colours of all nodes kept by a sharded caller (null: mrf_setup colours the graph)
    const uint8_t* m_bnd = nullptr;   // per node: 1 = boundary node of a sharded caller (own node with an edge into another rank's part) -> zone 0 of the schedule (k_mrf.hip); null: no marks
    int mrf_damp_period = 4;     // damped sweeps: 1, 1 + p, 1 + 2p, ... (4: the solver's definition, restated in the oracle; other values are experiment knobs)
    int mrf_force_generic = 0;   // test hook: every node takes the generic sweep kernel
    int mrf_wide = 1;            // class-1 nodes (neighbourhood columns of 33 .. 64 labels) through mrf_sweep8_kernel: 8 lanes x 8 labels (k_mrf.hip); 0: mrf_sweep4_kernel<16>
    bool m_wide_layout = false;  // ... as the last mrf_setup laid the records and runs out
    uint32_t m_range_nb = 0, m_range_ne = 0; std::vector<uint32_t> m_range_q;   // cached own share of every sub-class
    uint32_t m_sweep_no = 0;   // sweeps started since mrf_setup (1-based inside a sweep): sweeps 1, 5, 9, ... are damped
    mvs_mrf_params m_params{};
    // device-side stop rule (k_mrf.hip mrf_step): solver state in HBM, per-step reports through a pinned ring
    static constexpr uint32_t RING = 16;
    static constexpr uint32_t ICM_RING = 8;
    uint32_t* h_icm = nullptr; uint32_t* d_icm = nullptr; uint32_t icm_seq = 0;   // pinned "moved" counts of the ICM rounds, read a few rounds late
    mvs::DBuf<mvs_mrf_progress> m_state; mvs::DBuf<unsigned long long> m_hist; mvs::DBuf<uint32_t> m_ctl;   // m_ctl: {steps of the solve, sequence base} read by the step kernel
    // the sweep loop as a replayed hipGraph (api.hip): two sweeps (a damped and an undamped one) + their bookkeeping steps, captured on a
    // private stream, launched on the context's stream; re-captured per solve and pushed into the instantiated graph with hipGraphExecUpdate
    int mrf_graph = 1; hipStream_t cap_stream = nullptr; hipGraphExec_t sweep_exec = nullptr; uint32_t graph_launches = 0, graph_updates = 0, graph_instantiations = 0;
    mvs_mrf_progress* h_ring = nullptr; mvs_mrf_progress* d_ring = nullptr /* the same pinned slots as the device addresses them */; uint32_t steps_issued = 0; int mrf_lag = 1;
    int shard_peer_push = 1;   // sharded sweep loop: boundary runs stored straight into the peers' arrays where the communicator allows it (shard.hip PeerHub)
    // arrival of a report = its sequence number in the pinned word next to it (written after a system-scope fence): the host polls
    // memory, no event is recorded in the stream.  Sequence numbers never repeat within a context.
    uint32_t* h_seq = nullptr; uint32_t* d_seq = nullptr; uint32_t seq_base = 0;   // [0, RING): solver steps; [RING, RING + ICM_RING): ICM rounds
};

namespace mvs {
// RAII stage marker; no-op unless ctx->profile
struct Prof {
    mvs_ctx* c; size_t idx; bool on;
    Prof(mvs_ctx* ctx, const char* name) : c(ctx), idx(0), on(ctx->profile) {
        if (!on) return;
        auto get = [&]() { hipEvent_t e; if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); } else { MVS_HIP(hipEventCreate(&e)); } return e; };
        ProfSpan sp{name, get(), get(), true};
        MVS_HIP(hipEventRecord(sp.a, c->stream));
        idx = c->prof_spans.size(); c->prof_spans.push_back(sp);
    }
    void end() { if (on) { MVS_HIP(hipEventRecord(c->prof_spans[idx].b, c->stream)); on = false; } }
    ~Prof() { if (on) (void)hipEventRecord(c->prof_spans[idx].b, c->stream); }
};
// Adjacent spans that share their boundary events: one hipEventRecord per mark instead of two per span (an event record costs
// ~5 us of stream time; the solver loop alternates two short stages 42 times).
struct ProfChain {
    mvs_ctx* c; hipEvent_t prev; bool on;
    static hipEvent_t get(mvs_ctx* c) { hipEvent_t e; if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); } else { MVS_HIP(hipEventCreate(&e)); } return e; }
    explicit ProfChain(mvs_ctx* ctx) : c(ctx), prev(nullptr), on(ctx->profile) {}
    void begin() { if (on && !prev) { prev = get(c); MVS_HIP(hipEventRecord(prev, c->stream)); first = true; } }
    void mark(const char* name) {       // closes the span [previous mark, now)
        if (!on) return;
        begin();
        hipEvent_t e = get(c);
        MVS_HIP(hipEventRecord(e, c->stream));
        c->prof_spans.push_back(ProfSpan{name, prev, e, first});
        first = false; prev = e;
    }
    bool first = false;
};
// roctx range around the two windows the reference times (apps/texrecon/texrecon.cpp:118 "Calculating data costs", :126 "Running MRF
// optimization"): visible to rocprofv3 --marker-trace; libroctx64 is resolved at run time (no link-time dependency), absent = no-op
struct RoctxRange { explicit RoctxRange(const char* name); ~RoctxRange(); bool on; };
// device for the one-shot host entry points (mvs_data_costs, mvs_view_selection, ...): environment MVS_DEVICE, default 0
int default_device();
// which zone(s) of a colour phase mrf_sweep_phase sweeps (k_mrf.hip)
enum { MRF_PART_ALL = 0, MRF_PART_BOUNDARY = 1, MRF_PART_INTERIOR = 2 };
// reports through pinned host memory (k_mrf.hip)
void ensure_report_ring(mvs_ctx* ctx);
void report_u32(mvs_ctx* ctx, const uint32_t* d_src, uint32_t* d_dst, uint32_t seq_slot, uint32_t seq);
void wait_report(mvs_ctx* ctx, uint32_t seq_slot, uint32_t seq);
// up to 64 words from device memory to the host between two launches of ctx->stream: a one-block kernel stores them into a pinned buffer and
// announces them with a sequence number the host spins on (k_mrf.hip).  A 4-byte hipMemcpyAsync into pageable memory + hipStreamSynchronize
// holds the device idle for 22 us, this for 9 (scripts/probe/readback_cost.hip) -- a step has a dozen of them on its critical path.
void read_words(mvs_ctx* ctx, const void* d_src, void* out, uint32_t n_words);
void read_block(mvs_ctx* ctx, const void* d_src, void* out, size_t bytes);   // up to 4 KB, through pinned staging
// generic device exclusive scan (scan.hip): out[i] = sum_{k<i} in[i]; returns total via d_total (device, may be null)
void exclusive_scan_u32(mvs_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t* d_total);
// exact 64-bit total of a u32 array (blocking): the guard in front of scans whose total may pass 2^32
uint64_t sum_u32(mvs_ctx* ctx, const uint32_t* in, size_t n);
}  // namespace mvs
