// k_mrf.hip -- tex::view_selection on the GPU (libs/tex/view_selection.cpp:18-133).
//
// Model (restated from :27-82): node i = face, label set = {view_id + 1 : view_id
// in column i} with unaries = column costs, or the single label 0 when the column
// is empty; Potts edges of weight 1 between adjacent faces whose columns are both
// non-empty.  mapMAP (:93-118) is replaced by a GPU-resident tree-reweighted
// max-product solver -- colour-phased Gauss-Seidel sweeps (the adjacency graph is
// coloured, each colour class updates its messages in place in turn) + monotone
// ICM polish; the algorithm is specified to the float operation in DESIGN.md
// "MRF solver" and restated for the CPU in oracle/oracle.cpp -- labels are
// bit-identical by construction: a colour class is an independent set, min /
// argmin reductions are exact in any order, sums follow adjacency order,
// energies are 32.32 fixed-point integers.
//
// Work mapping: G lanes per node (G = 8 / 16 / 32 / 64 from the largest column), 4
// consecutive labels per lane; messages are 8-bit fixed point in HBM (four per 4-byte
// word), damped on odd sweeps only; per-edge cavity vectors go through an LDS tile for
// the label re-alignment gather; min / argmin are fused DPP butterflies.
#include "ctx.h"
#include <rocprim/rocprim.hpp>

namespace mvs {

namespace {

constexpr uint16_t MAP_NONE = 0xFFFF;
constexpr uint32_t MSG_BASE = MVS_MRF_MSG_BASE;   // first real message / map element (mvs_viewsel.h)

// Messages live in HBM as 8-bit fixed point over their range [0, lam], lam = 1 / rho (a message is a truncated,
// min-normalised cavity: 0 <= m <= lam by construction); arithmetic is fp32.  code = trunc(m * (255 / lam) + 0.5),
// value = code * (lam / 255): a quarter of the bytes of an fp32 layout, half of binary16, at the same solution
// quality (part of the solver's definition, restated in oracle/oracle.cpp).
typedef uint8_t msg_t;
struct MsgQ { float scale, step; };      // 255 / lam and lam / 255, fp32, the oracle computes them the same way
__device__ __forceinline__ MsgQ msg_q(float lam) { return MsgQ{255.0f / lam, lam / 255.0f}; }
__device__ __forceinline__ float msg_load(const msg_t* __restrict__ p, size_t i, MsgQ q) { return (float)p[i] * q.step; }
__device__ __forceinline__ uint32_t msg_code(float v, MsgQ q) { return (uint32_t)(v * q.scale + 0.5f); }   // v in [0, lam]: 0 .. 255

__device__ __forceinline__ unsigned long long fix32(float d) { return (unsigned long long)((double)d * 4294967296.0); }

// Butterfly partner exchange inside a lane group with DPP (VALU latency) instead of ds_bpermute (LDS
// latency): the pairings xor 1, xor 2 (quad_perm), i <-> 7-i (row_half_mirror) and i <-> 15-i (row_mirror)
// are involutions that merge groups of 2, 4, 8 and 16 lanes, which is all an all-reduce needs (min and
// argmin are exact in any order); only the 32- and 64-lane steps go through a shuffle.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
template <int STEP, int G>
__device__ __forceinline__ float partner_f(float v) {
    if (STEP == 1) return dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    else if (STEP == 2) return dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    else if (STEP == 4) return dpp_f<0x141>(v);  // row_half_mirror
    else if (STEP == 8) return dpp_f<0x140>(v);  // row_mirror
    else return __shfl_xor(v, STEP, G);
}
template <int STEP, int G>
__device__ __forceinline__ uint32_t partner_u(uint32_t v) {
    if (STEP == 1) return dpp_u<0xB1>(v);
    else if (STEP == 2) return dpp_u<0x4E>(v);
    else if (STEP == 4) return dpp_u<0x141>(v);
    else if (STEP == 8) return dpp_u<0x140>(v);
    else return __shfl_xor(v, STEP, G);
}
template <int G>
__device__ __forceinline__ float group_min(float v) {
    v = fminf(v, partner_f<1, G>(v)); v = fminf(v, partner_f<2, G>(v)); v = fminf(v, partner_f<4, G>(v));
    if (G >= 16) v = fminf(v, partner_f<8, G>(v));
    if (G >= 32) v = fminf(v, partner_f<16, G>(v));
    if (G >= 64) v = fminf(v, partner_f<32, G>(v));
    return v;
}
// first argmin: smallest value, ties -> smallest index
template <int STEP, int G>
__device__ __forceinline__ void argmin_step(float& bb, uint32_t& bt) {
    const float ob = partner_f<STEP, G>(bb); const uint32_t ot = partner_u<STEP, G>(bt);
    const bool take = ob < bb || (ob == bb && ot < bt);
    bb = take ? ob : bb; bt = take ? ot : bt;
}
template <int G>
__device__ __forceinline__ void group_argmin(float& bb, uint32_t& bt) {
    argmin_step<1, G>(bb, bt); argmin_step<2, G>(bb, bt); argmin_step<4, G>(bb, bt);
    if (G >= 16) argmin_step<8, G>(bb, bt);
    if (G >= 32) argmin_step<16, G>(bb, bt);
    if (G >= 64) argmin_step<32, G>(bb, bt);
}

// Fused forms for the hot sweep: v_min_f32 / v_min_u32 with the DPP modifier on src0, i.e. ONE VALU instruction per
// butterfly step where the compiler's v_mov_dpp + canonicalise + min takes three (plus the exec-mask branches it
// builds for a short-circuit argmin).  s_nop 1 = the two wait states a DPP read needs after a VALU write of the
// same register (the hazard recogniser does not look inside inline asm).  All lanes are active where these run.
#define MVS_DPP_OP(op, ctrl) asm("s_nop 1\n\t" op " %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v))
template <int STEP, int G>
__device__ __forceinline__ float min_step_f(float v) {
    float r;
    if (STEP == 1) MVS_DPP_OP("v_min_f32_dpp", "quad_perm:[1,0,3,2]");
    else if (STEP == 2) MVS_DPP_OP("v_min_f32_dpp", "quad_perm:[2,3,0,1]");
    else if (STEP == 4) MVS_DPP_OP("v_min_f32_dpp", "row_half_mirror");
    else if (STEP == 8) MVS_DPP_OP("v_min_f32_dpp", "row_mirror");
    else r = fminf(v, __shfl_xor(v, STEP, G));
    return r;
}
template <int STEP, int G>
__device__ __forceinline__ uint32_t min_step_u(uint32_t v) {
    uint32_t r;
    if (STEP == 1) MVS_DPP_OP("v_min_u32_dpp", "quad_perm:[1,0,3,2]");
    else if (STEP == 2) MVS_DPP_OP("v_min_u32_dpp", "quad_perm:[2,3,0,1]");
    else if (STEP == 4) MVS_DPP_OP("v_min_u32_dpp", "row_half_mirror");
    else if (STEP == 8) MVS_DPP_OP("v_min_u32_dpp", "row_mirror");
    else r = min(v, (uint32_t)__shfl_xor(v, STEP, G));
    return r;
}
#undef MVS_DPP_OP
template <int G>
__device__ __forceinline__ float group_min_fused(float v) {
    v = min_step_f<1, G>(v); v = min_step_f<2, G>(v); v = min_step_f<4, G>(v);
    if (G >= 16) v = min_step_f<8, G>(v);
    if (G >= 32) v = min_step_f<16, G>(v);
    if (G >= 64) v = min_step_f<32, G>(v);
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_min_fused(uint32_t v) {
    v = min_step_u<1, G>(v); v = min_step_u<2, G>(v); v = min_step_u<4, G>(v);
    if (G >= 16) v = min_step_u<8, G>(v);
    if (G >= 32) v = min_step_u<16, G>(v);
    if (G >= 64) v = min_step_u<32, G>(v);
    return v;
}

// ---- setup ----
__global__ void mrf_size_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                uint32_t F, uint32_t pad_mask, uint32_t* __restrict__ size, uint32_t* __restrict__ maxes /* [0]=kmax [1]=degmax */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = 0, deg = 0;
    if (i < F) {
        k = col_ptr[i + 1] - col_ptr[i];
        const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
        deg = e1 - e0;
        for (uint32_t e = e0; e < e1; ++e) {
            const uint32_t j = adj[e];
            const uint32_t kj = col_ptr[j + 1] - col_ptr[j];
            size[e] = (k > 0 && kj > 0) ? ((k + pad_mask) & ~pad_mask) : 0u;   // runs padded to a multiple of 4 (8-byte quads) or 16 elements (32-byte sectors)
        }
    }
    for (int o = 32; o > 0; o >>= 1) { k = max(k, (uint32_t)__shfl_xor(k, o, 64)); deg = max(deg, (uint32_t)__shfl_xor(deg, o, 64)); }
    // same-address atomics serialise (~12 ns each): only waves that can still raise a maximum issue one
    if ((threadIdx.x & 63) == 0) {
        if (k > __hip_atomic_load(&maxes[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxes[0], k);
        if (deg > __hip_atomic_load(&maxes[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxes[1], deg);
    }
}

// ---- colour-phased schedule ----
// Greedy colouring of the adjacency graph in the order of the keys (hash32(i), i) -- the oracle's mrf_colour -- by
// Jones-Plassmann rounds: a node colours itself (smallest colour no smaller-key neighbour holds) as soon as all its
// smaller-key neighbours are coloured.  Larger-key neighbours are necessarily still uncoloured at that moment, so the
// result equals the sequential greedy colouring whatever the interleaving; only the number of rounds varies.
__device__ __forceinline__ uint32_t mrf_hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ bool mrf_key_less(uint32_t a, uint32_t b) { const uint32_t ha = mrf_hash32(a), hb = mrf_hash32(b); return ha != hb ? ha < hb : a < b; }
constexpr uint32_t NO_COLOUR = 0xFFFFFFFFu;
__global__ void mrf_colour_init_kernel(uint32_t* __restrict__ colour, uint32_t* __restrict__ iota, uint32_t F) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < F) { colour[i] = NO_COLOUR; iota[i] = i; }
}
__global__ void mrf_colour_round_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, uint32_t F,
                                        uint32_t* colour, uint32_t* __restrict__ pending) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    if (__hip_atomic_load(colour + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != NO_COLOUR) return;
    unsigned long long used = 0ull;
    bool ready = true;
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        const uint32_t j = adj[e];
        if (j == i || !mrf_key_less(j, i)) continue;
        const uint32_t cj = __hip_atomic_load(colour + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cj == NO_COLOUR) { ready = false; break; }
        used |= 1ull << cj;
    }
    if (ready) __hip_atomic_store(colour + i, (uint32_t)__builtin_ctzll(~used), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *pending = 1u;                                       // racing stores of the same value
}
// colour_begin[c] = first position of colour >= c in the sorted colour array, c = 0 .. 64
__global__ void mrf_colour_begin_kernel(const uint32_t* __restrict__ sorted, uint32_t F, uint32_t* __restrict__ colour_begin) {
    const uint32_t c = threadIdx.x;
    if (c > 64u) return;
    uint32_t lo = 0, hi = F;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted[mid] < c) lo = mid + 1; else hi = mid; }
    colour_begin[c] = lo;
}
// own share of every colour class: positions of the ids in [nb, ne) inside perm[cb[c], cb[c + 1]) (ids ascending)
__global__ void mrf_phase_range_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ colour_begin, uint32_t C,
                                       uint32_t nb, uint32_t ne, uint32_t* __restrict__ out) {
    const uint32_t c = threadIdx.x;
    if (c >= C) return;
    const uint32_t cb = colour_begin[c], ce = colour_begin[c + 1];
    uint32_t lo = cb, hi = ce;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (perm[mid] < nb) lo = mid + 1; else hi = mid; }
    out[2 * c] = lo;
    hi = ce;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (perm[mid] < ne) lo = mid + 1; else hi = mid; }
    out[2 * c + 1] = lo;
}
// Message layout, SENDER-major: the runs a node WRITES (one per out-edge, in its list order, each as long as the
// receiver's padded label count) are contiguous, and nodes follow each other in (colour, id) order.  A colour phase
// therefore streams its previous-outgoing reads and its stores through one contiguous region (2 of the 3 message
// accesses per label); only the incoming reads gather 1 run out of each neighbour's block.
// size[e] belongs to the in-edge e = (i <- j) of its receiver i; the out-edge r = (j -> i) of j owns the same run.
__device__ __forceinline__ uint32_t mrf_reverse_edge(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, uint32_t from, uint32_t to) {
    uint32_t r = adj_ptr[to];
    const uint32_t r1 = adj_ptr[to + 1];
    while (r < r1 && adj[r] != from) ++r;
    return r < r1 ? r : 0xFFFFFFFFu;                          // position of `from` in the list of `to`
}
// nsz[b * (F + 1) + q] = message elements node perm[q] sends to receivers of colour b (b < n_col; with n_col == 1 all
// receivers count as colour 0 = plain sender-major).  One exclusive scan over the n_col * (F + 1) entries then yields the
// layout  [receiver colour][sender in (colour, id) order][out-edge in list order]:
//   * what a phase WRITES (and re-reads for damping) is one contiguous stream per receiver colour;
//   * what a phase READS as incoming messages is the whole super-region of its own colour -- every byte of it is consumed
//     in that phase, by receivers that follow each other roughly in memory order, so no fetched line is wasted (a plain
//     sender-major layout gathers one 90-byte run out of each neighbour's 270-byte block: 2x the bytes at C3).
constexpr int MAX_LAYOUT_COLOURS = 8;
__global__ void mrf_nodesize_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ colour, const uint32_t* __restrict__ adj_ptr,
                                    const uint32_t* __restrict__ adj, const uint32_t* __restrict__ size, uint32_t F, uint32_t n_col,
                                    uint32_t* __restrict__ nsz) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > F) return;
    uint32_t acc[MAX_LAYOUT_COLOURS];
#pragma unroll
    for (int b = 0; b < MAX_LAYOUT_COLOURS; ++b) acc[b] = 0;
    if (q < F) {
        const uint32_t j = perm[q];
        for (uint32_t r = adj_ptr[j]; r < adj_ptr[j + 1]; ++r) {
            const uint32_t i = adj[r], e = mrf_reverse_edge(adj_ptr, adj, j, i);
            if (e == 0xFFFFFFFFu) continue;
            const uint32_t b = (n_col > 1) ? colour[i] : 0u, sz = size[e];
#pragma unroll
            for (int c = 0; c < MAX_LAYOUT_COLOURS; ++c) acc[c] += ((uint32_t)c == b) ? sz : 0u;
        }
    }
#pragma unroll
    for (int b = 0; b < MAX_LAYOUT_COLOURS; ++b) if ((uint32_t)b < n_col) nsz[(size_t)b * (F + 1) + q] = acc[b];
}
__global__ void mrf_inoff_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ colour, const uint32_t* __restrict__ adj_ptr,
                                 const uint32_t* __restrict__ adj, const uint32_t* __restrict__ size, const uint32_t* __restrict__ noff,
                                 uint32_t F, uint32_t n_col, uint32_t* __restrict__ in_off) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= F) return;
    const uint32_t j = perm[q];
    uint32_t off[MAX_LAYOUT_COLOURS];
#pragma unroll
    for (int b = 0; b < MAX_LAYOUT_COLOURS; ++b) off[b] = ((uint32_t)b < n_col) ? noff[(size_t)b * (F + 1) + q] : 0u;
    for (uint32_t r = adj_ptr[j]; r < adj_ptr[j + 1]; ++r) {
        const uint32_t i = adj[r], e = mrf_reverse_edge(adj_ptr, adj, j, i);
        if (e == 0xFFFFFFFFu) continue;
        const uint32_t b = (n_col > 1) ? colour[i] : 0u, sz = size[e];
        uint32_t o = 0;
#pragma unroll
        for (int c = 0; c < MAX_LAYOUT_COLOURS; ++c) { if ((uint32_t)c == b) { o = off[c]; off[c] += sz; } }
        in_off[e] = o;
    }
}

__global__ void mrf_edge_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                uint32_t F, const uint32_t* __restrict__ in_off, const uint32_t* __restrict__ size, MrfEdge* __restrict__ edge) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        const uint32_t j = adj[e];
        uint32_t r = adj_ptr[j];
        const uint32_t r1 = adj_ptr[j + 1];
        while (r < r1 && adj[r] != i) ++r;
        MrfEdge m;
        m.in_off = MSG_BASE + in_off[e];                       // [0, MSG_BASE) is the reserved zero / identity run
        m.out_off = (r < r1) ? MSG_BASE + in_off[r] : 0u;
        m.kj = (size[e] > 0 && r < r1) ? (col_ptr[j + 1] - col_ptr[j]) : 0u;
        edge[e] = m;
    }
}

__global__ void mrf_identity_kernel(uint16_t* __restrict__ map) { map[threadIdx.x] = (uint16_t)threadIdx.x; }

// map[in_off(e) + t] = position of L_i[t] in L_j (binary search; lists ascending, calculate_data_costs.cpp:272)
// ident[e] = 1 iff the two label lists of edge e are identical (map == identity), i.e. the sender's out-edge rev(e) can skip it.
// With skip_ident (fast sweep path: identical-list edges read the reserved identity run instead) the map of such an
// edge is not even written -- three quarters of the edges on the synthetic scenes.
constexpr uint32_t MAP_TILE = 256;   // neighbour lists up to this length are searched in LDS
__global__ void __launch_bounds__(256) mrf_map_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const uint32_t* __restrict__ adj_ptr,
                               const uint32_t* __restrict__ adj, uint32_t F, const MrfEdge* __restrict__ edge, uint16_t* __restrict__ map,
                               uint8_t* __restrict__ ident, int skip_ident) {
    // 16 lanes per node.  The binary search is a chain of dependent loads: through global memory it is latency bound
    // (1.4 ms at C3), so the neighbour's list is first copied (coalesced) into the group's LDS tile.  A group never spans
    // waves and LDS operations of a wave execute in order, so the tile needs no barrier.
    __shared__ uint16_t s_l[16][MAP_TILE];
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t gl = threadIdx.x & 15;
    if (i >= F) return;
    uint16_t* tile = s_l[threadIdx.x >> 4];
    const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0;
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        const MrfEdge m = edge[e];
        if (m.kj == 0) continue;
        const uint32_t q0 = col_ptr[adj[e]];
        uint32_t same = (K == m.kj) ? 1u : 0u;
        if (same) {                                            // group-uniform: compare the two lists element by element (coalesced)
            for (uint32_t t = gl; t < K; t += 16) same &= (view_id[p0 + t] == view_id[q0 + t]) ? 1u : 0u;
            for (int o = 8; o > 0; o >>= 1) same &= __shfl_xor(same, o, 16);
        }
        if (gl == 0) ident[e] = (uint8_t)same;
        if (same && skip_ident) continue;
        const bool in_lds = m.kj <= MAP_TILE;                  // group-uniform
        if (in_lds) for (uint32_t t = gl; t < m.kj; t += 16) tile[t] = view_id[q0 + t];
        for (uint32_t t = gl; t < K; t += 16) {
            const uint16_t key = view_id[p0 + t];
            uint32_t lo = 0, hi = m.kj;
            if (in_lds) { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (tile[mid] < key) lo = mid + 1; else hi = mid; } }
            else { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (view_id[q0 + mid] < key) lo = mid + 1; else hi = mid; } }
            const uint16_t found = (lo < m.kj) ? (in_lds ? tile[lo] : view_id[q0 + lo]) : (uint16_t)0;
            map[m.in_off + t] = (lo < m.kj && found == key) ? (uint16_t)lo : MAP_NONE;
        }
    }
}

// One 48-byte descriptor per node (fast path, degree <= 3): everything a sweep needs to know about
// the node in a single 3 x 16-byte load instead of the col_ptr -> adj_ptr -> edge[] dependent chain.
__global__ void mrf_desc_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                const MrfEdge* __restrict__ edge, const uint8_t* __restrict__ ident, const uint32_t* __restrict__ perm,
                                uint32_t F, NodeDesc* __restrict__ desc) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;   // position in the (colour, id) order
    if (q >= F) return;
    const uint32_t i = perm[q];
    NodeDesc nd;
    nd.p0 = col_ptr[i]; nd.k = col_ptr[i + 1] - nd.p0;
    const uint32_t e0 = adj_ptr[i], deg = adj_ptr[i + 1] - e0;
    for (int d = 0; d < 3; ++d) {
        MrfEdge m; m.in_off = 0; m.out_off = 0; m.kj = 0;
        uint32_t flag = 0, nb = 0xFFFFFFFFu;
        if ((uint32_t)d < deg) nb = adj[e0 + d];
        if ((uint32_t)d < deg && nd.k > 0) {
            m = edge[e0 + d];
            // the message written over out-edge d is aligned with the neighbour's list: identity iff the lists are equal
            // (a symmetric property, so the in-edge's flag serves)
            if (m.kj && ident[e0 + d]) flag = 0x80000000u;
        }
        nd.in_off[d] = m.in_off; nd.out_off[d] = m.out_off; nd.kj[d] = m.kj | flag; nd.nbr[d] = nb;
    }
    nd.id = i; nd.pad_ = 0;
    desc[q] = nd;
}

// ---- one colour phase of a sweep; fast path: degree <= 3, K <= 4 * G ----
// ---- 4 labels per lane: lane gl owns labels 4*gl .. 4*gl+3 (one 4-byte access = four 8-bit messages, one 8-byte access =
// four u16 map entries), K <= 4 * G, so a 64-lane wave sweeps 64/G nodes per iteration at roughly the
// instruction count of one.  The kernel is VALU-issue bound (a wave64 VALU op occupies its SIMD for 4 cycles), so
// instructions per node is what counts, and the layout is arranged so that NO per-label masking is needed:
//   * elements [0, MSG_BASE) of both message buffers are zero for ever: an absent edge (degree < 3, or an empty
//     neighbour column) and every lane beyond the node's labels read their "incoming message" there;
//   * elements [0, MSG_BASE) of the map array are the identity: an edge whose two label lists are identical
//     (flag in the descriptor) reads its re-alignment map there instead of from its own run (no HBM traffic);
//   * the re-alignment gather c[p] goes through a per-group LDS tile with one extra slot holding +inf: MAP_NONE is
//     clamped onto that slot, so "label absent at the sender" needs no compare -- fmin(inf - cmin, 1/rho) = 1/rho;
//   * message runs are padded to multiples of 4 elements, so a lane stores all four of its values or none;
//   * the unary array has >= 4 floats of slack behind it (mrf_setup), so the 16-byte unary load needs no clamp.
// Values a lane computes for label slots beyond the column are garbage that never reaches a valid label: they are
// excluded from min / argmin by the ok[] mask (the only per-label selects left) and land in run padding.
// LDS operations of a wave execute in order and a lane group never spans waves, so the tile needs no barrier.
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <int G, bool DAMP, bool XCD, bool LATE_OLD>
__global__ void __launch_bounds__(256) mrf_sweep4_kernel(const NodeDesc* __restrict__ desc, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                         const uint16_t* __restrict__ map, msg_t* msg,
                                                         uint32_t* __restrict__ sel, uint32_t* __restrict__ lab, float* __restrict__ selcost,
                                                         uint32_t node_begin /* positions in the (colour, id) order */, uint32_t node_end, float rho, float alpha) {
    // in place: the nodes of one launch share a colour (an independent set), so no run is read by one node and
    // written by another; a node reads its old outgoing run before it overwrites it
    const msg_t* mo = msg; msg_t* mn = msg;
    constexpr int NPB = 256 / G;
    constexpr int TS = 4 * G + 4;                            // tile stride: 4G cavity values + the +inf slot (16-byte multiple)
    constexpr uint32_t IDENT = 0x80000000u;
    __shared__ __attribute__((aligned(16))) float s_c[NPB * TS];
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    float* __restrict__ tile = s_c + grp * TS;               // this group's 4G values, label-major
    if (gl == 0) tile[4 * G] = INFINITY;
    __syncthreads();
    const float omr = 1.0f - rho, lam = 1.0f / rho, oma = 1.0f - alpha;
    const MsgQ mq = msg_q(lam);
    const uint32_t stride = gridDim.x * NPB;
    uint32_t vb = blockIdx.x;
    if (XCD && (gridDim.x & 7u) == 0u) vb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    uint32_t i = node_begin + vb * NPB + grp;
    const uint32_t last = node_end - 1;
    NodeDesc nd = desc[min(i, last)];
    const uint32_t n_iter = (node_end - node_begin + stride - 1) / stride;
    const uint32_t t0 = 4u * gl;
    for (uint32_t it = 0; it < n_iter; ++it, i += stride) {
        const bool node_ok = i < node_end;
        const NodeDesc cur = nd;
        nd = desc[min(i + stride, last)];
        const uint32_t p0 = cur.p0, K = node_ok ? cur.k : 0u;
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ok[r] = t0 + r < K;
        // phase 1: addresses (always valid); phase 2: ALL loads as raw 8/16-byte words, issued back to back under
        // one wait (a load under a divergent branch would get its own exec region and s_waitcnt); phase 3: unpack.
        const uint32_t da = ok[0] ? p0 + t0 : 0u;
        uint32_t a_in[3], a_out[3], a_map[3], kj3[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t kjf = node_ok ? cur.kj[d] : 0u;
            kj3[d] = kjf & ~IDENT;
            const bool o0 = t0 < kj3[d];
            a_in[d] = (ok[0] && kj3[d] != 0u) ? cur.in_off[d] + t0 : t0;       // t0 < MSG_BASE: the zero run
            a_out[d] = o0 ? cur.out_off[d] + t0 : t0;
            a_map[d] = (o0 && !(kjf & IDENT)) ? cur.out_off[d] + t0 : t0;      // t0 < MSG_BASE: the identity run
        }
        const f32x4_a4 dv = *reinterpret_cast<const f32x4_a4*>(cost + da);
        uint32_t r_in[3], r_old[3]; uint2 r_map[3];   // four 8-bit messages per 4-byte word, four u16 map entries per 8-byte word
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            r_in[d] = *reinterpret_cast<const uint32_t*>(mo + a_in[d]);
            r_map[d] = *reinterpret_cast<const uint2*>(map + a_map[d]);
            if (!LATE_OLD) { if (DAMP) r_old[d] = *reinterpret_cast<const uint32_t*>(mo + a_out[d]); else r_old[d] = 0u; }
        }
        const float D[4] = {dv.x, dv.y, dv.z, dv.w};
        float in[3][4];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) in[d][r] = (float)((r_in[d] >> (8 * r)) & 0xFFu) * mq.step;   // v_cvt_f32_ubyte<r>
        }
        // decode: first argmin_t of b[t] = D[t] + rho * S[t] -- the group minimum, then the smallest label attaining it
        // (== the sequential "first minimum": comparisons are exact, +0 == -0 in both formulations)
        float bm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float S = ((0.0f + in[0][r]) + in[1][r]) + in[2][r];
            const float b = D[r] + rho * S;
            bm[r] = ok[r] ? b : INFINITY;
        }
        const float gm = group_min_fused<G>(fminf(fminf(bm[0], bm[1]), fminf(bm[2], bm[3])));
        uint32_t bt = (bm[3] == gm) ? t0 + 3u : 0xFFFFFFFFu;
        bt = (bm[2] == gm) ? t0 + 2u : bt; bt = (bm[1] == gm) ? t0 + 1u : bt; bt = (bm[0] == gm) ? t0 : bt;
        bt = group_min_fused<G>(bt);                          // every lane of the group holds the winner
        if (LATE_OLD) {
            // The previous outgoing message of edge (i -> j) is the run node j reads as "in" in this same sweep.  Issued
            // only now -- after this block's own "in" loads have landed (the decode above consumed them) -- the read hits
            // the CU's L1 / the XCD's L2 whenever j is swept by this block too, instead of racing the first fetch to HBM.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < 3; ++d) { if (DAMP) r_old[d] = *reinterpret_cast<const uint32_t*>(mo + a_out[d]); else r_old[d] = 0u; }
        }
        // label and unary of the decoded state (all the energy / ICM kernels need of a neighbour): loaded by every
        // lane (same address inside a group), consumed by the store at the end of the iteration
        const uint32_t pa = (bt < K) ? p0 + bt : 0u;          // bt < K whenever K > 0
        const uint32_t dec_view = view_id[pa];
        const float dec_cost = cost[pa];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int a = (d == 0) ? 1 : 0, b2 = (d == 2) ? 1 : 2;  // the two other slots, adjacency order
            float c[4], cm[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float oth = (0.0f + in[a][r]) + in[b2][r];
                c[r] = (D[r] + rho * oth) - omr * in[d][r];
                cm[r] = ok[r] ? c[r] : INFINITY;
            }
            const float cmin = group_min_fused<G>(fminf(fminf(cm[0], cm[1]), fminf(cm[2], cm[3])));
            *reinterpret_cast<float4*>(tile + t0) = make_float4(c[0], c[1], c[2], c[3]);
            const uint32_t mp[4] = {r_map[d].x & 0xFFFFu, r_map[d].x >> 16, r_map[d].y & 0xFFFFu, r_map[d].y >> 16};
            uint32_t w = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float cp = tile[min(mp[r], (uint32_t)(4 * G))];          // MAP_NONE -> the +inf slot
                const float raw = fminf(cp - cmin, lam);
                const float old = (float)((r_old[d] >> (8 * r)) & 0xFFu) * mq.step;
                w |= msg_code(DAMP ? (raw * oma + old * alpha) : raw, mq) << (8 * r);
            }
            if (t0 < kj3[d]) *reinterpret_cast<uint32_t*>(mn + cur.out_off[d] + t0) = w;      // one 4-byte store (runs are padded)
        }
        if (gl == 0 && node_ok) {
            /* K == 0: the single label 0 with unary 1 (view_selection.cpp:50-51,70-71) */
            const uint32_t id = cur.id;
            sel[id] = (K > 0u) ? bt : 0u; lab[id] = (K > 0u) ? dec_view + 1u : 0u; selcost[id] = (K > 0u) ? dec_cost : 1.0f;
        }
    }
}


// generic path: any degree, any K.  One wave per node, cavity vector through a global scratch row.
template <bool DAMP>
__global__ void __launch_bounds__(64) mrf_sweep_generic_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                               const uint32_t* __restrict__ adj_ptr, const MrfEdge* __restrict__ edge, const uint16_t* __restrict__ map,
                                                               msg_t* msg, const uint32_t* __restrict__ perm, uint32_t* __restrict__ sel,
                                                               uint32_t* __restrict__ lab, float* __restrict__ selcost,
                                                               float* __restrict__ scratch, uint32_t node_begin, uint32_t node_end, float rho, float alpha) {
    const msg_t* mo = msg; msg_t* mn = msg;                  // in place (one colour per launch)
    const int lane = threadIdx.x;
    if (node_begin + blockIdx.x >= node_end) return;
    const uint32_t i = perm[node_begin + blockIdx.x];
    const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0;
    if (K == 0) { if (lane == 0) { sel[i] = 0u; lab[i] = 0u; selcost[i] = 1.0f; } return; }
    const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
    const float omr = 1.0f - rho, lam = 1.0f / rho, oma = 1.0f - alpha;
    const MsgQ mq = msg_q(lam);
    float bb = INFINITY; uint32_t bt = 0xFFFFFFFFu;
    for (uint32_t t = lane; t < K; t += 64) {
        float S = 0.0f;
        for (uint32_t e = e0; e < e1; ++e) { const MrfEdge m = edge[e]; if (m.kj) S = S + msg_load(mo, m.in_off + t, mq); }
        const float b = cost[p0 + t] + rho * S;
        if (b < bb) { bb = b; bt = t; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(bb, o, 64); const uint32_t ot = __shfl_xor(bt, o, 64);
        if (ob < bb || (ob == bb && ot < bt)) { bb = ob; bt = ot; }
    }
    if (lane == 0) { sel[i] = bt; lab[i] = (uint32_t)view_id[p0 + bt] + 1u; selcost[i] = cost[p0 + bt]; }
    for (uint32_t e = e0; e < e1; ++e) {
        const MrfEdge m = edge[e];
        if (!m.kj) continue;  // wave-uniform
        float cmin = INFINITY;
        for (uint32_t t = lane; t < K; t += 64) {
            float oth = 0.0f;
            for (uint32_t e2 = e0; e2 < e1; ++e2) { if (e2 == e) continue; const MrfEdge m2 = edge[e2]; if (m2.kj) oth = oth + msg_load(mo, m2.in_off + t, mq); }
            const float c = (cost[p0 + t] + rho * oth) - omr * msg_load(mo, m.in_off + t, mq);
            scratch[p0 + t] = c;
            cmin = fminf(cmin, c);
        }
        for (int o = 32; o > 0; o >>= 1) cmin = fminf(cmin, __shfl_xor(cmin, o, 64));
        __syncthreads();
        for (uint32_t t2 = lane; t2 < m.kj; t2 += 64) {
            const uint16_t p = map[m.out_off + t2];
            const float raw = (p == MAP_NONE) ? lam : fminf(scratch[p0 + p] - cmin, lam);
            mn[m.out_off + t2] = (msg_t)msg_code(DAMP ? (raw * oma + msg_load(mo, m.out_off + t2, mq) * alpha) : raw, mq);
        }
        __syncthreads();
    }
}

// ---- exact energy of a labeling (32.32 fixed point) over nodes [node_begin, node_end) ----
// needs only the labels and selected unary costs: an edge is in the model iff both labels are
// non-zero (both columns non-empty, view_selection.cpp:29-42)
__global__ void __launch_bounds__(256) mrf_energy_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                                         const uint32_t* __restrict__ lab, const float* __restrict__ selcost,
                                                         uint32_t node_begin, uint32_t node_end, unsigned long long* __restrict__ out /* [0] energy, [1] cuts */) {
    unsigned long long unary = 0, cuts = 0;
    for (uint32_t i = node_begin + blockIdx.x * blockDim.x + threadIdx.x; i < node_end; i += gridDim.x * blockDim.x) {
        unary += fix32(selcost[i]);
        const uint32_t li = lab[i];
        if (li == 0u) continue;
        for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
            const uint32_t j = adj[e];
            if (j <= i) continue;                          /* :38 uni directional */
            const uint32_t lj = lab[j];
            cuts += (lj != 0u && lj != li);
        }
    }
    // no atomics (same-address atomics serialise at ~12 ns each): one partial pair per block, summed by
    // mrf_energy_reduce_kernel; integer sums, so the result does not depend on the launch geometry
    __shared__ unsigned long long su[4], sc[4];
    for (int o = 32; o > 0; o >>= 1) { unary += __shfl_xor(unary, o, 64); cuts += __shfl_xor(cuts, o, 64); }
    if ((threadIdx.x & 63) == 0) { su[threadIdx.x >> 6] = unary; sc[threadIdx.x >> 6] = cuts; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long u = su[0] + su[1] + su[2] + su[3], c = sc[0] + sc[1] + sc[2] + sc[3];
        out[2 * blockIdx.x] = u + (c << 32); out[2 * blockIdx.x + 1] = c;
    }
}
__global__ void __launch_bounds__(256) mrf_energy_reduce_kernel(const unsigned long long* __restrict__ partial, uint32_t n_blocks,
                                                                unsigned long long* __restrict__ out /* [0] energy, [1] cuts */) {
    unsigned long long e = 0, c = 0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 256u) { e += partial[2 * b]; c += partial[2 * b + 1]; }
    __shared__ unsigned long long su[4], sc[4];
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_xor(e, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { su[threadIdx.x >> 6] = e; sc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = su[0] + su[1] + su[2] + su[3]; out[1] = sc[0] + sc[1] + sc[2] + sc[3]; }
}

// ---- ICM polish: G lanes per node over its labels ----
template <int G>
__global__ void __launch_bounds__(256) mrf_icm_gain_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                           const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                                           const uint32_t* __restrict__ sel, const uint32_t* __restrict__ lab,
                                                           uint32_t node_begin, uint32_t node_end, float* __restrict__ gain, uint32_t* __restrict__ cand,
                                                           const uint32_t* __restrict__ list /* null: nodes [node_begin, node_end); else node ids list[node_begin .. node_end) */) {
    constexpr int NPB = 256 / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const uint32_t pos = node_begin + blockIdx.x * NPB + grp;
    const bool node_ok = pos < node_end;
    const uint32_t i = (list && node_ok) ? list[pos] : pos;
    const uint32_t p0 = node_ok ? col_ptr[i] : 0u;
    const uint32_t K = node_ok ? col_ptr[i + 1] - p0 : 0u;
    const uint32_t e0 = node_ok ? adj_ptr[i] : 0u, e1 = node_ok ? adj_ptr[i + 1] : 0u;
    const uint32_t cur_t = (K > 0) ? sel[i] : 0u;
    float best = INFINITY, cur = 0.0f; uint32_t bt = 0xFFFFFFFFu;
    // neighbour labels: the first three once (the manifold case), any further ones inside the loop
    uint32_t nl[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) nl[d] = (K > 0 && e0 + d < e1) ? lab[adj[e0 + d]] : 0u;
    for (uint32_t t = gl; t < K; t += G) {
        const uint32_t l = (uint32_t)view_id[p0 + t] + 1u;
        uint32_t diff = (nl[0] != 0u && nl[0] != l) + (nl[1] != 0u && nl[1] != l) + (nl[2] != 0u && nl[2] != l);
        for (uint32_t e = e0 + 3; e < e1; ++e) { const uint32_t lj = lab[adj[e]]; diff += (lj != 0u && lj != l); }
        const float en = cost[p0 + t] + (float)diff;
        if (en < best) { best = en; bt = t; }
        if (t == cur_t) cur = en;
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, G); const uint32_t ot = __shfl_xor(bt, o, G);
        if (ob < best || (ob == best && ot < bt)) { best = ob; bt = ot; }
        cur += __shfl_xor(cur, o, G);   // exactly one lane holds a non-zero term (or none: 0)
    }
    if (gl == 0 && node_ok) { gain[i] = (K > 0) ? (cur - best) : 0.0f; cand[i] = (K > 0) ? bt : 0u; }
}

// fast path (degree <= 3): persistent lane groups over the node descriptors, next descriptor prefetched
template <int G>
__global__ void __launch_bounds__(256) mrf_icm_gain_desc_kernel(const NodeDesc* __restrict__ desc, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                                const uint32_t* __restrict__ sel, const uint32_t* __restrict__ lab,
                                                                uint32_t node_begin, uint32_t node_end, float* __restrict__ gain, uint32_t* __restrict__ cand) {
    // node_begin / node_end are positions in the descriptor array ((colour, id) order); the node itself is cur.id
    constexpr int NPB = 256 / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const uint32_t stride = gridDim.x * NPB;
    uint32_t i = node_begin + blockIdx.x * NPB + grp;
    NodeDesc nd = {};
    if (i < node_end) nd = desc[i];
    const uint32_t n_iter = (node_end - node_begin + stride - 1) / stride;
    for (uint32_t it = 0; it < n_iter; ++it, i += stride) {
        const bool node_ok = i < node_end;
        const NodeDesc cur = nd;
        if (i + stride < node_end) nd = desc[i + stride];
        const uint32_t p0 = cur.p0, K = node_ok ? cur.k : 0u;
        const uint32_t id = cur.id;
        const uint32_t cur_t = (K > 0) ? sel[id] : 0u;
        uint32_t nl[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) nl[d] = (K > 0 && (cur.kj[d] & 0x7FFFFFFFu)) ? lab[cur.nbr[d]] : 0u;
        float best = INFINITY, cur_e = 0.0f; uint32_t bt = 0xFFFFFFFFu;
        for (uint32_t t = gl; t < K; t += G) {
            const uint32_t l = (uint32_t)view_id[p0 + t] + 1u;
            const uint32_t diff = (nl[0] != 0u && nl[0] != l) + (nl[1] != 0u && nl[1] != l) + (nl[2] != 0u && nl[2] != l);
            const float en = cost[p0 + t] + (float)diff;
            if (en < best) { best = en; bt = t; }
            if (t == cur_t) cur_e = en;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, G); const uint32_t ot = __shfl_xor(bt, o, G);
            if (ob < best || (ob == best && ot < bt)) { best = ob; bt = ot; }
            cur_e += __shfl_xor(cur_e, o, G);   // exactly one lane holds a non-zero term (or none: 0)
        }
        if (gl == 0 && node_ok) { gain[id] = (K > 0) ? (cur_e - best) : 0.0f; cand[id] = (K > 0) ? bt : 0u; }
    }
}

__global__ void __launch_bounds__(256) mrf_icm_apply_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                            const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                                            const float* __restrict__ gain, const uint32_t* __restrict__ cand,
                                                            uint32_t* sel, uint32_t* lab, float* selcost,
                                                            uint32_t node_begin, uint32_t node_end, uint32_t* __restrict__ moved /* [0] nodes moved, [1] list length */,
                                                            uint32_t* __restrict__ alist /* null, or: nodes whose gain has to be re-evaluated */) {
    const uint32_t i = node_begin + blockIdx.x * blockDim.x + threadIdx.x;
    bool mv = false;
    if (i < node_end) {
        const float gi = gain[i];
        if (gi > 0.0f) {   // only nodes with a non-empty column have a positive gain
            bool win = true;
            for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1] && win; ++e) {
                const uint32_t j = adj[e];
                if (lab[j] == 0u) continue;               // edge not in the model (gain[j] is 0 anyway)
                const float gj = gain[j];
                if (gj > gi || (gj == gi && j < i)) win = false;
            }
            if (win) mv = true;
        }
    }
    // all reads of neighbours' labels happen before any write: a winner's neighbours never win in the same
    // iteration (independent set), and lab[j] == 0 never changes
    if (mv) {
        const uint32_t p0 = col_ptr[i], t = cand[i];
        sel[i] = t; lab[i] = (uint32_t)view_id[p0 + t] + 1u; selcost[i] = cost[p0 + t];
        if (alist) {   // a node's gain depends on its column, its own label and its neighbours' labels only (duplicates in the list are harmless)
            const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
            uint32_t w = atomicAdd(&moved[1], 1u + (e1 - e0));
            alist[w++] = i;
            for (uint32_t e = e0; e < e1; ++e) alist[w++] = adj[e];
        }
    }
    const unsigned long long b = __ballot(mv);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(moved, (uint32_t)__popcll(b));
}

// argmin-unary start (max_sweeps == 0): sel / lab / selcost of every node
__global__ void mrf_argmin_unary_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost, uint32_t F,
                                        uint32_t* __restrict__ sel, uint32_t* __restrict__ lab, float* __restrict__ selcost) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0;
    uint32_t bt = 0;
    for (uint32_t t = 1; t < K; ++t) if (cost[p0 + t] < cost[p0 + bt]) bt = t;
    sel[i] = bt;
    lab[i] = K ? (uint32_t)view_id[p0 + bt] + 1u : 0u;
    selcost[i] = K ? cost[p0 + bt] : 1.0f;
}

/* label extraction (view_selection.cpp:120-132): labels are already decoded; range check + unseen count */
__global__ void mrf_labels_kernel(const uint32_t* __restrict__ lab, uint32_t node_begin, uint32_t node_end, uint32_t n_views,
                                  uint32_t* __restrict__ labels, uint32_t* __restrict__ bad_unseen /* [0] bad, [1] unseen */) {
    const uint32_t i = node_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_end) return;
    const uint32_t label = lab[i];
    if (label > n_views) atomicAdd(&bad_unseen[0], 1u);       /* :126-128 "Incorrect labeling" */
    if (label == 0u) atomicAdd(&bad_unseen[1], 1u);            /* :129 */
    labels[i - node_begin] = label;
}


// ---- device-side solver bookkeeping: one thread.  Same decisions, in the same arithmetic, as the host loop it
// replaces: best = min(best, e); stop iff sweep >= min_sweeps, sweep > window and
// double(hist[sweep - window] - best) < double(min_improvement) * double(hist[sweep - window]).
// With `partial` (single-GPU loop) the block first sums the energy kernel's per-block pairs itself and publishes them in
// energy_out -- one launch less per sweep than reduce + step; sharded callers pass the all-reduced pair in `energy`.
__global__ void __launch_bounds__(256) mrf_step_kernel(mvs_mrf_progress* __restrict__ st, unsigned long long* __restrict__ hist,
                                const unsigned long long* __restrict__ energy, const unsigned long long* __restrict__ partial, uint32_t n_partial,
                                unsigned long long* __restrict__ energy_out, int max_sweeps, int min_sweeps, int window,
                                float min_improvement) {
    __shared__ unsigned long long su[4], sc[4];
    unsigned long long e_sum = 0, c_sum = 0;
    if (partial) {
        for (uint32_t b = threadIdx.x; b < n_partial; b += 256u) { e_sum += partial[2 * b]; c_sum += partial[2 * b + 1]; }
        for (int o = 32; o > 0; o >>= 1) { e_sum += __shfl_xor(e_sum, o, 64); c_sum += __shfl_xor(c_sum, o, 64); }
        if ((threadIdx.x & 63) == 0) { su[threadIdx.x >> 6] = e_sum; sc[threadIdx.x >> 6] = c_sum; }
        __syncthreads();
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (partial) {
        e_sum = su[0] + su[1] + su[2] + su[3]; c_sum = sc[0] + sc[1] + sc[2] + sc[3];
        energy_out[0] = e_sum; energy_out[1] = c_sum;
    }
    if (st->stopped) { st->improved = 0u; return; }
    const uint32_t sw = st->sweep + 1u;
    const unsigned long long e0 = partial ? e_sum : energy[0];
    unsigned long long best = st->best;
    const bool imp = e0 < best;
    if (imp) best = e0;
    st->sweep = sw; st->improved = imp ? 1u : 0u; st->energy = e0; st->best = best;
    hist[sw] = best;
    bool stop = false;
    if ((int)sw >= min_sweeps && (int)sw > window) {
        const unsigned long long prev = hist[sw - (uint32_t)window];
        stop = (double)(prev - best) < (double)min_improvement * (double)prev;
    }
    if ((int)sw >= max_sweeps) stop = true;
    if (stop) { st->stopped = 1u; st->stop_sweep = sw; }
}
// best labeling := current decode, iff the step above saw an improvement (the flag is wave-uniform)
__global__ void __launch_bounds__(256) mrf_keep_best_if_kernel(const mvs_mrf_progress* __restrict__ st, const uint32_t* __restrict__ sel,
                                                               const uint32_t* __restrict__ lab, const float* __restrict__ cost,
                                                               uint32_t* __restrict__ best_sel, uint32_t* __restrict__ best_lab,
                                                               float* __restrict__ best_cost, uint32_t n) {
    if (!st->improved) return;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        best_sel[i] = sel[i]; best_lab[i] = lab[i]; best_cost[i] = cost[i];
    }
}

}  // namespace

// Builds the solver's edge tables for the active CSR (ctx->r_ptr / r_view / r_cost) and adjacency.
void mrf_setup(mvs_ctx* ctx, const mvs_mrf_params* params) {
    hipStream_t s = ctx->stream;
    const uint32_t F = ctx->csr_faces;
    ctx->m_params = *params;
    if (!(params->rho > 0.0f && params->rho <= 1.0f) || !(params->damping >= 0.0f && params->damping < 1.0f))
        throw StatusError(MVS_ERR_INVALID, "mrf params: need 0 < rho <= 1, 0 <= damping < 1");
    uint32_t E = 0;
    MVS_HIP(hipMemcpyAsync(&E, ctx->r_adj_ptr + F, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MVS_HIP(hipStreamSynchronize(s));
    ctx->m_size.ensure((size_t)E + 2); ctx->m_edge.ensure((size_t)E + 1); ctx->m_moved.ensure(8 + 2 * 64);
    ctx->m_n_adj = E;
    uint32_t* maxes = ctx->m_moved.p + 4;
    MVS_HIP(hipMemsetAsync(ctx->m_moved.p, 0, 8 * sizeof(uint32_t), s));
    MVS_HIP(hipMemsetAsync(ctx->m_size.p, 0, ((size_t)E + 2) * sizeof(uint32_t), s));
    const unsigned nb = (F + 255) / 256;
    if (F) { hipLaunchKernelGGL(mrf_size_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_adj_ptr, ctx->r_adj, F, (uint32_t)(ctx->mrf_run_pad == 16 ? 15 : 3), ctx->m_size.p, maxes); MVS_LAUNCH_CHECK(); }
    // ---- colour-phased schedule: colouring (Jones-Plassmann rounds), nodes in (colour, id) order ----
    ctx->m_colour.ensure((size_t)F + 2); ctx->m_perm.ensure((size_t)F + 2); ctx->m_tmp_a.ensure((size_t)MAX_LAYOUT_COLOURS * ((size_t)F + 1) + 72); ctx->m_tmp_b.ensure((size_t)MAX_LAYOUT_COLOURS * ((size_t)F + 1) + 2); ctx->m_tmp_c.ensure((size_t)F + 2);
    ctx->m_colours = 0; ctx->m_colour_begin.assign(66, 0); ctx->m_range_q.clear(); ctx->m_range_nb = ctx->m_range_ne = 0;
    ctx->m_sweep_no = 0;
    if (F) {
        uint32_t* pending = ctx->m_moved.p + 1;
        hipLaunchKernelGGL(mrf_colour_init_kernel, dim3(nb), dim3(256), 0, s, ctx->m_colour.p, ctx->m_tmp_a.p /* iota */, F); MVS_LAUNCH_CHECK();
        for (int round = 0;; ++round) {
            if (round >= 4096) throw StatusError(MVS_ERR_HIP, "graph colouring did not terminate");
            MVS_HIP(hipMemsetAsync(pending, 0, sizeof(uint32_t), s));
            for (int k = 0; k < 4; ++k) { hipLaunchKernelGGL(mrf_colour_round_kernel, dim3(nb), dim3(256), 0, s, ctx->r_adj_ptr, ctx->r_adj, F, ctx->m_colour.p, pending); MVS_LAUNCH_CHECK(); }
            uint32_t hp = 0;   // set if any of the four rounds left a node waiting
            MVS_HIP(hipMemcpyAsync(&hp, pending, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            MVS_HIP(hipStreamSynchronize(s));
            if (!hp) break;
        }
        // stable sort of the node ids by colour: perm = nodes in (colour, id) order; a colour class is a contiguous range
        size_t tmp_bytes = 0;
        MVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ctx->m_colour.p, ctx->m_tmp_b.p, ctx->m_tmp_a.p, ctx->m_perm.p, F, 0, 6, s));
        ctx->sort_tmp.ensure(tmp_bytes + 16);
        MVS_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->m_colour.p, ctx->m_tmp_b.p, ctx->m_tmp_a.p, ctx->m_perm.p, F, 0, 6, s));
        hipLaunchKernelGGL(mrf_colour_begin_kernel, dim3(1), dim3(128), 0, s, ctx->m_tmp_b.p, F, ctx->m_tmp_c.p); MVS_LAUNCH_CHECK();
        MVS_HIP(hipMemcpyAsync(ctx->m_colour_begin.data(), ctx->m_tmp_c.p, 65 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MVS_HIP(hipStreamSynchronize(s));
        ctx->m_colour_begin[65] = F;
        uint32_t C = 0; while (C < 64 && ctx->m_colour_begin[C] < F) ++C;   // colours 0 .. C-1 are in use (greedy colours are dense)
        ctx->m_colours = C;
        ctx->m_colour_begin.resize(C + 1); ctx->m_colour_begin[C] = F;
    }
    // message layout (sender-major, (colour, id) node order): in_off[e] for every directed edge e (adjacency order);
    // edges whose reverse is missing (asymmetric input) keep offset 0 and are disabled by mrf_edge_kernel
    DBuf<uint32_t>& in_off = ctx->m_sel2;  // temporary home, re-ensured below
    in_off.ensure(std::max<size_t>((size_t)E + 2, (size_t)F + 2));
    MVS_HIP(hipMemsetAsync(in_off.p, 0, ((size_t)E + 2) * sizeof(uint32_t), s));
    uint32_t h[3] = {0, 0, 0};
    if (F) {
        const uint32_t n_col = (ctx->m_colours >= 2 && ctx->m_colours <= (uint32_t)MAX_LAYOUT_COLOURS) ? ctx->m_colours : 1u;
        const size_t n_ent = (size_t)n_col * ((size_t)F + 1);
        ctx->m_tmp_a.ensure(n_ent + 72); ctx->m_tmp_b.ensure(n_ent + 2);
        hipLaunchKernelGGL(mrf_nodesize_kernel, dim3((F + 256) / 256), dim3(256), 0, s, ctx->m_perm.p, ctx->m_colour.p, ctx->r_adj_ptr, ctx->r_adj, ctx->m_size.p, F, n_col, ctx->m_tmp_a.p); MVS_LAUNCH_CHECK();
        exclusive_scan_u32(ctx, ctx->m_tmp_a.p, ctx->m_tmp_b.p, n_ent, nullptr);   // the last entry of every colour segment is 0, so scan[last] = total
        hipLaunchKernelGGL(mrf_inoff_kernel, dim3(nb), dim3(256), 0, s, ctx->m_perm.p, ctx->m_colour.p, ctx->r_adj_ptr, ctx->r_adj, ctx->m_size.p, ctx->m_tmp_b.p, F, n_col, in_off.p); MVS_LAUNCH_CHECK();
        MVS_HIP(hipMemcpyAsync(&h[0], ctx->m_tmp_b.p + (n_ent - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    MVS_HIP(hipMemcpyAsync(&h[1], maxes, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MVS_HIP(hipStreamSynchronize(s));
    ctx->m_total = (uint64_t)MSG_BASE + h[0]; ctx->m_kmax = h[1]; ctx->m_degmax = h[2];
    if (ctx->m_total >= 0xFFFFFFF0ull) throw StatusError(MVS_ERR_UNSUPPORTED, "message array exceeds 2^32 elements");
    if (F) { hipLaunchKernelGGL(mrf_edge_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_adj_ptr, ctx->r_adj, F, in_off.p, ctx->m_size.p, ctx->m_edge.p); MVS_LAUNCH_CHECK(); }
    ctx->m_map.ensure(ctx->m_total + 8); ctx->m_ident.ensure((size_t)E + 1);
    MVS_HIP(hipMemsetAsync(ctx->m_map.p, 0, (ctx->m_total + 8) * sizeof(uint16_t), s));   // run padding is read (and ignored): keep it defined
    hipLaunchKernelGGL(mrf_identity_kernel, dim3(1), dim3(MSG_BASE), 0, s, ctx->m_map.p); MVS_LAUNCH_CHECK();   // map[t] = t for t < MSG_BASE
    MVS_HIP(hipMemsetAsync(ctx->m_ident.p, 0, (size_t)E + 1, s));
    if (F) { hipLaunchKernelGGL(mrf_map_kernel, dim3((unsigned)(((size_t)F * 16 + 255) / 256)), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_adj_ptr, ctx->r_adj, F, ctx->m_edge.p, ctx->m_map.p, ctx->m_ident.p,
                               (ctx->m_degmax <= 3 && ctx->m_kmax <= 256 && ctx->csr_nnz >= 4) ? 1 : 0); MVS_LAUNCH_CHECK(); }
    ctx->m_desc.ensure((size_t)F + 1);
    if (F) { hipLaunchKernelGGL(mrf_desc_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_adj_ptr, ctx->r_adj, ctx->m_edge.p, ctx->m_ident.p, ctx->m_perm.p, F, ctx->m_desc.p); MVS_LAUNCH_CHECK(); }
    ctx->m_msg_a.ensure(ctx->m_total + 8);
    MVS_HIP(hipMemsetAsync(ctx->m_msg_a.p, 0, (ctx->m_total + 8) * sizeof(msg_t), s));   // zero codes, incl. the reserved zero run
    // the sweep reads unaries with unclamped 16-byte loads: a caller-owned cost array (mvs_ctx_costs_upload with device
    // pointers) is copied into the context's own buffer, which always has slack behind the last element
    if (ctx->r_cost != ctx->csr_cost.p && ctx->csr_nnz) {
        ctx->csr_cost.ensure(ctx->csr_nnz + 8);
        MVS_HIP(hipMemcpyAsync(ctx->csr_cost.p, ctx->r_cost, ctx->csr_nnz * sizeof(float), hipMemcpyDeviceToDevice, s));
        ctx->r_cost = ctx->csr_cost.p;
    }
    if (ctx->r_cost == ctx->csr_cost.p && ctx->csr_cost.cap >= ctx->csr_nnz + 8) MVS_HIP(hipMemsetAsync(ctx->csr_cost.p + ctx->csr_nnz, 0, 8 * sizeof(float), s));
    MVS_HIP(hipStreamSynchronize(s));  // in_off (m_sel2) is consumed; safe to reuse
    ctx->m_sel.ensure((size_t)F + 1); ctx->m_best_sel.ensure((size_t)F + 1); ctx->m_sel2.ensure((size_t)F + 1); ctx->m_cand.ensure((size_t)F + 1); ctx->m_gain.ensure((size_t)F + 1);
    ctx->m_lab.ensure((size_t)F + 1); ctx->m_best_lab.ensure((size_t)F + 1); ctx->m_cost.ensure((size_t)F + 1); ctx->m_best_cost.ensure((size_t)F + 1);
    // start state = argmin-unary decode everywhere: only needed when no sweep runs (ICM-only); otherwise the first
    // sweep (and, when sharded, the halo exchange that follows it) defines every label that is ever read
    MVS_HIP(hipMemsetAsync(ctx->m_lab.p, 0, ((size_t)F + 1) * sizeof(uint32_t), s));
    MVS_HIP(hipMemsetAsync(ctx->m_best_lab.p, 0, ((size_t)F + 1) * sizeof(uint32_t), s));
    if (F && params->max_sweeps <= 0) {
        hipLaunchKernelGGL(mrf_argmin_unary_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_cost, F, ctx->m_sel.p, ctx->m_lab.p, ctx->m_cost.p);
        MVS_LAUNCH_CHECK();
        MVS_HIP(hipMemcpyAsync(ctx->m_best_sel.p, ctx->m_sel.p, (size_t)F * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        MVS_HIP(hipMemcpyAsync(ctx->m_best_lab.p, ctx->m_lab.p, (size_t)F * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        MVS_HIP(hipMemcpyAsync(ctx->m_best_cost.p, ctx->m_cost.p, (size_t)F * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    MVS_HIP(hipMemsetAsync(ctx->m_gain.p, 0, ((size_t)F + 1) * sizeof(float), s));
    ctx->m_energy.ensure(4 + 2 * 2048);
    // device-side solver state: sweep 0, best = hist[0] = 2^64 - 1
    ctx->m_state.ensure(1); ctx->m_hist.ensure((size_t)std::max(params->max_sweeps, 0) + 2);
    mvs_mrf_progress init; memset(&init, 0, sizeof(init)); init.best = ~0ull; init.energy = ~0ull;
    if (!ctx->h_ring) {
        MVS_HIP(hipHostMalloc((void**)&ctx->h_ring, mvs_ctx::RING * sizeof(mvs_mrf_progress), hipHostMallocDefault));
        for (uint32_t k = 0; k < mvs_ctx::RING; ++k) MVS_HIP(hipEventCreateWithFlags(&ctx->ring_ev[k], hipEventDisableTiming));
    }
    ctx->h_ring[0] = init;
    MVS_HIP(hipMemcpyAsync(ctx->m_state.p, &ctx->h_ring[0], sizeof(init), hipMemcpyHostToDevice, s));
    MVS_HIP(hipMemsetAsync(ctx->m_hist.p, 0xFF, sizeof(unsigned long long), s));
    MVS_HIP(hipStreamSynchronize(s));
    ctx->steps_issued = 0; ctx->icm_dirty_valid = false;
}

// One bookkeeping step (see mrf_step_kernel); energy = device pointer to the (all-reduced) energy pair.
void mrf_step(mvs_ctx* ctx, const unsigned long long* energy) {
    hipStream_t s = ctx->stream;
    const mvs_mrf_params& P = ctx->m_params;
    if (!ctx->h_ring) throw StatusError(MVS_ERR_STATE, "mrf step before mrf setup");
    // energy == nullptr: the pair is still in per-block partials (mrf_energy(..., reduce = false)), summed by the step kernel
    hipLaunchKernelGGL(mrf_step_kernel, dim3(1), dim3(256), 0, s, ctx->m_state.p, ctx->m_hist.p, energy,
                       energy ? (const unsigned long long*)nullptr : ctx->m_energy.p + 4, energy ? 0u : ctx->m_energy_blocks, ctx->m_energy.p,
                       P.max_sweeps, P.min_sweeps, P.window, P.min_improvement);
    MVS_LAUNCH_CHECK();
    const uint32_t F = ctx->csr_faces;
    if (F) {
        hipLaunchKernelGGL(mrf_keep_best_if_kernel, dim3(std::min<unsigned>((F + 255) / 256, 2048u)), dim3(256), 0, s, ctx->m_state.p,
                           ctx->m_sel.p, ctx->m_lab.p, ctx->m_cost.p, ctx->m_best_sel.p, ctx->m_best_lab.p, ctx->m_best_cost.p, F);
        MVS_LAUNCH_CHECK();
    }
    ctx->icm_dirty_valid = false;   // the best labeling may change
    const uint32_t n = ++ctx->steps_issued, slot = n % mvs_ctx::RING;
    MVS_HIP(hipMemcpyAsync(&ctx->h_ring[slot], ctx->m_state.p, sizeof(mvs_mrf_progress), hipMemcpyDeviceToHost, s));
    MVS_HIP(hipEventRecord(ctx->ring_ev[slot], s));
}
void mrf_poll(mvs_ctx* ctx, uint32_t step, mvs_mrf_progress* out) {
    if (step == 0 || step > ctx->steps_issued || step + mvs_ctx::RING <= ctx->steps_issued)
        throw StatusError(MVS_ERR_INVALID, "mrf poll: step not among the last 16 issued");
    const uint32_t slot = step % mvs_ctx::RING;
    MVS_HIP(hipEventSynchronize(ctx->ring_ev[slot]));
    *out = ctx->h_ring[slot];
}

// Damping schedule (part of the solver's definition, restated in oracle/oracle.cpp): messages are damped with
// alpha = params.damping on ODD sweeps (1st, 3rd, ...) and written undamped on even sweeps.  Undamped sweeps oscillate
// (C3: 0.9 % higher final energy), but damping every second sweep suppresses that just as well as damping every sweep
// (C3: 44 sweeps to E = 1 111 890 with 0.2 on odd sweeps vs 47 to 1 110 970 with 0.1 on all) -- and an undamped sweep
// does not re-read its previous outgoing messages (6 nnz bytes of the ~27 nnz a damped sweep moves).
static float sweep_alpha(const mvs_ctx* ctx) { return (ctx->m_sweep_no & 1u) ? ctx->m_params.damping : 0.0f; }

template <int G>
static void launch_sweep4_g(mvs_ctx* ctx, uint32_t qb, uint32_t qe) {
    constexpr int NPB = 256 / G;
    const unsigned need = (qe - qb + NPB - 1) / NPB;
    const float rho = ctx->m_params.rho, alpha = sweep_alpha(ctx);
    // persistent lane groups: at most as many blocks as are resident at once (a partial second wave of
    // blocks would double the tail); mrf_blocks_per_cu > 0 overrides
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0; hipDeviceProp_t prop;
        MVS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mrf_sweep4_kernel<G, true, true, true>, 256, 0));
        MVS_HIP(hipGetDeviceProperties(&prop, ctx->device));
        resident = std::max(1, per_cu) * prop.multiProcessorCount;
    }
    unsigned blocks = ctx->mrf_blocks_per_cu > 0 ? 256u * (unsigned)ctx->mrf_blocks_per_cu : (unsigned)resident;
    blocks = std::max(1u, std::min(need, blocks));
    if (blocks > 8) blocks &= ~7u;   // multiple of the 8 XCDs
    msg_t* msg = reinterpret_cast<msg_t*>(ctx->m_msg_a.p);
#define SWEEP4_ARGS dim3(blocks), dim3(256), 0, ctx->stream, ctx->m_desc.p, ctx->r_view, ctx->r_cost, ctx->m_map.p, msg, ctx->m_sel.p, ctx->m_lab.p, ctx->m_cost.p, qb, qe, rho, alpha
    if (alpha != 0.0f) {
        if (ctx->mrf_late_old) { if (ctx->mrf_xcd) hipLaunchKernelGGL((mrf_sweep4_kernel<G, true, true, true>), SWEEP4_ARGS); else hipLaunchKernelGGL((mrf_sweep4_kernel<G, true, false, true>), SWEEP4_ARGS); }
        else { if (ctx->mrf_xcd) hipLaunchKernelGGL((mrf_sweep4_kernel<G, true, true, false>), SWEEP4_ARGS); else hipLaunchKernelGGL((mrf_sweep4_kernel<G, true, false, false>), SWEEP4_ARGS); }
    } else { if (ctx->mrf_xcd) hipLaunchKernelGGL((mrf_sweep4_kernel<G, false, true, false>), SWEEP4_ARGS); else hipLaunchKernelGGL((mrf_sweep4_kernel<G, false, false, false>), SWEEP4_ARGS); }
#undef SWEEP4_ARGS
}

// positions [qb, qe) in the (colour, id) order of the nodes of colour `phase` whose id lies in [nb0, ne0)
static void phase_range(mvs_ctx* ctx, uint32_t phase, uint32_t nb0, uint32_t ne0, uint32_t* qb, uint32_t* qe) {
    const uint32_t cb = ctx->m_colour_begin[phase], ce = ctx->m_colour_begin[phase + 1];
    if (nb0 == 0 && ne0 >= ctx->csr_faces) { *qb = cb; *qe = ce; return; }
    if (ctx->m_range_nb != nb0 || ctx->m_range_ne != ne0 || ctx->m_range_q.size() != 2 * (size_t)ctx->m_colours) {
        // a rank's own share of every colour class: one small kernel + read-back per (range, setup), then cached
        const uint32_t C = ctx->m_colours;
        ctx->m_moved.ensure(8 + 2 * 64);
        uint32_t* d = ctx->m_moved.p + 8;
        ctx->m_tmp_a.ensure(72);
        MVS_HIP(hipMemcpyAsync(ctx->m_tmp_a.p, ctx->m_colour_begin.data(), (C + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(mrf_phase_range_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->m_perm.p, ctx->m_tmp_a.p, C, nb0, ne0, d);
        MVS_LAUNCH_CHECK();
        ctx->m_range_q.assign(2 * (size_t)C, 0);
        MVS_HIP(hipMemcpyAsync(ctx->m_range_q.data(), d, 2 * C * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        ctx->m_range_nb = nb0; ctx->m_range_ne = ne0;
    }
    *qb = ctx->m_range_q[2 * phase]; *qe = ctx->m_range_q[2 * phase + 1];
}

// one colour phase of a sweep over the nodes of that colour with id in [nb0, ne0): in place
void mrf_sweep_phase(mvs_ctx* ctx, uint32_t phase, uint32_t nb0, uint32_t ne0) {
    if (phase == 0) ++ctx->m_sweep_no;      // sweeps are counted by their first phase (every caller runs the phases in order)
    if (phase >= ctx->m_colours || ne0 <= nb0) return;
    uint32_t qb, qe;
    phase_range(ctx, phase, nb0, ne0, &qb, &qe);
    if (qe <= qb) return;
    const uint32_t K = ctx->m_kmax;
    if (ctx->m_degmax <= 3 && K <= 256 && ctx->csr_nnz >= 4 && ctx->m_total > MSG_BASE) {
        if (K <= 32) launch_sweep4_g<8>(ctx, qb, qe);
        else if (K <= 64) launch_sweep4_g<16>(ctx, qb, qe);
        else if (K <= 128) launch_sweep4_g<32>(ctx, qb, qe);
        else launch_sweep4_g<64>(ctx, qb, qe);      // one node per wave: scenes with several hundred views per face
    } else {
        ctx->pq.ensure(ctx->csr_nnz + 1);  // scratch row per node (data-cost work buffer is free by now)
        const float rho = ctx->m_params.rho, alpha = sweep_alpha(ctx);
        msg_t* msg = reinterpret_cast<msg_t*>(ctx->m_msg_a.p);
        if (alpha != 0.0f)
            hipLaunchKernelGGL(mrf_sweep_generic_kernel<true>, dim3(qe - qb), dim3(64), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->r_adj_ptr, ctx->m_edge.p, ctx->m_map.p, msg, ctx->m_perm.p, ctx->m_sel.p, ctx->m_lab.p, ctx->m_cost.p, ctx->pq.p, qb, qe, rho, alpha);
        else
            hipLaunchKernelGGL(mrf_sweep_generic_kernel<false>, dim3(qe - qb), dim3(64), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->r_adj_ptr, ctx->m_edge.p, ctx->m_map.p, msg, ctx->m_perm.p, ctx->m_sel.p, ctx->m_lab.p, ctx->m_cost.p, ctx->pq.p, qb, qe, rho, alpha);
    }
    MVS_LAUNCH_CHECK();
}
// one sweep = every colour phase in turn (callers that shard the nodes exchange halos between the phases themselves)
void mrf_sweep(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    for (uint32_t ph = 0; ph < ctx->m_colours; ++ph) mrf_sweep_phase(ctx, ph, nb0, ne0);
}

// energy of the current decode (best == false) or of the best labeling over nodes [nb0, ne0)
// -> ctx->m_energy (device, 2 x u64), asynchronous
void mrf_energy(mvs_ctx* ctx, bool best, uint32_t nb0, uint32_t ne0, bool reduce) {
    const unsigned blocks = ne0 > nb0 ? std::min<unsigned>((ne0 - nb0 + 255) / 256, 2048u) : 0u;
    ctx->m_energy_blocks = blocks;
    ctx->m_energy.ensure(4 + 2 * 2048);
    unsigned long long* partial = ctx->m_energy.p + 4;
    if (blocks) {
        hipLaunchKernelGGL(mrf_energy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->r_adj_ptr, ctx->r_adj,
                           best ? ctx->m_best_lab.p : ctx->m_lab.p, best ? ctx->m_best_cost.p : ctx->m_cost.p, nb0, ne0, partial);
        MVS_LAUNCH_CHECK();
    }
    if (!reduce) return;   // the caller's mrf_step sums the partials
    hipLaunchKernelGGL(mrf_energy_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, blocks, ctx->m_energy.p);
    MVS_LAUNCH_CHECK();
}

void mrf_keep_best(mvs_ctx* ctx) {
    const size_t F = ctx->csr_faces;
    ctx->icm_dirty_valid = false;
    if (!F) return;
    MVS_HIP(hipMemcpyAsync(ctx->m_best_sel.p, ctx->m_sel.p, F * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_HIP(hipMemcpyAsync(ctx->m_best_lab.p, ctx->m_lab.p, F * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    MVS_HIP(hipMemcpyAsync(ctx->m_best_cost.p, ctx->m_cost.p, F * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
}

// ICM on the best labeling (in place): gains of nodes [nb0, ne0)
void mrf_icm_gain(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    if (ne0 <= nb0) return;
    const uint32_t K = ctx->m_kmax, n = ne0 - nb0;
    const bool whole = nb0 == 0 && ne0 == ctx->csr_faces;
#define ICM_G(GG, B, E, LIST) hipLaunchKernelGGL(mrf_icm_gain_kernel<GG>, dim3(((E) - (B) + (256 / GG) - 1) / (256 / GG)), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, \
                                     ctx->r_adj_ptr, ctx->r_adj, ctx->m_best_sel.p, ctx->m_best_lab.p, (B), (E), ctx->m_gain.p, ctx->m_cand.p, (LIST))
    // active set (unsharded calls only): after one full evaluation, only the nodes the last apply listed -- the nodes that
    // moved and their neighbours -- are re-evaluated; everybody else's stored gain / candidate are still the values a
    // full pass would compute.  Sharded callers exchange labels behind the library's back, so they evaluate all.
    if (whole && ctx->icm_dirty_valid) {
        uint32_t cnt = 0;
        MVS_HIP(hipMemcpyAsync(&cnt, ctx->m_moved.p + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        if (cnt) {
            const uint32_t* list = ctx->m_alist.p;
            if (K <= 8) ICM_G(8, 0u, cnt, list); else if (K <= 16) ICM_G(16, 0u, cnt, list); else if (K <= 32) ICM_G(32, 0u, cnt, list); else ICM_G(64, 0u, cnt, list);
            MVS_LAUNCH_CHECK();
        }
        return;
    }
    if (ctx->m_degmax <= 3 && whole) {   // descriptors are in (colour, id) order: whole-graph calls only
#define ICM_D(GG) hipLaunchKernelGGL(mrf_icm_gain_desc_kernel<GG>, dim3(std::max(1u, std::min<unsigned>((n + (256 / GG) - 1) / (256 / GG), 256u * 8u))), dim3(256), 0, ctx->stream, \
                                     ctx->m_desc.p, ctx->r_view, ctx->r_cost, ctx->m_best_sel.p, ctx->m_best_lab.p, nb0, ne0, ctx->m_gain.p, ctx->m_cand.p)
        if (K <= 8) ICM_D(8); else if (K <= 16) ICM_D(16); else if (K <= 48) ICM_D(16); else ICM_D(32);
#undef ICM_D
    } else {
        const uint32_t* none = nullptr;
        if (K <= 8) ICM_G(8, nb0, ne0, none); else if (K <= 16) ICM_G(16, nb0, ne0, none); else if (K <= 32) ICM_G(32, nb0, ne0, none); else ICM_G(64, nb0, ne0, none);
    }
#undef ICM_G
    MVS_LAUNCH_CHECK();
    ctx->icm_dirty_valid = whole;   // every gain is current: the next apply starts the list
}
void mrf_icm_apply(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    MVS_HIP(hipMemsetAsync(ctx->m_moved.p, 0, 2 * sizeof(uint32_t), ctx->stream));
    if (ne0 <= nb0) return;
    const bool whole = nb0 == 0 && ne0 == ctx->csr_faces;
    if (!whole) ctx->icm_dirty_valid = false;
    uint32_t* alist = nullptr;
    if (ctx->icm_dirty_valid) {   // winners form an independent set: at most (nodes + directed edges) entries
        ctx->m_alist.ensure((size_t)ctx->csr_faces + (size_t)ctx->m_n_adj + 16);
        alist = ctx->m_alist.p;
    }
    hipLaunchKernelGGL(mrf_icm_apply_kernel, dim3((ne0 - nb0 + 255) / 256), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->r_adj_ptr, ctx->r_adj,
                       ctx->m_gain.p, ctx->m_cand.p, ctx->m_best_sel.p, ctx->m_best_lab.p, ctx->m_best_cost.p, nb0, ne0, ctx->m_moved.p, alist);
    MVS_LAUNCH_CHECK();
}
// labels of nodes [nb0, ne0) of the best labeling into d_labels[0 .. ne0 - nb0); out = {bad, unseen}
void mrf_labels(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0, uint32_t* d_labels, uint32_t out[2]) {
    uint32_t* bu = ctx->m_moved.p + 2;
    MVS_HIP(hipMemsetAsync(bu, 0, 2 * sizeof(uint32_t), ctx->stream));
    if (ne0 > nb0) {
        hipLaunchKernelGGL(mrf_labels_kernel, dim3((ne0 - nb0 + 255) / 256), dim3(256), 0, ctx->stream, ctx->m_best_lab.p, nb0, ne0, ctx->csr_views, d_labels, bu);
        MVS_LAUNCH_CHECK();
    }
    MVS_HIP(hipMemcpyAsync(out, bu, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
}

}  // namespace mvs
