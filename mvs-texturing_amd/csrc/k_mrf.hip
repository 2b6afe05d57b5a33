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
#include <atomic>

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
// the code a message value `raw` in [0, lam] is stored as, damped against the old code (oracle.cpp msg_code):
// rne(fma(old, alpha, raw * oms)), oms = (1 - alpha) * scale, saturated at 255 -- v_cvt_pk_u8_f32 IS a saturating
// round-to-nearest-even conversion on gfx950 (scripts/probe/cvt_probe.hip), and it drops the byte into place
// 32-bit byte offsets from a wave-uniform base: the load takes the base from SGPRs and the offset from one VGPR
// (global_load ... v_off, s[base]) instead of a 64-bit address built with v_lshl_add_u64 per load
template <class T> __device__ __forceinline__ T ld_off(const void* base, uint32_t byte_off) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off); }
template <class T> __device__ __forceinline__ void st_off(void* base, uint32_t byte_off, T v) { *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v; }
// v_min_f32 without the canonicalising v_max_f32 x, x, x the compiler puts in front of fminf() on values it loaded from
// memory (no NaN can reach these operands: +inf and finite values only)
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <bool DAMP>
__device__ __forceinline__ uint32_t msg_pack_s(float raw_s, float alpha, float old_code, uint32_t byte, uint32_t word) {
    return __builtin_amdgcn_cvt_pk_u8_f32(DAMP ? __builtin_fmaf(old_code, alpha, raw_s) : raw_s, byte, word);
}
template <bool DAMP>
__device__ __forceinline__ uint32_t msg_pack(float raw, float oms, float alpha, float old_code, uint32_t byte, uint32_t word) {
    const float v = DAMP ? __builtin_fmaf(old_code, alpha, raw * oms) : raw * oms;
    return __builtin_amdgcn_cvt_pk_u8_f32(v, byte, word);
}

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
// Four group minima at once: the DPP steps of the four chains are interleaved, so a step's operand was written three
// instructions earlier and only the first block needs the DPP read-after-write wait states (one chain at a time costs
// `s_nop 1` plus the compiler's own padding in front of each of its 4 steps).
#define MVS_DPP4(nop, ctrl) asm(nop "v_min_f32_dpp %0, %4, %4 " ctrl " row_mask:0xf bank_mask:0xf\n\tv_min_f32_dpp %1, %5, %5 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
                                    "v_min_f32_dpp %2, %6, %6 " ctrl " row_mask:0xf bank_mask:0xf\n\tv_min_f32_dpp %3, %7, %7 " ctrl " row_mask:0xf bank_mask:0xf"          \
                                : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd) : "v"(a), "v"(b), "v"(c), "v"(d)); a = ra; b = rb; c = rc; d = rd
template <int G>
__device__ __forceinline__ void group_min_fused4(float& a, float& b, float& c, float& d) {
    if (G == 8 || G == 16) {
        float ra, rb, rc, rd;
        MVS_DPP4("s_nop 1\n\t", "quad_perm:[1,0,3,2]");
        MVS_DPP4("", "quad_perm:[2,3,0,1]");
        MVS_DPP4("", "row_half_mirror");
        if (G == 16) { MVS_DPP4("", "row_mirror"); }
    } else { a = group_min_fused<G>(a); b = group_min_fused<G>(b); c = group_min_fused<G>(c); d = group_min_fused<G>(d); }
}
#undef MVS_DPP4
template <int G>
__device__ __forceinline__ uint32_t group_min_fused(uint32_t v) {
    v = min_step_u<1, G>(v); v = min_step_u<2, G>(v); v = min_step_u<4, G>(v);
    if (G >= 16) v = min_step_u<8, G>(v);
    if (G >= 32) v = min_step_u<16, G>(v);
    if (G >= 64) v = min_step_u<32, G>(v);
    return v;
}

// ---- setup ----
// Node classes: the sweep is routed PER NODE, not per solve.  A node of degree <= 3 whose own column and whose in-model
// neighbours' columns hold at most 255 labels is a FAST node, swept by mrf_sweep4_kernel<G> with the lane-group width of its
// neighbourhood: G = 8 / 16 / 32 / 64 for kmx = max(K_i, K_j) <= 32 / 64 / 128 / 255 (classes 0 .. 3; the node's three outgoing runs are
// as long as the NEIGHBOURS' label lists, so they count).  Everything else -- a non-manifold edge (degree > 3), a column of more
// than 255 labels at the node or next to it -- is a GENERIC node (class 4): one wave per node, any degree, any K.  One
// non-manifold edge or one long column therefore costs a handful of generic nodes, not the whole solve, and the lane-group
// width follows the local column sizes instead of the largest column of the mesh.  A colour class is an independent set, so
// the order in which its nodes are swept -- hence the split into launches -- cannot change the result.
constexpr uint32_t CLS_GENERIC = 4;
__device__ __forceinline__ uint32_t mrf_node_class(uint32_t kmx, uint32_t deg, uint32_t force_generic) {
    if (force_generic || deg > 3u || kmx > 255u) return CLS_GENERIC;
    return kmx <= 32u ? 0u : kmx <= 64u ? 1u : kmx <= 128u ? 2u : 3u;
}
// (also rev[e] = position of the reverse directed edge (j -> i as an in-edge of j) of the in-edge e = (i <- j), 0xFFFFFFFF when the input is
//  asymmetric: found once here, read by the layout kernels below instead of being searched for again by each of them)
__global__ void mrf_size_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                uint32_t F, uint32_t pad_mask, uint32_t force_generic, uint32_t* __restrict__ size, uint8_t* __restrict__ cls,
                                uint32_t* __restrict__ maxes /* [0]=kmax [1]=degmax */, uint32_t* __restrict__ rev) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = 0, deg = 0;
    if (i < F) {
        k = col_ptr[i + 1] - col_ptr[i];
        const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
        deg = e1 - e0;
        uint32_t kmx = k;
        // four edges at a time, level by level (neighbour -> its column / its list bounds -> the head of its list): a thread's loads of one
        // level are independent of each other -- edge after edge this was a chain of 3 x 4 dependent round trips per node (0.11 ms at C3)
        for (uint32_t eb = e0; eb < e1; eb += 4) {
            uint32_t j[4], kj[4], r0[4], r1[4], head[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) j[q] = (eb + q < e1) ? adj[eb + q] : i;
#pragma unroll
            for (int q = 0; q < 4; ++q) { kj[q] = col_ptr[j[q] + 1] - col_ptr[j[q]]; r0[q] = adj_ptr[j[q]]; r1[q] = adj_ptr[j[q] + 1]; }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) head[q][t] = (r0[q] + t < r1[q]) ? adj[r0[q] + t] : 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (eb + q >= e1) continue;
                const uint32_t e = eb + q;
                size[e] = (k > 0 && kj[q] > 0) ? ((k + pad_mask) & ~pad_mask) : 0u;   // runs padded to a multiple of 4 (8-byte quads) or 16 elements (32-byte sectors)
                if (k > 0) kmx = max(kmx, kj[q]);
                uint32_t r = 0xFFFFFFFFu;                         // the FIRST position of i in j's list
#pragma unroll
                for (int t = 3; t >= 0; --t) r = (head[q][t] == i) ? r0[q] + t : r;
                if (r == 0xFFFFFFFFu) { uint32_t rr = r0[q] + 4u; while (rr < r1[q] && adj[rr] != i) ++rr; if (rr < r1[q]) r = rr; }   // (degree > 4)
                rev[e] = r;
            }
        }
        cls[i] = (uint8_t)mrf_node_class(kmx, deg, force_generic);
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
// `orig` (null: identity) = the caller's id of every node: the keys are those of the CALLER's numbering, so the colouring -- hence the
// labeling -- does not depend on the order the library keeps the nodes in (ctx.h "the library's own mesh layout").
__global__ void mrf_colour_round_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const uint32_t* __restrict__ orig, uint32_t F,
                                        uint32_t* colour, uint32_t* __restrict__ pending) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    if (__hip_atomic_load(colour + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != NO_COLOUR) return;
    unsigned long long used = 0ull;
    bool ready = true;
    const uint32_t oi = orig ? orig[i] : i;
    const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
    for (uint32_t eb = e0; eb < e1 && ready; eb += 4) {      // four neighbours at a time: their ids, keys and colours are independent loads
        uint32_t j[4], oj[4], cj[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) j[t] = (eb + t < e1) ? adj[eb + t] : i;
#pragma unroll
        for (int t = 0; t < 4; ++t) { oj[t] = orig ? orig[j[t]] : j[t]; cj[t] = __hip_atomic_load(colour + j[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (j[t] == i || !mrf_key_less(oj[t], oi)) continue;   // (a padded slot is the node itself)
            if (cj[t] == NO_COLOUR) ready = false; else used |= 1ull << cj[t];
        }
    }
    if (ready) {
        // 64 colours in use around one node: the mask (and the 6-bit class sort) cannot hold a 65th -- reported, not wrapped
        if (used == ~0ull) { pending[1] = 1u; used = 0ull; }
        __hip_atomic_store(colour + i, (uint32_t)__builtin_ctzll(~used), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else *pending = 1u;                                       // racing stores of the same value
}
// Schedule order = nodes sorted by SUB-CLASS key: fast nodes first, by (colour, class, id) -- key = 4 * colour + class < 256 --
// then the generic nodes by (colour, id) -- key = 256 + colour.  Every (colour, class) pair is one contiguous range of the
// order = one launch; the fast nodes (their records and descriptors) are positions [0, n_fast).
constexpr uint32_t N_SUB = 320;            // sub-classes 0 .. 319 (64 colours x 4 fast classes, 64 generic colour classes)
constexpr uint32_t SUB_GENERIC = 256;
// Every sub-class has two ZONES (sort key = 2 * sub-class + zone): zone 0 = the nodes a sharded caller marked as its BOUNDARY nodes
// (own nodes with an edge into another rank's part: ctx->m_bnd, shard.hip), zone 1 = everybody else.  A sharded rank sweeps the
// boundary zone of a colour first, hands its runs to the neighbours, and sweeps the interior while they travel; a colour class is an
// independent set, so the split changes no value.  Without marks (single context, building blocks) zone 0 is empty and the order is
// the plain (colour, class, id) order.
constexpr uint32_t N_KEY = 2 * N_SUB;
__global__ void mrf_sortkey_kernel(const uint32_t* __restrict__ colour, const uint8_t* __restrict__ cls, const uint8_t* __restrict__ bnd, uint32_t F, uint32_t* __restrict__ key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < F) { const uint32_t c = colour[i], k = cls[i]; key[i] = 2u * ((k == CLS_GENERIC) ? SUB_GENERIC + c : 4u * c + k) + ((bnd && bnd[i]) ? 0u : 1u); }
}
// sub_begin[k] = first position of a key >= k in the sorted key array, k = 0 .. N_KEY
__global__ void mrf_sub_begin_kernel(const uint32_t* __restrict__ sorted, uint32_t F, uint32_t* __restrict__ sub_begin) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > N_KEY) return;
    uint32_t lo = 0, hi = F;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted[mid] < c) lo = mid + 1; else hi = mid; }
    sub_begin[c] = lo;
}
// own share of every (sub-class, zone): positions of the ids in [nb, ne) inside perm[sb[k], sb[k + 1]) (ids ascending inside a key)
__global__ void mrf_sub_range_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ sub_begin,
                                     uint32_t nb, uint32_t ne, uint32_t* __restrict__ out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N_KEY) return;
    const uint32_t cb = sub_begin[c], ce = sub_begin[c + 1];
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
// nsz[b * (F + 1) + q] = message elements node perm[q] sends to receivers of colour b (b < n_col; with n_col == 1 all
// receivers count as colour 0 = plain sender-major).  One exclusive scan over the n_col * (F + 1) entries then yields the
// layout  [receiver colour][sender in (colour, id) order][out-edge in list order]:
//   * what a phase WRITES (and re-reads for damping) is one contiguous stream per receiver colour;
//   * what a phase READS as incoming messages is the whole super-region of its own colour -- every byte of it is consumed
//     in that phase, by receivers that follow each other roughly in memory order, so no fetched line is wasted (a plain
//     sender-major layout gathers one 90-byte run out of each neighbour's 270-byte block: 2x the bytes at C3).
constexpr int MAX_LAYOUT_COLOURS = 8;
__global__ void mrf_nodesize_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ colour, const uint32_t* __restrict__ adj_ptr,
                                    const uint32_t* __restrict__ adj, const uint32_t* __restrict__ size, const uint32_t* __restrict__ rev, uint32_t F, uint32_t n_col,
                                    uint32_t* __restrict__ nsz) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > F) return;
    uint32_t acc[MAX_LAYOUT_COLOURS];
#pragma unroll
    for (int b = 0; b < MAX_LAYOUT_COLOURS; ++b) acc[b] = 0;
    if (q < F) {
        const uint32_t j = perm[q];
        const uint32_t ra = adj_ptr[j], rz = adj_ptr[j + 1];
        for (uint32_t rb = ra; rb < rz; rb += 4) {           // four out-edges at a time: the loads of a level are independent
            uint32_t i[4], e[4], b[4], sz[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { const bool on = rb + t < rz; i[t] = on ? adj[rb + t] : j; e[t] = on ? rev[rb + t] : 0xFFFFFFFFu; }   // e = (i <- j): the in-edge of the receiver that owns this run
#pragma unroll
            for (int t = 0; t < 4; ++t) { b[t] = (n_col > 1) ? colour[i[t]] : 0u; sz[t] = (e[t] != 0xFFFFFFFFu) ? size[e[t]] : 0u; }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int c = 0; c < MAX_LAYOUT_COLOURS; ++c) acc[c] += ((uint32_t)c == b[t]) ? sz[t] : 0u;
        }
    }
#pragma unroll
    for (int b = 0; b < MAX_LAYOUT_COLOURS; ++b) if ((uint32_t)b < n_col) nsz[(size_t)b * (F + 1) + q] = acc[b];
}
__global__ void mrf_inoff_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ colour, const uint32_t* __restrict__ adj_ptr,
                                 const uint32_t* __restrict__ adj, const uint32_t* __restrict__ size, const uint32_t* __restrict__ rev, const uint32_t* __restrict__ noff,
                                 uint32_t F, uint32_t n_col, uint32_t* __restrict__ in_off) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= F) return;
    const uint32_t j = perm[q];
    uint32_t off[MAX_LAYOUT_COLOURS];
#pragma unroll
    for (int b = 0; b < MAX_LAYOUT_COLOURS; ++b) off[b] = ((uint32_t)b < n_col) ? noff[(size_t)b * (F + 1) + q] : 0u;
    const uint32_t ra = adj_ptr[j], rz = adj_ptr[j + 1];
    for (uint32_t rb = ra; rb < rz; rb += 4) {               // four out-edges at a time (see mrf_nodesize_kernel); offsets are handed out in list order
        uint32_t i[4], e[4], b[4], sz[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const bool on = rb + t < rz; i[t] = on ? adj[rb + t] : j; e[t] = on ? rev[rb + t] : 0xFFFFFFFFu; }
#pragma unroll
        for (int t = 0; t < 4; ++t) { b[t] = (n_col > 1) ? colour[i[t]] : 0u; sz[t] = (e[t] != 0xFFFFFFFFu) ? size[e[t]] : 0u; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (e[t] == 0xFFFFFFFFu) continue;
            uint32_t o = 0;
#pragma unroll
            for (int c = 0; c < MAX_LAYOUT_COLOURS; ++c) { if ((uint32_t)c == b[t]) { o = off[c]; off[c] += sz[t]; } }
            in_off[e[t]] = o;
        }
    }
}

__global__ void mrf_edge_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                uint32_t F, const uint32_t* __restrict__ in_off, const uint32_t* __restrict__ size, const uint32_t* __restrict__ rev, MrfEdge* __restrict__ edge) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
    for (uint32_t eb = e0; eb < e1; eb += 4) {               // four edges at a time: the loads of a level are independent
        uint32_t j[4], r[4], io[4], sz[4], oo[4], kj[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const bool on = eb + t < e1; j[t] = on ? adj[eb + t] : i; r[t] = on ? rev[eb + t] : 0xFFFFFFFFu; io[t] = on ? in_off[eb + t] : 0u; sz[t] = on ? size[eb + t] : 0u; }
#pragma unroll
        for (int t = 0; t < 4; ++t) { oo[t] = (r[t] != 0xFFFFFFFFu) ? in_off[r[t]] : 0u; kj[t] = col_ptr[j[t] + 1] - col_ptr[j[t]]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (eb + t >= e1) continue;
            const bool has = r[t] != 0xFFFFFFFFu;
            MrfEdge m;
            m.in_off = MSG_BASE + io[t];                       // [0, MSG_BASE) is the reserved zero / identity run
            m.out_off = has ? MSG_BASE + oo[t] : 0u;
            m.kj = (sz[t] > 0 && has) ? kj[t] : 0u;
            edge[eb + t] = m;
        }
    }
}

// map[in_off(e) + t] = position of L_i[t] in L_j (binary search; lists ascending, calculate_data_costs.cpp:272), for the in-edges
// e = (i <- j) whose SENDER j is a generic node: the generic sweep kernel re-aligns its outgoing runs through these 16-bit maps
// (fast senders carry byte maps in their records).
constexpr uint32_t MAP_TILE = 256;   // neighbour lists up to this length are searched in LDS
__global__ void __launch_bounds__(256) mrf_map_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const uint32_t* __restrict__ adj_ptr,
                               const uint32_t* __restrict__ adj, uint32_t F, const MrfEdge* __restrict__ edge, const uint8_t* __restrict__ cls, uint16_t* __restrict__ map) {
    // 16 lanes per node.  The binary search is a chain of dependent loads: through global memory it is latency bound,
    // so the neighbour's list is first copied (coalesced) into the group's LDS tile.  A group never spans
    // waves and LDS operations of a wave execute in order, so the tile needs no barrier.
    __shared__ uint16_t s_l[16][MAP_TILE];
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t gl = threadIdx.x & 15;
    if (i >= F) return;
    uint16_t* tile = s_l[threadIdx.x >> 4];
    const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0;
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        const MrfEdge m = edge[e];
        if (m.kj == 0 || cls[adj[e]] != CLS_GENERIC) continue;  // group-uniform
        const uint32_t q0 = col_ptr[adj[e]];
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

// ---- fast nodes (degree <= 3, columns of the neighbourhood <= 255 labels): per-node RECORDS + 48-byte descriptors ----
// The sweep streams, per node and in (colour, id) order, ONE read-only record instead of gathering from the CSR:
//   [ceil4(K) label words: cost code << 16 | view id]  [for every out-edge whose two label lists differ: ceil4(K_j) map bytes]
// padded to a multiple of four words.  A phase reads a contiguous run of records: no partially used lines (a phase used to
// read every third column of the CSR), the view id of the decoded label is already in a register (no gather), the unaries
// are 16-bit fixed point (2 instead of 4 bytes; part of the solver's definition, restated in oracle/oracle.cpp), the
// re-alignment maps are bytes (K <= 255; 0xFF = label absent at the sender) and exist only where the lists differ.
constexpr uint32_t REC_BASE = 256;            // words [0, REC_BASE) of the record array are zero: where masked lanes load
constexpr float COST_SCALE = 65535.0f;
__device__ __forceinline__ uint32_t cost_code(float c) { return (uint32_t)(c * COST_SCALE + 0.5f); }   // c in [0, 1]
__device__ __forceinline__ float cost_value(uint32_t code) { return (float)code * (1.0f / 65535.0f); }
// position of label `l` (> 0: view id l - 1) in the ascending view list of a column (calculate_data_costs.cpp:272); a labeling only ever
// holds labels of the node's own column, so the search finds it
__device__ __forceinline__ uint32_t label_position(const uint16_t* __restrict__ view_id, uint32_t p0, uint32_t K, uint32_t l) {
    uint32_t lo = 0, hi = K;
    const uint32_t key = l - 1u;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)view_id[p0 + mid] < key) lo = mid + 1; else hi = mid; }
    return lo < K ? lo : K - 1u;
}

// ident[e] = 1 iff the two label lists of the (valid) directed edge e are identical; 16 lanes per node.  The kernel is a chain
// of dependent gathers (edge -> neighbour -> its column -> its view ids), so three edges are in flight at a time.
// (Round 5 tried the symmetry -- of a pair only the node with the smaller id compares the lists and writes both flags: half the list
// gathers, but three more dependent look-ups per edge in front of them: 0.375 -> 0.52 ms.  Reverted.)
__global__ void __launch_bounds__(256) mrf_ident_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const uint32_t* __restrict__ adj_ptr,
                                                        const uint32_t* __restrict__ adj, uint32_t F, const MrfEdge* __restrict__ edge, const uint8_t* __restrict__ cls,
                                                        uint8_t* __restrict__ ident) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t gl = threadIdx.x & 15;
    if (i >= F || cls[i] == CLS_GENERIC) return;               // only fast nodes have records
    const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0;
    const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
    for (uint32_t e = e0; e < e1; e += 3) {
        uint32_t kj[3], q0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool on = e + k < e1;
            kj[k] = on ? edge[e + k].kj : 0u;
            q0[k] = col_ptr[on ? adj[e + k] : i];
        }
        uint32_t same = 0;   // bit k: edge e + k still looks identical
#pragma unroll
        for (int k = 0; k < 3; ++k) same |= (kj[k] != 0u && kj[k] == K) ? (1u << k) : 0u;
        if (same) {          // group-uniform
            for (uint32_t t = gl; t < K; t += 16) {
                const uint32_t v = view_id[p0 + t];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if ((same >> k) & 1u) { if (view_id[q0[k] + t] != v) same &= ~(1u << k); }
            }
            for (int o = 8; o > 0; o >>= 1) same &= __shfl_xor(same, o, 16);
        }
        if (gl < 3 && e + gl < e1 && kj[gl] != 0u) ident[e + gl] = (uint8_t)((same >> gl) & 1u);
    }
}
// rsz[q] = words of the record of node perm[q] (rsz[F] = 0)
// qpos[i] = position of node i in the (colour, id) order (the inverse of perm)
// (wide: class-1 nodes are swept by mrf_sweep8_kernel, whose lanes read 8 map bytes at once: their map sections are padded to 8 bytes)
__global__ void mrf_recsize_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ adj_ptr, const MrfEdge* __restrict__ edge,
                                   const uint8_t* __restrict__ ident, const uint8_t* __restrict__ cls, const uint32_t* __restrict__ perm, uint32_t F, uint32_t wide,
                                   uint32_t* __restrict__ rsz, uint32_t* __restrict__ qpos) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > F) return;
    uint32_t w = 0;
    if (q < F) {
        const uint32_t i = perm[q], K = col_ptr[i + 1] - col_ptr[i];
        qpos[i] = q;
        if (K && cls[i] != CLS_GENERIC) {
            w = (K + 3u) & ~3u;
            const bool w8 = wide != 0u && cls[i] == 1u;
            const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
            for (uint32_t eb = e0; eb < e1; eb += 4) {       // four edges at a time: independent loads
                uint32_t kj[4], id[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { const bool on = eb + t < e1; kj[t] = on ? edge[eb + t].kj : 0u; id[t] = on ? ident[eb + t] : 1u; }
#pragma unroll
                for (int t = 0; t < 4; ++t) if (kj[t] && !id[t]) w += w8 ? 2u * ((kj[t] + 7u) >> 3) : (kj[t] + 3u) >> 2;
            }
            w = (w + 3u) & ~3u;
        }
    }
    rsz[q] = w;
}
// fills the record of node i: 16 lanes per node, nodes in FACE order -- the columns are read as one sequential stream (in the
// (colour, id) order of the records a pass would touch every fourth column), the records are written where they belong;
// the node's own list sits in an LDS tile for the searches (a group never spans waves and LDS operations of a wave
// execute in order: no barrier)
__global__ void __launch_bounds__(256) mrf_record_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                         const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const MrfEdge* __restrict__ edge,
                                                         const uint8_t* __restrict__ ident, const uint8_t* __restrict__ cls, const uint32_t* __restrict__ qpos, const uint32_t* __restrict__ roff,
                                                         uint32_t F, uint32_t wide, uint32_t* __restrict__ rec) {
    __shared__ uint16_t s_l[16][256];
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t gl = threadIdx.x & 15;
    if (i >= F) return;
    const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0;
    const uint32_t ci = cls[i];
    if (K == 0 || ci == CLS_GENERIC) return;
    // A map byte is the SLOT of the sweep kernel's LDS tile that holds the sender's label: the tile is row-major over r = label mod 4
    // (slot = (at & 3) * G + (at >> 2) for position `at` in the sender's list, G = 8 << class lanes per node), so that the lanes of a
    // group read consecutive banks.  "Label absent at the sender" is the +inf slot 4 * G; for G = 64 the slots fill the byte range
    // 0 .. 254 and 0xFF marks absence (the kernel steers it to slot 256).
    // (a WIDE class-1 node -- mrf_sweep8_kernel: 8 lanes x 8 labels -- has the tile row-major over label mod 8: slot = (at & 7) * 8 + (at >> 3),
    //  "absent" = slot 64, and its map sections padded to 8 bytes)
    const bool w8 = wide != 0u && ci == 1u;
    const uint32_t rs = w8 ? 8u : 8u << ci, none_byte = w8 ? 64u : (ci < 3u ? 4u * rs : 0xFFu);
    const uint32_t lmask = w8 ? 7u : 3u, lshift = w8 ? 3u : 2u;
    const uint32_t q = qpos[i];
    uint16_t* tile = s_l[threadIdx.x >> 4];
    uint32_t* out = rec + REC_BASE + roff[q];
    const uint32_t K4 = (K + 3u) & ~3u;
    // the (at most three: a fast node) edges' sizes, flags and neighbour columns are requested before the column is read, not one edge
    // after the other behind it
    const uint32_t e0 = adj_ptr[i], deg = adj_ptr[i + 1] - e0;
    uint32_t kj3[3], q03[3];
    {
        uint32_t kj[3], idn[3], nb[3];
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) { const bool on = d < deg; kj[d] = on ? edge[e0 + d].kj : 0u; idn[d] = on ? (uint32_t)ident[e0 + d] : 1u; nb[d] = on ? adj[e0 + d] : i; }
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) { kj3[d] = (kj[d] != 0u && idn[d] == 0u) ? kj[d] : 0u; q03[d] = col_ptr[nb[d]]; }
    }
    for (uint32_t t = gl; t < K4; t += 16) {
        uint32_t w = 0u;
        if (t < K) { const uint32_t v = view_id[p0 + t]; tile[t] = (uint16_t)v; w = (cost_code(cost[p0 + t]) << 16) | v; }
        out[t] = w;
    }
    uint32_t pos = K4;
#pragma unroll
    for (uint32_t d = 0; d < 3; ++d) {
        const uint32_t kj = kj3[d];
        if (kj == 0) continue;                                 // group-uniform: not in the model, or identical lists (no map)
        const uint32_t q0 = q03[d], nw = w8 ? 2u * ((kj + 7u) >> 3) : (kj + 3u) >> 2;
        for (uint32_t wI = gl; wI < nw; wI += 16) {
            // positions of the RECEIVER's labels 4 wI .. 4 wI + 3 in this (the sender's) list: four lower-bound searches in
            // lockstep (the step count depends on K only), so their LDS reads are independent
            uint32_t key[4], lo[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) { const uint32_t t2 = 4u * wI + r; key[r] = (t2 < kj) ? (uint32_t)view_id[q0 + t2] : 0xFFFFFFFFu; }
            for (uint32_t n = K; n > 1u;) {
                const uint32_t half = n >> 1;
#pragma unroll
                for (uint32_t r = 0; r < 4; ++r) lo[r] += ((uint32_t)tile[lo[r] + half - 1u] < key[r]) ? half : 0u;
                n -= half;
            }
            uint32_t word = 0u;
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) {
                // lo = the last candidate position: the key sits there, or one further (beyond the list), or nowhere
                uint32_t at = lo[r] + (((uint32_t)tile[lo[r]] < key[r]) ? 1u : 0u);
                const uint32_t byte = (at < K && (uint32_t)tile[at < K ? at : 0u] == key[r]) ? (at & lmask) * rs + (at >> lshift) : none_byte;
                word |= byte << (8 * r);
            }
            out[pos + wI] = word;
        }
        pos += nw;
    }
    for (uint32_t t = pos + gl; t < ((pos + 3u) & ~3u); t += 16) out[t] = 0u;
}
__global__ void mrf_desc_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                const MrfEdge* __restrict__ edge, const uint8_t* __restrict__ ident, const uint32_t* __restrict__ perm,
                                const uint32_t* __restrict__ colour, const uint32_t* __restrict__ roff, uint32_t n_fast, NodeDesc* __restrict__ desc) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;   // position in the schedule order; the fast nodes come first
    if (q >= n_fast) return;
    const uint32_t i = perm[q];
    NodeDesc nd;
    const uint32_t k = col_ptr[i + 1] - col_ptr[i];
    nd.rec = k ? REC_BASE + roff[q] : 0u; nd.id = i; nd.kk = k;
    const uint32_t e0 = adj_ptr[i], deg = adj_ptr[i + 1] - e0, ci = colour[i];
    for (int d = 0; d < 3; ++d) {
        MrfEdge m; m.in_off = 0; m.out_off = 0; m.kj = 0;
        uint32_t flag_ident = 0, flag_low = 0, nb = i;          // absent neighbour: the node itself (a valid index; masked by kj = 0 wherever it is used)
        if ((uint32_t)d < deg) { nb = adj[e0 + d]; flag_low = (colour[nb] < ci) ? 1u : 0u; }
        if ((uint32_t)d < deg && k > 0) {
            m = edge[e0 + d];
            // the message written over out-edge d is aligned with the neighbour's list: identity iff the lists are equal
            // (a symmetric property, so the in-edge's flag serves)
            if (m.kj && ident[e0 + d]) flag_ident = 1u;
            if (m.kj == 0) { m.in_off = 0; m.out_off = 0; }    // edge not in the model: the sweep reads the reserved zero run
        }
        nd.in_off[d] = m.in_off | flag_low; nd.out_off[d] = m.out_off | flag_ident; nd.kk |= m.kj << (8 + 8 * d); nd.nbr[d] = nb;
    }
    desc[q] = nd;
}

// ---- one colour phase of a sweep; fast path: degree <= 3, K <= 255 (and K <= 4 * G) ----
// 4 labels per lane: lane gl owns labels 4*gl .. 4*gl+3 (one 16-byte load = four label words, one 4-byte access = four
// 8-bit messages or four map bytes), so a 64-lane wave sweeps 64/G nodes per iteration at roughly the instruction count
// of one.  The layout is arranged so that NO per-label masking of the loads is needed:
//   * elements [0, MSG_BASE) of the message buffer are zero for ever: an absent edge (degree < 3, or an empty
//     neighbour column) and every lane beyond the node's labels read their "incoming message" there; words
//     [0, REC_BASE) of the record array are zero as well (label / map words of masked lanes);
//   * an edge whose two label lists are identical (flag in the descriptor; 3/4 of the edges on the synthetic scenes)
//     has no map at all: the identity bytes 4*gl .. 4*gl+3 are a per-lane constant;
//   * the re-alignment gather c[p] goes through a per-group LDS tile with one extra slot holding +inf: 0xFF ("label
//     absent at the sender") is steered onto that slot, so it needs no compare -- fmin(inf - cmin, 1/rho) = 1/rho;
//   * message runs are padded to multiples of 4 elements, so a lane stores all four of its values or none.
// Values a lane computes for label slots beyond the column are garbage that never reaches a valid label: they are
// excluded from min / argmin by the ok[] mask (the only per-label selects left) and land in run padding.
// LDS operations of a wave execute in order and a lane group never spans waves, so the tile needs no barrier.
//
// Decode: the lane that owns the winning label writes the node's sel / label / unary into decode buffer st->w.
// Energy: E = sum_i D_i(l_i) + sum_(i,j) [l_i != l_j] is accumulated HERE, in 32.32 fixed point: the node adds its unary
// and one cut per model edge to a neighbour of a LOWER colour whose label differs -- that neighbour was swept in an
// earlier phase of this sweep, so its label is final, and every edge has exactly one higher-coloured end.  Integer sums:
// the per-block partials (partial[2 * block]) add up to the oracle's energy of the sweep whatever the launch geometry.
// (The access-pattern probes of rounds 4 and 5 -- scripts/sweep_probe.py: all accesses inside a 64 KB window, arithmetic removed, one stream
// dropped at a time, an edge-pair message layout, wider map loads -- are a PATCH on this kernel, scripts/probe/sweep4_probes.patch, which
// scripts/build_variant.py --patch applies to a copy of this file; the product source carries none of them.)
template <int G, bool DAMP, bool XCD, bool LATE_OLD>
__global__ void __launch_bounds__(256) mrf_sweep4_kernel(const NodeDesc* __restrict__ desc, const uint32_t* __restrict__ rec, msg_t* msg,
                                                         const mvs_mrf_progress* __restrict__ st, uint32_t* lab2, uint32_t buf_stride,
                                                         uint32_t node_begin /* positions in the (colour, id) order */, uint32_t node_end, float rho, float alpha,
                                                         unsigned long long* __restrict__ partial) {
    // in place: the nodes of one launch share a colour (an independent set), so no run is read by one node and
    // written by another; a node reads its old outgoing run before it overwrites it
    const msg_t* mo = msg; msg_t* mn = msg;
    constexpr int NPB = 256 / G;
    // Tile of a lane group: the 4G cavity values ROW-major over r = label mod 4 (label 4 * gl + r at slot r * G + gl) + the +inf slot 4G.
    // A ds_read_b32 / ds_write_b32 serves lanes 0-31 and 32-63 in one LDS cycle each when they hit 32 different banks (bank = word
    // address mod 32): with this layout the lanes of a group touch consecutive banks (exactly for identical label lists, 3/4 of the
    // edges), and the stride puts the 32 / G groups of a half-wave on disjoint banks (TS = G mod 32 for G < 32).  The label-major tile
    // (slot = label, one float4 write) made every gather a 4-way bank conflict: PMC SQ_LDS_BANK_CONFLICT = 56 % of the LDS cycles.
    // (One tile per group, reused by the three out-edges.  A variant with three tiles -- all writes first, then all twelve gathers, one
    // LDS round trip per node and 9 VGPRs fewer -- measured the same at C3 and 28 % slower where a quarter of the nodes take the
    // 64-lane class: profiles/EXPERIMENTS.md.)
    constexpr int TS = G == 8 ? 40 : G == 16 ? 80 : 4 * G + 4;
    __shared__ float s_c[NPB * TS];
    __shared__ unsigned long long s_e[8];
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    float* __restrict__ tile = s_c + grp * TS;
    // The stop rule runs on the device (mrf_step_kernel) while the host queues sweeps ahead of the reports it reads: a sweep queued
    // after the rule fired changes nothing anybody reads (the best labeling is frozen, the step kernel ignores its energy) -- it ends here
    if (st->stopped) return;
    if (gl == 0) tile[4 * G] = INFINITY;
    __syncthreads();
    const uint32_t wofs = st->w * buf_stride;                // decode buffer of this sweep (flipped by the step kernel when a sweep improves the best energy)
    uint32_t* lab = lab2 + wofs;
    const float lam = 1.0f / rho;
    const MsgQ mq = msg_q(lam);
    const float kappa = rho * mq.step, nstep = -mq.step, oms = (1.0f - alpha) * mq.scale, lam_s = lam * oms;   // the oracle's constants, fp32
    constexpr float HUGE_COST = 1e30f;                       // unary of the label slots beyond the column: never a minimum
    const uint32_t stride = gridDim.x * NPB;
    uint32_t vb = blockIdx.x;
    if (XCD && (gridDim.x & 7u) == 0u) vb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    uint32_t i = node_begin + vb * NPB + grp;
    const uint32_t last = node_end - 1;
    const uint32_t n_iter = (node_end - node_begin + stride - 1) / stride;
    const uint32_t t0 = 4u * gl;
    const uint32_t ident_word = 0x03020100u * (uint32_t)G + 0x01010101u * (uint32_t)gl;   // map bytes of an identical-list edge: the slots r * G + gl of the lane's own labels
    // Software pipeline, two nodes deep: while a lane group computes node `it`, the loads of node `it + 1` (label words,
    // three incoming runs, three map words, three neighbour labels and -- damped sweeps -- the three old outgoing runs)
    // and the descriptor of node `it + 2` are in flight.  The kernel had been latency bound: waves parked on memory 65 %
    // of their time with the vector unit 60 % busy and HBM at 3.5 TB/s (PMC, profiles/).  Same-colour nodes never touch
    // each other's runs or labels, so reading a node's inputs one iteration early cannot observe a write of this launch.
    // No address needs a select: a lane beyond the column / the neighbour's column reads whatever follows the run or
    // the record (valid memory -- both arrays carry slack -- and lines its neighbours fetch anyway) and its values never
    // reach a valid label; an absent edge has offsets 0 in the descriptor, i.e. the reserved zero run.
    struct Raw { uint4 lw; uint32_t in[3], map[3], nl[3], old[3]; };
    const uint32_t t0b = 4u * t0, glb = 4u * (uint32_t)gl;   // byte offsets of the lane's label words / map word inside a record
    auto issue = [&](const NodeDesc& d, Raw& r) {
        const uint32_t K = d.kk & 0xFFu, recb = 4u * d.rec;
        r.lw = ld_off<uint4>(rec, recb + t0b);
        uint32_t mposb = recb + 4u * ((K + 3u) & ~3u) + glb;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            r.in[e] = ld_off<uint32_t>(mo, (d.in_off[e] & ~3u) + t0);
            r.map[e] = ld_off<uint32_t>(rec, mposb);
            if (!(d.out_off[e] & 1u)) mposb += (((d.kk >> (8 + 8 * e)) & 0xFFu) + 3u) & ~3u;   // 4 bytes per 4 map entries
            r.nl[e] = ld_off<uint32_t>(lab, 4u * d.nbr[e]);     // an absent neighbour is recorded as the node itself
            if (DAMP) r.old[e] = ld_off<uint32_t>(mo, (d.out_off[e] & ~3u) + t0); else r.old[e] = d.out_off[e];
        }
    };
    NodeDesc cur = ld_off<NodeDesc>(desc, (uint32_t)sizeof(NodeDesc) * min(i, last));
    Raw rw; issue(cur, rw);
    NodeDesc nxt = ld_off<NodeDesc>(desc, (uint32_t)sizeof(NodeDesc) * min(i + stride, last));
    uint32_t acc_e = 0u, acc_c = 0u;                         // sum of the decoded labels' cost codes (< 2^32 per thread: <= 65535 per node) and cut edges
#pragma unroll 2
    for (uint32_t it = 0; it < n_iter; ++it, i += stride) {
        const bool node_ok = i < node_end;
        Raw rn; issue(nxt, rn);                                // node it + 1 (clamped to the last descriptor past the end: harmless reads)
        const NodeDesc nn = ld_off<NodeDesc>(desc, (uint32_t)sizeof(NodeDesc) * min(i + 2u * stride, last));  // node it + 2
        const uint32_t K = node_ok ? (cur.kk & 0xFFu) : 0u;
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ok[r] = t0 + r < K;
        uint32_t kj3[3], o_out[3]; bool ident[3], low[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            kj3[d] = node_ok ? ((cur.kk >> (8 + 8 * d)) & 0xFFu) : 0u;
            ident[d] = (cur.out_off[d] & 1u) != 0u; low[d] = (cur.in_off[d] & 1u) != 0u && kj3[d] != 0u;
            o_out[d] = cur.out_off[d] & ~3u;
        }
        const uint32_t lw[4] = {rw.lw.x, rw.lw.y, rw.lw.z, rw.lw.w};
        const uint32_t* r_in = rw.in; const uint32_t* r_map = rw.map; const uint32_t* nl = rw.nl; const uint32_t* r_old = rw.old;
        // The update on the 8-bit codes (oracle.cpp mrf_sweep is the definition): Sc = sum of the incoming codes (exact),
        // b = fma(rho * step, Sc, D), cs_e = fma(-step, code_e, b) * oms -- the reweighted cavity D + rho * sum_all - m_e in
        // damped code units, oms = (1 - alpha) * scale -- code' = rne(fma(old, alpha, min(cs[p] - cmin, lam * oms))).
        float cf[3][4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int d = 0; d < 3; ++d) cf[d][r] = (float)((r_in[d] >> (8 * r)) & 0xFFu);   // v_cvt_f32_ubyte<r>
            const float D = ok[r] ? cost_value(lw[r] >> 16) : HUGE_COST;
            b[r] = __builtin_fmaf(kappa, (cf[0][r] + cf[1][r]) + cf[2][r], D);
        }
        // the three cavities and the four group minima (of b and of each cavity) in one interleaved reduction
        float cv[3][4];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[d][r] = __builtin_fmaf(nstep, cf[d][r], b[r]) * oms;
        float gm = fminf(fminf(b[0], b[1]), fminf(b[2], b[3]));
        float cm0 = fminf(fminf(cv[0][0], cv[0][1]), fminf(cv[0][2], cv[0][3])), cm1 = fminf(fminf(cv[1][0], cv[1][1]), fminf(cv[1][2], cv[1][3])),
              cm2 = fminf(fminf(cv[2][0], cv[2][1]), fminf(cv[2][2], cv[2][3]));
        group_min_fused4<G>(gm, cm0, cm1, cm2);
        const float cmin3[3] = {cm0, cm1, cm2};
        // decode: first argmin_t b[t] -- the group minimum, then the smallest label attaining it (== the sequential
        // "first minimum": comparisons are exact)
        uint32_t bt = (b[3] == gm) ? t0 + 3u : 0xFFFFFFFFu;
        bt = (b[2] == gm) ? t0 + 2u : bt; bt = (b[1] == gm) ? t0 + 1u : bt; bt = (b[0] == gm) ? t0 : bt;
        bt = group_min_fused<G>(bt);                          // every lane of the group holds the winner
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            // the tile holds cs - cmin of edge d (the same subtraction the oracle performs after its gather); the extra slot stays +inf
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[r * G + gl] = cv[d][r] - cmin3[d];
            const uint32_t mw = ident[d] ? ident_word : r_map[d];
            uint32_t w = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t mp = (mw >> (8 * r)) & 0xFFu;
                const uint32_t slot = (G < 64) ? mp : ((mp == 0xFFu) ? (uint32_t)(4 * G) : mp);   // "absent at the sender" -> the +inf slot (G < 64: the records hold 4 * G)
                w = msg_pack_s<DAMP>(min_raw(tile[slot], lam_s), alpha, (float)((r_old[d] >> (8 * r)) & 0xFFu), (uint32_t)r, w);
            }
            if (t0 < kj3[d]) st_off<uint32_t>(mn, o_out[d] + t0, w);      // one 4-byte store (runs are padded)
        }
        // the lane that owns the winning label publishes the decode (K == 0: lane 0 publishes the single label 0 with
        // unary 1, view_selection.cpp:50-51,70-71) and accounts the node's share of the tracking energy (integer:
        // cost codes + 65535 per cut edge, oracle.cpp mrf_energy_sel)
        const bool owner = node_ok && ((K > 0u) ? ((bt >> 2) == (uint32_t)gl) : (gl == 0));
        if (owner) {
            const uint32_t r = bt & 3u;
            const uint32_t wsel = (r == 0u) ? lw[0] : (r == 1u) ? lw[1] : (r == 2u) ? lw[2] : lw[3];
            const uint32_t my_lab = (K > 0u) ? (wsel & 0xFFFFu) + 1u : 0u;
            const uint32_t my_code = (K > 0u) ? (wsel >> 16) : 65535u;
            const uint32_t idb = 4u * cur.id;
            // ONE store: the label.  The position in the column and the unary of the decoded label are not kept per sweep: whoever
            // needs them -- the polish, the reported energy -- derives them from the label once, after the sweeps (mrf_exact_cost_kernel)
            st_off<uint32_t>(lab, idb, my_lab);
            acc_e += my_code;
            acc_c += (low[0] && nl[0] != my_lab) + (low[1] && nl[1] != my_lab) + (low[2] && nl[2] != my_lab);
        }
        cur = nxt; rw = rn; nxt = nn;
    }
    // one partial pair per block (no atomics: same-address atomics serialise at ~12 ns each)
    unsigned long long e = acc_e, c = acc_c;
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_xor(e, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_e[threadIdx.x >> 6] = e; s_e[4 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        e = s_e[0] + s_e[1] + s_e[2] + s_e[3]; c = s_e[4] + s_e[5] + s_e[6] + s_e[7];
        partial[2 * blockIdx.x] = e + 65535ull * c; partial[2 * blockIdx.x + 1] = c;
    }
}


// ---- the same update with EIGHT labels per lane (class 1: neighbourhood columns of 33 .. 64 labels; option "mrf_wide") ----
// mrf_sweep4_kernel spends one memory instruction on 4 bytes of a run per lane: a 64-lane wave sweeps 4 nodes of this class per iteration
// and issues 20 loads / stores for them.  The probes of rounds 4 and 5 (EXPERIMENTS.md) say the kernel is bound by the NUMBER of memory
// instructions and requests per node, not by their bytes.  Here a node takes 8 lanes and a lane 8 consecutive labels: the three incoming
// runs, the three old outgoing runs, the three map words and the three stores are 8-byte accesses, the label words two 16-byte loads, and
// a wave sweeps 8 nodes per iteration with 21 memory instructions -- half the instructions per node, the same bytes, the same lines.
// Arithmetic, reduction results (min / first argmin are exact in any order), sums (adjacency order per label) and the message / record /
// decode layouts are those of mrf_sweep4_kernel: bit-identical results; only the map BYTES differ (a byte is the slot of THIS kernel's
// LDS tile: row-major over label mod 8, slot = (at & 7) * 8 + (at >> 3), "absent" = slot 64 -- mrf_record_kernel writes them per class).
template <bool DAMP, bool XCD>
__global__ void __launch_bounds__(256) mrf_sweep8_kernel(const NodeDesc* __restrict__ desc, const uint32_t* __restrict__ rec, msg_t* msg,
                                                         const mvs_mrf_progress* __restrict__ st, uint32_t* lab2, uint32_t buf_stride,
                                                         uint32_t node_begin, uint32_t node_end, float rho, float alpha,
                                                         unsigned long long* __restrict__ partial) {
    const msg_t* mo = msg; msg_t* mn = msg;                    // in place: one colour per launch (see mrf_sweep4_kernel)
    constexpr int G = 8, L = 8, NPB = 256 / G;
    constexpr int TS = 72;                                     // 8 x 8 slots + the +inf slot; 72 mod 32 = 8: the 4 groups of a half-wave sit on disjoint banks
    __shared__ float s_c[NPB * TS];
    __shared__ unsigned long long s_e[8];
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    float* __restrict__ tile = s_c + grp * TS;
    if (st->stopped) return;
    if (gl == 0) tile[L * G] = INFINITY;
    __syncthreads();
    const uint32_t wofs = st->w * buf_stride;
    uint32_t* lab = lab2 + wofs;
    const float lam = 1.0f / rho;
    const MsgQ mq = msg_q(lam);
    const float kappa = rho * mq.step, nstep = -mq.step, oms = (1.0f - alpha) * mq.scale, lam_s = lam * oms;
    constexpr float HUGE_COST = 1e30f;
    const uint32_t stride = gridDim.x * NPB;
    uint32_t vb = blockIdx.x;
    if (XCD && (gridDim.x & 7u) == 0u) vb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    uint32_t i = node_begin + vb * NPB + grp;
    const uint32_t last = node_end - 1;
    const uint32_t n_iter = (node_end - node_begin + stride - 1) / stride;
    const uint32_t t0 = (uint32_t)L * gl;
    // map bytes of an identical-list edge: the slots r * G + gl of the lane's own labels r = 0 .. 7
    const uint32_t ident_lo = 0x18100800u + 0x01010101u * (uint32_t)gl, ident_hi = 0x38302820u + 0x01010101u * (uint32_t)gl;
    struct Raw { uint4 lw0, lw1; uint2 in[3], map[3], old[3]; uint32_t nl[3]; };
    auto issue = [&](const NodeDesc& d, Raw& r) {
        const uint32_t K = d.kk & 0xFFu, recb = 4u * d.rec;
        r.lw0 = ld_off<uint4>(rec, recb + 32u * (uint32_t)gl);
        r.lw1 = ld_off<uint4>(rec, recb + 32u * (uint32_t)gl + 16u);
        uint32_t mposb = recb + 4u * ((K + 3u) & ~3u) + 8u * (uint32_t)gl;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            r.in[e] = ld_off<uint2>(mo, (d.in_off[e] & ~3u) + t0);
            r.map[e] = ld_off<uint2>(rec, mposb);
            if (!(d.out_off[e] & 1u)) mposb += (((d.kk >> (8 + 8 * e)) & 0xFFu) + 7u) & ~7u;   // (a wide node's map sections are padded to 8 bytes)
            r.nl[e] = ld_off<uint32_t>(lab, 4u * d.nbr[e]);
            if (DAMP) r.old[e] = ld_off<uint2>(mo, (d.out_off[e] & ~3u) + t0); else r.old[e] = make_uint2(0u, 0u);
        }
    };
    NodeDesc cur = ld_off<NodeDesc>(desc, (uint32_t)sizeof(NodeDesc) * min(i, last));
    Raw rw; issue(cur, rw);
    NodeDesc nxt = ld_off<NodeDesc>(desc, (uint32_t)sizeof(NodeDesc) * min(i + stride, last));
    uint32_t acc_e = 0u, acc_c = 0u;
    for (uint32_t it = 0; it < n_iter; ++it, i += stride) {
        const bool node_ok = i < node_end;
        Raw rn; issue(nxt, rn);
        const NodeDesc nn = ld_off<NodeDesc>(desc, (uint32_t)sizeof(NodeDesc) * min(i + 2u * stride, last));
        const uint32_t K = node_ok ? (cur.kk & 0xFFu) : 0u;
        uint32_t kj3[3], o_out[3]; bool ident[3], low[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            kj3[d] = node_ok ? ((cur.kk >> (8 + 8 * d)) & 0xFFu) : 0u;
            ident[d] = (cur.out_off[d] & 1u) != 0u; low[d] = (cur.in_off[d] & 1u) != 0u && kj3[d] != 0u;
            o_out[d] = cur.out_off[d] & ~3u;
        }
        const uint32_t lw[8] = {rw.lw0.x, rw.lw0.y, rw.lw0.z, rw.lw0.w, rw.lw1.x, rw.lw1.y, rw.lw1.z, rw.lw1.w};
        // (the update on the 8-bit codes: see mrf_sweep4_kernel; oracle.cpp mrf_sweep is the definition)
        float b[8], cv[3][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float cf[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) cf[d] = (float)((((r < 4) ? rw.in[d].x : rw.in[d].y) >> (8 * (r & 3))) & 0xFFu);
            const float D = (t0 + r < K) ? cost_value(lw[r] >> 16) : HUGE_COST;
            b[r] = __builtin_fmaf(kappa, (cf[0] + cf[1]) + cf[2], D);
#pragma unroll
            for (int d = 0; d < 3; ++d) cv[d][r] = __builtin_fmaf(nstep, cf[d], b[r]) * oms;
        }
        float gm = fminf(fminf(fminf(b[0], b[1]), fminf(b[2], b[3])), fminf(fminf(b[4], b[5]), fminf(b[6], b[7])));
        float cm[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) cm[d] = fminf(fminf(fminf(cv[d][0], cv[d][1]), fminf(cv[d][2], cv[d][3])), fminf(fminf(cv[d][4], cv[d][5]), fminf(cv[d][6], cv[d][7])));
        group_min_fused4<G>(gm, cm[0], cm[1], cm[2]);
        // decode: first argmin_t b[t] (the group minimum, then the smallest label attaining it)
        uint32_t bt = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 7; r >= 0; --r) bt = (b[r] == gm) ? t0 + (uint32_t)r : bt;
        bt = group_min_fused<G>(bt);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int r = 0; r < 8; ++r) tile[r * G + gl] = cv[d][r] - cm[d];
            const uint32_t mlo = ident[d] ? ident_lo : rw.map[d].x, mhi = ident[d] ? ident_hi : rw.map[d].y;
            uint32_t w0 = 0u, w1 = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t s0 = (mlo >> (8 * r)) & 0xFFu, s1 = (mhi >> (8 * r)) & 0xFFu;   // slots (64 = the +inf slot: "absent at the sender")
                w0 = msg_pack_s<DAMP>(min_raw(tile[s0], lam_s), alpha, (float)((rw.old[d].x >> (8 * r)) & 0xFFu), (uint32_t)r, w0);
                w1 = msg_pack_s<DAMP>(min_raw(tile[s1], lam_s), alpha, (float)((rw.old[d].y >> (8 * r)) & 0xFFu), (uint32_t)r, w1);
            }
            if (t0 < kj3[d]) st_off<uint2>(mn, o_out[d] + t0, make_uint2(w0, w1));   // one 8-byte store: with "mrf_wide" the runs are padded to 8 elements

        }
        const bool owner = node_ok && ((K > 0u) ? ((bt >> 3) == (uint32_t)gl) : (gl == 0));
        if (owner) {
            const uint32_t r = bt & 7u;
            uint32_t wsel = lw[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) wsel = (r == (uint32_t)k) ? lw[k] : wsel;
            const uint32_t my_lab = (K > 0u) ? (wsel & 0xFFFFu) + 1u : 0u;
            const uint32_t my_code = (K > 0u) ? (wsel >> 16) : 65535u;
            st_off<uint32_t>(lab, 4u * cur.id, my_lab);
            acc_e += my_code;
            acc_c += (low[0] && rw.nl[0] != my_lab) + (low[1] && rw.nl[1] != my_lab) + (low[2] && rw.nl[2] != my_lab);
        }
        cur = nxt; rw = rn; nxt = nn;
    }
    unsigned long long e = acc_e, c = acc_c;
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_xor(e, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_e[threadIdx.x >> 6] = e; s_e[4 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        e = s_e[0] + s_e[1] + s_e[2] + s_e[3]; c = s_e[4] + s_e[5] + s_e[6] + s_e[7];
        partial[2 * blockIdx.x] = e + 65535ull * c; partial[2 * blockIdx.x + 1] = c;
    }
}

// generic nodes: any degree, any K.  One BLOCK per node (persistent blocks striding over the range): the four waves split the
// labels for the belief vector b (which goes through a global scratch row: K is unbounded here) and then take the out-edges in
// turn, one edge per wave at a time -- a generic node is a chain of dependent gathers (edge record -> run -> map -> run), and
// with a handful of generic nodes per colour phase (one non-manifold edge, one long column) the launch lasts as long as ONE
// node's chain, so the chain is what is parallelised.  Same arithmetic as the fast kernel / the oracle; the node's share of the
// sweep's tracking energy (cost code of the decoded label + one cut per model edge to a lower-coloured neighbour whose label
// differs) goes into per-block partials like the fast kernel's.
template <bool DAMP>
__global__ void __launch_bounds__(256) mrf_sweep_generic_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                                const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, const MrfEdge* __restrict__ edge,
                                                                const uint16_t* __restrict__ map, const uint32_t* __restrict__ colour,
                                                                msg_t* msg, const uint32_t* __restrict__ perm, const mvs_mrf_progress* __restrict__ st,
                                                                uint32_t* lab2, uint32_t buf_stride,
                                                                float* scratch, uint32_t node_begin, uint32_t node_end, float rho, float alpha,
                                                                unsigned long long* __restrict__ partial) {
    const msg_t* mo = msg; msg_t* mn = msg;                  // in place (one colour per launch)
    __shared__ float s_b[4]; __shared__ uint32_t s_t[4]; __shared__ uint32_t s_cuts[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (st->stopped) return;                                 // a sweep queued after the stop rule fired (see mrf_sweep4_kernel)
    const uint32_t wofs = st->w * buf_stride;
    uint32_t* lab = lab2 + wofs;
    const float lam = 1.0f / rho;
    const MsgQ mq = msg_q(lam);
    const float kappa = rho * mq.step, nstep = -mq.step, oms = (1.0f - alpha) * mq.scale, lam_s = lam * oms;
    unsigned long long acc_e = 0ull, acc_c = 0ull;           // thread 0 accumulates
    for (uint32_t q = node_begin + blockIdx.x; q < node_end; q += gridDim.x) {   // block-uniform
        const uint32_t i = perm[q];
        const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0;
        if (K == 0) {   // the single label 0 with unary 1 (view_selection.cpp:50-51,70-71): cost code 65535, no model edge
            if (threadIdx.x == 0) { lab[i] = 0u; acc_e += 65535ull; }
            continue;
        }
        const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
        float bb = INFINITY; uint32_t bt = 0xFFFFFFFFu;
        for (uint32_t t = threadIdx.x; t < K; t += 256) {
            float Sc = 0.0f;
            for (uint32_t e = e0; e < e1; ++e) { const MrfEdge m = edge[e]; if (m.kj) Sc = Sc + (float)mo[m.in_off + t]; }   // adjacency order, like the oracle
            const float b = __builtin_fmaf(kappa, Sc, cost_value(cost_code(cost[p0 + t])));   // the unaries as the sweeps see them: 16-bit fixed point
            scratch[p0 + t] = b;
            if (b < bb) { bb = b; bt = t; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(bb, o, 64); const uint32_t ot = __shfl_xor(bt, o, 64);
            if (ob < bb || (ob == bb && ot < bt)) { bb = ob; bt = ot; }
        }
        if (lane == 0) { s_b[wave] = bb; s_t[wave] = bt; }
        __syncthreads();                                     // also orders the scratch row: written above, read below by other threads of the block
#pragma unroll
        for (int w = 0; w < 4; ++w) { const float ob = s_b[w]; const uint32_t ot = s_t[w]; if (w == 0 || ob < bb || (ob == bb && ot < bt)) { bb = ob; bt = ot; } }
        const uint32_t my_lab = (uint32_t)view_id[p0 + bt] + 1u, my_code = cost_code(cost[p0 + bt]);
        // cut edges to lower-coloured neighbours (their labels of this sweep are final): one edge per thread
        {
            const uint32_t ci = colour[i];
            uint32_t cuts = 0;
            for (uint32_t e = e0 + threadIdx.x; e < e1; e += 256) {
                const uint32_t j = adj[e];
                if (edge[e].kj != 0u && colour[j] < ci && lab[j] != my_lab) ++cuts;
            }
            for (int o = 32; o > 0; o >>= 1) cuts += __shfl_xor(cuts, o, 64);
            if (lane == 0) s_cuts[wave] = cuts;
        }
        // the out-edges, one per wave at a time
        for (uint32_t e = e0 + (uint32_t)wave; e < e1; e += 4) {
            const MrfEdge m = edge[e];
            if (!m.kj) continue;  // wave-uniform
            float cmin = INFINITY;
            for (uint32_t t = lane; t < K; t += 64) cmin = fminf(cmin, __builtin_fmaf(nstep, (float)mo[m.in_off + t], scratch[p0 + t]) * oms);
            for (int o = 32; o > 0; o >>= 1) cmin = fminf(cmin, __shfl_xor(cmin, o, 64));
            for (uint32_t t2 = lane; t2 < m.kj; t2 += 64) {
                const uint16_t p = map[m.out_off + t2];
                const float raw = (p == MAP_NONE) ? lam_s : fminf(__builtin_fmaf(nstep, (float)mo[m.in_off + p], scratch[p0 + p]) * oms - cmin, lam_s);
                mn[m.out_off + t2] = (msg_t)msg_pack_s<DAMP>(raw, alpha, (float)mo[m.out_off + t2], 0u, 0u);
            }
        }
        __syncthreads();                                     // s_cuts complete; s_b / s_t free for the next node
        if (threadIdx.x == 0) {
            lab[i] = my_lab;   // (position and unary of the label: derived after the sweeps, see mrf_sweep4_kernel)
            acc_e += my_code; acc_c += (unsigned long long)s_cuts[0] + s_cuts[1] + s_cuts[2] + s_cuts[3];
        }
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = acc_e + 65535ull * acc_c; partial[2 * blockIdx.x + 1] = acc_c; }
}

// ---- exact energy of a labeling (32.32 fixed point) over nodes [node_begin, node_end) ----
// needs only the labels and selected unary costs: an edge is in the model iff both labels are
// non-zero (both columns non-empty, view_selection.cpp:29-42)
__global__ void __launch_bounds__(256) mrf_energy_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                         const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                                         const uint32_t* __restrict__ lab, const float* __restrict__ selcost,
                                                         const mvs_mrf_progress* __restrict__ st /* non-null: the decode buffer st->w of lab / selcost */, uint32_t buf_stride,
                                                         uint32_t node_begin, uint32_t node_end, unsigned long long* __restrict__ out /* [0] energy, [1] cuts */) {
    // st != null: the TRACKING energy of the current decode (its unaries are dequantised 16-bit codes: integer units of
    // 1 / 65535, 65535 per cut edge); else the exact 32.32 fixed-point energy of the labeling given
    // (the sweeps keep only labels: the tracking energy looks the label's unary up in the table -- the 16-bit code the records hold)
    if (st) { const uint32_t wofs = st->w * buf_stride; lab += wofs; }
    unsigned long long unary = 0, cuts = 0;
    for (uint32_t i = node_begin + blockIdx.x * blockDim.x + threadIdx.x; i < node_end; i += gridDim.x * blockDim.x) {
        const uint32_t li = lab[i];
        if (st) { const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0; unary += (K && li) ? (unsigned long long)cost_code(cost[p0 + label_position(view_id, p0, K, li)]) : 65535ull; }
        else unary += fix32(selcost[i]);
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
        out[2 * blockIdx.x] = st ? u + 65535ull * c : u + (c << 32); out[2 * blockIdx.x + 1] = c;
    }
}
__global__ void __launch_bounds__(1024) mrf_energy_reduce_kernel(const unsigned long long* __restrict__ partial, uint32_t n_blocks,
                                                                 unsigned long long* __restrict__ out /* [0] energy, [1] cuts */,
                                                                 unsigned long long* __restrict__ out2 = nullptr /* a second copy (sharded solves: the slot the peers read) */) {
    unsigned long long e = 0, c = 0;
    const ulonglong2* pp = reinterpret_cast<const ulonglong2*>(partial);
#pragma unroll 8
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 1024u) { const ulonglong2 v = pp[b]; e += v.x; c += v.y; }
    __shared__ unsigned long long su[16], sc[16];
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_xor(e, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { su[threadIdx.x >> 6] = e; sc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) { e = 0; c = 0; for (int k = 0; k < 16; ++k) { e += su[k]; c += sc[k]; } out[0] = e; out[1] = c; if (out2) { out2[0] = e; out2[1] = c; } }
}

// ---- ICM polish: G lanes per node over its labels ----
template <int G>
__global__ void __launch_bounds__(256) mrf_icm_gain_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                           const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                                           const uint32_t* __restrict__ sel, const uint32_t* __restrict__ lab,
                                                           uint32_t node_begin, uint32_t node_end, float* __restrict__ gain, uint32_t* __restrict__ cand,
                                                           const uint32_t* __restrict__ list /* null: nodes [node_begin, node_end); else node ids list[node_begin .. node_end) */,
                                                           const uint32_t* __restrict__ list_count /* non-null: node_end = node_begin + *list_count */) {
    constexpr int NPB = 256 / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    if (list_count) node_end = node_begin + *list_count;   // the active list's length stays on the device: no read-back between the ICM rounds
    for (uint32_t pos = node_begin + blockIdx.x * NPB + grp; pos < node_end; pos += gridDim.x * NPB) {   // trip counts are uniform within a lane group
        const uint32_t i = list ? list[pos] : pos;
        const uint32_t p0 = col_ptr[i];
        const uint32_t K = col_ptr[i + 1] - p0;
        const uint32_t e0 = adj_ptr[i], e1 = adj_ptr[i + 1];
        const uint32_t cur_l = (K > 0) ? lab[i] : 0u;   // the node's current label (labels of a column are distinct: it names one position)
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
            if (l == cur_l) cur = en;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, G); const uint32_t ot = __shfl_xor(bt, o, G);
            if (ob < best || (ob == best && ot < bt)) { best = ob; bt = ot; }
            cur += __shfl_xor(cur, o, G);   // exactly one lane holds a non-zero term (or none: 0)
        }
        if (gl == 0) { gain[i] = (K > 0) ? (cur - best) : 0.0f; cand[i] = (K > 0) ? bt : 0u; }
    }
}

// fast path (degree <= 3): persistent lane groups over the node descriptors, next descriptor prefetched
template <int G>
__global__ void __launch_bounds__(256) mrf_icm_gain_desc_kernel(const NodeDesc* __restrict__ desc, const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
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
        const uint32_t K = node_ok ? (cur.kk & 0xFFu) : 0u;
        const uint32_t id = cur.id;
        const uint32_t p0 = (K > 0) ? col_ptr[id] : 0u;
        const uint32_t cur_l = (K > 0) ? lab[id] : 0u;
        uint32_t nl[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) nl[d] = (K > 0 && ((cur.kk >> (8 + 8 * d)) & 0xFFu)) ? lab[cur.nbr[d]] : 0u;
        float best = INFINITY, cur_e = 0.0f; uint32_t bt = 0xFFFFFFFFu;
        for (uint32_t t = gl; t < K; t += G) {
            const uint32_t l = (uint32_t)view_id[p0 + t] + 1u;
            const uint32_t diff = (nl[0] != 0u && nl[0] != l) + (nl[1] != 0u && nl[1] != l) + (nl[2] != 0u && nl[2] != l);
            const float en = cost[p0 + t] + (float)diff;
            if (en < best) { best = en; bt = t; }
            if (l == cur_l) cur_e = en;
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
                                                            uint32_t* __restrict__ alist /* null, or: nodes whose gain has to be re-evaluated */,
                                                            const uint32_t* __restrict__ orig /* null: identity; ties between equal gains go to the smaller id of the CALLER's numbering */) {
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
                if (gj > gi || (gj == gi && (orig ? orig[j] < orig[i] : j < i))) win = false;
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
                                  uint32_t* __restrict__ labels, uint32_t* __restrict__ bad_unseen /* [0] bad, [1] unseen */,
                                  const uint32_t* __restrict__ orig /* non-null: labels[orig[i]] = label of node i (the caller's numbering) */,
                                  const uint32_t* __restrict__ foreign /* may be null: labels outside their column, counted by mrf_exact_cost_kernel over the same range */) {
    const uint32_t i = node_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_end) return;
    if (foreign && i == node_begin && *foreign) atomicAdd(&bad_unseen[0], *foreign);
    const uint32_t label = lab[i];
    if (label > n_views) atomicAdd(&bad_unseen[0], 1u);       /* :126-128 "Incorrect labeling" */
    if (label == 0u) atomicAdd(&bad_unseen[1], 1u);            /* :129 */
    labels[orig ? orig[i] : i - node_begin] = label;
}


// ---- device-side solver bookkeeping: one block.  Same decisions, in the same arithmetic, as the host loop it
// replaces: best = min(best, e); stop iff sweep >= min_sweeps, sweep > window and
// double(hist[sweep - window] - best) < double(min_improvement) * double(hist[sweep - window]).
// With `partial` (single-GPU loop) the block first sums the per-block energy pairs the sweep kernels (fast path) or the
// energy kernel left behind and publishes them in energy_out; sharded callers pass the all-reduced pair in `energy`.
// "Keep the best labeling" is a flip of two indices: the sweep that improved the best energy wrote decode buffer w, which
// becomes best_w, and the next sweeps write the other buffer -- nothing is copied.  The report goes straight into the
// pinned ring slot (host memory), so a step is ONE launch.
// The step's number (which ring slot its report goes to, which sequence number announces it) is device state too -- ctl[0] counts
// the steps of the solve, ctl[1] is the solve's sequence base -- so a step's launch has the SAME arguments every time and the
// sweep loop can be replayed from a hipGraph.
__global__ void __launch_bounds__(1024) mrf_step_kernel(mvs_mrf_progress* __restrict__ st, unsigned long long* __restrict__ hist,
                                const unsigned long long* __restrict__ energy, const unsigned long long* __restrict__ partial, uint32_t n_partial,
                                unsigned long long* __restrict__ energy_out, mvs_mrf_progress* __restrict__ ring, uint32_t* __restrict__ ring_seq, uint32_t ring_slots,
                                uint32_t* __restrict__ ctl, int max_sweeps, int min_sweeps, int window, float min_improvement,
                                const unsigned long long* const* __restrict__ peer_tab = nullptr, uint32_t n_peer = 0, uint32_t peer_off = 0) {
    // peer_tab (sharded solves, peer-push transport): the energy pair of the sweep = the sum over the ranks' published pairs
    // peer_tab[q][peer_off], [peer_off + 1] -- read here, so that summing them is no launch of its own
    __shared__ unsigned long long su[16], sc[16];
    unsigned long long e_sum = 0, c_sum = 0;
    if (partial) {
        // up to 8192 pairs: 1024 threads, the loads of a thread independent of each other (a serial 256-thread loop was
        // a chain of 32 dependent round trips: 15 us per sweep)
        const ulonglong2* pp = reinterpret_cast<const ulonglong2*>(partial);
#pragma unroll 8
        for (uint32_t b = threadIdx.x; b < n_partial; b += 1024u) { const ulonglong2 v = pp[b]; e_sum += v.x; c_sum += v.y; }
        for (int o = 32; o > 0; o >>= 1) { e_sum += __shfl_xor(e_sum, o, 64); c_sum += __shfl_xor(c_sum, o, 64); }
        if ((threadIdx.x & 63) == 0) { su[threadIdx.x >> 6] = e_sum; sc[threadIdx.x >> 6] = c_sum; }
        __syncthreads();
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (partial) {
        e_sum = 0; c_sum = 0;
        for (int k = 0; k < 16; ++k) { e_sum += su[k]; c_sum += sc[k]; }
        energy_out[0] = e_sum; energy_out[1] = c_sum;
    } else if (peer_tab) {
        e_sum = 0; c_sum = 0;
        for (uint32_t q = 0; q < n_peer; ++q) { e_sum += peer_tab[q][peer_off]; c_sum += peer_tab[q][peer_off + 1u]; }
        energy_out[0] = e_sum; energy_out[1] = c_sum;
    }
    mvs_mrf_progress p = *st;
    if (p.stopped) { p.improved = 0u; }
    else {
        const uint32_t sw = p.sweep + 1u;
        const unsigned long long e0 = (partial || peer_tab) ? e_sum : energy[0];
        const bool imp = e0 < p.best;
        if (imp) { p.best = e0; p.best_w = p.w; p.w ^= 1u; }
        p.sweep = sw; p.improved = imp ? 1u : 0u; p.energy = e0;
        hist[sw] = p.best;
        bool stop = false;
        if ((int)sw >= min_sweeps && (int)sw > window) {
            const unsigned long long prev = hist[sw - (uint32_t)window];
            stop = (double)(prev - p.best) < (double)min_improvement * (double)prev;
        }
        if ((int)sw >= max_sweeps) stop = true;
        if (stop) { p.stopped = 1u; p.stop_sweep = sw; }
    }
    *st = p;
    const uint32_t n = ctl[0] + 1u; ctl[0] = n;
    if (ring) {   // pinned host memory: the report, a system-scope fence, then its sequence number -- the host polls the number
        const uint32_t slot = n % ring_slots;
        ring[slot] = p;
        __threadfence_system();
        __hip_atomic_store(ring_seq + slot, ctl[1] + n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// one value from device memory into a pinned slot, announced the same way (ICM "moved" counts)
__global__ void report_u32_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t* __restrict__ dst_seq, uint32_t seq) {
    *dst = *src;
    __threadfence_system();
    __hip_atomic_store(dst_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// n <= 64 words from device memory into a pinned buffer, announced like a report (read_words)
__global__ void report_words_kernel(const uint32_t* __restrict__ src, uint32_t n, uint32_t* __restrict__ dst, uint32_t* __restrict__ dst_seq, uint32_t seq) {
    if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(dst_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// best labeling := the current decode buffer (mvs_ctx_mrf_keep_best): the same index flip, unconditionally
__global__ void mrf_flip_kernel(mvs_mrf_progress* __restrict__ st) { st->best_w = st->w; st->w ^= 1u; }

}  // namespace

constexpr uint32_t EPART_BLOCKS = 2048;   // per colour phase: upper bound of the blocks of all its launches together (resident blocks)

// Builds the solver's edge tables for the active CSR (ctx->r_ptr / r_view / r_cost) and adjacency.
void mrf_setup(mvs_ctx* ctx, const mvs_mrf_params* params) {
    hipStream_t s = ctx->stream;
    const uint32_t F = ctx->csr_faces;
    ctx->m_params = *params;
    if (!(params->rho > 0.0f && params->rho <= 1.0f) || !(params->damping >= 0.0f && params->damping < 1.0f))
        throw StatusError(MVS_ERR_INVALID, "mrf params: need 0 < rho <= 1, 0 <= damping < 1");
    // window < 1 would index the energy history out of bounds (hist[sweep - window]); the others are nonsensical when negative
    if (params->window < 1 || params->min_sweeps < 0 || !(params->min_improvement >= 0.0f) || params->icm_iters < 0)
        throw StatusError(MVS_ERR_INVALID, "mrf params: need window >= 1, min_sweeps >= 0, min_improvement >= 0, icm_iters >= 0");
    uint32_t E = ctx->r_adj_edges;
    if (!ctx->r_adj_edges_known) {   // lists handed over on the device and used as they are: their length is only known there
        MVS_HIP(hipMemcpyAsync(&E, ctx->r_adj_ptr + F, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MVS_HIP(hipStreamSynchronize(s));
    }
    ctx->m_size.ensure((size_t)E + 2); ctx->m_edge.ensure((size_t)E + 1); ctx->m_moved.ensure(8 + 2 * 64);
    ctx->m_n_adj = E;
    uint32_t* maxes = ctx->m_moved.p + 4;
    MVS_HIP(hipMemsetAsync(ctx->m_moved.p, 0, 8 * sizeof(uint32_t), s));
    MVS_HIP(hipMemsetAsync(ctx->m_size.p, 0, ((size_t)E + 2) * sizeof(uint32_t), s));
    const unsigned nb = (F + 255) / 256;
    // degenerate inputs (fewer than four table entries, no edge at all) take the generic kernel throughout; "mrf_force_generic" is a test hook
    const uint32_t force_generic = (ctx->csr_nnz < 4 || E == 0 || ctx->mrf_force_generic) ? 1u : 0u;
    ctx->m_cls.ensure((size_t)F + 4);
    ctx->m_rev.ensure((size_t)E + 2);
    if (F) { hipLaunchKernelGGL(mrf_size_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_adj_ptr, ctx->r_adj, F, (uint32_t)(ctx->mrf_run_pad == 16 ? 15 : (ctx->mrf_wide ? 7 : 3)), force_generic, ctx->m_size.p, ctx->m_cls.p, maxes, ctx->m_rev.p); MVS_LAUNCH_CHECK(); }
    // ---- colour-phased schedule: colouring (Jones-Plassmann rounds), nodes in (colour, id) order ----
    ctx->m_colour.ensure((size_t)F + 2); ctx->m_perm.ensure((size_t)F + 2); ctx->m_tmp_a.ensure((size_t)MAX_LAYOUT_COLOURS * ((size_t)F + 1) + 72); ctx->m_tmp_b.ensure((size_t)MAX_LAYOUT_COLOURS * ((size_t)F + 1) + 2); ctx->m_tmp_c.ensure((size_t)F + 2);
    ctx->m_colours = 0; ctx->m_sub_begin.assign(N_KEY + 1, 0); ctx->m_n_fast = 0; ctx->m_range_q.clear(); ctx->m_range_nb = ctx->m_range_ne = 0;
    ctx->m_sweep_no = 0;
    if (F) {
        uint32_t* pending = ctx->m_moved.p + 1;   // [0] a node is still waiting, [1] a node saw all 64 colours around it
        hipLaunchKernelGGL(mrf_colour_init_kernel, dim3(nb), dim3(256), 0, s, ctx->m_colour.p, ctx->m_tmp_a.p /* iota */, F); MVS_LAUNCH_CHECK();
        // a sharded caller hands in the colouring it kept from its first solve (the colours follow from the adjacency alone, which a
        // shard pins at creation, like its layout): no rounds
        if (ctx->m_colour_in) MVS_HIP(hipMemcpyAsync(ctx->m_colour.p, ctx->m_colour_in, (size_t)F * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        for (int round = 0; !ctx->m_colour_in; ++round) {
            if (round >= 4096) throw StatusError(MVS_ERR_HIP, "graph colouring did not terminate");
            MVS_HIP(hipMemsetAsync(pending, 0, sizeof(uint32_t), s));
            // a batch of rounds per read-back: 16 first (meshes of 0.2 - 2 M faces need 12 - 16 rounds; a round past the eighth costs 4 - 10 us -- coloured
            // nodes leave at their first load -- a read-back 25 us of an idle device), then 4
            for (int k = 0; k < (round == 0 ? 16 : 4); ++k) { hipLaunchKernelGGL(mrf_colour_round_kernel, dim3(nb), dim3(256), 0, s, ctx->r_adj_ptr, ctx->r_adj, ctx->t_perm, F, ctx->m_colour.p, pending); MVS_LAUNCH_CHECK(); }
            uint32_t hp[2] = {0, 0};   // set if any round of the batch left a node waiting
            read_words(ctx, pending, hp, 2);
            if (hp[1]) throw StatusError(MVS_ERR_UNSUPPORTED, "adjacency graph needs more than 64 colours (a node with >= 64 mutually constrained neighbours)");
            if (!hp[0]) break;
        }
        MVS_HIP(hipMemsetAsync(ctx->m_moved.p, 0, 4 * sizeof(uint32_t), s));
        // stable sort of the node ids by sub-class key: perm = the schedule order; every (colour, class) pair is a contiguous range
        hipLaunchKernelGGL(mrf_sortkey_kernel, dim3(nb), dim3(256), 0, s, ctx->m_colour.p, ctx->m_cls.p, ctx->m_bnd, F, ctx->m_tmp_c.p); MVS_LAUNCH_CHECK();
        size_t tmp_bytes = 0;
        MVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ctx->m_tmp_c.p, ctx->m_tmp_b.p, ctx->m_tmp_a.p, ctx->m_perm.p, F, 0, 10, s));
        ctx->sort_tmp.ensure(tmp_bytes + 16);
        MVS_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->m_tmp_c.p, ctx->m_tmp_b.p, ctx->m_tmp_a.p, ctx->m_perm.p, F, 0, 10, s));
        ctx->m_sub.ensure(3 * (size_t)N_KEY + 8);   // [0, N_KEY]: sub_begin; behind it the own-share ranges of sub_range()
        hipLaunchKernelGGL(mrf_sub_begin_kernel, dim3((N_KEY + 256) / 256), dim3(256), 0, s, ctx->m_tmp_b.p, F, ctx->m_sub.p); MVS_LAUNCH_CHECK();
        read_block(ctx, ctx->m_sub.p, ctx->m_sub_begin.data(), (N_KEY + 1) * sizeof(uint32_t));
        const std::vector<uint32_t>& sb = ctx->m_sub_begin;
        uint32_t C = 0;   // colours 0 .. C-1 are in use (greedy colours are dense)
        for (uint32_t c = 0; c < 64; ++c) if (sb[2 * (4 * c + 4)] > sb[2 * (4 * c)] || sb[2 * (SUB_GENERIC + c + 1)] > sb[2 * (SUB_GENERIC + c)]) C = c + 1;
        ctx->m_colours = C;
        ctx->m_n_fast = sb[2 * SUB_GENERIC];
    }
    // message layout (sender-major, (colour, id) node order): in_off[e] for every directed edge e (adjacency order);
    // edges whose reverse is missing (asymmetric input) keep offset 0 and are disabled by mrf_edge_kernel
    DBuf<uint32_t>& in_off = ctx->m_sel2;  // temporary home, re-ensured below
    in_off.ensure(std::max<size_t>((size_t)E + 2, (size_t)F + 2));
    MVS_HIP(hipMemsetAsync(in_off.p, 0, ((size_t)E + 2) * sizeof(uint32_t), s));
    uint32_t h[3] = {0, 0, 0};
    if (F) {
        // the offsets are 32 bits wide: the total is bounded by 3 (degree) x (entries + 3 pad per column) on a manifold; beyond
        // 2^32 the scan below would wrap silently, so large inputs get an exact 64-bit total first
        if ((uint64_t)E * 4ull + 4ull * (uint64_t)ctx->csr_nnz >= 0xFFFFFFF0ull) {
            const uint64_t total = sum_u32(ctx, ctx->m_size.p, (size_t)E + 1);
            if ((uint64_t)MSG_BASE + total >= 0xFFFFFFF0ull) throw StatusError(MVS_ERR_UNSUPPORTED, "message array exceeds 2^32 elements (shard the graph: DESIGN.md multi-GPU)");
        }
        const uint32_t n_col = (ctx->m_colours >= 2 && ctx->m_colours <= (uint32_t)MAX_LAYOUT_COLOURS) ? ctx->m_colours : 1u;
        const size_t n_ent = (size_t)n_col * ((size_t)F + 1);
        ctx->m_tmp_a.ensure(n_ent + 72); ctx->m_tmp_b.ensure(n_ent + 2);
        hipLaunchKernelGGL(mrf_nodesize_kernel, dim3((F + 256) / 256), dim3(256), 0, s, ctx->m_perm.p, ctx->m_colour.p, ctx->r_adj_ptr, ctx->r_adj, ctx->m_size.p, ctx->m_rev.p, F, n_col, ctx->m_tmp_a.p); MVS_LAUNCH_CHECK();
        exclusive_scan_u32(ctx, ctx->m_tmp_a.p, ctx->m_tmp_b.p, n_ent, nullptr);   // the last entry of every colour segment is 0, so scan[last] = total
        hipLaunchKernelGGL(mrf_inoff_kernel, dim3(nb), dim3(256), 0, s, ctx->m_perm.p, ctx->m_colour.p, ctx->r_adj_ptr, ctx->r_adj, ctx->m_size.p, ctx->m_rev.p, ctx->m_tmp_b.p, F, n_col, in_off.p); MVS_LAUNCH_CHECK();
        MVS_HIP(hipMemcpyAsync(maxes + 2, ctx->m_tmp_b.p + (n_ent - 1), sizeof(uint32_t), hipMemcpyDeviceToDevice, s));   // (next to the two maxima: one read-back)
    } else MVS_HIP(hipMemsetAsync(maxes + 2, 0, sizeof(uint32_t), s));
    { uint32_t hm[3]; read_words(ctx, maxes, hm, 3); h[1] = hm[0]; h[2] = hm[1]; h[0] = hm[2]; }
    ctx->m_total = (uint64_t)MSG_BASE + h[0]; ctx->m_kmax = h[1]; ctx->m_degmax = h[2];
    if (ctx->m_total >= 0xFFFFFFF0ull) throw StatusError(MVS_ERR_UNSUPPORTED, "message array exceeds 2^32 elements");
    if (F) { hipLaunchKernelGGL(mrf_edge_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_adj_ptr, ctx->r_adj, F, in_off.p, ctx->m_size.p, ctx->m_rev.p, ctx->m_edge.p); MVS_LAUNCH_CHECK(); }
    ctx->m_ident.ensure((size_t)E + 1);
    MVS_HIP(hipMemsetAsync(ctx->m_ident.p, 0, (size_t)E + 1, s));
    ctx->m_fast = true;   // the sweep kernels of BOTH node classes accumulate the sweep's energy (callers no longer run the energy kernel per sweep)
    ctx->m_wide_layout = ctx->mrf_wide != 0;   // the records / runs built below are those of this variant: the sweeps follow the set-up, not the option
    const uint32_t n_fast = ctx->m_n_fast, n_generic = F - n_fast;
    if (n_fast) {
        // records + descriptors of the fast nodes.  Upper bound of the record array (no read-back): labels nnz + 3 F, maps <= one byte per message element
        const size_t rec_cap = (size_t)REC_BASE + ctx->csr_nnz + 8 * (size_t)F + ctx->m_total / 4 + (ctx->mrf_wide ? (size_t)ctx->m_n_adj : 0) + 1024;   // incl. slack for reads past the last record
        ctx->m_rec.ensure(rec_cap);
        MVS_HIP(hipMemsetAsync(ctx->m_rec.p, 0, REC_BASE * sizeof(uint32_t), s));
        hipLaunchKernelGGL(mrf_ident_kernel, dim3((unsigned)(((size_t)F * 16 + 255) / 256)), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_adj_ptr, ctx->r_adj, F, ctx->m_edge.p, ctx->m_cls.p, ctx->m_ident.p); MVS_LAUNCH_CHECK();
        uint32_t* rsz = ctx->m_tmp_a.p; uint32_t* roff = ctx->m_tmp_b.p;   // F + 1 entries each
        hipLaunchKernelGGL(mrf_recsize_kernel, dim3((F + 256) / 256), dim3(256), 0, s, ctx->r_ptr, ctx->r_adj_ptr, ctx->m_edge.p, ctx->m_ident.p, ctx->m_cls.p, ctx->m_perm.p, F, (uint32_t)(ctx->mrf_wide ? 1 : 0), rsz, ctx->m_tmp_c.p); MVS_LAUNCH_CHECK();
        exclusive_scan_u32(ctx, rsz, roff, (size_t)F + 1, nullptr);
        hipLaunchKernelGGL(mrf_record_kernel, dim3((unsigned)(((size_t)F * 16 + 255) / 256)), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->r_adj_ptr, ctx->r_adj,
                           ctx->m_edge.p, ctx->m_ident.p, ctx->m_cls.p, ctx->m_tmp_c.p /* qpos */, roff, F, (uint32_t)(ctx->mrf_wide ? 1 : 0), ctx->m_rec.p); MVS_LAUNCH_CHECK();
        ctx->m_desc.ensure((size_t)F + 1);
        hipLaunchKernelGGL(mrf_desc_kernel, dim3((n_fast + 255) / 256), dim3(256), 0, s, ctx->r_ptr, ctx->r_adj_ptr, ctx->r_adj, ctx->m_edge.p, ctx->m_ident.p, ctx->m_perm.p, ctx->m_colour.p, roff, n_fast, ctx->m_desc.p); MVS_LAUNCH_CHECK();
    }
    if (n_generic) {
        // 16-bit re-alignment maps of the runs the generic nodes SEND (indexed like the messages); only those elements are ever read
        ctx->m_map.ensure(ctx->m_total + 8);
        hipLaunchKernelGGL(mrf_map_kernel, dim3((unsigned)(((size_t)F * 16 + 255) / 256)), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_adj_ptr, ctx->r_adj, F, ctx->m_edge.p, ctx->m_cls.p, ctx->m_map.p); MVS_LAUNCH_CHECK();
        ctx->pq.ensure(ctx->csr_nnz + 1);   // scratch row per generic node (the data-cost work buffer is free by now)
    }
    ctx->m_msg_a.ensure(ctx->m_total + 1024);   // slack: lanes beyond a run read on (up to 4 * 64 elements)
    MVS_HIP(hipMemsetAsync(ctx->m_msg_a.p, 0, (ctx->m_total + 1024) * sizeof(msg_t), s));   // zero codes, incl. the reserved zero run
    // in_off lived in m_sel2: everything that reads it is queued on this stream ahead of the memsets below (and a re-allocation frees
    // through hipFree, which waits for the device), so no host synchronisation is needed before the buffer is reused
    // decode buffers: two of F + 1 entries each (see ctx.h)
    const size_t F1 = (size_t)F + 1;
    ctx->m_stride = (uint32_t)F1;
    ctx->m_sel.ensure(2 * F1); ctx->m_lab.ensure(2 * F1); ctx->m_cost.ensure(2 * F1);
    ctx->m_sel2.ensure(F1); ctx->m_cand.ensure(F1); ctx->m_gain.ensure(F1);
    MVS_HIP(hipMemsetAsync(ctx->m_sel.p, 0, 2 * F1 * sizeof(uint32_t), s));
    MVS_HIP(hipMemsetAsync(ctx->m_lab.p, 0, 2 * F1 * sizeof(uint32_t), s));
    MVS_HIP(hipMemsetAsync(ctx->m_cost.p, 0, 2 * F1 * sizeof(float), s));
    // start state = argmin-unary decode everywhere: only needed when no sweep runs (ICM-only); otherwise the first
    // sweep (and, when sharded, the halo exchange that follows it) defines every label that is ever read
    mvs_mrf_progress init; memset(&init, 0, sizeof(init)); init.best = ~0ull; init.energy = ~0ull; init.w = 0u; init.best_w = 1u;
    if (F && params->max_sweeps <= 0) {
        hipLaunchKernelGGL(mrf_argmin_unary_kernel, dim3(nb), dim3(256), 0, s, ctx->r_ptr, ctx->r_view, ctx->r_cost, F, ctx->m_sel.p, ctx->m_lab.p, ctx->m_cost.p);
        MVS_LAUNCH_CHECK();
        init.w = 1u; init.best_w = 0u;
    }
    ctx->b_sel = ctx->m_sel.p + init.best_w * F1; ctx->b_lab = ctx->m_lab.p + init.best_w * F1; ctx->b_cost = ctx->m_cost.p + init.best_w * F1;
    ctx->best_resolved = true;
    MVS_HIP(hipMemsetAsync(ctx->m_gain.p, 0, F1 * sizeof(float), s));
    // energy partials: [0, 4) the reduced pair; then 2 x EPART_BLOCKS per colour phase (sweep kernels) / 2 x 2048 (energy kernel)
    const size_t epart = 4 + 2 * (size_t)EPART_BLOCKS * std::max<size_t>(ctx->m_colours, 1);
    ctx->m_energy.ensure(epart);
    MVS_HIP(hipMemsetAsync(ctx->m_energy.p, 0, epart * sizeof(unsigned long long), s));   // slots a phase never writes stay zero
    // device-side solver state: sweep 0, best = hist[0] = 2^64 - 1
    ctx->m_state.ensure(1); ctx->m_hist.ensure((size_t)std::max(params->max_sweeps, 0) + 2);
    ensure_report_ring(ctx);
    ctx->seq_base += ctx->steps_issued;   // sequence numbers of this solve: seq_base + step (never reused within the context)
    ctx->h_ring[mvs_ctx::RING] = init;   // staging slot for the upload
    MVS_HIP(hipMemcpyAsync(ctx->m_state.p, &ctx->h_ring[mvs_ctx::RING], sizeof(init), hipMemcpyHostToDevice, s));
    // the step kernel's own counter: steps of this solve so far, and the solve's sequence base (staged in the pinned words behind the slots' sequence numbers)
    ctx->m_ctl.ensure(4);
    uint32_t* stage = ctx->h_seq + (mvs_ctx::RING + mvs_ctx::ICM_RING);
    stage[0] = 0u; stage[1] = ctx->seq_base;
    MVS_HIP(hipMemcpyAsync(ctx->m_ctl.p, stage, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    MVS_HIP(hipMemsetAsync(ctx->m_hist.p, 0xFF, sizeof(unsigned long long), s));
    ctx->steps_issued = 0; ctx->icm_dirty_valid = false; ctx->exact_valid = false;   // no synchronisation: the solver's launches queue behind the set-up on the same stream
}

// the best labeling's buffer, once the host needs it (ICM, final energy, labels): one read-back of the solver state
void resolve_best(mvs_ctx* ctx) {
    if (ctx->best_resolved) return;
    if (!ctx->m_state.p) throw StatusError(MVS_ERR_STATE, "mrf setup first");
    mvs_mrf_progress p;
    static_assert(sizeof(p) % 4 == 0 && sizeof(p) <= 64, "mvs_mrf_progress travels through read_words");
    read_words(ctx, ctx->m_state.p, &p, (uint32_t)(sizeof(p) / 4));
    const size_t o = (size_t)(p.best_w & 1u) * ctx->m_stride;
    ctx->b_sel = ctx->m_sel.p + o; ctx->b_lab = ctx->m_lab.p + o; ctx->b_cost = ctx->m_cost.p + o;
    ctx->best_resolved = true;
}

// One bookkeeping step (see mrf_step_kernel); energy = device pointer to the (all-reduced) energy pair, or null: the pair is
// still in per-block partials -- the sweep kernels' (fast path) or mrf_energy(..., reduce = false)'s -- summed by the step kernel.
void mrf_step(mvs_ctx* ctx, const unsigned long long* energy, const unsigned long long* const* peer_tab, uint32_t n_peer, uint32_t peer_off) {
    hipStream_t s = ctx->stream;
    const mvs_mrf_params& P = ctx->m_params;
    if (!ctx->h_ring) throw StatusError(MVS_ERR_STATE, "mrf step before mrf setup");
    const uint32_t n = ++ctx->steps_issued, slot = n % mvs_ctx::RING;
    const unsigned long long* partial = nullptr; uint32_t n_partial = 0;
    if (!energy && !peer_tab) { partial = ctx->m_energy.p + 4; n_partial = ctx->m_energy_from_sweep ? EPART_BLOCKS * std::max<uint32_t>(ctx->m_colours, 1u) : ctx->m_energy_blocks; }
    (void)n; (void)slot;   // the kernel derives both from its own step counter (ctl), which mirrors ctx->steps_issued
    hipLaunchKernelGGL(mrf_step_kernel, dim3(1), dim3(1024), 0, s, ctx->m_state.p, ctx->m_hist.p, energy, partial, n_partial, ctx->m_energy.p,
                       ctx->d_ring, ctx->d_seq, (uint32_t)mvs_ctx::RING, ctx->m_ctl.p, P.max_sweeps, P.min_sweeps, P.window, P.min_improvement, peer_tab, n_peer, peer_off);
    MVS_LAUNCH_CHECK();
    ctx->icm_dirty_valid = false; ctx->best_resolved = false; ctx->exact_valid = false;   // the best labeling may change
}
void mrf_poll(mvs_ctx* ctx, uint32_t step, mvs_mrf_progress* out) {
    if (step == 0 || step > ctx->steps_issued || step + mvs_ctx::RING <= ctx->steps_issued)
        throw StatusError(MVS_ERR_INVALID, "mrf poll: step not among the last 16 issued");
    const uint32_t slot = step % mvs_ctx::RING;
    wait_report(ctx, slot, ctx->seq_base + step);
    *out = ctx->h_ring[slot];
}

// ---- reports through pinned host memory ----
// Coherent (fine-grained) host memory: a value, a system-scope fence, then the slot's sequence number.  The host spins on the
// number -- no event in the stream (an event record costs the stream ~5 us, and the solver would record one per sweep).
void ensure_report_ring(mvs_ctx* ctx) {
    if (ctx->h_ring) return;
    constexpr uint32_t NS = mvs_ctx::RING + mvs_ctx::ICM_RING;
    MVS_HIP(hipHostMalloc((void**)&ctx->h_ring, (mvs_ctx::RING + 1) * sizeof(mvs_mrf_progress), hipHostMallocCoherent));
    MVS_HIP(hipHostGetDevicePointer((void**)&ctx->d_ring, ctx->h_ring, 0));
    MVS_HIP(hipHostMalloc((void**)&ctx->h_seq, (NS + 2) * sizeof(uint32_t), hipHostMallocCoherent));   // + 2 staging words (mrf_setup)
    MVS_HIP(hipHostGetDevicePointer((void**)&ctx->d_seq, ctx->h_seq, 0));
    MVS_HIP(hipHostMalloc((void**)&ctx->h_icm, mvs_ctx::ICM_RING * sizeof(uint32_t), hipHostMallocCoherent));
    MVS_HIP(hipHostGetDevicePointer((void**)&ctx->d_icm, ctx->h_icm, 0));
    for (uint32_t k = 0; k < NS; ++k) ctx->h_seq[k] = 0u;
    ctx->seq_base = 1u; ctx->icm_seq = 0x40000000u;   // no report carries the initial 0
}
void report_u32(mvs_ctx* ctx, const uint32_t* d_src, uint32_t* d_dst, uint32_t seq_slot, uint32_t seq) {
    hipLaunchKernelGGL(report_u32_kernel, dim3(1), dim3(1), 0, ctx->stream, d_src, d_dst, ctx->d_seq + seq_slot, seq);
    MVS_LAUNCH_CHECK();
}
void read_words(mvs_ctx* ctx, const void* d_src, void* out, uint32_t n_words) {
    if (n_words == 0 || n_words > 64) throw StatusError(MVS_ERR_INVALID, "read_words: 1 .. 64 words");
    if (!ctx->h_rb) {   // 64 data words, the sequence number, and behind it a 4 KB staging area for larger read-backs (read_block)
        MVS_HIP(hipHostMalloc((void**)&ctx->h_rb, (128 + 1024) * sizeof(uint32_t), hipHostMallocCoherent));
        MVS_HIP(hipHostGetDevicePointer((void**)&ctx->d_rb, ctx->h_rb, 0));
        ctx->h_rb[64] = 0u; ctx->rb_seq = 0u;
    }
    const uint32_t seq = ++ctx->rb_seq ? ctx->rb_seq : ++ctx->rb_seq;   // never the initial 0
    hipLaunchKernelGGL(report_words_kernel, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)d_src, n_words, ctx->d_rb, ctx->d_rb + 64, seq);
    MVS_LAUNCH_CHECK();
    const volatile uint32_t* p = ctx->h_rb + 64;
    for (uint64_t spins = 1; *p != seq; ++spins) {
        if ((spins & 0x3FFFu) == 0u) {   // now and then: is the stream still working on it?
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) { if (*p == seq) break; throw HipError("a device read-back did not arrive although the stream is idle"); }
            if (q != hipErrorNotReady) MVS_HIP(q);
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    memcpy(out, ctx->h_rb, n_words * sizeof(uint32_t));
}
// up to 4 KB: through the pinned staging area (a copy into PAGEABLE memory is staged by the driver and costs 10 us more)
void read_block(mvs_ctx* ctx, const void* d_src, void* out, size_t bytes) {
    if (bytes <= 64 * sizeof(uint32_t) && bytes % 4 == 0) { read_words(ctx, d_src, out, (uint32_t)(bytes / 4)); return; }
    if (bytes > 1024 * sizeof(uint32_t)) throw StatusError(MVS_ERR_INVALID, "read_block: at most 4 KB");
    if (!ctx->h_rb) { uint32_t dummy; read_words(ctx, d_src, &dummy, 1); }   // (allocates the pinned area)
    MVS_HIP(hipMemcpyAsync(ctx->h_rb + 128, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->h_rb + 128, bytes);
}
void wait_report(mvs_ctx* ctx, uint32_t seq_slot, uint32_t seq) {
    const volatile uint32_t* p = ctx->h_seq + seq_slot;
    for (uint64_t spins = 1; *p != seq; ++spins) {
        if ((spins & 0x3FFFu) == 0u) {   // now and then: is the stream still working on it?
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) { if (*p == seq) break; throw HipError("a device report did not arrive although the stream is idle"); }
            if (q != hipErrorNotReady) MVS_HIP(q);
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
}

// Damping schedule (part of the solver's definition, restated in oracle/oracle.cpp): messages are damped with
// alpha = params.damping on ODD sweeps (1st, 3rd, ...) and written undamped on even sweeps.  Undamped sweeps oscillate
// (C3: 0.9 % higher final energy), but damping every second sweep suppresses that just as well as damping every sweep
// (C3: 44 sweeps to E = 1 111 890 with 0.2 on odd sweeps vs 47 to 1 110 970 with 0.1 on all) -- and an undamped sweep
// does not re-read its previous outgoing messages.
// Round 6 re-scored the schedule in milliseconds (scripts/schedule_score.py, profiles/r06_schedule_score_*.json): an undamped launch is 14 %
// cheaper than a damped one, and damping every FOURTH sweep (1st, 5th, ...) holds the oscillation as well: the definition is period 4.
// (option "mrf_damp_period", an experiment knob: 4 = the definition; p damps sweeps 1, p + 1, 2p + 1, ...; 1 every sweep; 0 none)
static float sweep_alpha(const mvs_ctx* ctx) {
    const uint32_t p = (uint32_t)std::max(ctx->mrf_damp_period, 0);
    if (p == 0) return 0.0f;
    return (p == 1 || ctx->m_sweep_no % p == 1u) ? ctx->m_params.damping : 0.0f;
}

// one launch of the fast kernel over positions [qb, qe) (one (colour, class) range of the schedule order); its per-block energy
// partials go to slots [slot, slot + blocks) of the phase's region.  Returns the number of blocks (= slots) used.
template <int G>
static unsigned launch_sweep4_g(mvs_ctx* ctx, uint32_t phase, uint32_t qb, uint32_t qe, unsigned slot, unsigned slot_cap) {
    constexpr int NPB = 256 / G;
    const unsigned need = (qe - qb + NPB - 1) / NPB;
    const float rho = ctx->m_params.rho, alpha = sweep_alpha(ctx);
    // persistent lane groups: at most as many blocks as are resident at once (a partial second wave of
    // blocks would double the tail); mrf_blocks_per_cu > 0 overrides
    static std::atomic<int> resident_once{0};   // (the same value on every device of a node; in-process ranks race for it: atomic, idempotent)
    int resident = resident_once.load(std::memory_order_relaxed);
    if (resident == 0) {
        int per_cu = 0; hipDeviceProp_t prop;
        MVS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mrf_sweep4_kernel<G, true, true, true>, 256, 0));
        MVS_HIP(hipGetDeviceProperties(&prop, ctx->device));
        resident = std::max(1, per_cu) * prop.multiProcessorCount;
        resident_once.store(resident, std::memory_order_relaxed);
    }
    unsigned blocks = ctx->mrf_blocks_per_cu > 0 ? 256u * (unsigned)ctx->mrf_blocks_per_cu : (unsigned)resident;
    blocks = std::max(1u, std::min(std::min(need, blocks), slot_cap));
    if (blocks > 8) blocks &= ~7u;   // multiple of the 8 XCDs
    msg_t* msg = reinterpret_cast<msg_t*>(ctx->m_msg_a.p);
    unsigned long long* partial = ctx->m_energy.p + 4 + 2 * ((size_t)EPART_BLOCKS * phase + slot);
#define SWEEP4_ARGS dim3(blocks), dim3(256), 0, ctx->stream, ctx->m_desc.p, ctx->m_rec.p, msg, ctx->m_state.p, ctx->m_lab.p, ctx->m_stride, qb, qe, rho, alpha, partial
    if (alpha != 0.0f) {
        if (ctx->mrf_late_old) { if (ctx->mrf_xcd) hipLaunchKernelGGL((mrf_sweep4_kernel<G, true, true, true>), SWEEP4_ARGS); else hipLaunchKernelGGL((mrf_sweep4_kernel<G, true, false, true>), SWEEP4_ARGS); }
        else { if (ctx->mrf_xcd) hipLaunchKernelGGL((mrf_sweep4_kernel<G, true, true, false>), SWEEP4_ARGS); else hipLaunchKernelGGL((mrf_sweep4_kernel<G, true, false, false>), SWEEP4_ARGS); }
    } else { if (ctx->mrf_xcd) hipLaunchKernelGGL((mrf_sweep4_kernel<G, false, true, false>), SWEEP4_ARGS); else hipLaunchKernelGGL((mrf_sweep4_kernel<G, false, false, false>), SWEEP4_ARGS); }
#undef SWEEP4_ARGS
    MVS_LAUNCH_CHECK();
    return blocks;
}
// class 1 through mrf_sweep8_kernel (option "mrf_wide"): 8 lanes per node, 32 nodes per block
static unsigned launch_sweep8(mvs_ctx* ctx, uint32_t phase, uint32_t qb, uint32_t qe, unsigned slot, unsigned slot_cap) {
    constexpr int NPB = 256 / 8;
    const unsigned need = (qe - qb + NPB - 1) / NPB;
    const float rho = ctx->m_params.rho, alpha = sweep_alpha(ctx);
    static std::atomic<int> resident_once{0};   // (the same value on every device of a node; in-process ranks race for it: atomic, idempotent)
    int resident = resident_once.load(std::memory_order_relaxed);
    if (resident == 0) {
        int per_cu = 0; hipDeviceProp_t prop;
        MVS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mrf_sweep8_kernel<true, true>, 256, 0));
        MVS_HIP(hipGetDeviceProperties(&prop, ctx->device));
        resident = std::max(1, per_cu) * prop.multiProcessorCount;
        resident_once.store(resident, std::memory_order_relaxed);
    }
    unsigned blocks = ctx->mrf_blocks_per_cu > 0 ? 256u * (unsigned)ctx->mrf_blocks_per_cu : (unsigned)resident;
    blocks = std::max(1u, std::min(std::min(need, blocks), slot_cap));
    if (blocks > 8) blocks &= ~7u;
    msg_t* msg = reinterpret_cast<msg_t*>(ctx->m_msg_a.p);
    unsigned long long* partial = ctx->m_energy.p + 4 + 2 * ((size_t)EPART_BLOCKS * phase + slot);
#define SWEEP8_ARGS dim3(blocks), dim3(256), 0, ctx->stream, ctx->m_desc.p, ctx->m_rec.p, msg, ctx->m_state.p, ctx->m_lab.p, ctx->m_stride, qb, qe, rho, alpha, partial
    if (alpha != 0.0f) { if (ctx->mrf_xcd) hipLaunchKernelGGL((mrf_sweep8_kernel<true, true>), SWEEP8_ARGS); else hipLaunchKernelGGL((mrf_sweep8_kernel<true, false>), SWEEP8_ARGS); }
    else { if (ctx->mrf_xcd) hipLaunchKernelGGL((mrf_sweep8_kernel<false, true>), SWEEP8_ARGS); else hipLaunchKernelGGL((mrf_sweep8_kernel<false, false>), SWEEP8_ARGS); }
#undef SWEEP8_ARGS
    MVS_LAUNCH_CHECK();
    return blocks;
}
static unsigned launch_sweep_generic(mvs_ctx* ctx, uint32_t phase, uint32_t qb, uint32_t qe, unsigned slot, unsigned slot_cap) {
    const unsigned need = qe - qb;   // one block per node
    const unsigned blocks = std::max(1u, std::min(std::min(need, 256u * 4u), slot_cap));
    const float rho = ctx->m_params.rho, alpha = sweep_alpha(ctx);
    msg_t* msg = reinterpret_cast<msg_t*>(ctx->m_msg_a.p);
    unsigned long long* partial = ctx->m_energy.p + 4 + 2 * ((size_t)EPART_BLOCKS * phase + slot);
#define GENERIC_ARGS dim3(blocks), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->r_adj_ptr, ctx->r_adj, ctx->m_edge.p, ctx->m_map.p, ctx->m_colour.p, msg, ctx->m_perm.p, \
                     ctx->m_state.p, ctx->m_lab.p, ctx->m_stride, ctx->pq.p, qb, qe, rho, alpha, partial
    if (alpha != 0.0f) hipLaunchKernelGGL(mrf_sweep_generic_kernel<true>, GENERIC_ARGS);
    else hipLaunchKernelGGL(mrf_sweep_generic_kernel<false>, GENERIC_ARGS);
#undef GENERIC_ARGS
    MVS_LAUNCH_CHECK();
    return blocks;
}

// positions [qb, qe) in the schedule order of the nodes of (sub-class `sub`, zone) whose id lies in [nb0, ne0)
static void sub_range(mvs_ctx* ctx, uint32_t sub, uint32_t zone, uint32_t nb0, uint32_t ne0, uint32_t* qb, uint32_t* qe) {
    const uint32_t key = 2u * sub + zone;
    const uint32_t cb = ctx->m_sub_begin[key], ce = ctx->m_sub_begin[key + 1];
    if (ce <= cb || (nb0 == 0 && ne0 >= ctx->csr_faces)) { *qb = cb; *qe = ce; return; }
    if (ctx->m_range_nb != nb0 || ctx->m_range_ne != ne0 || ctx->m_range_q.size() != 2 * (size_t)N_KEY) {
        // a rank's own share of every (sub-class, zone): one small kernel + read-back per (range, setup), then cached
        uint32_t* d = ctx->m_sub.p + N_KEY + 2;   // (allocated by mrf_setup: ensure() here would drop sub_begin)
        hipLaunchKernelGGL(mrf_sub_range_kernel, dim3((N_KEY + 255) / 256), dim3(256), 0, ctx->stream, ctx->m_perm.p, ctx->m_sub.p, nb0, ne0, d);
        MVS_LAUNCH_CHECK();
        ctx->m_range_q.assign(2 * (size_t)N_KEY, 0);
        MVS_HIP(hipMemcpyAsync(ctx->m_range_q.data(), d, 2 * N_KEY * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        ctx->m_range_nb = nb0; ctx->m_range_ne = ne0;
    }
    *qb = ctx->m_range_q[2 * key]; *qe = ctx->m_range_q[2 * key + 1];
}

// one colour phase of a sweep over the nodes of that colour with id in [nb0, ne0): in place.  Up to five launches per zone, one per node
// class present in the colour (a uniform mesh has one); their energy partials share the phase's EPART_BLOCKS slots: the boundary zone's
// launches take slots [0, EPART_BOUNDARY), the interior's the rest (a slot nobody writes stays zero; the geometry of a phase's launches
// is the same in every sweep, so every slot that is ever written is rewritten by every sweep).
// part: MRF_PART_ALL = both zones, slots from 0 (one launch where the zones are adjacent in the schedule),
//       MRF_PART_BOUNDARY / MRF_PART_INTERIOR = one zone -- a sharded caller runs BOUNDARY, hands over, then INTERIOR.
// A solve keeps to one of the two modes (the slots of a launch differ between them).
void mrf_sweep_phase(mvs_ctx* ctx, uint32_t phase, uint32_t nb0, uint32_t ne0, int part) {
    if (phase == 0 && part != MRF_PART_INTERIOR) ++ctx->m_sweep_no;      // sweeps are counted by their first phase (every caller runs the phases in order, a split phase boundary first)
    if (phase >= ctx->m_colours || ne0 <= nb0) return;
    constexpr unsigned EPART_BOUNDARY = 256;
    struct Launch { uint32_t g, qb, qe; } L[10]; int n = 0;
    for (uint32_t g = 0; g < 5; ++g) {
        const uint32_t sub = g < 4 ? 4 * phase + g : SUB_GENERIC + phase;
        uint32_t b0 = 0, b1 = 0, i0 = 0, i1 = 0;
        if (part != MRF_PART_INTERIOR) sub_range(ctx, sub, 0, nb0, ne0, &b0, &b1);
        if (part != MRF_PART_BOUNDARY) sub_range(ctx, sub, 1, nb0, ne0, &i0, &i1);
        if (b1 > b0 && i1 > i0 && b1 == i0) { L[n++] = {g, b0, i1}; continue; }   // both zones, adjacent in the schedule (a single context's marks): one launch
        if (b1 > b0) L[n++] = {g, b0, b1};
        if (i1 > i0) L[n++] = {g, i0, i1};
    }
    unsigned slot = part == MRF_PART_INTERIOR ? EPART_BOUNDARY : 0u;
    const unsigned slot_end = part == MRF_PART_BOUNDARY ? EPART_BOUNDARY : EPART_BLOCKS;
    for (int k = 0; k < n; ++k) {
        const unsigned cap = slot_end - slot - (unsigned)(n - 1 - k);   // leaves at least one slot for every launch still to come
        const uint32_t qb = L[k].qb, qe = L[k].qe;
        switch (L[k].g) {
            case 0: slot += launch_sweep4_g<8>(ctx, phase, qb, qe, slot, cap); break;
            case 1: slot += ctx->m_wide_layout ? launch_sweep8(ctx, phase, qb, qe, slot, cap) : launch_sweep4_g<16>(ctx, phase, qb, qe, slot, cap); break;
            case 2: slot += launch_sweep4_g<32>(ctx, phase, qb, qe, slot, cap); break;
            case 3: slot += launch_sweep4_g<64>(ctx, phase, qb, qe, slot, cap); break;      // one node per wave: several hundred views per face
            default: slot += launch_sweep_generic(ctx, phase, qb, qe, slot, cap);
        }
    }
}
// one sweep = every colour phase in turn (callers that shard the nodes exchange halos between the phases themselves)
void mrf_sweep(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    for (uint32_t ph = 0; ph < ctx->m_colours; ++ph) mrf_sweep_phase(ctx, ph, nb0, ne0, MRF_PART_ALL);
    // the sweep kernels leave the sweep's energy behind as per-block partials (valid when the range is the whole graph)
    ctx->m_energy_from_sweep = nb0 == 0 && ne0 >= ctx->csr_faces && ctx->m_colours > 0;
}

// The solver tracks energies of the 16-bit unaries the sweeps see; the polish and everything reported use the exact costs:
// best_cost[i] := cost[col_ptr[i] + best_sel[i]] (1.0 for an empty column)
// The sweeps keep ONE word per node, its label.  Whatever works on the best labeling afterwards (polish, region moves, reported energy)
// needs the label's position in the column and its EXACT unary as well: derived here, once, from the label (idempotent).
__global__ void mrf_exact_cost_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                      const uint32_t* __restrict__ lab, uint32_t node_begin, uint32_t node_end, uint32_t* __restrict__ sel, float* __restrict__ selcost,
                                      uint32_t* __restrict__ foreign /* labels that are NOT entries of their node's column (counted into "Incorrect labeling") */) {
    const uint32_t i = node_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= node_end) return;
    const uint32_t p0 = col_ptr[i], K = col_ptr[i + 1] - p0, l = lab[i];
    const uint32_t t = (K && l) ? label_position(view_id, p0, K, l) : 0u;
    // label_position clamps: a stale or foreign label (a scatter through the building blocks, a table pruned behind the labeling) would
    // silently take a neighbour's position and unary -- it is counted instead
    if ((K != 0u) != (l != 0u) || (K && l && (uint32_t)view_id[p0 + t] + 1u != l)) atomicAdd(foreign, 1u);
    sel[i] = t;
    selcost[i] = K ? cost[p0 + t] : 1.0f;
}
void mrf_exact_costs(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    resolve_best(ctx);
    if (ne0 <= nb0) return;
    // the polish and the region moves keep position / label / unary of the best labeling consistent: derived once per labeling and range
    if (ctx->exact_valid && ctx->exact_nb == nb0 && ctx->exact_ne == ne0) return;
    ctx->exact_valid = true; ctx->exact_nb = nb0; ctx->exact_ne = ne0;
    MVS_HIP(hipMemsetAsync(ctx->m_moved.p + 6, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(mrf_exact_cost_kernel, dim3((ne0 - nb0 + 255) / 256), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, (const uint32_t*)ctx->b_lab, nb0, ne0, ctx->b_sel, ctx->b_cost, ctx->m_moved.p + 6);
    MVS_LAUNCH_CHECK();
}

// energy of the current decode (best == false) or of the best labeling over nodes [nb0, ne0)
// -> ctx->m_energy (device, 2 x u64), asynchronous
void mrf_energy(mvs_ctx* ctx, bool best, uint32_t nb0, uint32_t ne0, bool reduce) {
    const unsigned blocks = ne0 > nb0 ? std::min<unsigned>((ne0 - nb0 + 255) / 256, 2048u) : 0u;
    ctx->m_energy_blocks = blocks; ctx->m_energy_from_sweep = false;
    unsigned long long* partial = ctx->m_energy.p + 4;
    if (best) mrf_exact_costs(ctx, nb0, ne0);   // idempotent: whatever is reported about the best labeling uses the exact costs
    if (blocks) {
        if (best) hipLaunchKernelGGL(mrf_energy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->r_adj_ptr, ctx->r_adj, ctx->b_lab, ctx->b_cost, (const mvs_mrf_progress*)nullptr, 0u, nb0, ne0, partial);
        else hipLaunchKernelGGL(mrf_energy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->r_adj_ptr, ctx->r_adj, ctx->m_lab.p, ctx->m_cost.p, ctx->m_state.p, ctx->m_stride, nb0, ne0, partial);
        MVS_LAUNCH_CHECK();
    }
    if (!reduce) return;   // the caller's mrf_step sums the partials
    hipLaunchKernelGGL(mrf_energy_reduce_kernel, dim3(1), dim3(1024), 0, ctx->stream, partial, blocks, ctx->m_energy.p, (unsigned long long*)nullptr);
    MVS_LAUNCH_CHECK();
}
// the energy pair the fast-path sweep kernels of the last sweep left behind as per-block partials (own node range of a
// sharded caller, or the whole graph) -> ctx->m_energy (device), asynchronous
void mrf_sweep_energy_reduce(mvs_ctx* ctx, unsigned long long* out2) {
    const uint32_t n = EPART_BLOCKS * std::max<uint32_t>(ctx->m_colours, 1u);
    hipLaunchKernelGGL(mrf_energy_reduce_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->m_energy.p + 4, n, ctx->m_energy.p, out2);
    MVS_LAUNCH_CHECK();
}

// best labeling := the current decode (an index flip on the device)
void mrf_keep_best(mvs_ctx* ctx) {
    ctx->icm_dirty_valid = false; ctx->best_resolved = false; ctx->exact_valid = false;
    if (!ctx->m_state.p) throw StatusError(MVS_ERR_STATE, "mrf setup first");
    hipLaunchKernelGGL(mrf_flip_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->m_state.p);
    MVS_LAUNCH_CHECK();
}

// ICM on the best labeling (in place): gains of nodes [nb0, ne0)
void mrf_icm_gain(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    if (ne0 <= nb0) return;
    resolve_best(ctx);
    const uint32_t K = ctx->m_kmax, n = ne0 - nb0;
    const bool whole = nb0 == 0 && ne0 == ctx->csr_faces;
#define ICM_G(GG, B, E, LIST) hipLaunchKernelGGL(mrf_icm_gain_kernel<GG>, dim3(((E) - (B) + (256 / GG) - 1) / (256 / GG)), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, \
                                     ctx->r_adj_ptr, ctx->r_adj, ctx->b_sel, ctx->b_lab, (B), (E), ctx->m_gain.p, ctx->m_cand.p, (LIST), (const uint32_t*)nullptr)
    // active set (unsharded calls only): after one full evaluation, only the nodes the last apply listed -- the nodes that
    // moved and their neighbours -- are re-evaluated; everybody else's stored gain / candidate are still the values a
    // full pass would compute.  Sharded callers exchange labels behind the library's back, so they evaluate all.
    // The list's length is read on the device (fixed grid, grid-stride loop): no host round trip between the rounds.
    if (whole && ctx->icm_dirty_valid) {
        const uint32_t* list = ctx->m_alist.p; const uint32_t* cnt = ctx->m_moved.p + 1;
#define ICM_L(GG) hipLaunchKernelGGL(mrf_icm_gain_kernel<GG>, dim3(std::max(1u, std::min<unsigned>((n + (256 / GG) - 1) / (256 / GG), 1024u))), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, \
                                     ctx->r_adj_ptr, ctx->r_adj, ctx->b_sel, ctx->b_lab, 0u, 0u, ctx->m_gain.p, ctx->m_cand.p, list, cnt)
        if (K <= 8) ICM_L(8); else if (K <= 16) ICM_L(16); else if (K <= 32) ICM_L(32); else ICM_L(64);
#undef ICM_L
        MVS_LAUNCH_CHECK();
        return;
    }
    if (whole) {   // whole-graph calls: the fast nodes through their descriptors (schedule positions [0, n_fast)), the generic nodes as a list
        const uint32_t nf = ctx->m_n_fast;
        if (nf) {
#define ICM_D(GG) hipLaunchKernelGGL(mrf_icm_gain_desc_kernel<GG>, dim3(std::max(1u, std::min<unsigned>((nf + (256 / GG) - 1) / (256 / GG), 256u * 8u))), dim3(256), 0, ctx->stream, \
                                     ctx->m_desc.p, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->b_sel, ctx->b_lab, 0u, nf, ctx->m_gain.p, ctx->m_cand.p)
            if (K <= 8) ICM_D(8); else if (K <= 64) ICM_D(16); else ICM_D(32);   // 16 lanes up to 64 labels: 0.74 -> 0.62 ms of polish at C3 (K = 50)
#undef ICM_D
            MVS_LAUNCH_CHECK();
        }
        if (nf < n) ICM_G(64, nf, n, (const uint32_t*)ctx->m_perm.p);
    } else {
        const uint32_t* none = nullptr;
        // lanes per node: the label loop strides by the group size, so any size is exact; 16 lanes keep four nodes per wave busy up to
        // 64 labels (a 64-lane group would idle most of its lanes on the usual 40 - 50 candidates)
        if (K <= 8) ICM_G(8, nb0, ne0, none); else if (K <= 64) ICM_G(16, nb0, ne0, none); else if (K <= 128) ICM_G(32, nb0, ne0, none); else ICM_G(64, nb0, ne0, none);
    }
#undef ICM_G
    MVS_LAUNCH_CHECK();
    ctx->icm_dirty_valid = whole;   // every gain is current: the next apply starts the list
}
void mrf_icm_apply(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0) {
    MVS_HIP(hipMemsetAsync(ctx->m_moved.p, 0, 2 * sizeof(uint32_t), ctx->stream));
    if (ne0 <= nb0) return;
    resolve_best(ctx);
    const bool whole = nb0 == 0 && ne0 == ctx->csr_faces;
    if (!whole) ctx->icm_dirty_valid = false;
    uint32_t* alist = nullptr;
    if (ctx->icm_dirty_valid) {   // winners form an independent set: at most (nodes + directed edges) entries
        ctx->m_alist.ensure((size_t)ctx->csr_faces + (size_t)ctx->m_n_adj + 16);
        alist = ctx->m_alist.p;
    }
    hipLaunchKernelGGL(mrf_icm_apply_kernel, dim3((ne0 - nb0 + 255) / 256), dim3(256), 0, ctx->stream, ctx->r_ptr, ctx->r_view, ctx->r_cost, ctx->r_adj_ptr, ctx->r_adj,
                       ctx->m_gain.p, ctx->m_cand.p, ctx->b_sel, ctx->b_lab, ctx->b_cost, nb0, ne0, ctx->m_moved.p, alist, ctx->t_perm);
    MVS_LAUNCH_CHECK();
}
// labels of nodes [nb0, ne0) of the best labeling into d_labels[0 .. ne0 - nb0); out = {bad, unseen}
// (caller_order: the whole graph's labels at the caller's face ids -- ctx->t_perm; otherwise positions nb0 .. of the table's order)
void mrf_labels(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0, uint32_t* d_labels, uint32_t out[2], bool caller_order) {
    uint32_t* bu = ctx->m_moved.p + 2;
    resolve_best(ctx);
    MVS_HIP(hipMemsetAsync(bu, 0, 2 * sizeof(uint32_t), ctx->stream));
    if (ne0 > nb0) {
        const uint32_t* foreign = (ctx->exact_valid && ctx->exact_nb == nb0 && ctx->exact_ne == ne0) ? ctx->m_moved.p + 6 : (const uint32_t*)nullptr;
        hipLaunchKernelGGL(mrf_labels_kernel, dim3((ne0 - nb0 + 255) / 256), dim3(256), 0, ctx->stream, ctx->b_lab, nb0, ne0, ctx->csr_views, d_labels, bu, caller_order ? ctx->t_perm : (const uint32_t*)nullptr, foreign);
        MVS_LAUNCH_CHECK();
    }
    read_words(ctx, bu, out, 2);
}

}  // namespace mvs
