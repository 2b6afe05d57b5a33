// k_bvh.hip -- occlusion rays of calculate_face_projection_infos
// (libs/tex/calculate_data_costs.cpp:144 BVH build, :194-215 the three
// vertex->camera any-hit rays per (face, view)).
//
// rayint's acc::BVHTree is replaced by an implicit 4-ary BVH built ON THE GPU
// over Hilbert-sorted triangles (no pointers: node i of level L has children
// 4i..4i+3 of level L-1; a level-0 node's children are leaves of LEAF_T consecutive
// triangles).  One node = one 128-byte line holding the four child boxes.
// Traversal is stackless: one 4-bit pending-children mask per level packed in a
// 64-bit register.  The hit predicate (dmath.h ray_tri) is evaluated on the same
// triangles a brute-force loop would accept, boxes are padded and the slab test
// widened, so the boolean does not depend on the tree.
//
// The reference casts 3 rays per (face, view); a ray depends only on
// (vertex, view), so each distinct ray is traced ONCE (vertices are shared by
// ~6 faces): need bits -> ray kernel -> occluded bits, view-major bit matrices.
#include "ctx.h"
#include <rocprim/rocprim.hpp>
#ifndef MVS_LEAF_T
#define MVS_LEAF_T 16
#endif

namespace mvs {

namespace {

// triangles per leaf (consecutive in Hilbert order).  With the packet traversal a leaf round tests (64 / LEAF_T) candidate
// rays against LEAF_T triangles: bigger leaves mean fewer node visits and fuller rounds, more triangle tests per ray.
// Measured at C3 (A/B on one box): on the plain Hilbert order 4 / 8 / 16 triangles = 16.7 / 13.6 / 13.2 ms, with the
// median refinement below (tight leaves) 8 / 16 / 32 = 10.8 / 9.6 / 10.3 ms.
constexpr uint32_t LEAF_T = MVS_LEAF_T;
constexpr int LEAF_SLOTS = 64 / (int)LEAF_T;

__device__ __forceinline__ uint32_t f2ord(float f) {  // order-preserving float -> uint
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// scene_box: 6 ordered-uint words (min xyz, max xyz).  Two stages: every block leaves ONE box (butterflies + LDS), a second launch folds
// the <= 512 block boxes.  (Rounds 1 - 5 folded with atomics straight into the six words: same-address traffic from every wave serialises
// in one L2 channel -- 0.10 ms for a million vertices, rocprofv3 round 6; now ~0.01.)
__global__ void __launch_bounds__(256) bbox_kernel(const float* __restrict__ verts, uint32_t n_verts, uint32_t* __restrict__ part /* [gridDim.x][6] */) {
    __shared__ uint32_t s_red[4][6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n_verts; v += gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) { const float x = verts[3 * (size_t)v + a]; lo[a] = fminf(lo[a], x); hi[a] = fmaxf(hi[a], x); }
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
        if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][a] = f2ord(lo[a]); s_red[threadIdx.x >> 6][3 + a] = f2ord(hi[a]); }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        uint32_t v = s_red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? min(v, s_red[w][threadIdx.x]) : max(v, s_red[w][threadIdx.x]);
        part[6 * blockIdx.x + threadIdx.x] = v;
    }
}
__global__ void __launch_bounds__(64) bbox_fold_kernel(const uint32_t* __restrict__ part, uint32_t n_parts, uint32_t* __restrict__ box) {
    for (int a = 0; a < 6; ++a) {
        uint32_t v = a < 3 ? 0xFFFFFFFFu : 0u;
        for (uint32_t k = threadIdx.x; k < n_parts; k += 64) v = a < 3 ? min(v, part[6 * k + a]) : max(v, part[6 * k + a]);
        for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor(v, o, 64); v = a < 3 ? min(v, w) : max(v, w); }
        if (threadIdx.x == 0) box[a] = v;
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// 30-bit Hilbert index of a 10-bit lattice point (Skilling's transpose construction: undo the excess work of the
// Gray code level by level, then Gray-encode across the axes; the three 10-bit words interleave to the index).
// Consecutive indices are always lattice neighbours -- unlike the Z-order, whose jumps at octant boundaries put far
// apart triangles into one run of LEAF_T consecutive triangles (= one leaf of the implicit tree) and inflate its box:
// measured on the C3 mesh, 35 % fewer node visits and 45 % fewer (ray, leaf) pairs per packet than Morton order.
__device__ __forceinline__ uint32_t hilbert30(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t X[3] = {x, y, z};
#pragma unroll
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1) {
        const uint32_t P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    uint32_t t = 0;
#pragma unroll
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1) if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return (expand10(X[0]) << 2) | (expand10(X[1]) << 1) | expand10(X[2]);
}

__global__ void curve_key_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, uint32_t n_faces,
                              const uint32_t* __restrict__ box, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    float lo[3], ext[3];
    for (int a = 0; a < 3; ++a) { lo[a] = ord2f(box[a]); ext[a] = ord2f(box[3 + a]) - lo[a]; }
    const uint32_t* fv = faces + 3 * (size_t)f;
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float c = (verts[3 * (size_t)fv[0] + a] + verts[3 * (size_t)fv[1] + a] + verts[3 * (size_t)fv[2] + a]) * (1.0f / 3.0f);
        float t = ext[a] > 0.0f ? (c - lo[a]) / ext[a] : 0.0f;
        t = fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
        q[a] = (uint32_t)t;
    }
    keys[f] = hilbert30(q[0], q[1], q[2]);
    vals[f] = f;
}

// Hilbert index of a vertex (rays are launched in Hilbert order of their origin: 64 neighbouring
// vertices per wave = a compact patch of nearly parallel rays)
__global__ void vertex_key_kernel(const float* __restrict__ verts, uint32_t n_verts, const uint32_t* __restrict__ box,
                               uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_verts) return;
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(box[a]), ext = ord2f(box[3 + a]) - lo;
        float t = ext > 0.0f ? (verts[3 * (size_t)v + a] - lo) / ext : 0.0f;
        t = fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
        q[a] = (uint32_t)t;
    }
    keys[v] = hilbert30(q[0], q[1], q[2]);
    vals[v] = v;
}
__global__ void invert_perm_kernel(const uint32_t* __restrict__ perm, uint32_t n, uint32_t* __restrict__ inv) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) inv[perm[s]] = s;
}

// The LOWER levels of the order, in LDS.  A window of RW = 2048 consecutive triangles is a whole number of subtrees of the implicit tree;
// the levels above it are cut by k_kdorder.hip (round 6: the exact top-down cuts of the whole tree are worth 25 % of the ray stage; rounds
// 1 - 5 cut only inside windows of 512 of the Hilbert order).  Inside each window the triangles are re-partitioned top down: a segment is sorted
// along the longest axis of its centroids and cut in the middle (spatial median), recursively down to the leaves, so two
// binary levels = one 4-ary level.  The curve decides WHICH 512 triangles share a subtree, the median splits decide how
// they are grouped inside it: on the C3 mesh 26 % fewer leaf rounds and 34 % fewer (ray, leaf) pairs per packet
// (simulated; windows of 2048 would give 29 % / 38 %).  Deterministic: ties are broken by the triangle id.
constexpr int RW = 2048, RW_T = RW / 2;   // window and threads (one compare-exchange pair per thread); the levels above: k_kdorder.hip
// (Round 6 also tried the window in REGISTERS -- two elements per thread, distances 2 .. 64 as wave shuffles, only >= 128 through LDS: 0.54 ms
//  against 0.37 ms for this kernel at C3: a shuffle is an LDS-crossbar operation too, and every element fetched its partner instead of every
//  pair being visited once.)  Key and slot of a position are ONE LDS word: half the LDS instructions of separate arrays.
__global__ void __launch_bounds__(RW_T) refine_order_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, uint32_t* __restrict__ order,
                                                           uint32_t n_faces) {
    __shared__ float s_c[3][RW];          // centroid of the triangle in slot k (slots never move)
    __shared__ uint32_t s_id[RW];         // triangle id of slot k (0xFFFFFFFF: padding behind the last triangle)
    __shared__ uint32_t s_ks[RW];         // position i: key << 11 | slot -- key = the centroid's coordinate along the segment's longest axis, in 2^21 steps
                                          // of the segment's extent (ties -- centroids closer than 5e-7 of the extent -- are ranked by the triangle id):
                                          // ONE 4-byte LDS word per position; the kernel is
                                          // bound by the LDS traffic of its 266 compare-exchange passes (key + slot as 8 bytes: 0.32 ms at config 3)
    __shared__ uint32_t s_box[RW / (2 * LEAF_T)][6];   // centroid box per segment (ordered uints)
    const uint32_t w0 = blockIdx.x * RW;
    const int t = threadIdx.x;
    for (int i = t; i < RW; i += RW_T) {
        const uint32_t g = w0 + i;
        uint32_t id = 0xFFFFFFFFu; float c[3] = {INFINITY, INFINITY, INFINITY};
        if (g < n_faces) {
            id = order[g];
            const uint32_t* fv = faces + 3 * (size_t)id;
            for (int a = 0; a < 3; ++a) c[a] = (verts[3 * (size_t)fv[0] + a] + verts[3 * (size_t)fv[1] + a] + verts[3 * (size_t)fv[2] + a]) * (1.0f / 3.0f);
        }
        s_id[i] = id; s_c[0][i] = c[0]; s_c[1][i] = c[1]; s_c[2][i] = c[2]; s_ks[i] = (uint32_t)i;
    }
    __syncthreads();
    for (int seg = RW; seg > (int)LEAF_T; seg >>= 1) {
        const int nseg = RW / seg;
        for (int k = t; k < nseg * 6; k += RW_T) s_box[k / 6][k % 6] = (k % 6 < 3) ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        // centroid box of every segment: a wave holds 64 consecutive positions, i.e. whole segments or a 64-wide piece of
        // one -- butterfly min / max over min(seg, 64) lanes, then one lane per piece merges into LDS (same-address LDS
        // atomics from all 512 positions serialise 64-fold)
        for (int i = t; i < RW; i += RW_T) {
            const uint32_t sl = s_ks[i] & (uint32_t)(RW - 1);
            const bool ok = s_id[sl] != 0xFFFFFFFFu;
            uint32_t mn[3], mx[3];
            for (int a = 0; a < 3; ++a) { const uint32_t o = f2ord(s_c[a][sl]); mn[a] = ok ? o : 0xFFFFFFFFu; mx[a] = ok ? o : 0u; }
            const int w = seg < 64 ? seg : 64;
            for (int o = 1; o < w; o <<= 1)
                for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], (uint32_t)__shfl_xor(mn[a], o, 64)); mx[a] = max(mx[a], (uint32_t)__shfl_xor(mx[a], o, 64)); }
            if ((i & (w - 1)) == 0 && mn[0] != 0xFFFFFFFFu)
                for (int a = 0; a < 3; ++a) { atomicMin(&s_box[i / seg][a], mn[a]); atomicMax(&s_box[i / seg][3 + a], mx[a]); }
        }
        __syncthreads();
        for (int i = t; i < RW; i += RW_T) {
            const uint32_t* b = s_box[i / seg];
            int ax = 0;
            float best = 0.0f, lo = 0.0f;
            if (b[0] != 0xFFFFFFFFu) {   // segment holds at least one triangle
                const float e0 = ord2f(b[3]) - ord2f(b[0]), e1 = ord2f(b[4]) - ord2f(b[1]), e2 = ord2f(b[5]) - ord2f(b[2]);
                best = e0;
                if (e1 > best) { best = e1; ax = 1; }
                if (e2 > best) { best = e2; ax = 2; }
                lo = ord2f(b[ax]);
            }
            const uint32_t sl = s_ks[i] & (uint32_t)(RW - 1);
            uint32_t key = 0x1FFFFFu;                                      // padding behind the last triangle: the tail of every segment
            if (s_id[sl] != 0xFFFFFFFFu) key = best > 0.0f ? min((uint32_t)((s_c[ax][sl] - lo) / best * 2097151.0f), 0x1FFFFEu) : 0u;
            s_ks[i] = key << 11 | sl;
        }
        __syncthreads();
        // bitonic sort of every aligned segment of `seg` positions by (key, triangle id), ascending
        for (int k = 2; k <= seg; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int p = t;                                           // RW / 2 pairs, one per thread
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                const bool up = (k == seg) || ((i & k) == 0);
                const uint32_t ea = s_ks[i], eb = s_ks[l];
                const uint32_t ka = ea >> 11, kb = eb >> 11;
                // equal keys are ranked by the triangle id, not by the slot: the order a window ARRIVES in is not reproducible among ties
                // (k_kdorder.hip scatters with block-aggregated atomics), the order it leaves in must be
                const bool gt = ka > kb || (ka == kb && s_id[ea & (uint32_t)(RW - 1)] > s_id[eb & (uint32_t)(RW - 1)]);
                if (gt == up) { s_ks[i] = eb; s_ks[l] = ea; }
                // partners at distance j <= 64 live in the 128 positions this wave owns for all smaller j (LDS operations of
                // a wave execute in order): a block barrier is needed only while j > 64 or before the next k starts above 64
                if (j > 64 || (j == 1 && k >= 128)) __syncthreads();
            }
        __syncthreads();   // the next level (and the write-back) read positions other waves sorted
    }
    // padding keys are the largest: they stay at the tail of every segment, so the triangles are a prefix
    for (int i = t; i < RW; i += RW_T) { const uint32_t g = w0 + i; if (g < n_faces) order[g] = s_id[s_ks[i] & (uint32_t)(RW - 1)]; }
}

__device__ __forceinline__ float pad_from_box(const uint32_t* __restrict__ box) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = ord2f(box[a]); hi[a] = ord2f(box[3 + a]); }
    return scene_pad(lo, hi);
}

// triangles in curve order, 64 bytes each: {a, lo.x}, {e1 = b - a, lo.y}, {e2 = c - a, lo.z}, {hi, 0} with (lo, hi) = the
// triangle's box grown by pad (dmath.h tri_pad_box: the box clause of the hit predicate, evaluated once here instead of
// once per test -- 24 VALU instructions of every leaf test); slots >= n_faces are degenerate (never hit)
constexpr uint32_t TRI_F4 = 4;
__global__ void gather_tris_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, const uint32_t* __restrict__ order,
                                   uint32_t n_faces, uint32_t n_slots, const uint32_t* __restrict__ scene_box, float4* __restrict__ tris) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    float4 a = {0, 0, 0, 0}, e1 = {0, 0, 0, 0}, e2 = {0, 0, 0, 0}, h = {0, 0, 0, 0};
    if (s < n_faces) {
        const uint32_t* fv = faces + 3 * (size_t)(order ? order[s] : s);   // order == null: the faces are in curve order themselves
        const V3 pa = {verts[3 * (size_t)fv[0]], verts[3 * (size_t)fv[0] + 1], verts[3 * (size_t)fv[0] + 2]};
        const V3 pb = {verts[3 * (size_t)fv[1]], verts[3 * (size_t)fv[1] + 1], verts[3 * (size_t)fv[1] + 2]};
        const V3 pc = {verts[3 * (size_t)fv[2]], verts[3 * (size_t)fv[2] + 1], verts[3 * (size_t)fv[2] + 2]};
        const V3 d1 = pb - pa, d2 = pc - pa;
        V3 lo, hi;
        tri_pad_box(pa, d1, d2, pad_from_box(scene_box), &lo, &hi);
        a = {pa.x, pa.y, pa.z, lo.x}; e1 = {d1.x, d1.y, d1.z, lo.y}; e2 = {d2.x, d2.y, d2.z, lo.z}; h = {hi.x, hi.y, hi.z, 0.0f};
    }
    float4* t = tris + TRI_F4 * (size_t)s;
    t[0] = a; t[1] = e1; t[2] = e2; t[3] = h;
}
__device__ __forceinline__ bool tri_hit(const float4 A, const float4 E1, const float4 E2, const float4 H, const Ray& r) {
    return ray_tri_boxed(r, V3{A.x, A.y, A.z}, V3{E1.x, E1.y, E1.z}, V3{E2.x, E2.y, E2.z}, V3{A.w, E1.w, E2.w}, V3{H.x, H.y, H.z});
}

// level 0: node n -> child c = leaf 4n + c = triangles LEAF_T * (4n + c) .. + LEAF_T - 1.  Also emits the node's own box.
__global__ void build_level0_kernel(const float4* __restrict__ tris, uint32_t n_faces, uint32_t n_leaves, uint32_t n_nodes,
                                    const uint32_t* __restrict__ scene_box, Node4* __restrict__ nodes, float* __restrict__ own_box) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    float slo[3], shi[3];
    for (int a = 0; a < 3; ++a) { slo[a] = ord2f(scene_box[a]); shi[a] = ord2f(scene_box[3 + a]); }
    const float pad = scene_pad(slo, shi);
    Node4 nd;
    float nlo[3] = {INFINITY, INFINITY, INFINITY}, nhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t nchild = 0;
    for (int c = 0; c < 4; ++c) {
        const uint32_t leaf = 4 * n + c;
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {1e30f, 1e30f, 1e30f};
        if (leaf < n_leaves) {
            nchild = c + 1;
            for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
            for (uint32_t t = LEAF_T * leaf; t < LEAF_T * leaf + LEAF_T && t < n_faces; ++t) {
                const float4 A = tris[TRI_F4 * (size_t)t], E1 = tris[TRI_F4 * (size_t)t + 1], E2 = tris[TRI_F4 * (size_t)t + 2];
                const float pa[3] = {A.x, A.y, A.z};
                const float pb[3] = {A.x + E1.x, A.y + E1.y, A.z + E1.z};
                const float pc[3] = {A.x + E2.x, A.y + E2.y, A.z + E2.z};
                for (int a = 0; a < 3; ++a) {
                    lo[a] = fminf(lo[a], fminf(pa[a], fminf(pb[a], pc[a])));
                    hi[a] = fmaxf(hi[a], fmaxf(pa[a], fmaxf(pb[a], pc[a])));
                }
            }
            // ray_tri accepts only hit points inside (triangle box + pad): 4*pad keeps the slab test conservative
            for (int a = 0; a < 3; ++a) { lo[a] -= 4.0f * pad; hi[a] += 4.0f * pad; nlo[a] = fminf(nlo[a], lo[a]); nhi[a] = fmaxf(nhi[a], hi[a]); }
        }
        for (int a = 0; a < 3; ++a) { nd.b[c][a] = lo[a]; nd.b[c][3 + a] = hi[a]; }
    }
    nd.nchild = nchild;
    for (int k = 0; k < 7; ++k) nd.pad_[k] = 0;
    nodes[n] = nd;
    for (int a = 0; a < 3; ++a) { own_box[6 * (size_t)n + a] = nlo[a]; own_box[6 * (size_t)n + 3 + a] = nhi[a]; }
}

__global__ void build_level_kernel(const float* __restrict__ child_box, uint32_t n_children, uint32_t n_nodes,
                                   Node4* __restrict__ nodes, float* __restrict__ own_box) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    Node4 nd;
    float nlo[3] = {INFINITY, INFINITY, INFINITY}, nhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t nchild = 0;
    for (int c = 0; c < 4; ++c) {
        const uint32_t ch = 4 * n + c;
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {1e30f, 1e30f, 1e30f};
        if (ch < n_children) {
            nchild = c + 1;
            for (int a = 0; a < 3; ++a) {
                lo[a] = child_box[6 * (size_t)ch + a]; hi[a] = child_box[6 * (size_t)ch + 3 + a];
                nlo[a] = fminf(nlo[a], lo[a]); nhi[a] = fmaxf(nhi[a], hi[a]);
            }
        }
        for (int a = 0; a < 3; ++a) { nd.b[c][a] = lo[a]; nd.b[c][3 + a] = hi[a]; }
    }
    nd.nchild = nchild;
    for (int k = 0; k < 7; ++k) nd.pad_[k] = 0;
    nodes[n] = nd;
    for (int a = 0; a < 3; ++a) { own_box[6 * (size_t)n + a] = nlo[a]; own_box[6 * (size_t)n + 3 + a] = nhi[a]; }
}

// ---- vertex -> incident faces (for the need bits) ----
__global__ void vf_count_kernel(const uint32_t* __restrict__ faces, uint32_t n_faces, uint32_t* __restrict__ deg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n_faces) return;
    atomicAdd(&deg[faces[i]], 1u);
}
__global__ void vf_fill_kernel(const uint32_t* __restrict__ faces, uint32_t n_faces, const uint32_t* __restrict__ vf_ptr,
                               uint32_t* __restrict__ cursor, uint32_t* __restrict__ vf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n_faces) return;
    const uint32_t v = faces[i];
    const uint32_t k = atomicAdd(&cursor[v], 1u);
    vf[vf_ptr[v] + k] = i / 3;  // order within a vertex is irrelevant: only OR-ed
}

// ---- traversal: one wave = 64 rays from one surface patch to one camera sharing ONE traversal ----
// The 64 rays of a wave (64 vertices consecutive on the Hilbert curve, one view) walk the tree together: a node is visited when
// ANY live ray enters its box, so the node is one wave-uniform fetch (two wide scalar loads of the 128-byte line) and the child
// boxes are tested by all lanes at once.  At a leaf only the rays that enter the leaf's box are candidates (5 - 6 of 64), so the
// wave tests (candidate, triangle) PAIRS: a round = 4 candidates x 16 triangles, one pair per lane, rays parked in LDS, the
// candidate list through LDS, results back to the owning lanes through ballots.  The occlusion predicate is the OR over all
// triangles (dmath.h ray_tri: conservative box clause), so the booleans do not depend on the traversal -- the oracle's BVH and its
// brute force give the same bits.  Earlier traversals of this file (one per ray; shared without redistribution; round 2 of this
// one; top levels in LDS) are in the history -- profiles/r02_lds_bvh_ab.txt and DESIGN.md record what they measured.
// Instruction diet of a node visit (the kernel had been issue bound on both ports):
//  * slab test with packed math: a child's six bounds are three aligned SGPR pairs, so the six t = bound * inv - o * inv
//    are three v_pk_fma_f32 (full rate with a scalar-pair operand on gfx950: scripts/probe/pk_probe.hip);
//  * direction-sign specialisation: all rays of a packet go from one surface patch to one camera, so nearly always every
//    live lane has the same sign of d on each axis.  Then near / far planes are known statically -- near_x is lo.x if
//    d.x > 0 else hi.x -- and min / max per axis disappear: tn = max(max3(near), t0), tf = min(min3(far), t1), 5 instead of
//    13 instructions per child after the fma's.  fma is monotone in the bound for a fixed finite multiplier, so
//    min(t_lo, t_hi) IS t_near bit for bit: the hit masks equal the generic test's.  Packets with mixed signs, zero or
//    tiny direction components (|1/d| >= 1e30 or non-finite) take the generic variant OCT = 8 (fmin / fmax per axis);
//  * no per-level climbing: the pending-children nibbles of all levels sit in one 64-bit mask, s_ff1 finds the deepest
//    pending child, the ancestor's position in its level is a shift (heap order: node = (4^depth - 1) / 3 + position);
//  * a level-0 node's leaves are processed right after its visit, from the hit ballots still in registers;
//  * absent children are far-away points (build kernels), so the child-count mask is not needed.
// COUNT (option "count_rays"): node visits, leaf visits and leaf rounds of the launch -- the N_ray_nodes / N_ray_tris of the
// roofline accounting -- at the price of three atomics per wave.
typedef float f2 __attribute__((ext_vector_type(2)));

// m = 2 m + (mask != 0): shifts the "some live ray enters child c" bit into the child mask with two scalar instructions
__device__ __forceinline__ uint32_t shift_in_nonzero(uint32_t m, unsigned long long mask) {
    asm("s_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, %0" : "+s"(m) : "s"(mask) : "scc");
    return m;
}
__device__ __forceinline__ unsigned long long clear_bit(unsigned long long x, uint32_t bit) {
    asm("s_bitset0_b64 %0, %1" : "+s"(x) : "s"(bit));
    return x;
}

template <int OCT, bool COUNT>
__device__ __forceinline__ void packet3_traverse(const BvhDev& bvh, const float4 (*s_ray)[2], uint8_t* s_src, const int lane, const float pad,
                                                 const V3 inv, const V3 oi, const float t0, float t1,
                                                 unsigned long long& actm /* in: live rays; out: live rays never occluded */, uint32_t (&cnt)[3] /* node visits, leaf visits, rounds */) {
    constexpr bool SX = (OCT & 1) != 0, SY = (OCT & 2) != 0, SZ = (OCT & 4) != 0, GEN = OCT >= 8;
    const f2 I0 = {inv.x, inv.y}, I1 = {inv.z, inv.x}, I2 = {inv.y, inv.z};
    const f2 O0 = {-oi.x, -oi.y}, O1 = {-oi.z, -oi.x}, O2 = {-oi.y, -oi.z};
    const Node4* __restrict__ nodes = bvh.nodes;
    // A lane that is not live (never needed, or already occluded) gets the empty interval t1 = -1 < 0 <= t0: it enters no
    // box, so the hit ballots need no masking.
    t1 = __builtin_amdgcn_inverse_ballot_w64(actm) ? t1 : -1.0f;
    unsigned long long hm0 = 0ull, hm1 = 0ull, hm2 = 0ull, hm3 = 0ull;   // separate scalars, never an indexed array (that would live in scratch)
    // fetch(h): the 96 bytes of bounds of node h (h wave-uniform) arrive with two wide scalar loads; test(nd): the slab tests of its four children
    auto fetch = [&](uint32_t h) -> Node4 {
        if (COUNT) ++cnt[0];
        return *reinterpret_cast<const Node4*>(reinterpret_cast<const char*>(nodes) + (h << 7));   // 32-bit byte offset (build_bvh bounds h): base + offset addressing
    };
    auto test = [&](const Node4& nd) -> uint32_t {
        uint32_t m = 0;
#pragma unroll
        for (int c = 3; c >= 0; --c) {
            const f2 P0 = {nd.b[c][0], nd.b[c][1]}, P1 = {nd.b[c][2], nd.b[c][3]}, P2 = {nd.b[c][4], nd.b[c][5]};
            const f2 R0 = __builtin_elementwise_fma(P0, I0, O0);   // t(lo.x), t(lo.y)
            const f2 R1 = __builtin_elementwise_fma(P1, I1, O1);   // t(lo.z), t(hi.x)
            const f2 R2 = __builtin_elementwise_fma(P2, I2, O2);   // t(hi.y), t(hi.z)
            float tn, tf;
            if (GEN) {   // the round-2 sequence (fmin / fmax drop the NaN of inf - inf)
                tn = fmaxf(t0, fminf(R0.x, R1.y)); tf = fminf(t1, fmaxf(R0.x, R1.y));
                tn = fmaxf(tn, fminf(R0.y, R2.x)); tf = fminf(tf, fmaxf(R0.y, R2.x));
                tn = fmaxf(tn, fminf(R1.x, R2.y)); tf = fminf(tf, fmaxf(R1.x, R2.y));
            } else {
                const float nx = SX ? R1.y : R0.x, fx = SX ? R0.x : R1.y;
                const float ny = SY ? R2.x : R0.y, fy = SY ? R0.y : R2.x;
                const float nz = SZ ? R2.y : R1.x, fz = SZ ? R1.x : R2.y;
                tn = fmaxf(fmaxf(fmaxf(nx, ny), nz), t0);
                tf = fminf(fminf(fminf(fx, fy), fz), t1);
            }
            const unsigned long long hb = __builtin_amdgcn_ballot_w64(tn <= tf);
            if (c == 0) hm0 = hb; else if (c == 1) hm1 = hb; else if (c == 2) hm2 = hb; else hm3 = hb;
            m = shift_in_nonzero(m, hb);
        }
        return m;
    };
    auto visit = [&](uint32_t h) -> uint32_t { const Node4 nd = fetch(h); return test(nd); };
    // the leaves (children c of the level-0 node h0, c in m) whose boxes some live ray enters; stops when no ray is left (actm == 0)
    auto leaves = [&](uint32_t h0, uint32_t m) {
        const uint32_t leaf0 = (h0 - bvh.level_off[0]) * 4u;
        while (m != 0u) {
            const uint32_t c = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            // three scalar selects; the empty asm keeps the compiler from turning the chain into an indexed table in scratch
            unsigned long long hsel = (c == 1u) ? hm1 : hm0;
            asm("" : "+s"(hsel));
            hsel = (c == 2u) ? hm2 : hsel;
            asm("" : "+s"(hsel));
            hsel = (c == 3u) ? hm3 : hsel;
            const unsigned long long cb = hsel & actm;          // rays that are still live and enter this leaf's box
            if (cb != 0ull) {
                const bool cand = __builtin_amdgcn_inverse_ballot_w64(cb);
                const int n = __builtin_popcountll(cb);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cb, 0u));
                if (cand) s_src[rank] = (uint8_t)lane;
                const float4* __restrict__ tp = bvh.tris + TRI_F4 * (size_t)((leaf0 + c) * LEAF_T + ((uint32_t)lane & (LEAF_T - 1u)));
                const float4 A = tp[0], E1 = tp[1], E2 = tp[2], H = tp[3];
                unsigned long long hitm = 0ull;
                if (COUNT) { ++cnt[1]; cnt[2] += (uint32_t)((n + LEAF_SLOTS - 1) / LEAF_SLOTS); }
                for (int base = 0; base < n; base += LEAF_SLOTS) {
                    const int q = base + lane / (int)LEAF_T;     // slots >= n hold lane numbers of earlier lists (or the initial ones): harmless work, masked below
                    const int sl = (int)s_src[q];
                    const float4 r0 = s_ray[sl][0], r1 = s_ray[sl][1];
                    Ray rr; rr.o = V3{r0.x, r0.y, r0.z}; rr.tmin = r0.w; rr.d = V3{r1.x, r1.y, r1.z}; rr.tmax = r1.w; rr.pad = pad;
                    const bool h = tri_hit(A, E1, E2, H, rr) && q < n;
                    const unsigned long long hb = __builtin_amdgcn_ballot_w64(h);
                    if (hb != 0ull) {   // wave-uniform and rare: the owner of slot (rank - base) reads its LEAF_T result bits
                        const bool mine = cand && rank >= base && rank < base + LEAF_SLOTS &&
                                          ((hb >> ((int)LEAF_T * (rank - base))) & ((1ull << LEAF_T) - 1ull)) != 0ull;
                        hitm |= __builtin_amdgcn_ballot_w64(mine);
                    }
                }
                if (hitm != 0ull) {
                    actm &= ~hitm;
                    t1 = __builtin_amdgcn_inverse_ballot_w64(actm) ? t1 : -1.0f;
                    if (actm == 0ull) m = 0u;
                }
            }
        }
    };
    const uint32_t top = (uint32_t)bvh.top;
    uint32_t m = visit(1u);
    if (top == 0u) { leaves(1u, m); return; }
    unsigned long long masks = (unsigned long long)m << (4u * top);   // nibble L: hit children not yet visited of the current ancestor on level L
    uint32_t level = top, h = 1u;                                     // current node: heap index h on `level`
    if (masks == 0ull) return;
    const char* __restrict__ nbase = reinterpret_cast<const char*>(nodes);
    (void)nbase;
#ifndef MVS_RAY_PREFETCH
    do {
        const uint32_t idx = (uint32_t)__builtin_ctzll(masks);        // deepest level first = depth first
        masks = clear_bit(masks, idx);
        const uint32_t lv = idx >> 2;
        const uint32_t ch = ((h >> (2u * (lv - level))) << 2) | (idx & 3u);   // the ancestor on level lv, then its child
        m = visit(ch);
        if (lv != 1u) { masks |= (unsigned long long)m << (4u * (lv - 1u)); h = ch; level = lv - 1u; }
        else { h = ch >> 2; level = 1u; leaves(ch, m); masks = (actm == 0ull) ? 0ull : masks; }
    } while (masks != 0ull);
#else
    // (experiment, round 5: the node that follows a level-0 node in the walk does not depend on that node's leaves -- its fetch is issued
    //  BEFORE the leaves are processed, so its latency hides behind the triangle fetches and the rounds; profiles/EXPERIMENTS.md)
    uint32_t idx = (uint32_t)__builtin_ctzll(masks);
    masks = clear_bit(masks, idx);
    uint32_t lv = idx >> 2, ch = ((h >> (2u * (lv - level))) << 2) | (idx & 3u);
    Node4 nd = fetch(ch);
    for (;;) {
        m = test(nd);
        if (lv != 1u) {
            masks |= (unsigned long long)m << (4u * (lv - 1u)); h = ch; level = lv - 1u;
            if (masks == 0ull) break;
            idx = (uint32_t)__builtin_ctzll(masks); masks = clear_bit(masks, idx);
            lv = idx >> 2; ch = ((h >> (2u * (lv - level))) << 2) | (idx & 3u);
            nd = fetch(ch);
        } else {
            const uint32_t leaf_node = ch;
            h = ch >> 2; level = 1u;
            const bool more = masks != 0ull;
            if (more) {
                idx = (uint32_t)__builtin_ctzll(masks); masks = clear_bit(masks, idx);
                lv = idx >> 2; ch = ((h >> (2u * (lv - level))) << 2) | (idx & 3u);
                nd = fetch(ch);
            }
            leaves(leaf_node, m);
            if (!more || actm == 0ull) break;
        }
    }
#endif
}

// Residency: a 256-thread block is admitted per CU up to floor(800 / (ceil16(sgpr) + 16)) times (MI355X_MICROARCH.md "Residency").  The 98
// SGPRs the compiler takes by itself allow 6 blocks; amdgpu_num_sgpr(80) allows 8, and __launch_bounds__(256, 8) keeps the VGPRs at
// the 64 that 8 waves per SIMD leave (no spills).  The kernel is a chain of dependent node and triangle fetches per wave -- PMC
// (profiles/r03a_pmc_sq_c3.json): vector issue 41 %, waves parked on memory 46 % of their time -- so residency is throughput:
// 6.63 -> 6.33 (7 blocks) -> 6.02 ms (8 blocks) at BASELINE config 3, A/B of builds on one box (scripts/ab_libs.py).
template <bool XCD, bool COUNT>
__global__ void __launch_bounds__(256, 8) __attribute__((amdgpu_num_sgpr(80))) ray_packet3_kernel(const BvhDev bvh, const float* __restrict__ verts /* curve order: ctx->iv */,
                                                          const ViewParams* __restrict__ views, const unsigned long long* __restrict__ need,
                                                          unsigned long long* __restrict__ occl, uint32_t vwords, uint32_t n_verts, uint32_t n_views,
                                                          const uint32_t* __restrict__ scene_box, unsigned long long* __restrict__ counters) {
    __shared__ float4 s_ray[4][64][2];
    __shared__ uint8_t s_src[4][80];   // candidate lists; a round reads up to LEAF_SLOTS - 1 slots past the list
    // XCD-aware order: blocks are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), each with its own L2; with this map every
    // XCD walks a contiguous eighth of the (view, patch) sequence, so neighbouring packets -- which visit the same nodes -- share an L2
    uint32_t vblk = blockIdx.x;
    if (XCD && (gridDim.x & 7u) == 0u) vblk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint64_t wave = ((uint64_t)vblk * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wave >= (uint64_t)vwords * n_views) return;
    const uint32_t j = (uint32_t)(wave / vwords), vw = (uint32_t)(wave % vwords);
    const unsigned long long word = need[(size_t)j * vwords + vw];
    if (word == 0ull) return;  // occl is pre-zeroed
    const uint32_t s = vw * 64 + lane;
    const bool active = ((word >> lane) & 1ull) && s < n_verts;
    const uint32_t v = s < n_verts ? s : 0;
    const ViewParams& vp = views[j];
    const V3 o = {verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2]};
    const float pad = pad_from_box(scene_box);
    const Ray r = make_ray(o, V3{vp.pos[0], vp.pos[1], vp.pos[2]}, pad);
    s_ray[wv][lane][0] = make_float4(r.o.x, r.o.y, r.o.z, r.tmin);
    s_ray[wv][lane][1] = make_float4(r.d.x, r.d.y, r.d.z, r.tmax);
    s_src[wv][lane] = (uint8_t)lane; if (lane < 16) s_src[wv][64 + lane] = (uint8_t)lane;
    const V3 inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    const V3 oi = {r.o.x * inv.x, r.o.y * inv.y, r.o.z * inv.z};
    const float t0 = r.tmin * 0.999f, t1 = r.tmax * 1.001f;
    const unsigned long long actm = __builtin_amdgcn_ballot_w64(active);
    unsigned long long live = actm;
    // direction-sign octant of the packet over its live lanes; 8 = mixed / degenerate
    const float big = 1e30f;
    const unsigned long long okm = __builtin_amdgcn_ballot_w64(fabsf(inv.x) < big && fabsf(inv.y) < big && fabsf(inv.z) < big) & actm;
    const unsigned long long nxm = __builtin_amdgcn_ballot_w64(inv.x < 0.0f) & actm, nym = __builtin_amdgcn_ballot_w64(inv.y < 0.0f) & actm,
                             nzm = __builtin_amdgcn_ballot_w64(inv.z < 0.0f) & actm;
    int oct = 8;
    if (okm == actm && (nxm == 0ull || nxm == actm) && (nym == 0ull || nym == actm) && (nzm == 0ull || nzm == actm))
        oct = (nxm ? 1 : 0) | (nym ? 2 : 0) | (nzm ? 4 : 0);
    if (counters && lane == 0) { atomicAdd(&counters[10], 1ull); if (oct == 8) atomicAdd(&counters[11], 1ull); }   // diagnostics ("stats" option)
    uint32_t cnt[3] = {0u, 0u, 0u};
#define MVS_P3(O) packet3_traverse<O, COUNT>(bvh, s_ray[wv], s_src[wv], lane, pad, inv, oi, t0, t1, live, cnt)
    switch (oct) {
        case 0: MVS_P3(0); break; case 1: MVS_P3(1); break; case 2: MVS_P3(2); break; case 3: MVS_P3(3); break;
        case 4: MVS_P3(4); break; case 5: MVS_P3(5); break; case 6: MVS_P3(6); break; case 7: MVS_P3(7); break;
        default: MVS_P3(8); break;
    }
#undef MVS_P3
    if (lane == 0) occl[(size_t)j * vwords + vw] = actm & ~live;
    if (COUNT && lane == 0) { atomicAdd(&counters[8], (unsigned long long)cnt[0]); atomicAdd(&counters[9], (unsigned long long)cnt[1]); atomicAdd(&counters[13], (unsigned long long)cnt[2]); }
}

}  // namespace

// vertex -> incident faces (CSR vf_ptr / vf); order inside a vertex is arbitrary (atomic cursor): consumers only OR / search
void build_vertex_faces(mvs_ctx* ctx, const uint32_t* d_faces, uint32_t F, uint32_t NV) {
    hipStream_t s = ctx->stream;
    ctx->vf_ptr.ensure((size_t)NV + 1); ctx->vf_cursor.ensure((size_t)NV + 1); ctx->vf.ensure(3 * (size_t)F + 1);
    MVS_HIP(hipMemsetAsync(ctx->vf_cursor.p, 0, ((size_t)NV + 1) * sizeof(uint32_t), s));
    if (F) { hipLaunchKernelGGL(vf_count_kernel, dim3((3 * F + 255) / 256), dim3(256), 0, s, d_faces, F, ctx->vf_cursor.p); MVS_LAUNCH_CHECK(); }
    exclusive_scan_u32(ctx, ctx->vf_cursor.p, ctx->vf_ptr.p, (size_t)NV + 1, nullptr);
    MVS_HIP(hipMemsetAsync(ctx->vf_cursor.p, 0, ((size_t)NV + 1) * sizeof(uint32_t), s));
    if (F) { hipLaunchKernelGGL(vf_fill_kernel, dim3((3 * F + 255) / 256), dim3(256), 0, s, d_faces, F, ctx->vf_ptr.p, ctx->vf_cursor.p, ctx->vf.p); MVS_LAUNCH_CHECK(); }
}

// ---- the library's own mesh layout (ctx.h "the library's own mesh layout") ----
namespace {
__global__ void gather_verts_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ vperm, uint32_t n_verts, float* __restrict__ out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_verts) return;
    const size_t v = vperm[s];
    out[3 * (size_t)s] = verts[3 * v]; out[3 * (size_t)s + 1] = verts[3 * v + 1]; out[3 * (size_t)s + 2] = verts[3 * v + 2];
}
// out[p] = the face order[p] (null: p) of `faces` with its vertex indices mapped through vpos (null: kept)
__global__ void remap_faces_kernel(const uint32_t* __restrict__ faces, const uint32_t* __restrict__ order, const uint32_t* __restrict__ vpos, uint32_t n_faces,
                                   uint32_t* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_faces) return;
    const size_t f = order ? order[p] : p;
    const uint32_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
    out[3 * (size_t)p] = vpos ? vpos[a] : a; out[3 * (size_t)p + 1] = vpos ? vpos[b] : b; out[3 * (size_t)p + 2] = vpos ? vpos[c] : c;
}
// f_perm[p] = first[loc[p]] (the caller's id of the face at position p), f_pos = its inverse, normals gathered along
__global__ void finish_order_kernel(const uint32_t* __restrict__ first, const uint32_t* __restrict__ loc, const float* __restrict__ normals, uint32_t n_faces,
                                    uint32_t* __restrict__ f_perm, uint32_t* __restrict__ f_pos, float* __restrict__ out_normals) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_faces) return;
    const size_t f = first[loc[p]];
    f_perm[p] = (uint32_t)f; f_pos[f] = p;
    out_normals[3 * (size_t)p] = normals[3 * f]; out_normals[3 * (size_t)p + 1] = normals[3 * f + 1]; out_normals[3 * (size_t)p + 2] = normals[3 * f + 2];
}
__global__ void iota_kernel(uint32_t* __restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
void sort_by_key30(mvs_ctx* ctx, uint32_t* k_in, uint32_t* k_out, uint32_t* v_in, uint32_t* v_out, uint32_t n) {
    size_t tmp_bytes = 0;
    MVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, n, 0, 30, ctx->stream));
    ctx->sort_tmp.ensure(tmp_bytes + 16);
    MVS_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, n, 0, 30, ctx->stream));
}
}  // namespace

void kd_refine_order(mvs_ctx* ctx, const float* verts, const uint32_t* faces, uint32_t* order, uint32_t F, uint32_t window, uint32_t leaf_window);   // k_kdorder.hip
void build_scene_order(mvs_ctx* ctx);

// The upper levels of the order report "a cut had more equal keys than its tie list holds" through pinned memory, without a wait of their own
// (k_kdorder.hip).  Called where the caller synchronises anyway: true = that happened, the order has been rebuilt WITHOUT upper levels (and
// stays so until another mesh arrives) -- whatever was derived from the order since (BVH, bit matrices) must be derived again.
bool scene_order_commit(mvs_ctx* ctx) {
    if (!ctx->kd_pending) return false;
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    bool overflow = false;
    for (int l = 0; l < ctx->kd_pending; ++l) overflow = overflow || ctx->h_kd_flags[l] != 0u;
    ctx->kd_pending = 0;
    if (!overflow) return false;
    if (ctx->verbose) fprintf(stderr, "[mvs] face order: more equal centroid coordinates at a cut than the tie list holds; keeping the curve order above %d faces\n", 2048);
    ctx->kd_disabled = true;
    build_scene_order(ctx);
    return true;
}

// Lays the resident mesh out along a Hilbert curve: ctx->iv / ifc / inr (see ctx.h) and, with option "face_order" != 0, the face
// permutation f_perm / f_pos.  The caller's arrays are read with gathers exactly three times (the keys, the faces, the normals);
// everything after this function streams the copy.  Deterministic (stable sorts, ties by id): every rank of a sharded run derives
// the same order from the replicated mesh.
void build_scene_order(mvs_ctx* ctx) {
    const uint32_t F = ctx->n_faces, NV = ctx->n_verts;
    hipStream_t s = ctx->stream;
    ctx->scene_box.ensure(8 + 6 * 512);
    uint32_t* box = (uint32_t*)ctx->scene_box.p;
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    MVS_HIP(hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, s));
    ctx->mesh_ordered = false; ctx->tri_order = nullptr; ctx->kd_pending = 0;
    ctx->iv = ctx->d_verts; ctx->ifc = ctx->d_faces; ctx->inr = ctx->d_normals;
    if (F == 0 || NV == 0) return;
    { const uint32_t nb = std::min<uint32_t>((NV + 255) / 256, 512u);
      hipLaunchKernelGGL(bbox_kernel, dim3(nb), dim3(256), 0, s, ctx->d_verts, NV, box + 8); MVS_LAUNCH_CHECK();
      hipLaunchKernelGGL(bbox_fold_kernel, dim3(1), dim3(64), 0, s, (const uint32_t*)(box + 8), nb, box); MVS_LAUNCH_CHECK(); }
    const size_t n_max = std::max<size_t>(F, NV);
    ctx->sort_k.ensure(n_max); ctx->sort_k2.ensure(n_max); ctx->sort_v.ensure(n_max); ctx->sort_v2.ensure(F);
    // vertices along the curve (rays are launched in this order: 64 neighbouring vertices per wave = a compact patch of nearly parallel rays)
    ctx->vperm.ensure((size_t)NV + 1); ctx->vpos.ensure((size_t)NV + 1); ctx->i_verts.ensure(3 * (size_t)NV + 4);
    hipLaunchKernelGGL(vertex_key_kernel, dim3((NV + 255) / 256), dim3(256), 0, s, ctx->d_verts, NV, box, ctx->sort_k.p, ctx->sort_v.p);
    MVS_LAUNCH_CHECK();
    sort_by_key30(ctx, ctx->sort_k.p, ctx->sort_k2.p, ctx->sort_v.p, ctx->vperm.p, NV);
    hipLaunchKernelGGL(invert_perm_kernel, dim3((NV + 255) / 256), dim3(256), 0, s, ctx->vperm.p, NV, ctx->vpos.p);
    MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(gather_verts_kernel, dim3((NV + 255) / 256), dim3(256), 0, s, ctx->d_verts, ctx->vperm.p, NV, ctx->i_verts.p);
    MVS_LAUNCH_CHECK();
    ctx->iv = ctx->i_verts.p;
    ctx->i_faces.ensure(3 * (size_t)F + 4);
    static_assert(RW == 2 * RW_T && RW % (int)LEAF_T == 0 && RW >= 2048, "one pair per thread; k_kdorder.hip hands over at RW");
    const dim3 fg((F + 255) / 256), rg((F + RW - 1) / RW);
    // upper levels: exact top-down median cuts inside windows of ctx->bvh_window positions of the curve (0: the whole mesh; 1: none = the
    // order of rounds 1 - 5 with a larger LDS window); a mesh with thousands of EQUAL centroid coordinates at a cut keeps the curve order
    auto upper_levels = [&](const float* v, const uint32_t* f, uint32_t* order) {
        // (the upper levels cost ~0.4 ms of launches whatever the mesh and pay through the ray stage: measured at 2 M faces x 200 views
        //  -0.76 ms of rays for +0.50 ms here, at 200 k faces x 200 views -0.05 for +0.38 -- they run from bvh_upper_min_faces faces on; the rule
        //  looks at the MESH alone, so that mvs_partition_faces and every rank of a sharded run derive the same order whatever views are set)
        if (ctx->bvh_window != 1u && F >= ctx->bvh_upper_min_faces && !ctx->kd_disabled) kd_refine_order(ctx, v, f, order, F, ctx->bvh_window, (uint32_t)RW);
    };
    if (ctx->face_order != 0) {
        // faces along the curve: keys from the caller's arrays, the refinement and everything later on the copy
        ctx->f_tmp.ensure(3 * (size_t)F + 4); ctx->f_perm.ensure((size_t)F + 1); ctx->f_pos.ensure((size_t)F + 1); ctx->i_normals.ensure(3 * (size_t)F + 4);
        hipLaunchKernelGGL(curve_key_kernel, fg, dim3(256), 0, s, ctx->d_verts, ctx->d_faces, F, box, ctx->sort_k.p, ctx->sort_v.p);
        MVS_LAUNCH_CHECK();
        sort_by_key30(ctx, ctx->sort_k.p, ctx->sort_k2.p, ctx->sort_v.p, ctx->sort_v2.p, F);
        hipLaunchKernelGGL(remap_faces_kernel, fg, dim3(256), 0, s, ctx->d_faces, (const uint32_t*)ctx->sort_v2.p, (const uint32_t*)ctx->vpos.p, F, ctx->f_tmp.p);
        MVS_LAUNCH_CHECK();
        hipLaunchKernelGGL(iota_kernel, fg, dim3(256), 0, s, ctx->sort_v.p, F);
        MVS_LAUNCH_CHECK();
        upper_levels(ctx->i_verts.p, ctx->f_tmp.p, ctx->sort_v.p);
        hipLaunchKernelGGL(refine_order_kernel, rg, dim3(RW_T), 0, s, (const float*)ctx->i_verts.p, (const uint32_t*)ctx->f_tmp.p, ctx->sort_v.p, F);
        MVS_LAUNCH_CHECK();
        hipLaunchKernelGGL(remap_faces_kernel, fg, dim3(256), 0, s, (const uint32_t*)ctx->f_tmp.p, (const uint32_t*)ctx->sort_v.p, (const uint32_t*)nullptr, F, ctx->i_faces.p);
        MVS_LAUNCH_CHECK();
        hipLaunchKernelGGL(finish_order_kernel, fg, dim3(256), 0, s, (const uint32_t*)ctx->sort_v2.p, (const uint32_t*)ctx->sort_v.p, ctx->d_normals, F, ctx->f_perm.p, ctx->f_pos.p, ctx->i_normals.p);
        MVS_LAUNCH_CHECK();
        ctx->ifc = ctx->i_faces.p; ctx->inr = ctx->i_normals.p; ctx->mesh_ordered = true;
    } else {
        // the caller's face numbering is kept: only the BVH's triangle slots follow the curve
        hipLaunchKernelGGL(remap_faces_kernel, fg, dim3(256), 0, s, ctx->d_faces, (const uint32_t*)nullptr, (const uint32_t*)ctx->vpos.p, F, ctx->i_faces.p);
        MVS_LAUNCH_CHECK();
        ctx->ifc = ctx->i_faces.p;
        if (ctx->bvh_caller_order) { ctx->tri_order = nullptr; return; }   // experiment hook: the BVH's triangle slots ARE the caller's face order (scripts/bvh_order_probe.py)
        hipLaunchKernelGGL(curve_key_kernel, fg, dim3(256), 0, s, ctx->iv, ctx->ifc, F, box, ctx->sort_k.p, ctx->sort_v.p);
        MVS_LAUNCH_CHECK();
        sort_by_key30(ctx, ctx->sort_k.p, ctx->sort_k2.p, ctx->sort_v.p, ctx->sort_v2.p, F);
        upper_levels(ctx->iv, ctx->ifc, ctx->sort_v2.p);
        hipLaunchKernelGGL(refine_order_kernel, rg, dim3(RW_T), 0, s, ctx->iv, ctx->ifc, ctx->sort_v2.p, F);
        MVS_LAUNCH_CHECK();
        ctx->tri_order = ctx->sort_v2.p;
    }
}

// Builds the BVH and the vertex->face incidence for the resident mesh in the layout of build_scene_order.
void build_bvh(mvs_ctx* ctx) {
    const uint32_t F = ctx->n_faces, NV = ctx->n_verts;
    hipStream_t s = ctx->stream;
    const uint32_t n_leaves = (F + LEAF_T - 1) / LEAF_T;
    const uint32_t n_slots = ((n_leaves + 3) / 4) * 4 * LEAF_T;  // triangles padded to whole level-0 nodes
    uint32_t* box = (uint32_t*)ctx->scene_box.p;
    ctx->bvh_tris.ensure(TRI_F4 * (size_t)n_slots);
    hipLaunchKernelGGL(gather_tris_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, ctx->iv, ctx->ifc, ctx->tri_order, F, n_slots, box, ctx->bvh_tris.p);
    MVS_LAUNCH_CHECK();
    // level sizes
    BvhDev b{};
    uint32_t cnt = (n_leaves + 3) / 4; int L = 0;
    while (true) {
        if (L >= 16) throw HipError("BVH too deep");
        b.level_cnt[L] = cnt;
        if (cnt == 1) break;
        cnt = (cnt + 3) / 4; ++L;
    }
    b.top = L; b.n_leaves = n_leaves;
    // Levels are stored in 4-ary heap order with the root at index 1: level L (depth d = top - L) starts at 4^d, the node at
    // position p of its level is h = 4^d + p, its children are 4h + c, its ancestor k levels up is h >> 2k -- the traversal
    // needs no per-level table and climbs any number of levels with one shift.  Slots between the end of a level and the
    // next level's start are never touched (absent children are far-away points no ray reaches).
    for (int l = 0; l <= L; ++l) {
        uint64_t p4 = 1; for (int k = 0; k < L - l; ++k) p4 *= 4;
        if (p4 + b.level_cnt[l] > 0x01FFFFFFull) throw HipError("BVH too large");   // node byte offsets stay below 2^32
        b.level_off[l] = (uint32_t)p4;
    }
    const uint32_t off = b.level_off[0] + b.level_cnt[0];
    ctx->bvh_nodes.ensure(std::max<uint32_t>(off, 256u));   // 256: the LDS option stages a fixed prefix
    ctx->lvl_box_a.ensure(6 * (size_t)b.level_cnt[0]); ctx->lvl_box_b.ensure(6 * (size_t)std::max<uint32_t>(b.level_cnt[L > 0 ? 1 : 0], 1u));
    hipLaunchKernelGGL(build_level0_kernel, dim3((b.level_cnt[0] + 127) / 128), dim3(128), 0, s, ctx->bvh_tris.p, F, n_leaves, b.level_cnt[0], box, ctx->bvh_nodes.p + b.level_off[0], ctx->lvl_box_a.p);
    MVS_LAUNCH_CHECK();
    float* cur = ctx->lvl_box_a.p; float* nxt = ctx->lvl_box_b.p;
    for (int l = 1; l <= L; ++l) {
        hipLaunchKernelGGL(build_level_kernel, dim3((b.level_cnt[l] + 127) / 128), dim3(128), 0, s, cur, b.level_cnt[l - 1], b.level_cnt[l], ctx->bvh_nodes.p + b.level_off[l], nxt);
        MVS_LAUNCH_CHECK();
        std::swap(cur, nxt);
    }
    b.nodes = ctx->bvh_nodes.p; b.tris = ctx->bvh_tris.p;
    ctx->bvh = b;
    build_vertex_faces(ctx, ctx->ifc, F, NV);
}

void trace_rays(mvs_ctx* ctx) {
    const uint32_t vwords = (ctx->n_verts + 63) / 64;
    const uint64_t waves = (uint64_t)vwords * ctx->n_views;
    uint64_t blocks = (waves + 3) / 4;
    blocks = (blocks + 7) & ~7ull;   // multiple of 8 for the XCD-aware order (surplus waves exit)
    if (blocks > 0x7FFFFFFFull) throw HipError("ray grid too large");
    unsigned long long* counters = (ctx->stats || ctx->count_rays) ? ctx->counters.p : nullptr;
#define RAY_ARGS dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->bvh, ctx->iv, ctx->d_views.p, ctx->need_bits.p, ctx->occl_bits.p, \
                 vwords, ctx->n_verts, ctx->n_views, (const uint32_t*)ctx->scene_box.p, counters
    if (ctx->count_rays) { if (ctx->ray_xcd) hipLaunchKernelGGL((ray_packet3_kernel<true, true>), RAY_ARGS); else hipLaunchKernelGGL((ray_packet3_kernel<false, true>), RAY_ARGS); }
    else { if (ctx->ray_xcd) hipLaunchKernelGGL((ray_packet3_kernel<true, false>), RAY_ARGS); else hipLaunchKernelGGL((ray_packet3_kernel<false, false>), RAY_ARGS); }
#undef RAY_ARGS
    MVS_LAUNCH_CHECK();
}

}  // namespace mvs
