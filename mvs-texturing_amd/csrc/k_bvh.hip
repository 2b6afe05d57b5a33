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

// scene_box: 6 ordered-uint words (min xyz, max xyz), initialised to 0xFFFFFFFF x3, 0 x3
__global__ void bbox_kernel(const float* __restrict__ verts, uint32_t n_verts, uint32_t* __restrict__ box) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n_verts; v += gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) { const float x = verts[3 * (size_t)v + a]; lo[a] = fminf(lo[a], x); hi[a] = fmaxf(hi[a], x); }
    // same-address atomics serialise (~12 ns each): skip the ones that cannot move the bound any more
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
        if ((threadIdx.x & 63) == 0) {
            const uint32_t l = f2ord(lo[a]), h = f2ord(hi[a]);
            if (l < __hip_atomic_load(&box[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&box[a], l);
            if (h > __hip_atomic_load(&box[3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&box[3 + a], h);
        }
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

// Local refinement of the curve order.  A window of RW = 512 consecutive triangles is a whole number of subtrees of the
// implicit tree (16-triangle leaves: two level-1 nodes; 8-triangle leaves: one level-2 node).  Inside each window the triangles are re-partitioned top down: a segment is sorted
// along the longest axis of its centroids and cut in the middle (spatial median), recursively down to the leaves, so two
// binary levels = one 4-ary level.  The curve decides WHICH 512 triangles share a subtree, the median splits decide how
// they are grouped inside it: on the C3 mesh 26 % fewer leaf rounds and 34 % fewer (ray, leaf) pairs per packet
// (simulated; windows of 2048 would give 29 % / 38 %).  Deterministic: ties are broken by the triangle id.
constexpr int RW = 512;
__global__ void __launch_bounds__(256) refine_order_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, uint32_t* __restrict__ order,
                                                           uint32_t n_faces) {
    __shared__ float s_c[3][RW];          // centroid of the triangle in slot k (slots never move)
    __shared__ uint32_t s_id[RW];         // triangle id of slot k (0xFFFFFFFF: padding behind the last triangle)
    __shared__ float s_key[RW];           // sort key at position i
    __shared__ uint32_t s_slot[RW];       // slot at position i
    __shared__ uint32_t s_box[RW / (2 * LEAF_T)][6];   // centroid box per segment (ordered uints)
    const uint32_t w0 = blockIdx.x * RW;
    const int t = threadIdx.x;
    for (int i = t; i < RW; i += 256) {
        const uint32_t g = w0 + i;
        uint32_t id = 0xFFFFFFFFu; float c[3] = {INFINITY, INFINITY, INFINITY};
        if (g < n_faces) {
            id = order[g];
            const uint32_t* fv = faces + 3 * (size_t)id;
            for (int a = 0; a < 3; ++a) c[a] = (verts[3 * (size_t)fv[0] + a] + verts[3 * (size_t)fv[1] + a] + verts[3 * (size_t)fv[2] + a]) * (1.0f / 3.0f);
        }
        s_id[i] = id; s_c[0][i] = c[0]; s_c[1][i] = c[1]; s_c[2][i] = c[2]; s_slot[i] = (uint32_t)i;
    }
    __syncthreads();
    for (int seg = RW; seg > (int)LEAF_T; seg >>= 1) {
        const int nseg = RW / seg;
        for (int k = t; k < nseg * 6; k += 256) s_box[k / 6][k % 6] = (k % 6 < 3) ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        // centroid box of every segment: a wave holds 64 consecutive positions, i.e. whole segments or a 64-wide piece of
        // one -- butterfly min / max over min(seg, 64) lanes, then one lane per piece merges into LDS (same-address LDS
        // atomics from all 512 positions serialise 64-fold)
        for (int i = t; i < RW; i += 256) {
            const uint32_t sl = s_slot[i];
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
        for (int i = t; i < RW; i += 256) {
            const uint32_t* b = s_box[i / seg];
            int ax = 0;
            if (b[0] != 0xFFFFFFFFu) {   // segment holds at least one triangle
                const float e0 = ord2f(b[3]) - ord2f(b[0]), e1 = ord2f(b[4]) - ord2f(b[1]), e2 = ord2f(b[5]) - ord2f(b[2]);
                float best = e0;
                if (e1 > best) { best = e1; ax = 1; }
                if (e2 > best) { best = e2; ax = 2; }
            }
            s_key[i] = s_c[ax][s_slot[i]];
        }
        __syncthreads();
        // bitonic sort of every aligned segment of `seg` positions by (key, triangle id), ascending
        for (int k = 2; k <= seg; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int p = t;                                           // RW / 2 = 256 pairs, one per thread
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                const bool up = (k == seg) || ((i & k) == 0);
                const float ka = s_key[i], kb = s_key[l];
                const uint32_t sa = s_slot[i], sb = s_slot[l];
                const bool gt = ka > kb || (ka == kb && s_id[sa] > s_id[sb]);
                if (gt == up) { s_key[i] = kb; s_key[l] = ka; s_slot[i] = sb; s_slot[l] = sa; }
                // partners at distance j <= 64 live in the 128 positions this wave owns for all smaller j (LDS operations of
                // a wave execute in order): a block barrier is needed only while j > 64 or before the next k starts above 64
                if (j > 64 || (j == 1 && k >= 128)) __syncthreads();
            }
        __syncthreads();   // the next level (and the write-back) read positions other waves sorted
    }
    // padding keys are +inf with the largest id: they stay at the tail of every segment, so the triangles are a prefix
    for (int i = t; i < RW; i += 256) { const uint32_t g = w0 + i; if (g < n_faces) order[g] = s_id[s_slot[i]]; }
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
        const uint32_t* fv = faces + 3 * (size_t)order[s];
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

// ---- traversal ----
// Slab test of one child box: t = (bound - o) * inv per axis (subtract first: bound * inv - o * inv would cancel
// catastrophically for the nearby boxes that matter), interval [t0, t1] already widened by the caller.
__device__ __forceinline__ bool box_hit(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 inv, V3 o, float t0, float t1) {
    float ta = (lox - o.x) * inv.x, tb = (hix - o.x) * inv.x;
    float tn = fmaxf(t0, fminf(ta, tb)), tf = fminf(t1, fmaxf(ta, tb));   // fmin/fmax drop NaN (0 * inf)
    ta = (loy - o.y) * inv.y; tb = (hiy - o.y) * inv.y;
    tn = fmaxf(tn, fminf(ta, tb)); tf = fminf(tf, fmaxf(ta, tb));
    ta = (loz - o.z) * inv.z; tb = (hiz - o.z) * inv.z;
    tn = fmaxf(tn, fminf(ta, tb)); tf = fminf(tf, fmaxf(ta, tb));
    return tn <= tf;
}
__device__ __forceinline__ uint32_t node_hits(const Node4* __restrict__ nd, V3 o, V3 inv, float t0, float t1) {
    // 6 x 16-byte loads of one 128-byte line
    const float4* p = reinterpret_cast<const float4*>(nd);
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5];
    const uint32_t nchild = nd->nchild;
    const V3 oi = o;
    uint32_t m = 0;
    if (box_hit(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, inv, oi, t0, t1)) m |= 1u;
    if (box_hit(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, inv, oi, t0, t1)) m |= 2u;
    if (box_hit(q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, inv, oi, t0, t1)) m |= 4u;
    if (box_hit(q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, inv, oi, t0, t1)) m |= 8u;
    return m & ((1u << nchild) - 1u);
}

__device__ __forceinline__ bool tri_pre_hit(const float4* __restrict__ tris, uint32_t t, const Ray& r) {
    const float4* p = tris + TRI_F4 * (size_t)t;
    return tri_hit(p[0], p[1], p[2], p[3], r);
}

template <bool COUNT>
__device__ __forceinline__ bool any_hit(const BvhDev& bvh, const Ray& r, uint32_t& n_nodes, uint32_t& n_tris) {
    const V3 inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    const float t0 = r.tmin * 0.999f, t1 = r.tmax * 1.001f;
    int level = bvh.top;
    uint32_t node = 0;
    unsigned long long masks = (unsigned long long)node_hits(bvh.nodes + bvh.level_off[level], r.o, inv, t0, t1) << (4 * level);
    if (COUNT) n_nodes++;
    while (true) {
        const uint32_t m = (uint32_t)(masks >> (4 * level)) & 0xFu;
        if (m == 0) {
            if (level == bvh.top) return false;
            ++level; node >>= 2;
            continue;
        }
        const int c = __builtin_ctz(m);
        masks &= ~(1ull << (4 * level + c));
        const uint32_t child = node * 4 + c;
        if (level == 0) {
#pragma unroll
            for (uint32_t k = 0; k < LEAF_T; ++k) {
                if (COUNT) n_tris++;
                if (tri_pre_hit(bvh.tris, child * LEAF_T + k, r)) return true;
            }
        } else {
            --level; node = child;
            masks |= (unsigned long long)node_hits(bvh.nodes + bvh.level_off[level] + node, r.o, inv, t0, t1) << (4 * level);
            if (COUNT) n_nodes++;
        }
    }
}

// one wave per (view, 64-vertex word); lanes whose need bit is clear idle
template <bool COUNT>
__global__ void __launch_bounds__(256) ray_kernel(const BvhDev bvh, const float* __restrict__ verts, const uint32_t* __restrict__ vperm,
                                                  const ViewParams* __restrict__ views,
                                                  const unsigned long long* __restrict__ need, unsigned long long* __restrict__ occl,
                                                  uint32_t vwords, uint32_t n_verts, uint32_t n_views, const uint32_t* __restrict__ scene_box,
                                                  unsigned long long* __restrict__ counters) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (uint64_t)vwords * n_views) return;
    const uint32_t j = (uint32_t)(wave / vwords), vw = (uint32_t)(wave % vwords);   // view-major: needed patches form long runs (measured: patch-major is 40 % slower)
    const unsigned long long word = need[(size_t)j * vwords + vw];
    if (word == 0ull) return;  // occl is pre-zeroed
    const uint32_t sp = vw * 64 + lane;
    bool hit = false;
    uint32_t nn = 0, nt = 0;
    if (((word >> lane) & 1ull) && sp < n_verts) {
        const uint32_t v = vperm[sp];
        const ViewParams& vp = views[j];
        const V3 o = {verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2]};
        const Ray r = make_ray(o, V3{vp.pos[0], vp.pos[1], vp.pos[2]}, pad_from_box(scene_box));
        hit = any_hit<COUNT>(bvh, r, nn, nt);
    }
    const unsigned long long b = __ballot(hit);
    if (lane == 0) occl[(size_t)j * vwords + vw] = b;
    if (COUNT) {
        for (int o = 32; o > 0; o >>= 1) { nn += __shfl_xor(nn, o, 64); nt += __shfl_xor(nt, o, 64); }
        if (lane == 0) { atomicAdd(&counters[8], (unsigned long long)nn); atomicAdd(&counters[9], (unsigned long long)nt); }
    }
}

// ---- packet traversal: one wave = 64 nearly parallel rays sharing ONE traversal ----
// A node is visited when ANY still-active lane hits its box, so node and triangle addresses are
// wave-uniform (scalar / broadcast loads instead of 64 divergent gathers).  Lanes test every
// triangle of a visited leaf; testing a superset of triangles cannot change an any-hit result
// (the predicate is exact per triangle), so the booleans equal the per-ray traversal's.
__device__ __forceinline__ uint32_t node_hits_uniform(const Node4* __restrict__ nd, V3 o, V3 inv, float t0, float t1, bool active) {
    uint32_t m = 0;
    const uint32_t nchild = nd->nchild;
    const V3 oi = o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool h = box_hit(nd->lo(0, c), nd->lo(1, c), nd->lo(2, c), nd->hi(0, c), nd->hi(1, c), nd->hi(2, c), inv, oi, t0, t1);
        if (__ballot(active && h) != 0ull) m |= 1u << c;
    }
    return m & ((1u << nchild) - 1u);
}

template <bool COUNT>
__global__ void __launch_bounds__(256) ray_packet_kernel(const BvhDev bvh, const float* __restrict__ verts, const uint32_t* __restrict__ vperm,
                                                         const ViewParams* __restrict__ views, const unsigned long long* __restrict__ need,
                                                         unsigned long long* __restrict__ occl, uint32_t vwords, uint32_t n_verts, uint32_t n_views,
                                                         const uint32_t* __restrict__ scene_box, unsigned long long* __restrict__ counters) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (uint64_t)vwords * n_views) return;
    const uint32_t j = (uint32_t)(wave / vwords), vw = (uint32_t)(wave % vwords);   // view-major: needed patches form long runs (measured: patch-major is 40 % slower)
    const unsigned long long word = need[(size_t)j * vwords + vw];
    if (word == 0ull) return;  // occl is pre-zeroed
    const uint32_t s = vw * 64 + lane;
    bool active = ((word >> lane) & 1ull) && s < n_verts;
    const uint32_t v = vperm[s < n_verts ? s : 0];
    const ViewParams& vp = views[j];
    const V3 o = {verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2]};
    const Ray r = make_ray(o, V3{vp.pos[0], vp.pos[1], vp.pos[2]}, pad_from_box(scene_box));
    const V3 inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    const float t0 = r.tmin * 0.999f, t1 = r.tmax * 1.001f;
    bool hit = false;
    uint32_t nn = 0, nt = 0;
    int level = bvh.top;
    uint32_t node = 0;
    unsigned long long masks = (unsigned long long)node_hits_uniform(bvh.nodes + bvh.level_off[level], r.o, inv, t0, t1, active) << (4 * level);
    if (COUNT) nn++;
    while (true) {
        const uint32_t m = (uint32_t)(masks >> (4 * level)) & 0xFu;
        if (m == 0) {
            if (level == bvh.top) break;
            ++level; node >>= 2;
            continue;
        }
        const int c = __builtin_ctz(m);
        masks &= ~(1ull << (4 * level + c));
        const uint32_t child = node * 4 + c;   // wave-uniform
        if (level == 0) {
            const float4* __restrict__ tp = bvh.tris + TRI_F4 * (size_t)(child * LEAF_T);
#pragma unroll
            for (uint32_t k = 0; k < LEAF_T; ++k) {
                if (active && tri_hit(tp[TRI_F4 * k], tp[TRI_F4 * k + 1], tp[TRI_F4 * k + 2], tp[TRI_F4 * k + 3], r)) hit = true;
            }
            if (COUNT) nt += LEAF_T;
            active = active && !hit;
            if (__ballot(active) == 0ull) break;
        } else {
            --level; node = child;
            masks |= (unsigned long long)node_hits_uniform(bvh.nodes + bvh.level_off[level] + node, r.o, inv, t0, t1, active) << (4 * level);
            if (COUNT) nn++;
        }
    }
    const unsigned long long b = __ballot(hit);
    if (lane == 0) {
        occl[(size_t)j * vwords + vw] = b;
        if (COUNT) { atomicAdd(&counters[8], (unsigned long long)nn); atomicAdd(&counters[9], (unsigned long long)nt); }
    }
}

// ---- packet traversal with leaf work redistribution (ray_mode 2) ----
// Same shared traversal, but at a leaf only the lanes whose OWN ray hits the leaf box are candidates
// (5-6 of 64 on average), so instead of 64 lanes x 4 triangles the wave tests (candidate, triangle)
// PAIRS: lane L takes candidate L/4 and triangle L%4 (16 candidates per round).  Rays are parked in
// LDS once; the candidate list goes through LDS; results return to the owning lanes through a ballot.
// LDSN (mvs_set_option "lds_bvh_levels" > 0): the top levels of the tree (heap order: the first n_lds nodes) are staged in LDS by
// every block and read from there.  Measured A/B at BASELINE config 3 (profiles/r02_lds_bvh_ab.txt): NOT faster -- a node read
// from LDS lands in 32 VGPRs of every lane and is consumed as vector operands, whereas the default reads the node with two
// wide scalar loads into SGPRs (served by the scalar cache, which the top levels never leave) and feeds the fma's scalar
// operands.  Kept as an option for the record; the default is 0.
template <bool COUNT, bool XCD, bool LDSN>
__global__ void __launch_bounds__(256) ray_packet2_kernel(const BvhDev bvh, const float* __restrict__ verts, const uint32_t* __restrict__ vperm,
                                                          const ViewParams* __restrict__ views, const unsigned long long* __restrict__ need,
                                                          unsigned long long* __restrict__ occl, uint32_t vwords, uint32_t n_verts, uint32_t n_views,
                                                          const uint32_t* __restrict__ scene_box, unsigned long long* __restrict__ counters, uint32_t n_lds) {
    __shared__ float4 s_ray[4][64][2];
    __shared__ uint8_t s_src[4][64];
    extern __shared__ float4 s_top4[];
    if (LDSN) {
        for (uint32_t k = threadIdx.x; k < n_lds * 8u; k += 256u) s_top4[k] = reinterpret_cast<const float4*>(bvh.nodes)[k];
        __syncthreads();
    }
    const Node4* s_top = reinterpret_cast<const Node4*>(s_top4);
    // XCD-aware order: hardware block b runs on XCD b % 8 (observed; speed only), so each XCD gets a contiguous
    // eighth of the (view, vertex patch) sequence and with it a compact part of the BVH in its L2
    uint32_t vblk = blockIdx.x;
    if (XCD && (gridDim.x & 7u) == 0u) vblk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint64_t wave = ((uint64_t)vblk * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wave >= (uint64_t)vwords * n_views) return;
    const uint32_t j = (uint32_t)(wave / vwords), vw = (uint32_t)(wave % vwords);   // view-major: needed patches form long runs (measured: patch-major is 40 % slower)
    const unsigned long long word = need[(size_t)j * vwords + vw];
    if (word == 0ull) return;  // occl is pre-zeroed
    const uint32_t s = vw * 64 + lane;
    const bool active = ((word >> lane) & 1ull) && s < n_verts;
    const uint32_t v = vperm[s < n_verts ? s : 0];
    const ViewParams& vp = views[j];
    const V3 o = {verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2]};
    const float pad = pad_from_box(scene_box);
    const Ray r = make_ray(o, V3{vp.pos[0], vp.pos[1], vp.pos[2]}, pad);
    s_ray[wv][lane][0] = make_float4(r.o.x, r.o.y, r.o.z, r.tmin);
    s_ray[wv][lane][1] = make_float4(r.d.x, r.d.y, r.d.z, r.tmax);
    const V3 inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    const V3 oi = {r.o.x * inv.x, r.o.y * inv.y, r.o.z * inv.z};
    const float t0 = r.tmin * 0.999f, t1 = r.tmax * 1.001f;
    uint32_t nn = 0, nt = 0;
    int level = bvh.top;
    // The kernel is VALU bound (SQ_ACTIVE_INST_VALU = 98 % of the SIMD cycles at C3), so everything that is the same for
    // the whole wave lives in SGPRs: the set of live rays (actm) and of occluded rays (hitm) are 64-bit scalars, the four
    // per-child hit ballots of a node ARE the per-lane hit bits (bit L = lane L; hm0: those of the current level-0 node,
    // whose children are the leaves), a scalar mask becomes an execution mask again through inverse_ballot, ranks come
    // from v_mbcnt on the scalar mask.  A ballot of a bare float compare is the compare itself (it writes an SGPR pair).
    unsigned long long actm = __builtin_amdgcn_ballot_w64(active), hitm = 0ull;
    unsigned long long hm0[4] = {0ull, 0ull, 0ull, 0ull};
    auto visit = [&](uint32_t nidx, unsigned long long (&hm)[4]) -> uint32_t {
        uint32_t m = 0;
        Node4 ndv;
        if (LDSN && nidx < n_lds) ndv = s_top[nidx];   // wave-uniform branch
        else ndv = bvh.nodes[nidx];                    // the whole 128-byte line with two wide scalar loads
        const Node4* nd = &ndv;
        const uint32_t nchild = nd->nchild;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // t = bound * inv - o * inv as ONE fma (the product is exact inside it; oi is rounded once per ray): 12 instead of
            // 24 VALU operations per child.  Its absolute error, ulp(o * inv) <= 6e-8 |o| |inv|, is 400 times smaller than
            // the margin the boxes carry for exactly this purpose (a hit point lies >= 3 pad = 3e-5 max|coord| inside its leaf
            // box, i.e. 3 pad |inv| inside the slab), so the test stays conservative; inf - inf = NaN drops the axis (fmin/fmax).
            float ta = __builtin_fmaf(nd->lo(0, c), inv.x, -oi.x), tb = __builtin_fmaf(nd->hi(0, c), inv.x, -oi.x);
            float tn = fmaxf(t0, fminf(ta, tb)), tf = fminf(t1, fmaxf(ta, tb));
            ta = __builtin_fmaf(nd->lo(1, c), inv.y, -oi.y); tb = __builtin_fmaf(nd->hi(1, c), inv.y, -oi.y);
            tn = fmaxf(tn, fminf(ta, tb)); tf = fminf(tf, fmaxf(ta, tb));
            ta = __builtin_fmaf(nd->lo(2, c), inv.z, -oi.z); tb = __builtin_fmaf(nd->hi(2, c), inv.z, -oi.z);
            tn = fmaxf(tn, fminf(ta, tb)); tf = fminf(tf, fmaxf(ta, tb));
            hm[c] = __builtin_amdgcn_ballot_w64(tn <= tf) & actm;
            if (hm[c] != 0ull) m |= 1u << c;
        }
        return m & ((1u << nchild) - 1u);
    };
    unsigned long long hm_tmp[4];
    unsigned long long masks;
    const uint32_t off0 = bvh.level_off[0];   // heap index of the first level-0 node
    uint32_t node = 1;   // heap index, root = 1 (see build_bvh)
    if (level == 0) masks = (unsigned long long)visit(1u, hm0);
    else masks = (unsigned long long)visit(1u, hm_tmp) << (4 * level);
    if (COUNT) nn++;
    while (true) {
        const uint32_t m = (uint32_t)(masks >> (4 * level)) & 0xFu;
        if (m == 0) {
            if (level == bvh.top) break;
            ++level; node >>= 2;
            continue;
        }
        const int c = __builtin_ctz(m);
        masks &= ~(1ull << (4 * level + c));
        if (level == 0) {
            const uint32_t child = (node - off0) * 4 + c;   // leaf index, wave-uniform
            const unsigned long long hsel = (c == 0) ? hm0[0] : (c == 1) ? hm0[1] : (c == 2) ? hm0[2] : hm0[3];
            const unsigned long long cb = hsel & actm;          // rays that are still live and enter this leaf's box
            if (cb != 0ull) {
                const bool cand = __builtin_amdgcn_inverse_ballot_w64(cb);
                const int n = __builtin_popcountll(cb);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cb, 0u));
                if (cand) s_src[wv][rank] = (uint8_t)lane;
                const float4* __restrict__ tp = bvh.tris + TRI_F4 * (size_t)(child * LEAF_T + ((uint32_t)lane & (LEAF_T - 1u)));
                const float4 A = tp[0], E1 = tp[1], E2 = tp[2], H = tp[3];
                for (int base = 0; base < n; base += LEAF_SLOTS) {
                    const int q = base + lane / (int)LEAF_T;
                    const bool valid = q < n;
                    const int sl = valid ? (int)s_src[wv][q] : lane;
                    const float4 r0 = s_ray[wv][sl][0], r1 = s_ray[wv][sl][1];
                    Ray rr; rr.o = V3{r0.x, r0.y, r0.z}; rr.tmin = r0.w; rr.d = V3{r1.x, r1.y, r1.z}; rr.tmax = r1.w; rr.pad = pad;
                    const bool h = valid && tri_hit(A, E1, E2, H, rr);
                    const unsigned long long hb = __builtin_amdgcn_ballot_w64(h);
                    if (hb != 0ull) {   // wave-uniform and rare (hits are rare): the owner of slot (rank - base) reads its LEAF_T result bits
                        const bool mine = cand && rank >= base && rank < base + LEAF_SLOTS &&
                                          ((hb >> ((int)LEAF_T * (rank - base))) & ((1ull << LEAF_T) - 1ull)) != 0ull;
                        hitm |= __builtin_amdgcn_ballot_w64(mine);
                    }
                    if (COUNT) nt += 1;
                }
                actm &= ~hitm;
                if (actm == 0ull) break;
            }
        } else {
            --level; node = node * 4 + c;
            if (level == 0) masks |= (unsigned long long)visit(node, hm0);
            else masks |= (unsigned long long)visit(node, hm_tmp) << (4 * level);
            if (COUNT) nn++;
        }
    }
    if (lane == 0) {
        occl[(size_t)j * vwords + vw] = hitm;
        if (COUNT) { atomicAdd(&counters[8], (unsigned long long)nn); atomicAdd(&counters[9], (unsigned long long)nt); }
    }
}


// ---- packet traversal, round 3 (ray_mode 3, the default) ----
// The same shared traversal, leaf redistribution and predicate as ray_packet2_kernel -- identical booleans -- with the
// instruction count of a node visit cut by more than half (the kernel is issue bound on BOTH the vector and the scalar port):
//  * slab test with packed math: a child's six bounds are three aligned SGPR pairs, so the six t = bound * inv - o * inv
//    are three v_pk_fma_f32 (full rate with a scalar-pair operand on gfx950: scripts/probe/pk_probe.hip);
//  * direction-sign specialisation: all rays of a packet go from one surface patch to one camera, so nearly always every
//    live lane has the same sign of d on each axis.  Then near / far planes are known statically -- near_x is lo.x if
//    d.x > 0 else hi.x -- and min / max per axis disappear: tn = max(max3(near), t0), tf = min(min3(far), t1), 5 instead of
//    13 instructions per child after the fma's.  fma is monotone in the bound for a fixed finite multiplier, so
//    min(t_lo, t_hi) IS t_near bit for bit: the hit masks equal the generic test's.  Packets with mixed signs, zero or
//    tiny direction components (|1/d| >= 1e30 or non-finite) take the generic variant OCT = 8, which is the round-2 test;
//  * no per-level climbing: the pending-children nibbles of all levels sit in one 64-bit mask, s_ff1 finds the deepest
//    pending child, the ancestor's position in its level is a shift (heap order: node = (4^depth - 1) / 3 + position);
//  * a level-0 node's leaves are processed right after its visit, from the hit ballots still in registers -- nothing
//    per-node is carried around the loop (round 2 copied four 64-bit masks through every iteration);
//  * absent children are far-away points (build kernels), so the child-count mask is not needed.
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

template <int OCT>
__device__ __forceinline__ void packet3_traverse(const BvhDev& bvh, const float4 (*s_ray)[2], uint8_t* s_src, const int lane, const float pad,
                                                 const V3 inv, const V3 oi, const float t0, float t1,
                                                 unsigned long long& actm) {   // in: live rays; out: live rays never occluded
    constexpr bool SX = (OCT & 1) != 0, SY = (OCT & 2) != 0, SZ = (OCT & 4) != 0, GEN = OCT >= 8;
    const f2 I0 = {inv.x, inv.y}, I1 = {inv.z, inv.x}, I2 = {inv.y, inv.z};
    const f2 O0 = {-oi.x, -oi.y}, O1 = {-oi.z, -oi.x}, O2 = {-oi.y, -oi.z};
    const Node4* __restrict__ nodes = bvh.nodes;
    // A lane that is not live (never needed, or already occluded) gets the empty interval t1 = -1 < 0 <= t0: it enters no
    // box, so the hit ballots need no masking.
    t1 = __builtin_amdgcn_inverse_ballot_w64(actm) ? t1 : -1.0f;
    unsigned long long hm0 = 0ull, hm1 = 0ull, hm2 = 0ull, hm3 = 0ull;   // separate scalars, never an indexed array (that would live in scratch)
    auto visit = [&](uint32_t h) -> uint32_t {   // h wave-uniform: the 96 bytes of bounds arrive with two wide scalar loads
        const Node4 nd = *reinterpret_cast<const Node4*>(reinterpret_cast<const char*>(nodes) + (h << 7));   // 32-bit byte offset (build_bvh bounds h): base + offset addressing
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
    do {
        const uint32_t idx = (uint32_t)__builtin_ctzll(masks);        // deepest level first = depth first
        masks = clear_bit(masks, idx);
        const uint32_t lv = idx >> 2;
        const uint32_t ch = ((h >> (2u * (lv - level))) << 2) | (idx & 3u);   // the ancestor on level lv, then its child
        m = visit(ch);
        if (lv != 1u) { masks |= (unsigned long long)m << (4u * (lv - 1u)); h = ch; level = lv - 1u; }
        else { h = ch >> 2; level = 1u; leaves(ch, m); masks = (actm == 0ull) ? 0ull : masks; }
    } while (masks != 0ull);
}

// amdgpu_num_sgpr(80): a 256-thread block is admitted per CU up to floor(800 / (ceil16(sgpr) + 16)) times (MI355X_MICROARCH.md
// "Residency"): the 98 SGPRs the compiler takes by itself allow 6 blocks, 80 allow 7 (the VGPRs then stop at 7 waves per SIMD).  The
// kernel is a chain of dependent node and triangle fetches per wave -- PMC: vector issue 41 %, waves parked on memory 46 % of their
// time -- so residency is throughput: 6.63 -> 6.33 ms at BASELINE config 3 (A/B of two builds on one box, scripts/ab_libs.py).
template <bool XCD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) ray_packet3_kernel(const BvhDev bvh, const float* __restrict__ verts, const uint32_t* __restrict__ vperm,
                                                          const ViewParams* __restrict__ views, const unsigned long long* __restrict__ need,
                                                          unsigned long long* __restrict__ occl, uint32_t vwords, uint32_t n_verts, uint32_t n_views,
                                                          const uint32_t* __restrict__ scene_box, unsigned long long* __restrict__ counters) {
    __shared__ float4 s_ray[4][64][2];
    __shared__ uint8_t s_src[4][80];   // candidate lists; a round reads up to LEAF_SLOTS - 1 slots past the list
    uint32_t vblk = blockIdx.x;   // XCD-aware order, see ray_packet2_kernel
    if (XCD && (gridDim.x & 7u) == 0u) vblk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint64_t wave = ((uint64_t)vblk * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wave >= (uint64_t)vwords * n_views) return;
    const uint32_t j = (uint32_t)(wave / vwords), vw = (uint32_t)(wave % vwords);
    const unsigned long long word = need[(size_t)j * vwords + vw];
    if (word == 0ull) return;  // occl is pre-zeroed
    const uint32_t s = vw * 64 + lane;
    const bool active = ((word >> lane) & 1ull) && s < n_verts;
    const uint32_t v = vperm[s < n_verts ? s : 0];
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
#define MVS_P3(O) packet3_traverse<O>(bvh, s_ray[wv], s_src[wv], lane, pad, inv, oi, t0, t1, live)
    switch (oct) {
        case 0: MVS_P3(0); break; case 1: MVS_P3(1); break; case 2: MVS_P3(2); break; case 3: MVS_P3(3); break;
        case 4: MVS_P3(4); break; case 5: MVS_P3(5); break; case 6: MVS_P3(6); break; case 7: MVS_P3(7); break;
        default: MVS_P3(8); break;
    }
#undef MVS_P3
    if (lane == 0) occl[(size_t)j * vwords + vw] = actm & ~live;
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

// Builds the BVH and the vertex->face incidence for the resident mesh.
void build_bvh(mvs_ctx* ctx) {
    const uint32_t F = ctx->n_faces, NV = ctx->n_verts;
    hipStream_t s = ctx->stream;
    const uint32_t n_leaves = (F + LEAF_T - 1) / LEAF_T;
    const uint32_t n_slots = ((n_leaves + 3) / 4) * 4 * LEAF_T;  // triangles padded to whole level-0 nodes
    ctx->scene_box.ensure(8);
    uint32_t* box = (uint32_t*)ctx->scene_box.p;
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    MVS_HIP(hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(bbox_kernel, dim3(std::min<uint32_t>((NV + 255) / 256, 512u)), dim3(256), 0, s, ctx->d_verts, NV, box);
    MVS_LAUNCH_CHECK();
    ctx->sort_k.ensure(std::max<size_t>(F, NV)); ctx->sort_k2.ensure(std::max<size_t>(F, NV)); ctx->sort_v.ensure(std::max<size_t>(F, NV)); ctx->sort_v2.ensure(F);
    hipLaunchKernelGGL(curve_key_kernel, dim3((F + 255) / 256), dim3(256), 0, s, ctx->d_verts, ctx->d_faces, F, box, ctx->sort_k.p, ctx->sort_v.p);
    MVS_LAUNCH_CHECK();
    size_t tmp_bytes = 0;
    MVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ctx->sort_k.p, ctx->sort_k2.p, ctx->sort_v.p, ctx->sort_v2.p, F, 0, 30, s));
    ctx->sort_tmp.ensure(tmp_bytes + 16);
    MVS_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->sort_k.p, ctx->sort_k2.p, ctx->sort_v.p, ctx->sort_v2.p, F, 0, 30, s));
    static_assert(RW == 2 * 256 && RW % (int)LEAF_T == 0, "one pair per thread");
    hipLaunchKernelGGL(refine_order_kernel, dim3((F + RW - 1) / RW), dim3(256), 0, s, ctx->d_verts, ctx->d_faces, ctx->sort_v2.p, F);
    MVS_LAUNCH_CHECK();
    ctx->bvh_tris.ensure(TRI_F4 * (size_t)n_slots);
    hipLaunchKernelGGL(gather_tris_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, s, ctx->d_verts, ctx->d_faces, ctx->sort_v2.p, F, n_slots, box, ctx->bvh_tris.p);
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
    // vertices in Hilbert order (ray launch order)
    ctx->vperm.ensure((size_t)NV + 1); ctx->vpos.ensure((size_t)NV + 1);
    ctx->sort_k.ensure(std::max<size_t>(F, NV)); ctx->sort_k2.ensure(std::max<size_t>(F, NV)); ctx->sort_v.ensure(std::max<size_t>(F, NV));
    hipLaunchKernelGGL(vertex_key_kernel, dim3((NV + 255) / 256), dim3(256), 0, s, ctx->d_verts, NV, box, ctx->sort_k.p, ctx->sort_v.p);
    MVS_LAUNCH_CHECK();
    MVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ctx->sort_k.p, ctx->sort_k2.p, ctx->sort_v.p, ctx->vperm.p, NV, 0, 30, s));
    ctx->sort_tmp.ensure(tmp_bytes + 16);
    MVS_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->sort_k.p, ctx->sort_k2.p, ctx->sort_v.p, ctx->vperm.p, NV, 0, 30, s));
    hipLaunchKernelGGL(invert_perm_kernel, dim3((NV + 255) / 256), dim3(256), 0, s, ctx->vperm.p, NV, ctx->vpos.p);
    MVS_LAUNCH_CHECK();
    build_vertex_faces(ctx, ctx->d_faces, F, NV);
}

void trace_rays(mvs_ctx* ctx) {
    const uint32_t vwords = (ctx->n_verts + 63) / 64;
    const uint64_t waves = (uint64_t)vwords * ctx->n_views;
    uint64_t blocks = (waves + 3) / 4;
    if (ctx->ray_mode >= 2) blocks = (blocks + 7) & ~7ull;   // multiple of 8 for the XCD-aware order (surplus waves exit)
    if (blocks > 0x7FFFFFFFull) throw HipError("ray grid too large");
#define RAY_ARGS dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->bvh, ctx->d_verts, ctx->vperm.p, ctx->d_views.p, ctx->need_bits.p, ctx->occl_bits.p, \
                 vwords, ctx->n_verts, ctx->n_views, (const uint32_t*)ctx->scene_box.p, ctx->counters.p
    if (ctx->ray_mode == 3 && !ctx->count_rays && ctx->lds_bvh_levels <= 0) {
        blocks = (blocks + 7) & ~7ull;
        if (ctx->ray_xcd) hipLaunchKernelGGL(ray_packet3_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->bvh, ctx->d_verts, ctx->vperm.p, ctx->d_views.p,
                                             ctx->need_bits.p, ctx->occl_bits.p, vwords, ctx->n_verts, ctx->n_views, (const uint32_t*)ctx->scene_box.p, ctx->stats ? ctx->counters.p : nullptr);
        else hipLaunchKernelGGL(ray_packet3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->bvh, ctx->d_verts, ctx->vperm.p, ctx->d_views.p,
                                ctx->need_bits.p, ctx->occl_bits.p, vwords, ctx->n_verts, ctx->n_views, (const uint32_t*)ctx->scene_box.p, ctx->stats ? ctx->counters.p : nullptr);
    } else if (ctx->ray_mode >= 2) {
        // top levels in LDS (option, default off): levels counted from the root; at most what 48 KB hold, never the leaves' parents' level and below
        uint32_t n_lds = 0;
        if (ctx->lds_bvh_levels > 0) {
            const int lv = std::min(std::min(ctx->lds_bvh_levels, 4), (int)ctx->bvh.top);
            uint32_t w = 1; for (int l = 1; l < lv; ++l) w *= 4;
            n_lds = 2 * w;   // heap indices below 2 * 4^(lv - 1) hold the lv levels from the root (root = index 1)
        }
        const size_t lds = (size_t)n_lds * sizeof(Node4);
        if (ctx->count_rays) hipLaunchKernelGGL((ray_packet2_kernel<true, false, false>), RAY_ARGS, 0u);
        else if (n_lds) hipLaunchKernelGGL((ray_packet2_kernel<false, true, true>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, ctx->bvh, ctx->d_verts, ctx->vperm.p, ctx->d_views.p,
                                           ctx->need_bits.p, ctx->occl_bits.p, vwords, ctx->n_verts, ctx->n_views, (const uint32_t*)ctx->scene_box.p, ctx->counters.p, n_lds);
        else if (ctx->ray_xcd) hipLaunchKernelGGL((ray_packet2_kernel<false, true, false>), RAY_ARGS, 0u);
        else hipLaunchKernelGGL((ray_packet2_kernel<false, false, false>), RAY_ARGS, 0u);
    } else if (ctx->ray_mode == 1) {
        if (ctx->count_rays) hipLaunchKernelGGL(ray_packet_kernel<true>, RAY_ARGS); else hipLaunchKernelGGL(ray_packet_kernel<false>, RAY_ARGS);
    } else {
        if (ctx->count_rays) hipLaunchKernelGGL(ray_kernel<true>, RAY_ARGS); else hipLaunchKernelGGL(ray_kernel<false>, RAY_ARGS);
    }
#undef RAY_ARGS
    MVS_LAUNCH_CHECK();
}

}  // namespace mvs
