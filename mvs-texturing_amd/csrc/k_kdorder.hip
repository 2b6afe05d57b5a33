// k_kdorder.hip -- the UPPER levels of the face order behind the implicit BVH (k_bvh.hip): exact top-down median cuts on the device.
//
// The implicit 4-ary tree has no freedom but the ORDER of its triangles: a leaf is 16 consecutive triangles, a level-k node 4 consecutive
// level-(k-1) nodes.  Rounds 1 - 5 ordered them along a Hilbert curve and re-partitioned only inside windows of 512 (refine_order_kernel).
// Round 6 measured what the order is worth (scripts/bvh_order_probe.py, profiles/r06_bvh_order_probe_c3.json; BASELINE config 3, ray stage):
//     Hilbert + cuts inside 512-face windows                 39.9 node visits / 61.7 leaf rounds per packet   6.28 ms
//     ... inside 4 096 / 16 384 / 65 536 / 262 144 faces     35.4 / 32.6 / 30.7 / 28.6 visits                  5.50 / 5.19 / 5.14 / 4.90 ms
//     the whole tree cut top-down                            26.4 visits / 43.6 rounds                         4.74 ms
// (cuts that do not sit on the implicit tree's own child boundaries -- cells from a sample tree, whole Hilbert runs moved as units -- lose
//  nearly all of it: 6.2 / 5.4 ms.)  So the cuts have to be exact, by rank, at every level.
//
// One binary level for ALL nodes of a window size at once (the nodes are aligned ranges of the current order, a block's 4096 positions lie
// in one node): the node's k = cap / 2 smallest centroids along the longest axis of its centroid box go to its lower half.
//   * the k-th smallest key by a three-pass radix select (11 + 11 + 10 bits of the order-preserving uint of the float coordinate): every
//     block adds its elements to the node's histogram (LDS-private, flushed with atomics), the next pass finds the pivot bin itself;
//   * scatter: smaller keys below, larger above (block-aggregated atomic cursors: WHICH elements go below is exact and deterministic, their
//     order inside a half is not -- the next level selects by rank again and the last level, refine_order_kernel in LDS, sorts: the final
//     order is deterministic); keys EQUAL to the pivot are ranked by triangle id (a short per-node list, sorted by one block): ties never
//     make the order depend on scheduling.  The scatter also folds the children's centroid boxes (the next level's axes);
//   * a node filled to at most half its capacity (the mesh's tail) is not cut: it moves down one level unchanged.
// Five launches per level (three histogram passes, the scatter, ties + the children's axes), each streaming the window once; levels from the top window (262 144 faces of the Hilbert order, or the whole
// mesh with option "bvh_window" = 0) down to 2 x the LDS window of refine_order_kernel.
#include "ctx.h"

namespace mvs {

namespace {

constexpr uint32_t KD_T = 1024, KD_E = 4, KD_B = KD_T * KD_E;       // threads, elements per thread, positions per block
constexpr uint32_t KD_BINS = 2048, KD_TIE_CAP = 2048;

__device__ __forceinline__ uint32_t kd_f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float kd_ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// a thread's running centroid box (ordered uints), folded into a node's box ONCE per block: wave butterflies, the waves' results through LDS,
// then at most six atomics -- and only those that can still move a bound (same-address atomics serialise: one per wave and element cost
// the scatter 0.2 ms per level)
struct KdBox {
    uint32_t lo[3], hi[3];
    __device__ __forceinline__ void clear() { for (int a = 0; a < 3; ++a) { lo[a] = 0xFFFFFFFFu; hi[a] = 0u; } }
    __device__ __forceinline__ void add(bool on, float x, float y, float z) {
        if (!on) return;
        const float c[3] = {x, y, z};
        for (int a = 0; a < 3; ++a) { const uint32_t o = kd_f2ord(c[a]); lo[a] = min(lo[a], o); hi[a] = max(hi[a], o); }
    }
};
// s_red: 16 waves x 6 words; every thread of the block calls.  The block's box is STORED at dst (six words of its own): nothing is folded
// with atomics -- same-address traffic from hundreds of blocks serialises in one L2 channel (the first version of this file spent 0.17 ms
// of its centroid kernel and 25 of 27 us of every scatter there); kd_axis_kernel folds the block boxes of a node.
__device__ __forceinline__ void kd_box_store(KdBox bx, uint32_t* __restrict__ dst, uint32_t* s_red) {
    for (int a = 0; a < 3; ++a) for (int o = 32; o > 0; o >>= 1) { bx.lo[a] = min(bx.lo[a], (uint32_t)__shfl_xor(bx.lo[a], o, 64)); bx.hi[a] = max(bx.hi[a], (uint32_t)__shfl_xor(bx.hi[a], o, 64)); }
    const uint32_t wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; ++a) { s_red[6 * wave + a] = bx.lo[a]; s_red[6 * wave + 3 + a] = bx.hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const bool is_lo = threadIdx.x < 3;
        uint32_t v = is_lo ? 0xFFFFFFFFu : 0u;
        for (uint32_t w = 0; w < n_waves; ++w) { const uint32_t x = s_red[6 * w + threadIdx.x]; v = is_lo ? min(v, x) : max(v, x); }
        dst[threadIdx.x] = v;
    }
}
// the block's box in every thread's registers (res[0 .. 2] lower, res[3 .. 5] upper bounds)
__device__ __forceinline__ void kd_box_reduce(KdBox bx, uint32_t* s_red, uint32_t res[6]) {
    for (int a = 0; a < 3; ++a) for (int o = 32; o > 0; o >>= 1) { bx.lo[a] = min(bx.lo[a], (uint32_t)__shfl_xor(bx.lo[a], o, 64)); bx.hi[a] = max(bx.hi[a], (uint32_t)__shfl_xor(bx.hi[a], o, 64)); }
    const uint32_t wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; ++a) { s_red[6 * wave + a] = bx.lo[a]; s_red[6 * wave + 3 + a] = bx.hi[a]; }
    __syncthreads();
    for (int a = 0; a < 6; ++a) {
        uint32_t v = a < 3 ? 0xFFFFFFFFu : 0u;
        for (uint32_t w = 0; w < n_waves; ++w) { const uint32_t x = s_red[6 * w + a]; v = a < 3 ? min(v, x) : max(v, x); }
        res[a] = v;
    }
}
struct KdArrays { float* c[3]; uint32_t* id; };
struct KdPivot { uint32_t below, take, equal, ustar; };   // #{u < u*}, how many of the `equal` keys == u* go below (1 .. equal), u*

__device__ __forceinline__ int kd_axis(const uint32_t* __restrict__ b) {
    if (b[0] == 0xFFFFFFFFu) return 0;
    const float e0 = kd_ord2f(b[3]) - kd_ord2f(b[0]), e1 = kd_ord2f(b[4]) - kd_ord2f(b[1]), e2 = kd_ord2f(b[5]) - kd_ord2f(b[2]);
    int ax = 0; float best = e0;
    if (e1 > best) { best = e1; ax = 1; }
    if (e2 > best) { best = e2; ax = 2; }
    return ax;
}

// the bin of a 2048-bin histogram that holds the element of rank `k` (0-based), and the rank inside that bin: all threads of the block
// get the answer.  s_scan: KD_T + 32 words of scratch.
__device__ uint2 kd_find_bin(const uint32_t* __restrict__ hist, uint32_t k, uint32_t* s_scan) {
    const uint32_t t = threadIdx.x;
    const uint32_t a = hist[2 * t], b = hist[2 * t + 1];
    uint32_t v = a + b;
    // inclusive scan over the block's 1024 partial sums
    for (int o = 1; o < 64; o <<= 1) { const uint32_t n = __shfl_up(v, o, 64); if ((int)(t & 63) >= o) v += n; }
    if ((t & 63) == 63) s_scan[KD_T + (t >> 6)] = v;
    __syncthreads();
    if (t < 16) { uint32_t w = s_scan[KD_T + t]; for (int o = 1; o < 16; o <<= 1) { const uint32_t n = __shfl_up(w, o, 16); if ((int)t >= o) w += n; } s_scan[KD_T + 16 + t] = w; }
    __syncthreads();
    const uint32_t incl = v + ((t >> 6) ? s_scan[KD_T + 16 + (t >> 6) - 1] : 0u), excl = incl - (a + b);
    if (k >= excl && k < incl) { const bool first = k < excl + a; s_scan[0] = first ? 2 * t : 2 * t + 1; s_scan[1] = first ? k - excl : k - excl - a; }
    __syncthreads();
    const uint2 r = make_uint2(s_scan[0], s_scan[1]);
    __syncthreads();
    return r;
}

struct KdLevel {
    uint32_t F, cap;                 // node capacity of this level (a node = positions [j cap, (j + 1) cap) of the current order)
    const uint32_t* axis;            // [node] cut axis (kd_axis_kernel)
    uint32_t* hist;                  // [node][3][KD_BINS]
};

// PASS 0 / 1 / 2: histogram of the top 11 / middle 11 / low 10 bits of the keys (of those that fell into the pivot bins so far)
template <int PASS>
__global__ void __launch_bounds__(KD_T) kd_hist_kernel(KdLevel L, KdArrays in) {
    __shared__ uint32_t s_h[KD_BINS];
    __shared__ uint32_t s_scan[KD_T + 32];
    const uint32_t base = blockIdx.x * KD_B, j = base / L.cap, start = j * L.cap;
    if (start >= L.F) return;
    const uint32_t n = min(L.cap, L.F - start), k = L.cap / 2;
    if (n <= k) return;                                       // not cut at this level
    uint32_t* hist = L.hist + (size_t)j * 3 * KD_BINS;
    uint32_t prefix = 0;                                      // the key bits fixed by the earlier passes
    if (PASS >= 1) { const uint2 r = kd_find_bin(hist, k - 1, s_scan); prefix = r.x; if (PASS == 2) { const uint2 r2 = kd_find_bin(hist + KD_BINS, r.y, s_scan); prefix = (prefix << 11) | r2.x; } }
    for (uint32_t i = threadIdx.x; i < KD_BINS; i += KD_T) s_h[i] = 0u;
    __syncthreads();
    const float* __restrict__ key = in.c[L.axis[j]];
#pragma unroll
    for (uint32_t e = 0; e < KD_E; ++e) {
        const uint32_t p = base + threadIdx.x + e * KD_T;
        if (p < start + n) {
            const uint32_t u = kd_f2ord(key[p]);
            if (PASS == 0) atomicAdd(&s_h[u >> 21], 1u);
            else if (PASS == 1) { if ((u >> 21) == prefix) atomicAdd(&s_h[(u >> 10) & 2047u], 1u); }
            else { if ((u >> 10) == prefix) atomicAdd(&s_h[u & 1023u], 1u); }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < KD_BINS; i += KD_T) if (s_h[i]) atomicAdd(&hist[PASS * KD_BINS + i], s_h[i]);
}

// centroids of the triangles in the given order + one centroid box per block (side 0 of the block-box array)
__global__ void __launch_bounds__(KD_T) kd_centroid_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, const uint32_t* __restrict__ order, uint32_t F,
                                                           KdArrays out, uint32_t* __restrict__ bbox /* [block][2][6] */) {
    __shared__ uint32_t s_red[6 * 16];
    KdBox bx, none; bx.clear(); none.clear();
#pragma unroll
    for (uint32_t e = 0; e < KD_E; ++e) {
        const uint32_t p = blockIdx.x * KD_B + threadIdx.x + e * KD_T;
        if (p < F) {
            const uint32_t f = order[p];
            const uint32_t* fv = faces + 3 * (size_t)f;
            float c[3];
            for (int a = 0; a < 3; ++a) { c[a] = (verts[3 * (size_t)fv[0] + a] + verts[3 * (size_t)fv[1] + a] + verts[3 * (size_t)fv[2] + a]) * (1.0f / 3.0f); out.c[a][p] = c[a]; }
            out.id[p] = f;
            bx.add(true, c[0], c[1], c[2]);
        }
    }
    kd_box_store(bx, bbox + 12 * (size_t)blockIdx.x, s_red);
    kd_box_store(none, bbox + 12 * (size_t)blockIdx.x + 6, s_red);
}

// the cut axes of the TOP level's nodes = the longest axis of the fold of the block boxes kd_centroid_kernel left (side 0); one wave per node
// (the levels below get theirs from kd_tie_axis_kernel of the level above)
__global__ void __launch_bounds__(64) kd_axis_kernel(uint32_t F, uint32_t cap, const uint32_t* __restrict__ bbox, uint32_t n_nodes, uint32_t* __restrict__ axis) {
    const uint32_t j = blockIdx.x;
    if (j >= n_nodes) return;
    const uint32_t start = j * cap, n = min(cap, F - start);
    const uint32_t b0 = start / KD_B, nb = (n + KD_B - 1) / KD_B;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t k = threadIdx.x; k < nb; k += 64) {
        const uint32_t* b = bbox + 12 * (size_t)(b0 + k);
        for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], b[a]); hi[a] = max(hi[a], b[3 + a]); }
    }
    for (int a = 0; a < 3; ++a) for (int o = 32; o > 0; o >>= 1) { lo[a] = min(lo[a], (uint32_t)__shfl_xor(lo[a], o, 64)); hi[a] = max(hi[a], (uint32_t)__shfl_xor(hi[a], o, 64)); }
    if (threadIdx.x == 0) { const uint32_t b[6] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]}; axis[j] = (uint32_t)kd_axis(b); }
}

struct KdScatter {
    KdLevel L;
    uint32_t* cursor;                // [node][4]: next free slot below / above (non-tie elements), ties listed, overflow flag (node 0 only)
    uint32_t* tie;                   // [node][KD_TIE_CAP] source positions of the keys equal to the pivot (when they are not all taken)
    KdPivot* pivot;                  // [node]
    uint32_t* bbox;                  // [block][2][6]: centroid boxes of what this block sent below / above (folded by kd_tie_axis_kernel)
};

__global__ void __launch_bounds__(KD_T) kd_scatter_kernel(KdScatter S, KdArrays in, KdArrays out) {
    __shared__ uint32_t s_scan[KD_T + 32];
    __shared__ uint32_t s_cnt[2][16], s_base[2], s_red[6 * 16];
    const KdLevel& L = S.L;
    const uint32_t base = blockIdx.x * KD_B, j = base / L.cap, start = j * L.cap;
    if (start >= L.F) return;
    const uint32_t n = min(L.cap, L.F - start), k = L.cap / 2;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (n <= k) {   // not cut: the node moves down one level as it is (it is the LOWER child of its position: node 2 j of the next level)
        KdBox bx; bx.clear();
#pragma unroll
        for (uint32_t e = 0; e < KD_E; ++e) {
            const uint32_t p = base + t + e * KD_T;
            const bool on = p < start + n;
            float x = 0.f, y = 0.f, z = 0.f;
            if (on) { x = in.c[0][p]; y = in.c[1][p]; z = in.c[2][p]; out.c[0][p] = x; out.c[1][p] = y; out.c[2][p] = z; out.id[p] = in.id[p]; }
            bx.add(on, x, y, z);
        }
        KdBox none; none.clear();
        kd_box_store(bx, S.bbox + 12 * (size_t)blockIdx.x, s_red);
        kd_box_store(none, S.bbox + 12 * (size_t)blockIdx.x + 6, s_red);
        return;
    }
    const uint32_t* hist = L.hist + (size_t)j * 3 * KD_BINS;
    const uint2 r0 = kd_find_bin(hist, k - 1, s_scan);
    const uint2 r1 = kd_find_bin(hist + KD_BINS, r0.y, s_scan);
    const uint2 r2 = kd_find_bin(hist + 2 * KD_BINS, r1.y, s_scan);
    const uint32_t ustar = (r0.x << 21) | (r1.x << 10) | r2.x;
    const uint32_t equal = hist[2 * KD_BINS + r2.x], take = r2.y + 1u, below = (k - 1u) - r2.y;     // rank of the pivot among its equals is r2.y
    const bool ties = take < equal;                          // some, not all, of the equal keys go below: ranked by id in kd_tie_kernel
    if (base == start && t == 0) { KdPivot pv; pv.below = below; pv.take = take; pv.equal = equal; pv.ustar = ustar; S.pivot[j] = pv; }
    const int ax = (int)L.axis[j];
    uint32_t* cur = S.cursor + 4 * (size_t)j;
    // where the two halves start for NON-tie elements: below [start, ...), above [start + k + (equal - take), ...) when ties are listed
    const uint32_t lo0 = start, hi0 = start + k + (ties ? equal - take : 0u);
    KdBox bx_lo, bx_hi; bx_lo.clear(); bx_hi.clear();
#pragma unroll
    for (uint32_t e = 0; e < KD_E; ++e) {
        const uint32_t p = base + t + e * KD_T;
        const bool on = p < start + n;
        float c[3] = {0.f, 0.f, 0.f}; uint32_t id = 0u, u = 0u;
        if (on) { c[0] = in.c[0][p]; c[1] = in.c[1][p]; c[2] = in.c[2][p]; id = in.id[p]; u = kd_f2ord(c[ax]); }
        const bool is_tie = on && ties && u == ustar;
        const bool lo = on && !is_tie && (u < ustar || (u == ustar));     // (u == ustar without ties: all equal keys are taken)
        const bool hi = on && !is_tie && u > ustar;
        const unsigned long long ml = __ballot(lo), mh = __ballot(hi);
        if (lane == 0) { s_cnt[0][wave] = (uint32_t)__popcll(ml); s_cnt[1][wave] = (uint32_t)__popcll(mh); }
        __syncthreads();
        if (t < 2) {
            uint32_t tot = 0; for (int w = 0; w < 16; ++w) { const uint32_t v = s_cnt[t][w]; s_cnt[t][w] = tot; tot += v; }
            s_base[t] = tot ? atomicAdd(&cur[t], tot) : 0u;
        }
        __syncthreads();
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (lo || hi) {
            const uint32_t dst = lo ? lo0 + s_base[0] + s_cnt[0][wave] + (uint32_t)__popcll(ml & lt) : hi0 + s_base[1] + s_cnt[1][wave] + (uint32_t)__popcll(mh & lt);
            out.c[0][dst] = c[0]; out.c[1][dst] = c[1]; out.c[2][dst] = c[2]; out.id[dst] = id;
        }
        if (is_tie) { const uint32_t s = atomicAdd(&cur[2], 1u); if (s < KD_TIE_CAP) S.tie[(size_t)j * KD_TIE_CAP + s] = p; else S.cursor[3] = 1u; }
        bx_lo.add(lo, c[0], c[1], c[2]); bx_hi.add(hi, c[0], c[1], c[2]);
        __syncthreads();
    }
    kd_box_store(bx_lo, S.bbox + 12 * (size_t)blockIdx.x, s_red);
    kd_box_store(bx_hi, S.bbox + 12 * (size_t)blockIdx.x + 6, s_red);
}

// One block per node, after its scatter: (a) the keys equal to the node's pivot, when only some of them go below, are ranked by id (bitonic
// sort in LDS) and placed; (b) the centroid boxes of the node's two halves -- the block boxes its scatter left and the ties' -- are folded
// into the cut AXES of its children at the next level (no launch of their own, no same-address atomics).
__global__ void __launch_bounds__(KD_T) kd_tie_axis_kernel(KdScatter S, KdArrays in, KdArrays out, uint32_t* __restrict__ axis_next, uint32_t n_next) {
    __shared__ uint32_t s_id[KD_TIE_CAP], s_p[KD_TIE_CAP];
    __shared__ uint32_t s_red[6 * 16];
    const KdLevel& L = S.L;
    const uint32_t j = blockIdx.x, start = j * L.cap;
    if (start >= L.F) return;
    const uint32_t n = min(L.cap, L.F - start), k = L.cap / 2;
    KdBox bx_lo, bx_hi; bx_lo.clear(); bx_hi.clear();
    const uint32_t m = n <= k ? 0u : min(S.cursor[4 * (size_t)j + 2], KD_TIE_CAP);
    if (m) {   // (block-uniform)
        const KdPivot pv = S.pivot[j];
        for (uint32_t i = threadIdx.x; i < KD_TIE_CAP; i += KD_T) {
            const bool on = i < m;
            const uint32_t p = on ? S.tie[(size_t)j * KD_TIE_CAP + i] : 0u;
            s_p[i] = p; s_id[i] = on ? in.id[p] : 0xFFFFFFFFu;
        }
        __syncthreads();
        for (uint32_t kk = 2; kk <= KD_TIE_CAP; kk <<= 1)
            for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
                const uint32_t q = threadIdx.x, i = ((q & ~(jj - 1)) << 1) | (q & (jj - 1)), l = i | jj;
                const bool up = (i & kk) == 0;
                const uint32_t a = s_id[i], b = s_id[l];
                if ((a > b) == up) { s_id[i] = b; s_id[l] = a; const uint32_t pa = s_p[i]; s_p[i] = s_p[l]; s_p[l] = pa; }
                __syncthreads();
            }
        for (uint32_t i = threadIdx.x; i < KD_TIE_CAP; i += KD_T) {
            if (i >= m) continue;
            const bool lo = i < pv.take;
            const uint32_t p = s_p[i];
            const uint32_t dst = lo ? start + pv.below + i : start + k + (i - pv.take);
            const float c0 = in.c[0][p], c1 = in.c[1][p], c2 = in.c[2][p];
            out.c[0][dst] = c0; out.c[1][dst] = c1; out.c[2][dst] = c2; out.id[dst] = s_id[i];
            bx_lo.add(lo, c0, c1, c2); bx_hi.add(!lo, c0, c1, c2);
        }
    }
    if (!axis_next) return;
    // the children's boxes: this node's blocks (below = side 0, above = side 1) + the ties placed above
    const uint32_t b0 = start / KD_B, nb = (n + KD_B - 1) / KD_B;
    for (uint32_t q = threadIdx.x; q < nb; q += KD_T) {
        const uint32_t* b = S.bbox + 12 * (size_t)(b0 + q);
        for (int a = 0; a < 3; ++a) { bx_lo.lo[a] = min(bx_lo.lo[a], b[a]); bx_lo.hi[a] = max(bx_lo.hi[a], b[3 + a]); bx_hi.lo[a] = min(bx_hi.lo[a], b[6 + a]); bx_hi.hi[a] = max(bx_hi.hi[a], b[9 + a]); }
    }
    uint32_t r_lo[6], r_hi[6];
    kd_box_reduce(bx_lo, s_red, r_lo);
    kd_box_reduce(bx_hi, s_red, r_hi);
    if (threadIdx.x == 0) {
        if (2 * j < n_next) axis_next[2 * j] = (uint32_t)kd_axis(r_lo);
        if (2 * j + 1 < n_next) axis_next[2 * j + 1] = (uint32_t)kd_axis(r_hi);
    }
}

}  // namespace

// Re-partitions `order` (triangle ids, F entries, e.g. the Hilbert order) top-down inside aligned windows of `window` positions (0: the whole
// mesh), level by level down to node capacity `leaf_window` (a power of two >= 4096: what the LDS pass of k_bvh.hip takes over from).
// Nothing here waits for the device: the new order is written in any case, and one word per level -- "a node had more than KD_TIE_CAP keys
// equal to its pivot" (a degenerate mesh: elements were dropped, the order is NOT a permutation) -- travels to pinned host memory behind
// it.  The caller looks at those words at its next synchronisation (k_bvh.hip scene_order_commit) and falls back to the order without
// upper levels.
void kd_refine_order(mvs_ctx* ctx, const float* verts, const uint32_t* faces, uint32_t* order, uint32_t F, uint32_t window, uint32_t leaf_window) {
    hipStream_t s = ctx->stream;
    if (F <= leaf_window) return;
    uint64_t cap_top = leaf_window;
    while (cap_top < F && (window == 0 || cap_top < window)) cap_top *= 2;
    if (cap_top <= leaf_window) return;
    int levels = 0; for (uint64_t c = cap_top; c > leaf_window; c /= 2) ++levels;
    // per level: nodes, histograms, cursors, ties, pivots, axes, tie boxes; block boxes are per block (rewritten by every level)
    std::vector<size_t> node_off(levels + 2, 0);
    { uint64_t c = cap_top; for (int l = 0; l <= levels; ++l, c /= 2) node_off[l + 1] = node_off[l] + (size_t)((F + c - 1) / c); }
    const size_t n_nodes = node_off[levels];
    const unsigned blocks = (F + KD_B - 1) / KD_B;
    ctx->kd_hist.ensure(n_nodes * 3 * KD_BINS + 4); ctx->kd_cursor.ensure(n_nodes * 4 + 4); ctx->kd_tie.ensure(n_nodes * KD_TIE_CAP + 4);
    ctx->kd_pivot.ensure(n_nodes * 4 + 4); ctx->kd_box.ensure(12 * (size_t)blocks + node_off[levels + 1] + 16);
    for (int b = 0; b < 2; ++b) { for (int a = 0; a < 3; ++a) ctx->kd_c[b][a].ensure((size_t)F + 4); ctx->kd_id[b].ensure((size_t)F + 4); }
    uint32_t* bbox = ctx->kd_box.p; uint32_t* axis = bbox + 12 * (size_t)blocks;
    MVS_HIP(hipMemsetAsync(ctx->kd_hist.p, 0, n_nodes * 3 * KD_BINS * sizeof(uint32_t), s));
    MVS_HIP(hipMemsetAsync(ctx->kd_cursor.p, 0, (n_nodes * 4 + 4) * sizeof(uint32_t), s));
    // (should a tie list overflow, its elements are never placed: the holes must at least hold VALID ids -- the order is used by the kernels
    //  queued behind this function before scene_order_commit notices and rebuilds it)
    MVS_HIP(hipMemsetAsync(ctx->kd_id[1].p, 0, (size_t)F * sizeof(uint32_t), s));
    KdArrays A[2];
    for (int b = 0; b < 2; ++b) { for (int a = 0; a < 3; ++a) A[b].c[a] = ctx->kd_c[b][a].p; A[b].id = ctx->kd_id[b].p; }
    hipLaunchKernelGGL(kd_centroid_kernel, dim3(blocks), dim3(KD_T), 0, s, verts, faces, (const uint32_t*)order, F, A[0], bbox); MVS_LAUNCH_CHECK();
    int cur = 0;
    uint64_t cap = cap_top;
    for (int l = 0; l < levels; ++l, cap /= 2) {
        const uint32_t nn = (uint32_t)(node_off[l + 1] - node_off[l]);
        const uint32_t cap32 = (uint32_t)std::min<uint64_t>(cap, 0x80000000ull);
        if (l == 0) { hipLaunchKernelGGL(kd_axis_kernel, dim3(nn), dim3(64), 0, s, F, cap32, (const uint32_t*)bbox, nn, axis); MVS_LAUNCH_CHECK(); }
        KdScatter S;
        S.L.F = F; S.L.cap = cap32;
        S.L.axis = axis + node_off[l]; S.L.hist = ctx->kd_hist.p + node_off[l] * 3 * KD_BINS;
        S.cursor = ctx->kd_cursor.p + 4 * node_off[l]; S.tie = ctx->kd_tie.p + node_off[l] * KD_TIE_CAP;
        S.pivot = reinterpret_cast<KdPivot*>(ctx->kd_pivot.p) + node_off[l]; S.bbox = bbox;
        hipLaunchKernelGGL(kd_hist_kernel<0>, dim3(blocks), dim3(KD_T), 0, s, S.L, A[cur]); MVS_LAUNCH_CHECK();
        hipLaunchKernelGGL(kd_hist_kernel<1>, dim3(blocks), dim3(KD_T), 0, s, S.L, A[cur]); MVS_LAUNCH_CHECK();
        hipLaunchKernelGGL(kd_hist_kernel<2>, dim3(blocks), dim3(KD_T), 0, s, S.L, A[cur]); MVS_LAUNCH_CHECK();
        hipLaunchKernelGGL(kd_scatter_kernel, dim3(blocks), dim3(KD_T), 0, s, S, A[cur], A[cur ^ 1]); MVS_LAUNCH_CHECK();
        hipLaunchKernelGGL(kd_tie_axis_kernel, dim3(nn), dim3(KD_T), 0, s, S, A[cur], A[cur ^ 1], l + 1 < levels ? axis + node_off[l + 1] : (uint32_t*)nullptr,
                           (uint32_t)(node_off[l + 2] - node_off[l + 1])); MVS_LAUNCH_CHECK();
        cur ^= 1;
    }
    // more equal keys at a pivot than the tie list holds (flag of any level = word 3 of that level's first node): one word per level, to pinned memory
    if (!ctx->h_kd_flags) MVS_HIP(hipHostMalloc((void**)&ctx->h_kd_flags, 64 * sizeof(uint32_t), hipHostMallocDefault));
    for (int l = 0; l < levels && l < 64; ++l) { ctx->h_kd_flags[l] = 0u; MVS_HIP(hipMemcpyAsync(&ctx->h_kd_flags[l], ctx->kd_cursor.p + 4 * node_off[l] + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, s)); }
    ctx->kd_pending = std::min(levels, 64);
    MVS_HIP(hipMemcpyAsync(order, A[cur].id, (size_t)F * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
}

}  // namespace mvs
