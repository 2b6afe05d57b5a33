// k_dc.hip -- tex::calculate_data_costs on the GPU
// (libs/tex/calculate_data_costs.cpp:131-306).
//
// The reference walks view-by-view over all faces (:148-229) and scatters the
// survivors into per-face vectors under a critical section (:241-249).  Here:
//   cull_kernel       one thread per (face, view): culls of :171-191, 64 faces x 1 view per wave,
//                     result = one ballot word in a view-major bit matrix (no atomics);
//   need_kernel       which (vertex, view) rays are needed (OR over incident faces);
//   [k_bvh.hip]       each distinct ray once -> occluded bits;
//   info_kernel       get_face_info (:220) for the visible pairs, qualities written at the
//                     pair's rank among the pass bits (coalesced);
//   count/write       transposition view-major -> CSR by face (ascending view id = the
//                     std::sort of :272);
//   outlier_kernel    photometric_outlier_detection (:35-129) per face, fp64 in registers;
//   max / histogram / percentile / cost  = postprocess_face_infos (:278-302).
#include "ctx.h"

namespace mvs {

void prepare_views(mvs_ctx* ctx, bool need_gmi, const size_t* d_gmi_off, const size_t* d_mask_off);
void build_scene_order(mvs_ctx* ctx);
bool scene_order_commit(mvs_ctx* ctx);
void build_bvh(mvs_ctx* ctx);
void trace_rays(mvs_ctx* ctx);

namespace {

constexpr int VIEW_CHUNK = 32;
constexpr uint32_t HIST_BINS = 10000;  // calculate_data_costs.cpp:283

enum { C_BACK = 0, C_ANGLE = 1, C_OUTSIDE = 2, C_OCCL = 3, C_ZEROQ = 4, C_SURV = 5, C_RAYS = 6, C_PASS = 7, C_RNODES = 8, C_RTRIS = 9, C_REWALK = 12 };

__device__ __forceinline__ V3 ld3(const float* __restrict__ p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Diagnostic counters (cull reasons etc.): same-address atomics serialise at ~12 ns each, so they are
// optional (mvs_set_option "stats") and reduced per 256-thread block before touching memory.
template <int N>
__device__ __forceinline__ void block_count_add(const uint32_t (&c)[N], unsigned long long* __restrict__ counters, const int (&slot)[N]) {
    __shared__ uint32_t acc[N];
    if (threadIdx.x < N) acc[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) { const uint32_t w = wave_sum(c[k]); if ((threadIdx.x & 63) == 0 && w) atomicAdd(&acc[k], w); }
    __syncthreads();
    if (threadIdx.x < N && acc[threadIdx.x]) atomicAdd(&counters[slot[threadIdx.x]], (unsigned long long)acc[threadIdx.x]);
}

// ---- culls (calculate_data_costs.cpp:171-191) ----
// A block holds 256 faces in registers (vertices, normal, centre: gathered ONCE) and walks ALL views, 32 at a time: the face-major copy
// of a chunk's pass bits is one word per face.  (Until round 4 the grid had a second dimension over the view chunks, and every chunk's
// block gathered the same vertices again: 7 times at 200 views -- 0.5 GB of the kernel's 0.69 GB of HBM traffic at config 3.)
template <bool STATS>
__global__ void __launch_bounds__(256) cull_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, const float* __restrict__ normals,
                                                   const ViewParams* __restrict__ views, uint32_t n_views, uint32_t fb, uint32_t nf, uint32_t fwords,
                                                   float cos_limit, unsigned long long* __restrict__ pass, uint32_t* __restrict__ pass_face /* may be null */,
                                                   unsigned long long* __restrict__ counters) {
    const uint32_t lf = blockIdx.x * 256 + threadIdx.x;
    const bool wave_ok = (lf >> 6) < fwords;  // false: whole wave beyond the face range
    const bool act = lf < nf;
    const size_t f = (size_t)fb + (act ? lf : 0);
    const uint32_t i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const V3 v1 = ld3(verts, i0), v2 = ld3(verts, i1), v3 = ld3(verts, i2), nrm = ld3(normals, f);
    const int lane = threadIdx.x & 63;
    uint32_t cnt[4] = {0, 0, 0, 0};
    // the clear cases of the first two culls are decided without normalisations: dmath.h cull_pair_prefiltered
    const V3 centre = ((v1 + v2) + v3) / 3.0f;
    if (wave_ok) {
        // (blockIdx.y splits the chunks only when there are too few face blocks to fill the chip: a small mesh under many views)
        const uint32_t n_chunks = (n_views + VIEW_CHUNK - 1) / VIEW_CHUNK, per = (n_chunks + gridDim.y - 1) / gridDim.y;
        const uint32_t c_begin = blockIdx.y * per, c_end = min(c_begin + per, n_chunks);
        for (uint32_t chunk = c_begin, j0 = c_begin * VIEW_CHUNK; chunk < c_end; j0 += VIEW_CHUNK, ++chunk) {
            const uint32_t j1 = min(j0 + VIEW_CHUNK, n_views);
            uint32_t mine = 0;   // this face's pass bits for the views of the chunk (face-major copy for need_kernel: one load instead of 32)
            for (uint32_t j = j0; j < j1; ++j) {
                const ViewParams& vw = views[j];
                const int reason = act ? cull_pair_prefiltered(vw, v1, v2, v3, nrm, centre, cos_limit) : -1;
                if (STATS) { cnt[0] += reason == 1; cnt[1] += reason == 2; cnt[2] += reason == 3; cnt[3] += reason == 0; }
                const unsigned long long b = __ballot(reason == 0);
                if (lane == 0) pass[(size_t)j * fwords + (lf >> 6)] = b;
                mine |= (reason == 0 ? 1u : 0u) << (j - j0);
            }
            if (pass_face) pass_face[(size_t)chunk * ((size_t)fwords * 64u) + lf] = mine;
        }
    }
    if (STATS) { const int slot[4] = {C_BACK, C_ANGLE, C_OUTSIDE, C_PASS}; block_count_add<4>(cnt, counters, slot); }
}

// ---- which (vertex, view) rays are needed: OR of the pass bits of the incident faces ----
// (thread s handles vertex s of the library's copy of the mesh = the s-th vertex of the Hilbert curve: need / occluded bits are indexed by it)
template <bool STATS>
__global__ void __launch_bounds__(256) need_kernel(const uint32_t* __restrict__ vf_ptr, const uint32_t* __restrict__ vf, uint32_t n_verts, uint32_t n_views,
                                                   uint32_t fb, uint32_t nf, uint32_t fwords, uint32_t vwords,
                                                   const uint32_t* __restrict__ pass_face /* [view chunk][face]: the chunk's pass bits of a face (cull_kernel) */,
                                                   unsigned long long* __restrict__ need, unsigned long long* __restrict__ counters) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if ((v >> 6) >= vwords) return;  // whole wave beyond the vertex range
    const bool act = v < n_verts;
    const uint32_t vid = act ? v : 0;
    const uint32_t p0 = act ? vf_ptr[vid] : 0, p1 = act ? vf_ptr[vid + 1] : 0;
    const uint32_t j0 = blockIdx.y * VIEW_CHUNK, j1 = min(j0 + VIEW_CHUNK, n_views);
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;  // bit jj: some incident face passed for view j0 + jj
    for (uint32_t pb = p0; pb < p1; pb += 4) {   // four incident faces at a time: ids, then their pass words (independent loads per level)
        uint32_t lf[4], w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) lf[t] = (pb + t < p1) ? vf[pb + t] - fb : 0xFFFFFFFFu;  // wraps for faces below the range
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = (lf[t] < nf) ? pass_face[(size_t)blockIdx.y * ((size_t)fwords * 64u) + lf[t]] : 0u;
        acc |= (w[0] | w[1]) | (w[2] | w[3]);
    }
    uint32_t n_rays = 0;
    for (uint32_t j = j0; j < j1; ++j) {
        const unsigned long long b = __ballot((acc >> (j - j0)) & 1u);
        if (lane == 0) { need[(size_t)j * vwords + (v >> 6)] = b; n_rays += __popcll(b); }
    }
    if (STATS && lane == 0 && n_rays) atomicAdd(&counters[C_RAYS], (unsigned long long)n_rays);
}

__global__ void popc_kernel(const unsigned long long* __restrict__ words, uint32_t* __restrict__ cnt, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = (uint32_t)__popcll(words[i]);
}

// ---- get_face_info for the visible pairs (calculate_data_costs.cpp:194-228) ----
// WORDS (gradient term without outlier removal, lane-group sampler enabled): a footprint's pixels are read four per load and added as
// integers (dmath.h foot_walk_gmi_words) and the result is used iff foot_sums_certified proves it equal to the serial fp64 walk's;
// what it cannot certify (~ n 2^-28 of the footprints) and footprints with a degenerate edge go the way of the large ones (deferred).
template <int DATA_TERM, bool OUTLIER, bool VISTEST, bool STATS, bool WORDS>
__global__ void __launch_bounds__(256, WORDS ? 6 : 1) info_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, const ViewParams* __restrict__ views,
                                                   uint32_t n_views, uint32_t fb, uint32_t nf, uint32_t fwords, uint32_t vwords,
                                                   const unsigned long long* __restrict__ pass, const unsigned long long* __restrict__ occl,
                                                   const uint32_t* __restrict__ pass_base, float* __restrict__ pq, float* __restrict__ pcol,
                                                   unsigned long long* __restrict__ surv, unsigned long long* __restrict__ counters,
                                                   float defer_area /* footprints above this area are left to wave_info_kernel (+inf: none) */,
                                                   unsigned long long* __restrict__ defer_bits, int cert_shift) {
    static_assert(!WORDS || (DATA_TERM == 1 && !OUTLIER), "the word walk sums gradient magnitudes only");
    // the serial walk adds (double)u8 / 255.0 per pixel and channel: the 256 possible quotients, correctly rounded by the same
    // division, are looked up in LDS instead of divided (an fp64 division is ~15 double-precision instructions)
    __shared__ double s_q255[WORDS ? 1 : 256];
    if (!WORDS) {
        s_q255[threadIdx.x] = (double)threadIdx.x / 255.0;
        __syncthreads();
    }
    const uint32_t lf = blockIdx.x * 256 + threadIdx.x;
    const bool wave_ok = (lf >> 6) < fwords;  // false: whole wave beyond the face range
    const bool act = lf < nf;
    const size_t f = (size_t)fb + (act ? lf : 0);
    const uint32_t i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const V3 v1 = ld3(verts, i0), v2 = ld3(verts, i1), v3 = ld3(verts, i2);
    const uint32_t s0 = i0, s1 = i1, s2 = i2;  // bit positions of the 3 rays: the vertices of the library's copy are numbered along the curve the rays are launched in
    const uint32_t j0 = blockIdx.y * VIEW_CHUNK, j1 = min(j0 + VIEW_CHUNK, n_views);
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t cnt[3] = {0, 0, 0};  // occluded, zero quality, survivors
    // the chunk's pass words and their ranks: lane l < 32 fetches those of view j0 + l -- one round trip instead of one per view; the
    // views with an empty word (most of them: a wave's 64 faces see a fraction of the views) are finished here, by their lanes
    static_assert(VIEW_CHUNK <= 64, "one lane per view of the chunk");
    unsigned long long my_word = 0ull; uint32_t my_base = 0u;
    if (wave_ok && j0 + (uint32_t)lane < j1 && lane < VIEW_CHUNK) {
        const size_t widx = (size_t)(j0 + lane) * fwords + (lf >> 6);
        my_word = pass[widx]; my_base = pass_base[widx];
        if (my_word == 0ull) { surv[widx] = 0ull; if (defer_bits) defer_bits[widx] = 0ull; }
    }
    unsigned long long todo = __ballot(my_word != 0ull);
    while (todo) {
        const int jl = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t j = j0 + (uint32_t)jl;
        const size_t widx = (size_t)j * fwords + (lf >> 6);
        const unsigned long long word = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_word >> 32), jl) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_word, jl);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)my_base, jl);
        bool keep = false, deferred = false;
        if ((word >> lane) & 1ull) {
            bool visible = true;
            if (VISTEST) {
                const unsigned long long* o = occl + (size_t)j * vwords;
                visible = !(((o[s0 >> 6] >> (s0 & 63)) | (o[s1 >> 6] >> (s1 & 63)) | (o[s2 >> 6] >> (s2 & 63))) & 1ull);
            }
            FaceInfoOut fi; fi.quality = 0.0f; fi.mean_color[0] = fi.mean_color[1] = fi.mean_color[2] = 0.0f;
            if (visible) {
                if (WORDS) {
                    // face_info's control flow (dmath.h) around the integer walk
                    const ViewParams& view = views[j];
                    FootSetup fs;
                    foot_setup(view, v1, v2, v3, fs);
                    if (fs.area < FLT_EPSILON) {
                    } else if (fs.area > 0.5f) {
                        if (fs.area > defer_area) fi.quality = FOOT_DEFERRED;
                        else {
                            foot_edges(fs);
                            uint32_t n = 0, g = 0;
                            if (fs.fast) foot_walk_gmi_words(view, fs, &n, &g);
                            const double CG = (double)g / 255.0;
                            if (fs.fast && foot_sums_certified<1, false>(fs, n, 0.0, 0.0, 0.0, CG, cert_shift)) foot_finish<1, false>(view, fs, n, 0.0, 0.0, 0.0, CG, &fi);
                            else fi.quality = FOOT_DEFERRED;
                        }
                    } else foot_finish<1, false>(view, fs, 0u, 0.0, 0.0, 0.0, 0.0, &fi);
                } else face_info<DATA_TERM, OUTLIER>(views[j], v1, v2, v3, &fi, defer_area, s_q255);
                if (fi.quality == FOOT_DEFERRED) deferred = true;   // quality, survivor bit and counters come from wave_info_kernel
                else if (fi.quality == 0.0f) ++cnt[1]; else { keep = true; ++cnt[2]; }
            } else ++cnt[0];
            const size_t r = (size_t)base + __popcll(word & lt);
            pq[r] = fi.quality;
            if (OUTLIER) {
                rgb_to_ycbcr(fi.mean_color);  // :225
                pcol[3 * r] = fi.mean_color[0]; pcol[3 * r + 1] = fi.mean_color[1]; pcol[3 * r + 2] = fi.mean_color[2];
            }
        }
        const unsigned long long b = __ballot(keep);
        const unsigned long long db = defer_bits ? __ballot(deferred) : 0ull;
        if (lane == 0) { surv[widx] = b; if (defer_bits) defer_bits[widx] = db; }
    }
    if (STATS) { const int slot[3] = {C_OCCL, C_ZEROQ, C_SURV}; block_count_add<3>(cnt, counters, slot); }
}

// ---- large footprints: a lane GROUP per (face, view) pair ----
// The reference's rasteriser is a serial scan-line loop (texture_view.cpp:183-219): as one thread per pair a footprint of
// thousands of pixels keeps one lane busy while its 63 neighbours idle, and real captures are made of such footprints.  Here a
// footprint is walked by 16 lanes (2 scan lines x 8 pixels at a time; the spans come from the shared foot_row()), its pixels
// are read along rows and summed as INTEGERS (u8 values: exact, order independent); only the final division by 255 is fp64.  The
// reference adds the quotients u / 255.0 one by one in fp64, so its sum differs from this one by a few fp64 roundings
// (~n * 2^-53 relative).  The result is USED only under a certificate (dmath.h foot_sums_certified) that the conversion to float
// cannot tell the two apart -- it then IS the reference's value, bit for bit; the ~n * 2^-28 of the footprints the certificate
// cannot decide are re-walked in the reference's serial order (rewalk_info_kernel).  Footprints up to `defer_area` pixels take
// info_kernel (mvs_set_option "info_wave_area"; 0 = every footprint in the serial walk).
__global__ void defer_expand_kernel(const unsigned long long* __restrict__ defer_bits, const uint32_t* __restrict__ base, size_t n_words, uint2* __restrict__ list) {
    const size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    unsigned long long w = defer_bits[wi];
    uint32_t o = base[wi];
    while (w) { const int b = __ffsll((long long)w) - 1; w &= w - 1ull; list[o++] = make_uint2((uint32_t)wi, (uint32_t)(wi >> 32) << 8 | (uint32_t)b); }
}
template <int DATA_TERM, bool OUTLIER, bool STATS>
__global__ void __launch_bounds__(256) wave_info_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, const ViewParams* __restrict__ views,
                                                        uint32_t fb, uint32_t fwords, const uint2* __restrict__ list, uint32_t n_list,
                                                        const unsigned long long* __restrict__ pass, const uint32_t* __restrict__ pass_base,
                                                        float* __restrict__ pq, float* __restrict__ pcol, unsigned long long* __restrict__ surv,
                                                        unsigned long long* __restrict__ counters, int cert_shift, uint32_t* __restrict__ rewalk) {
    // 16 lanes per footprint (four footprints per wave), arranged as 2 scan lines x 8 pixels (words of four pixels for the gradient
    // term): the spans come from the shared foot_row(), the lanes stride through them -- no staging, no search, coalesced row reads
    constexpr int GL = 16, ROWS = 2, COLS = 8;
    const uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) / GL;
    const bool act = k < n_list;
    const int sub = threadIdx.x & (GL - 1), dy = sub / COLS, dx = sub % COLS;
    const uint2 rec = list[act ? k : 0];
    const size_t widx = (size_t)rec.x | ((size_t)(rec.y >> 8) << 32);
    const uint32_t bit = rec.y & 63u;
    const uint32_t j = (uint32_t)(widx / fwords), lf = (uint32_t)(widx % fwords) * 64u + bit;
    const size_t f = (size_t)fb + lf;
    const V3 v1 = ld3(verts, faces[3 * f]), v2 = ld3(verts, faces[3 * f + 1]), v3 = ld3(verts, faces[3 * f + 2]);
    const ViewParams& view = views[j];
    FootSetup s;
    foot_setup(view, v1, v2, v3, s);
    foot_edges(s);
    const int w = view.width;
    const uint8_t* image = view.rgb; const uint8_t* gimg = view.gmi;
    uint32_t n = 0, c0 = 0, c1 = 0, c2 = 0, g = 0;             // per-lane integer sums: <= 255 * (pixels / 16) each
    const int y_begin = (int)floorf(s.aabb_min_y), y_end = act ? (int)ceilf(s.aabb_max_y) : y_begin;   // (float)y < ceilf(max): y < an integer-valued float
    // Blocks of 16 scan lines: lane r of the group evaluates the span of line yb + r ONCE (foot_row: two correctly rounded divisions),
    // the steps below fetch their line's span from that lane -- one pair of divisions per line instead of one per lane and line
    const bool words = DATA_TERM == 1 && !OUTLIER && s.fast;   // group-uniform
#ifndef MVS_NARROW_PX
#define MVS_NARROW_PX 96.0f
#endif
    const bool narrow = s.aabb_max_x - s.aabb_min_x <= MVS_NARROW_PX;   // group-uniform
    for (int yb = y_begin; yb < y_end; yb += GL) {
        int xb_own = 0, xe_own = 0;
        if (yb + sub >= y_end || !foot_row(s, yb + sub, &xb_own, &xe_own) || xe_own <= xb_own) { xb_own = 0; xe_own = 0; }   // an empty span: the line is skipped
        if (words && narrow) {
            // a NARROW footprint (spans of a few words): lane r of the group sums line yb + r by itself, four words per step -- sixteen lines in
            // flight per footprint where the 2 x 8 arrangement below keeps four lines in flight with most of their eight lanes past the span
            const uint8_t* rowp = gimg + (size_t)(yb + sub) * w;
            int x0 = xb_own - (int)(reinterpret_cast<uintptr_t>(rowp + xb_own) & 3u);
            if (xe_own <= xb_own) x0 = xe_own;
            while (x0 < xe_own) {   // four words (16 pixels) per step: the loads of a step are independent of each other
                uint32_t v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (x0 + 4 * q < xe_own) ? *reinterpret_cast<const uint32_t*>(rowp + x0 + 4 * q) : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xw = x0 + 4 * q;
                    if (xw < xe_own) {
                        const int lo = max(xb_own - xw, 0), hi = min(xe_own - xw, 4);          // bytes [lo, hi) of the word are pixels of the span
                        const uint32_t m = (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & ~((1u << (8 * lo)) - 1u);
                        g = __builtin_amdgcn_sad_u8(v[q] & m, 0u, g); n += (uint32_t)(hi - lo);
                    }
                }
                x0 += 16;
            }
            continue;
        }
        if (words) {
            // gradient magnitudes only, whole spans: four pixels per load.  The words are aligned in ADDRESS space (the first one starts
            // at or up to three bytes before the span), bytes outside [xb, xe) are masked, v_sad_u8 adds the four bytes of a word in one
            // instruction.  Integer sums of the same pixels: identical result.  (gmi points into the context's own padded buffer: an
            // aligned word around a valid pixel is always inside it.)  A step covers 2 x 2 lines (this lane: lines r0 + dy and
            // r0 + 2 + dy), so that two loads are in flight per lane.
            for (int r0 = 0; r0 < GL && yb + r0 < y_end; r0 += 2 * ROWS) {
                int xb[2], xe[2], x0[2];
                const uint8_t* rowp[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = r0 + ROWS * q + dy;                  // < 16: the lane that holds the line's span
                    xb[q] = __shfl(xb_own, r, GL); xe[q] = __shfl(xe_own, r, GL);
                    rowp[q] = gimg + (size_t)(yb + r) * w;
                    x0[q] = xb[q] - (int)(reinterpret_cast<uintptr_t>(rowp[q] + xb[q]) & 3u) + 4 * dx;
                    if (xe[q] <= xb[q]) x0[q] = xe[q];              // an empty span (a skipped line, a line past y_end): nothing is fetched
                }
                while (x0[0] < xe[0] || x0[1] < xe[1]) {
                    uint32_t v[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) v[q] = (x0[q] < xe[q]) ? *reinterpret_cast<const uint32_t*>(rowp[q] + x0[q]) : 0u;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (x0[q] < xe[q]) {
                            const int lo = max(xb[q] - x0[q], 0), hi = min(xe[q] - x0[q], 4);      // bytes [lo, hi) of the word are pixels of the span
                            const uint32_t m = (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & ~((1u << (8 * lo)) - 1u);
                            g = __builtin_amdgcn_sad_u8(v[q] & m, 0u, g);
                            n += (uint32_t)(hi - lo);
                        }
                        x0[q] += 4 * COLS;
                    }
                }
            }
            continue;
        }
        for (int r0 = 0; r0 < GL && yb + r0 < y_end; r0 += ROWS) {
            const int y = yb + r0 + dy;
            const int xb = __shfl(xb_own, r0 + dy, GL), xe = __shfl(xe_own, r0 + dy, GL);
            for (int x = xb + dx; x < xe; x += COLS) {
                if (!s.fast && !foot_inside(s, x, y)) continue;
                const size_t pix = (size_t)x + (size_t)y * w;
                if (OUTLIER) { c0 += image[pix * 3 + 0]; c1 += image[pix * 3 + 1]; c2 += image[pix * 3 + 2]; }
                if (DATA_TERM == 1) g += gimg[pix];
                ++n;
            }
        }
    }
    for (int o = GL / 2; o > 0; o >>= 1) {
        n += __shfl_xor(n, o, GL); g += __shfl_xor(g, o, GL);
        if (OUTLIER) { c0 += __shfl_xor(c0, o, GL); c1 += __shfl_xor(c1, o, GL); c2 += __shfl_xor(c2, o, GL); }
    }
    uint32_t cnt[2] = {0, 0};   // zero quality, survivors
    if (sub == 0 && act) {
        FaceInfoOut fi; fi.quality = 0.0f; fi.mean_color[0] = fi.mean_color[1] = fi.mean_color[2] = 0.0f;
        // integer sums, one division each -- and a certificate that the result equals the reference's serial fp64 walk bit for bit
        // (dmath.h foot_sums_certified); the footprints it cannot decide (~ n 2^-28 of them) go to rewalk_info_kernel
        const double C0 = (double)c0 / 255.0, C1 = (double)c1 / 255.0, C2 = (double)c2 / 255.0, CG = (double)g / 255.0;
        const bool certified = foot_sums_certified<DATA_TERM, OUTLIER>(s, n, C0, C1, C2, CG, cert_shift);
        if (!certified) rewalk[atomicAdd(&counters[C_REWALK], 1ull)] = k;   // left to rewalk_info_kernel (keeps the serial walker's registers out of this kernel)
        else {
            foot_finish<DATA_TERM, OUTLIER>(view, s, n, C0, C1, C2, CG, &fi);
            const unsigned long long word = pass[widx];
            const size_t r = (size_t)pass_base[widx] + __popcll(word & ((1ull << bit) - 1ull));
            pq[r] = fi.quality;
            if (OUTLIER) { rgb_to_ycbcr(fi.mean_color); pcol[3 * r] = fi.mean_color[0]; pcol[3 * r + 1] = fi.mean_color[1]; pcol[3 * r + 2] = fi.mean_color[2]; }
            if (fi.quality != 0.0f) { atomicOr(&surv[widx], 1ull << bit); ++cnt[1]; } else ++cnt[0];
        }
    }
    if (STATS) { const int slot[2] = {C_ZEROQ, C_SURV}; block_count_add<2>(cnt, counters, slot); }
}
// The footprints wave_info_kernel could not certify, in the reference's serial fp64 order -- ONE WAVE per footprint.  What is serial by
// definition is the chain of fp64 additions (sum = fl(sum + u / 255.0), pixel after pixel); everything in front of it is not: the 64
// lanes take 64 consecutive pixels of a scan line, test / load / divide in parallel, and then every lane runs the same chain over the
// 64 quotients in pixel order (lane l's quotient comes through v_readlane; a pixel outside the footprint contributes +0.0, which
// changes no non-negative sum).  One thread per footprint walked a 3 000-pixel footprint at one dependent load + division per pixel:
// 0.35 ms for the 300 uncertified footprints of the real-like scene (of 2.8 M), on the critical path.
__device__ __forceinline__ double readlane_f64(double v, int l /* wave-uniform */) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int DATA_TERM, bool OUTLIER>
__device__ void face_info_wave(const ViewParams& view, V3 v1, V3 v2, V3 v3, FaceInfoOut* out) {   // == face_info (dmath.h), all lanes return the same
    const int lane = threadIdx.x & 63;
    FootSetup s;
    foot_setup(view, v1, v2, v3, s);
    out->quality = 0.0f;
    out->mean_color[0] = out->mean_color[1] = out->mean_color[2] = 0.0f;
    if (s.area < FLT_EPSILON) return;
    uint32_t num_samples = 0;
    double col0 = 0.0, col1 = 0.0, col2 = 0.0, gmi = 0.0;
    const int w = view.width;
    const uint8_t* image = view.rgb;
    const uint8_t* gimg = view.gmi;
    if (((DATA_TERM != 0) || OUTLIER) && s.area > 0.5f) {
        foot_edges(s);
        const float y_end = ceilf(s.aabb_max_y);
        for (int y = (int)floorf(s.aabb_min_y); (float)y < y_end; ++y) {
            int xb, xe;
            if (!foot_row(s, y, &xb, &xe)) continue;
            xb = __builtin_amdgcn_readfirstlane(xb); xe = __builtin_amdgcn_readfirstlane(xe);   // (every lane computed the same span)
            for (int x0 = xb; x0 < xe; x0 += 64) {
                const int x = x0 + lane;
                const bool in = x < xe && (s.fast || foot_inside(s, x, y));
                double q0 = 0.0, q1 = 0.0, q2 = 0.0, qg = 0.0;
                if (in) {
                    const size_t pix = (size_t)x + (size_t)y * w;
                    if (OUTLIER) { q0 = (double)image[pix * 3 + 0] / 255.0; q1 = (double)image[pix * 3 + 1] / 255.0; q2 = (double)image[pix * 3 + 2] / 255.0; }
                    if (DATA_TERM == 1) qg = (double)gimg[pix] / 255.0;
                }
                num_samples += (uint32_t)__popcll(__ballot(in));
                const int cnt = min(64, xe - x0);
                for (int l = 0; l < cnt; ++l) {
                    if (OUTLIER) { col0 += readlane_f64(q0, l); col1 += readlane_f64(q1, l); col2 += readlane_f64(q2, l); }
                    if (DATA_TERM == 1) gmi += readlane_f64(qg, l);
                }
            }
        }
    }
    foot_finish<DATA_TERM, OUTLIER>(view, s, num_samples, col0, col1, col2, gmi, out);
}
template <int DATA_TERM, bool OUTLIER, bool STATS>
__global__ void __launch_bounds__(64) rewalk_info_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ faces, const ViewParams* __restrict__ views,
                                                         uint32_t fb, uint32_t fwords, const uint2* __restrict__ list, const uint32_t* __restrict__ rewalk,
                                                         const unsigned long long* __restrict__ pass, const uint32_t* __restrict__ pass_base,
                                                         float* __restrict__ pq, float* __restrict__ pcol, unsigned long long* __restrict__ surv,
                                                         unsigned long long* __restrict__ counters) {
    const uint32_t n_rewalk = (uint32_t)counters[C_REWALK];
    for (uint32_t q = blockIdx.x; q < n_rewalk; q += gridDim.x) {   // one wave (= one block) per footprint
        const uint2 rec = list[rewalk[q]];
        const size_t widx = (size_t)rec.x | ((size_t)(rec.y >> 8) << 32);
        const uint32_t bit = rec.y & 63u;
        const uint32_t j = (uint32_t)(widx / fwords), lf = (uint32_t)(widx % fwords) * 64u + bit;
        const size_t f = (size_t)fb + lf;
        const V3 v1 = ld3(verts, faces[3 * f]), v2 = ld3(verts, faces[3 * f + 1]), v3 = ld3(verts, faces[3 * f + 2]);
        FaceInfoOut fi;
        face_info_wave<DATA_TERM, OUTLIER>(views[j], v1, v2, v3, &fi);
        if (threadIdx.x == 0) {
            const unsigned long long word = pass[widx];
            const size_t r = (size_t)pass_base[widx] + __popcll(word & ((1ull << bit) - 1ull));
            pq[r] = fi.quality;
            if (OUTLIER) { rgb_to_ycbcr(fi.mean_color); pcol[3 * r] = fi.mean_color[0]; pcol[3 * r + 1] = fi.mean_color[1]; pcol[3 * r + 2] = fi.mean_color[2]; }
            if (fi.quality != 0.0f) atomicOr(&surv[widx], 1ull << bit);
            if (STATS) atomicAdd(&counters[fi.quality != 0.0f ? C_SURV : C_ZEROQ], 1ull);
        }
    }
}

// ---- view-major bits -> CSR by face ----
__global__ void __launch_bounds__(256) csr_count_kernel(const unsigned long long* __restrict__ surv, uint32_t n_views, uint32_t nf, uint32_t fwords,
                                                        uint32_t* __restrict__ cnt) {
    const uint32_t lf = blockIdx.x * 256 + threadIdx.x;
    if ((lf >> 6) >= fwords) return;
    const int lane = threadIdx.x & 63;
    uint32_t c = 0;
    // eight views per batch: eight independent fetches in flight instead of a chain of one-word round trips (dc_csr 1.00 -> 0.89 ms at C3)
    const unsigned long long* col = surv + (lf >> 6);
    uint32_t j = 0;
    for (; j + 8 <= n_views; j += 8) {
        unsigned long long w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = col[(size_t)(j + k) * fwords];
#pragma unroll
        for (int k = 0; k < 8; ++k) c += (uint32_t)((w[k] >> lane) & 1ull);
    }
    for (; j < n_views; ++j) c += (uint32_t)((col[(size_t)j * fwords] >> lane) & 1ull);
    if (lf < nf) cnt[lf] = c;
    if (lf == 0) cnt[nf] = 0;  // sentinel so that the scan of nf + 1 entries yields col_ptr[nf] = nnz
}

// The per-face scatter of the surviving pairs into CSR columns, staged through LDS so that HBM sees coalesced rows.  A wave owns
// 64 consecutive faces, whose CSR columns form ONE contiguous chunk of the output; it fills the chunk segment by segment
// (SEG entries in LDS, each lane walking its views and depositing the entries that fall into the segment), then streams
// the segment out with full-width stores.  A direct scatter (one thread per face walking its views) writes one 2-byte and one
// 4-byte element per lane at addresses ~K entries apart: 32-byte sectors of which 2 or 4 bytes are useful (5.7 GB written for 0.53 GB at C3).
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int src /* wave-uniform */) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}
// The walk over the views is latency bound if every view costs a dependent chain of wave-uniform loads (survivor word ->
// pass word / base -> quality): 31 k waves x 2 segments x 200 serial round trips were 2.05 ms at C3.  Here lane L fetches
// the three words of view j0 + L (64 views per round trip), the views that have survivors among the wave's faces are
// then visited through readlane broadcasts four at a time, and the four quality gathers of a group are in flight together.
// OUTLIER: the mean colours (three floats per entry) travel with the qualities, in segments of a quarter of the size.
template <bool OUTLIER>
__global__ void __launch_bounds__(256) csr_write_staged_kernel(const unsigned long long* __restrict__ surv, const unsigned long long* __restrict__ pass,
                                                               const uint32_t* __restrict__ pass_base, const float* __restrict__ pq, const float* __restrict__ pcol,
                                                               uint32_t n_views, uint32_t nf, uint32_t fwords, const uint32_t* __restrict__ col_ptr,
                                                               uint16_t* __restrict__ view_id, float* __restrict__ quality, float* __restrict__ color,
                                                               uint32_t* __restrict__ max_bits /* non-null: also the maximum quality (:278-281), saving a pass over the table */) {
    constexpr int CSR_SEG = OUTLIER ? 512 : 3072;   // shadows the namespace constant: 4 x (2 + 4 [+ 12]) x SEG bytes of LDS per block (74 KB: two blocks per CU).
                                                    // 3072 holds the 64 x 45 entries of a wave at config 3 in ONE pass over the views (2048: two passes, 0.865 -> 0.787 ms;
                                                    // 4096: one block per CU, 1.15 ms; 1024: 0.835)
    __shared__ float s_q[4][CSR_SEG];
    __shared__ uint16_t s_v[4][CSR_SEG];
    __shared__ float s_c[4][OUTLIER ? 3 * CSR_SEG : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t word = blockIdx.x * 4 + wv;                 // one wave per 64 faces (no block-level barrier is used)
    if (word >= fwords) return;
    const uint32_t f0 = word * 64u, lf = f0 + lane;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t c0 = col_ptr[f0], c1 = col_ptr[min(f0 + 64u, nf)];
    const uint32_t k0 = col_ptr[min(lf, nf)];
    float qmax = 0.0f;   // :278 max_quality = 0.0f
    for (uint32_t segbase = c0; segbase < c1; segbase += CSR_SEG) {
        const uint32_t segend = min(segbase + (uint32_t)CSR_SEG, c1);
        uint32_t k = k0;
        for (uint32_t j0 = 0; j0 < n_views; j0 += 64u) {
            const uint32_t jl = j0 + (uint32_t)lane;
            const size_t widx = (size_t)min(jl, n_views - 1u) * fwords + word;
            const unsigned long long sw_l = (jl < n_views) ? surv[widx] : 0ull;
            const unsigned long long pw_l = pass[widx];
            const uint32_t pb_l = pass_base[widx];
            unsigned long long todo = __ballot(sw_l != 0ull);   // views of this round with survivors among the wave's faces
            while (todo != 0ull) {
                int b[4]; bool on[4]; uint32_t kk[4]; float q[4]; float cr[4], cg[4], cb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    b[u] = (todo != 0ull) ? (int)__builtin_ctzll(todo) : -1;
                    if (todo != 0ull) todo &= todo - 1ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    on[u] = false; kk[u] = 0u; q[u] = 0.0f; cr[u] = cg[u] = cb[u] = 0.0f;
                    if (b[u] >= 0) {                           // wave-uniform
                        const unsigned long long sw = readlane64(sw_l, b[u]);
                        if ((sw >> lane) & 1ull) {
                            if (k >= segbase && k < segend) {
                                const unsigned long long pw = readlane64(pw_l, b[u]);
                                const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)pb_l, b[u]);
                                on[u] = true; kk[u] = k - segbase;
                                const size_t r = (size_t)pb + __popcll(pw & lt);
                                q[u] = pq[r];
                                if (OUTLIER) { cr[u] = pcol[3 * r]; cg[u] = pcol[3 * r + 1]; cb[u] = pcol[3 * r + 2]; }
                            }
                            ++k;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (on[u]) {
                        s_v[wv][kk[u]] = (uint16_t)(j0 + (uint32_t)b[u]); s_q[wv][kk[u]] = q[u];
                        if (OUTLIER) { s_c[wv][3 * kk[u]] = cr[u]; s_c[wv][3 * kk[u] + 1] = cg[u]; s_c[wv][3 * kk[u] + 2] = cb[u]; }
                    }
            }
        }
        // LDS operations of one wave complete in order: the deposits above are visible to the reads below
        for (uint32_t i = lane; i < segend - segbase; i += 64u) { const float qv = s_q[wv][i]; view_id[segbase + i] = s_v[wv][i]; quality[segbase + i] = qv; qmax = fmaxf(qmax, qv); }
        if (OUTLIER) for (uint32_t i = lane; i < 3u * (segend - segbase); i += 64u) color[3 * (size_t)segbase + i] = s_c[wv][i];
    }
    if (max_bits) {   // qualities are > 0: uint order = float order; only a wave that can raise the maximum issues the atomic
        for (int o = 32; o > 0; o >>= 1) qmax = fmaxf(qmax, __shfl_xor(qmax, o, 64));
        if (lane == 0 && __float_as_uint(qmax) > __hip_atomic_load(max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_bits, __float_as_uint(qmax));
    }
}

// ---- photometric_outlier_detection (calculate_data_costs.cpp:35-129) ----
// One thread per face, fp64.  Row order = the reference's single-thread order:
// descending view id, i.e. the CSR run walked backwards.
// Eigen's FullPivLU is replaced by a plain full-pivoting 3x3 LU with Eigen's
// rank rule |pivot| > eps * 3 * |max pivot|; inverse = solve(I).
struct Lu3 {
    double lu[3][3]; int p[3], q[3]; double maxpivot; int nonzero;
    __device__ explicit Lu3(const double m[3][3]) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) lu[i][j] = m[i][j];
        for (int i = 0; i < 3; ++i) { p[i] = i; q[i] = i; }
        maxpivot = 0.0; nonzero = 3;
        for (int k = 0; k < 3; ++k) {
            int br = k, bc = k; double big = -1.0;
            for (int c = k; c < 3; ++c) for (int r = k; r < 3; ++r)
                if (fabs(lu[r][c]) > big) { big = fabs(lu[r][c]); br = r; bc = c; }
            if (big == 0.0) { nonzero = k; break; }
            if (big > maxpivot) maxpivot = big;
            if (br != k) { for (int c = 0; c < 3; ++c) { double t = lu[k][c]; lu[k][c] = lu[br][c]; lu[br][c] = t; } int t = p[k]; p[k] = p[br]; p[br] = t; }
            if (bc != k) { for (int r = 0; r < 3; ++r) { double t = lu[r][k]; lu[r][k] = lu[r][bc]; lu[r][bc] = t; } int t = q[k]; q[k] = q[bc]; q[bc] = t; }
            for (int r = k + 1; r < 3; ++r) lu[r][k] /= lu[k][k];
            for (int r = k + 1; r < 3; ++r) for (int c = k + 1; c < 3; ++c) lu[r][c] -= lu[r][k] * lu[k][c];
        }
    }
    __device__ bool invertible() const {
        const double thr = fabs(maxpivot) * (2.220446049250313e-16 * 3.0);
        int rank = 0;
        for (int i = 0; i < nonzero; ++i) rank += (fabs(lu[i][i]) > thr);
        return rank == 3;
    }
    __device__ void inverse(double inv[3][3]) const {
        for (int j = 0; j < 3; ++j) {
            double y[3];
            for (int i = 0; i < 3; ++i) y[i] = (p[i] == j) ? 1.0 : 0.0;
            for (int i = 1; i < 3; ++i) for (int k = 0; k < i; ++k) y[i] -= lu[i][k] * y[k];
            for (int i = 2; i >= 0; --i) { for (int k = i + 1; k < 3; ++k) y[i] -= lu[i][k] * y[k]; y[i] /= lu[i][i]; }
            for (int i = 0; i < 3; ++i) inv[q[i]][j] = y[i];
        }
    }
};

// multi_gauss_unnormalized (util.h:60-66)
__device__ __forceinline__ double multi_gauss_arg(const double x[3], const double mu[3], const double ci[3][3]) {
    double mr[3], w[3];
    for (int a = 0; a < 3; ++a) mr[a] = x[a] - mu[a];
    for (int b = 0; b < 3; ++b) w[b] = ((-0.5 * mr[0]) * ci[0][b] + (-0.5 * mr[1]) * ci[1][b]) + (-0.5 * mr[2]) * ci[2][b];
    return (w[0] * mr[0] + w[1] * mr[1]) + w[2] * mr[2];
}
__device__ __forceinline__ double multi_gauss(const double x[3], const double mu[3], const double ci[3][3]) { return exp(multi_gauss_arg(x, mu, ci)); }
// exp(arg) >= 6e-3 (the inlier test, :99-103) WITHOUT the exponential wherever the answer cannot depend on its rounding:
// exp is monotone and accurate to far better than 1e-9 relative, so an argument more than 1e-9 away from ln(6e-3) decides
// by itself; only arguments inside that band (practically never) evaluate exp().  The fp64 exp was 3/4 of the kernel.
__device__ __forceinline__ bool gauss_at_least_threshold(double arg) {
    const double x0 = -5.115995809754082;   // ln(6e-3)
    if (arg >= x0 + 1e-9) return true;
    if (arg < x0 - 1e-9) return false;       // also false for NaN, like exp(NaN) >= t
    return exp(arg) >= 6e-3;
}
// exp(arg) < 6e-3 (the final clamping test, :121): NOT the negation of the above for a NaN argument -- the reference leaves the
// quality untouched when the gauss value is NaN (NaN < t is false), e.g. after an ill-conditioned inverse that passed isInvertible
__device__ __forceinline__ bool gauss_below_threshold(double arg) {
    const double x0 = -5.115995809754082;   // ln(6e-3)
    if (arg < x0 - 1e-9) return true;
    if (arg >= x0 + 1e-9) return false;
    return exp(arg) < 6e-3;                  // false for NaN, like exp(NaN) < t
}

// photometric_outlier_detection (calculate_data_costs.cpp:35-129), one thread per face, every fp64 sum in the reference's
// order (descending view id: SURVEY.md 8a row D) -- the sums decide inlier sets through a threshold, so their order is part
// of the result.  The loop makes 10 x 12 passes over the face's colours.  LDS = true: the colours and the inlier flags of the
// block's 64 faces are staged once into LDS, laid out [entry][channel][lane] (conflict free: the lanes of a wave read the same
// entry of 64 different faces); every later pass is an LDS read.  Straight from HBM (LDS = false; columns longer than the
// 64 KB of LDS allow) a wave's loads hit 64 different lines per instruction, 120 times over: 97 ms at BASELINE config 3.
template <bool LDS>
__global__ void __launch_bounds__(64) outlier_kernel(const uint32_t* __restrict__ col_ptr, uint32_t nf, const float* __restrict__ color, float* __restrict__ quality,
                                                     uint8_t* __restrict__ inl_g, int mode, uint32_t kcap) {
    extern __shared__ float s_dyn[];
    const uint32_t lf = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x;
    if (lf >= nf) return;
    const int64_t p0 = col_ptr[lf], p1 = col_ptr[lf + 1];
    const int64_t n = p1 - p0;
    if (n == 0) return;
    float* sc = s_dyn; uint8_t* si = reinterpret_cast<uint8_t*>(s_dyn + (size_t)kcap * 192);
#define COL(k, a) (LDS ? sc[((k) * 3 + (a)) * 64 + lane] : color[3 * (p0 + (k)) + (a)])
#define INL(k) (LDS ? si[(k) * 64 + lane] : inl_g[p0 + (k)])
#define SET_INL(k, v) do { if (LDS) si[(k) * 64 + lane] = (v); else inl_g[p0 + (k)] = (v); } while (0)
    const double minimal_covariance = 5e-4;
    const float factor = (mode == MVS_OUTLIER_GAUSS_CLAMPING) ? 1.0f : 0.2f;
    for (int64_t k = 0; k < n; ++k) {
        if (LDS) { sc[(k * 3 + 0) * 64 + lane] = color[3 * (p0 + k)]; sc[(k * 3 + 1) * 64 + lane] = color[3 * (p0 + k) + 1]; sc[(k * 3 + 2) * 64 + lane] = color[3 * (p0 + k) + 2]; }
        SET_INL(k, 1);
    }
    int64_t n_in = n;
    double mean[3] = {0, 0, 0}, cov[3][3], ci[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int it = 0; it < 10; ++it) {
        if (n_in < 4) return;
        // every sum keeps the reference's order (its own sequence of additions); the three means, then the nine covariance
        // entries, are accumulated side by side in ONE walk each: independent fp64 chains, one read of the colours per walk
        // The walks are unrolled by four with the skipped entries turned into additions of +0.0 (exact: an accumulator that
        // starts at +0.0 never becomes -0.0, and x + 0.0 == x): straight-line code whose twelve colour reads are in flight
        // together -- with one wave per SIMD (the LDS footprint of the colours) nothing else hides their latency.
        {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            int64_t k = n - 1;
            for (; k >= 3; k -= 4) {
                float c[4][3]; bool in[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { in[u] = INL(k - u) != 0; c[u][0] = COL(k - u, 0); c[u][1] = COL(k - u, 1); c[u][2] = COL(k - u, 2); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s0 += (double)(in[u] ? c[u][0] : 0.0f); s1 += (double)(in[u] ? c[u][1] : 0.0f); s2 += (double)(in[u] ? c[u][2] : 0.0f); }
            }
            for (; k >= 0; --k) {
                if (!INL(k)) continue;
                s0 += (double)COL(k, 0); s1 += (double)COL(k, 1); s2 += (double)COL(k, 2);
            }
            mean[0] = s0 / (double)n_in; mean[1] = s1 / (double)n_in; mean[2] = s2 / (double)n_in;
        }
        {
            double sc9[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            auto acc = [&](double d0, double d1, double d2) {
                sc9[0][0] += d0 * d0; sc9[0][1] += d0 * d1; sc9[0][2] += d0 * d2;
                sc9[1][0] += d1 * d0; sc9[1][1] += d1 * d1; sc9[1][2] += d1 * d2;
                sc9[2][0] += d2 * d0; sc9[2][1] += d2 * d1; sc9[2][2] += d2 * d2;
            };
            int64_t k = n - 1;
            for (; k >= 3; k -= 4) {
                float c[4][3]; bool in[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { in[u] = INL(k - u) != 0; c[u][0] = COL(k - u, 0); c[u][1] = COL(k - u, 1); c[u][2] = COL(k - u, 2); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {   // a skipped entry contributes products of +0.0
                    const double d0 = in[u] ? (double)c[u][0] - mean[0] : 0.0, d1 = in[u] ? (double)c[u][1] - mean[1] : 0.0, d2 = in[u] ? (double)c[u][2] - mean[2] : 0.0;
                    acc(d0, d1, d2);
                }
            }
            for (; k >= 0; --k) {
                if (!INL(k)) continue;
                acc((double)COL(k, 0) - mean[0], (double)COL(k, 1) - mean[1], (double)COL(k, 2) - mean[2]);
            }
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a][b] = sc9[a][b] / (double)(n_in - 1);
        }
        double mx = 0.0;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) mx = (mx < fabs(cov[a][b])) ? fabs(cov[a][b]) : mx;
        if (mx < minimal_covariance) {
            for (int64_t k = 0; k < n; ++k) if (!INL(k)) quality[p0 + k] = 0.0f;
            return;
        }
        Lu3 lu(cov);
        if (!lu.invertible()) return;
        lu.inverse(ci);
        n_in = 0;
        {
            int64_t k = n - 1;
            for (; k >= 3; k -= 4) {   // four independent entries at a time: their reads and fp64 chains overlap
                double a4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const double c[3] = {(double)COL(k - u, 0), (double)COL(k - u, 1), (double)COL(k - u, 2)}; a4[u] = multi_gauss_arg(c, mean, ci); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint8_t in = gauss_at_least_threshold(a4[u]) ? 1 : 0; SET_INL(k - u, in); n_in += in; }
            }
            for (; k >= 0; --k) {
                const double c[3] = {(double)COL(k, 0), (double)COL(k, 1), (double)COL(k, 2)};
                const uint8_t in = gauss_at_least_threshold(multi_gauss_arg(c, mean, ci)) ? 1 : 0;
                SET_INL(k, in);
                n_in += in;
            }
        }
    }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) ci[a][b] *= (double)factor;
    for (int64_t k = 0; k < n; ++k) {
        const double c[3] = {(double)COL(k, 0), (double)COL(k, 1), (double)COL(k, 2)};
        if (mode == MVS_OUTLIER_GAUSS_DAMPING) quality[p0 + k] = (float)((double)quality[p0 + k] * multi_gauss(c, mean, ci));
        else if (gauss_below_threshold(multi_gauss_arg(c, mean, ci))) quality[p0 + k] = 0.0f;
    }
#undef COL
#undef INL
#undef SET_INL
}
__global__ void max_u32_kernel(const uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, v[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}

// remove quality == 0 entries (calculate_data_costs.cpp:268-270).  One wave per 64 consecutive faces, whose columns are one
// contiguous chunk: the chunk is STREAMED 64 entries at a time (coalesced), never walked column by column.
//   count: the ballot of "quality != 0" of a tile is handed to every lane, and lane f adds the bits that lie inside its own
//          column [k0, k1) -- no search for the face an entry belongs to;
//   copy:  the survivors of the chunk are a contiguous run of the output starting at dst_ptr[first face] (dst_ptr is the scan of
//          the counts), so the copy is a plain order-preserving compaction: ballot, prefix popcount, store.
__global__ void __launch_bounds__(256) nonzero_count_kernel(const uint32_t* __restrict__ col_ptr, uint32_t nf, const float* __restrict__ quality, uint32_t* __restrict__ cnt) {
    const uint32_t word = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t f0 = word * 64u, lf = f0 + lane;
    if (f0 > nf) return;                                          // wave-uniform (f0 == nf: only the sentinel entry cnt[nf] = 0)
    const uint32_t c0 = col_ptr[min(f0, nf)], c1 = col_ptr[min(f0 + 64u, nf)];
    const uint32_t k0 = col_ptr[min(lf, nf)], k1 = col_ptr[min(lf + 1u, nf)];
    uint32_t c = 0;
    for (uint32_t t0 = c0; t0 < c1; t0 += 64u) {
        const uint32_t i = t0 + lane;
        const unsigned long long nz = __ballot(i < c1 && quality[i] != 0.0f);
        // bits [lo, hi) of the tile belong to this lane's column
        const uint32_t lo = max(k0, t0) - t0, hi = min(max(k1, t0), t0 + 64u) - t0;
        if (hi > lo) {
            const unsigned long long m = ((hi >= 64u) ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
            c += (uint32_t)__popcll(nz & m);
        }
    }
    if (lf <= nf) cnt[lf] = (lf < nf) ? c : 0u;
}
__global__ void __launch_bounds__(256) nonzero_copy_kernel(const uint32_t* __restrict__ src_ptr, const uint32_t* __restrict__ dst_ptr, uint32_t nf,
                                                           const uint16_t* __restrict__ sv, const float* __restrict__ sq, uint16_t* __restrict__ dv, float* __restrict__ dq) {
    const uint32_t word = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t f0 = word * 64u;
    if (f0 >= nf) return;                                         // wave-uniform
    const uint32_t c0 = src_ptr[f0], c1 = src_ptr[min(f0 + 64u, nf)];
    uint32_t d = dst_ptr[f0];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t t0 = c0; t0 < c1; t0 += 64u) {
        const uint32_t i = t0 + lane;
        const float q = (i < c1) ? sq[i] : 0.0f;
        const bool keep = i < c1 && q != 0.0f;
        const unsigned long long b = __ballot(keep);
        if (keep) { const uint32_t o = d + (uint32_t)__popcll(b & lt); dv[o] = sv[i]; dq[o] = q; }
        d += (uint32_t)__popcll(b);
    }
}

// ---- postprocess_face_infos (calculate_data_costs.cpp:278-302) ----
__global__ void max_kernel(const float* __restrict__ q, size_t n, uint32_t* __restrict__ max_bits) {
    float m = 0.0f;  // :278 max_quality = 0.0f
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = smax(m, q[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // qualities are > 0: uint order = float order; same-address atomics serialise, so only a wave that can raise the maximum issues one
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __hip_atomic_load(max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_bits, __float_as_uint(m));
}

// Histogram::add_value (histogram.cpp:27-35) with LDS-privatised integer bins; hist[HIST_BINS] = num_values
__global__ void __launch_bounds__(1024) hist_kernel(const float* __restrict__ q, size_t n, const float* __restrict__ max_q, uint32_t* __restrict__ hist) {
    __shared__ uint32_t bins[HIST_BINS];
    for (uint32_t b = threadIdx.x; b < HIST_BINS; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    const float mx = *max_q;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        atomicAdd(&bins[hist_bin(q[i], mx, HIST_BINS)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < HIST_BINS; b += blockDim.x) { const uint32_t c = bins[b]; if (c) atomicAdd(&hist[b], c); }
}

// Histogram::get_approx_percentile (histogram.cpp:49-63)
// Histogram::get_approx_percentile (histogram.cpp:49-63): the reference walks the bins and returns the previous bin's
// bound as soon as float(num) / num_values > percentile, where num = counts of the bins before the current one.
// Parallel form: exclusive prefix sum of the counts, smallest bin index whose test fires, same float expressions.
__global__ void __launch_bounds__(1024) percentile_kernel(const uint32_t* __restrict__ hist, const float* __restrict__ max_q, float percentile, float* __restrict__ out,
                                                          unsigned long long* __restrict__ report /* [0] bits of max_q, [1] bits of the percentile: read back with the counters */) {
    constexpr uint32_t PER = (HIST_BINS + 1023u) / 1024u;
    __shared__ uint32_t s_sum[1024];
    __shared__ uint32_t s_first;
    const uint32_t t = threadIdx.x, b0 = t * PER;
    uint32_t local = 0;
    for (uint32_t k = 0; k < PER; ++k) if (b0 + k < HIST_BINS) local += hist[b0 + k];
    s_sum[t] = local;
    if (t == 0) s_first = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {                     // inclusive Hillis-Steele scan over the thread sums
        const uint32_t v = (t >= o) ? s_sum[t - o] : 0u;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    const float minv = 0.0f, maxv = *max_q;
    const int num_values = (int)hist[HIST_BINS];
    int num = (int)(s_sum[t] - local);                              // counts of all bins before b0
    uint32_t first = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < PER && b0 + k < HIST_BINS; ++k) {
        if (first == 0xFFFFFFFFu && (float)num / (float)num_values > percentile) first = b0 + k;
        num += (int)hist[b0 + k];
    }
    if (first != 0xFFFFFFFFu) atomicMin(&s_first, first);
    __syncthreads();
    if (t == 0) {
        const uint32_t i = s_first;
        if (i == 0xFFFFFFFFu) *out = maxv;
        else if (i == 0u) *out = minv;
        else *out = ((float)(i - 1u) / (float)(HIST_BINS - 1)) * (maxv - minv) + minv;
        if (report) { report[0] = __float_as_uint(maxv); report[1] = __float_as_uint(*out); }
    }
}

// cost = 1 - min(1, quality / percentile)  (calculate_data_costs.cpp:295-297)
__global__ void cost_kernel(const float* __restrict__ q, size_t n, const float* __restrict__ pctl, float* __restrict__ cost) {
    const float p = *pctl;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        cost[i] = 1.0f - smin(1.0f, q[i] / p);
}

__global__ void set_count_kernel(uint32_t* __restrict__ hist, uint32_t n) { hist[HIST_BINS] = n; }

}  // namespace

static uint32_t read_u32(mvs_ctx* ctx, const uint32_t* d) {
    uint32_t h = 0;
    read_words(ctx, d, &h, 1);
    return h;
}

// outlier detection over a pre-CSR: LDS-staged when the longest column fits (counts: per-face lengths on the device, or null
// with the longest column given in kmax_known)
static void launch_outlier(mvs_ctx* ctx, const uint32_t* ptr, const uint32_t* counts, uint32_t kmax_known, uint32_t nf, const float* col, float* q, uint8_t* inl, int mode);

// (re)compute per-view derived images and upload the view table
static void upload_views_and_prepare(mvs_ctx* ctx, bool need_gmi) {
    const uint32_t V = ctx->n_views;
    ctx->gmi_off.assign(V + 1, 0); ctx->mask_off.assign(V + 1, 0); ctx->msum_off.assign(V + 1, 0);
    for (uint32_t j = 0; j < V; ++j) {
        auto& v = ctx->h_views[j];
        v.mask_stride = (v.width + 31) / 32;
        v.msum_stride = ((v.mask_stride + 63) / 64) * 2;   // one bit per 32-pixel tile column, rows padded to 64 tiles (a wave's ballot)
        ctx->gmi_off[j + 1] = ctx->gmi_off[j] + (need_gmi ? (((size_t)v.width * v.height + 15) & ~(size_t)15) : 0);
        ctx->mask_off[j + 1] = ctx->mask_off[j] + (size_t)v.mask_stride * v.height;
        ctx->msum_off[j + 1] = ctx->msum_off[j] + (size_t)v.msum_stride * ((v.height + 31) / 32);
    }
    ctx->gmi_all.ensure(std::max<size_t>(ctx->gmi_off[V], 16));
    ctx->mask_all.ensure(std::max<size_t>(ctx->mask_off[V], 16));
    ctx->mask_zero.ensure(std::max<size_t>(ctx->mask_off[V], 16));
    ctx->mask_tmp.ensure(std::max<size_t>(ctx->mask_off[V], 16));
    ctx->msum_all.ensure(std::max<size_t>(ctx->msum_off[V], 16));
    for (uint32_t j = 0; j < V; ++j) {
        ctx->h_views[j].gmi = need_gmi ? ctx->gmi_all.p + ctx->gmi_off[j] : nullptr;
        ctx->h_views[j].mask = ctx->mask_all.p + ctx->mask_off[j];
        ctx->h_views[j].msum = ctx->msum_all.p + ctx->msum_off[j];
    }
    ctx->d_views.ensure(V);
    MVS_HIP(hipMemcpyAsync(ctx->d_views.p, ctx->h_views.data(), V * sizeof(ViewParams), hipMemcpyHostToDevice, ctx->stream));
    ctx->view_off.ensure(2 * ((size_t)V + 1));
    size_t* d_off = ctx->view_off.p;
    MVS_HIP(hipMemcpyAsync(d_off, ctx->gmi_off.data(), (V + 1) * sizeof(size_t), hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(hipMemcpyAsync(d_off + V + 1, ctx->mask_off.data(), (V + 1) * sizeof(size_t), hipMemcpyHostToDevice, ctx->stream));
    prepare_views(ctx, need_gmi, d_off, d_off + V + 1);
}

// phase 1: everything up to the per-face sorted infos + local max quality
static bool dc_phase1_once(mvs_ctx* ctx, const mvs_settings* st);
void dc_phase1(mvs_ctx* ctx, const mvs_settings* st) {
    // (a second pass only when the face order had to be rebuilt -- scene_order_commit: a mesh with thousands of equal centroid coordinates)
    if (dc_phase1_once(ctx, st)) (void)dc_phase1_once(ctx, st);
}
static bool dc_phase1_once(mvs_ctx* ctx, const mvs_settings* st) {
    if (!ctx->d_verts) throw StatusError(MVS_ERR_STATE, "scene not set (mesh + views)");
    /* calculate_data_costs.cpp:315-318 -- F is a uint32 here, so only the view guard can fire */
    if (ctx->n_views > 65535u) throw StatusError(MVS_ERR_TOO_MANY_VIEWS, "Exeeded maximal number of views");
    if (st->data_term != MVS_DATA_TERM_AREA && st->data_term != MVS_DATA_TERM_GMI) throw StatusError(MVS_ERR_INVALID, "bad data_term");
    if (st->outlier_removal < 0 || st->outlier_removal > 2) throw StatusError(MVS_ERR_INVALID, "bad outlier_removal");
    hipStream_t s = ctx->stream;
    const uint32_t V = ctx->n_views, fb = ctx->face_begin, nf = ctx->face_end - ctx->face_begin;
    const uint32_t fwords = (nf + 63) / 64, vwords = (ctx->n_verts + 63) / 64;
    const bool gmi = st->data_term == MVS_DATA_TERM_GMI, outl = st->outlier_removal != MVS_OUTLIER_NONE, vis = st->geometric_visibility_test != 0;
    ctx->dc_settings = *st; ctx->have_costs = false;
    ctx->t_perm = nullptr; ctx->t_pos = nullptr; ctx->u_valid = false;
    memset(&ctx->dc_stats, 0, sizeof(ctx->dc_stats));
    ctx->dc_stats.pairs = (uint64_t)nf * V;
    ctx->counters.ensure(64);
    MVS_HIP(hipMemsetAsync(ctx->counters.p, 0, 64 * sizeof(unsigned long long), s));

    const uint32_t nf_early = ctx->face_end - ctx->face_begin;
    if (nf_early == 0 || ctx->n_views == 0 || ctx->n_faces == 0) {
        // nothing to evaluate: every column is empty (the reference's loops simply do not execute)
        ctx->csr_ptr.ensure((size_t)nf_early + 2); ctx->csr_view.ensure(4); ctx->csr_q.ensure(4); ctx->csr_cost.ensure(4);
        MVS_HIP(hipMemsetAsync(ctx->csr_ptr.p, 0, ((size_t)nf_early + 2) * sizeof(uint32_t), s));
        ctx->csr_faces = nf_early; ctx->csr_views = ctx->n_views; ctx->csr_nnz = 0; ctx->dc_phase = 1;
        ctx->max_q.ensure(4); MVS_HIP(hipMemsetAsync(ctx->max_q.p, 0, 4 * sizeof(float), s));
        return false;
    }
    // the mesh in the library's own layout (faces [fb, fb + nf) are POSITIONS of that layout); a table over the whole mesh remembers
    // its order so that it crosses the ABI in the caller's numbering
    // The image preparation (:157-163) depends on nothing the face order or the BVH build (:144) produce, and neither fills the machine
    // (prep waits on memory half of its wave cycles, the order / BVH builds are a hundred short launches): prep runs on a second stream
    // beside BOTH.  The fork is recorded HERE, in front of the order's launches (recorded behind them -- as it was until the kernel timeline
    // of a step showed the main stream idle for 0.7 ms in front of the culls -- prep overlapped the BVH build only); prep's launches are
    // still QUEUED after the order's and the BVH's, because prep's flood fill makes the host wait (for its own stream only).
    const bool fork = ctx->dc_overlap_prep;
    if (fork) {
        if (!ctx->aux_stream) { MVS_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking)); MVS_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)); MVS_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming)); }
        MVS_HIP(hipEventRecord(ctx->ev_fork, s));                          // (the counters' memset above is what prep has to see)
        MVS_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
    }
    if (!(ctx->order_pinned && ctx->iv)) { Prof pr(ctx, "dc_order"); build_scene_order(ctx); }   // (pinned: a shard owns this layout, see ctx.h)
    // (a face RANGE's table remembers which faces its columns belong to -- t_perm = the range's slice of the order -- but has no inverse:
    //  it leaves as it is, columns in position order, and mvs_ctx_table_order names the faces)
    if (ctx->mesh_ordered) {
        if (fb == 0 && nf == ctx->n_faces) { ctx->t_perm = ctx->f_perm.p; ctx->t_pos = ctx->f_pos.p; }
        else { ctx->t_perm = ctx->f_perm.p + fb; ctx->t_pos = nullptr; }
    }
    if (!fork) { Prof pr(ctx, "dc_prep"); upload_views_and_prepare(ctx, gmi); }
    if (vis) { Prof pr(ctx, "dc_bvh_build"); build_bvh(ctx); }
    if (fork) {
        struct Swap { mvs_ctx* c; hipStream_t main; ~Swap() { c->stream = main; } } swap{ctx, s};
        ctx->stream = ctx->aux_stream;
        { Prof pr(ctx, "dc_prep"); upload_views_and_prepare(ctx, gmi); }
        MVS_HIP(hipEventRecord(ctx->ev_join, ctx->aux_stream));
        ctx->stream = s;
        MVS_HIP(hipStreamWaitEvent(s, ctx->ev_join, 0));
    }

    const size_t pw = (size_t)V * fwords;
    ctx->pass_bits.ensure(pw + 1); ctx->surv_bits.ensure(pw + 1); ctx->pass_base.ensure(pw + 2);
    const dim3 fgrid((nf + 255) / 256, (V + VIEW_CHUNK - 1) / VIEW_CHUNK);
    uint32_t* pass_face = nullptr;
    if (vis) { ctx->pass_face.ensure((size_t)fgrid.y * fwords * 64u + 1); pass_face = ctx->pass_face.p; }
    Prof pr_cull(ctx, "dc_cull");
    // one block per 256 faces walks all view chunks; the chunks are split over blockIdx.y only to reach ~4096 blocks on small meshes
    const dim3 cgrid(fgrid.x, std::max(1u, std::min<unsigned>(fgrid.y, (4096u + fgrid.x - 1) / fgrid.x)));
    if (ctx->stats)
        hipLaunchKernelGGL(cull_kernel<true>, cgrid, dim3(256), 0, s, ctx->iv, ctx->ifc, ctx->inr, ctx->d_views.p, V, fb, nf, fwords,
                           ctx->cos_limit, ctx->pass_bits.p, pass_face, ctx->counters.p);
    else
        hipLaunchKernelGGL(cull_kernel<false>, cgrid, dim3(256), 0, s, ctx->iv, ctx->ifc, ctx->inr, ctx->d_views.p, V, fb, nf, fwords,
                           ctx->cos_limit, ctx->pass_bits.p, pass_face, ctx->counters.p);
    MVS_LAUNCH_CHECK();
    pr_cull.end();
    if (vis) {
        const size_t vw = (size_t)V * vwords;
        ctx->need_bits.ensure(vw + 1); ctx->occl_bits.ensure(vw + 1);
        MVS_HIP(hipMemsetAsync(ctx->occl_bits.p, 0, vw * sizeof(unsigned long long), s));
        const dim3 vgrid((ctx->n_verts + 255) / 256, (V + VIEW_CHUNK - 1) / VIEW_CHUNK);
        Prof pr_need(ctx, "dc_need");
        if (ctx->stats)
            hipLaunchKernelGGL(need_kernel<true>, vgrid, dim3(256), 0, s, ctx->vf_ptr.p, ctx->vf.p, ctx->n_verts, V, fb, nf, fwords, vwords,
                               pass_face, ctx->need_bits.p, ctx->counters.p);
        else
            hipLaunchKernelGGL(need_kernel<false>, vgrid, dim3(256), 0, s, ctx->vf_ptr.p, ctx->vf.p, ctx->n_verts, V, fb, nf, fwords, vwords,
                               pass_face, ctx->need_bits.p, ctx->counters.p);
        MVS_LAUNCH_CHECK();
        pr_need.end();
        Prof pr_rays(ctx, "dc_rays");
        trace_rays(ctx);
        pr_rays.end();
    }
    // rank of every passing pair
    Prof pr_rank(ctx, "dc_rank_scan");
    hipLaunchKernelGGL(popc_kernel, dim3((unsigned)((pw + 255) / 256)), dim3(256), 0, s, ctx->pass_bits.p, ctx->pass_base.p, pw);
    MVS_LAUNCH_CHECK();
    ctx->max_q.ensure(4);
    uint32_t* d_total = (uint32_t*)(ctx->max_q.p + 2);
    // ranks, CSR pointers and nnz are 32 bits wide (the reference uses size_t containers): a scene whose passing pairs could
    // reach 2^32 gets an exact 64-bit count first and is refused instead of wrapping the scan (shard the faces: face ranges)
    if ((uint64_t)nf * V >= 0xFFFFFFF0ull && sum_u32(ctx, ctx->pass_base.p, pw) >= 0xFFFFFFF0ull)
        throw StatusError(MVS_ERR_UNSUPPORTED, "more than 2^32 (face, view) pairs pass the culls in one context: evaluate the faces in ranges (mvs_scene_set_face_range)");
    exclusive_scan_u32(ctx, ctx->pass_base.p, ctx->pass_base.p, pw, d_total);
    pr_rank.end();
    const uint32_t n_pass = read_u32(ctx, d_total);
    if (scene_order_commit(ctx)) return true;                 // (the stream is drained here anyway) the order was rebuilt: once more from the top
    ctx->pq.ensure((size_t)n_pass + 1);
    if (outl) ctx->pcol.ensure(3 * ((size_t)n_pass + 1));

    Prof pr_info(ctx, "dc_face_info");
    // footprints above info_wave_area pixels (and only footprints that are sampled at all: gmi or outlier removal) are left to
    // the wave-per-footprint kernel
    const bool defer = (gmi || outl) && ctx->info_wave_area > 0;
    ctx->dc_stats_deferred = 0;
    const bool words = defer && ctx->info_words && gmi && !outl;   // the integer walk needs the lane-group kernel behind it (for what it cannot certify)
    // (where info_kernel itself reads four pixels per load -- WORDS -- one lane keeps up with the lane group up to a few hundred pixels)
    const float defer_area = defer ? (float)(words ? std::max(ctx->info_wave_area, ctx->info_wave_area_words) : ctx->info_wave_area) : INFINITY;
    unsigned long long* defer_bits = nullptr;
    if (defer) { ctx->defer_bits.ensure(pw + 1); defer_bits = ctx->defer_bits.p; }
#define LAUNCH_INFO(DT, OL, VT, WD)                                                                                          \
    do { if (ctx->stats) LAUNCH_INFO2(DT, OL, VT, true, WD); else LAUNCH_INFO2(DT, OL, VT, false, WD); } while (0)
#define LAUNCH_INFO2(DT, OL, VT, ST, WD)                                                                                     \
    hipLaunchKernelGGL((info_kernel<DT, OL, VT, ST, WD>), fgrid, dim3(256), 0, s, ctx->iv, ctx->ifc, ctx->d_views.p, V, fb, nf, \
                       fwords, vwords, ctx->pass_bits.p, ctx->occl_bits.p, ctx->pass_base.p, ctx->pq.p, ctx->pcol.p,         \
                       ctx->surv_bits.p, ctx->counters.p, defer_area, defer_bits, ctx->info_cert_shift)
    if (gmi) { if (outl) { if (vis) LAUNCH_INFO(1, true, true, false); else LAUNCH_INFO(1, true, false, false); }
               else if (words) { if (vis) LAUNCH_INFO(1, false, true, true); else LAUNCH_INFO(1, false, false, true); }
               else      { if (vis) LAUNCH_INFO(1, false, true, false); else LAUNCH_INFO(1, false, false, false); } }
    else     { if (outl) { if (vis) LAUNCH_INFO(0, true, true, false); else LAUNCH_INFO(0, true, false, false); }
               else      { if (vis) LAUNCH_INFO(0, false, true, false); else LAUNCH_INFO(0, false, false, false); } }
#undef LAUNCH_INFO
#undef LAUNCH_INFO2
    MVS_LAUNCH_CHECK();
    if (defer) {
        // deferred pairs: bit matrix -> ranks -> list (no atomics), then one wave per pair
        ctx->defer_base.ensure(pw + 2);
        hipLaunchKernelGGL(popc_kernel, dim3((unsigned)((pw + 255) / 256)), dim3(256), 0, s, defer_bits, ctx->defer_base.p, pw); MVS_LAUNCH_CHECK();
        uint32_t* d_total2 = (uint32_t*)(ctx->max_q.p + 3);
        exclusive_scan_u32(ctx, ctx->defer_base.p, ctx->defer_base.p, pw, d_total2);
        const uint32_t n_def = read_u32(ctx, d_total2);
        ctx->dc_stats_deferred = n_def;
        if (n_def) {
            ctx->defer_list.ensure((size_t)n_def + 1);
            hipLaunchKernelGGL(defer_expand_kernel, dim3((unsigned)((pw + 255) / 256)), dim3(256), 0, s, defer_bits, ctx->defer_base.p, pw, ctx->defer_list.p); MVS_LAUNCH_CHECK();
            const dim3 wgrid((unsigned)(((size_t)n_def * 16 + 255) / 256));
#define WAVE_ARGS ctx->iv, ctx->ifc, ctx->d_views.p, fb, fwords, ctx->defer_list.p, n_def, ctx->pass_bits.p, ctx->pass_base.p, ctx->pq.p, ctx->pcol.p, ctx->surv_bits.p, ctx->counters.p, ctx->info_cert_shift, ctx->rewalk_list.p
#define REWALK_ARGS ctx->iv, ctx->ifc, ctx->d_views.p, fb, fwords, ctx->defer_list.p, ctx->rewalk_list.p, ctx->pass_bits.p, ctx->pass_base.p, ctx->pq.p, ctx->pcol.p, ctx->surv_bits.p, ctx->counters.p
#define LAUNCH_WAVE(DT, OL) do { if (ctx->stats) { hipLaunchKernelGGL((wave_info_kernel<DT, OL, true>), wgrid, dim3(256), 0, s, WAVE_ARGS); hipLaunchKernelGGL((rewalk_info_kernel<DT, OL, true>), rgrid, dim3(64), 0, s, REWALK_ARGS); } \
                                 else { hipLaunchKernelGGL((wave_info_kernel<DT, OL, false>), wgrid, dim3(256), 0, s, WAVE_ARGS); hipLaunchKernelGGL((rewalk_info_kernel<DT, OL, false>), rgrid, dim3(64), 0, s, REWALK_ARGS); } } while (0)
            ctx->rewalk_list.ensure((size_t)n_def + 1);
            // one wave per uncertified footprint, a fixed grid striding over the device-side count (no read-back): 2048 waves
            // cover what a scene produces (tens to hundreds) one each; the test hook that fails every certificate takes the grid-stride loop
            const dim3 rgrid(2048);
            if (gmi) { if (outl) LAUNCH_WAVE(1, true); else LAUNCH_WAVE(1, false); } else LAUNCH_WAVE(0, true);
#undef LAUNCH_WAVE
#undef WAVE_ARGS
#undef REWALK_ARGS
            MVS_LAUNCH_CHECK();
        }
    }
    pr_info.end();
    Prof pr_csr(ctx, "dc_csr");

    // CSR by face
    ctx->face_cnt.ensure((size_t)nf + 2);
    hipLaunchKernelGGL(csr_count_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, ctx->surv_bits.p, V, nf, fwords, ctx->face_cnt.p);
    MVS_LAUNCH_CHECK();
    DBuf<uint32_t>& ptr = outl ? ctx->pre_ptr : ctx->csr_ptr;
    ptr.ensure((size_t)nf + 2);
    exclusive_scan_u32(ctx, ctx->face_cnt.p, ptr.p, (size_t)nf + 1, nullptr);
    const uint32_t nnz_pre = read_u32(ctx, ptr.p + nf);
    ctx->dc_stats.nnz_pre = nnz_pre;
    if (outl) {
        ctx->pre_view.ensure((size_t)nnz_pre + 1); ctx->pre_q.ensure((size_t)nnz_pre + 1); ctx->pre_col.ensure(3 * ((size_t)nnz_pre + 1));
        ctx->pre_inl.ensure((size_t)nnz_pre + 1);
        hipLaunchKernelGGL(csr_write_staged_kernel<true>, dim3((fwords + 3) / 4), dim3(256), 0, s, ctx->surv_bits.p, ctx->pass_bits.p, ctx->pass_base.p,
                           ctx->pq.p, ctx->pcol.p, V, nf, fwords, ctx->pre_ptr.p, ctx->pre_view.p, ctx->pre_q.p, ctx->pre_col.p, (uint32_t*)nullptr);
        MVS_LAUNCH_CHECK();
        launch_outlier(ctx, ctx->pre_ptr.p, ctx->face_cnt.p /* still the per-face counts of csr_count_kernel */, 0u, nf, ctx->pre_col.p, ctx->pre_q.p, ctx->pre_inl.p, st->outlier_removal);
        hipLaunchKernelGGL(nonzero_count_kernel, dim3((nf / 64u + 1u + 3u) / 4u), dim3(256), 0, s, ctx->pre_ptr.p, nf, ctx->pre_q.p, ctx->face_cnt.p);
        MVS_LAUNCH_CHECK();
        ctx->csr_ptr.ensure((size_t)nf + 2);
        exclusive_scan_u32(ctx, ctx->face_cnt.p, ctx->csr_ptr.p, (size_t)nf + 1, nullptr);
        const uint32_t nnz = read_u32(ctx, ctx->csr_ptr.p + nf);
        ctx->csr_view.ensure((size_t)nnz + 1); ctx->csr_q.ensure((size_t)nnz + 1); ctx->csr_cost.ensure((size_t)nnz + 1);
        hipLaunchKernelGGL(nonzero_copy_kernel, dim3(((nf + 63u) / 64u + 3u) / 4u), dim3(256), 0, s, ctx->pre_ptr.p, ctx->csr_ptr.p, nf, ctx->pre_view.p, ctx->pre_q.p,
                           ctx->csr_view.p, ctx->csr_q.p);
        MVS_LAUNCH_CHECK();
        ctx->csr_nnz = nnz;
    } else {
        ctx->csr_view.ensure((size_t)nnz_pre + 1); ctx->csr_q.ensure((size_t)nnz_pre + 1); ctx->csr_cost.ensure((size_t)nnz_pre + 1);
        MVS_HIP(hipMemsetAsync(ctx->max_q.p, 0, 2 * sizeof(float), s));   // the kernel also leaves the maximum quality behind
        hipLaunchKernelGGL(csr_write_staged_kernel<false>, dim3((fwords + 3) / 4), dim3(256), 0, s, ctx->surv_bits.p, ctx->pass_bits.p, ctx->pass_base.p,
                           ctx->pq.p, (const float*)nullptr, V, nf, fwords, ctx->csr_ptr.p, ctx->csr_view.p, ctx->csr_q.p, (float*)nullptr, (uint32_t*)ctx->max_q.p);
        MVS_LAUNCH_CHECK();
        ctx->csr_nnz = nnz_pre;
    }
    pr_csr.end();
    Prof pr_post(ctx, "dc_post");
    ctx->csr_faces = nf; ctx->csr_views = V;
    // local max quality (:278-281): after outlier removal a pass of its own, otherwise left behind by the CSR kernel
    if (outl) {
        MVS_HIP(hipMemsetAsync(ctx->max_q.p, 0, 2 * sizeof(float), s));
        if (ctx->csr_nnz) {
            hipLaunchKernelGGL(max_kernel, dim3(1024), dim3(256), 0, s, ctx->csr_q.p, (size_t)ctx->csr_nnz, (uint32_t*)ctx->max_q.p);
            MVS_LAUNCH_CHECK();
        }
    }
    ctx->dc_phase = 1;
    return false;
}

// ---- label-space compression (BASELINE config 5: hundreds of candidate views per face) ----
// Keeps, per face, the `kmax` entries with the smallest (cost, view id) pairs -- ties to the smaller view id -- in ascending
// view order; columns of at most kmax entries are untouched.  NOT part of the reference (its model keeps every candidate,
// view_selection.cpp:46-47): an explicit option (mvs_set_option "max_labels" / mvs_ctx_prune_labels), off by default,
// restated identically in the oracle (its prune_labels entry point).  With kmax <= 255 every column fits the solver's fast path.
__global__ void prune_count_kernel(const uint32_t* __restrict__ col_ptr, uint32_t nf, uint32_t kmax, uint32_t* __restrict__ cnt) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f <= nf) cnt[f] = f < nf ? min(col_ptr[f + 1] - col_ptr[f], kmax) : 0u;
}
constexpr uint32_t PRUNE_TILE = 1024;   // columns up to this length are selected from registers (16 keys per lane)
__global__ void __launch_bounds__(256) prune_write_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                          const float* __restrict__ q, uint32_t nf, uint32_t kmax, const uint32_t* __restrict__ new_ptr,
                                                          uint16_t* __restrict__ view2, float* __restrict__ cost2, float* __restrict__ q2) {
    const uint32_t f = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (f >= nf) return;                                       // wave-uniform
    const uint32_t p0 = col_ptr[f], K = col_ptr[f + 1] - p0, o = new_ptr[f];
    if (K <= kmax) {
        for (uint32_t t = lane; t < K; t += 64) { view2[o + t] = view_id[p0 + t]; cost2[o + t] = cost[p0 + t]; if (q) q2[o + t] = q[p0 + t]; }
        return;
    }
    if (K <= PRUNE_TILE) {
        // Selection instead of ranking: the kmax smallest (cost, position) pairs are those below the kmax-th smallest cost T, plus
        // the first (kmax - #below) entries equal to T in position order.  T is built bit by bit (32 steps of "how many keys lie
        // below this candidate", a ballot and a popcount per register) on order-preserving integer keys held in registers:
        // ~400 instructions per column where ranking every entry against every other took K^2 / 64 LDS reads (K = 220: ~4400).
        constexpr int R = (int)(PRUNE_TILE / 64);
        uint32_t key[R];
        const uint32_t nreg = (K + 63u) / 64u;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t t = (uint32_t)r * 64u + lane;
            uint32_t k = 0xFFFFFFFFu;
            if ((uint32_t)r < nreg && t < K) { const uint32_t u = __float_as_uint(cost[p0 + t] + 0.0f); k = (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
            key[r] = k;   // entries beyond the column: the largest key (a real cost is never the NaN pattern that maps there)
        }
        auto below = [&](uint32_t T) {
            uint32_t n = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) if ((uint32_t)r < nreg) n += (uint32_t)__popcll(__ballot(key[r] < T));
            return n;
        };
        uint32_t T = 0;
        for (int bit = 31; bit >= 0; --bit) { const uint32_t Tc = T | (1u << bit); if (below(Tc) <= kmax - 1u) T = Tc; }
        const uint32_t need_eq = kmax - below(T);   // >= 1: the entry of rank kmax - 1 has key T
        const unsigned long long lt = (1ull << lane) - 1ull;
        uint32_t base = 0, eq_seen = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((uint32_t)r >= nreg) break;
            const uint32_t t = (uint32_t)r * 64u + lane;
            const bool in = t < K, eq = in && key[r] == T;
            const unsigned long long eqb = __ballot(eq);
            const bool keep = in && (key[r] < T || (eq && eq_seen + (uint32_t)__popcll(eqb & lt) < need_eq));
            const unsigned long long b = __ballot(keep);
            if (keep) {
                const uint32_t d = o + base + (uint32_t)__popcll(b & lt);
                view2[d] = view_id[p0 + t]; cost2[d] = cost[p0 + t]; if (q) q2[d] = q[p0 + t];
            }
            base += (uint32_t)__popcll(b); eq_seen += (uint32_t)__popcll(eqb);
        }
        return;
    }
    // longer columns: rank every entry against the column in global memory
    uint32_t base = 0;
    for (uint32_t t0 = 0; t0 < K; t0 += 64) {
        const uint32_t t = t0 + lane;
        bool keep = false;
        if (t < K) {
            const float c = cost[p0 + t];
            uint32_t rank = 0;                                 // entries ordered before t by (cost, position); positions ascend with the view id
            for (uint32_t u = 0; u < K; ++u) { const float cu = cost[p0 + u]; rank += (cu < c || (cu == c && u < t)) ? 1u : 0u; }
            keep = rank < kmax;
        }
        const unsigned long long b = __ballot(keep);
        if (keep) {
            const uint32_t d = o + base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            view2[d] = view_id[p0 + t]; cost2[d] = cost[p0 + t]; if (q) q2[d] = q[p0 + t];
        }
        base += (uint32_t)__popcll(b);
    }
}

// std::sort of a face's infos by view id (calculate_data_costs.cpp:272, operator< of FaceProjectionInfo, texture_view.h:31-33):
// one thread per face, insertion sort in place (columns are short; equal ids -- which a caller never produces -- keep their order)
__global__ void sort_columns_kernel(const uint32_t* __restrict__ col_ptr, uint32_t nf, uint16_t* __restrict__ view_id, float* __restrict__ q) {
    const uint32_t lf = blockIdx.x * blockDim.x + threadIdx.x;
    if (lf >= nf) return;
    const uint32_t p0 = col_ptr[lf], p1 = col_ptr[lf + 1];
    for (uint32_t a = p0 + 1; a < p1; ++a) {
        const uint16_t v = view_id[a]; const float x = q[a];
        uint32_t b = a;
        while (b > p0 && view_id[b - 1] > v) { view_id[b] = view_id[b - 1]; q[b] = q[b - 1]; --b; }
        view_id[b] = v; q[b] = x;
    }
}

// tex::postprocess_face_infos (calculate_data_costs.cpp:253-306, exported at texturing.h:71-74) on caller-provided infos:
// per face outlier detection in the order the infos are given (:265-267), erase quality == 0 (:268-270), sort by view id
// (:272); then the global maximum, the histogram and the percentile (:278-288) and the costs (:291-298) -- phases 2 and 3.
// The arrays arrive with every face's list REVERSED: outlier_kernel walks a list back to front (its other caller holds
// the lists ascending by view, the order the reference reaches them in is descending).
void dc_postprocess(mvs_ctx* ctx, uint32_t nf, uint32_t n_views, const uint32_t* h_ptr, const uint16_t* h_view_rev, const float* h_q_rev,
                    const float* h_col_rev, const mvs_settings* st) {
    if (st->outlier_removal < 0 || st->outlier_removal > 2) throw StatusError(MVS_ERR_INVALID, "bad outlier_removal");
    if (n_views > 65535u) throw StatusError(MVS_ERR_TOO_MANY_VIEWS, "Exeeded maximal number of views");
    hipStream_t s = ctx->stream;
    const bool outl = st->outlier_removal != MVS_OUTLIER_NONE;
    const uint32_t n = h_ptr[nf];
    ctx->dc_settings = *st; ctx->have_costs = false;
    ctx->t_perm = nullptr; ctx->t_pos = nullptr; ctx->u_valid = false;   // the caller's infos, the caller's order
    memset(&ctx->dc_stats, 0, sizeof(ctx->dc_stats));
    ctx->dc_stats.nnz_pre = n;
    ctx->counters.ensure(64);
    MVS_HIP(hipMemsetAsync(ctx->counters.p, 0, 64 * sizeof(unsigned long long), s));
    ctx->pre_ptr.ensure((size_t)nf + 2); ctx->pre_view.ensure((size_t)n + 1); ctx->pre_q.ensure((size_t)n + 1);
    MVS_HIP(hipMemcpyAsync(ctx->pre_ptr.p, h_ptr, ((size_t)nf + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (n) {
        MVS_HIP(hipMemcpyAsync(ctx->pre_view.p, h_view_rev, (size_t)n * sizeof(uint16_t), hipMemcpyHostToDevice, s));
        MVS_HIP(hipMemcpyAsync(ctx->pre_q.p, h_q_rev, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
    }
    if (outl && n) {
        ctx->pre_col.ensure(3 * ((size_t)n + 1)); ctx->pre_inl.ensure((size_t)n + 1);
        MVS_HIP(hipMemcpyAsync(ctx->pre_col.p, h_col_rev, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
        uint32_t kmax_h = 0; for (uint32_t i = 0; i < nf; ++i) kmax_h = std::max(kmax_h, h_ptr[i + 1] - h_ptr[i]);
        launch_outlier(ctx, ctx->pre_ptr.p, nullptr, kmax_h, nf, ctx->pre_col.p, ctx->pre_q.p, ctx->pre_inl.p, st->outlier_removal);
    }
    ctx->face_cnt.ensure((size_t)nf + 2); ctx->csr_ptr.ensure((size_t)nf + 2);
    uint32_t nnz = n;
    if (outl) {   // the zero-quality erase belongs to the outlier branch (:265-271): without outlier removal every info stays
        hipLaunchKernelGGL(nonzero_count_kernel, dim3((nf / 64u + 1u + 3u) / 4u), dim3(256), 0, s, ctx->pre_ptr.p, nf, ctx->pre_q.p, ctx->face_cnt.p);
        MVS_LAUNCH_CHECK();
        exclusive_scan_u32(ctx, ctx->face_cnt.p, ctx->csr_ptr.p, (size_t)nf + 1, nullptr);
        nnz = read_u32(ctx, ctx->csr_ptr.p + nf);
    } else MVS_HIP(hipMemcpyAsync(ctx->csr_ptr.p, ctx->pre_ptr.p, ((size_t)nf + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    ctx->csr_view.ensure((size_t)nnz + 1); ctx->csr_q.ensure((size_t)nnz + 1); ctx->csr_cost.ensure((size_t)nnz + 8);
    if (nf) {
        if (outl) {
            hipLaunchKernelGGL(nonzero_copy_kernel, dim3(((nf + 63u) / 64u + 3u) / 4u), dim3(256), 0, s, ctx->pre_ptr.p, ctx->csr_ptr.p, nf, ctx->pre_view.p, ctx->pre_q.p, ctx->csr_view.p, ctx->csr_q.p);
            MVS_LAUNCH_CHECK();
        } else if (n) {
            MVS_HIP(hipMemcpyAsync(ctx->csr_view.p, ctx->pre_view.p, (size_t)n * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
            MVS_HIP(hipMemcpyAsync(ctx->csr_q.p, ctx->pre_q.p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(sort_columns_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, ctx->csr_ptr.p, nf, ctx->csr_view.p, ctx->csr_q.p);
        MVS_LAUNCH_CHECK();
    }
    ctx->csr_nnz = nnz; ctx->csr_faces = nf; ctx->csr_views = n_views;
    ctx->max_q.ensure(4);
    MVS_HIP(hipMemsetAsync(ctx->max_q.p, 0, 2 * sizeof(float), s));
    if (nnz) { hipLaunchKernelGGL(max_kernel, dim3(1024), dim3(256), 0, s, ctx->csr_q.p, (size_t)nnz, (uint32_t*)ctx->max_q.p); MVS_LAUNCH_CHECK(); }
    ctx->dc_phase = 1;
}

static void launch_outlier(mvs_ctx* ctx, const uint32_t* ptr, const uint32_t* counts, uint32_t kmax_known, uint32_t nf, const float* col, float* q, uint8_t* inl, int mode) {
    hipStream_t s = ctx->stream;
    if (!nf) return;
    constexpr uint32_t LDS_PER_ENTRY = 64 * (3 * sizeof(float) + 1);   // 64 faces x (three floats + the inlier flag)
    uint32_t kmax = kmax_known;
    if (counts) {
        ctx->counters.ensure(64);
        uint32_t* d = reinterpret_cast<uint32_t*>(ctx->counters.p + 48);
        MVS_HIP(hipMemsetAsync(d, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(max_u32_kernel, dim3(std::min<unsigned>((nf + 255) / 256, 1024u)), dim3(256), 0, s, counts, nf, d); MVS_LAUNCH_CHECK();
        kmax = read_u32(ctx, d);
    }
    const uint64_t lds = (uint64_t)std::max<uint32_t>(kmax, 1u) * LDS_PER_ENTRY;
    if (lds <= 64u * 1024u) hipLaunchKernelGGL(outlier_kernel<true>, dim3((nf + 63) / 64), dim3(64), (size_t)lds, s, ptr, nf, col, q, inl, mode, std::max<uint32_t>(kmax, 1u));
    else hipLaunchKernelGGL(outlier_kernel<false>, dim3((nf + 63) / 64), dim3(64), 0, s, ptr, nf, col, q, inl, mode, 0u);
    MVS_LAUNCH_CHECK();
}

// phase 2: histogram of the local qualities against the (possibly all-reduced) maximum (:283-286)
void dc_phase2(mvs_ctx* ctx) {
    if (ctx->dc_phase != 1) throw StatusError(MVS_ERR_STATE, "dc_phase2 needs dc_phase1");
    hipStream_t s = ctx->stream;
    Prof pr(ctx, "dc_post");
    ctx->hist.ensure(HIST_BINS + 8);
    MVS_HIP(hipMemsetAsync(ctx->hist.p, 0, (HIST_BINS + 8) * sizeof(uint32_t), s));
    if (ctx->csr_nnz) {
        hipLaunchKernelGGL(hist_kernel, dim3(512), dim3(1024), 0, s, ctx->csr_q.p, (size_t)ctx->csr_nnz, ctx->max_q.p, ctx->hist.p);
        MVS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(set_count_kernel, dim3(1), dim3(1), 0, s, ctx->hist.p, (uint32_t)ctx->csr_nnz);
    MVS_LAUNCH_CHECK();
    ctx->dc_phase = 2;
}

// phase 3: percentile from the (possibly all-reduced) histogram, cost write (:288-298)
void dc_prune_labels(mvs_ctx* ctx, uint32_t kmax);
void dc_phase3(mvs_ctx* ctx, mvs_dc_stats* stats) {
    if (ctx->dc_phase != 2) throw StatusError(MVS_ERR_STATE, "dc_phase3 needs dc_phase2");
    hipStream_t s = ctx->stream;
    ctx->pctl.ensure(4);
    Prof pr(ctx, "dc_post");
    hipLaunchKernelGGL(percentile_kernel, dim3(1), dim3(1024), 0, s, ctx->hist.p, ctx->max_q.p, 0.995f, ctx->pctl.p, ctx->counters.p + 14);
    MVS_LAUNCH_CHECK();
    if (ctx->csr_nnz) {
        hipLaunchKernelGGL(cost_kernel, dim3(2048), dim3(256), 0, s, ctx->csr_q.p, (size_t)ctx->csr_nnz, ctx->pctl.p, ctx->csr_cost.p);
        MVS_LAUNCH_CHECK();
    }
    pr.end();
    unsigned long long hc[16];   // counters + (percentile_kernel's report) max quality and percentile: one read-back
    read_words(ctx, ctx->counters.p, hc, (uint32_t)(sizeof(hc) / 4));
    float mq, pc; { const uint32_t a = (uint32_t)hc[14], b = (uint32_t)hc[15]; memcpy(&mq, &a, 4); memcpy(&pc, &b, 4); }
    mvs_dc_stats& S = ctx->dc_stats;
    S.cull_backface = hc[C_BACK]; S.cull_angle = hc[C_ANGLE]; S.cull_outside = hc[C_OUTSIDE]; S.cull_occluded = hc[C_OCCL];
    S.cull_zero_quality = hc[C_ZEROQ]; S.rays = hc[C_RAYS]; S.ray_nodes = hc[C_RNODES]; S.ray_tris = hc[C_RTRIS] * 16ull /* leaf visits -> triangles: 16 per leaf (k_bvh.hip MVS_LEAF_T) */; S.ray_leaf_rounds = hc[13];
    S.ray_packets = hc[10]; S.ray_packets_generic = hc[11]; S.footprints_lane_group = ctx->dc_stats_deferred; S.footprints_rewalked = hc[C_REWALK];
    S.nnz = ctx->csr_nnz; S.max_quality = mq; S.percentile = pc;
    ctx->r_ptr = ctx->csr_ptr.p; ctx->r_view = ctx->csr_view.p; ctx->r_cost = ctx->csr_cost.p; ctx->csr_q_valid = true;
    ctx->have_costs = true; ctx->dc_phase = 3;
    if (ctx->max_labels > 0) { dc_prune_labels(ctx, (uint32_t)ctx->max_labels); S.nnz = ctx->csr_nnz; }   // label-space compression (option, off by default)
    if (stats) *stats = S;
}

// prunes the active table of the context in place (see prune_write_kernel); the table must be the context's own
void dc_prune_labels(mvs_ctx* ctx, uint32_t kmax) {
    if (!ctx->have_costs) throw StatusError(MVS_ERR_STATE, "no data costs on the device");
    if (kmax == 0) return;
    if (ctx->r_ptr != ctx->csr_ptr.p) throw StatusError(MVS_ERR_STATE, "label pruning works on the context's own table (not on caller-owned device arrays)");
    hipStream_t s = ctx->stream;
    Prof pr(ctx, "dc_prune");
    const uint32_t nf = ctx->csr_faces;
    const bool have_q = ctx->csr_q_valid;
    ctx->face_cnt.ensure((size_t)nf + 2); ctx->pre_ptr.ensure((size_t)nf + 2);
    hipLaunchKernelGGL(prune_count_kernel, dim3((nf + 256) / 256), dim3(256), 0, s, ctx->csr_ptr.p, nf, kmax, ctx->face_cnt.p); MVS_LAUNCH_CHECK();
    exclusive_scan_u32(ctx, ctx->face_cnt.p, ctx->pre_ptr.p, (size_t)nf + 1, nullptr);
    const uint32_t nnz2 = read_u32(ctx, ctx->pre_ptr.p + nf);
    if (nnz2 == ctx->csr_nnz) return;                          // no column is longer than kmax
    ctx->pre_view.ensure((size_t)nnz2 + 1); ctx->pre_q.ensure((size_t)nnz2 + 1); ctx->pcol.ensure((size_t)nnz2 + 8);
    if (nf) {
        hipLaunchKernelGGL(prune_write_kernel, dim3((unsigned)(((size_t)nf * 64 + 255) / 256)), dim3(256), 0, s, ctx->csr_ptr.p, ctx->csr_view.p, ctx->csr_cost.p,
                           have_q ? ctx->csr_q.p : (const float*)nullptr, nf, kmax, ctx->pre_ptr.p, ctx->pre_view.p, ctx->pcol.p, ctx->pre_q.p);
        MVS_LAUNCH_CHECK();
    }
    ctx->csr_view.ensure((size_t)nnz2 + 1); ctx->csr_cost.ensure((size_t)nnz2 + 8); ctx->csr_q.ensure((size_t)nnz2 + 1);
    MVS_HIP(hipMemcpyAsync(ctx->csr_ptr.p, ctx->pre_ptr.p, ((size_t)nf + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    if (nnz2) {
        MVS_HIP(hipMemcpyAsync(ctx->csr_view.p, ctx->pre_view.p, (size_t)nnz2 * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
        MVS_HIP(hipMemcpyAsync(ctx->csr_cost.p, ctx->pcol.p, (size_t)nnz2 * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (have_q) MVS_HIP(hipMemcpyAsync(ctx->csr_q.p, ctx->pre_q.p, (size_t)nnz2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    ctx->csr_nnz = nnz2; ctx->dc_stats.nnz = nnz2; ctx->u_valid = false;
    ctx->r_ptr = ctx->csr_ptr.p; ctx->r_view = ctx->csr_view.p; ctx->r_cost = ctx->csr_cost.p;
}

}  // namespace mvs
