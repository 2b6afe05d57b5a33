// dmath_host.cpp -- TEST SHIM: compiles dmath.h (the arithmetic the HIP kernels
// execute per (face, view) pair) as plain host C++ so that tests/ can compare
// each function with the oracle on a machine without a GPU.  Not a code path of
// the product: nothing in the library or the Python package calls into this file.
#include "dmath.h"
#include <vector>
#include <cstring>

using namespace mvs;

extern "C" {

struct dmh_view {
    float pos[3], viewdir[3], K[9], w2c[16];
    int32_t width, height;
    const uint8_t* rgb;
    const uint8_t* gmi;
    const uint32_t* mask;  // bit-packed, rows padded to 32-bit words, may be null
};

static ViewParams to_vp(const dmh_view* v) {
    ViewParams p;
    memcpy(p.pos, v->pos, sizeof(p.pos)); memcpy(p.viewdir, v->viewdir, sizeof(p.viewdir));
    memcpy(p.K, v->K, sizeof(p.K)); memcpy(p.w2c, v->w2c, sizeof(p.w2c));
    p.width = v->width; p.height = v->height; p.mask_stride = (v->width + 31) / 32;
    p.rgb = v->rgb; p.gmi = v->gmi; p.mask = v->mask; p.msum = nullptr; p.msum_stride = 0;
    return p;
}

float dmh_cos_limit(void) { return host_cos_limit(); }

// reasons[f * n_views + j] = cull_pair(...) for every (face, view)
void dmh_cull_all(const dmh_view* views, uint32_t n_views, const float* verts, const uint32_t* faces, const float* normals,
                  uint32_t n_faces, float cos_limit, int8_t* reasons) {
    for (uint32_t j = 0; j < n_views; ++j) {
        const ViewParams vp = to_vp(views + j);
        for (uint32_t f = 0; f < n_faces; ++f) {
            const uint32_t* fv = faces + 3 * (size_t)f;
            const V3 v1 = {verts[3 * fv[0]], verts[3 * fv[0] + 1], verts[3 * fv[0] + 2]};
            const V3 v2 = {verts[3 * fv[1]], verts[3 * fv[1] + 1], verts[3 * fv[1] + 2]};
            const V3 v3 = {verts[3 * fv[2]], verts[3 * fv[2] + 1], verts[3 * fv[2] + 2]};
            const V3 n = {normals[3 * f], normals[3 * f + 1], normals[3 * f + 2]};
            reasons[(size_t)f * n_views + j] = (int8_t)cull_pair(vp, v1, v2, v3, n, cos_limit);
        }
    }
}

// the culls kernel's entry point (clear cases without normalisations) next to plain cull_pair, for n independent pairs:
// view pose, triangle, normal given per pair; out[2 * i] = cull_pair, out[2 * i + 1] = cull_pair_prefiltered
void dmh_cull_pairs(const dmh_view* base, uint32_t n, const float* pos, const float* viewdir, const float* tri9, const float* normal,
                    float cos_limit, int8_t* out) {
    for (uint32_t i = 0; i < n; ++i) {
        ViewParams vp = to_vp(base);
        for (int a = 0; a < 3; ++a) { vp.pos[a] = pos[3 * i + a]; vp.viewdir[a] = viewdir[3 * i + a]; }
        const float* t = tri9 + 9 * (size_t)i;
        const V3 v1 = {t[0], t[1], t[2]}, v2 = {t[3], t[4], t[5]}, v3 = {t[6], t[7], t[8]}, nr = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
        const V3 centre = ((v1 + v2) + v3) / 3.0f;
        out[2 * i] = (int8_t)cull_pair(vp, v1, v2, v3, nr, cos_limit);
        out[2 * i + 1] = (int8_t)cull_pair_prefiltered(vp, v1, v2, v3, nr, centre, cos_limit);
    }
}

// quality / YCbCr mean colour of one (face, view) pair
void dmh_face_info(const dmh_view* view, int data_term, int outlier, const float* v1, const float* v2, const float* v3,
                   float* quality, float* ycbcr) {
    const ViewParams vp = to_vp(view);
    FaceInfoOut fi;
    const V3 a = {v1[0], v1[1], v1[2]}, b = {v2[0], v2[1], v2[2]}, c = {v3[0], v3[1], v3[2]};
    if (data_term == 1) { if (outlier) face_info<1, true>(vp, a, b, c, &fi); else face_info<1, false>(vp, a, b, c, &fi); }
    else { if (outlier) face_info<0, true>(vp, a, b, c, &fi); else face_info<0, false>(vp, a, b, c, &fi); }
    *quality = fi.quality;
    if (outlier) rgb_to_ycbcr(fi.mean_color);
    ycbcr[0] = fi.mean_color[0]; ycbcr[1] = fi.mean_color[1]; ycbcr[2] = fi.mean_color[2];
}

// brute-force any-hit of the (origin -> view_pos) visibility ray over all triangles
int dmh_ray_any_hit(const float* verts, const uint32_t* faces, uint32_t n_faces, uint32_t n_verts, const float* origin, const float* view_pos) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t v = 0; v < n_verts; ++v)
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], verts[3 * (size_t)v + a]); hi[a] = fmaxf(hi[a], verts[3 * (size_t)v + a]); }
    const Ray r = make_ray(V3{origin[0], origin[1], origin[2]}, V3{view_pos[0], view_pos[1], view_pos[2]}, scene_pad(lo, hi));
    for (uint32_t f = 0; f < n_faces; ++f) {
        const uint32_t* fv = faces + 3 * (size_t)f;
        const V3 a = {verts[3 * fv[0]], verts[3 * fv[0] + 1], verts[3 * fv[0] + 2]};
        const V3 b = {verts[3 * fv[1]], verts[3 * fv[1] + 1], verts[3 * fv[1] + 2]};
        const V3 c = {verts[3 * fv[2]], verts[3 * fv[2] + 1], verts[3 * fv[2] + 2]};
        if (ray_tri(r, a, b - a, c - a)) return 1;
    }
    return 0;
}

// luminance + Sobel magnitude of a whole image with the kernels' integer arithmetic
void dmh_gradient_magnitude(const uint8_t* rgb, int w, int h, uint8_t* gmi) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            if (y == 0 || y == h - 1 || x == 0 || x == w - 1) { gmi[(size_t)y * w + x] = 0; continue; }
            int l[3][3];
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const uint8_t* p = rgb + ((size_t)(y + dy) * w + (x + dx)) * 3;
                    l[dy + 1][dx + 1] = luminance_u8(p[0], p[1], p[2]);
                }
            const int sx = (l[0][2] - l[0][0]) + 2 * (l[1][2] - l[1][0]) + (l[2][2] - l[2][0]);
            const int sy = (l[2][0] - l[0][0]) + 2 * (l[2][1] - l[0][1]) + (l[2][2] - l[0][2]);
            gmi[(size_t)y * w + x] = isqrt_clamp255(sx * sx + sy * sy);
        }
}

uint32_t dmh_hist_bin(float value, float maxv) { return hist_bin(value, maxv, 10000u); }

// Exactness certificate of the lane-group footprint sampler (dmath.h foot_sums_certified) against what it certifies: `trials` random
// footprints of 33 .. max_n pixels (u8 values from a random sub-range, as in an image region), random area; the integer-sum result
// next to the serial fp64 sums of the quotients in three orders (forward = the reference's, backward, strided).
// out[0] = certified, out[1] = trials where some serial order gives another float than the integer sum (gmi term or a colour mean),
// out[2] = ... among the CERTIFIED ones (must be 0), out[3] = trials
void dmh_foot_cert_trials(uint64_t seed, uint32_t trials, uint32_t max_n, int shift, uint64_t* out) {
    uint64_t n_cert = 0, n_mis = 0, n_bad = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : n_cert, n_mis, n_bad)
    for (uint32_t t = 0; t < trials; ++t) {
        uint64_t x = seed * 0x9E3779B97F4A7C15ull + (uint64_t)t * 0xD1B54A32D192ED03ull + 1ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        const uint32_t n = 33u + (uint32_t)(rnd() % (max_n - 32u));
        const uint32_t lo = (uint32_t)(rnd() % 200u), span = 1u + (uint32_t)(rnd() % (256u - lo));
        std::vector<uint8_t> px(4 * (size_t)n);
        for (auto& p : px) p = (uint8_t)(lo + rnd() % span);
        FootSetup s{}; s.area = 0.5f * (float)n * (0.8f + 0.4f * (float)(rnd() % 1000u) / 1000.0f);
        uint64_t S[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < n; ++k) for (int c = 0; c < 4; ++c) S[c] += px[4 * (size_t)k + c];
        const double C[4] = {(double)S[0] / 255.0, (double)S[1] / 255.0, (double)S[2] / 255.0, (double)S[3] / 255.0};
        const bool cert = foot_sums_certified<1, true>(s, n, C[0], C[1], C[2], C[3], shift);
        const float qi = (float)foot_gmi_term(C[3], n, s.area), m0 = foot_mean(C[0], n), m1 = foot_mean(C[1], n), m2 = foot_mean(C[2], n);
        bool mis = false;
        for (int order = 0; order < 3; ++order) {
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t i = order == 0 ? k : order == 1 ? n - 1 - k : (uint32_t)(((uint64_t)k * 7919u) % n);
                for (int c = 0; c < 4; ++c) a[c] += (double)px[4 * (size_t)i + c] / 255.0;
            }
            mis = mis || (float)foot_gmi_term(a[3], n, s.area) != qi || foot_mean(a[0], n) != m0 || foot_mean(a[1], n) != m1 || foot_mean(a[2], n) != m2;
        }
        n_cert += cert; n_mis += mis; n_bad += (cert && mis);
    }
    out[0] = n_cert; out[1] = n_mis; out[2] = n_bad; out[3] = trials;
}

// isqrt_clamp255 against integer arithmetic for every n in [0, n_max): the number of disagreements (must be 0)
uint64_t dmh_isqrt_mismatches(uint32_t n_max) {
    uint64_t bad = 0;
    uint32_t k = 0;                                    // k = floor(sqrt(n)), advanced incrementally
    for (uint32_t n = 0; n < n_max; ++n) {
        while ((uint64_t)(k + 1) * (k + 1) <= n) ++k;
        bad += isqrt_clamp255((int)n) != (uint8_t)(k < 255u ? k : 255u);
    }
    return bad;
}

// the integer word walk of the footprint samplers (foot_walk_gmi_words, 2 / 3 / 4 scan lines per iteration) against the plain loop over
// the same spans, on random triangles in random images whose width is NOT a multiple of four (every alignment of a span's first and
// last word occurs).  out[0] = trials with a sampled `fast` footprint, out[1] = disagreements in (pixel count, sum) -- must be 0,
// out[2] = pixels walked
void dmh_word_walk_trials(uint64_t seed, uint32_t trials, uint64_t* out) {
    uint64_t n_fast = 0, n_bad = 0, n_px = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : n_fast, n_bad, n_px)
    for (uint32_t t = 0; t < trials; ++t) {
        uint64_t x = seed * 0x9E3779B97F4A7C15ull + (uint64_t)t * 0xD1B54A32D192ED03ull + 1ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        auto unit = [&]() { return (float)(rnd() % 1000003u) / 1000003.0f; };
        const int w = 33 + (int)(rnd() % 90u), h = 20 + (int)(rnd() % 60u);
        std::vector<uint8_t> buf((size_t)w * h + 16);
        for (auto& b : buf) b = (uint8_t)(rnd() >> 11);
        ViewParams vp; memset(&vp, 0, sizeof(vp));
        vp.width = w; vp.height = h; vp.gmi = buf.data() + 8;            // readable a few bytes before and after, like the context's buffer
        FootSetup s{};
        const float cx = 2.0f + unit() * (float)(w - 4), cy = 2.0f + unit() * (float)(h - 4), r = 0.6f + unit() * unit() * 30.0f;
        auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
        V2* p[3] = {&s.p1, &s.p2, &s.p3};
        for (int k = 0; k < 3; ++k) { p[k]->x = clampf(cx + (2.0f * unit() - 1.0f) * r, 0.0f, (float)w - 1.01f); p[k]->y = clampf(cy + (2.0f * unit() - 1.0f) * r, 0.0f, (float)h - 1.01f); }
        foot_setup_px(s);
        if (s.area <= 0.5f) continue;
        foot_edges(s);
        if (!s.fast) continue;
        uint32_t n0 = 0, g0 = 0;
        const float y_end = ceilf(s.aabb_max_y);
        for (int y = (int)floorf(s.aabb_min_y); (float)y < y_end; ++y) {
            int xb, xe;
            if (!foot_row(s, y, &xb, &xe)) continue;
            for (int xx = xb; xx < xe; ++xx) { g0 += vp.gmi[(size_t)xx + (size_t)y * w]; ++n0; }
        }
        uint32_t n[3], g[3];
        foot_walk_gmi_words<2>(vp, s, &n[0], &g[0]); foot_walk_gmi_words<3>(vp, s, &n[1], &g[1]); foot_walk_gmi_words<4>(vp, s, &n[2], &g[2]);
        ++n_fast; n_px += n0;
        for (int k = 0; k < 3; ++k) n_bad += (n[k] != n0 || g[k] != g0);
    }
    out[0] = n_fast; out[1] = n_bad; out[2] = n_px;
}

}  // extern "C"
