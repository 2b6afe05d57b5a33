// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h header comment).
//
// PARITY UNPINNED for the rows whose arithmetic lives in MVE / rayint / Eigen / mapMAP: the reference has no
// tests/fixtures and its hot path cannot be built here.  PINNED against the reference's own
// code where its sources compile without those libraries -- TextureView's masks, valid_pixel and get_face_info (rows B2, B3, C),
// Tri (row C), Histogram percentile (row D2), SparseTable / .spt (row E), UniGraph
// lists and get_subgraphs (rows G, f3), Settings defaults: oracle/_ref (Makefile target `ref`) compiles those sources
// from /root/reference and tests/test_reference_pins.py compares.  Every function cites the reference lines it
// restates.  Compile with -O2 -ffp-contract=off -fno-fast-math so that the
// float operation order written here is the order executed.
//
// Conventions restated from the absent MVE math library (recollection,
// "MVE-compatible by recollection" -- DEFINED HERE):
//   * Vector dot / Matrix*Vector = std::inner_product from T(0), left to right;
//   * Vector / scalar divides each component (no reciprocal);
//   * Matrix4f::mult(v, w): row . v (3 terms, left to right) then += w * m[row][3].
#include "oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct V3 { float x, y, z; };
struct V2 { float x, y; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { return a / norm(a); }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 load3(const float* p) { return {p[0], p[1], p[2]}; }

// ---------------------------------------------------------------------------
// TextureView::get_pixel_coords  (texture_view.h:161-166)
//   pixel = projection * world_to_cam.mult(vertex, 1.0f); pixel /= pixel[2];
//   return (pixel[0] - 0.5f, pixel[1] - 0.5f)
inline V2 pixel_coords(const orc_view& v, V3 p) {
    const float* m = v.w2c;
    float c[3];
    for (int i = 0; i < 3; ++i)
        c[i] = ((m[4 * i] * p.x + m[4 * i + 1] * p.y) + m[4 * i + 2] * p.z) + 1.0f * m[4 * i + 3];
    const float* k = v.K;
    float q[3];
    for (int i = 0; i < 3; ++i)
        q[i] = (k[3 * i] * c[0] + k[3 * i + 1] * c[1]) + k[3 * i + 2] * c[2];
    return {q[0] / q[2] - 0.5f, q[1] / q[2] - 0.5f};
}

// TextureView::valid_pixel  (texture_view.cpp:253-281); mask may be null
inline bool valid_pixel(const orc_view& v, const uint8_t* mask, V2 px) {
    const int width = v.width, height = v.height;
    const float x = px.x, y = px.y;
    bool valid = (x >= 0.0f && x < static_cast<float>(width - 1) &&
                  y >= 0.0f && y < static_cast<float>(height - 1));
    if (valid && mask) {
        float cx = std::max(0.0f, std::min(static_cast<float>(width - 1), x));
        float cy = std::max(0.0f, std::min(static_cast<float>(height - 1), y));
        int const floor_x = static_cast<int>(cx);
        int const floor_y = static_cast<int>(cy);
        int const floor_xp1 = std::min(floor_x + 1, width - 1);
        int const floor_yp1 = std::min(floor_y + 1, height - 1);
        valid = mask[floor_x + floor_y * width] && mask[floor_x + floor_yp1 * width] &&
                mask[floor_xp1 + floor_y * width] && mask[floor_xp1 + floor_yp1 * width];
    }
    return valid;
}

// mve::Image<uint8_t>::linear_at (MVE, absent; recollection -- DEFINED HERE):
// clamp to the image, bilinear weights w0*w2, w1*w2, w0*w3, w1*w3, and
// math::interpolate<unsigned char> rounds with +0.5f.
inline uint8_t linear_at(const uint8_t* img, int w, int h, int chans, float x, float y, int c) {
    x = std::max(0.0f, std::min(static_cast<float>(w - 1), x));
    y = std::max(0.0f, std::min(static_cast<float>(h - 1), y));
    int const floor_x = static_cast<int>(x);
    int const floor_y = static_cast<int>(y);
    int const floor_xp1 = std::min(floor_x + 1, w - 1);
    int const floor_yp1 = std::min(floor_y + 1, h - 1);
    float const w1 = x - static_cast<float>(floor_x);
    float const w0 = 1.0f - w1;
    float const w3 = y - static_cast<float>(floor_y);
    float const w2 = 1.0f - w3;
    int const rowstride = w * chans;
    int const row1 = floor_y * rowstride, row2 = floor_yp1 * rowstride;
    int const col1 = floor_x * chans, col2 = floor_xp1 * chans;
    float const v1 = img[row1 + col1 + c], v2 = img[row1 + col2 + c];
    float const v3 = img[row2 + col1 + c], v4 = img[row2 + col2 + c];
    return static_cast<uint8_t>(((v1 * (w0 * w2) + v2 * (w1 * w2)) + v3 * (w0 * w3)) + v4 * (w1 * w3) + 0.5f);
}

struct FaceInfo {  // FaceProjectionInfo (texture_view.h:26-34)
    uint16_t view_id;
    float quality;
    float mean_color[3];
};

// Tri (tri.cpp:12-24, tri.h:58-84) -- pinned against the reference's own tri.{h,cpp} through oracle/_ref
// (tests/test_reference_pins.py): constructor (detT, aabb from the unsorted points), get_area, inside.
struct TriR { float detT, min_x, min_y, max_x, max_y; };
inline TriR tri_make(V2 t1, V2 t2, V2 t3) {
    TriR r;
    const float T0 = t1.x - t3.x, T1 = t2.x - t3.x, T2 = t1.y - t3.y, T3 = t2.y - t3.y;
    r.detT = T0 * T3 - T2 * T1;
    r.min_x = std::min(t1.x, std::min(t2.x, t3.x)); r.min_y = std::min(t1.y, std::min(t2.y, t3.y));
    r.max_x = std::max(t1.x, std::max(t2.x, t3.x)); r.max_y = std::max(t1.y, std::max(t2.y, t3.y));
    return r;
}
inline float tri_area(V2 t1, V2 t2, V2 t3) {
    const float ux = t2.x - t1.x, uy = t2.y - t1.y, vx = t3.x - t1.x, vy = t3.y - t1.y;
    return 0.5f * std::abs(ux * vy - uy * vx);
}
inline bool tri_inside(V2 t1, V2 t2, V2 t3, float detT, float x, float y) {
    float const dx = (x - t3.x), dy = (y - t3.y);
    float const alpha = ((t2.y - t3.y) * dx + (t3.x - t2.x) * dy) / detT;
    if (alpha < 0.0f || alpha > 1.0f) return false;
    float const beta = ((t3.y - t1.y) * dx + (t1.x - t3.x) * dy) / detT;
    if (beta < 0.0f || beta > 1.0f) return false;
    if (alpha + beta > 1.0f) return false;
    return true;
}

// TextureView::get_face_info  (texture_view.cpp:134-251) with Tri (tri.cpp:12-24, tri.h:58-84)
void get_face_info(const orc_view& view, const uint8_t* gmi_img, V3 v1, V3 v2, V3 v3,
                   const orc_settings& st, FaceInfo* info) {
    V2 p1 = pixel_coords(view, v1), p2 = pixel_coords(view, v2), p3 = pixel_coords(view, v3);
    // Tri ctor (tri.cpp:12-24): detT and aabb from the UNSORTED points; Tri::get_area (tri.h:79-84)
    const V2 t1 = p1, t2 = p2, t3 = p3;
    const TriR tri = tri_make(t1, t2, t3);
    const float detT = tri.detT;
    const float aabb_min_x = tri.min_x, aabb_min_y = tri.min_y, aabb_max_x = tri.max_x, aabb_max_y = tri.max_y;
    const float area = tri_area(t1, t2, t3);

    if (area < std::numeric_limits<float>::epsilon()) { info->quality = 0.0f; return; }

    std::size_t num_samples = 0;
    double colors[3] = {0.0, 0.0, 0.0};
    double gmi = 0.0;
    const int w = view.width;
    const uint8_t* image = view.rgb;
    const bool sampling_necessary = st.data_term != 0 || st.outlier_removal != 0;

    if (sampling_necessary && area > 0.5f) {
        /* Sort pixels in ascending order of y (texture_view.cpp:163-167) */
        while (true)
            if (p1.y <= p2.y)
                if (p2.y <= p3.y) break;
                else std::swap(p2, p3);
            else std::swap(p1, p2);
        float const m1 = (p1.y - p3.y) / (p1.x - p3.x);
        float const b1 = p1.y - m1 * p1.x;
        float const m2 = (p1.y - p2.y) / (p1.x - p2.x);
        float const b2 = p1.y - m2 * p1.x;
        float const m3 = (p2.y - p3.y) / (p2.x - p3.x);
        float const b3 = p2.y - m3 * p2.x;
        bool fast_sampling_possible = std::isfinite(m1) && m2 != 0.0f && std::isfinite(m2) &&
                                      m3 != 0.0f && std::isfinite(m3);
        for (int y = std::floor(aabb_min_y); y < std::ceil(aabb_max_y); ++y) {
            float min_x = aabb_min_x - 0.5f;
            float max_x = aabb_max_x + 0.5f;
            if (fast_sampling_possible) {
                float const cy = static_cast<float>(y) + 0.5f;
                min_x = (cy - b1) / m1;
                if (cy <= p2.y) max_x = (cy - b2) / m2;
                else max_x = (cy - b3) / m3;
                if (min_x >= max_x) std::swap(min_x, max_x);
                if (min_x < aabb_min_x || min_x > aabb_max_x) continue;
                if (max_x < aabb_min_x || max_x > aabb_max_x) continue;
            }
            for (int x = std::floor(min_x + 0.5f); x < std::ceil(max_x - 0.5f); ++x) {
                const float cx = static_cast<float>(x) + 0.5f;
                const float cy = static_cast<float>(y) + 0.5f;
                if (!fast_sampling_possible) {
                    if (!tri_inside(t1, t2, t3, detT, cx, cy)) continue;   /* Tri::inside (tri.h:58-77) */
                }
                if (st.outlier_removal != 0) {
                    for (int i = 0; i < 3; i++)
                        colors[i] += static_cast<double>(image[(x + y * w) * 3 + i]) / 255.0;
                }
                if (st.data_term == 1) {
                    gmi += static_cast<double>(gmi_img[x + y * w]) / 255.0;
                }
                ++num_samples;
            }
        }
    }

    const int h = view.height;
    if (st.data_term == 1) {
        if (num_samples > 0) {
            gmi = (gmi / num_samples) * area;
        } else {
            double gmv1 = static_cast<double>(linear_at(gmi_img, w, h, 1, p1.x, p1.y, 0)) / 255.0;
            double gmv2 = static_cast<double>(linear_at(gmi_img, w, h, 1, p2.x, p2.y, 0)) / 255.0;
            double gmv3 = static_cast<double>(linear_at(gmi_img, w, h, 1, p3.x, p3.y, 0)) / 255.0;
            gmi = ((gmv1 + gmv2 + gmv3) / 3.0) * area;
        }
    }
    if (st.outlier_removal != 0) {
        if (num_samples > 0) {
            for (int i = 0; i < 3; ++i) info->mean_color[i] = static_cast<float>(colors[i] / num_samples);
        } else {
            for (int i = 0; i < 3; ++i) {
                double c1 = static_cast<double>(linear_at(image, w, h, 3, p1.x, p1.y, i)) / 255.0;
                double c2 = static_cast<double>(linear_at(image, w, h, 3, p2.x, p2.y, i)) / 255.0;
                double c3 = static_cast<double>(linear_at(image, w, h, 3, p3.x, p3.y, i)) / 255.0;
                info->mean_color[i] = static_cast<float>(((c1 + c2) + c3) / 3.0);
            }
        }
    }
    switch (st.data_term) {
        case 0: info->quality = area; break;
        case 1: info->quality = static_cast<float>(gmi); break;
    }
}

// mve::image::color_rgb_to_ycbcr<float> (MVE, absent; recollection -- DEFINED HERE)
inline void rgb_to_ycbcr(float* v) {
    float out[3];
    out[0] = v[0] * 0.299f + v[1] * 0.587f + v[2] * 0.114f;
    out[1] = v[0] * -0.168736f + v[1] * -0.331264f + v[2] * 0.5f + 0.5f;
    out[2] = v[0] * 0.5f + v[1] * -0.418688f + v[2] * -0.081312f + 0.5f;
    v[0] = out[0]; v[1] = out[1]; v[2] = out[2];
}

// ---------------------------------------------------------------------------
// Ray / triangle any-hit.  rayint's acc::BVHTree is absent and
// the reference uses it purely as a boolean (calculate_data_costs.cpp:201-212),
// so the test is DEFINED HERE: Moeller-Trumbore in fp32 on {a, e1 = b-a, e2 = c-a},
// fixed operation order with fused multiply-adds (cross = fma(a.y, b.z, -(a.z b.y)) ...,
// dot = fma(a.z, b.z, fma(a.y, b.y, a.x b.x))), division free: with s = sign(det) the
// scaled barycentrics s u det, s v det and the scaled distance s t det are compared with
// 0, |det|, tmin |det|, tmax |det| (as ad - (us + vs) >= 0, fma(-tmin, ad, ts) >= 0,
// fma(tmax, ad, -ts) >= 0 -- no epsilon anywhere); a ray passing all of these gets
// t = (t det) / det and its computed hit point must lie in the triangle's bounding
// box grown by `pad` (pad = 1e-5 * max(scene extent, max |coordinate|) + 1e-30).
// The last clause makes any conservative box culling exact: the result is the OR
// over ALL triangles, i.e. independent of the acceleration structure (tests compare
// the BVH with the brute-force loop).
inline float fdot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
inline V3 fcross(V3 a, V3 b) {
    V3 r;
    r.x = __builtin_fmaf(a.y, b.z, -(a.z * b.y));
    r.y = __builtin_fmaf(a.z, b.x, -(a.x * b.z));
    r.z = __builtin_fmaf(a.x, b.y, -(a.y * b.x));
    return r;
}
inline bool ray_tri(V3 orig, V3 dir, float tmin, float tmax, float pad, V3 a, V3 b, V3 c) {
    const V3 e1 = b - a, e2 = c - a;
    const V3 pv = fcross(dir, e2);
    const float det = fdot(e1, pv);
    const V3 tv = orig - a;
    const float un = fdot(tv, pv);
    const V3 qv = fcross(tv, e1);
    const float vn = fdot(dir, qv);
    const float tn = fdot(e2, qv);
    const float sg = det < 0.0f ? -1.0f : 1.0f;
    const float ad = det * sg, us = un * sg, vs = vn * sg, ts = tn * sg;
    const float w = ad - (us + vs);
    const float g1 = __builtin_fmaf(-tmin, ad, ts), g2 = __builtin_fmaf(tmax, ad, -ts);
    if (!(ad > 0.0f && std::fmin(std::fmin(std::fmin(std::fmin(us, vs), w), g1), g2) >= 0.0f)) return false;
    const float t = tn / det;
    const V3 bb = a + e1, cc = a + e2;
    const float h[3] = {orig.x + t * dir.x, orig.y + t * dir.y, orig.z + t * dir.z};
    const float lo[3] = {std::fmin(a.x, std::fmin(bb.x, cc.x)), std::fmin(a.y, std::fmin(bb.y, cc.y)), std::fmin(a.z, std::fmin(bb.z, cc.z))};
    const float hi[3] = {std::fmax(a.x, std::fmax(bb.x, cc.x)), std::fmax(a.y, std::fmax(bb.y, cc.y)), std::fmax(a.z, std::fmax(bb.z, cc.z))};
    for (int k = 0; k < 3; ++k) if (!(h[k] >= lo[k] - pad && h[k] <= hi[k] + pad)) return false;
    return true;
}

}  // namespace

// A deliberately simple BVH (median split on the largest centroid axis, <= 4
// triangles per leaf).  Boxes are padded and the slab test is widened so that
// box culling is conservative w.r.t. ray_tri's rounding: tests compare it with
// the brute-force loop.
struct orc_bvh {
    struct Node { float bmin[3], bmax[3]; uint32_t left, right, first, count; };
    std::vector<Node> nodes;
    std::vector<uint32_t> tri;  // permuted triangle ids
    float pad;
};

namespace {

void bvh_build_rec(orc_bvh& b, const orc_mesh& m, std::vector<V3>& cent, uint32_t node,
                   uint32_t first, uint32_t count) {
    float bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = first; i < first + count; ++i) {
        const uint32_t t = b.tri[i];
        for (int k = 0; k < 3; ++k) {
            const float* p = m.verts + 3 * (size_t)m.faces[3 * (size_t)t + k];
            for (int a = 0; a < 3; ++a) { bmin[a] = std::min(bmin[a], p[a]); bmax[a] = std::max(bmax[a], p[a]); }
        }
        const float c[3] = {cent[t].x, cent[t].y, cent[t].z};
        for (int a = 0; a < 3; ++a) { cmin[a] = std::min(cmin[a], c[a]); cmax[a] = std::max(cmax[a], c[a]); }
    }
    for (int a = 0; a < 3; ++a) { b.nodes[node].bmin[a] = bmin[a] - 4.0f * b.pad; b.nodes[node].bmax[a] = bmax[a] + 4.0f * b.pad; }
    b.nodes[node].first = first; b.nodes[node].count = count; b.nodes[node].left = b.nodes[node].right = 0;
    if (count <= 4) return;
    int axis = 0;
    if (cmax[1] - cmin[1] > cmax[axis] - cmin[axis]) axis = 1;
    if (cmax[2] - cmin[2] > cmax[axis] - cmin[axis]) axis = 2;
    const uint32_t mid = first + count / 2;
    std::nth_element(b.tri.begin() + first, b.tri.begin() + mid, b.tri.begin() + first + count,
                     [&](uint32_t x, uint32_t y) {
                         const float cx = axis == 0 ? cent[x].x : axis == 1 ? cent[x].y : cent[x].z;
                         const float cy = axis == 0 ? cent[y].x : axis == 1 ? cent[y].y : cent[y].z;
                         return cx < cy || (cx == cy && x < y);
                     });
    const uint32_t l = (uint32_t)b.nodes.size();
    b.nodes.push_back({}); b.nodes.push_back({});
    b.nodes[node].left = l; b.nodes[node].right = l + 1; b.nodes[node].count = 0;
    bvh_build_rec(b, m, cent, l, first, mid - first);
    bvh_build_rec(b, m, cent, l + 1, mid, first + count - mid);
}

inline bool ray_box(const orc_bvh::Node& n, V3 o, V3 inv, float tmin, float tmax) {
    float t0 = tmin, t1 = tmax;
    const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; ++a) {
        const float ta = (n.bmin[a] - oo[a]) * ii[a], tb = (n.bmax[a] - oo[a]) * ii[a];
        t0 = std::fmax(t0, std::fmin(ta, tb));  // fmin/fmax drop NaN (0 * inf)
        t1 = std::fmin(t1, std::fmax(ta, tb));
    }
    return t0 <= t1 * 1.0000005f + 1e-7f;
}

struct RayCounters { uint64_t nodes = 0, tris = 0; };

// pad = 1e-5 * max(scene extent, max |coordinate|) + 1e-30 over the exact vertex min/max
float scene_pad(const orc_mesh& m) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t v = 0; v < m.n_verts; ++v)
        for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], m.verts[3 * (size_t)v + a]); hi[a] = std::fmax(hi[a], m.verts[3 * (size_t)v + a]); }
    float ext = 0.0f, mag = 0.0f;
    for (int a = 0; a < 3; ++a) {
        ext = std::fmax(ext, hi[a] - lo[a]);
        mag = std::fmax(mag, std::fmax(std::abs(lo[a]), std::abs(hi[a])));
    }
    return 1e-5f * std::fmax(ext, mag) + 1e-30f;
}

// the any-hit query itself (what the reference asks of acc::BVHTree::intersect), for a ray already set up
bool any_hit_ray(const orc_bvh* b, const orc_mesh& m, float pad, V3 origin, V3 dir, float tmin, float tmax, bool brute, RayCounters* rc) {
    auto tri_hit = [&](uint32_t t) {
        const uint32_t* f = m.faces + 3 * (size_t)t;
        return ray_tri(origin, dir, tmin, tmax, pad, load3(m.verts + 3 * (size_t)f[0]),
                       load3(m.verts + 3 * (size_t)f[1]), load3(m.verts + 3 * (size_t)f[2]));
    };
    if (brute) {
        for (uint32_t t = 0; t < m.n_faces; ++t) { if (rc) rc->tris++; if (tri_hit(t)) return true; }
        return false;
    }
    const V3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    const float bt0 = tmin * 0.999f, bt1 = tmax * 1.001f;
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const orc_bvh::Node& n = b->nodes[stack[--sp]];
        if (rc) rc->nodes++;
        if (!ray_box(n, origin, inv, bt0, bt1)) continue;
        if (n.count) {
            for (uint32_t i = n.first; i < n.first + n.count; ++i) {
                if (rc) rc->tris++;
                if (tri_hit(b->tri[i])) return true;
            }
        } else { stack[sp++] = n.left; stack[sp++] = n.right; }
    }
    return false;
}

bool any_hit(const orc_bvh* b, const orc_mesh& m, float pad, V3 origin, V3 view_pos, bool brute, RayCounters* rc) {
    /* calculate_data_costs.cpp:201-206 */
    V3 dir = view_pos - origin;
    const float tmax = norm(dir);
    const float tmin = tmax * 0.0001f;
    dir = dir / norm(dir);
    return any_hit_ray(b, m, pad, origin, dir, tmin, tmax, brute, rc);
}

// ---------------------------------------------------------------------------
// photometric_outlier_detection (calculate_data_costs.cpp:35-129).
// Eigen is absent: mean / covariance are sequential sums, FullPivLU is a plain
// full-pivoting 3x3 LU with Eigen's rank rule (|pivot| > eps*3*|maxpivot|) and
// inverse = solve(I) -- DEFINED HERE.  `infos` is in the single-thread order of
// the reference: DESCENDING view id.
struct Lu3 {
    double lu[3][3]; int p[3], q[3]; double maxpivot; int nonzero;
    explicit Lu3(const double m[3][3]) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) lu[i][j] = m[i][j];
        for (int i = 0; i < 3; ++i) { p[i] = i; q[i] = i; }
        maxpivot = 0.0; nonzero = 3;
        for (int k = 0; k < 3; ++k) {
            int br = k, bc = k; double big = -1.0;
            for (int c = k; c < 3; ++c) for (int r = k; r < 3; ++r)  // column-major, first max wins
                if (std::abs(lu[r][c]) > big) { big = std::abs(lu[r][c]); br = r; bc = c; }
            if (big == 0.0) { nonzero = k; break; }
            if (big > maxpivot) maxpivot = big;
            if (br != k) { for (int c = 0; c < 3; ++c) std::swap(lu[k][c], lu[br][c]); std::swap(p[k], p[br]); }
            if (bc != k) { for (int r = 0; r < 3; ++r) std::swap(lu[r][k], lu[r][bc]); std::swap(q[k], q[bc]); }
            for (int r = k + 1; r < 3; ++r) lu[r][k] /= lu[k][k];
            for (int r = k + 1; r < 3; ++r) for (int c = k + 1; c < 3; ++c) lu[r][c] -= lu[r][k] * lu[k][c];
        }
    }
    bool invertible() const {
        const double thr = std::abs(maxpivot) * (std::numeric_limits<double>::epsilon() * 3.0);
        int rank = 0;
        for (int i = 0; i < nonzero; ++i) rank += (std::abs(lu[i][i]) > thr);
        return rank == 3;
    }
    void inverse(double inv[3][3]) const {  // solve P A Q = L U for A x = e_j
        for (int j = 0; j < 3; ++j) {
            double y[3];
            for (int i = 0; i < 3; ++i) y[i] = (p[i] == j) ? 1.0 : 0.0;
            for (int i = 1; i < 3; ++i) for (int k = 0; k < i; ++k) y[i] -= lu[i][k] * y[k];
            for (int i = 2; i >= 0; --i) { for (int k = i + 1; k < 3; ++k) y[i] -= lu[i][k] * y[k]; y[i] /= lu[i][i]; }
            for (int i = 0; i < 3; ++i) inv[q[i]][j] = y[i];
        }
    }
};

// multi_gauss_unnormalized (util.h:60-66): exp(-0.5 * mr * Cinv * mr^T), left to right
inline double multi_gauss(const double x[3], const double mu[3], const double ci[3][3]) {
    double mr[3], w[3];
    for (int a = 0; a < 3; ++a) mr[a] = x[a] - mu[a];
    for (int b = 0; b < 3; ++b)
        w[b] = ((-0.5 * mr[0]) * ci[0][b] + (-0.5 * mr[1]) * ci[1][b]) + (-0.5 * mr[2]) * ci[2][b];
    return std::exp((w[0] * mr[0] + w[1] * mr[1]) + w[2] * mr[2]);
}

bool photometric_outlier_detection(std::vector<FaceInfo>* infos, const orc_settings& st) {
    if (infos->size() == 0) return true;
    double const gauss_rejection_threshold = 6e-3;
    double const minimal_covariance = 5e-4;
    int const outlier_detection_iterations = 10;
    int const minimal_num_inliers = 4;
    float outlier_removal_factor;
    switch (st.outlier_removal) {
        case 2: outlier_removal_factor = 1.0f; break;  // clamping
        case 1: outlier_removal_factor = 0.2f; break;  // damping
        default: return true;
    }
    const size_t n = infos->size();
    std::vector<uint32_t> is_inlier(n, 1);
    size_t n_in = n;
    double var_mean[3] = {0, 0, 0}, cov[3][3], cov_inv[3][3] = {{0}};
    for (int it = 0; it < outlier_detection_iterations; ++it) {
        if ((int)n_in < minimal_num_inliers) return false;
        for (int a = 0; a < 3; ++a) {
            double s = 0.0;
            for (size_t r = 0; r < n; ++r) if (is_inlier[r]) s += (double)(*infos)[r].mean_color[a];
            var_mean[a] = s / (double)n_in;
        }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
            double s = 0.0;
            for (size_t r = 0; r < n; ++r) if (is_inlier[r])
                s += ((double)(*infos)[r].mean_color[a] - var_mean[a]) * ((double)(*infos)[r].mean_color[b] - var_mean[b]);
            cov[a][b] = s / double(n_in - 1);
        }
        double mx = 0.0;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) mx = std::max(mx, std::abs(cov[a][b]));
        if (mx < minimal_covariance) {
            for (size_t r = 0; r < n; ++r) if (!is_inlier[r]) (*infos)[r].quality = 0.0f;
            return true;
        }
        Lu3 lu(cov);
        if (!lu.invertible()) return false;
        lu.inverse(cov_inv);
        n_in = 0;
        for (size_t r = 0; r < n; ++r) {
            const double c[3] = {(*infos)[r].mean_color[0], (*infos)[r].mean_color[1], (*infos)[r].mean_color[2]};
            double g = multi_gauss(c, var_mean, cov_inv);
            is_inlier[r] = (g >= gauss_rejection_threshold ? 1 : 0);
            n_in += is_inlier[r];
        }
    }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov_inv[a][b] *= outlier_removal_factor;
    for (FaceInfo& info : *infos) {
        const double c[3] = {info.mean_color[0], info.mean_color[1], info.mean_color[2]};
        double g = multi_gauss(c, var_mean, cov_inv);
        if (st.outlier_removal == 1) info.quality *= g;
        else if (g < gauss_rejection_threshold) info.quality = 0.0f;
    }
    return true;
}

}  // namespace

extern "C" {

// TextureView::generate_validity_mask (texture_view.cpp:42-94): flood fill from
// the 4 corners through pixels whose channel sum is 0; those become invalid.
void orc_validity_mask(const uint8_t* rgb, int w, int h, uint8_t* mask) {
    std::fill(mask, mask + (size_t)w * h, (uint8_t)1);
    std::vector<uint8_t> checked((size_t)w * h, 0);
    std::vector<std::pair<int, int>> queue;
    auto push = [&](int x, int y) {
        if (!checked[(size_t)y * w + x]) { checked[(size_t)y * w + x] = 255; queue.emplace_back(x, y); }
    };
    push(0, 0); push(0, h - 1); push(w - 1, 0); push(w - 1, h - 1);
    while (!queue.empty()) {
        auto [x, y] = queue.back(); queue.pop_back();  // visit order does not change the result
        const uint8_t* p = rgb + ((size_t)y * w + x) * 3;
        int sum = p[0] + p[1] + p[2];
        if (sum == 0) {
            mask[(size_t)y * w + x] = 0;
            const int nx[4] = {x + 1, x, x - 1, x}, ny[4] = {y, y + 1, y, y - 1};
            for (int i = 0; i < 4; ++i)
                if (0 <= nx[i] && nx[i] < w && 0 <= ny[i] && ny[i] < h) push(nx[i], ny[i]);
        }
    }
}

// TextureView::generate_gradient_magnitude (texture_view.cpp:102-107):
//   bw = desaturate<uint8_t>(image, DESATURATE_LUMINANCE); gmi = sobel_edge<uint8_t>(bw).
// MVE is absent (recollection -- DEFINED HERE): luminance = T(0.30 * r + 0.59f * g + 0.11f * b)
// (the 0.30 literal is a double in MVE's image_tools.h), truncation to u8;
// sobel_edge: 3x3 Sobel gx, gy in double, out = T(min(255, sqrt(gx^2 + gy^2))),
// border pixels 0.
void orc_gradient_magnitude(const uint8_t* rgb, int w, int h, uint8_t* gmi) {
    std::vector<uint8_t> bw((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t* v = rgb + 3 * i;
        bw[i] = static_cast<uint8_t>(0.30 * v[0] + 0.59f * v[1] + 0.11f * v[2]);
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t pos = (size_t)y * w + x;
            if (y == 0 || y == h - 1 || x == 0 || x == w - 1) { gmi[pos] = 0; continue; }
            auto at = [&](int xx, int yy) { return (double)bw[(size_t)yy * w + xx]; };
            const double gx = 1.0 * at(x + 1, y - 1) - 1.0 * at(x - 1, y - 1) + 2.0 * at(x + 1, y) -
                              2.0 * at(x - 1, y) + 1.0 * at(x + 1, y + 1) - 1.0 * at(x - 1, y + 1);
            const double gy = 1.0 * at(x - 1, y + 1) - 1.0 * at(x - 1, y - 1) + 2.0 * at(x, y + 1) -
                              2.0 * at(x, y - 1) + 1.0 * at(x + 1, y + 1) - 1.0 * at(x + 1, y - 1);
            const double g = std::sqrt(gx * gx + gy * gy);
            gmi[pos] = static_cast<uint8_t>(std::min(255.0, g));
        }
}

// TextureView::erode_validity_mask (texture_view.cpp:109-132), restated
// literally: border pixels are cleared in the OLD mask (which is swapped away),
// interior invalid pixels clear their 3x3 neighbourhood in the copy.
void orc_erode_validity_mask(uint8_t* mask, int w, int h) {
    std::vector<uint8_t> eroded(mask, mask + (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            if (x == 0 || x == w - 1 || y == 0 || y == h - 1) { mask[x + (size_t)y * w] = 0; continue; }
            if (mask[x + (size_t)y * w]) continue;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) eroded[(x + i) + (size_t)(y + j) * w] = 0;
        }
    std::copy(eroded.begin(), eroded.end(), mask);
}

orc_bvh* orc_bvh_build(const orc_mesh* mesh) {
    orc_bvh* b = new orc_bvh;
    const orc_mesh& m = *mesh;
    b->pad = scene_pad(m);
    std::vector<V3> cent(m.n_faces);
    b->tri.resize(m.n_faces);
    for (uint32_t t = 0; t < m.n_faces; ++t) {
        b->tri[t] = t;
        const uint32_t* f = m.faces + 3 * (size_t)t;
        cent[t] = (load3(m.verts + 3 * (size_t)f[0]) + load3(m.verts + 3 * (size_t)f[1]) + load3(m.verts + 3 * (size_t)f[2])) / 3.0f;
    }
    b->nodes.reserve((size_t)m.n_faces);
    b->nodes.push_back({});
    if (m.n_faces) bvh_build_rec(*b, m, cent, 0, 0, m.n_faces);
    return b;
}
void orc_bvh_free(orc_bvh* b) { delete b; }

int orc_ray_occluded(const orc_bvh* b, const orc_mesh* mesh, const float origin[3],
                     const float view_pos[3], int brute) {
    return any_hit(b, *mesh, scene_pad(*mesh), load3(origin), load3(view_pos), brute != 0, nullptr) ? 1 : 0;
}
// the bare query for a ray the CALLER set up: oracle/_ref's stand-in for acc::BVHTree::intersect forwards the reference's
// own rays here (tests/test_reference_pins.py)
int orc_ray_hit(const orc_bvh* b, const orc_mesh* mesh, const float origin[3], const float dir[3], float tmin, float tmax, int brute) {
    return any_hit_ray(b, *mesh, b ? b->pad : scene_pad(*mesh), load3(origin), load3(dir), tmin, tmax, brute != 0, nullptr) ? 1 : 0;
}
// photometric_outlier_detection (calculate_data_costs.cpp:35-129) on one face's infos, in the order given: mean_color[3n]
// (already YCbCr), quality[n] updated in place; returns the function's bool
int orc_outlier_detection(uint32_t n, const float* mean_color, float* quality, int outlier_removal) {
    std::vector<FaceInfo> infos(n);
    for (uint32_t i = 0; i < n; ++i) {
        infos[i].view_id = (uint16_t)i; infos[i].quality = quality[i];
        for (int a = 0; a < 3; ++a) infos[i].mean_color[a] = mean_color[3 * (size_t)i + a];
    }
    orc_settings st; st.data_term = 0; st.outlier_removal = outlier_removal; st.geometric_visibility_test = 0;
    const bool ok = photometric_outlier_detection(&infos, st);
    for (uint32_t i = 0; i < n; ++i) quality[i] = infos[i].quality;
    return ok ? 1 : 0;
}

// Histogram (histogram.cpp:22-63): add_value + get_approx_percentile
float orc_percentile(const float* values, uint64_t n, float max_value, float percentile) {
    const float min = 0.0f, max = max_value;
    std::vector<unsigned int> bins(10000, 0);
    int num_values = 0;
    for (uint64_t i = 0; i < n; ++i) {
        float clamped_value = std::max(min, std::min(max, values[i]));
        std::size_t index = floor(((clamped_value - min) / (max - min)) * (bins.size() - 1));
        bins[index]++;
        ++num_values;
    }
    int num = 0;
    float upper_bound = min;
    for (std::size_t i = 0; i < bins.size(); ++i) {
        if (static_cast<float>(num) / num_values > percentile) return upper_bound;
        num += bins[i];
        upper_bound = (static_cast<float>(i) / (bins.size() - 1)) * (max - min) + min;
    }
    return max;
}

// tex::calculate_data_costs (calculate_data_costs.cpp:308-323)
//   = calculate_face_projection_infos (:131-251) + postprocess_face_infos (:253-306)
int orc_data_costs(const orc_mesh* mesh, const orc_view* views, uint32_t n_views,
                   const orc_settings* settings, uint32_t face_begin, uint32_t face_end,
                   int bvh_mode, int n_threads, orc_csr* out, orc_dc_stats* stats) {
    const orc_mesh& m = *mesh;
    const orc_settings& st = *settings;
    /* :315-318 (num_faces is a uint32 here, so only the view guard can fire) */
    if (n_views > std::numeric_limits<std::uint16_t>::max()) return 2;
    if (face_end > m.n_faces) face_end = m.n_faces;
    if (face_begin > face_end) face_begin = face_end;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    orc_dc_stats S; memset(&S, 0, sizeof(S));
    const uint32_t nf = face_end - face_begin;
    S.pairs = (uint64_t)nf * n_views;

    double t0 = now_s();
    orc_bvh* bvh = nullptr;
    if (st.geometric_visibility_test && bvh_mode == 0) bvh = orc_bvh_build(mesh);  /* :144 */
    const float pad = scene_pad(m);
    S.t_bvh = now_s() - t0;

    // per-view lists of (face, info): the reference's thread-local vectors (:150,227-228)
    std::vector<std::vector<std::pair<uint32_t, FaceInfo>>> per_view(n_views);
    double t_prep = 0.0;
    uint64_t c_back = 0, c_angle = 0, c_out = 0, c_occ = 0, c_zero = 0, c_rays = 0, c_nodes = 0, c_tris = 0;
    t0 = now_s();
#pragma omp parallel for schedule(dynamic) num_threads(n_threads) \
    reduction(+ : t_prep, c_back, c_angle, c_out, c_occ, c_zero, c_rays, c_nodes, c_tris)
    for (int jj = 0; jj < (int)n_views; ++jj) {
        const uint16_t j = (uint16_t)jj;
        const orc_view& view = views[j];
        const int w = view.width, h = view.height;
        double tp = now_s();
        std::vector<uint8_t> mask((size_t)w * h), gmi;
        orc_validity_mask(view.rgb, w, h, mask.data());           /* :158 */
        if (st.data_term == 1) {                                   /* :160-163 */
            gmi.resize((size_t)w * h);
            orc_gradient_magnitude(view.rgb, w, h, gmi.data());
            orc_erode_validity_mask(mask.data(), w, h);
        }
        t_prep += now_s() - tp;
        const V3 view_pos = load3(view.pos), viewing_direction = load3(view.viewdir);
        auto& outv = per_view[j];
        RayCounters rc;
        for (uint32_t face_id = face_begin; face_id < face_end; ++face_id) {  /* :168-229 */
            const uint32_t* f = m.faces + 3 * (size_t)face_id;
            const V3 v1 = load3(m.verts + 3 * (size_t)f[0]);
            const V3 v2 = load3(m.verts + 3 * (size_t)f[1]);
            const V3 v3 = load3(m.verts + 3 * (size_t)f[2]);
            const V3 face_normal = load3(m.face_normals + 3 * (size_t)face_id);
            const V3 face_center = ((v1 + v2) + v3) / 3.0f;
            const V3 view_to_face_vec = normalized(face_center - view_pos);
            const V3 face_to_view_vec = normalized(view_pos - face_center);
            /* Backface and basic frustum culling (:183-185) */
            float viewing_angle = dot(face_to_view_vec, face_normal);
            if (viewing_angle < 0.0f || dot(viewing_direction, view_to_face_vec) < 0.0f) { c_back++; continue; }
            /* :187  std::acos(float) > MATH_DEG2RAD(75.0f) -- the macro is a double expression */
            if (std::acos(viewing_angle) > (75.0f * (3.14159265358979323846264338327950288 / 180.0))) { c_angle++; continue; }
            /* :191 */
            if (!(valid_pixel(view, mask.data(), pixel_coords(view, v1)) &&
                  valid_pixel(view, mask.data(), pixel_coords(view, v2)) &&
                  valid_pixel(view, mask.data(), pixel_coords(view, v3)))) { c_out++; continue; }
            if (st.geometric_visibility_test) {  /* :194-215 */
                bool visible = true;
                const V3* samples[] = {&v1, &v2, &v3};
                for (int k = 0; k < 3; ++k) {
                    c_rays++;
                    if (any_hit(bvh, m, pad, *samples[k], view_pos, bvh_mode != 0, &rc)) { visible = false; break; }
                }
                if (!visible) { c_occ++; continue; }
            }
            FaceInfo info = {j, 0.0f, {0.0f, 0.0f, 0.0f}};
            get_face_info(view, gmi.data(), v1, v2, v3, st, &info);  /* :220 */
            if (info.quality == 0.0) { c_zero++; continue; }          /* :222 */
            rgb_to_ycbcr(info.mean_color);                            /* :225 */
            outv.emplace_back(face_id, info);
        }
        c_nodes += rc.nodes; c_tris += rc.tris;
    }
    S.t_infos = now_s() - t0; S.t_prep = t_prep;
    S.cull_backface = c_back; S.cull_angle = c_angle; S.cull_outside = c_out; S.cull_occluded = c_occ;
    S.cull_zero_quality = c_zero; S.rays = c_rays; S.ray_nodes = c_nodes; S.ray_tris = c_tris;
    if (bvh) orc_bvh_free(bvh);

    /* :241-249 with one thread: views ascending, each appended in reverse => per face DESCENDING view id */
    t0 = now_s();
    std::vector<std::vector<FaceInfo>> infos(nf);
    for (int j = (int)n_views - 1; j >= 0; --j) {
        for (auto& pr : per_view[j]) infos[pr.first - face_begin].push_back(pr.second);
        S.nnz_pre += per_view[j].size();
        std::vector<std::pair<uint32_t, FaceInfo>>().swap(per_view[j]);
    }
    /* postprocess_face_infos (:253-306) */
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads)
    for (int64_t i = 0; i < (int64_t)nf; ++i) {
        std::vector<FaceInfo>& fi = infos[i];
        if (st.outlier_removal != 0) {
            photometric_outlier_detection(&fi, st);
            fi.erase(std::remove_if(fi.begin(), fi.end(), [](FaceInfo const& x) { return x.quality == 0.0f; }), fi.end());
        }
        std::sort(fi.begin(), fi.end(), [](FaceInfo const& a, FaceInfo const& b) { return a.view_id < b.view_id; });
    }
    float max_quality = 0.0f;
    uint64_t nnz = 0;
    for (auto& fi : infos) { nnz += fi.size(); for (auto& x : fi) max_quality = std::max(max_quality, x.quality); }
    std::vector<float> allq; allq.reserve(nnz);
    for (auto& fi : infos) for (auto& x : fi) allq.push_back(x.quality);
    float percentile = orc_percentile(allq.data(), nnz, max_quality, 0.995f);
    S.max_quality = max_quality; S.percentile = percentile;

    out->n_faces = nf; out->n_views = n_views; out->nnz = nnz;
    out->col_ptr = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)nf + 1));
    out->view_id = (uint16_t*)malloc(sizeof(uint16_t) * std::max<uint64_t>(nnz, 1));
    out->cost = (float*)malloc(sizeof(float) * std::max<uint64_t>(nnz, 1));
    out->quality = (float*)malloc(sizeof(float) * std::max<uint64_t>(nnz, 1));
    uint64_t k = 0;
    for (uint32_t i = 0; i < nf; ++i) {
        out->col_ptr[i] = (uint32_t)k;
        for (auto& x : infos[i]) {
            /* Clamp to percentile and normalize (:295-297) */
            float normalized_quality = std::min(1.0f, x.quality / percentile);
            float data_cost = (1.0f - normalized_quality);
            out->view_id[k] = x.view_id; out->cost[k] = data_cost; out->quality[k] = x.quality; ++k;
        }
    }
    out->col_ptr[nf] = (uint32_t)k;
    S.t_post = now_s() - t0;
    if (stats) *stats = S;
    return 0;
}

// Label-space compression (NOT in the reference; the option of the same name of the product, include/mvs_viewsel.h
// mvs_ctx_prune_labels): per face the kmax entries with the smallest (cost, view id) pairs, in ascending view order.
void orc_prune_labels(const orc_csr* in, uint32_t kmax, orc_csr* out) {
    const uint32_t nf = in->n_faces;
    out->n_faces = nf; out->n_views = in->n_views;
    out->col_ptr = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)nf + 1));
    uint64_t n = 0;
    for (uint32_t i = 0; i < nf; ++i) { out->col_ptr[i] = (uint32_t)n; n += std::min(in->col_ptr[i + 1] - in->col_ptr[i], kmax); }
    out->col_ptr[nf] = (uint32_t)n; out->nnz = n;
    out->view_id = (uint16_t*)malloc(sizeof(uint16_t) * std::max<uint64_t>(n, 1));
    out->cost = (float*)malloc(sizeof(float) * std::max<uint64_t>(n, 1));
    out->quality = (float*)malloc(sizeof(float) * std::max<uint64_t>(n, 1));
    std::vector<uint32_t> order;
    for (uint32_t i = 0; i < nf; ++i) {
        const uint32_t p0 = in->col_ptr[i], K = in->col_ptr[i + 1] - p0;
        order.resize(K);
        for (uint32_t t = 0; t < K; ++t) order[t] = t;
        if (K > kmax) {
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return in->cost[p0 + a] != in->cost[p0 + b] ? in->cost[p0 + a] < in->cost[p0 + b] : a < b; });
            order.resize(kmax);
            std::sort(order.begin(), order.end());
        }
        uint32_t d = out->col_ptr[i];
        for (uint32_t t : order) { out->view_id[d] = in->view_id[p0 + t]; out->cost[d] = in->cost[p0 + t]; out->quality[d] = in->quality ? in->quality[p0 + t] : 0.0f; ++d; }
    }
}

// Row f4: mve::image::image_undistort_k2k4 / image_undistort_vsfm as generate_texture_views.cpp:153-165 applies them
// (dist0 != 0: k2k4 if dist1 != 0, else vsfm).  MVE is absent -- DEFINED HERE from recollection of mve/image_tools.h:
// output pixel (x, y) -> centred, divided by max(w, h); rsq over flen^2; k2k4 factor 1 + rsq k2 + rsq^2 k4; vsfm inverts
// r_u = r_d (1 + k1 r_d^2) with 8 Newton steps in fp64 from r_d = r_u; source positions beyond half a pixel outside stay
// black, others are clamped and sampled with linear_at.
void orc_undistort(const uint8_t* rgb, int w, int h, float flen_f, float dist0, float dist1, uint8_t* out) {
    const size_t bytes = (size_t)w * h * 3;
    if (dist0 == 0.0f) { memcpy(out, rgb, bytes); return; }
    const double flen = flen_f, d0 = dist0, d1 = dist1;
    const double width_half = (double)w / 2.0, height_half = (double)h / 2.0, norm = (double)std::max(w, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double fx = ((double)x - width_half) / norm, fy = ((double)y - height_half) / norm;
            double factor;
            bool ok = true;
            if (d1 != 0.0) {
                const double rsq = (fx * fx + fy * fy) / (flen * flen);
                factor = 1.0 + rsq * d0 + (rsq * rsq) * d1;
            } else {
                const double ru = std::sqrt(fx * fx + fy * fy) / flen;
                double rd = ru;
                // guarded Newton (see k_prep.hip undistort_kernel): no safely positive derivative, or no convergence -> no source pixel
                for (int it = 0; it < 8 && ok; ++it) {
                    const double den = (3.0 * d0) * (rd * rd) + 1.0;
                    if (!(den > 1e-3)) ok = false; else rd = rd - (((d0 * rd) * rd) * rd + rd - ru) / den;
                }
                if (ok) { const double res = ((d0 * rd) * rd) * rd + rd - ru; ok = (res < 0.0 ? -res : res) <= 1e-9 * (1.0 + ru); }
                factor = ru > 0.0 ? rd / ru : 1.0;
            }
            fx = (fx * factor) * norm + width_half;
            fy = (fy * factor) * norm + height_half;
            uint8_t* o = out + ((size_t)y * w + x) * 3;
            if (!ok) { o[0] = o[1] = o[2] = 0; continue; }
            if (!(fx >= -0.5 && fx <= (double)w - 0.5 && fy >= -0.5 && fy <= (double)h - 0.5)) { o[0] = o[1] = o[2] = 0; continue; }
            fx = std::max(0.0, std::min((double)w - 1.0, fx));
            fy = std::max(0.0, std::min((double)h - 1.0, fy));
            for (int c = 0; c < 3; ++c) o[c] = linear_at(rgb, w, h, 3, (float)fx, (float)fy, c);
        }
}

void orc_csr_free(orc_csr* c) {
    free(c->col_ptr); free(c->view_id); free(c->cost); free(c->quality);
    memset(c, 0, sizeof(*c));
}

}  // extern "C"

// ===========================================================================
// View selection  (view_selection.cpp:18-133)
//
// Model construction is restated from the reference (:27-82): node i has the
// label set {view_id + 1 : view_id in column(i)} with unaries = column costs,
// or the single label 0 (cost 1.0) when its column is empty; an edge (i, j)
// of weight 1 (Potts) exists iff j is adjacent to i and BOTH columns are
// non-empty.  Energy: E(l) = sum_i D_i(l_i) + sum_{(i,j)} [l_i != l_j].
//
// The SOLVER is DEFINED HERE (mapMAP fa526e0 is absent -- SURVEY.md 0.2, 7):
// tree-reweighted max-product message passing over the face adjacency graph with a
// colour-phased Gauss-Seidel schedule (below), followed by a monotone ICM polish.  Every quantity is
// specified down to the float operation order so that the HIP implementation
// can be bit-exact at any number of GPUs / partitions:
//
//  * Directed edge e = (i <- j), j = adj[e], valid iff both columns non-empty.
//    Message m_e has K_i entries aligned with i's label list; moff[e] is the
//    exclusive prefix sum of (valid ? K_i : 0) in adjacency-CSR order;
//    map[moff[e] + t] = position of L_i[t] inside L_j, or 0xFFFF.  m = 0 at start.
//  * Schedule: the adjacency graph is coloured greedily in the order of the keys (hash32(i), i) (mrf_colour); one
//    sweep = for colour 0, 1, ...: every node of that colour (an independent set) updates IN PLACE from the
//    current messages.  Messages are STORED as 8-bit codes over their range [0, lam], lam = 1 / rho (a message is a
//    truncated, min-normalised cavity): value = code * step, step = lam / 255, scale = 255 / lam.  The update of node i,
//    written on the codes (mrf_sweep below is the definition, fma = IEEE fused multiply-add):
//      Sc[t] = sum over valid e of code_e[t]                 (exact)
//      b[t]  = fma(rho * step, Sc[t], D[t]);  sel_i = first argmin_t b[t]      D = the 16-bit fixed-point unaries
//      for each valid e (to j):  c[t] = fma(-step, code_e[t], b[t]);  cmin = min_t c[t]
//        ( = D + rho * sum_all m - m_e = (D + rho * sum_others m) - (1 - rho) * m_e, the tree-reweighted cavity )
//        for t' < K_j:  p = map[moff[rev e] + t']
//          ( c and lam taken in damped code units: cs = c * oms, lam_s = lam * oms, oms = (1 - alpha) * scale )
//          raw   = (p == NONE) ? lam_s : fminf(cs[p] - cmin_s, lam_s)
//          code' = rne( fma(old code, alpha, raw) ), saturated at 255
//        with alpha = damping on every fourth sweep (1st, 5th, ...: MRF_DAMP_PERIOD) and 0 on the others.
//  * After each sweep the decoded labeling's energy is evaluated exactly in
//    32.32 fixed point (integer sums are order independent); the best labeling
//    so far is kept.  Stop like StopWhenReturnsDiminish(5, 0.01)
//    (view_selection.cpp:84): when the best energy improved by less than min_improvement (default 0.2 %)
//    over the last `window` (default 5) sweeps (and at least min_sweeps ran), or at max_sweeps.
//  * ICM polish: every node computes its best label given the neighbours'
//    labels and its gain; a node moves iff gain > 0 and (gain, -index) beats
//    all its neighbours' (an independent set => energy strictly decreases).
// ===========================================================================
namespace {

const uint16_t MAP_NONE = 0xFFFF;
uint64_t* g_trace = nullptr; int g_trace_len = 0;
// optional per-sweep energy trace (experiments)

inline uint64_t fix32(float d) { return (uint64_t)((double)d * 4294967296.0); }

// Messages are STORED as 8-bit fixed point over their range [0, lam], lam = 1 / rho (a message is a truncated,
// min-normalised cavity, so 0 <= m <= lam by construction): value = code * step, step = lam / 255, scale = 255 / lam
// (both evaluated in fp32).  Part of the solver's definition: the labelings reach the same energy
// as with binary16 or fp32 messages at half / a quarter of the bytes.
struct MsgQ { float scale, step; };
inline MsgQ msg_q(float lam) { return MsgQ{255.0f / lam, lam / 255.0f}; }
// the code a message is stored as: `raw_s` = its value in damped code units, (1 - alpha) * scale * value, in
// [0, (1 - alpha) * 255]; damped against the old code (alpha = 0: undamped)
inline uint32_t msg_code(float raw_s, float alpha, uint32_t old_code) {
    const long code = lrintf(__builtin_fmaf((float)old_code, alpha, raw_s));   // round to nearest even
    return (uint32_t)(code > 255 ? 255 : code);
}

struct Mrf {
    uint32_t F = 0;
    const uint32_t* col_ptr = nullptr; const uint16_t* view_id = nullptr; const float* cost = nullptr;
    const float* qcost = nullptr;   // the unaries as the sweeps see them (16-bit fixed point, below)
    const uint32_t* adj_ptr = nullptr; const uint32_t* adj = nullptr;
    std::vector<uint8_t> valid;     // per directed edge
    std::vector<uint32_t> rev;      // per directed edge
    std::vector<uint64_t> moff;     // per directed edge (+1)
    std::vector<uint16_t> map;      // per message entry
    uint32_t K(uint32_t i) const { return col_ptr[i + 1] - col_ptr[i]; }
};

void mrf_setup(Mrf& g) {
    const uint32_t F = g.F;
    const uint32_t E = g.adj_ptr[F];
    g.valid.assign(E, 0); g.rev.assign(E, 0); g.moff.assign((size_t)E + 1, 0);
    uint64_t off = 0;
    for (uint32_t i = 0; i < F; ++i)
        for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1]; ++e) {
            const uint32_t j = g.adj[e];
            g.valid[e] = (g.K(i) > 0 && g.K(j) > 0) ? 1 : 0;
            uint32_t r = g.adj_ptr[j];
            while (r < g.adj_ptr[j + 1] && g.adj[r] != i) ++r;
            g.rev[e] = r;
            g.moff[e] = off;
            if (g.valid[e]) off += g.K(i);
        }
    g.moff[E] = off;
    g.map.assign(off, MAP_NONE);
    for (uint32_t i = 0; i < F; ++i)
        for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1]; ++e) {
            if (!g.valid[e]) continue;
            const uint32_t j = g.adj[e];
            const uint16_t* Li = g.view_id + g.col_ptr[i]; const uint16_t* Lj = g.view_id + g.col_ptr[j];
            const uint32_t Ki = g.K(i), Kj = g.K(j);
            uint32_t q = 0;
            for (uint32_t t = 0; t < Ki; ++t) {  // both lists ascending (calculate_data_costs.cpp:272)
                while (q < Kj && Lj[q] < Li[t]) ++q;
                if (q < Kj && Lj[q] == Li[t]) g.map[g.moff[e] + t] = (uint16_t)q;
            }
        }
}

// exact energy in 32.32 fixed point (qcode == nullptr), or the solver's TRACKING energy: the same sum over the 16-bit
// unaries the sweeps see, in units of 1 / 65535 (sum of cost codes + 65535 per cut edge) -- pure integer arithmetic
uint64_t mrf_energy_sel(const Mrf& g, const std::vector<uint32_t>& sel, uint64_t* cuts_out, int n_threads = 1, const uint16_t* qcode = nullptr) {
    const float* cost = g.cost;
    uint64_t unary = 0, cuts = 0;
#pragma omp parallel for schedule(static) num_threads(n_threads) reduction(+ : unary, cuts)
    for (int64_t ii = 0; ii < (int64_t)g.F; ++ii) {
        const uint32_t i = (uint32_t)ii;
        if (g.K(i) == 0) { unary += qcode ? 65535u : fix32(1.0f); continue; }   /* view_selection.cpp:70-71 */
        unary += qcode ? (uint64_t)qcode[g.col_ptr[i] + sel[i]] : fix32(cost[g.col_ptr[i] + sel[i]]);
        const uint16_t li = g.view_id[g.col_ptr[i] + sel[i]];
        for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1]; ++e) {
            const uint32_t j = g.adj[e];
            if (!g.valid[e] || j <= i) continue;                /* :38 uni directional, i < adj_face */
            if (g.view_id[g.col_ptr[j] + sel[j]] != li) ++cuts;
        }
    }
    if (cuts_out) *cuts_out = cuts;
    return qcode ? unary + 65535ull * cuts : unary + (cuts << 32);
}

// Greedy colouring of the adjacency graph in the order of the keys (hash32(i), i): node i takes the smallest
// colour no earlier neighbour holds.  DEFINED HERE (part of the solver's schedule): the GPU reaches the same
// colouring with Jones-Plassmann rounds (a node colours itself once all neighbours with a smaller key have).
inline uint32_t mrf_hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
int mrf_colour(const Mrf& g, std::vector<uint8_t>& colour) {
    std::vector<uint32_t> order(g.F);
    for (uint32_t i = 0; i < g.F; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [](uint32_t a, uint32_t b) { const uint32_t ha = mrf_hash32(a), hb = mrf_hash32(b); return ha != hb ? ha < hb : a < b; });
    colour.assign(g.F, 255);
    int n_colours = 0;
    for (uint32_t i : order) {
        uint64_t used = 0;
        for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1]; ++e) { const uint8_t cj = colour[g.adj[e]]; if (cj != 255) used |= 1ull << cj; }
        uint8_t c = 0; while (used & (1ull << c)) ++c;      // degree <= 48 (UniGraph lists are de-duplicated): c <= 48
        colour[i] = c; n_colours = std::max(n_colours, (int)c + 1);
    }
    return n_colours;
}

constexpr uint32_t MRF_DAMP_PERIOD = 4u;   // damped sweeps: 1, 5, 9, ... (csrc/k_mrf.hip sweep_alpha restates it)

// One phase of a sweep: all nodes of colour `phase` (an independent set) recompute their outgoing messages IN PLACE
// from the current messages -- colour-phased Gauss-Seidel.  Within a phase no node reads what another writes.
// Damping schedule: alpha = P.damping on every FOURTH sweep (1st, 5th, 9th, ...), none on the others (round 6: scored in milliseconds
// on configs 2, 3 and the real-like scene -- profiles/r06_schedule_score_*.json; rounds 1 - 5 damped the odd sweeps).
// Messages live as their 8-bit codes (value = code * step); the update is written on the codes, with explicit fused
// multiply-adds (IEEE fma: one rounding, identical on the CPU and the GPU):
//   Sc[t] = sum of the incoming codes (small integers: exact in fp32 in any order)
//   b[t]  = fma(rho * step, Sc[t], D[t])                       sel_i = first argmin_t b[t]
//   c[t]  = fma(-step, code_e[t], b[t]) * oms                  ( D + rho * sum_all - m_e, the reweighted cavity, in damped
//                                                                code units: oms = (1 - alpha) * scale, lam_s = lam * oms )
//   raw   = (p == NONE) ? lam_s : fmin(c[p] - cmin, lam_s)
//   code' = rne( fma(old_code, alpha, raw) ), saturated to 255
void mrf_sweep(const Mrf& g, const orc_mrf_params& P, std::vector<uint8_t>& msg,
               std::vector<uint32_t>& sel, int n_threads, const uint8_t* colour, int phase, uint32_t sweep_no) {
    const float lam = 1.0f / P.rho;
    const float alpha = (sweep_no % MRF_DAMP_PERIOD == 1u) ? P.damping : 0.0f;
    const MsgQ mq = msg_q(lam);
    const float kappa = P.rho * mq.step, nstep = -mq.step, oms = (1.0f - alpha) * mq.scale, lam_s = lam * oms;
#pragma omp parallel num_threads(n_threads)
    {
        std::vector<float> b, c;
#pragma omp for schedule(dynamic, 1024)
        for (int64_t ii = 0; ii < (int64_t)g.F; ++ii) {
            const uint32_t i = (uint32_t)ii, Ki = g.K(i);
            if (Ki == 0 || colour[i] != phase) continue;
            const float* D = g.qcost + g.col_ptr[i];
            const uint32_t e0 = g.adj_ptr[i], e1 = g.adj_ptr[i + 1];
            // decode
            b.resize(Ki); c.resize(Ki);
            uint32_t best_t = 0; float best_b = 0.0f;
            for (uint32_t t = 0; t < Ki; ++t) {
                float Sc = 0.0f;
                for (uint32_t e = e0; e < e1; ++e) if (g.valid[e]) Sc = Sc + (float)msg[g.moff[e] + t];
                b[t] = __builtin_fmaf(kappa, Sc, D[t]);
                if (t == 0 || b[t] < best_b) { best_b = b[t]; best_t = t; }
            }
            sel[i] = best_t;
            // outgoing messages
            for (uint32_t e = e0; e < e1; ++e) {
                if (!g.valid[e]) continue;
                float cmin = 0.0f;                               // of the cavity in (damped) code units: c * oms
                for (uint32_t t = 0; t < Ki; ++t) {
                    c[t] = __builtin_fmaf(nstep, (float)msg[g.moff[e] + t], b[t]) * oms;
                    if (t == 0 || c[t] < cmin) cmin = c[t];
                }
                const uint32_t j = g.adj[e], r = g.rev[e], Kj = g.K(j);
                const uint64_t o = g.moff[r];
                for (uint32_t t2 = 0; t2 < Kj; ++t2) {
                    const uint16_t p = g.map[o + t2];
                    const float raw = (p == MAP_NONE) ? lam_s : std::fmin(c[p] - cmin, lam_s);
                    msg[o + t2] = (uint8_t)msg_code(raw, alpha, msg[o + t2]);
                }
            }
        }
    }
}

// one ICM iteration from labels `sel`; returns number of nodes moved
uint32_t mrf_icm_iter(const Mrf& g, std::vector<uint32_t>& sel, std::vector<float>& gain,
                      std::vector<uint32_t>& cand, int n_threads) {
#pragma omp parallel for schedule(dynamic, 1024) num_threads(n_threads)
    for (int64_t ii = 0; ii < (int64_t)g.F; ++ii) {
        const uint32_t i = (uint32_t)ii, Ki = g.K(i);
        gain[i] = 0.0f; cand[i] = 0;
        if (Ki == 0) continue;
        const float* D = g.cost + g.col_ptr[i]; const uint16_t* L = g.view_id + g.col_ptr[i];
        float best = 0.0f, cur = 0.0f; uint32_t bt = 0;
        for (uint32_t t = 0; t < Ki; ++t) {
            uint32_t diff = 0;
            for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1]; ++e) {
                if (!g.valid[e]) continue;
                const uint32_t j = g.adj[e];
                diff += (g.view_id[g.col_ptr[j] + sel[j]] != L[t]);
            }
            const float en = D[t] + (float)diff;
            if (t == 0 || en < best) { best = en; bt = t; }
            if (t == sel[i]) cur = en;
        }
        gain[i] = cur - best; cand[i] = bt;
    }
    uint32_t moved = 0;
    std::vector<uint32_t> nsel(sel);
#pragma omp parallel for schedule(dynamic, 1024) num_threads(n_threads) reduction(+ : moved)
    for (int64_t ii = 0; ii < (int64_t)g.F; ++ii) {
        const uint32_t i = (uint32_t)ii;
        if (!(gain[i] > 0.0f)) continue;
        bool win = true;
        for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1] && win; ++e) {
            if (!g.valid[e]) continue;
            const uint32_t j = g.adj[e];
            if (gain[j] > gain[i] || (gain[j] == gain[i] && j < i)) win = false;
        }
        if (win) { nsel[i] = cand[i]; ++moved; }
    }
    sel.swap(nsel);
    return moved;
}

// Region moves (option region_rounds; a two-level step in the spirit of mapMAP's multilevel contraction of same-label regions,
// view_selection.cpp:103-115 use_multilevel -- DEFINED HERE): a REGION = connected component of equally labelled faces over
// the model's edges, named by its smallest face.  A region may take the label l of a neighbouring region if every one of its
// faces has l among its candidates; the energy changes by  sum_i (D_i(l) - D_i(l_i))  -  #edges to neighbours labelled l
// (those edges stop being cut; every other boundary edge stays cut).  All in 32.32 fixed point: exact, order independent.
// Per region the best candidate (largest gain, ties to the smaller label); a region moves iff its gain is positive and
// beats the gains of all neighbouring regions (ties to the smaller region id) -- an independent set, so the energy drops by
// exactly the sum of the gains.  Returns the number of regions that moved.
uint32_t mrf_region_round(const Mrf& g, std::vector<uint32_t>& sel) {
    const uint32_t F = g.F;
    std::vector<uint32_t> lab(F), root(F);
    for (uint32_t i = 0; i < F; ++i) { lab[i] = g.K(i) ? (uint32_t)g.view_id[g.col_ptr[i] + sel[i]] + 1u : 0u; root[i] = i; }
    auto find = [&](uint32_t x) { while (root[x] != x) { root[x] = root[root[x]]; x = root[x]; } return x; };
    for (uint32_t i = 0; i < F; ++i)
        for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1]; ++e) {
            const uint32_t j = g.adj[e];
            if (!g.valid[e] || lab[j] != lab[i]) continue;
            uint32_t a = find(i), b = find(j);
            if (a != b) { if (a < b) root[b] = a; else root[a] = b; }      // the smaller id stays the root
        }
    for (uint32_t i = 0; i < F; ++i) root[i] = find(i);
    std::vector<uint64_t> keys;                                            // (region << 16 | label) per directed cut edge
    for (uint32_t i = 0; i < F; ++i)
        for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1]; ++e) {
            const uint32_t j = g.adj[e];
            if (g.valid[e] && lab[j] != lab[i]) keys.push_back(((uint64_t)root[i] << 16) | lab[j]);
        }
    std::sort(keys.begin(), keys.end());
    std::vector<uint64_t> ck; std::vector<uint32_t> cnt;
    for (size_t k = 0; k < keys.size(); ++k) { if (k == 0 || keys[k] != keys[k - 1]) { ck.push_back(keys[k]); cnt.push_back(0); } cnt.back()++; }
    std::vector<uint64_t> cur(F, 0), sum(ck.size(), 0); std::vector<uint32_t> size(F, 0), have(ck.size(), 0);
    for (uint32_t i = 0; i < F; ++i) {
        const uint32_t R = root[i], p0 = g.col_ptr[i], K = g.K(i);
        size[R]++; cur[R] += K ? fix32(g.cost[p0 + sel[i]]) : fix32(1.0f);
        const size_t c0 = std::lower_bound(ck.begin(), ck.end(), (uint64_t)R << 16) - ck.begin();
        for (size_t c = c0; c < ck.size() && (ck[c] >> 16) == R; ++c) {
            const uint16_t v = (uint16_t)((ck[c] & 0xFFFFu) - 1u);
            const uint16_t* L = g.view_id + p0;
            const uint16_t* it = std::lower_bound(L, L + K, v);
            if (it != L + K && *it == v) { have[c]++; sum[c] += fix32(g.cost[p0 + (it - L)]); }
        }
    }
    std::vector<int64_t> gain(F, 0); std::vector<uint32_t> bestl(F, 0);
    for (size_t c = 0; c < ck.size(); ++c) {
        const uint32_t R = (uint32_t)(ck[c] >> 16), l = (uint32_t)(ck[c] & 0xFFFFu);
        if (have[c] != size[R]) continue;
        const int64_t gn = (int64_t)(cur[R] - sum[c]) + ((int64_t)cnt[c] << 32);
        if (gn > gain[R] || (gn == gain[R] && gn > 0 && l < bestl[R])) { gain[R] = gn; bestl[R] = l; }   // candidates ascend in l: ties keep the first
    }
    std::vector<uint8_t> lose(F, 0);
    for (uint32_t i = 0; i < F; ++i)
        for (uint32_t e = g.adj_ptr[i]; e < g.adj_ptr[i + 1]; ++e) {
            const uint32_t j = g.adj[e];
            if (!g.valid[e] || lab[j] == lab[i]) continue;
            const uint32_t R = root[i], S = root[j];
            if (gain[S] > gain[R] || (gain[S] == gain[R] && S < R)) lose[R] = 1;
        }
    uint32_t moved = 0;
    for (uint32_t i = 0; i < F; ++i) {
        const uint32_t R = root[i];
        if (gain[R] <= 0 || lose[R]) continue;
        const uint32_t p0 = g.col_ptr[i], K = g.K(i);
        const uint16_t v = (uint16_t)(bestl[R] - 1u);
        const uint16_t* L = g.view_id + p0;
        sel[i] = (uint32_t)(std::lower_bound(L, L + K, v) - L);
        if (i == R) ++moved;
    }
    return moved;
}

}  // namespace

extern "C" {

// For the pins against the reference's own texture_view.cpp (oracle/_ref, tests/test_reference_pins.py): an IDENTITY
// camera (projection = I, world_to_cam = I) maps the vertex (x + 0.5, y + 0.5, 1) to the pixel coordinates (x, y) exactly
// in any accumulation order, so the mask / valid_pixel / get_face_info logic can be compared on chosen 2D inputs.
static orc_view identity_view(const uint8_t* rgb, int w, int h) {
    orc_view v; memset(&v, 0, sizeof(v));
    v.K[0] = v.K[4] = v.K[8] = 1.0f; v.w2c[0] = v.w2c[5] = v.w2c[10] = v.w2c[15] = 1.0f; v.viewdir[2] = 1.0f;
    v.width = w; v.height = h; v.rgb = rgb;
    return v;
}
// out[k] = TextureView::valid_pixel(xy[k]) after generate_validity_mask() (+ erode_validity_mask() if erode)
void orc_valid_pixel_map(const uint8_t* rgb, int w, int h, int erode, const float* xy, uint32_t n, uint8_t* out) {
    std::vector<uint8_t> mask((size_t)w * h);
    orc_validity_mask(rgb, w, h, mask.data());
    if (erode) orc_erode_validity_mask(mask.data(), w, h);
    const orc_view v = identity_view(rgb, w, h);
    for (uint32_t k = 0; k < n; ++k) out[k] = valid_pixel(v, mask.data(), V2{xy[2 * k], xy[2 * k + 1]}) ? 1 : 0;
}
// TextureView::get_face_info of n triangles given by 3D vertices (9 floats each) under the identity camera
void orc_face_info(const uint8_t* rgb, const uint8_t* gmi, int w, int h, int data_term, int outlier, const float* verts, uint32_t n,
                   float* quality, float* color) {
    const orc_view v = identity_view(rgb, w, h);
    orc_settings st; st.data_term = data_term; st.outlier_removal = outlier; st.geometric_visibility_test = 0;
    for (uint32_t k = 0; k < n; ++k) {
        FaceInfo fi; fi.quality = 0.0f; fi.mean_color[0] = fi.mean_color[1] = fi.mean_color[2] = 0.0f;
        const float* p = verts + 9 * (size_t)k;
        get_face_info(v, gmi, V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, V3{p[6], p[7], p[8]}, st, &fi);
        quality[k] = fi.quality; for (int i = 0; i < 3; ++i) color[3 * k + i] = fi.mean_color[i];
    }
}
// Tri as get_face_info uses it: out = {area, aabb min_x, min_y, max_x, max_y}; inside[k] for the n query points
void orc_tri(const float p[6], float out[5], const float* xy, uint32_t n, uint8_t* inside) {
    const V2 t1 = {p[0], p[1]}, t2 = {p[2], p[3]}, t3 = {p[4], p[5]};
    const TriR r = tri_make(t1, t2, t3);
    out[0] = tri_area(t1, t2, t3); out[1] = r.min_x; out[2] = r.min_y; out[3] = r.max_x; out[4] = r.max_y;
    for (uint32_t k = 0; k < n; ++k) inside[k] = tri_inside(t1, t2, t3, r.detT, xy[2 * k], xy[2 * k + 1]) ? 1 : 0;
}
uint32_t orc_msg_code(float raw, float rho, float alpha, uint32_t old_code) {
    const MsgQ q = msg_q(1.0f / rho);
    return msg_code(raw * ((1.0f - alpha) * q.scale), alpha, old_code);
}
void orc_mrf_set_trace(uint64_t* buf, int len) { g_trace = buf; g_trace_len = len; }

void orc_mrf_default_params(orc_mrf_params* p) {
    p->max_sweeps = 200; p->min_sweeps = 20; p->window = 5; p->min_improvement = 0.005f;
    p->damping = 0.2f; p->rho = 0.8f; p->icm_iters = 50; p->region_rounds = 0;
}

int orc_view_selection(const orc_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj,
                       const orc_mrf_params* params, int n_threads, uint32_t* labels,
                       orc_mrf_stats* stats) {
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    orc_mrf_params P; if (params) P = *params; else orc_mrf_default_params(&P);
    orc_mrf_stats S; memset(&S, 0, sizeof(S));
    double t0 = now_s();
    Mrf g; g.F = costs->n_faces; g.col_ptr = costs->col_ptr; g.view_id = costs->view_id; g.cost = costs->cost;
    g.adj_ptr = adj_ptr; g.adj = adj;
    mrf_setup(g);
    // The sweeps read the unaries as 16-bit fixed point over [0, 1] (costs are 1 - min(1, q / percentile),
    // calculate_data_costs.cpp:291-296): code = trunc(c * 65535 + 0.5), value = code * (1 / 65535), fp32 on both sides.
    // Part of the solver's definition: the GPU streams {view id, cost code} as one 32-bit word per label.  The energies
    // that drive the stop rule and the choice of the best sweep are those of the SAME quantised unaries ("tracking energy");
    // the ICM polish and the energy that is reported use the exact costs.
    std::vector<float> qcost(costs->nnz); std::vector<uint16_t> qcode(costs->nnz);
    for (uint64_t k = 0; k < costs->nnz; ++k) {
        qcode[k] = (uint16_t)(uint32_t)(costs->cost[k] * 65535.0f + 0.5f);
        qcost[k] = (float)(int32_t)qcode[k] * (1.0f / 65535.0f);
    }
    g.qcost = qcost.data();
    const uint64_t M = g.moff[g.adj_ptr[g.F]];
    std::vector<uint8_t> msg(M, 0);   // 8-bit codes
    std::vector<uint8_t> colour;
    const int n_colours = mrf_colour(g, colour);
    std::vector<uint32_t> sel(g.F, 0), best_sel(g.F, 0);
    S.t_setup = now_s() - t0; t0 = now_s();
    uint64_t best_e = ~0ull, best_cuts = 0;
    std::vector<uint64_t> hist; hist.push_back(~0ull);
    uint32_t s = 0;
    for (s = 1; (int)s <= P.max_sweeps; ++s) {
        for (int phase = 0; phase < n_colours; ++phase) mrf_sweep(g, P, msg, sel, n_threads, colour.data(), phase, s);
        uint64_t cuts; const uint64_t e = mrf_energy_sel(g, sel, &cuts, n_threads, qcode.data());
        if (e < best_e) { best_e = e; best_cuts = cuts; best_sel = sel; }
        hist.push_back(best_e);
        if (g_trace && (int)s <= g_trace_len) g_trace[s - 1] = e;
        if ((int)s >= P.min_sweeps && (int)s > P.window) {
            const uint64_t prev = hist[s - P.window];
            if ((double)(prev - best_e) < (double)P.min_improvement * (double)prev) break;
        }
    }
    S.sweeps = std::min<uint32_t>(s, (uint32_t)P.max_sweeps);
    if (P.max_sweeps <= 0) {  // no message passing: start ICM from the argmin-unary labeling
        for (uint32_t i = 0; i < g.F; ++i) {
            uint32_t bt = 0;
            for (uint32_t t = 1; t < g.K(i); ++t) if (g.cost[g.col_ptr[i] + t] < g.cost[g.col_ptr[i] + bt]) bt = t;
            best_sel[i] = bt;
        }
    }
    std::vector<float> gain(g.F); std::vector<uint32_t> cand(g.F);
    int it = 0;
    for (; it < P.icm_iters; ++it) if (mrf_icm_iter(g, best_sel, gain, cand, n_threads) == 0) break;
    S.icm_iters = (uint32_t)it;
    /* region moves (off by default: region_rounds = 0), each round followed by a fresh ICM polish */
    for (int r = 0; r < P.region_rounds; ++r) {
        const uint32_t m = mrf_region_round(g, best_sel);
        if (m == 0) break;
        S.region_rounds++; S.region_moves += m;
        for (it = 0; it < P.icm_iters; ++it) { S.icm_iters++; if (mrf_icm_iter(g, best_sel, gain, cand, n_threads) == 0) break; }
    }
    best_e = mrf_energy_sel(g, best_sel, &best_cuts);
    /* label extraction (view_selection.cpp:120-132) */
    for (uint32_t i = 0; i < g.F; ++i) {
        if (g.K(i) == 0) { labels[i] = 0; S.unseen++; }
        else labels[i] = (uint32_t)g.view_id[g.col_ptr[i] + best_sel[i]] + 1u;
    }
    S.energy_fixed = best_e; S.energy = (double)best_e / 4294967296.0; S.cut_edges = best_cuts;
    S.t_solve = now_s() - t0;
    if (stats) *stats = S;
    return 0;
}

uint64_t orc_energy(const orc_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj,
                    const uint32_t* labels, uint64_t* cut_edges) {
    const uint32_t F = costs->n_faces;
    uint64_t unary = 0, cuts = 0;
    auto K = [&](uint32_t i) { return costs->col_ptr[i + 1] - costs->col_ptr[i]; };
    for (uint32_t i = 0; i < F; ++i) {
        if (K(i) == 0) { if (labels[i] != 0) return ~0ull; unary += fix32(1.0f); continue; }
        bool found = false;
        for (uint32_t k = costs->col_ptr[i]; k < costs->col_ptr[i + 1]; ++k)
            if ((uint32_t)costs->view_id[k] + 1u == labels[i]) { unary += fix32(costs->cost[k]); found = true; break; }
        if (!found) return ~0ull;
        for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
            const uint32_t j = adj[e];
            if (j <= i || K(j) == 0) continue;
            if (labels[j] != labels[i]) ++cuts;
        }
    }
    if (cut_edges) *cut_edges = cuts;
    return unary + (cuts << 32);
}

// A LOWER BOUND on the minimum of E over all labelings (the LP relaxation's dual, by MPLP block coordinate ascent:
// Globerson & Jaakkola 2007), in fp64.  No labeling -- mapMAP's included -- has a smaller energy, so
// (E(our labels) - bound) / bound bounds how much better ANY solver could do: the quality statement that stands in for
// the unavailable label-for-label comparison with mapMAP.  Dual variables lam_{e->i}(x_i) per edge side with
// lam_{e->i}(x_i) + lam_{e->j}(x_j) <= [x_i != x_j] maintained by every update; bound = sum_i min_x (D_i(x) + sum_e lam_{e->i}(x))
// + 1 per face without candidates (as orc_energy counts them).  trace[0..iters) receives the bound after each round.
double orc_mrf_lower_bound(const orc_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj, int iters, int n_threads, double* trace) {
    const uint32_t F = costs->n_faces;
    auto K = [&](uint32_t i) { return costs->col_ptr[i + 1] - costs->col_ptr[i]; };
    struct Edge { uint32_t i, j; size_t oi, oj; };   // message offsets of the two sides
    std::vector<Edge> edges;
    size_t total = 0;
    double constant = 0.0;
    for (uint32_t i = 0; i < F; ++i) {
        if (K(i) == 0) { constant += 1.0; continue; }
        for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
            const uint32_t j = adj[e];
            if (j <= i || K(j) == 0) continue;
            edges.push_back({i, j, total, total + K(i)});
            total += (size_t)K(i) + K(j);
        }
    }
    std::vector<double> lam(total, 0.0), b((size_t)costs->col_ptr[F]);
    for (size_t k = 0; k < b.size(); ++k) b[k] = (double)costs->cost[k];
    // greedy edge colouring: edges of one colour share no node, so a colour class is updated in parallel (any order ascends)
    std::vector<std::vector<uint32_t>> classes;
    {
        std::vector<std::vector<bool>> used(F);
        for (uint32_t e = 0; e < edges.size(); ++e) {
            auto &ui = used[edges[e].i], &uj = used[edges[e].j];
            size_t c = 0;
            while ((c < ui.size() && ui[c]) || (c < uj.size() && uj[c])) ++c;
            if (ui.size() <= c) ui.resize(c + 1, false);
            if (uj.size() <= c) uj.resize(c + 1, false);
            ui[c] = uj[c] = true;
            if (classes.size() <= c) classes.resize(c + 1);
            classes[c].push_back(e);
        }
    }
    if (n_threads <= 0) n_threads = 1;
    auto bound = [&]() {
        double s = constant;
        for (uint32_t i = 0; i < F; ++i) {
            if (K(i) == 0) continue;
            double m = INFINITY;
            for (uint32_t k = costs->col_ptr[i]; k < costs->col_ptr[i + 1]; ++k) m = std::min(m, b[k]);
            s += m;
        }
        return s;
    };
    for (int it = 0; it < iters; ++it) {
        for (const auto& cls : classes) {
#pragma omp parallel for schedule(static) num_threads(n_threads) if (cls.size() > 2048)
        for (int64_t ce = 0; ce < (int64_t)cls.size(); ++ce) {
            const Edge& ed = edges[cls[ce]];
            const uint32_t ci = costs->col_ptr[ed.i], cj = costs->col_ptr[ed.j], ki = K(ed.i), kj = K(ed.j);
            std::vector<double> ai(ki), aj(kj);
            double mi = INFINITY, mj = INFINITY;
            for (uint32_t x = 0; x < ki; ++x) { ai[x] = b[ci + x] - lam[ed.oi + x]; mi = std::min(mi, ai[x]); }
            for (uint32_t x = 0; x < kj; ++x) { aj[x] = b[cj + x] - lam[ed.oj + x]; mj = std::min(mj, aj[x]); }
            // min over the other side of [x != x'] + a(x'): 1 + its minimum, or its value at the same view (lists ascend by view id)
            uint32_t y = 0;
            for (uint32_t x = 0; x < ki; ++x) {
                const uint16_t v = costs->view_id[ci + x];
                while (y < kj && costs->view_id[cj + y] < v) ++y;
                double m = 1.0 + mj;
                if (y < kj && costs->view_id[cj + y] == v) m = std::min(m, aj[y]);
                const double l = -0.5 * ai[x] + 0.5 * m;
                lam[ed.oi + x] = l; b[ci + x] = ai[x] + l;
            }
            y = 0;
            for (uint32_t x = 0; x < kj; ++x) {
                const uint16_t v = costs->view_id[cj + x];
                while (y < ki && costs->view_id[ci + y] < v) ++y;
                double m = 1.0 + mi;
                if (y < ki && costs->view_id[ci + y] == v) m = std::min(m, ai[y]);
                const double l = -0.5 * aj[x] + 0.5 * m;
                lam[ed.oj + x] = l; b[cj + x] = aj[x] + l;
            }
        }
        }
        if (trace) trace[it] = bound();
    }
    // feasibility, checked rather than assumed (rounding): the largest violation of lam_i(x) + lam_j(x') <= [x != x'] is subtracted
    double slack = 0.0;
    for (const Edge& ed : edges) {
        const uint32_t ci = costs->col_ptr[ed.i], cj = costs->col_ptr[ed.j], ki = K(ed.i), kj = K(ed.j);
        double mxi = -INFINITY, mxj = -INFINITY, same = -INFINITY;
        for (uint32_t x = 0; x < ki; ++x) mxi = std::max(mxi, lam[ed.oi + x]);
        for (uint32_t x = 0; x < kj; ++x) mxj = std::max(mxj, lam[ed.oj + x]);
        uint32_t y = 0;
        for (uint32_t x = 0; x < ki; ++x) {
            const uint16_t v = costs->view_id[ci + x];
            while (y < kj && costs->view_id[cj + y] < v) ++y;
            if (y < kj && costs->view_id[cj + y] == v) same = std::max(same, lam[ed.oi + x] + lam[ed.oj + y]);
        }
        slack += std::max(0.0, std::max(same, mxi + mxj - 1.0));
    }
    return bound() - slack;
}

int orc_icm_baseline(const orc_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj,
                     int max_iters, uint32_t* labels) {
    orc_mrf_params P; orc_mrf_default_params(&P);
    P.max_sweeps = 0; P.icm_iters = max_iters;
    return orc_view_selection(costs, adj_ptr, adj, &P, 0, labels, nullptr);
}

}  // extern "C"

// ===========================================================================
// Row f1: the two stages immediately BEFORE the path.
// ===========================================================================
extern "C" {

// tex::build_adjacency_graph (build_adjacency_graph.cpp:16-53) + UniGraph::add_edge (uni_graph.h:86-93).
// MeshInfo::get_faces_for_edge (MVE, absent) returns the faces containing both vertices; for a manifold edge
// that is {self, other}.  For an edge shared by more than two faces its order depends on MeshInfo internals:
// DEFINED HERE as ascending face id.  Output: adjacency lists flattened in list order (malloc'ed).
int orc_build_adjacency(uint32_t n_faces, const uint32_t* faces, uint32_t** adj_ptr_out, uint32_t** adj_out) {
    struct EdgeRec { uint64_t key; uint32_t face; };
    std::vector<EdgeRec> recs; recs.reserve((size_t)n_faces * 3);
    auto ekey = [](uint32_t a, uint32_t b) { return (uint64_t)std::min(a, b) << 32 | std::max(a, b); };
    for (uint32_t f = 0; f < n_faces; ++f)
        for (int k = 0; k < 3; ++k) recs.push_back({ekey(faces[3 * (size_t)f + k], faces[3 * (size_t)f + (k + 1) % 3]), f});
    std::vector<EdgeRec> sorted(recs);
    std::stable_sort(sorted.begin(), sorted.end(), [](const EdgeRec& a, const EdgeRec& b) { return a.key < b.key; });
    /* mve::MeshInfo::get_faces_for_edge(v1, v2) (MVE, absent; recollection -- DEFINED HERE): the faces in v1's list that are also in
     * v2's list, ascending.  For v1 != v2 these are the faces owning the edge; for a face with a repeated vertex the reference asks
     * for the "edge" (a, a) and gets EVERY face at a. */
    uint32_t n_verts = 0;
    for (size_t i = 0; i < 3 * (size_t)n_faces; ++i) n_verts = std::max(n_verts, faces[i] + 1);
    std::vector<std::vector<uint32_t>> vfaces(n_verts);
    for (uint32_t f = 0; f < n_faces; ++f)
        for (int k = 0; k < 3; ++k) { auto& l = vfaces[faces[3 * (size_t)f + k]]; if (l.empty() || l.back() != f) l.push_back(f); }
    auto faces_for_edge = [&](uint64_t key, std::vector<uint32_t>* out) {
        if ((uint32_t)(key >> 32) == (uint32_t)key) { const auto& l = vfaces[(uint32_t)key]; out->insert(out->end(), l.begin(), l.end()); return; }
        auto lo = std::lower_bound(sorted.begin(), sorted.end(), key, [](const EdgeRec& r, uint64_t k) { return r.key < k; });
        for (; lo != sorted.end() && lo->key == key; ++lo) out->push_back(lo->face);
    };
    std::vector<std::vector<uint32_t>> lists(n_faces);
    for (uint32_t f = 0; f < n_faces; ++f) {
        std::vector<uint32_t> adj_faces;                                   /* build_adjacency_graph.cpp:31-34 */
        for (int k = 0; k < 3; ++k) faces_for_edge(recs[3 * (size_t)f + k].key, &adj_faces);
        for (uint32_t g : adj_faces) {
            if (g == f) continue;                                          /* :41 avoid self referencing */
            auto& lf = lists[f];
            if (std::find(lf.begin(), lf.end(), g) != lf.end()) continue;  /* :43 has_edge */
            lf.push_back(g); lists[g].push_back(f);                        /* uni_graph.h:89-90 */
        }
    }
    uint32_t* ptr = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)n_faces + 1));
    uint64_t total = 0;
    for (uint32_t f = 0; f < n_faces; ++f) { ptr[f] = (uint32_t)total; total += lists[f].size(); }
    ptr[n_faces] = (uint32_t)total;
    uint32_t* adj = (uint32_t*)malloc(sizeof(uint32_t) * std::max<uint64_t>(total, 1));
    for (uint32_t f = 0; f < n_faces; ++f) std::copy(lists[f].begin(), lists[f].end(), adj + ptr[f]);
    *adj_ptr_out = ptr; *adj_out = adj;
    return 0;
}

// tex::prepare_mesh (prepare_mesh.cpp:14-70): remove_redundant_faces + ensure_normals.
// A face is redundant iff some face with a LARGER id that touches one of its vertices consists only of vertices
// of this face (:27-44).  Face normals (MVE TriangleMesh::recalc_normals, absent -- DEFINED HERE): normalised
// (b - a) x (c - a), the zero vector when its length is 0.  Returns the number of kept faces.
uint32_t orc_prepare_mesh(uint32_t n_verts, const float* verts, uint32_t n_faces, const uint32_t* faces,
                          uint32_t* faces_out, float* normals_out) {
    std::vector<std::vector<uint32_t>> vfaces(n_verts);
    for (uint32_t f = 0; f < n_faces; ++f) for (int k = 0; k < 3; ++k) vfaces[faces[3 * (size_t)f + k]].push_back(f);
    uint32_t kept = 0;
    for (uint32_t f = 0; f < n_faces; ++f) {
        const uint32_t* fv = faces + 3 * (size_t)f;
        bool redundant = false;
        for (int j = 0; !redundant && j < 3; ++j)
            for (uint32_t g : vfaces[fv[j]]) {
                if (redundant) break;
                if (f < g) {
                    bool identical = true;
                    for (int l = 0; l < 3; ++l) {
                        const uint32_t v = faces[3 * (size_t)g + l];
                        if (std::find(fv, fv + 3, v) == fv + 3) { identical = false; break; }
                    }
                    redundant = identical;
                }
            }
        if (redundant) continue;
        for (int k = 0; k < 3; ++k) faces_out[3 * (size_t)kept + k] = fv[k];
        const V3 a = load3(verts + 3 * (size_t)fv[0]), b = load3(verts + 3 * (size_t)fv[1]), c = load3(verts + 3 * (size_t)fv[2]);
        V3 n = cross(b - a, c - a);
        const float len = norm(n);
        if (len != 0.0f) n = n / len;
        normals_out[3 * (size_t)kept] = n.x; normals_out[3 * (size_t)kept + 1] = n.y; normals_out[3 * (size_t)kept + 2] = n.z;
        ++kept;
    }
    return kept;
}

}  // extern "C"

// ---- row f3 ----
// UniGraph::get_subgraphs (uni_graph.cpp:21-55), statement by statement, on the flattened adjacency lists.
static void get_subgraphs_of_label(uint32_t n_faces, const uint32_t* adj_ptr, const uint32_t* adj, const uint32_t* labels,
                                   uint32_t label, std::vector<std::vector<uint32_t>>* subgraphs) {
    std::vector<bool> used(n_faces, false);                                     /* :25 */
    for (uint32_t i = 0; i < n_faces; ++i) {                                    /* :27 */
        if (labels[i] == label && !used[i]) {                                   /* :28 */
            subgraphs->push_back(std::vector<uint32_t>());                      /* :29 */
            std::deque<uint32_t> queue;                                         /* :31 (std::list used as a FIFO) */
            queue.push_back(i); used[i] = true;                                 /* :33-34 */
            while (!queue.empty()) {                                            /* :36 */
                const uint32_t node = queue.front(); queue.pop_front();         /* :37-38 */
                subgraphs->back().push_back(node);                              /* :40 */
                for (uint32_t e = adj_ptr[node]; e < adj_ptr[node + 1]; ++e) {  /* :43-44 */
                    const uint32_t adj_node = adj[e];                           /* :45 */
                    if (labels[adj_node] == label && !used[adj_node]) {         /* :47 */
                        queue.push_back(adj_node); used[adj_node] = true;       /* :48-49 */
                    }
                }
            }
        }
    }
}

extern "C" {
// generate_texture_patches.cpp:469-475 calls it for label = i + 1 of every view; label 0 (unseen faces) is
// included here because the hole filling of the same file walks those components too.
uint32_t orc_get_subgraphs(uint32_t n_faces, const uint32_t* adj_ptr, const uint32_t* adj, const uint32_t* labels,
                           uint32_t n_labels, uint32_t* label_ptr, uint32_t** comp_ptr_out, uint32_t* comp_faces) {
    std::vector<uint32_t> comp_ptr; comp_ptr.push_back(0);
    uint32_t filled = 0;
    for (uint32_t label = 0; label < n_labels; ++label) {
        label_ptr[label] = (uint32_t)comp_ptr.size() - 1;
        std::vector<std::vector<uint32_t>> subgraphs;
        get_subgraphs_of_label(n_faces, adj_ptr, adj, labels, label, &subgraphs);
        for (const auto& sg : subgraphs) {
            for (uint32_t f : sg) comp_faces[filled++] = f;
            comp_ptr.push_back(filled);
        }
    }
    label_ptr[n_labels] = (uint32_t)comp_ptr.size() - 1;
    uint32_t* cp = (uint32_t*)malloc(sizeof(uint32_t) * comp_ptr.size());
    std::copy(comp_ptr.begin(), comp_ptr.end(), cp);
    *comp_ptr_out = cp;
    return (uint32_t)comp_ptr.size() - 1;
}

}  // extern "C"
