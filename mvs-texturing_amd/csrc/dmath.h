// dmath.h -- per-(face, view) arithmetic of the data-cost path as
// __host__ __device__ inline functions, shared by the HIP kernels.
//
// Each function states the reference lines it implements.  The float operation
// order is part of the contract (the library is compiled with
// -ffp-contract=off and correctly rounded fp32 divide / sqrt), so that results
// are bit-identical to a scalar IEEE-754 evaluation of the same expressions.
// The header also compiles as plain host C++ (MVS_HD empty) for the CPU-side
// unit test of the kernels' arithmetic (csrc/dmath_host.cpp).
#pragma once
#include <stdint.h>
#include <math.h>
#include <float.h>
#include <string.h>

#if defined(__HIPCC__)
#define MVS_HD __host__ __device__ __forceinline__
#else
#define MVS_HD inline
#endif

namespace mvs {

struct V3 { float x, y, z; };
struct V2 { float x, y; };

MVS_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
MVS_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
MVS_HD V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
MVS_HD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MVS_HD float norm(V3 a) { return sqrtf(dot(a, a)); }
MVS_HD V3 normalized(V3 a) { return a / norm(a); }
MVS_HD V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// std::min / std::max semantics (matter only for NaN / signed zero)
MVS_HD float smin(float a, float b) { return (b < a) ? b : a; }
MVS_HD float smax(float a, float b) { return (a < b) ? b : a; }
MVS_HD int imin(int a, int b) { return (b < a) ? b : a; }

// Device-side mirror of the TextureView fields read by the path
// (libs/tex/texture_view.h:43-48) plus the per-view derived images.
struct ViewParams {
    float pos[3];
    float viewdir[3];
    float K[9];
    float w2c[12];  // first three rows of world_to_cam
    int32_t width, height;
    int32_t mask_stride;        // 32-bit words per mask row
    int32_t msum_stride;        // 32-bit words per row of the tile summary
    const uint8_t* rgb;         // width*height*3
    const uint8_t* gmi;         // width*height (gradient magnitude), may be null
    const uint32_t* mask;       // bit-packed validity mask, rows padded to 32-bit words; NULL on the device copy of a view whose mask
                                // is all ones (k_prep.hip mask_trivial_kernel: nothing was flooded, the usual case): no look-up at all
    const uint32_t* msum;       // tile summary of the mask (may be null): bit (tx, ty) set iff every mask bit of the pixels
                                // [32 tx, 32 tx + 32] x [32 ty, 32 ty + 32] (clipped to the image) is set -- the four bits valid_pixel
                                // reads around a position in tile (tx, ty) are then known without reading them
};

// TextureView::get_pixel_coords  (texture_view.h:161-166)
MVS_HD V2 pixel_coords(const ViewParams& v, V3 p) {
    const float* m = v.w2c;
    const float c0 = ((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + 1.0f * m[3];
    const float c1 = ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + 1.0f * m[7];
    const float c2 = ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + 1.0f * m[11];
    const float* k = v.K;
    const float q0 = (k[0] * c0 + k[1] * c1) + k[2] * c2;
    const float q1 = (k[3] * c0 + k[4] * c1) + k[5] * c2;
    const float q2 = (k[6] * c0 + k[7] * c1) + k[8] * c2;
    return {q0 / q2 - 0.5f, q1 / q2 - 0.5f};
}

MVS_HD bool mask_bit(const ViewParams& v, int x, int y) {
    return (v.mask[(size_t)y * v.mask_stride + (x >> 5)] >> (x & 31)) & 1u;
}

// TextureView::valid_pixel  (texture_view.cpp:253-281)
MVS_HD bool valid_pixel(const ViewParams& v, V2 px) {
    const int width = v.width, height = v.height;
    const float x = px.x, y = px.y;
    bool valid = (x >= 0.0f && x < (float)(width - 1) && y >= 0.0f && y < (float)(height - 1));
    if (valid && v.mask) {
        const float cx = smax(0.0f, smin((float)(width - 1), x));
        const float cy = smax(0.0f, smin((float)(height - 1), y));
        const int floor_x = (int)cx, floor_y = (int)cy;
        const int floor_xp1 = imin(floor_x + 1, width - 1), floor_yp1 = imin(floor_y + 1, height - 1);
        valid = mask_bit(v, floor_x, floor_y) && mask_bit(v, floor_x, floor_yp1) &&
                mask_bit(v, floor_xp1, floor_y) && mask_bit(v, floor_xp1, floor_yp1);
    }
    return valid;
}

// Culling tests of calculate_face_projection_infos (calculate_data_costs.cpp:171-191).
// Returns 0 = passes, 1 = backface / behind camera (:183-185), 2 = angle (:187-188),
// 3 = projects outside the valid image area (:191).
// cos_limit replaces `std::acos(viewing_angle) > MATH_DEG2RAD(75.0f)`: it is the
// smallest float c with !(acosf(c) > 75 deg), found on the host with the host's
// acosf, so `viewing_angle < cos_limit` decides identically (acosf is monotone).
// the last cull (calculate_data_costs.cpp:191): all three vertices project onto valid pixels
// valid_pixel of the three vertices at once, branch free: the reference's short-circuit chain (three vertices x four mask bits, each
// read only if everything before it held) is twelve DEPENDENT memory latencies per (face, view) pair on the GPU -- the culls kernel
// spent 72 % of its wave cycles parked on them (profiles/r03a_pmc_sq_c3.json).  Here the three range tests come first, then all
// twelve mask words are requested together and ANDed: one latency.  Same boolean: a conjunction does not depend on the order of its
// terms; for a position inside the range the reference's clamps are identities, and a pair with a position outside it is false
// already and looks nothing up.  Round 4: the look-ups are the exception -- a view whose mask is all ones has mask == NULL on the
// device, and elsewhere a per-tile summary answers for every position whose 32 x 32 tile (+ 1 pixel) is wholly valid.
MVS_HD bool valid_pixels3(const ViewParams& v, V2 a, V2 b, V2 c) {
    const int width = v.width, height = v.height;
    const float wm = (float)(width - 1), hm = (float)(height - 1);
    const V2 p[3] = {a, b, c};
    bool in[3];
    for (int k = 0; k < 3; ++k) in[k] = (p[k].x >= 0.0f && p[k].x < wm && p[k].y >= 0.0f && p[k].y < hm);
    bool valid = in[0] && in[1] && in[2];
    if (v.mask && valid) {
        int fx[3], fy[3];
        for (int k = 0; k < 3; ++k) { fx[k] = (int)p[k].x; fy[k] = (int)p[k].y; }
        // Tile summary first (384 bytes per 2048 x 1536 view: cache resident): where the three tiles are wholly valid -- everywhere
        // but along the rim of a flooded region -- the twelve mask bits are ones without being read.  Same boolean by construction.
        uint32_t whole = 0u;
        if (v.msum) {
            whole = 1u;
            for (int k = 0; k < 3; ++k) whole &= v.msum[(size_t)(fy[k] >> 5) * v.msum_stride + (fx[k] >> 10)] >> ((fx[k] >> 5) & 31);
        }
        if (!(whole & 1u)) {
            uint32_t bits = 1u;
            for (int k = 0; k < 3; ++k) {
                const int fx1 = imin(fx[k] + 1, width - 1), fy1 = imin(fy[k] + 1, height - 1);
                const uint32_t* r0 = v.mask + (size_t)fy[k] * v.mask_stride; const uint32_t* r1 = v.mask + (size_t)fy1 * v.mask_stride;
                const uint32_t w00 = r0[fx[k] >> 5], w01 = r1[fx[k] >> 5], w10 = r0[fx1 >> 5], w11 = r1[fx1 >> 5];
                bits &= (w00 >> (fx[k] & 31)) & (w01 >> (fx[k] & 31)) & (w10 >> (fx1 & 31)) & (w11 >> (fx1 & 31));
            }
            valid = (bits & 1u) != 0u;
        }
    }
    return valid;
}
MVS_HD int cull_pixels(const ViewParams& v, V3 v1, V3 v2, V3 v3) {
    return valid_pixels3(v, pixel_coords(v, v1), pixel_coords(v, v2), pixel_coords(v, v3)) ? 0 : 3;
}
MVS_HD int cull_pair(const ViewParams& v, V3 v1, V3 v2, V3 v3, V3 face_normal, float cos_limit) {
    const V3 view_pos = {v.pos[0], v.pos[1], v.pos[2]};
    const V3 viewing_direction = {v.viewdir[0], v.viewdir[1], v.viewdir[2]};
    const V3 face_center = ((v1 + v2) + v3) / 3.0f;
    const V3 view_to_face_vec = normalized(face_center - view_pos);
    const V3 face_to_view_vec = normalized(view_pos - face_center);
    const float viewing_angle = dot(face_to_view_vec, face_normal);
    if (viewing_angle < 0.0f || dot(viewing_direction, view_to_face_vec) < 0.0f) return 1;
    if (viewing_angle < cos_limit) return 2;
    return cull_pixels(v, v1, v2, v3);
}

// cull_pair with its clear cases decided WITHOUT the two normalisations (2 sqrt + 6 correctly rounded divisions per pair): the
// culls kernel's entry point; identical reason codes (tests/test_host.py compares it with cull_pair on adversarial pairs).
//  * first cull (:183-185: the face looks away from the view, or lies behind it).  With d = view_pos - centre exactly as
//    cull_pair forms it, the reference's viewing_angle = dot(d / |d|, n) evaluated in fp32 differs from dot(d, n) / |d| by less
//    than 4e-7 A / |d|, A = sum |d_k n_k|, and the unnormalised fp32 dot from its exact value by less than 2e-7 A: dot(d, n) <
//    -4e-6 A therefore implies viewing_angle < 0 with a tenfold margin; likewise for dot(viewdir, -d);
//  * angle cull (:187-188), only when both dot products are as clearly on the FRONT side (reason 1 excluded): s = sqrt(dd)
//    differs from |d| by less than 4e-7 |d| (correctly rounded on the host, 1 ulp on the device: both inside the margin), so
//    comparing un -+ 4e-6 A with cos_limit s (1 +- 4e-6) decides `viewing_angle < cos_limit`;
//  * everything in between takes cull_pair unchanged.  (A NaN cos_limit -- host_cos_limit's failure value -- fails both
//    comparisons and falls through to cull_pair.)
MVS_HD int cull_pair_prefiltered(const ViewParams& vw, V3 v1, V3 v2, V3 v3, V3 nrm, V3 centre /* ((v1 + v2) + v3) / 3 */, float cos_limit) {
    const V3 d = V3{vw.pos[0], vw.pos[1], vw.pos[2]} - centre;
    const float un = (d.x * nrm.x + d.y * nrm.y) + d.z * nrm.z, an = (fabsf(d.x * nrm.x) + fabsf(d.y * nrm.y)) + fabsf(d.z * nrm.z);
    const float uv = (d.x * vw.viewdir[0] + d.y * vw.viewdir[1]) + d.z * vw.viewdir[2],
                av = (fabsf(d.x * vw.viewdir[0]) + fabsf(d.y * vw.viewdir[1])) + fabsf(d.z * vw.viewdir[2]);
    if ((un < -4e-6f * an) || (uv > 4e-6f * av)) return 1;      // dot(viewdir, centre - pos) = -uv < 0
    const bool clear_front = (un > 4e-6f * an) && (uv < -4e-6f * av);
    const float dd = (d.x * d.x + d.y * d.y) + d.z * d.z;
#if defined(__HIP_DEVICE_COMPILE__)
    const float ls = cos_limit * __builtin_amdgcn_sqrtf(dd);
#else
    const float ls = cos_limit * sqrtf(dd);
#endif
    if (clear_front && un + 4e-6f * an <= ls * (1.0f - 4e-6f)) return 2;
    if (clear_front && un - 4e-6f * an >= ls * (1.0f + 4e-6f)) return cull_pixels(vw, v1, v2, v3);
    return cull_pair(vw, v1, v2, v3, nrm, cos_limit);
}

// Visibility ray of calculate_data_costs.cpp:200-206: origin = vertex,
// dir = normalised (view_pos - origin), tmax = |view_pos - origin|, tmin = 1e-4 tmax.
// `pad` is the scene-scale slack of the hit predicate (see ray_tri).
struct Ray { V3 o, d; float tmin, tmax, pad; };
MVS_HD Ray make_ray(V3 origin, V3 view_pos, float pad) {
    Ray r;
    r.o = origin;
    V3 dir = view_pos - origin;
    r.tmax = norm(dir);
    r.tmin = r.tmax * 0.0001f;
    r.d = dir / norm(dir);
    r.pad = pad;
    return r;
}
// pad = 1e-5 * max(scene extent, largest |coordinate|) + 1e-30 (scene box = exact min/max of the vertices)
MVS_HD float scene_pad(const float lo[3], const float hi[3]) {
    float ext = 0.0f, mag = 0.0f;
    for (int a = 0; a < 3; ++a) {
        ext = fmaxf(ext, hi[a] - lo[a]);
        mag = fmaxf(mag, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
    }
    return 1e-5f * fmaxf(ext, mag) + 1e-30f;
}

// Ray / triangle any-hit test on a triangle given as {a, e1 = b - a, e2 = c - a}.
// rayint (acc::BVHTree, calculate_data_costs.cpp:23,144,209) is not available and
// the reference only uses the boolean, so the predicate is defined by this
// library (DESIGN.md "Occlusion rays"): Moeller-Trumbore in fp32 with an exact
// operation order (fused multiply-adds, listed at ray_tri_boxed), in the division-free
// form -- with s = sign(det) the scaled barycentrics s u det, s v det and the scaled
// distance s t det are compared with 0, |det|, tmin |det| and tmax |det| (no slack
// anywhere) -- and, for a ray that
// passes all of these, t = (t det) / det (correctly rounded) AND the computed hit
// point o + t d must lie inside the triangle's bounding box grown by `pad`.
// The last clause makes box culling provably conservative (a grazing ray whose
// rounded barycentrics are accepted although it misses the triangle's box is
// rejected by every implementation alike), so the OR over all triangles does not
// depend on the acceleration structure.  Only a real crossing pays for the
// division and the box clause; everything before is branch free.
// The box clause is evaluated against (lo, hi) = triangle box grown by pad, which the GPU stores with the triangle
// (tri_pad_box below, evaluated once at build time instead of once per test -- the same fp32 expressions, so the
// predicate is bit-identical either way).
MVS_HD void tri_pad_box(V3 a, V3 e1, V3 e2, float pad, V3* lo, V3* hi) {
    const V3 b = a + e1, c = a + e2;
    lo->x = fminf(a.x, fminf(b.x, c.x)) - pad; hi->x = fmaxf(a.x, fmaxf(b.x, c.x)) + pad;
    lo->y = fminf(a.y, fminf(b.y, c.y)) - pad; hi->y = fmaxf(a.y, fmaxf(b.y, c.y)) + pad;
    lo->z = fminf(a.z, fminf(b.z, c.z)) - pad; hi->z = fmaxf(a.z, fmaxf(b.z, c.z)) + pad;
}
// The arithmetic of the predicate, fixed to the operation (IEEE fma = one rounding; the lane-pair kernel evaluates two
// triangles per lane with the packed forms of exactly these operations):
//   cross(a, b) = { fma(a.y, b.z, -(a.z b.y)), fma(a.z, b.x, -(a.x b.z)), fma(a.x, b.y, -(a.y b.x)) }
//   dot(a, b)   = fma(a.z, b.z, fma(a.y, b.y, a.x b.x))
//   pv = cross(d, e2), det = dot(e1, pv), tv = o - a, un = dot(tv, pv), qv = cross(tv, e1), vn = dot(d, qv), tn = dot(e2, qv)
//   sg = det < 0 ? -1 : 1; ad = det sg, us = un sg, vs = vn sg, ts = tn sg           (exact)
//   w = ad - (us + vs), g1 = fma(-tmin, ad, ts), g2 = fma(tmax, ad, -ts)
//   pre = ad > 0 && min(min(min(min(us, vs), w), g1), g2) >= 0                         (min = fminf)
//   hit = pre && the point o + (tn / det) d lies in [lo, hi]                           (division correctly rounded)
MVS_HD float dot_fma(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
MVS_HD V3 cross_fma(V3 a, V3 b) {
    return V3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}
MVS_HD bool ray_tri_point_in_box(const Ray& r, float tn, float det, V3 lo, V3 hi) {
    const float t = tn / det;
    const float hx = r.o.x + t * r.d.x, hy = r.o.y + t * r.d.y, hz = r.o.z + t * r.d.z;
    return hx >= lo.x && hx <= hi.x && hy >= lo.y && hy <= hi.y && hz >= lo.z && hz <= hi.z;
}
MVS_HD bool ray_tri_boxed(const Ray& r, V3 a, V3 e1, V3 e2, V3 lo, V3 hi) {
    const V3 pv = cross_fma(r.d, e2);
    const float det = dot_fma(e1, pv);
    const V3 tv = r.o - a;
    const float un = dot_fma(tv, pv);             // u det
    const V3 qv = cross_fma(tv, e1);
    const float vn = dot_fma(r.d, qv);            // v det
    const float tn = dot_fma(e2, qv);             // t det
    const float sg = det < 0.0f ? -1.0f : 1.0f;
    const float ad = det * sg, us = un * sg, vs = vn * sg, ts = tn * sg;
    const float w = ad - (us + vs);
    const float g1 = __builtin_fmaf(-r.tmin, ad, ts), g2 = __builtin_fmaf(r.tmax, ad, -ts);
    const bool pre = (ad > 0.0f) & (fminf(fminf(fminf(fminf(us, vs), w), g1), g2) >= 0.0f);   // & not &&: branch free
    if (!pre) return false;
    return ray_tri_point_in_box(r, tn, det, lo, hi);
}
MVS_HD bool ray_tri(const Ray& r, V3 a, V3 e1, V3 e2) {
    V3 lo, hi;
    tri_pad_box(a, e1, e2, r.pad, &lo, &hi);
    return ray_tri_boxed(r, a, e1, e2, lo, hi);
}

// mve::Image<uint8_t>::linear_at as used at texture_view.cpp:226-229,240-243
// (MVE is not available: clamp, bilinear weights w0*w2, w1*w2, w0*w3, w1*w3,
// round with +0.5f -- DESIGN.md "MVE semantics").
MVS_HD uint8_t linear_at(const uint8_t* img, int w, int h, int chans, float x, float y, int c) {
    x = smax(0.0f, smin((float)(w - 1), x));
    y = smax(0.0f, smin((float)(h - 1), y));
    const int floor_x = (int)x, floor_y = (int)y;
    const int floor_xp1 = imin(floor_x + 1, w - 1), floor_yp1 = imin(floor_y + 1, h - 1);
    const float w1 = x - (float)floor_x, w0 = 1.0f - w1;
    const float w3 = y - (float)floor_y, w2 = 1.0f - w3;
    const int rowstride = w * chans;
    const size_t row1 = (size_t)floor_y * rowstride, row2 = (size_t)floor_yp1 * rowstride;
    const int col1 = floor_x * chans, col2 = floor_xp1 * chans;
    const float v1 = img[row1 + col1 + c], v2 = img[row1 + col2 + c];
    const float v3 = img[row2 + col1 + c], v4 = img[row2 + col2 + c];
    return (uint8_t)(((v1 * (w0 * w2) + v2 * (w1 * w2)) + v3 * (w0 * w3)) + v4 * (w1 * w3) + 0.5f);
}

struct FaceInfoOut { float quality; float mean_color[3]; };

// TextureView::get_face_info (texture_view.cpp:134-251) with Tri (tri.cpp:12-24, tri.h:58-84), split into the pieces the
// two footprint walkers share -- every float expression exists once, in the reference's operation order:
//   foot_setup   projection, Tri (bounding box, area, detT), y-sort of the vertices, edge equations   (:139-180)
//   foot_row     the pixel span [xb, xe) of scan line y, or false where the reference `continue`s        (:187-200)
//   foot_inside  Tri::inside for the slow path (an edge slope is 0 / infinite)                          (:205, tri.h:58-77)
//   foot_finish  quality and mean colour from the sums, incl. the no-sample fallback                    (:222-250)
struct FootSetup {
    V2 p1, p2, p3;            // sorted by ascending y
    V2 t1, t2, t3;            // Tri keeps the unsorted points
    float detT, aabb_min_x, aabb_min_y, aabb_max_x, aabb_max_y, area;
    float m1, b1, m2, b2, m3, b3;
    bool fast;
};
MVS_HD void foot_setup_px(FootSetup& s);
MVS_HD void foot_setup(const ViewParams& view, V3 v1, V3 v2, V3 v3, FootSetup& s) {
    s.p1 = pixel_coords(view, v1); s.p2 = pixel_coords(view, v2); s.p3 = pixel_coords(view, v3);
    foot_setup_px(s);
}
// the part of foot_setup after the projection (p1 .. p3 given in pixel coordinates)
MVS_HD void foot_setup_px(FootSetup& s) {
    s.t1 = s.p1; s.t2 = s.p2; s.t3 = s.p3;
    const float T0 = s.t1.x - s.t3.x, T1 = s.t2.x - s.t3.x, T2 = s.t1.y - s.t3.y, T3 = s.t2.y - s.t3.y;
    s.detT = T0 * T3 - T2 * T1;
    s.aabb_min_x = smin(s.t1.x, smin(s.t2.x, s.t3.x)); s.aabb_min_y = smin(s.t1.y, smin(s.t2.y, s.t3.y));
    s.aabb_max_x = smax(s.t1.x, smax(s.t2.x, s.t3.x)); s.aabb_max_y = smax(s.t1.y, smax(s.t2.y, s.t3.y));
    const float ux = s.t2.x - s.t1.x, uy = s.t2.y - s.t1.y, vx = s.t3.x - s.t1.x, vy = s.t3.y - s.t1.y;
    s.area = 0.5f * fabsf(ux * vy - uy * vx);
}
// the part of the setup only a sampled footprint needs (texture_view.cpp:163-180)
MVS_HD void foot_edges(FootSetup& s) {
    while (true) {   // sort by ascending y (:163-167)
        if (s.p1.y <= s.p2.y) {
            if (s.p2.y <= s.p3.y) break;
            V2 t = s.p2; s.p2 = s.p3; s.p3 = t;
        } else { V2 t = s.p1; s.p1 = s.p2; s.p2 = t; }
    }
    s.m1 = (s.p1.y - s.p3.y) / (s.p1.x - s.p3.x); s.b1 = s.p1.y - s.m1 * s.p1.x;
    s.m2 = (s.p1.y - s.p2.y) / (s.p1.x - s.p2.x); s.b2 = s.p1.y - s.m2 * s.p1.x;
    s.m3 = (s.p2.y - s.p3.y) / (s.p2.x - s.p3.x); s.b3 = s.p2.y - s.m3 * s.p2.x;
    s.fast = isfinite(s.m1) && s.m2 != 0.0f && isfinite(s.m2) && s.m3 != 0.0f && isfinite(s.m3);
}
MVS_HD bool foot_row(const FootSetup& s, int y, int* xb, int* xe) {
    float min_x = s.aabb_min_x - 0.5f, max_x = s.aabb_max_x + 0.5f;
    const float cy = (float)y + 0.5f;
    if (s.fast) {
        min_x = (cy - s.b1) / s.m1;
        if (cy <= s.p2.y) max_x = (cy - s.b2) / s.m2;
        else max_x = (cy - s.b3) / s.m3;
        if (min_x >= max_x) { float t = min_x; min_x = max_x; max_x = t; }
        if (min_x < s.aabb_min_x || min_x > s.aabb_max_x) return false;
        if (max_x < s.aabb_min_x || max_x > s.aabb_max_x) return false;
    }
    *xb = (int)floorf(min_x + 0.5f);
    *xe = (int)ceilf(max_x - 0.5f);            // the reference loops while (float)x < ceilf(max_x - 0.5f): x < an integer-valued float
    return true;
}
MVS_HD bool foot_inside(const FootSetup& s, int x, int y) {   // Tri::inside (tri.h:58-77)
    const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
    const float dx = cx - s.t3.x, dy = cy - s.t3.y;
    const float alpha = ((s.t2.y - s.t3.y) * dx + (s.t3.x - s.t2.x) * dy) / s.detT;
    if (alpha < 0.0f || alpha > 1.0f) return false;
    const float beta = ((s.t3.y - s.t1.y) * dx + (s.t1.x - s.t3.x) * dy) / s.detT;
    if (beta < 0.0f || beta > 1.0f) return false;
    if (alpha + beta > 1.0f) return false;
    return true;
}
// the two expressions of foot_finish that depend on the pixel sums (num_samples > 0): texture_view.cpp:222-224 and :235-236
MVS_HD double foot_gmi_term(double gmi_sum, uint32_t num_samples, float area) { return (gmi_sum / (double)num_samples) * (double)area; }
MVS_HD float foot_mean(double col_sum, uint32_t num_samples) { return (float)(col_sum / (double)num_samples); }
template <int DATA_TERM, bool OUTLIER>
MVS_HD void foot_finish(const ViewParams& view, const FootSetup& s, uint32_t num_samples, double col0, double col1, double col2, double gmi, FaceInfoOut* out) {
    const int w = view.width, h = view.height;
    const uint8_t* image = view.rgb;
    const uint8_t* gimg = view.gmi;
    if (DATA_TERM == 1) {
        if (num_samples > 0) {
            gmi = foot_gmi_term(gmi, num_samples, s.area);
        } else {
            const double g1 = (double)linear_at(gimg, w, h, 1, s.p1.x, s.p1.y, 0) / 255.0;
            const double g2 = (double)linear_at(gimg, w, h, 1, s.p2.x, s.p2.y, 0) / 255.0;
            const double g3 = (double)linear_at(gimg, w, h, 1, s.p3.x, s.p3.y, 0) / 255.0;
            gmi = (((g1 + g2) + g3) / 3.0) * (double)s.area;
        }
    }
    if (OUTLIER) {
        if (num_samples > 0) {
            out->mean_color[0] = foot_mean(col0, num_samples);
            out->mean_color[1] = foot_mean(col1, num_samples);
            out->mean_color[2] = foot_mean(col2, num_samples);
        } else {
            for (int i = 0; i < 3; ++i) {
                const double c1 = (double)linear_at(image, w, h, 3, s.p1.x, s.p1.y, i) / 255.0;
                const double c2 = (double)linear_at(image, w, h, 3, s.p2.x, s.p2.y, i) / 255.0;
                const double c3 = (double)linear_at(image, w, h, 3, s.p3.x, s.p3.y, i) / 255.0;
                out->mean_color[i] = (float)(((c1 + c2) + c3) / 3.0);
            }
        }
    }
    out->quality = (DATA_TERM == 0) ? s.area : (float)gmi;
}

// Exactness certificate for pixel sums that were NOT accumulated in the reference's order (the footprint samplers of k_dc.hip add
// the u8 values as integers and divide once: sum_k = fl(S_k / 255.0)).  The reference adds the n quotients fl(u / 255.0) one by one
// in fp64; all terms are non-negative, so whatever the order its sum R lies within (1 +- 2^-53)^(n + 1) of sum_k.  The expressions
// above are a division by n > 0 and a product with the area > 0, each rounded once, then the conversion to float: with
// Q = expr(sum_k), expr(R) lies within (1 +- 2^-53)^(n + 5) of Q, and the conversion to float is monotone.  So if both ends of
// [Q (1 - eps), Q (1 + eps)], eps = (n + 12) 2^-53 rounded up to an even multiple (both factors exact; the slack covers the rounding
// of the two products), give the same float, expr(R) gives it too: (float)Q IS the reference's result bit for bit, and foot_finish
// (sum_k) returns exactly that.  False: the interval straddles a float rounding boundary (probability ~ n 2^-28) and the caller has
// to repeat the walk serially.  shift > 0 widens eps by that many bits (test hook).  num_samples == 0: the sums are not used at all.
MVS_HD bool foot_value_certified(double q, double dn, double up) { return (float)(q * dn) == (float)(q * up); }
template <int DATA_TERM, bool OUTLIER>
MVS_HD bool foot_sums_certified(const FootSetup& s, uint32_t num_samples, double col0, double col1, double col2, double gmi, int shift) {
    if (num_samples == 0) return true;
    const double eps = ldexp((double)((num_samples + 13u) & ~1u), -53 + shift), dn = 1.0 - eps, up = 1.0 + eps;
    bool ok = true;
    if (DATA_TERM == 1) ok = foot_value_certified(foot_gmi_term(gmi, num_samples, s.area), dn, up);
    if (OUTLIER) ok = ok && foot_value_certified(col0 / (double)num_samples, dn, up) && foot_value_certified(col1 / (double)num_samples, dn, up) &&
                      foot_value_certified(col2 / (double)num_samples, dn, up);
    return ok;
}

// Integer walk of a `fast` footprint for the gradient term (one thread per footprint): the same pixels as the scan-line loop of
// face_info below, read as aligned 32-bit words (four pixels; bytes outside the span masked) and added as integers -- what
// wave_info_kernel does with a lane group, here with one lane.  ROWS scan lines per iteration, so that as many loads are in flight
// where the serial walk waits for every pixel in turn (its fp64 sum is a dependent chain by definition).  The caller certifies the
// sums (foot_sums_certified) before using them.  gmi must be readable up to three bytes before / after a span (the context's padded
// buffer).
MVS_HD uint32_t bytes_sum4(uint32_t v, uint32_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sad_u8(v, 0u, acc);
#else
    return acc + (v & 0xFFu) + ((v >> 8) & 0xFFu) + ((v >> 16) & 0xFFu) + (v >> 24);
#endif
}
template <int ROWS = 2>
MVS_HD void foot_walk_gmi_words(const ViewParams& view, const FootSetup& s, uint32_t* num_samples, uint32_t* gmi_sum) {
    const uint8_t* gimg = view.gmi;
    const int w = view.width;
    uint32_t n = 0, g = 0;
    const int y_end = (int)ceilf(s.aabb_max_y);                 // (float)y < ceilf(max): y < an integer-valued float
    for (int y = (int)floorf(s.aabb_min_y); y < y_end; y += ROWS) {
        int xb[ROWS], xe[ROWS], x0[ROWS];
        const uint8_t* row0 = gimg + (size_t)y * w;
        bool more = false;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (y + r >= y_end || !foot_row(s, y + r, &xb[r], &xe[r]) || xe[r] <= xb[r]) { xb[r] = 0; xe[r] = 0; }
            x0[r] = xb[r] - (int)(reinterpret_cast<uintptr_t>(row0 + r * w + xb[r]) & 3u);   // the aligned word that holds the first pixel
            if (xe[r] <= xb[r]) x0[r] = xe[r];                 // an empty span (a skipped line, a line past y_end): nothing is fetched
            more = more || x0[r] < xe[r];
        }
        while (more) {
            uint32_t v[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) v[r] = (x0[r] < xe[r]) ? *reinterpret_cast<const uint32_t*>(row0 + r * w + x0[r]) : 0u;
            more = false;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                if (x0[r] < xe[r]) {
                    const int lo = xb[r] > x0[r] ? xb[r] - x0[r] : 0, hi = xe[r] - x0[r] < 4 ? xe[r] - x0[r] : 4;   // bytes [lo, hi) are pixels of the span
                    const uint32_t m = (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & ~((1u << (8 * lo)) - 1u);
                    g = bytes_sum4(v[r] & m, g);
                    n += (uint32_t)(hi - lo);
                }
                x0[r] += 4;
                more = more || x0[r] < xe[r];
            }
        }
    }
    *num_samples = n; *gmi_sum = g;
}

constexpr float FOOT_DEFERRED = -1.0f;   // quality marker: "sampled by the wave-per-footprint kernel" (qualities are >= 0)

// One sequential scan-line walk per (face, view): the fp64 accumulation order is the reference's, which is what makes
// qualities bit-exact.  DATA_TERM: 0 = area, 1 = gmi; OUTLIER: colours are accumulated iff true.  A footprint that has to
// be sampled and whose area exceeds defer_area is NOT walked here: quality = FOOT_DEFERRED.
template <int DATA_TERM, bool OUTLIER>
MVS_HD void face_info(const ViewParams& view, V3 v1, V3 v2, V3 v3, FaceInfoOut* out, float defer_area = INFINITY,
                      const double* u8_over_255 = nullptr /* optional table of (double)v / 255.0, v = 0..255: the same correctly rounded
                                                             quotients the walk would compute, looked up instead of divided */) {
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
    const bool sampling_necessary = (DATA_TERM != 0) || OUTLIER;

    if (sampling_necessary && s.area > 0.5f) {
        if (s.area > defer_area) { out->quality = FOOT_DEFERRED; return; }
        foot_edges(s);
        const float y_end = ceilf(s.aabb_max_y);
        for (int y = (int)floorf(s.aabb_min_y); (float)y < y_end; ++y) {
            int xb, xe;
            if (!foot_row(s, y, &xb, &xe)) continue;
            for (int x = xb; x < xe; ++x) {
                if (!s.fast && !foot_inside(s, x, y)) continue;
                const size_t pix = (size_t)x + (size_t)y * w;
                if (OUTLIER) {
                    const uint8_t c0 = image[pix * 3 + 0], c1 = image[pix * 3 + 1], c2 = image[pix * 3 + 2];
                    col0 += u8_over_255 ? u8_over_255[c0] : (double)c0 / 255.0;
                    col1 += u8_over_255 ? u8_over_255[c1] : (double)c1 / 255.0;
                    col2 += u8_over_255 ? u8_over_255[c2] : (double)c2 / 255.0;
                }
                if (DATA_TERM == 1) { const uint8_t g = gimg[pix]; gmi += u8_over_255 ? u8_over_255[g] : (double)g / 255.0; }
                ++num_samples;
            }
        }
    }
    foot_finish<DATA_TERM, OUTLIER>(view, s, num_samples, col0, col1, col2, gmi, out);
}

// mve::image::color_rgb_to_ycbcr<float> as applied at calculate_data_costs.cpp:225
MVS_HD void rgb_to_ycbcr(float* v) {
    const float r = v[0], g = v[1], b = v[2];
    v[0] = r * 0.299f + g * 0.587f + b * 0.114f;
    v[1] = r * -0.168736f + g * -0.331264f + b * 0.5f + 0.5f;
    v[2] = r * 0.5f + g * -0.418688f + b * -0.081312f + 0.5f;
}

// Luminance + Sobel of TextureView::generate_gradient_magnitude (texture_view.cpp:102-107):
// mve::image::desaturate<uint8_t>(DESATURATE_LUMINANCE) then sobel_edge<uint8_t>
// (MVE not available -- DESIGN.md "MVE semantics").
MVS_HD uint8_t luminance_u8(uint8_t r, uint8_t g, uint8_t b) {
    return (uint8_t)(0.30 * (double)r + (double)(0.59f * (float)g) + (double)(0.11f * (float)b));
}
// floor(min(255, sqrt(n))) for n >= 0, exact
MVS_HD uint8_t isqrt_clamp255(int n) {
    // m = min(n, 255^2) + 1/2 is exact in fp32 (17 bits).  For k^2 <= min(n, 255^2) < (k+1)^2:  k^2 + 1/2 <= m <= (k+1)^2 - 1/2, so
    // k + 1/(4k+1) < sqrt(m) < (k+1) - 1/(4k+4): at k <= 255 both gaps are a thousand fp32 ulps, and truncating ANY square root
    // accurate to a few ulps gives k -- perfect squares included, which is what the half is for.  On the device that root is
    // v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence the compile flags would expand sqrtf into.
    const float m = (float)(n < 255 * 255 ? n : 255 * 255) + 0.5f;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint8_t)(int)__builtin_amdgcn_sqrtf(m);
#else
    return (uint8_t)(int)sqrtf(m);
#endif
}

// Histogram::add_value bin index (histogram.cpp:28-30) for min = 0
MVS_HD uint32_t hist_bin(float value, float maxv, uint32_t num_bins) {
    const float clamped = smax(0.0f, smin(maxv, value));
    return (uint32_t)floorf(((clamped - 0.0f) / (maxv - 0.0f)) * (float)(num_bins - 1));
}

// Host-side constant for cull_pair: the smallest float c with
// !( (double)std::acos(c) > MATH_DEG2RAD(75.0f) ) -- calculate_data_costs.cpp:187 evaluated
// with the HOST's acosf (the one a CPU build of the reference would call).  Returns NaN if
// acosf is not monotone around the threshold.
inline float host_cos_limit() {
    const double limit = 75.0f * (3.14159265358979323846264338327950288 / 180.0);
    auto f = [](uint32_t u) { float x; memcpy(&x, &u, 4); return x; };
    auto culled = [&](uint32_t u) { return (double)acosf(f(u)) > limit; };
    uint32_t lo = 0x3E000000u /* 0.125: culled */, hi = 0x3F000000u /* 0.5: kept */;
    while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (culled(mid)) lo = mid; else hi = mid; }
    for (uint32_t k = 1; k <= 4096; ++k) if (!culled(hi - k) || culled(hi + k - 1)) return NAN;
    return f(hi);
}

}  // namespace mvs
