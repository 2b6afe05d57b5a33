// Synthetic scene generator (host C++, deterministic, seeded).
//
// Produces exactly the inputs the reference's hot path consumes:
//   * mesh arrays as read at libs/tex/calculate_data_costs.cpp:136-138
//     (faces 3F u32, vertices, per-face normals),
//   * the TextureView camera fields of libs/tex/texture_view.h:43-48
//     (pos, viewdir, 3x3 projection, 4x4 world_to_cam, width, height) plus an
//     in-memory RGB8 image (what TextureView::load_image would have decoded),
//   * the face adjacency lists with the ordering semantics of
//     libs/tex/build_adjacency_graph.cpp:16-53 + UniGraph::add_edge
//     (libs/tex/uni_graph.h:86-93).
//
// It is an INPUT PRODUCER shared by tests, bench.py and the oracle harness; it
// is neither part of the product's compute path nor of the oracle.  Only
// + - * / sqrt floor are used on floats so that results do not depend on libm.
//
// SURVEY.md section 8(d) fixes the recipe: geodesic icosphere of frequency n
// (F = 20 n^2), optional radial value-noise displacement, pinhole cameras on
// a Fibonacci sphere (or the 6 axis directions) looking at the origin,
// procedural photo-consistent RGB8 images without any (0,0,0) pixel.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <unordered_map>
#include <algorithm>

extern "C" {

typedef struct {
    uint32_t n_verts;
    uint32_t n_faces;
    float*    verts;    // 3 * n_verts
    uint32_t* faces;    // 3 * n_faces
    float*    normals;  // 3 * n_faces
    uint32_t* adj_ptr;  // n_faces + 1
    uint32_t* adj;      // adj_ptr[n_faces]
} synth_mesh;

typedef struct {
    float pos[3];
    float viewdir[3];
    float K[9];
    float w2c[16];
    int32_t width;
    int32_t height;
} synth_camera;

}  // extern "C"

namespace {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 normalized(V3 a) { float n = std::sqrt(dot(a, a)); return {a.x / n, a.y / n, a.z / n}; }

inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
inline uint32_t hash3(int32_t x, int32_t y, int32_t z, uint32_t seed) {
    uint32_t h = seed;
    h = hash32(h ^ (uint32_t)x * 0x9E3779B1U);
    h = hash32(h ^ (uint32_t)y * 0x85EBCA77U);
    h = hash32(h ^ (uint32_t)z * 0xC2B2AE3DU);
    return h;
}
// lattice value in [-1, 1]
inline float lattice(int32_t x, int32_t y, int32_t z, uint32_t seed) {
    return (float)(hash3(x, y, z, seed) & 0xFFFF) * (2.0f / 65535.0f) - 1.0f;
}
inline float smooth(float t) { return t * t * (3.0f - 2.0f * t); }
// trilinear value noise, deterministic (+,-,*,floor only)
float value_noise(V3 p, uint32_t seed) {
    float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    int32_t ix = (int32_t)fx, iy = (int32_t)fy, iz = (int32_t)fz;
    float tx = smooth(p.x - fx), ty = smooth(p.y - fy), tz = smooth(p.z - fz);
    float c000 = lattice(ix, iy, iz, seed),     c100 = lattice(ix + 1, iy, iz, seed);
    float c010 = lattice(ix, iy + 1, iz, seed), c110 = lattice(ix + 1, iy + 1, iz, seed);
    float c001 = lattice(ix, iy, iz + 1, seed), c101 = lattice(ix + 1, iy, iz + 1, seed);
    float c011 = lattice(ix, iy + 1, iz + 1, seed), c111 = lattice(ix + 1, iy + 1, iz + 1, seed);
    float x00 = c000 + (c100 - c000) * tx, x10 = c010 + (c110 - c010) * tx;
    float x01 = c001 + (c101 - c001) * tx, x11 = c011 + (c111 - c011) * tx;
    float y0 = x00 + (x10 - x00) * ty, y1 = x01 + (x11 - x01) * ty;
    return y0 + (y1 - y0) * tz;
}
float fbm3(V3 p, uint32_t seed) {
    float s = 0.0f, a = 1.0f, f = 2.0f;
    for (int o = 0; o < 3; ++o) {
        s += a * value_noise(p * f, seed + 977u * (uint32_t)o);
        a *= 0.5f; f *= 2.0f;
    }
    return s * (1.0f / 1.75f);
}
inline float tri_wave(float x) {  // period 1, range [-1, 1]
    float d = x - std::floor(x + 0.5f);
    return 4.0f * std::fabs(d) - 1.0f;
}

const float ICO_T = 1.6180339887498949f;
const V3 ICO_V[12] = {
    {-1, ICO_T, 0}, {1, ICO_T, 0}, {-1, -ICO_T, 0}, {1, -ICO_T, 0},
    {0, -1, ICO_T}, {0, 1, ICO_T}, {0, -1, -ICO_T}, {0, 1, -ICO_T},
    {ICO_T, 0, -1}, {ICO_T, 0, 1}, {-ICO_T, 0, -1}, {-ICO_T, 0, 1}};
const int ICO_F[20][3] = {
    {0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11},
    {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
    {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9},
    {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};

}  // namespace

extern "C" {

// Geodesic icosphere of frequency n: 20 n^2 faces, 10 n^2 + 2 vertices.
// displacement_amp == 0 -> plain unit sphere (BASELINE config 1);
// otherwise r = 1 + amp * fbm3(p * 1.5) (3 octaves of seeded value noise).
int synth_icosphere(uint32_t n, float displacement_amp, uint32_t seed, synth_mesh* out) {
    if (n == 0 || !out) return -1;
    const uint64_t F = 20ull * n * n, NV = 10ull * n * n + 2;
    if (F > 0xFFFFFFFFull) return -2;
    std::vector<V3> verts; verts.reserve(NV);
    std::vector<uint32_t> faces; faces.reserve(3 * F);
    // vertex dedupe: corners by base vertex id, edge points by (lo, hi, k-from-lo)
    std::unordered_map<uint64_t, uint32_t> shared;
    shared.reserve(30ull * n + 64);
    std::vector<uint32_t> grid((size_t)(n + 1) * (n + 1));
    auto norm_ico = [](V3 v) { return normalized(v); };
    for (int bf = 0; bf < 20; ++bf) {
        const int ia = ICO_F[bf][0], ib = ICO_F[bf][1], ic = ICO_F[bf][2];
        const V3 A = norm_ico(ICO_V[ia]), B = norm_ico(ICO_V[ib]), C = norm_ico(ICO_V[ic]);
        for (uint32_t i = 0; i <= n; ++i) {
            for (uint32_t j = 0; i + j <= n; ++j) {
                const uint32_t k = n - i - j;  // weights: A:k, B:i, C:j
                uint64_t key = ~0ull;
                if (i == 0 && j == 0) key = (uint64_t)ia << 48 | (uint64_t)ia << 32;
                else if (k == 0 && j == 0) key = (uint64_t)ib << 48 | (uint64_t)ib << 32;
                else if (k == 0 && i == 0) key = (uint64_t)ic << 48 | (uint64_t)ic << 32;
                else if (j == 0) {  // edge A-B, i steps from A
                    int lo = std::min(ia, ib), hi = std::max(ia, ib);
                    uint32_t t = (lo == ia) ? i : n - i;
                    key = (uint64_t)lo << 48 | (uint64_t)hi << 32 | t;
                } else if (i == 0) {  // edge A-C, j steps from A
                    int lo = std::min(ia, ic), hi = std::max(ia, ic);
                    uint32_t t = (lo == ia) ? j : n - j;
                    key = (uint64_t)lo << 48 | (uint64_t)hi << 32 | t;
                } else if (k == 0) {  // edge B-C, j steps from B
                    int lo = std::min(ib, ic), hi = std::max(ib, ic);
                    uint32_t t = (lo == ib) ? j : n - j;
                    key = (uint64_t)lo << 48 | (uint64_t)hi << 32 | t;
                }
                uint32_t idx;
                bool fresh = true;
                if (key != ~0ull) {
                    auto it = shared.find(key);
                    if (it != shared.end()) { idx = it->second; fresh = false; }
                }
                if (fresh) {
                    V3 p = A * ((float)k / (float)n) + B * ((float)i / (float)n) + C * ((float)j / (float)n);
                    p = normalized(p);
                    idx = (uint32_t)verts.size();
                    verts.push_back(p);
                    if (key != ~0ull) shared.emplace(key, idx);
                }
                grid[(size_t)i * (n + 1) + j] = idx;
            }
        }
        auto G = [&](uint32_t i, uint32_t j) { return grid[(size_t)i * (n + 1) + j]; };
        for (uint32_t i = 0; i < n; ++i) {
            for (uint32_t j = 0; i + j < n; ++j) {
                uint32_t a = G(i, j), b = G(i + 1, j), c = G(i, j + 1);
                faces.push_back(a); faces.push_back(b); faces.push_back(c);
                if (i + j + 1 < n) {
                    uint32_t d = G(i + 1, j + 1);
                    faces.push_back(b); faces.push_back(d); faces.push_back(c);
                }
            }
        }
    }
    if (verts.size() != NV || faces.size() != 3 * F) return -3;
    if (displacement_amp != 0.0f) {
        for (auto& p : verts) {
            float r = 1.0f + displacement_amp * fbm3(p * 1.5f, seed);
            p = p * r;
        }
    }
    out->n_verts = (uint32_t)NV;
    out->n_faces = (uint32_t)F;
    out->verts = (float*)malloc(sizeof(float) * 3 * NV);
    out->faces = (uint32_t*)malloc(sizeof(uint32_t) * 3 * F);
    out->normals = (float*)malloc(sizeof(float) * 3 * F);
    memcpy(out->verts, verts.data(), sizeof(float) * 3 * NV);
    // outward CCW winding + face normals = normalised (b-a) x (c-a)
    for (uint64_t f = 0; f < F; ++f) {
        uint32_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
        V3 pa = verts[a], pb = verts[b], pc = verts[c];
        V3 nrm = cross(pb - pa, pc - pa);
        if (dot(nrm, pa + pb + pc) < 0.0f) {
            std::swap(b, c); std::swap(pb, pc);
            nrm = cross(pb - pa, pc - pa);
        }
        nrm = normalized(nrm);
        out->faces[3 * f] = a; out->faces[3 * f + 1] = b; out->faces[3 * f + 2] = c;
        out->normals[3 * f] = nrm.x; out->normals[3 * f + 1] = nrm.y; out->normals[3 * f + 2] = nrm.z;
    }
    out->adj_ptr = nullptr; out->adj = nullptr;
    return 0;
}

// Face adjacency with the semantics of build_adjacency_graph.cpp:16-53:
// for face i (ascending), for its edges (v1,v2),(v2,v3),(v3,v1), every OTHER
// face sharing that edge gets an undirected graph edge unless already present;
// UniGraph::add_edge appends to BOTH lists, so list order = global insertion order.
int synth_build_adjacency(synth_mesh* m) {
    const uint32_t F = m->n_faces;
    std::unordered_map<uint64_t, std::vector<uint32_t>> edge_faces;
    edge_faces.reserve((size_t)F * 2);
    auto ekey = [](uint32_t a, uint32_t b) {
        return (uint64_t)std::min(a, b) << 32 | std::max(a, b);
    };
    for (uint32_t f = 0; f < F; ++f) {
        const uint32_t* v = m->faces + 3 * (size_t)f;
        edge_faces[ekey(v[0], v[1])].push_back(f);
        edge_faces[ekey(v[1], v[2])].push_back(f);
        edge_faces[ekey(v[2], v[0])].push_back(f);
    }
    std::vector<std::vector<uint32_t>> lists(F);
    for (uint32_t f = 0; f < F; ++f) {
        const uint32_t* v = m->faces + 3 * (size_t)f;
        const uint64_t keys[3] = {ekey(v[0], v[1]), ekey(v[1], v[2]), ekey(v[2], v[0])};
        for (int e = 0; e < 3; ++e) {
            for (uint32_t g : edge_faces[keys[e]]) {
                if (g == f) continue;
                auto& lf = lists[f];
                if (std::find(lf.begin(), lf.end(), g) != lf.end()) continue;
                lf.push_back(g);
                lists[g].push_back(f);
            }
        }
    }
    m->adj_ptr = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)F + 1));
    uint64_t total = 0;
    for (uint32_t f = 0; f < F; ++f) { m->adj_ptr[f] = (uint32_t)total; total += lists[f].size(); }
    m->adj_ptr[F] = (uint32_t)total;
    m->adj = (uint32_t*)malloc(sizeof(uint32_t) * std::max<uint64_t>(total, 1));
    for (uint32_t f = 0; f < F; ++f)
        std::copy(lists[f].begin(), lists[f].end(), m->adj + m->adj_ptr[f]);
    return 0;
}

void synth_mesh_free(synth_mesh* m) {
    free(m->verts); free(m->faces); free(m->normals); free(m->adj_ptr); free(m->adj);
    memset(m, 0, sizeof(*m));
}

// Pinhole camera at `pos` looking at the origin.  MVE camera convention
// (x right, y down, z forward); world_to_cam = [R | -R pos]; viewdir = third
// row of R; K = [f 0 w/2; 0 f h/2; 0 0 1] (what CameraInfo::fill_calibration
// yields for paspect = 1, ppoint = (0.5, 0.5)).
static void make_camera(double px, double py, double pz, float focal_px, int w, int h,
                        synth_camera* c) {
    V3 pos = {(float)px, (float)py, (float)pz};
    V3 f = normalized(pos * -1.0f);
    V3 up = (std::fabs(f.z) > 0.99f) ? V3{0, 1, 0} : V3{0, 0, 1};
    V3 r = normalized(cross(f, up));
    V3 d = cross(f, r);
    const V3 rows[3] = {r, d, f};
    memset(c, 0, sizeof(*c));
    c->pos[0] = pos.x; c->pos[1] = pos.y; c->pos[2] = pos.z;
    c->viewdir[0] = f.x; c->viewdir[1] = f.y; c->viewdir[2] = f.z;
    for (int i = 0; i < 3; ++i) {
        c->w2c[4 * i + 0] = rows[i].x; c->w2c[4 * i + 1] = rows[i].y; c->w2c[4 * i + 2] = rows[i].z;
        c->w2c[4 * i + 3] = -dot(rows[i], pos);
    }
    c->w2c[15] = 1.0f;
    c->K[0] = focal_px; c->K[2] = 0.5f * (float)w;
    c->K[4] = focal_px; c->K[5] = 0.5f * (float)h;
    c->K[8] = 1.0f;
    c->width = w; c->height = h;
}

// layout 0: the 6 axis directions (n_views must be 6); layout 1: Fibonacci sphere.
// zoom_odd scales the focal length of odd-indexed cameras (> 1 crops the sphere, exercising the
// valid_pixel cull of calculate_data_costs.cpp:191).
int synth_cameras2(uint32_t n_views, int layout, float radius, int w, int h, float zoom_odd, float zoom, synth_camera* out) {
    // sphere of radius ~1.1 spans ~90 % of the short image side at zoom 1; zoom > 1 crops every view to a patch of the
    // surface (large footprints, few candidate views per face: the shape of a real capture)
    const float rs = 1.1f;
    const float tan_half = rs / std::sqrt(radius * radius - rs * rs);
    const float focal = (0.45f * (float)std::min(w, h) / tan_half) * zoom;
    if (layout == 0) {
        if (n_views != 6) return -1;
        const double ax[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
        for (int k = 0; k < 6; ++k)
            make_camera(ax[k][0] * radius, ax[k][1] * radius, ax[k][2] * radius, (k & 1) ? focal * zoom_odd : focal, w, h, out + k);
        return 0;
    }
    const double golden = 2.399963229728653;  // pi * (3 - sqrt(5))
    for (uint32_t k = 0; k < n_views; ++k) {
        double z = 1.0 - (2.0 * k + 1.0) / (double)n_views;
        double rr = std::sqrt(std::max(0.0, 1.0 - z * z));
        double phi = golden * (double)k;
        make_camera(radius * rr * std::cos(phi), radius * rr * std::sin(phi), radius * z, (k & 1) ? focal * zoom_odd : focal, w, h, out + k);
    }
    return 0;
}

int synth_cameras(uint32_t n_views, int layout, float radius, int w, int h, float zoom_odd, synth_camera* out) {
    return synth_cameras2(n_views, layout, radius, w, h, zoom_odd, 1.0f, out);
}

// Procedural RGB8 image for one camera: a colour field defined on the unit
// sphere (so views are photo-consistent), seen through the pinhole, plus
// per-pixel sensor noise; background is dim hash noise.  No pixel is (0,0,0)
// (the validity flood fill of texture_view.cpp:42-94 would eat those).
// black_corner > 0 paints a black square of that size into the (0,0) corner
// to exercise the flood fill / erosion path.
void synth_render(const synth_camera* cam, uint32_t view_index, uint32_t seed, int black_corner,
                  uint8_t* rgb) {
    const int w = cam->width, h = cam->height;
    const float f = cam->K[0], cx = cam->K[2], cy = cam->K[5];
    const V3 o = {cam->pos[0], cam->pos[1], cam->pos[2]};
    const V3 r0 = {cam->w2c[0], cam->w2c[1], cam->w2c[2]};
    const V3 r1 = {cam->w2c[4], cam->w2c[5], cam->w2c[6]};
    const V3 r2 = {cam->w2c[8], cam->w2c[9], cam->w2c[10]};
    const float oo = dot(o, o) - 1.0f;
    // small images are rendered serially: waking a team of hundreds of threads per call costs far more than the pixels
#pragma omp parallel for schedule(static) if ((size_t)w * (size_t)h >= 262144)
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            uint8_t* px = rgb + ((size_t)y * w + x) * 3;
            if (x < black_corner && y < black_corner) { px[0] = px[1] = px[2] = 0; continue; }
            const float dx = ((float)x + 0.5f - cx) / f, dy = ((float)y + 0.5f - cy) / f;
            const V3 d = r0 * dx + r1 * dy + r2;  // R^T (dx, dy, 1)
            const float a = dot(d, d), b = 2.0f * dot(o, d);
            const float disc = b * b - 4.0f * a * oo;
            const uint32_t hn = hash3(x, y, (int32_t)view_index, seed ^ 0x5bd1e995U);
            if (disc > 0.0f) {
                const float s = (-b - std::sqrt(disc)) / (2.0f * a);
                const V3 p = o + d * s;
                const float detail = value_noise(p * 48.0f, seed + 17u);
                const float coarse = value_noise(p * 6.0f, seed + 29u);
                float c[3];
                c[0] = 128.0f + 50.0f * tri_wave(1.5f * p.x + 0.10f) + 30.0f * coarse + 42.0f * detail;
                c[1] = 128.0f + 50.0f * tri_wave(1.5f * p.y + 0.35f) - 30.0f * coarse + 42.0f * detail;
                c[2] = 128.0f + 50.0f * tri_wave(1.5f * p.z + 0.60f) + 20.0f * coarse - 42.0f * detail;
                for (int k = 0; k < 3; ++k) {
                    float v = c[k] + (float)((hn >> (8 * k)) & 7u) - 3.5f;
                    v = std::min(255.0f, std::max(16.0f, v));
                    px[k] = (uint8_t)v;
                }
            } else {
                px[0] = (uint8_t)(40u + (hn & 63u));
                px[1] = (uint8_t)(40u + ((hn >> 8) & 63u));
                px[2] = (uint8_t)(40u + ((hn >> 16) & 63u));
            }
        }
    }
}

}  // extern "C"
