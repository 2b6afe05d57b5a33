// bvh_order.cpp -- EXPERIMENT tool (scripts/bvh_order_probe.py; not part of the product): triangle orders for the implicit 4-ary BVH of
// csrc/k_bvh.hip (leaves = 16 consecutive triangles, a level-k node = 4 consecutive level-(k-1) nodes), built on the host top-down:
// every node's range is cut at the multiples of its children's capacity (so that the partition IS the implicit tree), each binary cut
// along the axis that minimises the surface-area cost  SA(left box) * n_left + SA(right box) * n_right  (the SAH of a cut whose
// position the layout fixes), the triangles ordered by centroid along that axis.  mode 0 = the longest axis of the centroid box
// (plain object-median kd order), mode 1 = the SAH axis choice.
//   g++ -O2 -fopenmp -shared -fPIC -o libbvh_order.so bvh_order.cpp
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {
struct Tri { float c[3], lo[3], hi[3]; uint32_t id; };
struct Box { float lo[3], hi[3]; void clear() { for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
             void add(const Tri& t) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], t.lo[a]); hi[a] = std::max(hi[a], t.hi[a]); } }
             double area() const { const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2]; return (x < 0 || y < 0 || z < 0) ? 0.0 : 2.0 * (x * y + y * z + z * x); } };

void cut(Tri* t, size_t n, size_t at, int mode) {   // orders t so that t[0 .. at) / t[at .. n) are the two sides of the best cut at position `at`
    if (at == 0 || at >= n) return;
    int best = 0;
    if (mode == 0) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = 0; k < n; ++k) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], t[k].c[a]); hi[a] = std::max(hi[a], t[k].c[a]); }
        for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[best] - lo[best]) best = a;
    } else {
        double best_cost = INFINITY;
        std::vector<Tri> tmp(t, t + n);
        for (int a = 0; a < 3; ++a) {
            std::nth_element(tmp.begin(), tmp.begin() + at, tmp.end(), [a](const Tri& x, const Tri& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.id < y.id); });
            Box l, r; l.clear(); r.clear();
            for (size_t k = 0; k < at; ++k) l.add(tmp[k]);
            for (size_t k = at; k < n; ++k) r.add(tmp[k]);
            const double cost = l.area() * (double)at + r.area() * (double)(n - at);
            if (cost < best_cost) { best_cost = cost; best = a; }
        }
    }
    const int a = best;
    std::nth_element(t, t + at, t + n, [a](const Tri& x, const Tri& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.id < y.id); });
}
void build(Tri* t, size_t n, size_t cap, int mode) {   // node of capacity `cap` triangles holding n <= cap
    if (cap <= 16 || n <= 16) return;
    const size_t c = cap / 4;
    if (n <= c) { build(t, n, c, mode); return; }
    if (n > 2 * c) { cut(t, n, 2 * c, mode); cut(t, 2 * c, c, mode); if (n > 3 * c) cut(t + 2 * c, n - 2 * c, c, mode); }
    else cut(t, n, c, mode);
    const bool par = n > (1u << 14);
#pragma omp parallel for schedule(dynamic, 1) if (par)
    for (int k = 0; k < 4; ++k) { const size_t b = (size_t)k * c; if (b < n) build(t + b, std::min(c, n - b), c, mode); }
}
}  // namespace

extern "C" int bvh_order(uint32_t n_verts, const float* verts, uint32_t n_faces, const uint32_t* faces, int mode, uint32_t* perm_out) {
    (void)n_verts;
    std::vector<Tri> t(n_faces);
    for (uint32_t f = 0; f < n_faces; ++f) {
        Tri& x = t[f]; x.id = f;
        for (int a = 0; a < 3; ++a) {
            const float p = verts[3 * faces[3 * f] + a], q = verts[3 * faces[3 * f + 1] + a], r = verts[3 * faces[3 * f + 2] + a];
            x.c[a] = (p + q + r) * (1.0f / 3.0f); x.lo[a] = std::min(p, std::min(q, r)); x.hi[a] = std::max(p, std::max(q, r));
        }
    }
    size_t cap = 16; while (cap < n_faces) cap *= 4;
    build(t.data(), n_faces, cap, mode);
    for (uint32_t k = 0; k < n_faces; ++k) perm_out[k] = t[k].id;
    return 0;
}
