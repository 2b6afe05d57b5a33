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
void build(Tri* t, size_t n, size_t cap, int mode, size_t stop = 16) {   // node of capacity `cap` elements holding n <= cap; cuts down to segments of `stop`
    if (cap <= stop || n <= stop) return;
    const size_t c = cap / 4;
    if (n <= c) { build(t, n, c, mode, stop); return; }
    if (n > 2 * c) { cut(t, n, 2 * c, mode); cut(t, 2 * c, c, mode); if (n > 3 * c) cut(t + 2 * c, n - 2 * c, c, mode); }
    else cut(t, n, c, mode);
    const bool par = n > (1u << 14);
#pragma omp parallel for schedule(dynamic, 1) if (par)
    for (int k = 0; k < 4; ++k) { const size_t b = (size_t)k * c; if (b < n) build(t + b, std::min(c, n - b), c, mode, stop); }
}
// one pass over SUPER-ELEMENTS: aligned groups of `se` consecutive elements of t (a whole number of implicit subtrees) move as units inside
// aligned windows of `win` groups, cut top-down to segments of `stop` groups; a partial last group stays last (its key is +inf on the GPU)
void se_pass(std::vector<Tri>& t, size_t se, size_t win, size_t stop, int mode) {
    const size_t n = t.size(), n_se = (n + se - 1) / se, n_full = n / se;
    std::vector<Tri> g(n_se);
    for (size_t k = 0; k < n_se; ++k) {
        Tri& x = g[k]; x.id = (uint32_t)k;
        double c[3] = {0, 0, 0}; size_t m = 0;
        for (int a = 0; a < 3; ++a) { x.lo[a] = INFINITY; x.hi[a] = -INFINITY; }
        for (size_t i = k * se; i < std::min(n, (k + 1) * se); ++i, ++m) for (int a = 0; a < 3; ++a) { c[a] += t[i].c[a]; x.lo[a] = std::min(x.lo[a], t[i].lo[a]); x.hi[a] = std::max(x.hi[a], t[i].hi[a]); }
        for (int a = 0; a < 3; ++a) x.c[a] = (float)(c[a] / (double)std::max<size_t>(m, 1));
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (long long b = 0; b < (long long)n_full; b += (long long)win) build(g.data() + b, std::min<size_t>(win, n_full - b), win, mode, stop);   // (the partial group, if any, is not among the first n_full)
    std::vector<Tri> out; out.reserve(n);
    for (size_t k = 0; k < n_se; ++k) { const size_t src = g[k].id; for (size_t i = src * se; i < std::min(n, (src + 1) * se); ++i) out.push_back(t[i]); }
    t.swap(out);
}
}  // namespace

// ---- sample tree: the top levels from a SAMPLE of the triangles (what one GPU block can hold in LDS), everybody else descends its planes ----
struct Plane { int axis; float thr; int left, right; };   // children: >= 0 = plane index, < 0 = ~cell
void sample_tree(Tri* t, size_t n, size_t stop, std::vector<Plane>& planes, int& n_cells, int self) {
    // halving cuts along the longest centroid axis down to cells of <= stop samples
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t k = 0; k < n; ++k) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], t[k].c[a]); hi[a] = std::max(hi[a], t[k].c[a]); }
    int ax = 0; for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
    const size_t h = n / 2;
    std::nth_element(t, t + h, t + n, [ax](const Tri& x, const Tri& y) { return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.id < y.id); });
    planes[self].axis = ax; planes[self].thr = t[h].c[ax];
    if (h <= stop) planes[self].left = ~(n_cells++); else { planes.push_back(Plane()); const int c = (int)planes.size() - 1; planes[self].left = c; sample_tree(t, h, stop, planes, n_cells, c); }
    if (n - h <= stop) planes[self].right = ~(n_cells++); else { planes.push_back(Plane()); const int c = (int)planes.size() - 1; planes[self].right = c; sample_tree(t + h, n - h, stop, planes, n_cells, c); }
}

// window_cap > 0: the cuts are made only INSIDE aligned windows of window_cap (= 16 * 4^k) triangles of the order `base` (null: the caller's
// order) -- what a GPU pass that holds one window in LDS could do on top of the library's Hilbert order; 0 = the whole tree top-down
extern "C" int bvh_order(uint32_t n_verts, const float* verts, uint32_t n_faces, const uint32_t* faces, int mode, const uint32_t* base, uint32_t window_cap, uint32_t* perm_out) {
    (void)n_verts;
    std::vector<Tri> t(n_faces);
    for (uint32_t k = 0; k < n_faces; ++k) {
        const uint32_t f = base ? base[k] : k;
        Tri& x = t[k]; x.id = f;
        for (int a = 0; a < 3; ++a) {
            const float p = verts[3 * faces[3 * f] + a], q = verts[3 * faces[3 * f + 1] + a], r = verts[3 * faces[3 * f + 2] + a];
            x.c[a] = (p + q + r) * (1.0f / 3.0f); x.lo[a] = std::min(p, std::min(q, r)); x.hi[a] = std::max(p, std::max(q, r));
        }
    }
    if (mode >= 20) {   // top levels from a sample of window_cap triangles (cells of 16 samples), cell-major order (inside a cell: the base order), then faces inside windows of 4096
        const size_t ns = std::min<size_t>(window_cap, n_faces);
        std::vector<Tri> smp(ns);
        for (size_t k = 0; k < ns; ++k) smp[k] = t[(size_t)((double)k * n_faces / ns)];
        std::vector<Plane> planes(1); planes.reserve(4 * ns); int n_cells = 0;
        sample_tree(smp.data(), ns, 16, planes, n_cells, 0);
        std::vector<std::pair<uint64_t, uint32_t> > key(n_faces);
        for (uint32_t k = 0; k < n_faces; ++k) {
            int p = 0;
            while (p >= 0) { const Plane& pl = planes[p]; p = (t[k].c[pl.axis] < pl.thr) ? pl.left : pl.right; }
            key[k] = std::make_pair(((uint64_t)(uint32_t)~p << 32) | k, k);
        }
        std::sort(key.begin(), key.end());
        std::vector<Tri> o(n_faces);
        for (uint32_t k = 0; k < n_faces; ++k) o[k] = t[key[k].second];
        t.swap(o);
#pragma omp parallel for schedule(dynamic, 4)
        for (long long b = 0; b < (long long)n_faces; b += 4096) build(t.data() + b, std::min<size_t>(4096, n_faces - b), 4096, 0);
    } else if (mode >= 10) {   // hierarchical passes over the order `base`: groups of 4096 over everything, groups of 64 inside windows of 262144, triangles inside windows of 4096
        const int m = mode - 10;
        size_t top = 64; while (top * 4096 < n_faces) top *= 4;
        se_pass(t, 4096, top, 64, m);
        se_pass(t, 64, 4096, 64, m);
#pragma omp parallel for schedule(dynamic, 4)
        for (long long b = 0; b < (long long)n_faces; b += 4096) build(t.data() + b, std::min<size_t>(4096, n_faces - b), 4096, m);
    } else if (window_cap) {
#pragma omp parallel for schedule(dynamic, 4)
        for (long long b = 0; b < (long long)n_faces; b += window_cap) build(t.data() + b, std::min<size_t>(window_cap, n_faces - b), window_cap, mode);
    } else {
        size_t cap = 16; while (cap < n_faces) cap *= 4;
        build(t.data(), n_faces, cap, mode);
    }
    for (uint32_t k = 0; k < n_faces; ++k) perm_out[k] = t[k].id;
    return 0;
}
