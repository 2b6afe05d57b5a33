// oracle/_ref -- the REFERENCE's own code for the pieces of the path that compile without MVE / rayint / Eigen /
// mapMAP: Histogram (libs/tex/histogram.{h,cpp}), UniGraph (libs/tex/uni_graph.{h,cpp}), SparseTable
// (libs/tex/sparse_table.h), Tri (libs/tex/tri.{h,cpp}, rect.h), TextureView's mask / valid_pixel / get_face_info logic
// (libs/tex/texture_view.{h,cpp}), the binary vector files of util.h and the Settings defaults (libs/tex/settings.h).  Their sources are compiled where they lie
// under /root/reference (oracle/Makefile, target `ref`); this file only adds extern "C" entry points so that the tests
// can pin the oracle's restatements of SURVEY.md rows C (Tri), D2, E, G / f3 and the defaults against the real thing.
// TEST INFRASTRUCTURE ONLY (never loaded by the product).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "settings.h"
#include "histogram.h"
#include "uni_graph.h"
#include "sparse_table.h"
#include "tri.h"
#include "texture_view.h"
#include "util.h"
#include <mve/image_tools.h>
#include <cstdio>

typedef SparseTable<std::uint32_t, std::uint16_t, float> RefDataCosts;   // == tex::DataCosts (libs/tex/texturing.h:36)

extern "C" {

// postprocess_face_infos' percentile (calculate_data_costs.cpp:283-288): Histogram(0, max, bins), add_value, get_approx_percentile
float ref_percentile(const float* v, std::uint64_t n, float min, float max, std::uint32_t bins, float percentile) {
    Histogram h(min, max, bins);
    for (std::uint64_t i = 0; i < n; ++i) h.add_value(v[i]);
    return h.get_approx_percentile(percentile);
}

// tex::Settings as default constructed (settings.h:85-95)
void ref_settings_defaults(std::int32_t out[3]) {
    tex::Settings s;
    out[0] = (std::int32_t)s.data_term; out[1] = (std::int32_t)s.outlier_removal; out[2] = s.geometric_visibility_test ? 1 : 0;
}

// A UniGraph whose adjacency lists equal the given CSR lists, built ONLY through add_edge: face i adds its larger
// neighbours in its list order (smaller neighbours were added when they were processed) -- the insertion pattern of
// build_adjacency_graph.cpp:31-47.  Dumps the resulting lists so the caller can check the replay reproduced them.
static UniGraph make_graph(std::uint32_t F, const std::uint32_t* adj_ptr, const std::uint32_t* adj) {
    UniGraph g(F);
    for (std::uint32_t i = 0; i < F; ++i)
        for (std::uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e)
            if (adj[e] > i) g.add_edge(i, adj[e]);
    return g;
}
std::uint64_t ref_unigraph_lists(std::uint32_t F, const std::uint32_t* adj_ptr, const std::uint32_t* adj,
                                 std::uint32_t* out_ptr /* F + 1 */, std::uint32_t* out_adj /* adj_ptr[F] */) {
    const UniGraph g = make_graph(F, adj_ptr, adj);
    std::uint32_t n = 0;
    for (std::uint32_t i = 0; i < F; ++i) {
        out_ptr[i] = n;
        std::vector<std::size_t> const& l = g.get_adj_nodes(i);
        for (std::size_t k = 0; k < l.size(); ++k) { if (n < adj_ptr[F]) out_adj[n] = (std::uint32_t)l[k]; ++n; }
    }
    out_ptr[F] = n;
    return g.num_edges();
}

// UniGraph::get_subgraphs(label) (uni_graph.cpp:21-55): components in the reference's order, faces in BFS queue order
std::uint32_t ref_get_subgraphs(std::uint32_t F, const std::uint32_t* adj_ptr, const std::uint32_t* adj, const std::uint32_t* labels,
                                std::uint32_t label, std::uint32_t* comp_ptr /* up to F + 1 */, std::uint32_t* comp_faces /* up to F */) {
    UniGraph g = make_graph(F, adj_ptr, adj);
    for (std::uint32_t i = 0; i < F; ++i) g.set_label(i, labels[i]);
    std::vector<std::vector<std::size_t> > sub;
    g.get_subgraphs(label, &sub);
    std::uint32_t n = 0;
    for (std::size_t c = 0; c < sub.size(); ++c) {
        comp_ptr[c] = n;
        for (std::size_t k = 0; k < sub[c].size(); ++k) comp_faces[n++] = (std::uint32_t)sub[c][k];
    }
    comp_ptr[sub.size()] = n;
    return (std::uint32_t)sub.size();
}

// Tri (tri.{h,cpp}): out = {get_area, aabb min_x, min_y, max_x, max_y}; inside[k] = Tri::inside(xy[2k], xy[2k+1])
void ref_tri(const float p[6], float out[5], const float* xy, std::uint32_t n, std::uint8_t* inside) {
    const Tri tri(math::Vec2f(p[0], p[1]), math::Vec2f(p[2], p[3]), math::Vec2f(p[4], p[5]));
    const Rect<float> bb = tri.get_aabb();
    out[0] = tri.get_area(); out[1] = bb.min_x; out[2] = bb.min_y; out[3] = bb.max_x; out[4] = bb.max_y;
    for (std::uint32_t k = 0; k < n; ++k) inside[k] = tri.inside(xy[2 * k], xy[2 * k + 1]) ? 1 : 0;
}

// ---- TextureView (texture_view.{h,cpp}) under an identity camera, image attached with bind_image ----
static tex::TextureView make_view(const std::uint8_t* rgb, int w, int h) {
    char name[64]; std::snprintf(name, sizeof(name), "%dx%d", w, h);
    tex::TextureView tv(0, mve::CameraInfo(), name);
    mve::ByteImage::Ptr img = mve::ByteImage::create(w, h, 3);
    std::memcpy(img->get_data_pointer(), rgb, (std::size_t)w * h * 3);
    tv.bind_image(img);
    return tv;
}
// out[k] = valid_pixel(xy[k]) after generate_validity_mask() (+ erode_validity_mask() if erode)   (texture_view.cpp:42-94,109-132,253-281)
void ref_valid_pixel_map(const std::uint8_t* rgb, int w, int h, int erode, const float* xy, std::uint32_t n, std::uint8_t* out) {
    tex::TextureView tv = make_view(rgb, w, h);
    tv.generate_validity_mask();
    if (erode) tv.erode_validity_mask();
    for (std::uint32_t k = 0; k < n; ++k) out[k] = tv.valid_pixel(math::Vec2f(xy[2 * k], xy[2 * k + 1])) ? 1 : 0;
}
// get_face_info (texture_view.cpp:134-251) of n triangles given by 3D vertices (9 floats each); gmi = the gradient
// magnitude image generate_gradient_magnitude() is to install (the Sobel arithmetic itself is MVE's and not pinned)
void ref_face_info(const std::uint8_t* rgb, const std::uint8_t* gmi, int w, int h, int data_term, int outlier, const float* verts, std::uint32_t n,
                   float* quality, float* color) {
    tex::TextureView tv = make_view(rgb, w, h);
    mve::ByteImage::Ptr g = mve::ByteImage::create(w, h, 1);
    std::memcpy(g->get_data_pointer(), gmi, (std::size_t)w * h);
    mve::image::next_gradient_magnitude() = g;
    tv.generate_gradient_magnitude();
    tex::Settings st;
    st.data_term = (tex::DataTerm)data_term; st.outlier_removal = (tex::OutlierRemoval)outlier;
    for (std::uint32_t k = 0; k < n; ++k) {
        const float* p = verts + 9 * (std::size_t)k;
        tex::FaceProjectionInfo info; info.view_id = 0; info.quality = 0.0f; info.mean_color = math::Vec3f(0.0f);
        tv.get_face_info(math::Vec3f(p[0], p[1], p[2]), math::Vec3f(p[3], p[4], p[5]), math::Vec3f(p[6], p[7], p[8]), &info, st);
        quality[k] = info.quality; for (int i = 0; i < 3; ++i) color[3 * k + i] = info.mean_color[i];
    }
}

// SparseTable::save_to_file / load_from_file (sparse_table.h:112-187) on a table filled by set_value in CSR order
int ref_spt_write(const char* path, std::uint32_t cols, std::uint16_t rows, const std::uint32_t* col_ptr, const std::uint16_t* view_id, const float* cost) {
    try {
        RefDataCosts t(cols, rows);
        for (std::uint32_t i = 0; i < cols; ++i)
            for (std::uint32_t k = col_ptr[i]; k < col_ptr[i + 1]; ++k) t.set_value(i, view_id[k], cost[k]);
        RefDataCosts::save_to_file(t, path);
    } catch (std::exception&) { return 1; }
    return 0;
}
// reads a .spt into CSR arrays (caller sized: col_ptr cols + 1, the others nnz_cap); returns nnz, or -1 on any exception
std::int64_t ref_spt_read(const char* path, std::uint32_t cols, std::uint16_t rows, std::uint32_t* col_ptr, std::uint16_t* view_id, float* cost,
                          std::uint64_t nnz_cap) {
    try {
        RefDataCosts t(cols, rows);
        RefDataCosts::load_from_file(path, &t);
        std::uint64_t n = 0;
        for (std::uint32_t i = 0; i < cols; ++i) {
            col_ptr[i] = (std::uint32_t)n;
            RefDataCosts::Column const& c = t.col(i);
            for (std::size_t k = 0; k < c.size(); ++k) { if (n < nnz_cap) { view_id[n] = c[k].first; cost[n] = c[k].second; } ++n; }
        }
        col_ptr[cols] = (std::uint32_t)n;
        return (std::int64_t)n;
    } catch (std::exception&) { return -1; }
}

// vector_to_file<std::size_t> / vector_from_file<std::size_t> (util.h:104-131): the labeling file of texrecon.cpp:130-136
int ref_vec_write(const char* path, const std::uint32_t* labels, std::uint32_t n) {
    try { std::vector<std::size_t> v(labels, labels + n); vector_to_file<std::size_t>(path, v); } catch (std::exception&) { return 1; }
    return 0;
}
std::int64_t ref_vec_read(const char* path, std::uint32_t* labels, std::uint32_t cap) {
    try {
        std::vector<std::size_t> v = vector_from_file<std::size_t>(path);
        for (std::size_t i = 0; i < v.size() && i < cap; ++i) labels[i] = (std::uint32_t)v[i];
        return (std::int64_t)v.size();
    } catch (std::exception&) { return -1; }
}

}  // extern "C"
