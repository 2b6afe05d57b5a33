// oracle/_ref -- the REFERENCE's own code for the path and its neighbours: tex::view_selection's model construction and
// decode (libs/tex/view_selection.cpp), tex::calculate_data_costs
// with photometric_outlier_detection, calculate_face_projection_infos and postprocess_face_infos
// (libs/tex/calculate_data_costs.cpp), TextureView (libs/tex/texture_view.{h,cpp}), Tri (libs/tex/tri.{h,cpp}, rect.h),
// Histogram (libs/tex/histogram.{h,cpp}), UniGraph (libs/tex/uni_graph.{h,cpp}), SparseTable (libs/tex/sparse_table.h),
// prepare_mesh / build_adjacency_graph (libs/tex/prepare_mesh.cpp, build_adjacency_graph.cpp),
// the binary vector files of util.h and the Settings defaults (libs/tex/settings.h).  The sources are compiled where they
// lie under /root/reference (oracle/Makefile, target `ref`) against oracle/ref_stubs, which stands in for the headers
// of the absent libraries (MVE, rayint, Eigen, mapMAP): containers, recorders, and the oracle's definitions of their arithmetic.  This file
// only adds extern "C" entry points so that the tests can pin the oracle's restatements of SURVEY.md rows A, B, C, D, D1,
// D2, E, F, G / f3, H, f1 and the defaults against the real thing.
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
#include "texturing.h"
#include <mve/image_tools.h>
#include <mve/image_io.h>
#include <acc/bvh_tree.h>
#include "mapmap/full.h"
#include <cstdio>

// defined (not static) in libs/tex/calculate_data_costs.cpp:35 but declared in no header
// (MVS_DROPIN_BUILD: this file compiled against integration/view_selection_mi355x.cpp instead of upstream's two sources)
namespace tex { bool photometric_outlier_detection(std::vector<FaceProjectionInfo>* infos, Settings const& settings); }

typedef SparseTable<std::uint32_t, std::uint16_t, float> RefDataCosts;   // == tex::DataCosts (libs/tex/texturing.h:36)

#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

// threads of the reference's own OpenMP loops (calculate_data_costs.cpp:148-153,260) in the OpenMP build of this library
// (oracle/Makefile target `ref_omp`: bench.py's baseline leg); returns the count in effect (1 in the serial build)
int ref_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

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

// ---- the path's data-cost half, whole: tex::calculate_data_costs (calculate_data_costs.cpp:308-323) = the culls and
// the ray set-up / order / early exit of calculate_face_projection_infos (:131-251), get_face_info, the colour-space change,
// photometric_outlier_detection, the zero-quality erase, the sort, max / histogram / percentile and the final normalisation
// of postprocess_face_infos (:253-306) -- the reference's own code, compiled without OpenMP (= its --num_threads 1 order).
// Supplied by the test rather than computed here, because they are arithmetic of ABSENT libraries: the per-view camera
// arrays (mve::CameraInfo), the gradient-magnitude images (mve desaturate + sobel_edge) and the boolean answer to each
// any-hit ray (rayint), which the stand-in acc::BVHTree forwards to `ray_fn` with the ray exactly as the reference set it up.
struct RefView { float pos[3], viewdir[3], K[9], w2c[16]; std::int32_t width, height; const std::uint8_t* rgb; };   // layout of orc_view
std::int64_t ref_calculate_data_costs(std::uint32_t n_verts, const float* verts, std::uint32_t n_faces, const std::uint32_t* faces,
                                      const float* face_normals, const RefView* views, const std::uint8_t* const* gmi, std::uint32_t n_views,
                                      int data_term, int outlier_removal, int geometric_visibility_test,
                                      int (*ray_fn)(const void*, const void*, const float*, const float*, float, float, int),
                                      const void* ray_bvh, const void* ray_mesh, int ray_brute,
                                      std::uint32_t* col_ptr /* n_faces + 1 */, std::uint16_t* view_id, float* cost, std::uint64_t cap,
                                      std::uint64_t* rays_cast) {
    try {
        mve::TriangleMesh::Ptr mesh = mve::TriangleMesh::create();
        for (std::uint32_t v = 0; v < n_verts; ++v) mesh->get_vertices().push_back(math::Vec3f(verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]));
        mesh->get_faces().assign(faces, faces + 3 * (std::size_t)n_faces);
        for (std::uint32_t f = 0; f < n_faces; ++f)
            mesh->get_face_normals().push_back(math::Vec3f(face_normals[3 * f], face_normals[3 * f + 1], face_normals[3 * f + 2]));

        mve::image::file_registry().clear();
        mve::image::gradient_registry().clear();
        std::vector<tex::TextureView> texture_views;
        for (std::uint32_t j = 0; j < n_views; ++j) {
            RefView const& rv = views[j];
            mve::CameraInfo cam;
            std::memcpy(cam.K, rv.K, sizeof(cam.K)); std::memcpy(cam.w2c, rv.w2c, sizeof(cam.w2c));
            std::memcpy(cam.pos, rv.pos, sizeof(cam.pos)); std::memcpy(cam.dir, rv.viewdir, sizeof(cam.dir));
            char name[64]; std::snprintf(name, sizeof(name), "%dx%d#%u", rv.width, rv.height, j);
            mve::ByteImage::Ptr img = mve::ByteImage::create(rv.width, rv.height, 3);
            std::memcpy(img->get_data_pointer(), rv.rgb, (std::size_t)rv.width * rv.height * 3);
            mve::image::file_registry()[name] = img;
            if (gmi && gmi[j]) {
                mve::ByteImage::Ptr g = mve::ByteImage::create(rv.width, rv.height, 1);
                std::memcpy(g->get_data_pointer(), gmi[j], (std::size_t)rv.width * rv.height);
                mve::image::gradient_registry()[img.get()] = g;
            }
            texture_views.push_back(tex::TextureView(j, cam, name));
        }
        acc::RayHook& hook = acc::ray_hook();
        hook.fn = ray_fn; hook.bvh = ray_bvh; hook.mesh = ray_mesh; hook.brute = ray_brute; hook.calls = 0;

        tex::Settings st;
        st.data_term = (tex::DataTerm)data_term; st.outlier_removal = (tex::OutlierRemoval)outlier_removal;
        st.geometric_visibility_test = geometric_visibility_test != 0;
        tex::DataCosts data_costs(n_faces, n_views);
        tex::calculate_data_costs(mesh, &texture_views, st, &data_costs);

        if (rays_cast) *rays_cast = hook.calls;
        hook.fn = nullptr;
        mve::image::file_registry().clear();
        mve::image::gradient_registry().clear();
        std::uint64_t n = 0;
        for (std::uint32_t i = 0; i < n_faces; ++i) {
            col_ptr[i] = (std::uint32_t)n;
            tex::DataCosts::Column const& c = data_costs.col(i);
            for (std::size_t k = 0; k < c.size(); ++k) { if (n < cap) { view_id[n] = c[k].first; cost[n] = c[k].second; } ++n; }
        }
        col_ptr[n_faces] = (std::uint32_t)n;
        return (std::int64_t)n;
    } catch (std::exception& e) { std::fprintf(stderr, "ref_calculate_data_costs: %s\n", e.what()); return -1; }
}

// Image lifetime inside tex::calculate_data_costs (calculate_data_costs.cpp:157 tv.load_image() ... :231 tv.release_image(): upstream holds ONE
// view's images at a time): n_views views of w x h over a one-triangle mesh, every load a fresh copy; view `fail_view` (< 0: none) cannot be
// loaded.  out[0] = most images alive at once during the call, beyond those alive before it; out[1] = views whose image is still loaded
// when the call has returned or thrown; returns 0, or 1 with the exception's text in err.
int ref_image_lifetime(std::uint32_t n_views, int w, int h, int fail_view, long* out, char* err, int err_cap) {
    mve::image::file_registry().clear(); mve::image::gradient_registry().clear(); mve::image::unreadable_files().clear();
    mve::TriangleMesh::Ptr mesh = mve::TriangleMesh::create();
    mesh->get_vertices().push_back(math::Vec3f(0.0f, 0.0f, 0.0f)); mesh->get_vertices().push_back(math::Vec3f(1.0f, 0.0f, 0.0f)); mesh->get_vertices().push_back(math::Vec3f(0.0f, 1.0f, 0.0f));
    const std::uint32_t tri[3] = {0, 1, 2};
    mesh->get_faces().assign(tri, tri + 3);
    mesh->get_face_normals().push_back(math::Vec3f(0.0f, 0.0f, 1.0f));
    std::vector<tex::TextureView> views;
    for (std::uint32_t j = 0; j < n_views; ++j) {
        mve::CameraInfo cam;
        const float K[9] = {(float)w, 0.0f, w * 0.5f, 0.0f, (float)w, h * 0.5f, 0.0f, 0.0f, 1.0f};
        const float w2c[16] = {1, 0, 0, -0.3f, 0, -1, 0, 0.3f, 0, 0, -1, 2.0f, 0, 0, 0, 1};     // a camera at (0.3, 0.3, 2) looking down -z
        const float pos[3] = {0.3f, 0.3f, 2.0f}, dir[3] = {0.0f, 0.0f, -1.0f};
        std::memcpy(cam.K, K, sizeof(K)); std::memcpy(cam.w2c, w2c, sizeof(w2c)); std::memcpy(cam.pos, pos, sizeof(pos)); std::memcpy(cam.dir, dir, sizeof(dir));
        char name[64]; std::snprintf(name, sizeof(name), "%dx%d#%u", w, h, j);
        mve::ByteImage::Ptr img = mve::ByteImage::create(w, h, 3);
        for (int p = 0; p < w * h * 3; ++p) img->get_data_pointer()[p] = (std::uint8_t)(40 + (p * 7 + j) % 200);
        mve::image::file_registry()[name] = img;
        if ((int)j == fail_view) mve::image::unreadable_files().insert(name);
        views.push_back(tex::TextureView(j, cam, name));
    }
    acc::RayHook& hook = acc::ray_hook();
    hook.fn = nullptr; hook.brute = 0; hook.calls = 0;
    tex::Settings st; st.geometric_visibility_test = false; st.data_term = tex::DATA_TERM_AREA;   // (the stand-in gradient images are keyed by the registered image: copies have none)
    tex::DataCosts data_costs(1, n_views);
    mve::image::load_copies() = true;
    const long before = mve::ImageCensus::live().load();
    mve::ImageCensus::peak().store(before);
    int rc = 0;
    try { tex::calculate_data_costs(mesh, &views, st, &data_costs); }
    catch (std::exception& e) { rc = 1; if (err && err_cap > 0) { std::strncpy(err, e.what(), (std::size_t)err_cap - 1); err[err_cap - 1] = 0; } }
    mve::image::load_copies() = false;
    out[0] = mve::ImageCensus::peak().load() - before;
    out[1] = mve::ImageCensus::live().load() - before;     // copies that are still bound to a view (get_image() asserts instead of answering)
    mve::image::file_registry().clear(); mve::image::gradient_registry().clear(); mve::image::unreadable_files().clear();
    return rc;
}

// tex::postprocess_face_infos (calculate_data_costs.cpp:253-306, exported at texturing.h:71-74) on caller-supplied infos,
// in the order given: the reference's own outlier detection / erase / sort / max / histogram / percentile / set_value
std::int64_t ref_postprocess_face_infos(std::uint32_t n_faces, std::uint32_t n_views, const std::uint32_t* info_ptr, const std::uint16_t* view_id,
                                        const float* quality, const float* mean_color, int outlier_removal,
                                        std::uint32_t* col_ptr /* n_faces + 1 */, std::uint16_t* view_out, float* cost_out, std::uint64_t cap) {
    try {
        tex::FaceProjectionInfos infos(n_faces);
        for (std::uint32_t i = 0; i < n_faces; ++i)
            for (std::uint32_t k = info_ptr[i]; k < info_ptr[i + 1]; ++k) {
                tex::FaceProjectionInfo fi;
                fi.view_id = view_id[k]; fi.quality = quality[k];
                fi.mean_color = math::Vec3f(mean_color[3 * k], mean_color[3 * k + 1], mean_color[3 * k + 2]);
                infos[i].push_back(fi);
            }
        tex::Settings st; st.outlier_removal = (tex::OutlierRemoval)outlier_removal;
        tex::DataCosts data_costs(n_faces, n_views);
        tex::postprocess_face_infos(st, &infos, &data_costs);
        std::uint64_t n = 0;
        for (std::uint32_t i = 0; i < n_faces; ++i) {
            col_ptr[i] = (std::uint32_t)n;
            tex::DataCosts::Column const& c = data_costs.col(i);
            for (std::size_t k = 0; k < c.size(); ++k) { if (n < cap) { view_out[n] = c[k].first; cost_out[n] = c[k].second; } ++n; }
        }
        col_ptr[n_faces] = (std::uint32_t)n;
        return (std::int64_t)n;
    } catch (std::exception& e) { std::fprintf(stderr, "ref_postprocess_face_infos: %s\n", e.what()); return -1; }
}

#ifndef MVS_DROPIN_BUILD   // (the drop-in build -- Makefile target `dropin` -- has no calculate_data_costs.cpp, hence no such internal symbol)
// photometric_outlier_detection (calculate_data_costs.cpp:35-129) on one face's infos in the order given
int ref_outlier_detection(std::uint32_t n, const float* mean_color, float* quality, int outlier_removal) {
    std::vector<tex::FaceProjectionInfo> infos(n);
    for (std::uint32_t i = 0; i < n; ++i) {
        infos[i].view_id = (std::uint16_t)i; infos[i].quality = quality[i];
        infos[i].mean_color = math::Vec3f(mean_color[3 * i], mean_color[3 * i + 1], mean_color[3 * i + 2]);
    }
    tex::Settings st; st.outlier_removal = (tex::OutlierRemoval)outlier_removal;
    const bool ok = tex::photometric_outlier_detection(&infos, st);
    for (std::uint32_t i = 0; i < n; ++i) quality[i] = infos[i].quality;
    return ok ? 1 : 0;
}
#endif

// ---- the path's labeling half: tex::view_selection (view_selection.cpp:18-133), the reference's own model construction
// (edges between faces that both have candidate views, label sets view_id + 1 / {0}, unary tables, Potts weight) and its
// decode (label_from_offset, the "Incorrect labeling" guard, set_label).  mapMAP is absent: the stand-in mapmap/full.h
// records the model and asks `solve` for one label offset per node; inside `solve` the caller reads the model with
// ref_model_sizes / ref_model_get and answers with ref_model_set_offsets.
void ref_model_sizes(std::uint64_t out[3]) {
    mapmap::Model const& m = mapmap::model();
    std::uint64_t total = 0;
    for (std::size_t i = 0; i < m.labels.size(); ++i) total += m.labels[i].size();
    out[0] = m.n_nodes; out[1] = m.edge_weights.size(); out[2] = total;
}
// edges[2E], weights[E], label_ptr[n + 1], labels[total], costs[total] (-1 entries where a node's cost vector is shorter
// than its label vector), misc[16] = {potts, window, ratio, components_updated, compress, all set_unary calls consistent,
// ctrl[0..8], seed}
void ref_model_get(std::uint32_t* edges, float* weights, std::uint32_t* label_ptr, std::int32_t* labels, float* costs, double* misc) {
    mapmap::Model const& m = mapmap::model();
    for (std::size_t e = 0; e < m.edges.size(); ++e) edges[e] = m.edges[e];
    for (std::size_t e = 0; e < m.edge_weights.size(); ++e) weights[e] = m.edge_weights[e];
    std::uint32_t n = 0;
    for (std::size_t i = 0; i < m.labels.size(); ++i) {
        label_ptr[i] = n;
        for (std::size_t k = 0; k < m.labels[i].size(); ++k) {
            labels[n] = m.labels[i][k];
            costs[n] = (i < m.costs.size() && k < m.costs[i].size() && m.costs[i].size() == m.labels[i].size()) ? m.costs[i][k] : -1.0f;
            ++n;
        }
    }
    label_ptr[m.labels.size()] = n;
    bool unaries_ok = m.unary_set.size() == m.n_nodes;
    for (std::size_t i = 0; i < m.unary_set.size(); ++i) unaries_ok = unaries_ok && m.unary_set[i] == 1;
    misc[0] = m.potts; misc[1] = m.term_window; misc[2] = m.term_ratio; misc[3] = m.components_updated; misc[4] = m.compress; misc[5] = unaries_ok;
    for (int i = 0; i < 9; ++i) misc[6 + i] = m.ctrl[i];
    misc[15] = (double)m.seed;
}
void ref_model_set_offsets(const std::int32_t* offsets) {
    mapmap::SolveHook& h = mapmap::solve_hook();
    for (std::size_t i = 0; i < h.offsets.size(); ++i) h.offsets[i] = offsets[i];
}
// returns 0, or 1 if view_selection threw (what() copied to err); labels_out[i] = graph->get_label(i) afterwards
int ref_view_selection(std::uint32_t n_faces, std::uint16_t n_views, const std::uint32_t* col_ptr, const std::uint16_t* view_id, const float* cost,
                       const std::uint32_t* adj_ptr, const std::uint32_t* adj, int (*solve)(void*), std::uint32_t* labels_out, char* err, int err_len) {
    try {
        tex::DataCosts data_costs(n_faces, n_views);
        for (std::uint32_t i = 0; i < n_faces; ++i)
            for (std::uint32_t k = col_ptr[i]; k < col_ptr[i + 1]; ++k) data_costs.set_value(i, view_id[k], cost[k]);
        UniGraph graph = make_graph(n_faces, adj_ptr, adj);
        mapmap::solve_hook().fn = solve; mapmap::solve_hook().user = nullptr;
        tex::view_selection(data_costs, &graph, tex::Settings());
        mapmap::solve_hook().fn = nullptr;
        for (std::uint32_t i = 0; i < n_faces; ++i) labels_out[i] = (std::uint32_t)graph.get_label(i);
        return 0;
    } catch (std::exception& e) {
        mapmap::solve_hook().fn = nullptr;
        if (err && err_len > 0) std::snprintf(err, (std::size_t)err_len, "%s", e.what());
        return 1;
    }
}

// ---- row f1: tex::prepare_mesh (prepare_mesh.cpp:14-70) and tex::build_adjacency_graph (build_adjacency_graph.cpp:16-53) ----
static mve::TriangleMesh::Ptr make_mesh(std::uint32_t n_verts, const float* verts, std::uint32_t n_faces, const std::uint32_t* faces) {
    mve::TriangleMesh::Ptr mesh = mve::TriangleMesh::create();
    for (std::uint32_t v = 0; v < n_verts; ++v) mesh->get_vertices().push_back(math::Vec3f(verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]));
    mesh->get_faces().assign(faces, faces + 3 * (std::size_t)n_faces);
    return mesh;
}
std::uint32_t ref_prepare_mesh(std::uint32_t n_verts, const float* verts, std::uint32_t n_faces, const std::uint32_t* faces,
                               std::uint32_t* faces_out, float* normals_out) {
    mve::TriangleMesh::Ptr mesh = make_mesh(n_verts, verts, n_faces, faces);
    mve::MeshInfo mesh_info(mesh);
    tex::prepare_mesh(&mesh_info, mesh);
    std::uint32_t const kept = (std::uint32_t)(mesh->get_faces().size() / 3);
    for (std::size_t i = 0; i < mesh->get_faces().size(); ++i) faces_out[i] = mesh->get_faces()[i];
    for (std::uint32_t f = 0; f < kept; ++f) for (int a = 0; a < 3; ++a) normals_out[3 * f + a] = mesh->get_face_normals()[f][a];
    return kept;
}
std::uint64_t ref_build_adjacency(std::uint32_t n_verts, std::uint32_t n_faces, const std::uint32_t* faces, std::uint32_t* out_ptr, std::uint32_t* out_adj,
                                  std::uint64_t cap) {
    std::vector<float> zeros(3 * (std::size_t)n_verts, 0.0f);
    mve::TriangleMesh::Ptr mesh = make_mesh(n_verts, zeros.data(), n_faces, faces);
    mve::MeshInfo mesh_info(mesh);
    UniGraph graph(n_faces);
    tex::build_adjacency_graph(mesh, mesh_info, &graph);
    std::uint64_t n = 0;
    for (std::uint32_t i = 0; i < n_faces; ++i) {
        out_ptr[i] = (std::uint32_t)n;
        std::vector<std::size_t> const& l = graph.get_adj_nodes(i);
        for (std::size_t k = 0; k < l.size(); ++k) { if (n < cap) out_adj[n] = (std::uint32_t)l[k]; ++n; }
    }
    out_ptr[n_faces] = (std::uint32_t)n;
    return graph.num_edges();
}

// ---- row f2: tex::generate_texture_views on a SCENE FOLDER (generate_texture_views.cpp:67-157 from_images_and_camera_files: the pairing of
// <prefix>.cam with the image file next to it in the sorted listing, the two-line .cam parsing, the choice of the undistortion model, the
// view ids, the name of the rewritten image) -- the reference's own code on real files of a test directory.  Everything it parsed comes
// back as one JSON text: per view (in id order) {id, image_file, trans, rot, flen, dist, paspect, ppoint}, then the undistortion calls and
// the files handed to save_png_file.  (mve::CameraInfo's fields and string setters, util::Tokenizer and the directory listing are
// stand-ins: oracle/ref_stubs.)  Returns the text's length, -1 on an exception, -2 if `cap` is too small.
#ifndef MVS_DROPIN_BUILD
int ref_scene_folder_views(const char* path, const char* tmp_dir, char* out, int cap) {
    try {
        mve::camera_log().clear(); mve::image::header_requests().clear(); mve::image::undistort_log().clear(); mve::image::saved_files().clear();
        std::vector<tex::TextureView> views;
        tex::generate_texture_views(path, &views, tmp_dir);
        std::vector<mve::CameraInfo> const& cams = mve::camera_log();
        std::vector<std::string> const& files = mve::image::header_requests();
        if (cams.size() != views.size() || files.size() != views.size()) return -1;
        std::string js = "{\"views\": [";
        char buf[256];
        // generate_texture_views sorts the views by id; the logs are in construction order = id order for a scene folder (no OpenMP here)
        for (std::size_t k = 0; k < views.size(); ++k) {
            mve::CameraInfo const& c = cams[k];
            std::snprintf(buf, sizeof(buf), "%s{\"id\": %zu, \"image_file\": \"", k ? ", " : "", views[k].get_id()); js += buf; js += files[k]; js += "\", \"trans\": [";
            for (int i = 0; i < 3; ++i) { std::snprintf(buf, sizeof(buf), "%s%.9g", i ? ", " : "", (double)c.trans[i]); js += buf; }
            js += "], \"rot\": [";
            for (int i = 0; i < 9; ++i) { std::snprintf(buf, sizeof(buf), "%s%.9g", i ? ", " : "", (double)c.rot[i]); js += buf; }
            std::snprintf(buf, sizeof(buf), "], \"flen\": %.9g, \"dist\": [%.9g, %.9g], \"paspect\": %.9g, \"ppoint\": [%.9g, %.9g]}",
                          (double)c.flen, (double)c.dist[0], (double)c.dist[1], (double)c.paspect, (double)c.ppoint[0], (double)c.ppoint[1]);
            js += buf;
        }
        js += "], \"undistort\": [";
        for (std::size_t k = 0; k < mve::image::undistort_log().size(); ++k) {
            mve::image::UndistortCall const& u = mve::image::undistort_log()[k];
            std::snprintf(buf, sizeof(buf), "%s{\"model\": \"%s\", \"flen\": %.9g, \"d0\": %.9g, \"d1\": %.9g}", k ? ", " : "", u.model == 0 ? "k2k4" : "vsfm", (double)u.flen, (double)u.d0, (double)u.d1);
            js += buf;
        }
        js += "], \"saved\": [";
        for (std::size_t k = 0; k < mve::image::saved_files().size(); ++k) { js += k ? ", \"" : "\""; js += mve::image::saved_files()[k]; js += "\""; }
        js += "]}";
        if ((int)js.size() + 1 > cap) return -2;
        std::memcpy(out, js.c_str(), js.size() + 1);
        return (int)js.size();
    } catch (std::exception& e) { std::fprintf(stderr, "ref_scene_folder_views: %s\n", e.what()); return -1; }
}
#endif

}  // extern "C"
