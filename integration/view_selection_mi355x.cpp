// view_selection_mi355x.cpp -- the link-time drop-in for upstream mvs-texturing: ONE translation unit that replaces
// libs/tex/calculate_data_costs.cpp and libs/tex/view_selection.cpp in the `tex` target.  It is compiled against the REFERENCE's
// own headers (libs/tex/texturing.h, texture_view.h, sparse_table.h, uni_graph.h, settings.h -- all untouched) and defines the three
// symbols texturing.h:66-80 declares, with upstream's signatures, exception texts and console lines; the bodies marshal to the
// C ABI of include/mvs_viewsel.h and link libmvs_viewsel.so.
//
// In this repository the file is BUILT and TESTED: oracle/Makefile target `dropin` compiles it -- together with upstream's own
// texture_view.cpp, tri.cpp, histogram.cpp, uni_graph.cpp, where they lie -- against /root/reference/libs/tex and the stand-in
// MVE headers of oracle/ref_stubs into oracle/_ref/libtexdrop.so; tests/test_integration_tu.py (-m gpu) sends the same
// mve::TriangleMesh / std::vector<tex::TextureView> / tex::DataCosts / UniGraph objects once through upstream's
// tex::calculate_data_costs (oracle/_ref/libtexref.so) and once through this file and compares the containers.
//
// TextureView keeps `projection` and `world_to_cam` private (texture_view.h:43-48) and offers no accessor.  A maintainer adds two
// one-line getters and defines MVS_TEXTUREVIEW_HAS_ACCESSORS (INTEGRATION.md shows the diff); WITHOUT touching upstream the two
// members are reached below through explicit template instantiation, which the standard exempts from access checking
// ([temp.spec]/6) -- that is the configuration the test builds, so the reference's headers stay byte for byte what they are.
#include "texturing.h"

#include <cstdint>
#include <cstring>
#include <exception>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mvs_viewsel.h"            // this repository: include/

namespace {

#ifndef MVS_TEXTUREVIEW_HAS_ACCESSORS
template <class Tag, typename Tag::type Member> struct Expose { friend typename Tag::type reach(Tag) { return Member; } };
struct ProjectionTag { typedef math::Matrix3f tex::TextureView::* type; friend type reach(ProjectionTag); };
struct WorldToCamTag { typedef math::Matrix4f tex::TextureView::* type; friend type reach(WorldToCamTag); };
template struct Expose<ProjectionTag, &tex::TextureView::projection>;
template struct Expose<WorldToCamTag, &tex::TextureView::world_to_cam>;
math::Matrix3f const & projection_of(tex::TextureView const & tv) { return tv.*reach(ProjectionTag()); }
math::Matrix4f const & world_to_cam_of(tex::TextureView const & tv) { return tv.*reach(WorldToCamTag()); }
#else
math::Matrix3f const & projection_of(tex::TextureView const & tv) { return tv.get_projection(); }
math::Matrix4f const & world_to_cam_of(tex::TextureView const & tv) { return tv.get_world_to_cam(); }
#endif

void check(mvs_status st) {     // the exceptions of calculate_data_costs.cpp:315-318 and view_selection.cpp:126-128
    if (st == MVS_OK) return;
    if (st == MVS_ERR_TOO_MANY_FACES) throw std::runtime_error("Exeeded maximal number of faces");
    if (st == MVS_ERR_TOO_MANY_VIEWS) throw std::runtime_error("Exeeded maximal number of views");
    if (st == MVS_ERR_LABELING) throw std::runtime_error("Incorrect labeling");
    throw std::runtime_error(std::string("mvs_viewsel: ") + mvs_last_error());
}

mvs_settings c_settings(tex::Settings const & settings) {
    mvs_settings st;
    st.data_term = static_cast<int32_t>(settings.data_term);                 // DATA_TERM_AREA = 0, DATA_TERM_GMI = 1 (settings.h:59-62)
    st.outlier_removal = static_cast<int32_t>(settings.outlier_removal);     // NONE / GAUSS_DAMPING / GAUSS_CLAMPING = 0 / 1 / 2 (:69-73)
    st.geometric_visibility_test = settings.geometric_visibility_test ? 1 : 0;
    return st;
}

}  // namespace

TEX_NAMESPACE_BEGIN

void
calculate_data_costs(mve::TriangleMesh::ConstPtr mesh, std::vector<TextureView> * texture_views,
    Settings const & settings, DataCosts * data_costs) {

    std::size_t const num_faces = mesh->get_faces().size() / 3;
    std::size_t const num_views = texture_views->size();

    if (num_faces > std::numeric_limits<std::uint32_t>::max())               // calculate_data_costs.cpp:315-318, before any work
        throw std::runtime_error("Exeeded maximal number of faces");
    if (num_views > std::numeric_limits<std::uint16_t>::max())
        throw std::runtime_error("Exeeded maximal number of views");

    mvs_mesh m;                                                               // :136-138
    std::memset(&m, 0, sizeof(m));
    m.n_faces = static_cast<std::uint32_t>(num_faces);
    m.n_verts = static_cast<std::uint32_t>(mesh->get_vertices().size());
    static float const no_floats[3] = {0.0f, 0.0f, 0.0f};
    static std::uint32_t const no_index[3] = {0, 0, 0};
    m.verts = m.n_verts ? *mesh->get_vertices()[0] : no_floats;               // math::Vec3f = three packed floats
    m.faces = num_faces ? mesh->get_faces().data() : no_index;
    m.face_normals = num_faces ? *mesh->get_face_normals()[0] : no_floats;

    std::vector<mvs_view> views(num_views);
    for (std::size_t j = 0; j < num_views; ++j) {
        TextureView & tv = texture_views->at(j);
        math::Vec3f const pos = tv.get_pos(), dir = tv.get_viewing_direction();
        std::memcpy(views[j].pos, *pos, sizeof(views[j].pos));
        std::memcpy(views[j].viewdir, *dir, sizeof(views[j].viewdir));
        std::memcpy(views[j].K, *projection_of(tv), sizeof(views[j].K));      // row major, like math::Matrix
        std::memcpy(views[j].w2c, *world_to_cam_of(tv), sizeof(views[j].w2c));
        views[j].width = tv.get_width(); views[j].height = tv.get_height();   // (known from the image header: texture_view.cpp:21-40)
        views[j].rgb = nullptr;                                               // the pixels come through the image source below
    }
    // Upstream holds ONE decoded image at a time (:157 tv.load_image() ... :231 tv.release_image()).  The library asks for the pixels view by
    // view, on this thread, a few views at a time, and hands every one back as soon as it is on the device -- also when something fails
    // half way (a missing file throws inside load_image: the exception is kept, the views loaded so far are released, then it is rethrown).
    struct Images {
        std::vector<TextureView> * views; std::exception_ptr error;
        static std::uint8_t const * acquire(void * user, std::uint32_t j) {
            Images * self = static_cast<Images *>(user);
            try {
                TextureView & tv = self->views->at(j);
                tv.load_image();                                              // :157 (decoding stays on the host)
                return tv.get_image()->get_data_pointer();                    // mve::ByteImage, three interleaved channels
            } catch (...) { self->error = std::current_exception(); return nullptr; }
        }
        static void release(void * user, std::uint32_t j) { static_cast<Images *>(user)->views->at(j).release_image(); }   // :231
    } images{texture_views, nullptr};
    mvs_image_source source;
    source.acquire = &Images::acquire; source.release = &Images::release; source.user = &images; source.max_in_flight = 4;
    mvs_settings st = c_settings(settings);
    mvs_dc_stats stats;
    std::memset(&stats, 0, sizeof(stats));
    mvs_view const no_view = mvs_view();
    // The table arrives in chunks of 65 536 faces while the next chunk is still on the bus: this fill (:291-298) hides the download.
    // The table also stays on the device, fingerprinted there, for the tex::view_selection that follows (texrecon.cpp:100,121).
    mvs_status const rc = mvs_data_costs_stream_from(&m, num_views ? views.data() : &no_view, static_cast<std::uint32_t>(num_views), &source, &st,
        [](void * user, std::uint32_t first, std::uint32_t n, std::uint32_t const * ptr, std::uint16_t const * view, float const * cost) {
            DataCosts * dc = static_cast<DataCosts *>(user);
            for (std::uint32_t i = 0; i < n; ++i)
                for (std::uint32_t k = ptr[i]; k < ptr[i + 1]; ++k)
                    dc->set_value(first + i, view[k - ptr[0]], cost[k - ptr[0]]);
        }, data_costs, nullptr, &stats);
    if (images.error) std::rethrow_exception(images.error);                   // upstream's own exception (e.g. util::Exception of a missing image)
    check(rc);

    std::cout << "\tMaximum quality of a face within an image: " << stats.max_quality << std::endl;       // :304-305
    std::cout << "\tClamping qualities to " << stats.percentile << " within normalization." << std::endl;
}

void
postprocess_face_infos(Settings const & settings, FaceProjectionInfos * face_projection_infos,
    DataCosts * data_costs) {                                                 // texturing.h:71-74; calculate_data_costs.cpp:253-306

    std::vector<std::uint32_t> ptr(1, 0); std::vector<std::uint16_t> view; std::vector<float> quality, color;
    for (std::vector<FaceProjectionInfo> const & infos : *face_projection_infos) {        // the caller's order is kept
        for (FaceProjectionInfo const & info : infos) {
            view.push_back(info.view_id); quality.push_back(info.quality);
            color.insert(color.end(), *info.mean_color, *info.mean_color + 3);
        }
        ptr.push_back(static_cast<std::uint32_t>(view.size()));
    }
    mvs_settings st = c_settings(settings);
    mvs_csr csr; mvs_dc_stats stats;
    std::memset(&csr, 0, sizeof(csr)); std::memset(&stats, 0, sizeof(stats));
    check(mvs_postprocess_face_infos(static_cast<std::uint32_t>(face_projection_infos->size()), static_cast<std::uint32_t>(data_costs->rows()),
        ptr.data(), view.data(), quality.data(), color.data(), &st, &csr, &stats));
    for (std::uint32_t i = 0; i < csr.n_faces; ++i)                           // :291-298
        for (std::uint32_t k = csr.col_ptr[i]; k < csr.col_ptr[i + 1]; ++k)
            data_costs->set_value(i, csr.view_id[k], csr.cost[k]);
    mvs_csr_free(&csr);
    for (std::vector<FaceProjectionInfo> & infos : *face_projection_infos) infos = std::vector<FaceProjectionInfo>();   // :300-301

    std::cout << "\tMaximum quality of a face within an image: " << stats.max_quality << std::endl;
    std::cout << "\tClamping qualities to " << stats.percentile << " within normalization." << std::endl;
}

void
view_selection(DataCosts const & data_costs, UniGraph * graph, Settings const &) {
    std::uint32_t const F = static_cast<std::uint32_t>(data_costs.cols());
    std::vector<std::uint32_t> col_ptr(F + 1, 0), adj_ptr(F + 1, 0), adj, labels(F, 0);
    for (std::uint32_t i = 0; i < F; ++i) {
        col_ptr[i + 1] = col_ptr[i] + static_cast<std::uint32_t>(data_costs.col(i).size());
        for (std::size_t n : graph->get_adj_nodes(i)) adj.push_back(static_cast<std::uint32_t>(n));   // list order matters: message sums follow it
        adj_ptr[i + 1] = static_cast<std::uint32_t>(adj.size());
    }
    if (adj.empty()) adj.push_back(0);                                         // keep .data() non-null
    std::uint64_t const nnz = col_ptr[F];
    // Is this the table calculate_data_costs left on the device?  One read of the container, nothing copied.
    std::uint64_t fp = mvs_fp_mix(F, data_costs.rows()) + mvs_fp_mix(nnz, 1);
    for (std::uint32_t i = 0; i < F; ++i) {
        fp += mvs_fp_mix(i, col_ptr[i + 1]);
        std::uint64_t k = col_ptr[i];
        for (std::pair<std::uint16_t, float> const & e : data_costs.col(i)) { // view_selection.cpp:46-47,65-66
            std::uint32_t bits; std::memcpy(&bits, &e.second, 4);
            fp += mvs_fp_mix((1ull << 40) + k++, std::uint64_t(e.first) << 32 | bits);
        }
    }
    mvs_mrf_stats stats;
    std::memset(&stats, 0, sizeof(stats));
    mvs_status rc = mvs_view_selection_cached(fp, F, static_cast<std::uint32_t>(data_costs.rows()), nnz, adj_ptr.data(), adj.data(), nullptr,
        labels.data(), &stats);
    if (rc == MVS_ERR_STATE) {                 // a table that was modified, loaded with -D or computed elsewhere: flatten and upload
        std::vector<std::uint16_t> view; std::vector<float> cost;
        view.reserve(nnz + 1); cost.reserve(nnz + 1);
        for (std::uint32_t i = 0; i < F; ++i)
            for (std::pair<std::uint16_t, float> const & e : data_costs.col(i)) { view.push_back(e.first); cost.push_back(e.second); }
        if (view.empty()) { view.push_back(0); cost.push_back(0.0f); }
        mvs_csr csr;
        csr.n_faces = F; csr.n_views = static_cast<std::uint32_t>(data_costs.rows()); csr.nnz = nnz;
        csr.col_ptr = col_ptr.data(); csr.view_id = view.data(); csr.cost = cost.data();
        rc = mvs_view_selection(&csr, adj_ptr.data(), adj.data(), nullptr, labels.data(), &stats);
    }
    check(rc);                                                                 // throws "Incorrect labeling" (:126-128)
    for (std::uint32_t i = 0; i < F; ++i) graph->set_label(i, labels[i]);     // :130
    std::cout << '\t' << stats.unseen << " faces have not been seen" << std::endl;       // :132
}

TEX_NAMESPACE_END
