// tex_viewsel.hpp -- header-only C++ mirror of the reference's interface for the
// view-selection path, on top of the C ABI of mvs_viewsel.h.
//
// Same names, argument meaning and error behaviour as nmoehrle/mvs-texturing:
//   tex::Settings / DataTerm / OutlierRemoval        libs/tex/settings.h:59-95
//   tex::TextureView (the members the path reads)    libs/tex/texture_view.h:39-117
//   SparseTable<C,R,T>, tex::DataCosts               libs/tex/sparse_table.h:29-187, texturing.h:36
//   UniGraph, tex::Graph                             libs/tex/uni_graph.h:20-138, texturing.h:35
//   tex::calculate_data_costs                        libs/tex/texturing.h:66-69
//   tex::view_selection                              libs/tex/texturing.h:79-80
// so that the call sequence of apps/texrecon/texrecon.cpp:88-136 compiles against
// this header unchanged.  MVE is not required: the mesh is any object with
// get_faces() / get_vertices() / get_face_normals() (mve::TriangleMesh has them;
// tex::SimpleMesh below is a stand-in), images are plain RGB8 buffers.
// INTEGRATION.md shows how the same marshalling replaces the bodies of
// calculate_data_costs.cpp / view_selection.cpp inside an upstream checkout.
#ifndef TEX_VIEWSEL_HPP
#define TEX_VIEWSEL_HPP

#include <algorithm>
#include <array>
#include <cstdint>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "mvs_viewsel.h"

/* ---------------- SparseTable (libs/tex/sparse_table.h) ---------------- */
#define TEX_SPARSE_TABLE_HEADER "SPT"
#define TEX_SPARSE_TABLE_VERSION "0.2"

template <typename C, typename R, typename T>
class SparseTable {
public:
    typedef std::vector<std::pair<R, T> > Column;
    typedef std::vector<std::pair<C, T> > Row;

private:
    std::vector<Column> column_wise_data;
    std::vector<Row> row_wise_data;
    std::size_t nnz;

public:
    SparseTable() : nnz(0) {}
    SparseTable(C cols, R rows) : nnz(0) { column_wise_data.resize(cols); row_wise_data.resize(rows); }
    C cols() const { return static_cast<C>(column_wise_data.size()); }
    R rows() const { return static_cast<R>(row_wise_data.size()); }
    Column const& col(C id) const { return column_wise_data[id]; }
    Row const& row(R id) const { return row_wise_data[id]; }
    void set_value(C col, R row, T value) {
        column_wise_data[col].push_back(std::pair<R, T>(row, value));
        row_wise_data[row].push_back(std::pair<C, T>(col, value));
        nnz++;
    }
    std::size_t get_nnz(void) const { return nnz; }

    /* sparse_table.h:112-136 */
    static void save_to_file(SparseTable const& t, std::string const& filename) {
        std::ofstream out(filename.c_str(), std::ios::binary);
        if (!out.good()) throw std::runtime_error("Could not open " + filename);
        out << TEX_SPARSE_TABLE_HEADER << " " << TEX_SPARSE_TABLE_VERSION << " " << t.cols() << " " << t.rows() << " " << t.get_nnz() << std::endl;
        for (C col = 0; col < t.cols(); ++col)
            for (auto const& e : t.col(col)) {
                out.write((char const*)&col, sizeof(C)); out.write((char const*)&e.first, sizeof(R)); out.write((char const*)&e.second, sizeof(T));
            }
    }
    /* sparse_table.h:138-187 */
    static void load_from_file(std::string const& filename, SparseTable* t) {
        std::ifstream in(filename.c_str(), std::ios::binary);
        if (!in.good()) throw std::runtime_error("Could not open " + filename);
        std::string header, version;
        in >> header;
        if (header != TEX_SPARSE_TABLE_HEADER) throw std::runtime_error("Not a SparseTable file!");
        in >> version;
        if (version != TEX_SPARSE_TABLE_VERSION) throw std::runtime_error("Incompatible version of SparseTable file!");
        C cols; R rows; std::size_t n;
        in >> cols >> rows >> n;
        if (cols != t->cols() || rows != t->rows()) throw std::runtime_error("SparseTable has different dimension!");
        std::string rest; std::getline(in, rest);
        for (std::size_t i = 0; i < n; ++i) {
            C col; R row; T value;
            in.read((char*)&col, sizeof(C)); in.read((char*)&row, sizeof(R)); in.read((char*)&value, sizeof(T));
            t->set_value(col, row, value);
        }
    }
};

/* ---------------- UniGraph (libs/tex/uni_graph.h) ---------------- */
class UniGraph {
    std::vector<std::vector<std::size_t> > adj_lists;
    std::vector<std::size_t> labels;
    std::size_t edges;
    /* All labels' subgraphs come out of ONE GPU pass; generate_texture_patches.cpp:469-475 asks label by label (once per view), so the
     * result of the pass is kept until a label or an edge changes (set_label / add_edge are the only mutators): V calls cost one pass. */
    mutable bool cache_valid = false;
    mutable std::size_t cache_labels = 0;
    mutable mvs_subgraphs cache{};
    void drop_cache() const { if (cache_valid) { mvs_subgraphs_free(&cache); cache_valid = false; } }

public:
    explicit UniGraph(std::size_t nodes) : edges(0) { adj_lists.resize(nodes); labels.resize(nodes); }
    UniGraph(const UniGraph& o) : adj_lists(o.adj_lists), labels(o.labels), edges(o.edges) {}
    UniGraph& operator=(const UniGraph& o) { if (this != &o) { drop_cache(); adj_lists = o.adj_lists; labels = o.labels; edges = o.edges; } return *this; }
    ~UniGraph() { drop_cache(); }
    bool has_edge(std::size_t n1, std::size_t n2) const {
        auto const& l = adj_lists[n1];
        return std::find(l.begin(), l.end(), n2) != l.end();
    }
    void add_edge(std::size_t n1, std::size_t n2) {
        if (!has_edge(n1, n2)) { drop_cache(); adj_lists[n1].push_back(n2); adj_lists[n2].push_back(n1); ++edges; }
    }
    std::size_t num_edges() const { return edges; }
    std::size_t num_nodes() const { return adj_lists.size(); }
    void set_label(std::size_t n, std::size_t label) { if (labels[n] != label) drop_cache(); labels[n] = label; }
    std::size_t get_label(std::size_t n) const { return labels[n]; }
    std::vector<std::size_t> const& get_adj_nodes(std::size_t node) const { return adj_lists[node]; }

    /* uni_graph.cpp:21-55: the connected subgraphs of nodes labelled `label`, appended to *subgraphs in ascending
     * order of their smallest node, each in BFS queue order.  generate_texture_patches.cpp:469-475 asks for every
     * label in turn; the GPU computes all labels in one pass (mvs_get_subgraphs) whose result is kept (see above):
     * the first call pays the pass, the following ones slice it.  (Not thread safe, like the reference's member.) */
    void get_subgraphs(std::size_t label, std::vector<std::vector<std::size_t> >* subgraphs) const {
        if (!cache_valid || label >= cache_labels) {
            drop_cache();
            std::size_t n_labels = label + 1;
            for (std::size_t l : labels) n_labels = std::max(n_labels, l + 1);
            get_all_subgraphs(n_labels, &cache);
            cache_valid = true; cache_labels = n_labels;
        }
        const mvs_subgraphs& sg = cache;
        for (std::uint32_t c = sg.label_ptr[label]; c < sg.label_ptr[label + 1]; ++c)
            subgraphs->push_back(std::vector<std::size_t>(sg.comp_faces + sg.comp_ptr[c], sg.comp_faces + sg.comp_ptr[c + 1]));
    }
    /* all labels 0 .. n_labels - 1 in one GPU pass; free with mvs_subgraphs_free */
    void get_all_subgraphs(std::size_t n_labels, mvs_subgraphs* out) const {
        std::vector<std::uint32_t> adj_ptr(adj_lists.size() + 1, 0), adj, lab(labels.size());
        for (std::size_t i = 0; i < adj_lists.size(); ++i) {
            for (std::size_t g : adj_lists[i]) adj.push_back(static_cast<std::uint32_t>(g));
            adj_ptr[i + 1] = static_cast<std::uint32_t>(adj.size());
            lab[i] = static_cast<std::uint32_t>(labels[i]);
        }
        if (adj.empty()) adj.push_back(0);
        if (mvs_get_subgraphs(static_cast<std::uint32_t>(adj_lists.size()), adj_ptr.data(), adj.data(), lab.data(),
                              static_cast<std::uint32_t>(n_labels), out) != MVS_OK)
            throw std::runtime_error(std::string("mvs_viewsel: ") + mvs_last_error());
    }
};

namespace tex {

/* ---------------- settings (libs/tex/settings.h:59-95) ---------------- */
enum DataTerm { DATA_TERM_AREA = 0, DATA_TERM_GMI = 1 };
enum SmoothnessTerm { SMOOTHNESS_TERM_POTTS = 0 };
enum OutlierRemoval { OUTLIER_REMOVAL_NONE = 0, OUTLIER_REMOVAL_GAUSS_DAMPING = 1, OUTLIER_REMOVAL_GAUSS_CLAMPING = 2 };
enum ToneMapping { TONE_MAPPING_NONE = 0, TONE_MAPPING_GAMMA = 1 };

struct Settings {
    bool verbose = false;
    DataTerm data_term = DATA_TERM_GMI;
    SmoothnessTerm smoothness_term = SMOOTHNESS_TERM_POTTS;
    OutlierRemoval outlier_removal = OUTLIER_REMOVAL_NONE;
    ToneMapping tone_mapping = TONE_MAPPING_NONE;   /* settings.h:86 (read downstream of the path: texture patch generation) */
    bool geometric_visibility_test = true;
    bool global_seam_leveling = true;
    bool local_seam_leveling = true;
    bool hole_filling = true;
    bool keep_unseen_faces = false;
};

/* ---------------- TextureView: the members the path reads (texture_view.h:39-117) ---------------- */
class TextureView {
    std::size_t id;
    std::array<float, 3> pos, viewdir;
    std::array<float, 9> projection;      // math::Matrix3f, row major
    std::array<float, 16> world_to_cam;   // math::Matrix4f, row major
    int width, height;
    std::shared_ptr<std::vector<std::uint8_t> > image;   // RGB8, width * height * 3 (mve::ByteImage layout)

public:
    TextureView(std::size_t id, const float pos_[3], const float viewdir_[3], const float K[9], const float w2c[16], int width, int height)
        : id(id), width(width), height(height) {
        std::copy(pos_, pos_ + 3, pos.begin()); std::copy(viewdir_, viewdir_ + 3, viewdir.begin());
        std::copy(K, K + 9, projection.begin()); std::copy(w2c, w2c + 16, world_to_cam.begin());
    }
    std::size_t get_id(void) const { return id; }
    const float* get_pos(void) const { return pos.data(); }
    const float* get_viewing_direction(void) const { return viewdir.data(); }
    const float* get_projection(void) const { return projection.data(); }
    const float* get_world_to_cam(void) const { return world_to_cam.data(); }
    int get_width(void) const { return width; }
    int get_height(void) const { return height; }
    /* bind_image / get_image / release_image: texture_view.h:147-151,187-189,203-206 */
    void bind_image(std::shared_ptr<std::vector<std::uint8_t> > new_image) { image = std::move(new_image); }
    std::shared_ptr<std::vector<std::uint8_t> > get_image(void) const { return image; }
    void release_image(void) { image.reset(); }
};

typedef std::vector<TextureView> TextureViews;
typedef UniGraph Graph;
typedef SparseTable<std::uint32_t, std::uint16_t, float> DataCosts;
/** texture_view.h:26-34 */
struct FaceProjectionInfo {
    std::uint16_t view_id;
    float quality;
    float mean_color[3];   /* math::Vec3f upstream: YCbCr, filled when outlier removal is on */
    bool operator<(FaceProjectionInfo const& other) const { return view_id < other.view_id; }
};
typedef std::vector<std::vector<FaceProjectionInfo> > FaceProjectionInfos;   /* texturing.h:38 */

/** Stand-in for mve::TriangleMesh::ConstPtr with the three getters of calculate_data_costs.cpp:136-138. */
struct SimpleMesh {
    std::vector<unsigned int> faces;     // 3 per face
    std::vector<float> vertices;         // xyz per vertex
    std::vector<float> face_normals;     // xyz per face
    std::vector<unsigned int> const& get_faces() const { return faces; }
    std::vector<float> const& get_vertices() const { return vertices; }
    std::vector<float> const& get_face_normals() const { return face_normals; }
    typedef std::shared_ptr<const SimpleMesh> ConstPtr;
};

namespace detail {
/** wall-clock milliseconds of the adapter's own stages in the last calculate_data_costs / view_selection of this thread (the library's
 *  share: mvs_last_call_profile()).  table_fill_ms is the caller's container: SparseTable::set_value, two push_backs per entry
 *  (sparse_table.h:105-110) -- the reference's own fill at calculate_data_costs.cpp:291-298 pays the same. */
struct AdapterTiming { double marshal_ms = 0, library_ms = 0, table_fill_ms = 0, flatten_ms = 0, graph_ms = 0, set_labels_ms = 0; std::string library_profile; };
inline AdapterTiming& last_timing() { static thread_local AdapterTiming t; return t; }
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void throw_status(mvs_status st) {
    /* the reference throws std::runtime_error with these texts (calculate_data_costs.cpp:315-318, view_selection.cpp:126-128) */
    switch (st) {
        case MVS_OK: return;
        case MVS_ERR_TOO_MANY_FACES: throw std::runtime_error("Exeeded maximal number of faces");
        case MVS_ERR_TOO_MANY_VIEWS: throw std::runtime_error("Exeeded maximal number of views");
        case MVS_ERR_LABELING: throw std::runtime_error("Incorrect labeling");
        default: throw std::runtime_error(std::string("mvs_viewsel: ") + mvs_last_error());
    }
}
template <class V> const float* flat(V const& v) { return reinterpret_cast<const float*>(v.data()); }  // vector<float> or vector<math::Vec3f>
/** fn(slice, n_slices) for every slice, on up to `want` threads; slices that get no thread (built without -pthread, thread limit) run here */
template <class Fn> void for_slices(unsigned want, Fn fn) {
    unsigned const n = std::max(1u, std::min(want, std::max(1u, std::thread::hardware_concurrency())));
    std::vector<std::thread> th;
    unsigned started = 1;   // slice 0 is the caller's
    try { for (; started < n; ++started) th.emplace_back(fn, started, n); } catch (std::system_error const&) {}
    fn(0u, n);
    for (unsigned t = started; t < n; ++t) fn(t, n);
    for (auto& x : th) x.join();
}
/** the caller's chunk of the streamed table into its SparseTable (calculate_data_costs.cpp:291-298) */
inline void fill_chunk(void* user, std::uint32_t first_face, std::uint32_t n_faces, const std::uint32_t* col_ptr, const std::uint16_t* view_id, const float* cost) {
    DataCosts* dc = static_cast<DataCosts*>(user);
    std::uint32_t const base = col_ptr[0];
    for (std::uint32_t i = 0; i < n_faces; ++i)
        for (std::uint32_t k = col_ptr[i]; k < col_ptr[i + 1]; ++k) dc->set_value(first_face + i, view_id[k - base], cost[k - base]);
}
}  // namespace detail

/**
 * Calculates the data costs for each face and texture view combination,
 * if the face is visible within the texture view.   (libs/tex/texturing.h:62-69)
 * `data_costs` must be pre-sized by the caller to (faces, views) as at texrecon.cpp:98.
 * Views must have their image bound (the reference loads it from disk at calculate_data_costs.cpp:157 and
 * releases it at :231; this adapter likewise leaves the views released).
 */
template <class MeshConstPtr>
void calculate_data_costs(MeshConstPtr mesh, TextureViews* texture_views, Settings const& settings, DataCosts* data_costs) {
    double const t0 = detail::now_ms();
    std::size_t const num_faces = mesh->get_faces().size() / 3;
    std::size_t const num_views = texture_views->size();
    if (num_faces > std::numeric_limits<std::uint32_t>::max()) throw std::runtime_error("Exeeded maximal number of faces");
    if (num_views > std::numeric_limits<std::uint16_t>::max()) throw std::runtime_error("Exeeded maximal number of views");

    mvs_mesh m;
    m.n_faces = static_cast<std::uint32_t>(num_faces);
    m.n_verts = static_cast<std::uint32_t>(mesh->get_vertices().size() * sizeof(mesh->get_vertices()[0]) / (3 * sizeof(float)));
    m.verts = detail::flat(mesh->get_vertices());
    m.faces = mesh->get_faces().data();
    m.face_normals = detail::flat(mesh->get_face_normals());
    /* the images of this mirror type are bound by the caller before the call and released here on EVERY exit path (:231); the replacement
     * translation unit for the real tex::TextureView loads them view by view instead (integration/view_selection_mi355x.cpp, mvs_image_source) */
    struct ReleaseAll { std::vector<TextureView>* v; ~ReleaseAll() { for (TextureView& tv : *v) tv.release_image(); } } release_all{texture_views};
    std::vector<mvs_view> views(num_views);
    for (std::size_t j = 0; j < num_views; ++j) {
        TextureView const& tv = texture_views->at(j);
        if (!tv.get_image()) throw std::runtime_error("TextureView without image");
        std::memcpy(views[j].pos, tv.get_pos(), sizeof(float) * 3);
        std::memcpy(views[j].viewdir, tv.get_viewing_direction(), sizeof(float) * 3);
        std::memcpy(views[j].K, tv.get_projection(), sizeof(float) * 9);
        std::memcpy(views[j].w2c, tv.get_world_to_cam(), sizeof(float) * 16);
        views[j].width = tv.get_width(); views[j].height = tv.get_height();
        views[j].rgb = tv.get_image()->data();
    }
    mvs_settings st;
    st.data_term = settings.data_term; st.outlier_removal = settings.outlier_removal;
    st.geometric_visibility_test = settings.geometric_visibility_test ? 1 : 0;
    mvs_dc_stats stats;
    detail::AdapterTiming& T = detail::last_timing(); T = detail::AdapterTiming();
    double const t1 = detail::now_ms();
    T.marshal_ms = t1 - t0;
    /* the table arrives in chunks of faces while the next chunk is still on the bus: the fill below (the caller's container, :291-298)
     * hides the download; the table also stays on the device for the view_selection that follows (texrecon.cpp:100,121) */
    detail::throw_status(mvs_data_costs_stream(&m, views.data(), static_cast<std::uint32_t>(num_views), &st, &detail::fill_chunk, data_costs, nullptr, &stats));
    double const t2 = detail::now_ms();
    T.library_profile = mvs_last_call_profile();
    /* library_ms = up to the first chunk; table_fill_ms = the chunk loop (set_value for every entry + whatever of the download it did not hide) */
    {
        double chunks = 0.0; std::size_t const at = T.library_profile.find("\"chunks_and_callbacks_ms\": ");
        if (at != std::string::npos) chunks = std::atof(T.library_profile.c_str() + at + 27);
        T.table_fill_ms = chunks; T.library_ms = (t2 - t1) - chunks;
    }
    std::cout << "\tMaximum quality of a face within an image: " << stats.max_quality << std::endl;     /* :304-305 */
    std::cout << "\tClamping qualities to " << stats.percentile << " within normalization." << std::endl;
}

/** Postprocessing of the per face projection infos into data costs   (libs/tex/texturing.h:71-74,
 * calculate_data_costs.cpp:253-306).  As upstream: the infos are consumed (the vector is cleared, :301) and data_costs
 * must be pre-sized to (faces, views). */
inline void postprocess_face_infos(Settings const& settings, FaceProjectionInfos* face_projection_infos, DataCosts* data_costs) {
    std::size_t const F = face_projection_infos->size();
    std::vector<std::uint32_t> ptr(F + 1, 0);
    std::vector<std::uint16_t> view_id; std::vector<float> quality, color;
    for (std::size_t i = 0; i < F; ++i) {
        for (FaceProjectionInfo const& info : face_projection_infos->at(i)) {
            view_id.push_back(info.view_id); quality.push_back(info.quality);
            color.push_back(info.mean_color[0]); color.push_back(info.mean_color[1]); color.push_back(info.mean_color[2]);
        }
        ptr[i + 1] = static_cast<std::uint32_t>(view_id.size());
    }
    mvs_settings st;
    st.data_term = settings.data_term; st.outlier_removal = settings.outlier_removal;
    st.geometric_visibility_test = settings.geometric_visibility_test ? 1 : 0;
    mvs_csr csr; std::memset(&csr, 0, sizeof(csr));
    mvs_dc_stats stats;
    detail::throw_status(mvs_postprocess_face_infos(static_cast<std::uint32_t>(F), data_costs->rows(), ptr.data(), view_id.data(), quality.data(),
                                                    color.data(), &st, &csr, &stats));
    for (std::uint32_t i = 0; i < csr.n_faces; ++i)          /* calculate_data_costs.cpp:291-298 */
        for (std::uint32_t k = csr.col_ptr[i]; k < csr.col_ptr[i + 1]; ++k) data_costs->set_value(i, csr.view_id[k], csr.cost[k]);
    mvs_csr_free(&csr);
    face_projection_infos->clear();                           /* :301 */
    std::cout << "\tMaximum quality of a face within an image: " << stats.max_quality << std::endl;     /* :304-305 */
    std::cout << "\tClamping qualities to " << stats.percentile << " within normalization." << std::endl;
}

/** Runs the view selection procedure and saves the labeling in the graph   (libs/tex/texturing.h:76-80) */
inline void view_selection(DataCosts const& data_costs, UniGraph* graph, Settings const&) {
    detail::AdapterTiming& T = detail::last_timing(); T = detail::AdapterTiming();
    double const t0 = detail::now_ms();
    std::uint32_t const F = data_costs.cols();
    /* Is this the table calculate_data_costs just handed out (texrecon.cpp:100,121)?  Then it is still on the device.  Its fingerprint
     * (mvs_viewsel.h) is summed over the caller's container column by column -- the container is READ once, nothing is copied. */
    std::vector<std::uint32_t> col_ptr(F + 1, 0);
    for (std::uint32_t i = 0; i < F; ++i) col_ptr[i + 1] = col_ptr[i] + static_cast<std::uint32_t>(data_costs.col(i).size());
    std::uint64_t const nnz = col_ptr[F];
    unsigned const want = static_cast<unsigned>(std::min<std::uint64_t>(32u, nnz / (1u << 20) + 1u));
    std::vector<std::uint64_t> part(std::max(1u, std::min(want, std::max(1u, std::thread::hardware_concurrency()))), 0);
    detail::for_slices(static_cast<unsigned>(part.size()), [&](unsigned t, unsigned of) {
        std::uint64_t h = 0;
        for (std::uint32_t i = static_cast<std::uint32_t>(static_cast<std::uint64_t>(F) * t / of); i < static_cast<std::uint32_t>(static_cast<std::uint64_t>(F) * (t + 1) / of); ++i) {
            h += mvs_fp_mix(i, col_ptr[i + 1]);
            std::uint64_t k = col_ptr[i];
            for (auto const& e : data_costs.col(i)) {
                std::uint32_t bits; std::memcpy(&bits, &e.second, 4);
                h += mvs_fp_mix((1ull << 40) + k, (static_cast<std::uint64_t>(e.first) << 32) | bits); ++k;
            }
        }
        part[t] = h;
    });
    std::uint64_t fp = mvs_fp_mix(F, data_costs.rows()) + mvs_fp_mix(nnz, 1);
    for (std::uint64_t v : part) fp += v;
    double const t1 = detail::now_ms();
    T.flatten_ms = t1 - t0;   /* the time spent READING the caller's container (fingerprint walk; plus the flatten below if the table is not the parked one) */
    /* UniGraph adjacency lists flattened in list order */
    std::vector<std::uint32_t> adj_ptr(F + 1, 0);
    for (std::uint32_t i = 0; i < F; ++i) adj_ptr[i + 1] = adj_ptr[i] + static_cast<std::uint32_t>(graph->get_adj_nodes(i).size());
    std::vector<std::uint32_t> adj(static_cast<std::size_t>(adj_ptr[F]) + 1, 0);
    detail::for_slices(static_cast<unsigned>(std::min<std::uint32_t>(16u, F / 65536u + 1u)), [&](unsigned t, unsigned of) {
        for (std::uint32_t i = static_cast<std::uint32_t>(static_cast<std::uint64_t>(F) * t / of); i < static_cast<std::uint32_t>(static_cast<std::uint64_t>(F) * (t + 1) / of); ++i) {
            std::size_t k = adj_ptr[i];
            for (std::size_t n : graph->get_adj_nodes(i)) adj[k++] = static_cast<std::uint32_t>(n);
        }
    });
    double const t2 = detail::now_ms();
    T.graph_ms = t2 - t1;
    std::vector<std::uint32_t> labels(F, 0);
    mvs_mrf_stats stats;
    std::cout << "\tOptimizing:" << std::endl;
    mvs_status st = mvs_view_selection_cached(fp, F, data_costs.rows(), nnz, adj_ptr.data(), adj.data(), nullptr, labels.data(), &stats);
    double t3 = detail::now_ms();
    if (st == MVS_ERR_STATE) {
        /* not the parked table (changed by the caller, loaded from a file, computed elsewhere): flatten it and hand it over */
        std::vector<std::uint16_t> view_id(nnz + 1); std::vector<float> cost(nnz + 1);
        detail::for_slices(static_cast<unsigned>(std::min<std::uint32_t>(8u, F / 65536u + 1u)), [&](unsigned t, unsigned of) {
            for (std::uint32_t i = static_cast<std::uint32_t>(static_cast<std::uint64_t>(F) * t / of); i < static_cast<std::uint32_t>(static_cast<std::uint64_t>(F) * (t + 1) / of); ++i) {
                std::size_t k = col_ptr[i];
                for (auto const& e : data_costs.col(i)) { view_id[k] = e.first; cost[k] = e.second; ++k; }
            }
        });
        double const tf = detail::now_ms();
        T.flatten_ms += tf - t3;
        mvs_csr csr;
        csr.n_faces = F; csr.n_views = data_costs.rows(); csr.nnz = nnz;
        csr.col_ptr = col_ptr.data(); csr.view_id = view_id.data(); csr.cost = cost.data();
        st = mvs_view_selection(&csr, adj_ptr.data(), adj.data(), nullptr, labels.data(), &stats);
        t3 = detail::now_ms();
        T.library_ms = t3 - tf;
    } else T.library_ms = t3 - t2;
    detail::throw_status(st);
    T.library_profile = mvs_last_call_profile();
    std::cout << "\t\t" << stats.sweeps << " sweeps\t" << stats.energy << std::endl;
    for (std::uint32_t i = 0; i < F; ++i) graph->set_label(i, labels[i]);                  /* view_selection.cpp:130 */
    T.set_labels_ms = detail::now_ms() - t3;
    std::cout << '\t' << stats.unseen << " faces have not been seen" << std::endl;      /* :132 */
}

}  // namespace tex

#endif  // TEX_VIEWSEL_HPP
