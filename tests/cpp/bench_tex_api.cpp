// Timing of the REAL drop-in path: tex::calculate_data_costs + tex::view_selection through include/tex_viewsel.hpp -- host
// containers in (mesh, TextureViews with bound images), DataCosts / UniGraph out -- i.e. the window apps/texrecon/texrecon.cpp:96-127
// as texrecon would run it against this library.  Prints one JSON object.  Usage: bench_tex_api <n> <views> <width> <height> <reps> <out.json>  (the adapter itself prints the reference's progress lines to stdout)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "tex_viewsel.hpp"

extern "C" {
typedef struct { uint32_t n_verts, n_faces; float* verts; uint32_t* faces; float* normals; uint32_t* adj_ptr; uint32_t* adj; } synth_mesh;
typedef struct { float pos[3], viewdir[3], K[9], w2c[16]; int32_t width, height; } synth_camera;
int synth_icosphere(uint32_t n, float amp, uint32_t seed, synth_mesh* out);
int synth_build_adjacency(synth_mesh* m);
void synth_mesh_free(synth_mesh* m);
int synth_cameras(uint32_t n_views, int layout, float radius, int w, int h, float zoom_odd, synth_camera* out);
void synth_render(const synth_camera* cam, uint32_t view_index, uint32_t seed, int black_corner, uint8_t* rgb);
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "usage: %s n views width height reps out.json\n", argv[0]); return 2; }
    const uint32_t n = std::atoi(argv[1]), V = std::atoi(argv[2]); const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
    const int reps = std::atoi(argv[5]);
    std::FILE* out = std::fopen(argv[6], "w"); if (!out) return 5;
    synth_mesh sm; if (synth_icosphere(n, 0.05f, 1234, &sm) || synth_build_adjacency(&sm)) return 3;
    auto mesh = std::make_shared<tex::SimpleMesh>();
    mesh->faces.assign(sm.faces, sm.faces + 3 * (size_t)sm.n_faces);
    mesh->vertices.assign(sm.verts, sm.verts + 3 * (size_t)sm.n_verts);
    mesh->face_normals.assign(sm.normals, sm.normals + 3 * (size_t)sm.n_faces);
    std::vector<synth_camera> cams(V); if (synth_cameras(V, 1, 3.0f, W, H, 1.0f, cams.data())) return 4;
    std::vector<std::shared_ptr<std::vector<std::uint8_t> > > images(V);
#pragma omp parallel for schedule(dynamic)
    for (int j = 0; j < (int)V; ++j) {
        images[j] = std::make_shared<std::vector<std::uint8_t> >((size_t)W * H * 3);
        synth_render(&cams[j], (uint32_t)j, 99, 0, images[j]->data());
    }
    std::size_t const num_faces = mesh->get_faces().size() / 3;
    tex::Settings settings;
    std::fprintf(out, "{\"faces\": %zu, \"views\": %u, \"width\": %d, \"height\": %d, \"runs\": [", num_faces, V, W, H);
    for (int rep = 0; rep < reps; ++rep) {
        tex::TextureViews texture_views;
        for (uint32_t j = 0; j < V; ++j) {
            texture_views.emplace_back(j, cams[j].pos, cams[j].viewdir, cams[j].K, cams[j].w2c, W, H);
            texture_views.back().bind_image(images[j]);
        }
        tex::Graph graph(num_faces);
        for (uint32_t i = 0; i < sm.n_faces; ++i)
            for (uint32_t e = sm.adj_ptr[i]; e < sm.adj_ptr[i + 1]; ++e) graph.add_edge(i, sm.adj[e]);
        double const t0 = now_ms();
        tex::DataCosts data_costs(static_cast<std::uint32_t>(num_faces), static_cast<std::uint16_t>(texture_views.size()));   /* texrecon.cpp:98 */
        tex::calculate_data_costs(tex::SimpleMesh::ConstPtr(mesh), &texture_views, settings, &data_costs);                   /* :100 */
        double const t1 = now_ms();
        tex::detail::AdapterTiming const dc = tex::detail::last_timing();
        tex::view_selection(data_costs, &graph, settings);                                                                    /* :121 */
        double const t2 = now_ms();
        tex::detail::AdapterTiming const vs = tex::detail::last_timing();
        std::size_t labelled = 0; for (std::size_t i = 0; i < graph.num_nodes(); ++i) labelled += graph.get_label(i) != 0;
        std::fprintf(out, "%s{\"calculate_data_costs_ms\": %.3f, \"view_selection_ms\": %.3f, \"dropin_ms\": %.3f, \"nnz\": %zu, \"labelled\": %zu,\n"
                    "  \"calculate_data_costs\": {\"marshal_ms\": %.3f, \"library_ms\": %.3f, \"table_fill_ms\": %.3f, \"library\": %s},\n"
                    "  \"view_selection\": {\"flatten_ms\": %.3f, \"graph_ms\": %.3f, \"library_ms\": %.3f, \"set_labels_ms\": %.3f, \"library\": %s}}",
                    rep ? ",\n " : "", t1 - t0, t2 - t1, t2 - t0, (std::size_t)data_costs.get_nnz(), labelled,
                    dc.marshal_ms, dc.library_ms, dc.table_fill_ms, dc.library_profile.c_str(),
                    vs.flatten_ms, vs.graph_ms, vs.library_ms, vs.set_labels_ms, vs.library_profile.c_str());
    }
    std::fprintf(out, "]}\n"); std::fclose(out);
    synth_mesh_free(&sm);
    return 0;
}
