// The call sequence of apps/texrecon/texrecon.cpp:88-136 against include/tex_viewsel.hpp:
//   tex::Graph graph(num_faces); [build adjacency]; tex::DataCosts data_costs(num_faces, views);
//   tex::calculate_data_costs(mesh, &texture_views, settings, &data_costs);  [save .spt]
//   tex::view_selection(data_costs, &graph, settings);                        [save labeling .vec]
// on a synthetic scene from csrc/scene_synth.cpp.  Usage: test_tex_api <out_prefix> [n] [views]
#include <cstdio>
#include <cstdlib>
#include <fstream>
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

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out_prefix [n] [views]\n", argv[0]); return 2; }
    const std::string prefix = argv[1];
    const uint32_t n = argc > 2 ? std::atoi(argv[2]) : 8, V = argc > 3 ? std::atoi(argv[3]) : 8;
    const int W = 320, H = 240;
    synth_mesh sm; if (synth_icosphere(n, 0.2f, 1234, &sm) || synth_build_adjacency(&sm)) return 3;
    auto mesh = std::make_shared<tex::SimpleMesh>();
    mesh->faces.assign(sm.faces, sm.faces + 3 * (size_t)sm.n_faces);
    mesh->vertices.assign(sm.verts, sm.verts + 3 * (size_t)sm.n_verts);
    mesh->face_normals.assign(sm.normals, sm.normals + 3 * (size_t)sm.n_faces);
    std::vector<synth_camera> cams(V); if (synth_cameras(V, 1, 3.0f, W, H, 1.4f, cams.data())) return 4;
    tex::TextureViews texture_views;
    for (uint32_t j = 0; j < V; ++j) {
        auto img = std::make_shared<std::vector<std::uint8_t> >((size_t)W * H * 3);
        synth_render(&cams[j], j, 99, j == 0 ? 20 : 0, img->data());
        texture_views.emplace_back(j, cams[j].pos, cams[j].viewdir, cams[j].K, cams[j].w2c, W, H);
        texture_views.back().bind_image(img);
    }
    std::size_t const num_faces = mesh->get_faces().size() / 3;
    tex::Graph graph(num_faces);                                                  /* texrecon.cpp:91-92 */
    for (uint32_t i = 0; i < sm.n_faces; ++i)
        for (uint32_t e = sm.adj_ptr[i]; e < sm.adj_ptr[i + 1]; ++e) graph.add_edge(i, sm.adj[e]);
    tex::Settings settings;                                                       /* defaults: gmi / none / visibility test */
    tex::DataCosts data_costs(static_cast<std::uint32_t>(num_faces), static_cast<std::uint16_t>(texture_views.size()));  /* :98 */
    try {
        tex::calculate_data_costs(tex::SimpleMesh::ConstPtr(mesh), &texture_views, settings, &data_costs);             /* :100 */
        tex::DataCosts::save_to_file(data_costs, prefix + "_data_costs.spt");                                          /* :104 */
        tex::view_selection(data_costs, &graph, settings);                                                              /* :121 */
    } catch (std::runtime_error& e) {
        std::fprintf(stderr, "\tOptimization failed: %s\n", e.what());                                                  /* :123 */
        return 1;
    }
    /* the table calculate_data_costs handed out was still on the device: view_selection found it by its fingerprint */
    bool const cached = tex::detail::last_timing().library_profile.find("mvs_view_selection_cached") != std::string::npos;
    std::printf("view_selection on the parked table: %s\n", cached ? "yes" : "no");
    std::vector<std::size_t> labeling(graph.num_nodes());                         /* :130-136 */
    for (std::size_t i = 0; i < graph.num_nodes(); ++i) labeling[i] = graph.get_label(i);
    std::ofstream out((prefix + "_labeling.vec").c_str(), std::ios::binary);
    out.write(reinterpret_cast<const char*>(labeling.data()), labeling.size() * sizeof(std::size_t));
    /* the guards of calculate_data_costs.cpp:317-318 surface as the reference's exception text */
    tex::DataCosts reload(static_cast<std::uint32_t>(num_faces), static_cast<std::uint16_t>(texture_views.size()));
    tex::DataCosts::load_from_file(prefix + "_data_costs.spt", &reload);          /* texrecon.cpp:110 */
    if (reload.get_nnz() != data_costs.get_nnz()) return 5;
    {   /* the same table again (nothing is parked any more) and a table loaded from the file: both take the flatten-and-upload route, same labels */
        tex::Graph g2(num_faces);
        for (uint32_t i = 0; i < sm.n_faces; ++i)
            for (uint32_t e = sm.adj_ptr[i]; e < sm.adj_ptr[i + 1]; ++e) g2.add_edge(i, sm.adj[e]);
        tex::view_selection(reload, &g2, settings);
        bool const uploaded = tex::detail::last_timing().library_profile.find("\"mvs_view_selection\"") != std::string::npos;
        for (std::size_t i = 0; i < graph.num_nodes(); ++i) if (g2.get_label(i) != graph.get_label(i)) return 7;
        std::printf("view_selection on a reloaded table: %s, same labels\n", uploaded ? "uploaded" : "parked");
    }
    /* generate_texture_patches.cpp:469-475: subgraphs of every label == the reference's loop (uni_graph.cpp:21-55) */
    std::size_t n_patches = 0;
    auto reference = [&](std::size_t label) {
        std::vector<std::vector<std::size_t> > want;
        std::vector<bool> used(graph.num_nodes(), false);
        for (std::size_t i = 0; i < graph.num_nodes(); ++i) {
            if (graph.get_label(i) != label || used[i]) continue;
            want.push_back(std::vector<std::size_t>());
            std::vector<std::size_t> queue(1, i); used[i] = true;
            for (std::size_t h = 0; h < queue.size(); ++h) {
                want.back().push_back(queue[h]);
                for (std::size_t a : graph.get_adj_nodes(queue[h]))
                    if (graph.get_label(a) == label && !used[a]) { queue.push_back(a); used[a] = true; }
            }
        }
        return want;
    };
    for (std::size_t label = 0; label <= texture_views.size(); ++label) {
        std::vector<std::vector<std::size_t> > got;
        graph.get_subgraphs(label, &got);      /* one GPU pass for all labels, kept until a label changes: this loop pays it once */
        if (got != reference(label)) return 6;
        n_patches += got.size();
    }
    {   /* a label that changes drops the kept result: the next call sees the new labeling */
        std::size_t const old_label = graph.get_label(0), new_label = (old_label + 1) % (texture_views.size() + 1);
        graph.set_label(0, new_label);
        for (std::size_t label : {old_label, new_label}) {
            std::vector<std::vector<std::size_t> > got;
            graph.get_subgraphs(label, &got);
            if (got != reference(label)) return 8;
        }
        graph.set_label(0, old_label);
    }
    std::printf("patches=%zu\n", n_patches);
    std::printf("ok faces=%zu views=%zu nnz=%zu\n", num_faces, texture_views.size(), data_costs.get_nnz());
    synth_mesh_free(&sm);
    return 0;
}
