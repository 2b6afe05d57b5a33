// Stand-in for MVE's mve/mesh_info.h: the two queries the reference's prepare_mesh.cpp / build_adjacency_graph.cpp make --
// the faces incident to a vertex (mesh_info[v].faces) and the faces containing an edge (get_faces_for_edge, APPENDING to
// the output).  MVE orders a vertex's faces as a fan around the vertex; here they are in ascending face order, which is
// what the oracle assumes.  The order only matters where an edge is shared by three or more faces (non-manifold); on
// manifold meshes the adjacency lists the reference builds are decided by its own edge order v1v2, v2v3, v3v1 alone.
// Test infrastructure only (oracle/_ref).
#ifndef MVS_REF_STUB_MVE_MESH_INFO_H
#define MVS_REF_STUB_MVE_MESH_INFO_H
#include <algorithm>
#include <cstddef>
#include <vector>
#include "mve/mesh.h"
namespace mve {
class MeshInfo {
public:
    typedef std::vector<std::size_t> AdjacentFaces;
    struct VertexInfo { AdjacentFaces faces; };
    MeshInfo() {}
    explicit MeshInfo(TriangleMesh::ConstPtr mesh) { initialize(mesh); }
    void clear() { info.clear(); }
    void initialize(TriangleMesh::ConstPtr mesh) {
        info.assign(mesh->get_vertices().size(), VertexInfo());
        TriangleMesh::FaceList const& f = mesh->get_faces();
        for (std::size_t i = 0; i < f.size(); ++i) {
            AdjacentFaces& l = info[f[i]].faces;
            if (l.empty() || l.back() != i / 3) l.push_back(i / 3);
        }
    }
    VertexInfo const& operator[](std::size_t v) const { return info[v]; }
    std::size_t size() const { return info.size(); }
    void get_faces_for_edge(std::size_t v1, std::size_t v2, std::vector<std::size_t>* out) const {
        AdjacentFaces const& a = info[v1].faces;
        AdjacentFaces const& b = info[v2].faces;
        for (std::size_t k = 0; k < a.size(); ++k)
            if (std::find(b.begin(), b.end(), a[k]) != b.end()) out->push_back(a[k]);
    }
private:
    std::vector<VertexInfo> info;
};
}  // namespace mve
#endif
