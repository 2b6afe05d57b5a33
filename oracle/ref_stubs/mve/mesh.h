// Stand-in for MVE's mve/mesh.h: the three arrays calculate_data_costs.cpp reads (vertices, face indices, face normals)
// as plain std::vectors with MVE's accessor names.  ensure_normals (called by the reference's prepare_mesh.cpp:63) is
// arithmetic of the absent library, restated as the oracle defines it: n = (b - a) x (c - a), divided by its length
// unless that is zero -- an ASSUMPTION; vertex normals are not read by the path and are not computed.
// Test infrastructure only (oracle/_ref).
#ifndef MVS_REF_STUB_MVE_MESH_H
#define MVS_REF_STUB_MVE_MESH_H
#include <memory>
#include <vector>
#include "math/vector.h"
namespace mve {
class TriangleMesh {
public:
    typedef std::shared_ptr<TriangleMesh> Ptr;
    typedef std::shared_ptr<TriangleMesh const> ConstPtr;
    typedef std::vector<math::Vec3f> VertexList;
    typedef std::vector<math::Vec3f> NormalList;
    typedef std::vector<unsigned int> FaceList;
    static Ptr create() { return Ptr(new TriangleMesh()); }
    VertexList& get_vertices() { return vertices; }
    VertexList const& get_vertices() const { return vertices; }
    FaceList& get_faces() { return faces; }
    FaceList const& get_faces() const { return faces; }
    NormalList& get_face_normals() { return face_normals; }
    NormalList const& get_face_normals() const { return face_normals; }
    void ensure_normals(bool face, bool /*vertex*/) {
        if (!face || face_normals.size() == faces.size() / 3) return;
        face_normals.clear();
        for (std::size_t i = 0; i + 2 < faces.size(); i += 3) {
            math::Vec3f const u = vertices[faces[i + 1]] - vertices[faces[i]], v = vertices[faces[i + 2]] - vertices[faces[i]];
            math::Vec3f n(u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]);
            float const len = n.norm();
            if (len != 0.0f) n = n / len;
            face_normals.push_back(n);
        }
    }
private:
    VertexList vertices;
    FaceList faces;
    NormalList face_normals;
};
}  // namespace mve
#endif
