// Stand-in for MVE's mve/mesh.h: the three arrays calculate_data_costs.cpp reads (vertices, face indices, face normals)
// as plain std::vectors with MVE's accessor names.  No mesh arithmetic (normals are supplied by the caller).
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
private:
    VertexList vertices;
    FaceList faces;
    NormalList face_normals;
};
}  // namespace mve
#endif
