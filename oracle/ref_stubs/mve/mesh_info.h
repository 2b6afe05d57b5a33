// Stand-in for MVE's mve/mesh_info.h: texturing.h only names the type in declarations oracle/_ref never calls.
#ifndef MVS_REF_STUB_MVE_MESH_INFO_H
#define MVS_REF_STUB_MVE_MESH_INFO_H
namespace mve { class MeshInfo; }
#endif
