// Stand-in for MVE's mve/camera.h.  CameraInfo::fill_* is camera arithmetic of the absent library; oracle/_ref does not
// pin it: a CameraInfo here simply CARRIES the four arrays a TextureView keeps (projection, world_to_cam, position,
// viewing direction), which the test supplies already computed.  Default constructed it is the identity camera
// (projection = I, world_to_cam = I, position 0, direction +z), for which TextureView::get_pixel_coords maps a vertex
// (x + 0.5, y + 0.5, 1) to the pixel coordinates (x, y) exactly.
// The FILE-LEVEL members the reference's generate_texture_views.cpp fills from a .cam file -- trans, rot, flen, dist, paspect, ppoint, the
// two *_from_string setters -- are plain data here (defaults by recollection of MVE: flen 0, paspect 1, ppoint (0.5, 0.5), dist 0,
// identity pose); every camera a TextureView is constructed from is logged (camera_log) so that a test can read what the reference
// parsed out of a file.
#ifndef MVS_REF_STUB_MVE_CAMERA_H
#define MVS_REF_STUB_MVE_CAMERA_H
#include <sstream>
#include <string>
#include <vector>
namespace mve {
struct CameraInfo;
inline std::vector<CameraInfo>& camera_log();
struct CameraInfo {
    float K[9], w2c[16], pos[3], dir[3];
    float flen, dist[2], paspect, ppoint[2], trans[3], rot[9];
    CameraInfo() {
        for (int i = 0; i < 9; ++i) K[i] = (i % 4 == 0) ? 1.0f : 0.0f;
        for (int i = 0; i < 16; ++i) w2c[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        pos[0] = pos[1] = pos[2] = 0.0f;
        dir[0] = dir[1] = 0.0f; dir[2] = 1.0f;
        flen = 0.0f; dist[0] = dist[1] = 0.0f; paspect = 1.0f; ppoint[0] = ppoint[1] = 0.5f;
        trans[0] = trans[1] = trans[2] = 0.0f;
        for (int i = 0; i < 9; ++i) rot[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    }
    void set_translation_from_string(std::string const& s) { std::stringstream ss(s); for (int i = 0; i < 3; ++i) ss >> trans[i]; }
    void set_rotation_from_string(std::string const& s) { std::stringstream ss(s); for (int i = 0; i < 9; ++i) ss >> rot[i]; }
    void fill_calibration(float* k, float, float) const { camera_log().push_back(*this); for (int i = 0; i < 9; ++i) k[i] = K[i]; }   // (called once per TextureView constructed)
    void fill_world_to_cam(float* m) const { for (int i = 0; i < 16; ++i) m[i] = w2c[i]; }
    void fill_camera_pos(float* p) const { for (int i = 0; i < 3; ++i) p[i] = pos[i]; }
    void fill_viewing_direction(float* d) const { for (int i = 0; i < 3; ++i) d[i] = dir[i]; }
};
inline std::vector<CameraInfo>& camera_log() { static std::vector<CameraInfo> l; return l; }
}  // namespace mve
#endif
