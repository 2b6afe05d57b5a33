// Stand-in for MVE's mve/camera.h.  CameraInfo::fill_* is camera arithmetic of the absent library; the oracle/_ref
// tests do not pin it: every camera here is the identity (projection = I, world_to_cam = I, position 0, direction +z),
// so that TextureView::get_pixel_coords maps a vertex (x + 0.5, y + 0.5, 1) to the pixel coordinates (x, y) exactly.
#ifndef MVS_REF_STUB_MVE_CAMERA_H
#define MVS_REF_STUB_MVE_CAMERA_H
namespace mve {
struct CameraInfo {
    void fill_calibration(float* k, float, float) const { for (int i = 0; i < 9; ++i) k[i] = (i % 4 == 0) ? 1.0f : 0.0f; }
    void fill_world_to_cam(float* m) const { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    void fill_camera_pos(float* p) const { p[0] = p[1] = p[2] = 0.0f; }
    void fill_viewing_direction(float* d) const { d[0] = d[1] = 0.0f; d[2] = 1.0f; }
};
}  // namespace mve
#endif
