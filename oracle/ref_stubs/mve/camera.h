// Stand-in for MVE's mve/camera.h.  CameraInfo::fill_* is camera arithmetic of the absent library; oracle/_ref does not
// pin it: a CameraInfo here simply CARRIES the four arrays a TextureView keeps (projection, world_to_cam, position,
// viewing direction), which the test supplies already computed.  Default constructed it is the identity camera
// (projection = I, world_to_cam = I, position 0, direction +z), for which TextureView::get_pixel_coords maps a vertex
// (x + 0.5, y + 0.5, 1) to the pixel coordinates (x, y) exactly.
#ifndef MVS_REF_STUB_MVE_CAMERA_H
#define MVS_REF_STUB_MVE_CAMERA_H
namespace mve {
struct CameraInfo {
    float K[9], w2c[16], pos[3], dir[3];
    CameraInfo() {
        for (int i = 0; i < 9; ++i) K[i] = (i % 4 == 0) ? 1.0f : 0.0f;
        for (int i = 0; i < 16; ++i) w2c[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        pos[0] = pos[1] = pos[2] = 0.0f;
        dir[0] = dir[1] = 0.0f; dir[2] = 1.0f;
    }
    void fill_calibration(float* k, float, float) const { for (int i = 0; i < 9; ++i) k[i] = K[i]; }
    void fill_world_to_cam(float* m) const { for (int i = 0; i < 16; ++i) m[i] = w2c[i]; }
    void fill_camera_pos(float* p) const { for (int i = 0; i < 3; ++i) p[i] = pos[i]; }
    void fill_viewing_direction(float* d) const { for (int i = 0; i < 3; ++i) d[i] = dir[i]; }
};
}  // namespace mve
#endif
