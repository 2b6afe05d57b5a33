// Stand-in for MVE's mve/bundle_io.h: just enough for the reference's generate_texture_views.cpp (from_nvm_scene) to COMPILE; oracle/_ref
// never loads a bundle (load_nvm_bundle throws).  Test infrastructure only.
#ifndef MVS_REF_STUB_MVE_BUNDLE_IO_H
#define MVS_REF_STUB_MVE_BUNDLE_IO_H
#include <memory>
#include <string>
#include <vector>
#include "mve/camera.h"
#include "util/exception.h"
namespace mve {
struct NVMCameraInfo { std::string filename; float radial_distortion; };
class Bundle {
public:
    typedef std::shared_ptr<Bundle> Ptr;
    typedef std::vector<CameraInfo> Cameras;
    Cameras& get_cameras() { return cameras; }
private:
    Cameras cameras;
};
inline Bundle::Ptr load_nvm_bundle(std::string const&, std::vector<NVMCameraInfo>*) { throw util::Exception("oracle/_ref: bundles are not supported by the stand-in"); }
}  // namespace mve
#endif
