// Stand-in for MVE's mve/scene.h + view.h: just enough for the reference's generate_texture_views.cpp (from_mve_scene) to COMPILE;
// oracle/_ref never opens an MVE scene (Scene::create throws).  Test infrastructure only.
#ifndef MVS_REF_STUB_MVE_SCENE_H
#define MVS_REF_STUB_MVE_SCENE_H
#include <memory>
#include <string>
#include <vector>
#include "mve/camera.h"
#include "util/exception.h"
#include "util/file_system.h"
namespace mve {
enum ImageType { IMAGE_TYPE_UNKNOWN, IMAGE_TYPE_UINT8, IMAGE_TYPE_FLOAT };
class View {
public:
    typedef std::shared_ptr<View> Ptr;
    struct ImageProxy { std::string filename; int channels; ImageProxy() : channels(0) {} };
    bool has_image(std::string const&, ImageType) const { return false; }
    ImageProxy const* get_image_proxy(std::string const&) const { return nullptr; }
    std::string get_name() const { return std::string(); }
    std::string get_directory() const { return std::string(); }
    int get_id() const { return 0; }
    CameraInfo const& get_camera() const { return cam; }
private:
    CameraInfo cam;
};
class Scene {
public:
    typedef std::shared_ptr<Scene> Ptr;
    typedef std::vector<View::Ptr> ViewList;
    static Ptr create(std::string const&) { throw util::Exception("oracle/_ref: MVE scenes are not supported by the stand-in"); }
    ViewList const& get_views() const { return views; }
    View::Ptr get_view_by_id(std::size_t i) { return i < views.size() ? views[i] : View::Ptr(); }
private:
    ViewList views;
};
}  // namespace mve
#endif
