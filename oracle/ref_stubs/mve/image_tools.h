// Stand-in for MVE's mve/image_tools.h.  desaturate / sobel_edge are image arithmetic of the absent library and are NOT
// pinned by oracle/_ref: desaturate passes the image through and sobel_edge returns the gradient-magnitude image the
// test registered for that image (gradient_registry, or next_gradient_magnitude() as the fallback), so that
// get_face_info can be driven in GMI mode.
#ifndef MVS_REF_STUB_MVE_IMAGE_TOOLS_H
#define MVS_REF_STUB_MVE_IMAGE_TOOLS_H
#include <map>
#include <vector>
#include "mve/image.h"
namespace mve { namespace image {
enum DesaturateType { DESATURATE_MAXIMUM, DESATURATE_LIGHTNESS, DESATURATE_LUMINOSITY, DESATURATE_LUMINANCE, DESATURATE_AVERAGE };
inline ByteImage::Ptr& next_gradient_magnitude() { static ByteImage::Ptr p; return p; }
inline std::map<const void*, ByteImage::Ptr>& gradient_registry() { static std::map<const void*, ByteImage::Ptr> r; return r; }
template <typename T> typename Image<T>::Ptr desaturate(typename Image<T>::ConstPtr img, DesaturateType) { return std::const_pointer_cast<Image<T> >(img); }
template <typename T> typename Image<T>::Ptr sobel_edge(typename Image<T>::ConstPtr img) {
    std::map<const void*, ByteImage::Ptr>::const_iterator it = gradient_registry().find(static_cast<const void*>(img.get()));
    return it != gradient_registry().end() ? it->second : next_gradient_magnitude();
}
// image_undistort_k2k4 / image_undistort_vsfm: pixel arithmetic of the absent library (restated independently for the product's row f4);
// here they only RECORD which model the reference's generate_texture_views.cpp chose, with which parameters, and hand the image back
struct UndistortCall { int model /* 0 k2k4, 1 vsfm */; float flen, d0, d1; };
inline std::vector<UndistortCall>& undistort_log() { static std::vector<UndistortCall> l; return l; }
template <typename T> typename Image<T>::Ptr image_undistort_k2k4(typename Image<T>::ConstPtr img, float flen, float k2, float k4) {
    UndistortCall c = {0, flen, k2, k4}; undistort_log().push_back(c); return std::const_pointer_cast<Image<T> >(img);
}
template <typename T> typename Image<T>::Ptr image_undistort_vsfm(typename Image<T>::ConstPtr img, float flen, float k1) {
    UndistortCall c = {1, flen, k1, 0.0f}; undistort_log().push_back(c); return std::const_pointer_cast<Image<T> >(img);
}
template <typename T> typename Image<T>::Ptr crop(typename Image<T>::ConstPtr img, int, int, int, int, T const*) { return std::const_pointer_cast<Image<T> >(img); }
} }  // namespace mve::image
#endif
