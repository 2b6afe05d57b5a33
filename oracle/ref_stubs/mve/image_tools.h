// Stand-in for MVE's mve/image_tools.h.  desaturate / sobel_edge are image arithmetic of the absent library and are NOT
// pinned by oracle/_ref: generate_gradient_magnitude() simply receives the gradient-magnitude image the test chose
// (set_next_gradient_magnitude), so that get_face_info can be driven in GMI mode.
#ifndef MVS_REF_STUB_MVE_IMAGE_TOOLS_H
#define MVS_REF_STUB_MVE_IMAGE_TOOLS_H
#include "mve/image.h"
namespace mve { namespace image {
enum DesaturateType { DESATURATE_MAXIMUM, DESATURATE_LIGHTNESS, DESATURATE_LUMINOSITY, DESATURATE_LUMINANCE, DESATURATE_AVERAGE };
inline ByteImage::Ptr& next_gradient_magnitude() { static ByteImage::Ptr p; return p; }
template <typename T> typename Image<T>::Ptr desaturate(typename Image<T>::ConstPtr img, DesaturateType) { return std::const_pointer_cast<Image<T> >(img); }
template <typename T> typename Image<T>::Ptr sobel_edge(typename Image<T>::ConstPtr) { return next_gradient_magnitude(); }
template <typename T> typename Image<T>::Ptr crop(typename Image<T>::ConstPtr img, int, int, int, int, T const*) { return std::const_pointer_cast<Image<T> >(img); }
} }  // namespace mve::image
#endif
