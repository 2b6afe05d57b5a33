// Stand-in for MVE's mve/image_color.h.  The conversion is arithmetic of the absent library, restated here exactly as
// the oracle restates it (ITU-R BT.601 weights, +0.5 offsets on Cb / Cr, left-to-right sums) -- an ASSUMPTION: what
// oracle/_ref pins is WHERE the reference applies it (after the zero-quality cull, before outlier detection).
#ifndef MVS_REF_STUB_MVE_IMAGE_COLOR_H
#define MVS_REF_STUB_MVE_IMAGE_COLOR_H
namespace mve { namespace image {
template <typename T> void color_rgb_to_ycbcr(T* v) {
    T out[3];
    out[0] = v[0] * T(0.299) + v[1] * T(0.587) + v[2] * T(0.114);
    out[1] = v[0] * T(-0.168736) + v[1] * T(-0.331264) + v[2] * T(0.5) + T(0.5);
    out[2] = v[0] * T(0.5) + v[1] * T(-0.418688) + v[2] * T(-0.081312) + T(0.5);
    v[0] = out[0]; v[1] = out[1]; v[2] = out[2];
}
} }  // namespace mve::image
#endif
