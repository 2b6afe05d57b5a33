// Stand-in for MVE's math/functions.h: the reference's util.h uses math::clamp in a debug colour helper (get_jet_color)
// that oracle/_ref never calls.
#ifndef MVS_REF_STUB_MATH_FUNCTIONS_H
#define MVS_REF_STUB_MATH_FUNCTIONS_H
namespace math {
template <typename T> T clamp(T const& v, T const& lo = T(0), T const& hi = T(1)) { return v < lo ? lo : (hi < v ? hi : v); }
}  // namespace math
#endif
