// Stand-in for MVE's math/functions.h: the reference's util.h uses math::clamp in a debug colour helper (get_jet_color)
// that oracle/_ref never calls.  MATH_DEG2RAD (math/defines.h in MVE) is a DOUBLE expression, as the oracle assumes:
// calculate_data_costs.cpp:187 compares acos(float) with 75.0f * (pi / 180.0) in double.
#ifndef MVS_REF_STUB_MATH_FUNCTIONS_H
#define MVS_REF_STUB_MATH_FUNCTIONS_H
#define MATH_PI 3.14159265358979323846264338327950288
#define MATH_DEG2RAD(x) ((x) * (MATH_PI / 180.0))
namespace math {
template <typename T> T clamp(T const& v, T const& lo = T(0), T const& hi = T(1)) { return v < lo ? lo : (hi < v ? hi : v); }
}  // namespace math
#endif
