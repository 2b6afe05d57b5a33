// Stand-in for MVE's math/vector.h, just enough for the reference's tri.{h,cpp} to compile UNCHANGED for oracle/_ref:
// element storage, element access, construction from components and the component-wise difference.  None of these
// involves an operation order, so nothing about the reference's arithmetic is decided here (the inner products etc.
// of the real library are NOT provided: sources that need them stay unbuildable).  Test infrastructure only.
#ifndef MVS_REF_STUB_MATH_VECTOR_H
#define MVS_REF_STUB_MATH_VECTOR_H
#include <algorithm>   // the real header pulls these in; tri.{h,cpp} rely on std::min / std::max / std::abs through it
#include <cmath>
namespace math {
template <typename T, int N>
class Vector {
public:
    Vector() { for (int i = 0; i < N; ++i) v[i] = T(0); }
    Vector(T a, T b) { static_assert(N == 2, "2 components"); v[0] = a; v[1] = b; }
    Vector(T a, T b, T c) { static_assert(N == 3, "3 components"); v[0] = a; v[1] = b; v[2] = c; }
    T& operator[](int i) { return v[i]; }
    T const& operator[](int i) const { return v[i]; }
    Vector operator-(Vector const& o) const { Vector r; for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i]; return r; }
private:
    T v[N];
};
typedef Vector<float, 2> Vec2f;
typedef Vector<float, 3> Vec3f;
}  // namespace math
#endif
