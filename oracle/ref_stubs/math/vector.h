// Stand-in for MVE's math/vector.h, just enough for the reference's tri.{h,cpp} and texture_view.{h,cpp} to compile
// UNCHANGED for oracle/_ref: element storage and access, construction, conversion, and COMPONENT-WISE operators.
// A component-wise operator has no operation order to decide, with one exception that is a convention of the absent
// library and therefore an assumption here (the same one the oracle makes): `vector / scalar` divides every component
// (it does not multiply by a reciprocal).
// For the reference's calculate_data_costs.cpp (its culls and ray set-up) the inner product, norm and normalisation are
// provided as well; their operation order IS a convention of the absent library and is an assumption here, the one the
// oracle states: dot = ((a0 b0 + a1 b1) + a2 b2), norm = sqrt(dot(v, v)), normalize divides every component by norm().
// What oracle/_ref pins with them is the reference's CONTROL FLOW around this arithmetic, not the arithmetic.
// Test infrastructure only.
#ifndef MVS_REF_STUB_MATH_VECTOR_H
#define MVS_REF_STUB_MATH_VECTOR_H
#include <algorithm>   // the real header pulls these in; tri.{h,cpp} / texture_view.cpp rely on std::min / std::max / std::abs / std::swap through it
#include <cmath>
#include <cstdint>
#include <iostream>
#include <limits>
namespace math {
template <typename T, int N>
class Vector {
public:
    Vector() { for (int i = 0; i < N; ++i) v[i] = T(0); }
    explicit Vector(T const& a) { for (int i = 0; i < N; ++i) v[i] = a; }
    Vector(T const& a, T const& b) { static_assert(N == 2, "2 components"); v[0] = a; v[1] = b; }
    Vector(T const& a, T const& b, T const& c) { static_assert(N == 3, "3 components"); v[0] = a; v[1] = b; v[2] = c; }
    Vector(T const& a, T const& b, T const& c, T const& d) { static_assert(N == 4, "4 components"); v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
    template <typename U> Vector(Vector<U, N> const& o) { for (int i = 0; i < N; ++i) v[i] = T(o[i]); }
    T& operator[](int i) { return v[i]; }
    T const& operator[](int i) const { return v[i]; }
    T* operator*() { return v; }
    T const* operator*() const { return v; }
    Vector operator-(Vector const& o) const { Vector r; for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i]; return r; }
    Vector operator+(Vector const& o) const { Vector r; for (int i = 0; i < N; ++i) r.v[i] = v[i] + o.v[i]; return r; }
    Vector& operator+=(Vector const& o) { for (int i = 0; i < N; ++i) v[i] += o.v[i]; return *this; }
    Vector operator/(T const& s) const { Vector r; for (int i = 0; i < N; ++i) r.v[i] = v[i] / s; return r; }
    Vector& operator/=(T const& s) { for (int i = 0; i < N; ++i) v[i] /= s; return *this; }
    Vector operator*(T const& s) const { Vector r; for (int i = 0; i < N; ++i) r.v[i] = v[i] * s; return r; }
    T& operator()(int i) { return v[i]; }
    T const& operator()(int i) const { return v[i]; }
    T dot(Vector const& o) const { T s = v[0] * o.v[0]; for (int i = 1; i < N; ++i) s = s + v[i] * o.v[i]; return s; }
    T square_norm() const { return dot(*this); }
    T norm() const { return std::sqrt(square_norm()); }
    Vector& normalize() { T const n = norm(); for (int i = 0; i < N; ++i) v[i] /= n; return *this; }
    Vector normalized() const { return Vector(*this).normalize(); }
private:
    T v[N];
};
typedef Vector<float, 2> Vec2f;
typedef Vector<float, 3> Vec3f;
typedef Vector<float, 4> Vec4f;
typedef Vector<double, 3> Vec3d;
typedef Vector<int, 2> Vec2i;
typedef Vector<unsigned char, 3> Vec3uc;
}  // namespace math
#endif
