// Stand-in for MVE's math/matrix.h: element storage and access, plus the two products TextureView::get_pixel_coords uses.
// The products accumulate left to right (a convention of the absent library); the oracle/_ref tests only ever use
// IDENTITY matrices, for which every accumulation order gives the same, exact result -- projection arithmetic is NOT
// what these tests pin.  See math/vector.h in this directory.
#ifndef MVS_REF_STUB_MATH_MATRIX_H
#define MVS_REF_STUB_MATH_MATRIX_H
#include "math/vector.h"
namespace math {
template <typename T, int R, int C>
class Matrix {
public:
    Matrix() { for (int i = 0; i < R * C; ++i) m[i] = T(0); }
    T& operator[](int i) { return m[i]; }
    T const& operator[](int i) const { return m[i]; }
    T* operator*() { return m; }
    T const* operator*() const { return m; }
    Vector<T, R> operator*(Vector<T, C> const& v) const {
        Vector<T, R> r;
        for (int i = 0; i < R; ++i) { T s = T(0); for (int k = 0; k < C; ++k) s += m[i * C + k] * v[k]; r[i] = s; }
        return r;
    }
    Vector<T, C - 1> mult(Vector<T, C - 1> const& v, T const& w) const {   // upper rows of M * (v, w)
        Vector<T, C - 1> r;
        for (int i = 0; i < C - 1; ++i) { T s = T(0); for (int k = 0; k < C - 1; ++k) s += m[i * C + k] * v[k]; r[i] = s + w * m[i * C + C - 1]; }
        return r;
    }
private:
    T m[R * C];
};
typedef Matrix<float, 2, 2> Matrix2f;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 4> Matrix4f;
}  // namespace math
#endif
