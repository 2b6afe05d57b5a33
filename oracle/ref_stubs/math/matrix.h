// Stand-in for MVE's math/matrix.h: element storage and access, plus the two products TextureView::get_pixel_coords uses.
// The products accumulate left to right, the homogeneous term last (a convention of the absent library and an assumption
// here, the one the oracle states in pixel_coords).  The TextureView unit pins use IDENTITY matrices, for which every
// accumulation order gives the same, exact result; the whole-path pin (ref_calculate_data_costs) uses real cameras and
// therefore pins the reference's control flow GIVEN this convention, not the convention.  See math/vector.h here.
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
    T& operator()(int r, int c) { return m[r * C + c]; }
    T const& operator()(int r, int c) const { return m[r * C + c]; }
    T* operator*() { return m; }
    T const* operator*() const { return m; }
    Vector<T, R> operator*(Vector<T, C> const& v) const {
        Vector<T, R> r;
        for (int i = 0; i < R; ++i) { T s = m[i * C] * v[0]; for (int k = 1; k < C; ++k) s = s + m[i * C + k] * v[k]; r[i] = s; }
        return r;
    }
    Vector<T, C - 1> mult(Vector<T, C - 1> const& v, T const& w) const {   // upper rows of M * (v, w)
        Vector<T, C - 1> r;
        for (int i = 0; i < C - 1; ++i) { T s = m[i * C] * v[0]; for (int k = 1; k < C - 1; ++k) s = s + m[i * C + k] * v[k]; r[i] = s + w * m[i * C + C - 1]; }
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
