// Stand-in for MVE's math/matrix.h: element storage and access only (tri.cpp fills a Matrix2f element by element).
// See math/vector.h in this directory.
#ifndef MVS_REF_STUB_MATH_MATRIX_H
#define MVS_REF_STUB_MATH_MATRIX_H
namespace math {
template <typename T, int R, int C>
class Matrix {
public:
    Matrix() { for (int i = 0; i < R * C; ++i) m[i] = T(0); }
    T& operator[](int i) { return m[i]; }
    T const& operator[](int i) const { return m[i]; }
private:
    T m[R * C];
};
typedef Matrix<float, 2, 2> Matrix2f;
}  // namespace math
#endif
