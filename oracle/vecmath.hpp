// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// Dependency-free stand-ins for the few Eigen operations the reference hot path uses
// (Eigen is not available in the build container).
//   * Quaterniond(w,x,y,z).toRotationMatrix()   (Eigen/src/Geometry/Quaternion.h)
//   * Vector3d::normalized()  -> returns the input unchanged when the squared norm is 0
//   * fixed-size mat*vec products: sum in k = 0,1,2 order
#pragma once
#include <cmath>
#include <cfloat>
#include <algorithm>

namespace orc {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() = default;
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(const V3 &a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, const V3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(const V3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(const V3 &a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 &operator+=(V3 &a, const V3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double sqnorm(const V3 &a) { return dot(a, a); }
inline double norm(const V3 &a) { return std::sqrt(sqnorm(a)); }
// Eigen: normalized() divides by sqrt(squaredNorm) only when squaredNorm > 0.
inline V3 normalized(const V3 &a) {
    double z = sqnorm(a);
    if (z > 0.0) return a / std::sqrt(z);
    return a;
}
inline V3 vabs(const V3 &a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }

struct V4 {
    double w = 0, x = 0, y = 0, z = 0;   // quaternion order of the reference: (w,x,y,z) = quat(0..3)
    double operator[](int i) const { return i == 0 ? w : (i == 1 ? x : (i == 2 ? y : z)); }
    double &operator[](int i) { return i == 0 ? w : (i == 1 ? x : (i == 2 ? y : z)); }
};

struct M3 {
    double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    static M3 identity() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0; return r; }
};
inline V3 mul(const M3 &A, const V3 &v) {
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
            A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
// A^T v
inline V3 mulT(const M3 &A, const V3 &v) {
    return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z,
            A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
            A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
inline M3 mul(const M3 &A, const M3 &B) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return r;
}
inline M3 transpose(const M3 &A) {
    M3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[j][i];
    return r;
}

// Eigen::Quaterniond(w,x,y,z).toRotationMatrix(); no normalisation of the quaternion.
inline M3 quat_to_rot(const V4 &q) {
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 R;
    R.m[0][0] = 1.0 - (tyy + tzz); R.m[0][1] = txy - twz;         R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz;         R.m[1][1] = 1.0 - (txx + tzz); R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy;         R.m[2][1] = tyz + twx;         R.m[2][2] = 1.0 - (txx + tyy);
    return R;
}

} // namespace orc
