// Host-side FP64 vector / 3x3 matrix helpers for the scene front end.
// Operation order follows the reference so that prepared data (matrices, camera
// corners, triangle normals) is bit-identical: src/vector.h:25-57 (dot is
// x*x'+y*y'+z*z' left to right, Normalize multiplies by 1/len),
// src/matrix.cpp:10-133 (row-vector * matrix, cofactor inverse, rotations).
#pragma once
#include <cmath>

namespace qdh {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() = default;
    V3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(const V3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, const V3& a) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3& a, const V3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(const V3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalized(V3 a) {
    const double m = 1. / length(a);   // reciprocal then three multiplies (src/vector.h:29)
    a.x *= m; a.y *= m; a.z *= m;
    return a;
}

struct M3 {
    double m[3][3];
    explicit M3(double diag = 0.) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) m[i][j] = (i == j ? diag : 0.);
    }
};

inline V3 operator*(const V3& v, const M3& a) {
    return V3(v.x * a.m[0][0] + v.y * a.m[1][0] + v.z * a.m[2][0],
              v.x * a.m[0][1] + v.y * a.m[1][1] + v.z * a.m[2][1],
              v.x * a.m[0][2] + v.y * a.m[1][2] + v.z * a.m[2][2]);
}

inline M3 operator*(const M3& a, const M3& b) {
    M3 r(0.);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) r.m[i][j] += a.m[i][k] * b.m[k][j];
    return r;
}

inline double determinant(const M3& a) {
    return a.m[0][0] * a.m[1][1] * a.m[2][2] - a.m[0][0] * a.m[1][2] * a.m[2][1] -
           a.m[0][1] * a.m[1][0] * a.m[2][2] + a.m[0][1] * a.m[1][2] * a.m[2][0] +
           a.m[0][2] * a.m[1][0] * a.m[2][1] - a.m[0][2] * a.m[1][1] * a.m[2][0];
}

inline double minor_signed(const M3& a, int ii, int jj) {
    int r[2], c[2], rc = 0, cc = 0;
    for (int i = 0; i < 3; ++i) if (i != ii) r[rc++] = i;
    for (int j = 0; j < 3; ++j) if (j != jj) c[cc++] = j;
    double t = a.m[r[0]][c[0]] * a.m[r[1]][c[1]] - a.m[r[1]][c[0]] * a.m[r[0]][c[1]];
    return ((ii + jj) % 2) ? -t : t;
}

inline M3 inverse(const M3& a) {
    const double d = determinant(a);
    if (std::fabs(d) < 1e-12) return a;   // singular: returned unchanged (src/matrix.cpp:70-71)
    const double rd = 1. / d;
    M3 r(0.);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = rd * minor_signed(a, j, i);
    return r;
}

inline M3 transpose(const M3& a) {
    M3 r(0.);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}

inline M3 rot_x(double ang) {
    const double S = std::sin(ang), C = std::cos(ang);
    M3 r(1.);
    r.m[1][1] = C; r.m[2][1] = S; r.m[1][2] = -S; r.m[2][2] = C;
    return r;
}
inline M3 rot_y(double ang) {
    const double S = std::sin(ang), C = std::cos(ang);
    M3 r(1.);
    r.m[0][0] = C; r.m[2][0] = -S; r.m[0][2] = S; r.m[2][2] = C;
    return r;
}
inline M3 rot_z(double ang) {
    const double S = std::sin(ang), C = std::cos(ang);
    M3 r(1.);
    r.m[0][0] = C; r.m[1][0] = S; r.m[0][1] = -S; r.m[1][1] = C;
    return r;
}

constexpr double kPi = 3.141592653589793238;          // src/constants.h:10
inline double to_radians(double deg) { return deg / 180. * kPi; }  // src/utils.h:19

struct Transform {   // src/transform.cpp:5-42
    M3 m{1.}, inv{1.}, invT{1.};
    V3 offset;
    void scale(double x, double y, double z) {
        M3 s(x); s.m[1][1] = y; s.m[2][2] = z;
        m = m * s; inv = inverse(m); invT = transpose(inv);
    }
    void rotate(double yaw, double pitch, double roll) {
        m = m * rot_x(to_radians(pitch)) * rot_y(to_radians(yaw)) * rot_z(to_radians(roll));
        inv = inverse(m); invT = transpose(inv);
    }
    void translate(const V3& v) { offset = v; }   // overwrites, does not accumulate
};

}  // namespace qdh
