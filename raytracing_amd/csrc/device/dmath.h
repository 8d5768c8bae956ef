// Device-side FP64 vector / FP32 colour helpers.  Operation order is the reference's
// (src/vector.h:25-57, src/vector.cpp:56-90, src/matrix.cpp:10-17, src/color.h:36-41):
// dot = x*x' + y*y' + z*z' left to right, Normalize multiplies by one reciprocal, every
// colour * scalar narrows the scalar to float first.  Built with -ffp-contract=off so
// no FMA is formed (the reference is a baseline x86-64 build without FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QD_DEV __device__ __forceinline__

// A reference to constant data that every lane of the wave addresses alike (the scene description, the launch parameters),
// for a function that is not inlined: the address is made wave-uniform and the data read from the constant address space,
// i.e. with scalar loads (see kernarg_scene in qdtrace.hip).
template <class T> QD_DEV const T& uniform_const(const T& r) {
    const uint64_t a = (uint64_t)&r;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    typedef const T __attribute__((address_space(4))) * ConstPtr;
    return *(const T*)(ConstPtr)(((uint64_t)hi << 32) | (uint64_t)lo);
}
namespace qd {

constexpr double kPi = 3.141592653589793238;   // src/constants.h:10
constexpr double kInf = 1e99;                  // src/constants.h:11

struct DV { double x, y, z; };
QD_DEV DV mk(double x, double y, double z) { DV v; v.x = x; v.y = y; v.z = z; return v; }
QD_DEV DV operator+(const DV& a, const DV& b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
QD_DEV DV operator-(const DV& a, const DV& b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
QD_DEV DV operator-(const DV& a) { return mk(-a.x, -a.y, -a.z); }
QD_DEV DV operator*(const DV& a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
QD_DEV DV operator*(double s, const DV& a) { return mk(a.x * s, a.y * s, a.z * s); }
QD_DEV DV operator/(const DV& a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
QD_DEV double dot(const DV& a, const DV& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
QD_DEV DV cross(const DV& a, const DV& b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
QD_DEV double len_sqr(const DV& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
QD_DEV double dsqrt(double x) { return __dsqrt_rn(x); }   // must be correctly rounded to match sqrtsd
QD_DEV double len(const DV& a) { return dsqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
QD_DEV DV normalize(DV a) { const double m = 1. / len(a); a.x *= m; a.y *= m; a.z *= m; return a; }
QD_DEV double comp(const DV& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
QD_DEV double sqr(double a) { return a * a; }
// std::min / std::max as the reference calls them: (b < a) ? b : a and (a < b) ? b : a.  Unlike fmin / fmax they hand a NaN
// first argument through, which decides where the heightfield DDA of a ray with dir.x == 0 on a grid line stops.
QD_DEV double smin(double a, double b) { return b < a ? b : a; }
QD_DEV double smax(double a, double b) { return a < b ? b : a; }
template <class P> QD_DEV DV ld3(P p) { return mk(p[0], p[1], p[2]); }
QD_DEV DV mul_mat(const DV& v, const double* m) {
    return mk(v.x * m[0] + v.y * m[3] + v.z * m[6], v.x * m[1] + v.y * m[4] + v.z * m[7], v.x * m[2] + v.y * m[5] + v.z * m[8]);
}
QD_DEV bool is_between(double x, double lo, double hi, double eps) { return lo - eps <= x && x <= hi + eps; }
QD_DEV bool is_zero(double a, double eps) { return -eps <= a && a <= eps; }
QD_DEV int sign_of(double x) { return x > 1e-6 ? +1 : (x < -1e-6 ? -1 : 0); }

QD_DEV DV reflect(DV in, const DV& n) {                         // src/vector.cpp:56-62
    in = normalize(in);
    in = in - (2 * n) * dot(n, in);
    return normalize(in);
}
QD_DEV DV faceforward(const DV& ray, const DV& n) { return dot(ray, n) < 0 ? n : -n; }
QD_DEV DV refract(const DV& in, const DV& n, double ratio) {    // src/vector.cpp:72-82
    const double ndoti = dot(in, n);
    const double k = 1. - sqr(ratio) * (1. - sqr(ndoti));
    DV r = mk(0, 0, 0);
    if (k >= 0.) r = ratio * in - (ratio * ndoti + dsqrt(k)) * n;
    return r;
}
QD_DEV double fresnel_ratio(const DV& in, const DV& n, double ratio) {   // src/vector.cpp:84-90
    const double f = sqr((1. - ratio) / (1 + ratio));
    return f + (1. - f) * pow(1. + dot(in, n), 5.);
}
QD_DEV void orthonormal_system(const DV& in, DV& r1, DV& r2) {          // src/utils.cpp:49-61
    const DV s0 = mk(-0.267261242, 0.534522484, -0.801783726), s1 = mk(+0.483368245, 0.096673649, +0.870062840);
    r1 = fabs(dot(in, s0)) > 0.99 ? cross(in, s1) : cross(in, s0);
    r1 = normalize(r1);
    r2 = normalize(cross(in, r1));
}

struct Col { float r, g, b; };
QD_DEV Col mkc(float r, float g, float b) { Col c; c.r = r; c.g = g; c.b = b; return c; }
QD_DEV Col operator+(const Col& a, const Col& b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }
QD_DEV Col operator-(const Col& a, const Col& b) { return mkc(a.r - b.r, a.g - b.g, a.b - b.b); }
QD_DEV Col operator*(const Col& a, const Col& b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }
QD_DEV Col operator*(const Col& a, float m) { return mkc(a.r * m, a.g * m, a.b * m); }
QD_DEV Col operator/(const Col& a, float d) { return mkc(a.r / d, a.g / d, a.b / d); }
QD_DEV Col ldc(const float* p) { return mkc(p[0], p[1], p[2]); }

// Path-hashed random stream for glossy reflection: identical to oracle.cpp's rng_mode 1.
QD_DEV uint32_t mix32(uint32_t h) { h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16; return h; }
QD_DEV uint32_t path_child(uint32_t parent, uint32_t slot, uint32_t sample) {
    return mix32(parent ^ mix32(0x9e3779b9U + slot * 0x85ebca6bU + sample * 0xc2b2ae35U));
}
QD_DEV double random01(uint32_t path, uint32_t sample, uint32_t draw) {
    const uint32_t h = mix32(path ^ mix32(0x68bc21ebU + sample * 0x02e5be93U + draw * 0x967a889bU));
    return (h & 0x7fffffffU) / 2147483647.0;
}
QD_DEV uint32_t pixel_path(int x, int y, int sub) { return mix32((uint32_t)x * 0x9e3779b1U ^ mix32((uint32_t)y * 0x85ebca77U + (uint32_t)sub)); }

}  // namespace qd
