// Device textures, environment lookup and world-space hit assembly.
// Restates src/texture.cpp, src/bitmap.cpp:44-50,293-310, src/environment.cpp:58-108 and
// Node::Intersect's back-transform (src/geometry.cpp:331-338).
#pragma once
#include "dgeom.h"

namespace qd {

struct Info {        // IntersectionInfo, src/geometry.h:13-22
    DV ip, normal;
    double distance, u, v;
    DV rayDir, dNdx, dNdy;
};

QD_DEV Col bmp_pixel(const DScene& S, const qd_bitmap& b, unsigned x, unsigned y) {
    if (b.width == 0 || x >= b.width || y >= b.height) return mkc(0, 0, 0);
    return ldc(S.texels + (b.texel_off + x + (size_t)y * b.width) * 3);
}
QD_DEV Col bmp_bilinear(const DScene& S, const qd_bitmap& b, double x, double y) {
    const unsigned xx = static_cast<unsigned>(floor(x)), yy = static_cast<unsigned>(floor(y));
    const double p = x - xx, q = y - yy;
    const unsigned w = b.width, h = b.height;
    const unsigned xr = w ? (xx + 1) % w : 0, yd = h ? (yy + 1) % h : 0;
    const Col center = bmp_pixel(S, b, xx, yy), right = bmp_pixel(S, b, xr, yy);
    const Col down = bmp_pixel(S, b, xx, yd), dr = bmp_pixel(S, b, xr, yd);
    return center * (float)(1 - p) * (float)(1 - q) + right * (float)p * (float)(1 - q) + down * (float)(1 - p) * (float)q + dr * (float)p * (float)q;
}
QD_DEV Col bitmap_helper_color(const DScene& S, const qd_texture& t, double u, double v) {   // src/texture.cpp:139-199
    const qd_bitmap b = S.bitmaps[t.bitmap];
    const unsigned width = b.width, height = b.height;
    if (width == 0 || height == 0) return mkc(0, 0, 0);
    if (t.bilinear) {
        double x = u * t.scaling * width, y = v * t.scaling * height;
        int xx = static_cast<int>(floor(x)), yy = static_cast<int>(floor(y));
        const double dx = x - xx, dy = y - yy;
        xx = (int)((unsigned)xx % width);      // int %= unsigned: modulo of the two's-complement value (parity trap 28)
        yy = (int)((unsigned)yy % height);
        x = static_cast<double>(xx) + dx; y = static_cast<double>(yy) + dy;
        return bmp_bilinear(S, b, x, y);
    }
    int x = static_cast<int>(floor(u * t.scaling * width));
    int y = static_cast<int>(floor(v * t.scaling * height));
    return bmp_pixel(S, b, (unsigned)x % width, (unsigned)y % height);
}

// RICH = false: the scene holds only constant, checker and Fresnel textures (qd_scene_create checks); the Mandelbrot
// loop, the procedural sines and the bitmap fetches are not compiled into that copy of the caller.
template <bool RICH>
QD_DEV Col tex_sample(const DScene& S, int ti, const Info& info) {
    const qd_texture t = S.textures[ti];
    if (!RICH && t.type != QD_TEX_CONST && t.type != QD_TEX_CHECKER && t.type != QD_TEX_FRESNEL) return mkc(0, 0, 0);
    switch (t.type) {
        case QD_TEX_CONST: return ldc(t.color1);
        case QD_TEX_CHECKER: {
            const int x = (int)floor(info.u * t.scaling / 5.), y = (int)floor(info.v * t.scaling / 5.);
            return ((x + y) % 2 == 0) ? ldc(t.color1) : ldc(t.color2);
        }
        case QD_TEX_MANDELBROT: {
            const double x0 = info.u / t.scaling, y0 = info.v / t.scaling;
            double x = 0, y = 0; unsigned i;
            for (i = 0; i < 128 && sqr(x) + sqr(y) < 4; ++i) { const double tmp = sqr(x) - sqr(y) + x0; y = 2 * x * y + y0; x = tmp; }
            const Col c1 = ldc(t.color1), c2 = ldc(t.color2);
            if (i == 0) return c1;
            if (i < 16) return c1 * 0.75f + c2 * 0.25f;
            if (i < 32) return c1 * 0.5f + c2 * 0.5f;
            if (i < 64) return c1 * 0.25f + c2 * 0.75f;
            return c2;
        }
        case QD_TEX_PROCEDURAL: {
            Col r = mkc(0, 0, 0);
            for (int i = 0; i < t.i1; ++i) {
                const qd_proc_line l = S.proc_lines[t.i0 + i];
                r = r + (ldc(l.color_u) * (float)sin(info.u * l.freq_u) + ldc(l.color_v) * (float)sin(info.v * l.freq_v));
            }
            return r;
        }
        case QD_TEX_BITMAP: return bitmap_helper_color(S, t, info.u, info.v);
        case QD_TEX_FRESNEL: {
            double f;
            if (dot(info.normal, info.rayDir) > 0) f = fresnel_ratio(info.rayDir, -info.normal, 1. / t.p0);
            else f = fresnel_ratio(info.rayDir, info.normal, t.p0);
            const float ff = static_cast<float>(f);
            return mkc(ff, ff, ff);
        }
        default: return mkc(0, 0, 0);
    }
}

QD_DEV void tex_modify_normal(const DScene& S, int ti, Info& info) {
    const qd_texture t = S.textures[ti];
    if (t.type == QD_TEX_BUMPMAP) {                                   // src/texture.cpp:267-274
        const Col bump = bitmap_helper_color(S, t, info.u, info.v);
        info.normal = info.normal + (info.dNdx * (double)bump.r + info.dNdy * (double)bump.g) * t.p0;
        info.normal = normalize(info.normal);
    } else if (t.type == QD_TEX_BUMPS) {                              // src/texture.cpp:276-296
        const float strength = (float)t.p0;
        if (strength <= 0.f) return;
        const float freqX[3] = {0.5f, 1.21f, 1.9f}, freqZ[3] = {0.4f, 1.13f, 1.81f};
        const float fm = 0.2f;
        const float intensity[3] = {0.1f, 0.08f, 0.05f};
        double dx = 0, dy = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dx += sin((double)(fm * freqX[i]) * info.u) * (double)intensity[i] * (double)strength;
            dy += sin((double)(fm * freqZ[i]) * info.v) * (double)intensity[i] * (double)strength;
        }
        info.normal = info.normal + (dx * info.dNdx + dy * info.dNdy);
        info.normal = normalize(info.normal);
    }
}

QD_DEV Col env_side(const DScene& S, int face, double x, double y) {
    const qd_bitmap b = S.bitmaps[S.env_faces[face]];
    const double nx = (x + 1) * 0.5 * (b.width - 1);
    const double ny = (y + 1) * 0.5 * (b.height - 1);
    return S.env_bilinear ? bmp_bilinear(S, b, nx, ny) : bmp_pixel(S, b, static_cast<unsigned>(nx), static_cast<unsigned>(ny));
}
QD_DEV Col env_lookup(const DScene& S, const DV& vec) {               // src/environment.cpp:70-108
    double mv = fabs(vec.x); int md = 0;
    if (fabs(vec.y) > mv) { mv = fabs(vec.y); md = 1; }
    if (fabs(vec.z) > mv) { mv = fabs(vec.z); md = 2; }
    const DV t = vec / fabs(comp(vec, md));
    const int state = ((comp(t, md) < 0) ? 0 : 3) + md;
    switch (state) {
        case 0: return env_side(S, 0, t.z, -t.y);
        case 1: return env_side(S, 1, t.x, -t.z);
        case 2: return env_side(S, 2, t.x, t.y);
        case 3: return env_side(S, 3, -t.z, -t.y);
        case 4: return env_side(S, 4, t.x, t.z);
        default: return env_side(S, 5, t.x, -t.y);
    }
}

// Per-node object-space ray (Node::Intersect, src/geometry.cpp:321-327)
struct ObjRay { DV o, d; double len; };
QD_DEV ObjRay to_object(const qd_node& n, const DV& start, const DV& dir) {
    ObjRay r;
    r.o = mul_mat(start - ld3(n.offset), n.Minv);
    r.d = mul_mat(dir, n.Minv);
    r.len = len(r.d);
    r.d = normalize(r.d);
    return r;
}

}  // namespace qd
