// Heightfield preparation on the host: gaussian blur, max-heights, normals and the
// 2^k "high map" skip structure.  Follows src/heightfield.cpp:102-214 and :260-290.
#include "heightfield.h"

#include <algorithm>
#include <cmath>

namespace qdh {

namespace {
const float kLargeFloat = 1e17f;   // src/constants.h:4
inline double sqr(double a) { return a * a; }
inline int nearest_int(double x) { return (int)std::floor(x + 0.5); }
inline int clampi(int x, int a, int b) { return std::min(std::max(x, a), b); }
}  // namespace

PrepHooks g_prep_hooks;
int g_prep_hook_calls = 0;

bool HeightfieldData::prepare(Image& bmp, double blur, std::string& err) {
    width = bmp.w; height = bmp.h;
    const uint32_t W = width, H = height;
    heights.assign((size_t)W * H, 0.f);
    float min_y = kLargeFloat, max_y = -kLargeFloat;

    const int r_dev = blur <= 0. ? 0 : std::min(128, nearest_int(3 * blur));
    if (g_prep_hooks.heightfield && W && H && bmp.ok() && (blur <= 0. || r_dev >= 1)) {
        // the same passes as below on the device; the gaussian weights need libm's exp and are made here
        std::vector<float> gauss((size_t)r_dev * r_dev);
        for (int y = 0; y < r_dev; ++y)
            for (int x = 0; x < r_dev; ++x)
                gauss[(size_t)y * r_dev + x] = static_cast<float>(std::exp(-(sqr(x) + sqr(y)) / (2 * sqr(blur))) / (2 * kPi * sqr(blur)));
        maxheights.assign((size_t)W * H, 0.f);
        std::vector<double> nrm((size_t)W * H * 3);
        if (g_prep_hooks.heightfield(W, H, bmp.rgb.data(), gauss.data(), r_dev, heights.data(), maxheights.data(), nrm.data()) != 0) {
            err = "device heightfield preparation failed";
            return false;
        }
        ++g_prep_hook_calls;
        normals.resize((size_t)W * H);
        for (size_t i = 0; i < normals.size(); ++i) normals[i] = V3(nrm[i * 3], nrm[i * 3 + 1], nrm[i * 3 + 2]);
        for (float h : heights) { min_y = std::min(min_y, h); max_y = std::max(max_y, h); }
        bmin = V3(0, min_y, 0);
        bmax = V3(double(W), max_y, double(H));
        return true;
    }

    // CPU version of the same per-cell passes the device kernels run (csrc/device/prep.h): a grey plane, the blurred
    // heights, then one function of the cell index for the max-heights and one for the normals.  The order of the float
    // additions of the blur (rows of the window top to bottom, left to right inside a row) is what fixes the bits.
    const size_t cells = (size_t)W * H;
    std::vector<float> grey(cells);
    for (size_t i = 0; i < cells; ++i) {
        const float* px = bmp.ok() ? &bmp.rgb[i * 3] : nullptr;
        grey[i] = px ? (px[0] + px[1] + px[2]) / 3 : 0.f;
    }
    if (blur <= 0.) heights = grey;
    else {
        const int r = r_dev;                                     // window half-width: weights for offsets 0 .. r-1
        std::vector<float> weight((size_t)std::max(r, 0) * std::max(r, 0));
        for (int j = 0; j < r; ++j)
            for (int i = 0; i < r; ++i)
                weight[(size_t)j * r + i] = static_cast<float>(std::exp(-(sqr(i) + sqr(j)) / (2 * sqr(blur))) / (2 * kPi * sqr(blur)));
        auto blurred = [&](uint32_t cx, uint32_t cy) {
            float acc = 0.f;
            for (int oy = 1 - r; oy < r; ++oy) {
                const uint32_t sy = cy + (uint32_t)oy;           // wraps below zero: treated as outside, which reads black
                const float* wrow = &weight[(size_t)std::abs(oy) * r];
                for (int ox = 1 - r; ox < r; ++ox) {
                    const uint32_t sx = cx + (uint32_t)ox;
                    const float sample = (sx < W && sy < H) ? grey[(size_t)sy * W + sx] : 0.f;
                    acc += wrow[std::abs(ox)] * sample;
                }
            }
            return acc;
        };
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) heights[(size_t)y * W + x] = blurred(x, y);
    }
    for (float h : heights) { min_y = std::min(min_y, h); max_y = std::max(max_y, h); }
    bmin = V3(0, min_y, 0);
    bmax = V3(double(W), max_y, double(H));

    // per cell: the highest of its (up to four) corner heights, and the normal of the forward-difference plane of the
    // nearest cell that has both forward neighbours (the last column and row repeat the one before)
    auto corner_max = [&](uint32_t x, uint32_t y) {
        const size_t i = (size_t)y * W + x;
        float top = heights[i];
        const bool right = x + 1 < W, below = y + 1 < H;
        if (right) top = std::max(top, heights[i + 1]);
        if (below) top = std::max(top, heights[i + W]);
        if (right && below) top = std::max(top, heights[i + W + 1]);
        return top;
    };
    auto cell_normal = [&](uint32_t x, uint32_t y) {
        if (W < 2 || H < 2) return V3();
        const size_t i = (size_t)std::min(y, H - 2) * W + std::min(x, W - 2);
        const float here = heights[i];
        const V3 along_x(1, heights[i + 1] - here, 0), along_z(0, heights[i + W] - here, 1);
        return normalized(cross(along_z, along_x));
    };
    maxheights.resize(cells);
    normals.resize(cells);
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            maxheights[(size_t)y * W + x] = corner_max(x, y);
            normals[(size_t)y * W + x] = cell_normal(x, y);
        }
    return true;
}

bool HeightfieldData::build_highmap(std::string& err) {
    const uint32_t W = width, H = height;
    max_k = (int)std::ceil(std::log((double)std::max(W, H)) / std::log(2.0));
    if (max_k > 16) max_k = 16;   // OptimizationHighMap holds 16 levels (src/heightfield.h:39)
    highmap.assign((size_t)W * H * 16, 0.f);
    if (g_prep_hooks.highmap && W && H) {
        if (g_prep_hooks.highmap(W, H, heights.data(), max_k, highmap.data()) != 0) { err = "device high-map preparation failed"; return false; }
        ++g_prep_hook_calls;
        return true;
    }
    auto get_h = [&](int x, int y) { x = clampi(x, 0, (int)W - 1); y = clampi(y, 0, (int)H - 1); return heights[(size_t)y * W + x]; };
    auto get_hi = [&](int x, int y, int k) { x = clampi(x, 0, (int)W - 1); y = clampi(y, 0, (int)H - 1); return highmap[((size_t)y * W + x) * 16 + k]; };
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            float h = get_h((int)x, (int)y);
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) h = std::max(h, get_h((int)x + dx, (int)y + dy));
            highmap[((size_t)y * W + x) * 16 + 0] = h;
        }
    for (int k = 1; k < max_k; ++k)
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                const int off = 1 << (k - 1);
                const float a = get_hi((int)x - off, (int)y - off, k - 1), b = get_hi((int)x + off, (int)y - off, k - 1);
                const float c = get_hi((int)x - off, (int)y + off, k - 1), d = get_hi((int)x + off, (int)y + off, k - 1);
                highmap[((size_t)y * W + x) * 16 + k] = std::max(std::max(a, b), std::max(c, d));
            }
    return true;
}

}  // namespace qdh
