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

    if (blur <= 0.) {
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                float p[3]; bmp.get(x, y, p);
                const float h = (p[0] + p[1] + p[2]) / 3;
                heights[(size_t)y * W + x] = h;
                min_y = std::min(min_y, h); max_y = std::max(max_y, h);
            }
    } else {
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                float p[3]; bmp.get(x, y, p);
                const float f = (p[0] + p[1] + p[2]) / 3;
                bmp.set(x, y, f, f, f);
            }
        static float gauss[128][128];
        const int r = std::min(128, nearest_int(3 * blur));
        for (int y = 0; y < r; ++y)
            for (int x = 0; x < r; ++x)
                gauss[y][x] = static_cast<float>(std::exp(-(sqr(x) + sqr(y)) / (2 * sqr(blur))) / (2 * kPi * sqr(blur)));
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                float sum = 0.f;
                for (int dy = -r + 1; dy < r; ++dy)
                    for (int dx = -r + 1; dx < r; ++dx) {
                        float p[3];
                        bmp.get(x + (uint32_t)dx, y + (uint32_t)dy, p);   // unsigned wrap -> black outside
                        sum += gauss[std::abs(dy)][std::abs(dx)] * p[0];
                    }
                heights[(size_t)y * W + x] = sum;
                min_y = std::min(min_y, sum); max_y = std::max(max_y, sum);
            }
    }
    bmin = V3(0, min_y, 0);
    bmax = V3(double(W), max_y, double(H));

    // max of the (up to) four corners of each cell (PopulateMaxHeights)
    maxheights.assign((size_t)W * H, 0.f);
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            float mh = heights[(size_t)y * W + x];
            if (x < W - 1) mh = std::max(mh, heights[(size_t)y * W + x + 1]);
            if (y < H - 1) {
                mh = std::max(mh, heights[(size_t)(y + 1) * W + x]);
                if (x < W - 1) mh = std::max(mh, heights[(size_t)(y + 1) * W + x + 1]);
            }
            maxheights[(size_t)y * W + x] = mh;
        }

    // forward-difference normals (PopulateNormals); last column / row copy their neighbour
    normals.assign((size_t)W * H, V3());
    for (uint32_t y = 0; y + 1 < H; ++y)
        for (uint32_t x = 0; x + 1 < W; ++x) {
            const float h0 = heights[(size_t)y * W + x];
            const float hdx = heights[(size_t)y * W + x + 1];
            const float hdy = heights[(size_t)(y + 1) * W + x];
            const V3 vdx(1, hdx - h0, 0), vdy(0, hdy - h0, 1);
            normals[(size_t)y * W + x] = normalized(cross(vdy, vdx));
        }
    if (W >= 2)
        for (uint32_t y = 0; y < H; ++y) normals[(size_t)y * W + W - 1] = normals[(size_t)y * W + W - 2];
    if (H >= 2)
        for (uint32_t x = 0; x < W; ++x) normals[(size_t)(H - 1) * W + x] = normals[(size_t)(H - 2) * W + x];
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
