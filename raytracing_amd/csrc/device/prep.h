// Scene preparation on the device: the table passes the reference runs in BeginRender -- heightfield
// grey / gaussian blur / max-heights / normals (src/heightfield.cpp:102-214), the 2^k "high map" (:260-290) and
// Bitmap::Differentiate for bump maps (src/bitmap.cpp:312-326) -- as HIP kernels behind host-pointer entry points, so that
// the host front end (libqdhost, qdh_set_prep_hooks) can hand them its buffers.  Every table is bit-identical to the host
// code's: one thread per cell runs that cell's arithmetic in the host's order (float sums of the blur ascending in
// dy, dx; -ffp-contract=off; correctly rounded division and sqrt), and what needs libm (the gaussian weights, exp) is
// computed by the caller and passed in.  Layout: cells row-major as in the scene description; the blur reads a grey
// plane through the caches (a 512 x 512 field is 1 MB: it lives in L2), the high map is the scene's 16-floats-per-cell
// array, one launch per level because level k reads level k - 1 of neighbours 2^(k-1) cells away.
#pragma once

namespace {

// std::max(a, b) as the host code evaluates it
__device__ __forceinline__ float prep_max(float a, float b) { return a < b ? b : a; }

// plain mean of the channels, Color::intensity (src/color.h:18)
__global__ void prep_grey_kernel(const float* __restrict__ rgb, float* __restrict__ grey, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    grey[i] = (rgb[i * 3] + rgb[i * 3 + 1] + rgb[i * 3 + 2]) / 3;
}

// heights = sum over the (2r-1)^2 window of gauss[|dy|][|dx|] * grey, black outside (src/heightfield.cpp:112-122)
__global__ void prep_blur_kernel(const float* __restrict__ grey, const float* __restrict__ gauss, int r, uint32_t W, uint32_t H,
                                 float* __restrict__ heights) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    float sum = 0.f;
    for (int dy = -r + 1; dy < r; ++dy) {
        const uint32_t yy = y + (uint32_t)dy;                    // unsigned wrap: outside reads black, as Bitmap::GetPixel
        const float* grow = gauss + (size_t)(dy < 0 ? -dy : dy) * r;
        for (int dx = -r + 1; dx < r; ++dx) {
            const uint32_t xx = x + (uint32_t)dx;
            const float p = (xx < W && yy < H) ? grey[(size_t)yy * W + xx] : 0.f;
            sum += grow[dx < 0 ? -dx : dx] * p;
        }
    }
    heights[(size_t)y * W + x] = sum;
}

// PopulateMaxHeights + PopulateNormals (src/heightfield.cpp:170-214): max of the cell's (up to) four corners; forward
// difference normal, the last column / row repeating their neighbour's (the column is copied first, then the row)
__global__ void prep_cells_kernel(const float* __restrict__ heights, uint32_t W, uint32_t H, float* __restrict__ maxheights,
                                  double* __restrict__ normals) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    float mh = heights[i];
    if (x < W - 1) mh = prep_max(mh, heights[i + 1]);
    if (y < H - 1) {
        mh = prep_max(mh, heights[i + W]);
        if (x < W - 1) mh = prep_max(mh, heights[i + W + 1]);
    }
    maxheights[i] = mh;
    double nx = 0., ny = 0., nz = 0.;
    if (W >= 2 && H >= 2) {
        const uint32_t cx = x < W - 1 ? x : W - 2, cy = y < H - 1 ? y : H - 2;
        const size_t c = (size_t)cy * W + cx;
        const float h0 = heights[c], hdx = heights[c + 1], hdy = heights[c + W];
        const double ax = 0., ay = (double)(hdy - h0), az = 1.;   // vdy
        const double bx = 1., by = (double)(hdx - h0), bz = 0.;   // vdx
        nx = ay * bz - az * by; ny = az * bx - ax * bz; nz = ax * by - ay * bx;      // cross(vdy, vdx)
        const double m = 1. / __dsqrt_rn(nx * nx + ny * ny + nz * nz);
        nx *= m; ny *= m; nz *= m;
    }                                                             // a field one cell wide or high has no interior: zero
    normals[i * 3] = nx; normals[i * 3 + 1] = ny; normals[i * 3 + 2] = nz;
}

__device__ __forceinline__ uint32_t prep_clamp(int v, uint32_t n) { return v < 0 ? 0u : ((uint32_t)v >= n ? n - 1 : (uint32_t)v); }

// level 0: max over the clamped 3 x 3 neighbourhood; level k: max of the four level k-1 entries 2^(k-1) cells away diagonally
__global__ void prep_highmap_kernel(const float* __restrict__ heights, uint32_t W, uint32_t H, int k, float* __restrict__ highmap) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    float h;
    if (k == 0) {
        h = heights[(size_t)y * W + x];
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
                h = prep_max(h, heights[(size_t)prep_clamp((int)y + dy, H) * W + prep_clamp((int)x + dx, W)]);
    } else {
        const int off = 1 << (k - 1);
        auto at = [&](int xx, int yy) { return highmap[((size_t)prep_clamp(yy, H) * W + prep_clamp(xx, W)) * 16 + (k - 1)]; };
        const float a = at((int)x - off, (int)y - off), b = at((int)x + off, (int)y - off);
        const float c = at((int)x - off, (int)y + off), d = at((int)x + off, (int)y + off);
        h = prep_max(prep_max(a, b), prep_max(c, d));
    }
    highmap[((size_t)y * W + x) * 16 + k] = h;
}

// (I(x,y) - I(x+1,y), I(x,y) - I(x,y+1), 0) with wrap-around, I = plain mean of the channels
__global__ void prep_differentiate_kernel(const float* __restrict__ rgb, uint32_t W, uint32_t H, float* __restrict__ out) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    auto inten = [&](uint32_t xx, uint32_t yy) { const float* p = rgb + ((size_t)yy * W + xx) * 3; return (p[0] + p[1] + p[2]) / 3; };
    const float me = inten(x, y);
    float* o = out + ((size_t)y * W + x) * 3;
    o[0] = me - inten((x + 1) % W, y);
    o[1] = me - inten(x, (y + 1) % H);
    o[2] = 0.f;
}

// device buffers of one prep call, freed on every path out
struct PrepBufs {
    std::vector<void*> p;
    ~PrepBufs() { for (void* q : p) if (q) (void)hipFree(q); }
    template <class T> int get(T** out, size_t n) {
        void* q = nullptr;
        HIP_TRY(hipMalloc(&q, (n ? n : 1) * sizeof(T)));
        p.push_back(q); *out = (T*)q;
        return 0;
    }
};

}  // namespace

extern "C" int qd_prep_heightfield(uint32_t width, uint32_t height, const float* rgb, const float* gauss, int r,
                                   float* heights, float* maxheights, double* normals) {
    if (!rgb || !heights || !maxheights || !normals || width == 0 || height == 0) return set_err("qd_prep_heightfield: bad argument");
    if (r < 0 || r > 128 || (r > 0 && !gauss)) return set_err("qd_prep_heightfield: blur radius outside 0..128, or no weights");
    if (g_device < 0) return set_err("qd_prep_heightfield: qd_init has not been called");
    const size_t n = (size_t)width * height;
    PrepBufs B;
    float *d_rgb, *d_grey, *d_h, *d_mh, *d_g = nullptr; double* d_n;
    if (B.get(&d_rgb, n * 3) || B.get(&d_grey, n) || B.get(&d_h, n) || B.get(&d_mh, n) || B.get(&d_n, n * 3)) return -1;
    HIP_TRY(hipMemcpy(d_rgb, rgb, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(prep_grey_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_rgb, d_grey, n);
    const dim3 blk(64, 4), grd((width + 63) / 64, (height + 3) / 4);
    float* d_heights = d_grey;                                   // no blur: the heights are the grey values
    if (r > 0) {
        if (B.get(&d_g, (size_t)r * r)) return -1;
        HIP_TRY(hipMemcpy(d_g, gauss, (size_t)r * r * sizeof(float), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(prep_blur_kernel, grd, blk, 0, nullptr, d_grey, d_g, r, width, height, d_h);
        d_heights = d_h;
    }
    hipLaunchKernelGGL(prep_cells_kernel, grd, blk, 0, nullptr, d_heights, width, height, d_mh, d_n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(heights, d_heights, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(maxheights, d_mh, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(normals, d_n, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int qd_prep_highmap(uint32_t width, uint32_t height, const float* heights, int max_k, float* highmap) {
    if (!heights || !highmap || width == 0 || height == 0 || max_k < 0 || max_k > 16) return set_err("qd_prep_highmap: bad argument");
    if (g_device < 0) return set_err("qd_prep_highmap: qd_init has not been called");
    const size_t n = (size_t)width * height;
    PrepBufs B;
    float *d_h, *d_map;
    if (B.get(&d_h, n) || B.get(&d_map, n * 16)) return -1;
    HIP_TRY(hipMemcpy(d_h, heights, n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_map, 0, n * 16 * sizeof(float)));
    const dim3 blk(64, 4), grd((width + 63) / 64, (height + 3) / 4);
    for (int k = 0; k < (max_k > 1 ? max_k : 1); ++k)            // level 0 always exists, levels 1 .. max_k - 1 follow
        hipLaunchKernelGGL(prep_highmap_kernel, grd, blk, 0, nullptr, d_h, width, height, k, d_map);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(highmap, d_map, n * 16 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int qd_prep_differentiate(uint32_t width, uint32_t height, float* rgb) {
    if (!rgb || width == 0 || height == 0) return set_err("qd_prep_differentiate: bad argument");
    if (g_device < 0) return set_err("qd_prep_differentiate: qd_init has not been called");
    const size_t n = (size_t)width * height;
    PrepBufs B;
    float *d_in, *d_out;
    if (B.get(&d_in, n * 3) || B.get(&d_out, n * 3)) return -1;
    HIP_TRY(hipMemcpy(d_in, rgb, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    const dim3 blk(64, 4), grd((width + 63) / 64, (height + 3) / 4);
    hipLaunchKernelGGL(prep_differentiate_kernel, grd, blk, 0, nullptr, d_in, width, height, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rgb, d_out, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
