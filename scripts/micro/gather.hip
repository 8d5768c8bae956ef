// Micro-benchmark: dependent random gathers, the access pattern of the KD walk.
// Each lane chases indices through a large array; per step it loads K x 16 bytes from one 128-byte line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int K, int SPREAD>
__global__ __launch_bounds__(64, 4) void chase(const u32x4* __restrict__ arr, unsigned n_lines, int steps, unsigned* out) {
    unsigned idx = (blockIdx.x * 64u + threadIdx.x) * 2654435761u % n_lines;
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        const u32x4* p = arr + (size_t)idx * 8;      // 128-byte line = 8 x 16 bytes
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = p[k * SPREAD];
        unsigned h = idx;
#pragma unroll
        for (int k = 0; k < K; ++k) h = h * 31u + v[k].x + v[k].w;
        acc += h;
        idx = (h * 2654435761u) % n_lines;
    }
    out[blockIdx.x * 64u + threadIdx.x] = acc;
}

template <int K, int SPREAD>
void run(const char* name, const u32x4* arr, unsigned n_lines, unsigned* out, int waves_per_cu) {
    const int steps = 2000, grid = 256 * waves_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((chase<K, SPREAD>), dim3(grid), dim3(64), 0, 0, arr, n_lines, 200, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((chase<K, SPREAD>), dim3(grid), dim3(64), 0, 0, arr, n_lines, steps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double lane_steps = (double)grid * 64 * steps;
    printf("%-28s lines %9u waves/CU %2d: %7.2f ms  %6.1f ns/wave-step  %6.2f G lane-steps/s\n", name, n_lines, waves_per_cu, ms,
           ms * 1e6 / steps, lane_steps / ms / 1e6);
}

int main(int argc, char** argv) {
    const bool calibrate = argc > 1;   // one case only: FETCH_SIZE calibration for 16-byte random gathers (lines = 4096 waves x 64 x 2200)
    const size_t max_lines = (size_t)1 << 24;        // 2 GiB
    u32x4* arr; unsigned* out;
    hipMalloc(&arr, max_lines * 128); hipMalloc(&out, 256 * 16 * 64 * 4);
    std::vector<unsigned> h(max_lines * 32);
    unsigned x = 12345u;
    for (auto& w : h) { x = x * 1664525u + 1013904223u; w = x; }
    hipMemcpy(arr, h.data(), max_lines * 128, hipMemcpyHostToDevice);
    if (calibrate) { run<1, 1>("1 x 16B", arr, 1u << 24, out, 16); return 0; }
    for (unsigned lines : {1u << 14, 1u << 19, 1u << 24}) {    // 2 MiB (L2), 64 MiB (MALL), 2 GiB (HBM)
        for (int w : {8, 16}) {
            run<1, 1>("1 x 16B", arr, lines, out, w);
            run<2, 1>("2 x 16B adjacent", arr, lines, out, w);
            run<3, 1>("3 x 16B adjacent", arr, lines, out, w);
            run<5, 1>("5 x 16B adjacent", arr, lines, out, w);
            run<2, 4>("2 x 16B, 64B apart", arr, lines, out, w);
        }
    }
    return 0;
}
