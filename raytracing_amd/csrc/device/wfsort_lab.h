// EXPERIMENT, compiled in and OFF unless the environment names it (profiles/r03_experiments.md, section 8): does the walk
// get cheaper when the rays of a generation that share a wave are chosen for coherence?
//
//   QD_WF_SORT=1   generation g + 1 is ordered by (shadow ray?, octant of the direction) before its walk -- a STABLE radix
//                  sort, so rays of one key stay in the order they were appended in, which is the order of the pixels
//                  (neighbouring lanes start from neighbouring hit points);
//   QD_WF_SORT=2   ... by (shadow ray?, octant, 3 bits of every direction component relative to the largest);
//   QD_WF_SORT=3   no sort: the same synchronisation and timing, the figure to compare with.
//
// The host waits for every generation (it reads the generation's size back), so frame times mean nothing in this mode.  The
// figures are summed device times between HIP events: the walk + shade launches of every generation, and the sort of every
// generation; both are printed per frame on stderr when the scene is destroyed.  Not a product path: wavefront form only,
// one stream at a time.
#pragma once

struct WfSortLab {
    int mode = -1;
    uint32_t* keys[2] = {nullptr, nullptr};
    uint32_t* idx[2] = {nullptr, nullptr};
    uint64_t* tmp = nullptr;
    void* cub = nullptr;
    size_t cub_bytes = 0, n_cap = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double ms_trace = 0., ms_sort = 0.;
    uint64_t frames = 0, rays_sorted = 0, gens_sorted = 0;
};
static WfSortLab g_lab;

static int wf_sort_mode() {
    if (g_lab.mode < 0) {
        const char* e = getenv("QD_WF_SORT");
        g_lab.mode = e ? atoi(e) : 0;
        if (g_lab.mode < 0 || g_lab.mode > 3) g_lab.mode = 0;
    }
    return g_lab.mode;
}

__global__ __launch_bounds__(256) void wf_sort_keys_kernel(const uint64_t* Q, size_t cap, uint32_t n, int mode, uint32_t* keys, uint32_t* idx) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const double dx = __longlong_as_double((long long)Q[3 * cap + i]), dy = __longlong_as_double((long long)Q[4 * cap + i]),
                 dz = __longlong_as_double((long long)Q[5 * cap + i]);
    const uint32_t shadow = ((uint32_t)(Q[7 * cap + i] >> 32) & kFlagShadow) ? 1u : 0u;
    uint32_t key = (shadow << 3) | (dx > 0 ? 1u : 0u) | (dy > 0 ? 2u : 0u) | (dz > 0 ? 4u : 0u);
    if (mode == 2) {
        const double m = fmax(fmax(fabs(dx), fabs(dy)), fabs(dz));
        const double s = m > 0. ? 7.999 / m : 0.;
        const uint32_t qx = min(7u, (uint32_t)(fabs(dx) * s)), qy = min(7u, (uint32_t)(fabs(dy) * s)), qz = min(7u, (uint32_t)(fabs(dz) * s));
        key = (key << 9) | (qx << 6) | (qy << 3) | qz;
    }
    keys[i] = key;
    idx[i] = i;
}

// dst is compact (word w of entry i at [w * n + i]); src is the queue (stride cap)
__global__ __launch_bounds__(256) void wf_sort_permute_kernel(const uint64_t* src, size_t cap, const uint32_t* idx, uint32_t n, uint64_t* dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    if (j >= n) return;                                       // (cannot happen: idx is a permutation of 0 .. n - 1)
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)kEntryWords; ++w) dst[(size_t)w * n + i] = src[(size_t)w * cap + j];
}

static void wf_lab_free() {
    for (int k = 0; k < 2; ++k) {
        if (g_lab.keys[k]) (void)hipFree(g_lab.keys[k]);
        if (g_lab.idx[k]) (void)hipFree(g_lab.idx[k]);
        g_lab.keys[k] = g_lab.idx[k] = nullptr;
    }
    if (g_lab.tmp) (void)hipFree(g_lab.tmp);
    if (g_lab.cub) (void)hipFree(g_lab.cub);
    g_lab.tmp = nullptr; g_lab.cub = nullptr; g_lab.cub_bytes = 0; g_lab.n_cap = 0;
}

// The stream has been drained by the caller.  queue: generation `gen_next`'s ray queue (stride cap), count at *count_dev.
static int wf_lab_sort(uint64_t* queue, size_t cap, const uint32_t* count_dev, uint32_t limit, hipStream_t stream) {
    uint32_t n = 0;
    if (hipMemcpy(&n, count_dev, sizeof n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    n = std::min<uint32_t>(n, limit);
    n = (uint32_t)std::min<size_t>(n, cap);
    if (n < 2u || g_lab.mode == 3) return 0;
    if (n > g_lab.n_cap) {
        wf_lab_free();
        const size_t want = (size_t)n + n / 4 + 1024;
        for (int k = 0; k < 2; ++k) {
            if (hipMalloc((void**)&g_lab.keys[k], want * sizeof(uint32_t)) != hipSuccess) return -1;
            if (hipMalloc((void**)&g_lab.idx[k], want * sizeof(uint32_t)) != hipSuccess) return -1;
        }
        if (hipMalloc((void**)&g_lab.tmp, want * (size_t)kEntryWords * sizeof(uint64_t)) != hipSuccess) return -1;
        g_lab.n_cap = want;
    }
    const int bits = g_lab.mode == 2 ? 13 : 4;
    size_t need = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, need, g_lab.keys[0], g_lab.keys[1], g_lab.idx[0], g_lab.idx[1], (int)n, 0, bits, stream) != hipSuccess) return -1;
    if (need > g_lab.cub_bytes) {
        if (g_lab.cub) (void)hipFree(g_lab.cub);
        g_lab.cub = nullptr; g_lab.cub_bytes = 0;
        if (hipMalloc(&g_lab.cub, need + 256) != hipSuccess) return -1;
        g_lab.cub_bytes = need + 256;
    }
    const unsigned blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(wf_sort_keys_kernel, dim3(blocks), dim3(256), 0, stream, (const uint64_t*)queue, cap, n, g_lab.mode, g_lab.keys[0], g_lab.idx[0]);
    if (hipGetLastError() != hipSuccess) return -1;
    size_t bytes = g_lab.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(g_lab.cub, bytes, g_lab.keys[0], g_lab.keys[1], g_lab.idx[0], g_lab.idx[1], (int)n, 0, bits, stream) != hipSuccess) return -1;
    hipLaunchKernelGGL(wf_sort_permute_kernel, dim3(blocks), dim3(256), 0, stream, (const uint64_t*)queue, cap, (const uint32_t*)g_lab.idx[1], n, g_lab.tmp);
    if (hipGetLastError() != hipSuccess) return -1;
    for (size_t w = 0; w < (size_t)kEntryWords; ++w)
        if (hipMemcpyAsync(queue + w * cap, g_lab.tmp + w * n, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream) != hipSuccess) return -1;
    g_lab.rays_sorted += n; g_lab.gens_sorted++;
    return 0;
}

static void wf_lab_report() {
    if (g_lab.mode <= 0 || g_lab.frames == 0) return;
    fprintf(stderr, "QD_WF_SORT=%d: %llu frames; per frame: walk + shade launches %.3f ms, sort %.3f ms (%.2f M rays in %.1f generations)\n",
            g_lab.mode, (unsigned long long)g_lab.frames, g_lab.ms_trace / g_lab.frames, g_lab.ms_sort / g_lab.frames,
            (double)g_lab.rays_sorted / g_lab.frames / 1e6, (double)g_lab.gens_sorted / g_lab.frames);
    g_lab.frames = 0; g_lab.ms_trace = g_lab.ms_sort = 0.; g_lab.rays_sorted = g_lab.gens_sorted = 0;
    wf_lab_free();
}
