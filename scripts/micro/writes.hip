// Micro-benchmark: what the L2 does with repeated writes.  Each kernel makes REPS rounds of stores to a footprint that
// fits the L2 many times over; the PMC counters (TCC_WRITE_sum at the L2's input, TCC_EA0_WRREQ_sum / WRITE_SIZE at its
// output) tell whether the stores are combined in the L2 (write-back) or passed on one by one.
//   scratch_rw   a per-lane array with a run-time index (lives in scratch memory)
//   global_rw    the same pattern on lane-interleaved global memory
//   atomic_f32   float atomicAdd on a small global array (agent scope)
// usage: writes <kernel 0|1|2>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int REPS = 256, WORDS = 64;

__global__ __launch_bounds__(64) void scratch_rw(const int* __restrict__ idx, int* out) {
    int a[WORDS];
    for (int i = 0; i < WORDS; ++i) a[i] = i;
    int acc = 0;
    for (int r = 0; r < REPS; ++r) {
        const int k = idx[r] & (WORDS - 1);       // uniform, unknown at compile time
        a[k] = acc + r;
        acc += a[(k * 5 + 3) & (WORDS - 1)];
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(64) void global_rw(const int* __restrict__ idx, int* buf, int* out) {
    int* a = buf + (size_t)blockIdx.x * 64 * WORDS + threadIdx.x;       // word w of lane l at a[w * 64]
    for (int i = 0; i < WORDS; ++i) a[i * 64] = i;
    int acc = 0;
    for (int r = 0; r < REPS; ++r) {
        const int k = idx[r] & (WORDS - 1);
        a[k * 64] = acc + r;
        acc += a[((k * 5 + 3) & (WORDS - 1)) * 64];
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(64) void atomic_f32(float* px) {
    for (int r = 0; r < REPS; ++r) atomicAdd(&px[(blockIdx.x * 64 + threadIdx.x + r * 64) & 65535], 1.0f);
}

int main(int argc, char** argv) {
    const int which = argc > 1 ? atoi(argv[1]) : 0, grid = 1024;
    int *idx, *buf, *out; float* px;
    int h[REPS]; for (int r = 0; r < REPS; ++r) h[r] = (r * 37 + 11) & 63;
    hipMalloc(&idx, sizeof h); hipMemcpy(idx, h, sizeof h, hipMemcpyHostToDevice);
    hipMalloc(&buf, (size_t)grid * 64 * WORDS * 4); hipMalloc(&out, grid * 64 * 4); hipMalloc(&px, 65536 * 4); hipMemset(px, 0, 65536 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    if (which == 0) hipLaunchKernelGGL(scratch_rw, dim3(grid), dim3(64), 0, 0, idx, out);
    if (which == 1) hipLaunchKernelGGL(global_rw, dim3(grid), dim3(64), 0, 0, idx, buf, out);
    if (which == 2) hipLaunchKernelGGL(atomic_f32, dim3(grid), dim3(64), 0, 0, px);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("kernel %d: %.3f ms; store instructions per wave %d (+%d to fill), waves %d, footprint %.1f MiB -> %0.f 64-byte store requests if nothing is combined\n",
           which, ms, REPS, WORDS, grid, grid * 64.0 * WORDS * 4 / 1048576, (double)grid * (REPS + WORDS) * 4);
    return hipDeviceSynchronize() != hipSuccess;
}
