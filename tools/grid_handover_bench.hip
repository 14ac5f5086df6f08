// How long does one grid-wide hand-over take inside a persistent kernel?  (measurement aid, not part of the library)
//
// A persistent forward pass over all layers would keep its edge slabs in LDS and cross the grid twice per layer: every
// workgroup publishes a partial row of BatchNorm sums (device-scope stores), announces itself on an arrival counter and
// waits until all have arrived, then reads the partial rows of all workgroups.  This program measures that hand-over
// alone with the grid the edge kernels use (512 workgroups of 256 threads, two per CU), in three forms:
//   A  counter only: atomic arrival + spin on the counter (the floor of any grid barrier)
//   B  counter + every workgroup publishes a row of 2 H floats (sc1 stores) before arriving
//   C  as B, and after the barrier every workgroup reads all 512 rows back (what a BatchNorm over the batch needs),
//      one column per thread, batched 8 loads deep
//   D  counter only, two levels: groups of 16 workgroups arrive on a counter of their own (32 addresses instead of one),
//      the last of a group arrives on the top counter, everyone polls the top counter
// Printed: microseconds per hand-over (total kernel time / rounds, launch excluded by differencing two round counts).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o graph-convnet-uelb_amd/lib/grid_handover_bench.bin tools/grid_handover_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int H2 = 128;      // 2 H floats per partial row at H = 64

__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_handover(int rounds, unsigned* __restrict__ counter, float* __restrict__ rows, float* __restrict__ out) {
    const int t = threadIdx.x, nwg = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (MODE >= 1 && t < H2) st_dev(rows + (size_t)blockIdx.x * H2 + t, (float)(r + t));
        __syncthreads();
        if (t == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)nwg * (unsigned)(r + 1);
            // bounded: if the grid were not resident as a whole the wait ends anyway (and the timing is meaningless)
            for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
                __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (MODE >= 2 && t < H2) {
            float s = 0.f;
            for (int w0 = 0; w0 < nwg; w0 += 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = ld_dev(rows + (size_t)(w0 + k) * H2 + t);
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            }
            acc += s;
        }
    }
    if (acc == 123.456f) out[t] = acc;
}

__global__ __launch_bounds__(256, 2) void k_handover_tree(int rounds, unsigned* __restrict__ top, unsigned* __restrict__ leaf) {
    const int t = threadIdx.x, ngroups = (gridDim.x + 15) / 16, g = blockIdx.x / 16;
    const int gsize = min(16, (int)gridDim.x - 16 * g);
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();
        if (t == 0) {
            const unsigned prev = __hip_atomic_fetch_add(leaf + 32 * g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1 == (unsigned)gsize * (unsigned)(r + 1)) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)ngroups * (unsigned)(r + 1);
            for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(top, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
                __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}
static double run_tree(int grid, int rounds, unsigned* top, unsigned* leaf) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(top, 0, 4); (void)hipMemset(leaf, 0, 4 * 32 * 64);
    k_handover_tree<<<grid, 256>>>(4, top, leaf);
    (void)hipDeviceSynchronize();
    (void)hipMemset(top, 0, 4); (void)hipMemset(leaf, 0, 4 * 32 * 64);
    (void)hipEventRecord(e0);
    k_handover_tree<<<grid, 256>>>(rounds, top, leaf);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
}

template <int MODE>
static double run(int grid, int rounds, unsigned* counter, float* rows, float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(counter, 0, 4);
    k_handover<MODE><<<grid, 256>>>(4, counter, rows, out);
    (void)hipDeviceSynchronize();
    (void)hipMemset(counter, 0, 4);
    (void)hipEventRecord(e0);
    k_handover<MODE><<<grid, 256>>>(rounds, counter, rows, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
}

int main() {
    unsigned* counter;
    float *rows, *out;
    (void)hipMalloc(&counter, 4);
    (void)hipMalloc(&rows, sizeof(float) * 1024 * H2);
    (void)hipMalloc(&out, 4096);
    unsigned* leaf;
    (void)hipMalloc(&leaf, 4 * 32 * 64);
    for (int grid : {256, 512}) {
        const double a1 = run<0>(grid, 50, counter, rows, out), a2 = run<0>(grid, 250, counter, rows, out);
        const double b1 = run<1>(grid, 50, counter, rows, out), b2 = run<1>(grid, 250, counter, rows, out);
        const double c1 = run<2>(grid, 50, counter, rows, out), c2 = run<2>(grid, 250, counter, rows, out);
        const double d1 = run_tree(grid, 50, counter, leaf), d2 = run_tree(grid, 250, counter, leaf);
        printf("grid %d x 256 threads: hand-over  A counter only %.2f us   B + publish a row %.2f us   C + read all rows back %.2f us   D two-level counter %.2f us\n", grid,
               (a2 - a1) / 200.0, (b2 - b1) / 200.0, (c2 - c1) / 200.0, (d2 - d1) / 200.0);
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("error: %s\n", hipGetErrorString(err));
    return 0;
}
