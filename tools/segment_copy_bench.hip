// What does the memory system deliver for the access pattern of the wide edge kernels?  (measurement aid, not part of
// the library)
//
// The edge tensors are [B, V, V, C, H] fp32 with H = 300: a row is 1,200 contiguous bytes, and the rows j of one
// (b, i, c) unit lie C rows = 60,000 bytes apart.  A workgroup of the wide kernels walks the rows of a unit 64 at a time
// (16 rows per sub-step, 32 threads x 3 float4 per row) and takes the units of a contiguous range one after the other,
// c fastest.  This program moves the same bytes (2 tensors read, 1 or 2 written, 1.23 GB each) in three row orders:
//   contiguous   the workgroup's rows are consecutive in memory (the ceiling for 1,200-byte rows)
//   unit-major   the kernels' order: 128 rows of stride C per unit, units c fastest
//   c-blocked    a step takes 16 rows j x 4 neighbouring c (4,800 contiguous bytes per j)
// with 1, 2 or 4 sub-steps of 16 rows requested before the first is used, one or two 512-thread workgroups per CU.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o graph-convnet-uelb_amd/lib/segment_copy_bench.bin tools/segment_copy_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int H = 300, G = H / 4, C = 50, VJ = 128;      // 75 float4 per row; 128 rows j per unit

// row index of local row r (0 .. 63) of step n of a workgroup that owns steps [s0, s1)
template <int ORDER>
__device__ __forceinline__ size_t row_of(size_t step, int r) {
    if (ORDER == 0) return step * 64 + r;
    if (ORDER == 1) {           // unit = step / 2 -> (bi, c), c fastest; rows j = 64 (step & 1) + r
        const size_t unit = step >> 1, bi = unit / C, c = unit % C;
        return (bi * VJ + 64 * (step & 1) + r) * C + c;
    }
    // c-blocked: a (bi, c block of 4) pair holds 128 j x 4 c = 512 rows = 8 steps of 16 j x 4 c; C padded to 52 is avoided by
    // letting the last block of a bi take c = 46 .. 49 (overlap with the block before it: a little extra traffic, not less)
    const size_t blk = step >> 3, bi = blk / 13, cb = blk % 13;
    const int c0 = (int)std::min<size_t>(4 * cb, C - 4);
    const int j = 16 * (int)(step & 7) + (r >> 2);
    return (bi * VJ + j) * C + c0 + (r & 3);
}

template <int ORDER, int DEPTH, int NW>
__global__ __launch_bounds__(512) void k_seg(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o1, float* __restrict__ o2, size_t nsteps) {
    const int t = threadIdx.x, arow = t >> 5, g0 = t & 31;
    const size_t s0 = nsteps * blockIdx.x / gridDim.x, s1 = nsteps * (blockIdx.x + 1) / gridDim.x;
    for (size_t st = s0; st < s1; ++st) {
#pragma unroll
        for (int q0 = 0; q0 < 4; q0 += DEPTH) {
            f32x4 va[DEPTH][3], vb[DEPTH][3];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const size_t row = row_of<ORDER>(st, 16 * (q0 + d) + arow);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int g = g0 + 32 * k;
                    const size_t off = g < G ? row * H + 4 * g : 0;
                    va[d][k] = *reinterpret_cast<const f32x4*>(a + off);
                    vb[d][k] = *reinterpret_cast<const f32x4*>(b + off);
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const size_t row = row_of<ORDER>(st, 16 * (q0 + d) + arow);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int g = g0 + 32 * k;
                    if (g < G) {
                        *reinterpret_cast<f32x4*>(o1 + row * H + 4 * g) = va[d][k] + vb[d][k];
                        if (NW == 2) *reinterpret_cast<f32x4*>(o2 + row * H + 4 * g) = va[d][k] * vb[d][k];
                    }
                }
            }
        }
    }
}

template <int ORDER, int DEPTH, int NW>
static void run(const char* name, const float* a, const float* b, float* o1, float* o2, size_t nsteps, hipEvent_t e0, hipEvent_t e1) {
    for (int per_cu : {1, 2}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            k_seg<ORDER, DEPTH, NW><<<256 * per_cu, 512>>>(a, b, o1, o2, nsteps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) best = std::min(best, ms);
        }
        const double bytes = (2.0 + NW) * nsteps * 64 * H * 4;
        printf("%-11s %d sub-steps in flight, 2 read + %d written, %d workgroup(s) per CU: %7.3f ms  %5.2f TB/s\n", name, DEPTH, NW, per_cu, best, bytes / best / 1e9);
    }
}

int main() {
    const size_t nbi = 160, rows = nbi * VJ * C;
    size_t nsteps = rows / 64;          // 1,024,000 rows, 16,000 steps
    const size_t bytes = rows * H * 4;
    float *a, *b, *o1, *o2;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o1, bytes)); CK(hipMalloc(&o2, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%zu rows of %d bytes per tensor (%.2f GB)\n", rows, H * 4, bytes / 1e9);
#define ALL(ORDER, name) \
    run<ORDER, 1, 1>(name, a, b, o1, o2, nsteps, e0, e1); run<ORDER, 2, 1>(name, a, b, o1, o2, nsteps, e0, e1); run<ORDER, 4, 1>(name, a, b, o1, o2, nsteps, e0, e1); \
    run<ORDER, 1, 2>(name, a, b, o1, o2, nsteps, e0, e1); run<ORDER, 4, 2>(name, a, b, o1, o2, nsteps, e0, e1);
    ALL(0, "contiguous") ALL(1, "unit-major")
    nsteps = nbi * 13 * 8;      // 13 blocks of 4 c per (b, i), 8 steps each
    ALL(2, "c-blocked")
    // how fast can a single CU go when the others leave the memory system alone?  (fewer workgroups than CUs)
    nsteps = rows / 64 / 8;
    for (int grid : {16, 32, 64, 128, 256}) {
        for (int depth : {1, 4}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (depth == 1) k_seg<1, 1, 1><<<grid, 512>>>(a, b, o1, o2, nsteps * grid / 256);
                else k_seg<1, 4, 1><<<grid, 512>>>(a, b, o1, o2, nsteps * grid / 256);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0) best = std::min(best, ms);
            }
            const double bytes = 3.0 * (nsteps * grid / 256) * 64 * H * 4;
            printf("unit-major, %3d workgroups (one per CU), %d sub-steps in flight: %7.3f ms  %5.2f TB/s  %6.1f GB/s per workgroup\n", grid, depth, best, bytes / best / 1e9, bytes / best / 1e6 / grid);
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
