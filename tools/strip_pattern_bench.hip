// What does the memory system deliver for the access pattern of the strip edge kernels at 64 channels?  (measurement
// aid, not part of the library)
//
// Edge tensors [B, V, V, C, H] fp32 with H = 64: a row is 256 contiguous bytes and the rows j of a (b, i, c) unit lie
// C rows = 25,600 bytes apart (C = 100).  The forward kernel reads two tensors and writes two per 16-row tile; its
// workgroups (128 threads, four to six per CU) take units o, o + nslots, ... of an XCD's eighth of all units.  This
// program moves the same bytes with a plain kernel of the same shape in three unit orders:
//   rows       consecutive rows (the ceiling for this kernel shape)
//   i-fastest  the kernel's order: unit o = (b, c, i), i fastest - workgroups side by side work on different rows i of one
//              (b, c), 5 MB apart
//   c-fastest  unit o = (b, i, c), c fastest - workgroups side by side work on neighbouring c of one (b, i): the 256-byte
//              rows they touch at about the same time are adjacent in memory
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o graph-convnet-uelb_amd/lib/strip_pattern_bench.bin tools/strip_pattern_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int H = 64, C = 100, V = 208, B = 4, NJT = V / 16;

template <int ORDER, int NW>
__global__ __launch_bounds__(128) void k_pat(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o1, float* __restrict__ o2) {
    const int t = threadIdx.x, arow = t >> 4, g = t & 15;           // rows arow and arow + 8, float4 column group g
    const int units = B * V * C;
    const int nslots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int per_xcd = (units + 7) >> 3;
    const int o_hi = min(units, (xcd + 1) * per_xcd);
    for (int o = xcd * per_xcd + slot; o < o_hi; o += nslots) {
        int bi, c;
        if (ORDER == 1) { const int bc = o / V, i = o - bc * V, bb = bc / C; c = bc - bb * C; bi = bb * V + i; }
        else { bi = o / C; c = o - bi * C; }
        f32x4 va[2], vb[2];
        auto row_of = [&](int jt, int r) -> size_t {
            if (ORDER == 0) return ((size_t)o * NJT + jt) * 16 + r;
            return ((size_t)bi * V + jt * 16 + r) * C + c;
        };
        auto load = [&](int jt) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const size_t off = row_of(jt, arow + 8 * k) * H + 4 * g;
                va[k] = *reinterpret_cast<const f32x4*>(a + off);
                vb[k] = *reinterpret_cast<const f32x4*>(b + off);
            }
        };
        load(0);
        for (int jt = 0; jt < NJT; ++jt) {
            f32x4 ca[2] = {va[0], va[1]}, cb[2] = {vb[0], vb[1]};
            if (jt + 1 < NJT) load(jt + 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const size_t off = row_of(jt, arow + 8 * k) * H + 4 * g;
                *reinterpret_cast<f32x4*>(o1 + off) = ca[k] + cb[k];
                if (NW == 2) *reinterpret_cast<f32x4*>(o2 + off) = ca[k] * cb[k];
            }
        }
    }
}

template <int ORDER, int NW>
static void run(const char* name, const float* a, const float* b, float* o1, float* o2, hipEvent_t e0, hipEvent_t e1) {
    for (int grid : {1024, 1536, 2048}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            k_pat<ORDER, NW><<<grid, 128>>>(a, b, o1, o2);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) best = std::min(best, ms);
        }
        const double bytes = (2.0 + NW) * B * V * (double)V * C * H * 4;
        printf("%-10s 2 read + %d written, %4d workgroups of 128 threads: %7.3f ms  %5.2f TB/s\n", name, NW, grid, best, bytes / best / 1e9);
    }
}

int main() {
    const size_t rows = (size_t)B * V * V * C, bytes = rows * H * 4;
    float *a, *b, *o1, *o2;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o1, bytes)); CK(hipMalloc(&o2, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%zu rows of %d bytes per tensor (%.2f GB)\n", rows, H * 4, bytes / 1e9);
    run<0, 2>("rows", a, b, o1, o2, e0, e1);
    run<1, 2>("i-fastest", a, b, o1, o2, e0, e1);
    run<2, 2>("c-fastest", a, b, o1, o2, e0, e1);
    run<0, 1>("rows", a, b, o1, o2, e0, e1);
    run<1, 1>("i-fastest", a, b, o1, o2, e0, e1);
    run<2, 1>("c-fastest", a, b, o1, o2, e0, e1);
    CK(hipDeviceSynchronize());
    return 0;
}
