// What does this box's memory system deliver to a plain streaming kernel?  (measurement aid, not part of the library)
// Copy (1 read + 1 write), read-only and write-only sweeps over 4 GiB with 16 bytes per lane, plain and non-temporal
// accesses, 1 to 8 independent 16-byte accesses in flight per lane, 256 to 1024 threads per workgroup, 1 to 16
// workgroups of 256 threads' worth per CU.  Prints the best and a few representative lines.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o graph-convnet-uelb_amd/lib/copy_peak_bench.bin tools/copy_peak_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0 copy, 1 read-only (sum), 2 write-only.  NT: non-temporal accesses.  U: independent accesses per lane per trip.
template <int MODE, bool NT, int U>
__global__ void k_stream(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4, float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 v[U];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = f32x4{1.f, 2.f, 3.f, (float)u};
        }
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

struct Line { double tbs; std::string what; };

template <int MODE, bool NT, int U>
static void sweep(std::vector<Line>& out, const f32x4* src, f32x4* dst, size_t n4, float* sink, hipEvent_t e0, hipEvent_t e1) {
    for (int block : {256, 512, 1024})
        for (int per_cu : {1, 2, 4, 8, 16}) {
            const int grid = 256 * per_cu * 256 / block;
            if (grid < 256) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                k_stream<MODE, NT, U><<<grid, block>>>(src, dst, n4, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0) best = std::min(best, ms);
            }
            const double bytes = (MODE == 0 ? 2.0 : 1.0) * n4 * 16;
            char buf[200];
            snprintf(buf, sizeof buf, "%s %s  %d in flight per lane, %4d threads x %5d workgroups: %7.3f ms", MODE == 0 ? "copy " : (MODE == 1 ? "read " : "write"),
                     NT ? "non-temporal" : "plain       ", U, block, grid, best);
            out.push_back({bytes / best / 1e9, buf});
        }
}

int main() {
    const size_t bytes = (size_t)4 << 30, n4 = bytes / 16;
    f32x4 *src, *dst;
    float* sink;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Line> copy, rd, wr;
    sweep<0, false, 1>(copy, src, dst, n4, sink, e0, e1); sweep<0, false, 4>(copy, src, dst, n4, sink, e0, e1); sweep<0, false, 8>(copy, src, dst, n4, sink, e0, e1);
    sweep<0, true, 1>(copy, src, dst, n4, sink, e0, e1); sweep<0, true, 4>(copy, src, dst, n4, sink, e0, e1); sweep<0, true, 8>(copy, src, dst, n4, sink, e0, e1);
    sweep<1, false, 4>(rd, src, dst, n4, sink, e0, e1); sweep<1, true, 4>(rd, src, dst, n4, sink, e0, e1); sweep<1, false, 8>(rd, src, dst, n4, sink, e0, e1);
    sweep<2, false, 4>(wr, src, dst, n4, sink, e0, e1); sweep<2, true, 4>(wr, src, dst, n4, sink, e0, e1);
    for (auto* v : {&copy, &rd, &wr}) {
        std::sort(v->begin(), v->end(), [](const Line& a, const Line& b) { return a.tbs > b.tbs; });
        printf("best five of %zu settings, then the median and the worst:\n", v->size());
        for (size_t k = 0; k < v->size(); ++k)
            if (k < 5 || k == v->size() / 2 || k + 1 == v->size()) printf("  %6.2f TB/s  %s\n", (*v)[k].tbs, (*v)[k].what.c_str());
    }
    return 0;
}
