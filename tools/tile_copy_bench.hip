// Copy bandwidth of the edge-tensor access pattern: tiles of 16 rows x 256 bytes whose rows lie C * 256 bytes apart
// (rows j of one (b, i, c) in a [B][V][V][C][H] tensor, H = 64) against tiles of 4096 contiguous bytes (what a
// [B][C][V][V][H] layout would give), at several workgroup counts.  Build: hipcc --offload-arch=gfx950 -O3 -o tile_copy_bench tile_copy_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// tensor [BV][V][C][64]; unit = (bi, c), tiles of 16 rows j.  mode 0: strided rows (as stored), mode 1: contiguous tiles.
// order 0: units (bi, c) c fastest; order 1: (c, bi) bi fastest.  NR tensors read, NW written per tile.
template <int NR, int NW>
__global__ __launch_bounds__(256) void k_tiles(const float* __restrict__ src, float* __restrict__ dst, size_t tensor_floats,
                                               long long ntiles, int V, int C, int mode, int order, long long BV) {
    const int t = threadIdx.x, row = t >> 4, g = t & 15;
    const int njt = V / 16;
    const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    auto off_of = [&](long long tl) -> size_t {
        if (mode == 1) return (size_t)tl * 1024 + (size_t)row * 64 + g * 4;
        long long u = tl / njt; const int jt = (int)(tl - u * njt);
        long long bi; int c;
        if (order == 0) { bi = u / C; c = (int)(u - bi * C); } else { c = (int)(u / BV); bi = u - (long long)c * BV; }
        return (((size_t)bi * V + jt * 16 + row) * C + c) * 64 + g * 4;
    };
    if (t0 >= t1) return;
    f32x4 v[NR];
    size_t off = off_of(t0);
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + k * tensor_floats + off);
    for (long long tl = t0; tl < t1; ++tl) {
        const size_t noff = off_of(tl + 1 < t1 ? tl + 1 : tl);
        f32x4 n[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) n[k] = *reinterpret_cast<const f32x4*>(src + k * tensor_floats + noff);
        f32x4 s = v[0];
#pragma unroll
        for (int k = 1; k < NR; ++k) s += v[k];
#pragma unroll
        for (int k = 0; k < NW; ++k) *reinterpret_cast<f32x4*>(dst + k * tensor_floats + off) = s + (float)k;
#pragma unroll
        for (int k = 0; k < NR; ++k) v[k] = n[k];
        off = noff;
    }
}

int main(int argc, char** argv) {
    const int B = 4, V = 192, C = 100;          // V a multiple of 16 keeps the two modes on the same bytes
    const long long BV = (long long)B * V;
    const size_t tensor_floats = (size_t)BV * V * C * 64;
    const long long ntiles = BV * C * (V / 16);
    float *src, *dst;
    CK(hipMalloc(&src, 4 * tensor_floats * sizeof(float)));
    CK(hipMalloc(&dst, 2 * tensor_floats * sizeof(float)));
    CK(hipMemset(src, 1, 4 * tensor_floats * sizeof(float)));
    CK(hipMemset(dst, 0, 2 * tensor_floats * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("tensor %.2f GB, %lld tiles\n", tensor_floats * 4 / 1e9, ntiles);
    const int grids[] = {512, 1024, 2048, 4096, 16384};
    for (int cfg = 0; cfg < 3; ++cfg) {          // (reads, writes): (1,1) copy, (2,2) forward-like, (4,1) backward-like
        const int nr = cfg == 0 ? 1 : (cfg == 1 ? 2 : 4), nw = cfg == 0 ? 1 : (cfg == 1 ? 2 : 1);
        for (int mode = 0; mode < 2; ++mode)
            for (int order = 0; order < (mode == 0 ? 2 : 1); ++order)
                for (int grid : grids) {
                    float best = 1e30f;
                    for (int rep = 0; rep < 4; ++rep) {
                        CK(hipEventRecord(e0));
                        if (cfg == 0) k_tiles<1, 1><<<grid, 256>>>(src, dst, tensor_floats, ntiles, V, C, mode, order, BV);
                        else if (cfg == 1) k_tiles<2, 2><<<grid, 256>>>(src, dst, tensor_floats, ntiles, V, C, mode, order, BV);
                        else k_tiles<4, 1><<<grid, 256>>>(src, dst, tensor_floats, ntiles, V, C, mode, order, BV);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep > 0 && ms < best) best = ms;
                    }
                    const double bytes = (double)(nr + nw) * tensor_floats * 4;
                    printf("reads %d writes %d  %s order %s  grid %5d: %7.3f ms  %6.2f TB/s\n", nr, nw, mode ? "contiguous" : "strided   ",
                           order ? "(c,bi)" : "(bi,c)", grid, best, bytes / best / 1e9);
                    fflush(stdout);
                }
    }
    return 0;
}
