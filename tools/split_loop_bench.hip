// Microbenchmark of a split-bf16 product loop for the wide edge kernels (measurement aid, not part of the library).
//
// An fp32 value x is cut into three bf16 pieces x = x1 + x2 + x3 (each piece the round-to-nearest bf16 of what the
// pieces before it left over: the cut is exact, 8 + 8 + 8 significant bits).  A product a * w is then
//     a1 w1 + (a1 w2 + a2 w1) + (a2 w2 + a1 w3 + a3 w1)   + terms below 2^-26 |a w| that are dropped,
// six bf16 MFMA products accumulated in fp32.  v_mfma_f32_16x16x32_bf16 does 16 x 16 x 32 multiply-adds in about 16
// cycles where v_mfma_f32_16x16x4_f32 needs 8 x 32 cycles for the same block: 96 against 256 cycles per block.
//
// What is measured: ROWS = 16 NS rows of activations in LDS as three bf16 planes (row stride 640 B, 16-byte chunks
// swizzled so that the operand reads are conflict-free), against all column blocks of an H x H weight matrix whose
// three planes are streamed from L2 in fragment order, as k_edge_fwd_w's product phase would run it; no staging phase
// and no epilogue.  Printed: time, fp32-equivalent TFLOP/s (2 ROWS H^2 per step), and the largest error of the result
// against a float64 product next to that of the fp32 MFMA loop of the library on the same data.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I graph-convnet-uelb_amd/csrc -o graph-convnet-uelb_amd/lib/split_loop_bench.bin tools/split_loop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "kernels_wide.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SKB = GCN_WIDE_MAX_H / 32;          // k blocks of 32
constexpr int SROWB = GCN_WIDE_MAX_H * 2;         // bytes per LDS row of one plane

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// two fp32 values -> three packed bf16 pairs (low half = first value)
struct Split3 { uint32_t p[3]; };
__device__ __forceinline__ Split3 split2(float x, float y) {
    Split3 r;
    r.p[0] = cvt_pk_bf16(x, y);
    x -= __builtin_bit_cast(float, r.p[0] << 16); y -= __builtin_bit_cast(float, r.p[0] & 0xffff0000u);
    r.p[1] = cvt_pk_bf16(x, y);
    x -= __builtin_bit_cast(float, r.p[1] << 16); y -= __builtin_bit_cast(float, r.p[1] & 0xffff0000u);
    r.p[2] = cvt_pk_bf16(x, y);
    return r;
}

// W fp32 [H][H] -> Wt[plane][nb][kb][lane] (u32x4 = 8 bf16): lane (rs, q) holds W[16 nb + rs][32 kb + 8 q .. + 7]
__global__ void k_presplit(const float* __restrict__ W, u32x4* __restrict__ Wt, int H, int NB) {
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4;
    const int frag = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (frag >= NB * SKB) return;
    const int nb = frag / SKB, kb = frag - nb * SKB;
    const int n = 16 * nb + rs;
    u32x4 o[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 32 * kb + 8 * q + 2 * e;
        const float x = (n < H && k < H) ? W[(size_t)n * H + k] : 0.f, y = (n < H && k + 1 < H) ? W[(size_t)n * H + k + 1] : 0.f;
        const Split3 sp = split2(x, y);
                o[0][e] = sp.p[0]; o[1][e] = sp.p[1]; o[2][e] = sp.p[2];
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) Wt[((size_t)p * NB * SKB + frag) * 64 + lane] = o[p];
}

// byte offset of the 16-byte chunk (row, chunk) inside a plane
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * SROWB + ((chunk ^ (2 * ((row >> 1) & 3))) << 4); }

// NS row tiles x NC column blocks (nb, nb + 4) per pass.  acc[c][s] += sum over k.
template <int NS, int NC>
__device__ __forceinline__ void split_gemm(const unsigned char* planes, int plane_bytes, const u32x4* __restrict__ Wt, int NB, int nb,
                                           int lane, int rs, int q, f32x4 (&o)[NC][NS]) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) o[c][s] = f4zero();
    const size_t pstride = (size_t)NB * SKB * 64;
    u32x4 w[NC][3], wn[NC][3];
    auto loadw = [&](u32x4 (&d)[NC][3], int kb) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int nbc = min(nb + 4 * c, NB - 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) d[c][p] = Wt[p * pstride + ((size_t)nbc * SKB + min(kb, SKB - 1)) * 64 + lane];
        }
    };
    loadw(w, 0);
    for (int kb = 0; kb < SKB; ++kb) {
        loadw(wn, kb + 1);
        u32x4 a[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                a[s][p] = *reinterpret_cast<const u32x4*>(planes + p * plane_bytes + lds_off(16 * s + rs, 4 * kb + q));
        __builtin_amdgcn_sched_barrier(0);
#define MM(c, s, pw, pa) o[c][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[c][pw]), __builtin_bit_cast(bf16x8, a[s][pa]), o[c][s], 0, 0, 0)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int s = 0; s < NS; ++s) { MM(c, s, 2, 0); }
#pragma unroll
            for (int s = 0; s < NS; ++s) { MM(c, s, 0, 2); }
#pragma unroll
            for (int s = 0; s < NS; ++s) { MM(c, s, 1, 1); }
#pragma unroll
            for (int s = 0; s < NS; ++s) { MM(c, s, 1, 0); }
#pragma unroll
            for (int s = 0; s < NS; ++s) { MM(c, s, 0, 1); }
#pragma unroll
            for (int s = 0; s < NS; ++s) { MM(c, s, 0, 0); }
        }
#undef MM
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int p = 0; p < 3; ++p) w[c][p] = wn[c][p];
    }
}

// The same product with the weight fragments requested D k blocks ahead (a ring of D register sets).
template <int NS, int NC, int D>
__device__ __forceinline__ void split_gemm_d(const unsigned char* planes, int plane_bytes, const u32x4* __restrict__ Wt, int NB, int nb,
                                             int lane, int rs, int q, f32x4 (&o)[NC][NS]) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) o[c][s] = f4zero();
    const size_t pstride = (size_t)NB * SKB * 64;
    u32x4 w[D][NC][3];
    auto loadw = [&](u32x4 (&d)[NC][3], int kb) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int nbc = min(nb + 4 * c, NB - 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) d[c][p] = Wt[p * pstride + ((size_t)nbc * SKB + min(kb, SKB - 1)) * 64 + lane];
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) loadw(w[d], d);
    for (int kb0 = 0; kb0 < SKB; kb0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int kb = kb0 + d;
            if (kb < SKB) {
                u32x4 a[NS][3];
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        a[s][p] = *reinterpret_cast<const u32x4*>(planes + p * plane_bytes + lds_off(16 * s + rs, 4 * kb + q));
                __builtin_amdgcn_sched_barrier(0);
#define MM(c, s, pw, pa) o[c][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[d][c][pw]), __builtin_bit_cast(bf16x8, a[s][pa]), o[c][s], 0, 0, 0)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) { MM(c, s, 2, 0); }
#pragma unroll
                    for (int s = 0; s < NS; ++s) { MM(c, s, 0, 2); }
#pragma unroll
                    for (int s = 0; s < NS; ++s) { MM(c, s, 1, 1); }
#pragma unroll
                    for (int s = 0; s < NS; ++s) { MM(c, s, 1, 0); }
#pragma unroll
                    for (int s = 0; s < NS; ++s) { MM(c, s, 0, 1); }
#pragma unroll
                    for (int s = 0; s < NS; ++s) { MM(c, s, 0, 0); }
                }
#undef MM
                __builtin_amdgcn_sched_barrier(0);
            }
            loadw(w[d], kb + D);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// NC column blocks at once with the LDS operands of one tile read while the 6 NC MFMAs of the tile before it run (two
// register sets): the LDS reads of a k block are then 1 / NC of what one block per pass needs, and never waited for.
// D = 9 selects it.
template <int NS, int NC, bool ASM = false>
__device__ __forceinline__ void split_gemm_t(const unsigned char* planes, int plane_bytes, const u32x4* __restrict__ Wt, int NB, int nb,
                                             int lane, int rs, int q, f32x4 (&o)[NC][NS]) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) o[c][s] = f4zero();
    const size_t pstride = (size_t)NB * SKB * 64;
    u32x4 w[NC][3], wn[NC][3];
    auto loadw = [&](u32x4 (&d)[NC][3], int kb) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int nbc = min(nb + 4 * c, NB - 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) d[c][p] = Wt[p * pstride + ((size_t)nbc * SKB + min(kb, SKB - 1)) * 64 + lane];
        }
    };
    auto read_tile = [&](u32x4 (&a)[3], int s, int kb) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const u32x4*>(planes + p * plane_bytes + lds_off(16 * s + rs, 4 * min(kb, SKB - 1) + q));
    };
    loadw(w, 0);
    u32x4 a[2][3];
    read_tile(a[0], 0, 0);
    static_assert(SKB % 2 == 0, "two k blocks per trip");
    for (int kb0 = 0; kb0 < SKB; kb0 += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int kb = kb0 + d;
            loadw(wn, kb + 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int cur = (NS * d + s) & 1;
                if (s + 1 < NS) read_tile(a[cur ^ 1], s + 1, kb);
                else read_tile(a[cur ^ 1], 0, kb + 1);
                __builtin_amdgcn_sched_barrier(0);
#define MM(pw, pa) _Pragma("unroll") for (int c = 0; c < NC; ++c) { \
        if constexpr (ASM) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o[c][s]) : "v"(w[c][pw]), "v"(a[cur][pa])); \
        else o[c][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[c][pw]), __builtin_bit_cast(bf16x8, a[cur][pa]), o[c][s], 0, 0, 0); }
                MM(2, 0); MM(0, 2); MM(1, 1); MM(1, 0); MM(0, 1); MM(0, 0);
#undef MM
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int p = 0; p < 3; ++p) w[c][p] = wn[c][p];
        }
    }
}

template <int NS, int NC, int D = 1>
__global__ __launch_bounds__(256) void k_split(const float* __restrict__ X, const u32x4* __restrict__ Wt, float* __restrict__ out, int H,
                                               int iters, int write_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dynb[];
    const WideIdx ix = wide_idx();
    const int NB = (H + 15) / 16;
    constexpr int ROWS = 16 * NS, PB = ROWS * SROWB;
    // staging: thread (arow, g0) cuts float4 groups of its row; a group of 8 consecutive k is one 16-byte chunk per plane
    for (int r = ix.t >> 4; r < ROWS; r += 16) {
        for (int ch = ix.t & 15; ch < GCN_WIDE_MAX_H / 8; ch += 16) {
            u32x4 o[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * ch + 2 * e;
                const float x = k < H ? X[(size_t)r * H + k] : 0.f, y = k + 1 < H ? X[(size_t)r * H + k + 1] : 0.f;
                const Split3 sp = split2(x, y);
                o[0][e] = sp.p[0]; o[1][e] = sp.p[1]; o[2][e] = sp.p[2];
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dynb + p * PB + lds_off(r, ch)) = o[p];
        }
    }
    __syncthreads();
    f32x4 keep = f4zero();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int nb = ix.w; nb < NB; nb += 4 * NC) {
            f32x4 o[NC][NS];
            if (NC == 2 && D < 8 && nb + 4 >= NB) {
                f32x4 o1[1][NS];
                split_gemm<NS, 1>(dynb, PB, Wt, NB, nb, ix.lane, ix.rs, ix.q, o1);
#pragma unroll
                for (int s = 0; s < NS; ++s) { o[0][s] = o1[0][s]; o[NC - 1][s] = f4zero(); }
            } else {
                if constexpr (D == 9) split_gemm_t<NS, NC>(dynb, PB, Wt, NB, nb, ix.lane, ix.rs, ix.q, o);
                else if constexpr (D == 8) { split_gemm_t<NS, NC, true>(dynb, PB, Wt, NB, nb, ix.lane, ix.rs, ix.q, o); asm volatile("s_nop 7\n s_nop 7" ::: "memory"); }
                else if constexpr (D == 1) split_gemm<NS, NC>(dynb, PB, Wt, NB, nb, ix.lane, ix.rs, ix.q, o);
                else split_gemm_d<NS, NC, D>(dynb, PB, Wt, NB, nb, ix.lane, ix.rs, ix.q, o);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    keep += o[c][s];
                    const int col = 16 * (nb + 4 * c) + 4 * ix.q;
                    if (write_out && nb + 4 * c < NB && col < H) st4(out + (size_t)(16 * s + ix.rs) * H + col, o[c][s]);
                }
        }
        __syncthreads();
    }
    if (keep[0] == 123.456f) out[ix.t] = keep[1] + keep[2] + keep[3];
}

// the library's fp32 loop on the same data, for the error comparison and the time
template <int NS>
__global__ __launch_bounds__(256, 2) void k_f32(const float* __restrict__ X, const float* __restrict__ Wm, float* __restrict__ out, int H,
                                                int iters, int write_out) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float (*tile)[WLS] = reinterpret_cast<float (*)[WLS]>(dyn);
    const WideIdx ix = wide_idx();
    const int NB = (H + 15) / 16;
    for (int e = ix.t; e < 16 * WNS * WLS; e += 256) {
        const int r = e / WLS, c = e % WLS;
        dyn[e] = (c < H && r < 16 * NS) ? X[(size_t)r * H + c] : 0.f;
    }
    const rsrc_t rW = make_rsrc(Wm, (uint32_t)H * H * 4);
    __syncthreads();
    f32x4 keep = f4zero();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int nb = ix.w; nb < NB; nb += 4) {
            f32x4 o[WNS];
            wide_gemm_ns<0, NS>(tile, rW, H, NB, nb, ix.rs, ix.q, o);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                keep += o[s];
                const int col = 16 * nb + 4 * ix.q;
                if (write_out && col < H) st4(out + (size_t)(16 * s + ix.rs) * H + col, o[s]);
            }
        }
        __syncthreads();
    }
    if (keep[0] == 123.456f) out[ix.t] = keep[1] + keep[2] + keep[3];
}

static double max_err(const std::vector<float>& got, const std::vector<double>& ref, int rows, int H, double* scale) {
    double e = 0, s = 0;
    for (int i = 0; i < rows * H; ++i) { e = std::max(e, std::fabs((double)got[i] - ref[i])); s = std::max(s, std::fabs(ref[i])); }
    *scale = s;
    return e;
}

template <typename K, typename... A>
static float time_kernel(K kern, int grid, size_t lds, int iters, A... args) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<grid, 256, lds>>>(args..., 2, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<grid, 256, lds>>>(args..., iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  error: %s\n", hipGetErrorString(err));
    return ms;
}

template <int NS, int NC, int D = 1>
static void run_split(const float* X, const u32x4* Wt, float* out, int H, int wgs_per_cu, const std::vector<double>& ref) {
    const size_t lds_min = (size_t)3 * 16 * NS * SROWB;
    const size_t lds = wgs_per_cu == 1 ? std::max(lds_min, (size_t)90 * 1024) : lds_min;
    if (wgs_per_cu * lds > 160 * 1024) { printf("split bf16 NS=%d NC=%d wg/cu=%d: does not fit LDS\n", NS, NC, wgs_per_cu); return; }
    const int iters = 40, grid = 256 * wgs_per_cu;
    const float ms = time_kernel(k_split<NS, NC, D>, grid, lds, iters, X, Wt, out, H);
    k_split<NS, NC, D><<<1, 256, lds>>>(X, Wt, out, H, 1, 1);
    std::vector<float> h(16 * NS * H);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    double scale;
    const double e = max_err(h, ref, 16 * NS, H, &scale);
    const double flops = 2.0 * 16 * NS * H * (double)H * iters * grid;
    printf("split bf16 x6  rows/step=%d colblocks/pass=%d ahead=%d wg/cu=%d: %.3f ms, %.1f TFLOP/s fp32-equivalent (unpadded); max err %.3e of max |out| %.3e = %.2e\n",
           16 * NS, NC, D, wgs_per_cu, ms, flops / (ms * 1e-3) * 1e-12, e, scale, e / scale);
}
template <int NS>
static void run_f32(const float* X, const float* W, float* out, int H, int wgs_per_cu, const std::vector<double>& ref) {
    const size_t lds = wgs_per_cu == 1 ? 90 * 1024 : sizeof(float) * 16 * WNS * WLS;
    const int iters = 40, grid = 256 * wgs_per_cu;
    const float ms = time_kernel(k_f32<NS>, grid, lds, iters, X, W, out, H);
    k_f32<NS><<<1, 256, lds>>>(X, W, out, H, 1, 1);
    std::vector<float> h(16 * NS * H);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    double scale;
    const double e = max_err(h, ref, 16 * NS, H, &scale);
    const double flops = 2.0 * 16 * NS * H * (double)H * iters * grid;
    printf("fp32 16x16x4   rows/step=%d                  wg/cu=%d: %.3f ms, %.1f TFLOP/s (unpadded); max err %.3e of max |out| %.3e = %.2e\n",
           16 * NS, wgs_per_cu, ms, flops / (ms * 1e-3) * 1e-12, e, scale, e / scale);
}

int main() {
    const int H = 300, NB = (H + 15) / 16, ROWS = 64;
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hX((size_t)ROWS * H), hW((size_t)H * H);
    for (auto& v : hX) v = nd(gen) * 3.f;
    for (auto& v : hW) v = nd(gen) * 0.06f;
    std::vector<double> ref((size_t)ROWS * H);
    for (int r = 0; r < ROWS; ++r)
        for (int n = 0; n < H; ++n) {
            double s = 0;
            for (int k = 0; k < H; ++k) s += (double)hX[(size_t)r * H + k] * (double)hW[(size_t)n * H + k];
            ref[(size_t)r * H + n] = s;
        }
    float *X, *W, *out;
    u32x4* Wt;
    hipMalloc(&X, hX.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&out, (size_t)ROWS * H * 4 + 4096);
    hipMalloc(&Wt, (size_t)3 * NB * SKB * 64 * 16);
    hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    k_presplit<<<(NB * SKB + 3) / 4, 256>>>(W, Wt, H, NB);
    hipDeviceSynchronize();
    // tile-pipelined form (ahead=9), one workgroup of four wavefronts per CU = one wavefront per SIMD
    run_split<4, 1, 9>(X, Wt, out, H, 1, ref);
    run_split<4, 2, 9>(X, Wt, out, H, 1, ref);
    run_split<4, 3, 9>(X, Wt, out, H, 1, ref);
    run_split<4, 5, 9>(X, Wt, out, H, 1, ref);
    run_split<4, 5, 8>(X, Wt, out, H, 1, ref);
    run_split<4, 3, 8>(X, Wt, out, H, 1, ref);
    run_split<4, 2, 8>(X, Wt, out, H, 1, ref);
    run_split<4, 1, 8>(X, Wt, out, H, 1, ref);
    for (int wpc = 1; wpc <= 2; ++wpc) {
        run_f32<3>(X, W, out, H, wpc, ref);
        run_f32<2>(X, W, out, H, wpc, ref);
        run_split<2, 1>(X, Wt, out, H, wpc, ref);
        run_split<2, 2>(X, Wt, out, H, wpc, ref);
        run_split<3, 1>(X, Wt, out, H, wpc, ref);
        run_split<3, 2>(X, Wt, out, H, wpc, ref);
        run_split<4, 1>(X, Wt, out, H, wpc, ref);
        run_split<4, 2>(X, Wt, out, H, wpc, ref);
        run_split<2, 1, 2>(X, Wt, out, H, wpc, ref);
        run_split<2, 1, 3>(X, Wt, out, H, wpc, ref);
        run_split<2, 1, 4>(X, Wt, out, H, wpc, ref);
        run_split<2, 2, 2>(X, Wt, out, H, wpc, ref);
        run_split<2, 2, 3>(X, Wt, out, H, wpc, ref);
        run_split<3, 1, 2>(X, Wt, out, H, wpc, ref);
        run_split<3, 1, 3>(X, Wt, out, H, wpc, ref);
        run_split<3, 1, 4>(X, Wt, out, H, wpc, ref);
        run_split<3, 2, 2>(X, Wt, out, H, wpc, ref);
    }
    return 0;
}
