// Microbenchmark of the product loop of the wide edge kernels (measurement aid, not part of the library):
// 48 LDS rows x H columns against 16-column blocks of an H x H weight matrix streamed from L2, as in k_edge_fwd_w's
// phase B, without the staging phase and the epilogue.  Prints the fp32 MFMA pipe utilisation for each loop variant
// at one and at two 256-thread workgroups per CU.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I graph-convnet-uelb_amd/csrc -o graph-convnet-uelb_amd/lib/wide_loop_bench.bin tools/wide_loop_bench.hip
//   (the binary travels to the GPU box with the tree; run it there as ./graph-convnet-uelb_amd/lib/wide_loop_bench.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels_wide.hpp"

// Software-pipelined form of wide_gemm_ns (weights two groups of four k blocks ahead in three register sets, the LDS
// operands of a k block requested while the MFMAs of the block before run).  It is no faster than wide_gemm_ns in any
// setting measured here, which is why the library does not carry it.
template <int NS>
__device__ __forceinline__ void wide_gemm_p(const float (*tile)[WLS], rsrc_t rW, int H, int NB, int nb, int rs, int q,
                                            f32x4 (&o)[WNS]) {
    f32x4 a[NS], b[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { a[s] = f4zero(); b[s] = f4zero(); }
    const int n = 16 * nb + rs;
    const uint32_t rowoff = n < H ? (uint32_t)n * (uint32_t)H * 4u : kOob;
    const int NG = (NB + 3) / 4, last_kb = 4 * NG - 1;
    const float* lbase = &tile[rs][4 * q];
    auto loadw = [&](f32x4 (&wf)[4], int g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 16 * (4 * g + u) + 4 * q;
            wf[u] = bld4(rW, (4 * g + u < NB && k < H && n < H) ? rowoff + (uint32_t)k * 4u : kOob);
        }
    };
    auto loadl = [&](f32x4 (&in)[NS], int kb) {
        const float* p = lbase + 16 * min(kb, last_kb);
#pragma unroll
        for (int s = 0; s < NS; ++s) in[s] = ld4(p + 16 * s * WLS);
    };
    auto mm = [&](const f32x4& w, const f32x4 (&in)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            a[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], in[s][0], a[s], 0, 0, 0);
            b[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], in[s][1], b[s], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            a[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], in[s][2], a[s], 0, 0, 0);
            b[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], in[s][3], b[s], 0, 0, 0);
        }
    };
    f32x4 w0[4], w1[4], w2[4], i0[NS], i1[NS];
    loadw(w0, 0);
    loadw(w1, 1);
    loadl(i0, 0);
    auto group = [&](const f32x4 (&wc)[4], f32x4 (&wn)[4], int g) {
        loadw(wn, g + 2);
        loadl(i1, 4 * g + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(wc[0], i0);
        __builtin_amdgcn_sched_barrier(0);
        loadl(i0, 4 * g + 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(wc[1], i1);
        __builtin_amdgcn_sched_barrier(0);
        loadl(i1, 4 * g + 3);
        __builtin_amdgcn_sched_barrier(0);
        mm(wc[2], i0);
        __builtin_amdgcn_sched_barrier(0);
        loadl(i0, 4 * g + 4);
        __builtin_amdgcn_sched_barrier(0);
        mm(wc[3], i1);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int g = 0; g < NG; g += 3) {
        group(w0, w2, g);
        if (g + 1 < NG) group(w1, w0, g + 1);
        if (g + 2 < NG) group(w2, w1, g + 2);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] = a[s] + b[s];
}

template <int VAR>
__global__ __launch_bounds__(256, 2) void k_loop(const float* __restrict__ Wm, float* __restrict__ out, int H, int iters,
                                                 unsigned long long* __restrict__ clk) {
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float (*tile)[WLS] = reinterpret_cast<float (*)[WLS]>(dyn);
    const WideIdx ix = wide_idx();
    const int NB = (H + 15) / 16;
    for (int e = ix.t; e < 16 * WNS * WLS; e += 256) dyn[e] = (e % WLS) < H ? 0.001f * (float)((e * 7) % 13) : 0.f;
    const rsrc_t rW = make_rsrc(Wm, (uint32_t)H * H * 4);
    __syncthreads();
    f32x4 acc = f4zero();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int kk = 0; kk < WKB; ++kk) {
            const int nb = ix.w + 4 * kk;
            if (nb < NB) {
                f32x4 o[WNS];
                if constexpr (VAR < 3) wide_gemm_ns<VAR, 3>(tile, rW, H, NB, nb, ix.rs, ix.q, o);
                else if constexpr (VAR == 3) wide_gemm_p<3>(tile, rW, H, NB, nb, ix.rs, ix.q, o);
                else if constexpr (VAR == 8 || VAR == 9 || VAR == 10) {
                    // 32x32x2 MFMAs, the same flops per column block as the 16x16x4 loops (two 32-row tiles x 16 of the 32
                    // features are counted: one item = 32 features x 64 rows per TWO column blocks).
                    // 8: registers only; 9: registers only, operands rotating over eight registers;
                    // 10: weights streamed (one group ahead), LDS operands one k block ahead
                    typedef float f32x16 __attribute__((ext_vector_type(16)));
                    f32x16 c0, c1;
                    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
                    const int NBK = (NB + 3) & ~3;           // k blocks of 16 -> 8 MFMAs of k = 2 each per 32x32 tile
                    // per column block of 16 features x 48 rows the 16x16x4 loop issues 12 * NBK MFMAs = 12 NBK * 2048 flop;
                    // the same flop count here: 6 NBK MFMAs of 4096 flop
                    const int n_mfma = 6 * NBK;
                    if constexpr (VAR == 8) {
                        const float x = (float)ix.rs, y = (float)ix.q;
                        for (int k = 0; k < n_mfma / 2; ++k) {
                            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
                            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0);
                        }
                    } else if constexpr (VAR == 9) {
                        float x[8];
                        for (int u = 0; u < 8; ++u) x[u] = (float)(ix.rs + u);
                        for (int k = 0; k < n_mfma / 8; ++k) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], x[7 - u], c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u + 4], x[3 - u], c1, 0, 0, 0);
                            }
                        }
                    } else {
                        const int l32 = ix.lane & 31, blk = ix.lane >> 5;
                        const int n = 32 * (nb >> 1) + l32;
                        const uint32_t rowoff = n < H ? (uint32_t)n * (uint32_t)H * 4u : kOob;
                        const float* lbase = &tile[l32][4 * blk];
                        f32x4 wa[2], wb[2], i0[2], i1[2];
                        // a group = 16 k = two b128 per lane (k = 16 g + 8 h + 4 blk + e)
                        auto loadw = [&](f32x4 (&wf)[2], int g) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int k = 16 * g + 8 * h + 4 * blk;
                                wf[h] = bld4(rW, (k < H && n < H) ? rowoff + (uint32_t)k * 4u : kOob);
                            }
                        };
                        auto loadl = [&](f32x4 (&in)[2], int g, int h) {
                            const float* p = lbase + 16 * min(g, NBK - 1) + 8 * h;
                            in[0] = ld4(p); in[1] = ld4(p + 16 * WLS);          // rows l32 and 16 + l32 stand in for 32 + l32
                        };
                        auto mm = [&](const f32x4& w, const f32x4 (&in)[2]) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], in[0][e], c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], in[1][e], c1, 0, 0, 0);
                            }
                        };
                        // 6 NBK MFMAs = NBK * 3/4 ... keep it simple: NBK groups of 16 k give 16 MFMAs each for a 32 x 64 item,
                        // i.e. 16 NBK MFMAs per item; an item stands for 32/16 * 64/48 = 2.67 column blocks: run 3/8 of the groups
                        const int ng = (3 * NBK + 7) / 8;
                        loadw(wa, 0);
                        loadl(i0, 0, 0);
                        for (int g = 0; g < ng; g += 2) {
                            loadw(wb, g + 1);
                            loadl(i1, g, 1);
                            __builtin_amdgcn_sched_barrier(0);
                            mm(wa[0], i0);
                            __builtin_amdgcn_sched_barrier(0);
                            loadl(i0, g + 1, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            mm(wa[1], i1);
                            __builtin_amdgcn_sched_barrier(0);
                            loadw(wa, g + 2);
                            loadl(i1, g + 1, 1);
                            __builtin_amdgcn_sched_barrier(0);
                            mm(wb[0], i0);
                            __builtin_amdgcn_sched_barrier(0);
                            loadl(i0, g + 2, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            mm(wb[1], i1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    for (int s = 0; s < 3; ++s) o[s] = f32x4{c0[0] + c1[4 * s], c0[1] + c1[5], c0[2 + s] + c1[9], c0[7] + c1[15]};
                }
                else if constexpr (VAR == 5 || VAR == 6 || VAR == 7) {
                    // 5: LDS operands as in wide_gemm_p, weights fixed in registers; 6: weights streamed, LDS operands fixed;
                    // 7: as 5 with one 16-row tile's operand (ds_read_b128) feeding all three chains
                    f32x4 a[3], b[3], i0[3], i1[3], w0[4];
                    for (int s = 0; s < 3; ++s) { a[s] = f4zero(); b[s] = f4zero(); i0[s] = f4zero() + (float)s; i1[s] = i0[s]; }
                    const float* lbase = &tile[ix.rs][4 * ix.q];
                    const int NBK = (NB + 3) & ~3;
                    const int n = 16 * nb + ix.rs;
                    for (int u = 0; u < 4; ++u) w0[u] = f4zero() + (float)(u + ix.q);
                    auto mm = [&](const f32x4& w, const f32x4 (&in)[3]) {
#pragma unroll
                        for (int s = 0; s < 3; ++s) {
                            a[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], in[s][0], a[s], 0, 0, 0);
                            b[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], in[s][1], b[s], 0, 0, 0);
                        }
#pragma unroll
                        for (int s = 0; s < 3; ++s) {
                            a[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], in[s][2], a[s], 0, 0, 0);
                            b[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], in[s][3], b[s], 0, 0, 0);
                        }
                    };
                    if constexpr (VAR == 5 || VAR == 7) {
                        constexpr int NL = VAR == 5 ? 3 : 1;
                        for (int s = 0; s < NL; ++s) i0[s] = ld4(lbase + 16 * s * WLS);
                        for (int kb = 0; kb < NBK; kb += 2) {
                            for (int s = 0; s < NL; ++s) i1[s] = ld4(lbase + 16 * s * WLS + 16 * (kb + 1));
                            __builtin_amdgcn_sched_barrier(0);
                            mm(w0[kb & 3], i0);
                            __builtin_amdgcn_sched_barrier(0);
                            for (int s = 0; s < NL; ++s) i0[s] = ld4(lbase + 16 * s * WLS + 16 * min(kb + 2, NBK - 1));
                            __builtin_amdgcn_sched_barrier(0);
                            mm(w0[(kb + 1) & 3], i1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        f32x4 w1[4];
                        auto loadw = [&](f32x4 (&wf)[4], int g) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int k = 16 * (4 * g + u) + 4 * ix.q;
                                wf[u] = bld4(rW, (4 * g + u < NB && k < H && n < H) ? (uint32_t)n * (uint32_t)H * 4u + (uint32_t)k * 4u : kOob);
                            }
                        };
                        loadw(w0, 0);
                        for (int g = 0; g < NBK / 4; g += 2) {
                            loadw(w1, g + 1);
                            __builtin_amdgcn_sched_barrier(0);
                            for (int u = 0; u < 4; ++u) mm(w0[u], i0);
                            __builtin_amdgcn_sched_barrier(0);
                            loadw(w0, g + 2);
                            __builtin_amdgcn_sched_barrier(0);
                            if (g + 1 < NBK / 4) for (int u = 0; u < 4; ++u) mm(w1[u], i0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    for (int s = 0; s < 3; ++s) o[s] = a[s] + b[s];
                } else {          // registers only: the issue rate of the matrix pipe itself
                    f32x4 a[3], b[3];
                    for (int s = 0; s < 3; ++s) { a[s] = f4zero(); b[s] = f4zero(); }
                    const float x = (float)ix.rs, y = (float)ix.q;
                    for (int k = 0; k < 4 * ((NB + 3) & ~3) / 2; ++k) {
#pragma unroll
                        for (int s = 0; s < 3; ++s) {
                            a[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[s], 0, 0, 0);
                            b[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b[s], 0, 0, 0);
                        }
                    }
                    for (int s = 0; s < 3; ++s) o[s] = a[s] + b[s];
                }
                acc += o[0] + o[1] + o[2];
            }
        }
        __syncthreads();
    }
    if (acc[0] == 123.456f) out[ix.t] = acc[1] + acc[2] + acc[3];
    if (blockIdx.x == 0 && ix.t == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

template <int VAR>
void run(const char* name, const float* W, float* out, int H, int wgs_per_cu) {
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    const int iters = 40, grid = 256 * wgs_per_cu;
    // one workgroup per CU: ask for more than half of the LDS
    const size_t lds = wgs_per_cu == 1 ? 90 * 1024 : sizeof(float) * 16 * WNS * WLS;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_loop<VAR><<<grid, 256, lds>>>(W, out, H, 2, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_loop<VAR><<<grid, 256, lds>>>(W, out, H, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int NB = (H + 15) / 16, NBK = (NB + 3) & ~3;
    // MFMAs per SIMD: wavefront w of every workgroup on the CU takes the blocks w, w + 4, ...; the busiest SIMD has ceil(NB / 4)
    const double mfma_busiest = (double)wgs_per_cu * iters * ((NB + 3) / 4) * (4.0 * NB) * 3;     // unpadded k steps
    const double mfma_issued = (double)wgs_per_cu * iters * ((NB + 3) / 4) * (4.0 * NBK) * 3;
    unsigned long long hc[2];
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)hc[0] / ((double)hc[1] / 100.0);      // wall clock counts at 100 MHz
    const double cyc = ms * 1e-3 * 2.4e9;
    const double flops = 2.0 * 48 * H * (double)H * iters * grid;
    printf("%-28s H=%d wg/cu=%d: %.3f ms, pipe busy (busiest SIMD, useful / issued) %.3f / %.3f, %.1f TFLOP/s unpadded, shader clock %.0f MHz\n", name, H, wgs_per_cu, ms,
           mfma_busiest * 32 / cyc, mfma_issued * 32 / cyc, flops / (ms * 1e-3) * 1e-12, mhz);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  error: %s\n", hipGetErrorString(err));
}

int main() {
    const int H = 300;
    float *W, *out;
    hipMalloc(&W, sizeof(float) * H * H);
    hipMalloc(&out, 4096);
    std::vector<float> h(H * H);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)(i % 17);
    hipMemcpy(W, h.data(), sizeof(float) * H * H, hipMemcpyHostToDevice);
    for (int wpc = 1; wpc <= 2; ++wpc) {
        run<0>("no prefetch", W, out, H, wpc);
        run<1>("weights one group ahead", W, out, H, wpc);
        run<2>("weights two groups ahead", W, out, H, wpc);
        run<3>("pipelined (wide_gemm_p)", W, out, H, wpc);
        run<4>("registers only", W, out, H, wpc);
        run<5>("LDS operands only", W, out, H, wpc);
        run<7>("one LDS read per 12 MFMAs", W, out, H, wpc);
        run<6>("streamed weights only", W, out, H, wpc);
        run<8>("32x32x2 registers only", W, out, H, wpc);
        run<9>("32x32x2 rotating registers", W, out, H, wpc);
        run<10>("32x32x2 streamed + LDS", W, out, H, wpc);
    }
    return 0;
}
