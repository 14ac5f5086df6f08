// Layer and head kernels for hidden sizes above 64 (up to GCN_WIDE_MAX_H): the same two-phase structure as
// kernels_strip.hpp with 256-thread workgroups, but
//   * phase A: thread (arow, g0) walks the float4 column groups g0, g0 + 16, ... of its row,
//   * phase B: wavefront w walks the 16-column blocks w, w + 4, ... of the output,
//   * the H x H weights do not fit in registers or LDS: each 16-row tile streams the weight rows of its
//     column block from memory (L2-resident) as MFMA A operands.
// One LDS tile per operand (single buffered, two barriers per step); the head kernels take up to three 16-row
// tiles per step against one fetch of the weight rows (wide_gemm_ns).  The math of each kernel is stated above its
// argument block in kernels_fused.hpp.
#pragma once
#include "common.hpp"
#include "kernels_fused.hpp"
#include "kernels_bwd.hpp"
#include "kernels_strip.hpp"

constexpr int GCN_WIDE_MAX_H = 320;
constexpr int WNB = GCN_WIDE_MAX_H / 16;       // column blocks
constexpr int WLS = GCN_WIDE_MAX_H + 8;        // LDS row stride (floats), 8 mod 64: the 128-bit MFMA operand reads of lane (rs, q) at row rs, column 4 q fall on 16 different bank quads in every 16-lane group of ds_read_b128 (4 mod 64 is two-way conflicted)
constexpr int WKA = GCN_WIDE_MAX_H / 64;       // column groups per phase-A thread
constexpr int WKB = (WNB + 3) / 4;             // column blocks per wavefront

struct WideIdx {
    int t, lane, rs, q, w, arow, g0;
};
__device__ __forceinline__ WideIdx wide_idx() {
    WideIdx x;
    x.t = threadIdx.x; x.lane = x.t & 63; x.rs = x.lane & 15; x.q = x.lane >> 4; x.w = x.t >> 6;
    x.arow = x.t >> 4; x.g0 = x.t & 15;
    return x;
}

// Output columns 16 nb + 4 q .. + 3 of row rs of tile * W^T, W row-major [H][H] behind descriptor rW.
// The weight rows come from L2 in groups of four 16-column blocks.  PIPE: two register sets alternate so that the
// next group's loads are in flight during the current group's MFMAs (groups past the end use out-of-range offsets:
// no memory traffic, zeros); worth its 16 registers only where the kernel's occupancy does not drop for it (the
// backward kernels, already at one workgroup per CU: 7.5 -> 6.3 ms per layer at H = 300; the forward edge kernel
// loses its second workgroup and gets slower).
template <bool PIPE = false>
__device__ __forceinline__ f32x4 wide_gemm(const float (*tile)[WLS], rsrc_t rW, int H, int NB, int nb, int rs, int q) {
    f32x4 acc0 = f4zero(), acc1 = f4zero();
    const int n = 16 * nb + rs;
    const uint32_t rowoff = n < H ? (uint32_t)n * (uint32_t)H * 4u : kOob;
    auto load = [&](f32x4 (&wf)[4], int s0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 16 * (s0 + u) + 4 * q;
            wf[u] = bld4(rW, (s0 + u < NB && k < H && n < H) ? rowoff + (uint32_t)k * 4u : kOob);
        }
    };
    auto compute = [&](const f32x4 (&wf)[4], int s0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (s0 + u < NB) {
                const f32x4 in = ld4(&tile[rs][16 * (s0 + u) + 4 * q]);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][0], in[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][1], in[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][2], in[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][3], in[3], acc1, 0, 0, 0);
            }
        }
    };
    f32x4 wa[4];
    if constexpr (PIPE) {
        f32x4 wb[4];
        load(wa, 0);
        for (int s0 = 0; s0 < NB; s0 += 8) {
            load(wb, s0 + 4);
            compute(wa, s0);
            load(wa, s0 + 8);
            compute(wb, s0 + 4);
        }
    } else {
        for (int s0 = 0; s0 < NB; s0 += 4) {
            load(wa, s0);
            compute(wa, s0);
        }
    }
    return acc0 + acc1;
}

__device__ __forceinline__ f32x4 bn_relu_res4(const f32x4 tp, const f32x4 rin, const float* mu, const float* rsd,
                                              const float* ga, const float* be) {
    const f32x4 pm = ld4(mu), pr = ld4(rsd), pg = ld4(ga), pb = ld4(be);
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float bn = (tp[k] - pm[k]) * pr[k] * pg[k] + pb[k];
        v[k] = (bn > 0.f ? bn : 0.f) + rin[k];
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// Node forward (NodeFwdArgs): one workgroup per 16-row tile, grid-stride.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_node_fwd_w(NodeFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[16][WLS];
    __shared__ __attribute__((aligned(16))) float cl[4][GCN_WIDE_MAX_H];
    const WideIdx ix = wide_idx();
    const int H = a.H, NB = (H + 15) / 16, HP = 16 * NB;
    clds_fill(cl[0], a.apply ? a.pmean : nullptr, H, HP); clds_fill(cl[1], a.apply ? a.prstd : nullptr, H, HP);
    clds_fill(cl[2], a.apply ? a.pgamma : nullptr, H, HP); clds_fill(cl[3], a.apply ? a.pbeta : nullptr, H, HP);
    const DropCfg dc = drop_cfg(a.rng, a.drop_sid, a.drop_p, a.drop_on && a.apply);
    rsrc_t rW[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rW[k] = make_rsrc(a.W[k], (uint32_t)H * H * 4);
    const int ntiles = (a.rows + 15) / 16;
    __syncthreads();
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int row = tl * 16 + ix.arow;
        for (int g = ix.g0; g < HP / 4; g += 16) {
            const int acol = 4 * g;
            f32x4 val = f4zero();
            if (row < a.rows && acol < H) {
                const size_t off = (size_t)row * H + acol;
                if (!a.apply) {
                    val = ld4(a.x_in + off);
                } else {
                    val = bn_relu_res4(ld4(a.prev_tmp + off), ld4(a.prev_in + off), &cl[0][acol], &cl[1][acol], &cl[2][acol], &cl[3][acol]);
                    if (dc.on) val *= drop_mul4(dc, drop_rowhash(dc, (uint32_t)row), g);
                    if (a.x_out != nullptr) st4(a.x_out + off, val);
                }
            }
            st4(&tile[ix.arow][acol], val);
        }
        __syncthreads();
        const int rowb = tl * 16 + ix.rs;
        for (int nb = ix.w; nb < NB; nb += 4) {
            const int bcol = 16 * nb + 4 * ix.q;
            const bool valid = rowb < a.rows && bcol < H;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                f32x4 o = wide_gemm<true>(tile, rW[k], H, NB, nb, ix.rs, ix.q);
                if (valid) st4(a.y[k] + (size_t)rowb * H + bcol, o + ld4(a.bias[k] + bcol));
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Up to WNS 16-row tiles (LDS rows 16 s .. 16 s + 15) against the same weight rows: every group of weight rows
// fetched from L2 feeds WNS times the MFMAs of wide_gemm.  These kernels are bound by the L2 traffic of the
// re-streamed weights (the whole H x H matrix per step of a workgroup), so rows per step is what sets their speed.
// ------------------------------------------------------------------------------------------------
constexpr int WNS = 3;
// x[0] <- x[1] <- ... <- x[WKB-1] <- x[0]: after WKB calls every entry is back in place
__device__ __forceinline__ void rot(f32x4 (&x)[WKB]) {
    const f32x4 t = x[0];
#pragma unroll
    for (int k = 0; k + 1 < WKB; ++k) x[k] = x[k + 1];
    x[WKB - 1] = t;
}

// The k loop runs over groups of four 16-column blocks without a per-block test: the caller keeps the LDS columns
// from 16 NB up to the next multiple of 64 at zero (wide_zero_pad), and weight blocks past NB load as zeros.  The LDS
// operands of half a group are requested before its first MFMA.
template <int PIPE, int NS>
__device__ __forceinline__ void wide_gemm_ns(const float (*tile)[WLS], rsrc_t rW, int H, int NB, int nb, int rs, int q,
                                             f32x4 (&o)[WNS], unsigned long long* wait_acc = nullptr) {
    f32x4 a[NS], b[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { a[s] = f4zero(); b[s] = f4zero(); }
    const int n = 16 * nb + rs;
    const uint32_t rowoff = n < H ? (uint32_t)n * (uint32_t)H * 4u : kOob;
    auto load = [&](f32x4 (&wf)[4], int s0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 16 * (s0 + u) + 4 * q;
            wf[u] = bld4(rW, (s0 + u < NB && k < H && n < H) ? rowoff + (uint32_t)k * 4u : kOob);
        }
    };
    auto compute = [&](const f32x4 (&wf)[4], int s0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 in[NS][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < NS; ++s) in[s][u] = ld4(&tile[16 * s + rs][16 * (s0 + 2 * h + u) + 4 * q]);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads above the MFMAs (the scheduler would sink each to its use)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    a[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2 * h + u][0], in[s][u][0], a[s], 0, 0, 0);
                    b[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2 * h + u][1], in[s][u][1], b[s], 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    a[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2 * h + u][2], in[s][u][2], a[s], 0, 0, 0);
                    b[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2 * h + u][3], in[s][u][3], b[s], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x4 wa[4];
    if constexpr (PIPE == 2) {              // weight groups requested two groups ahead of their MFMAs
        f32x4 wb[4], wc[4];
        load(wa, 0);
        load(wb, 4);
        for (int s0 = 0; s0 < NB; s0 += 12) {
            load(wc, s0 + 8);
            __builtin_amdgcn_sched_barrier(0);
            compute(wa, s0);
            load(wa, s0 + 12);
            __builtin_amdgcn_sched_barrier(0);
            if (s0 + 4 < NB) compute(wb, s0 + 4);
            load(wb, s0 + 16);
            __builtin_amdgcn_sched_barrier(0);
            if (s0 + 8 < NB) compute(wc, s0 + 8);
        }
    } else if constexpr (PIPE == 1) {       // one group ahead
        f32x4 wb[4];
        load(wa, 0);
        for (int s0 = 0; s0 < NB; s0 += 8) {
            load(wb, s0 + 4);
            __builtin_amdgcn_sched_barrier(0);
            compute(wa, s0);
            load(wa, s0 + 8);
            __builtin_amdgcn_sched_barrier(0);
            if (s0 + 4 < NB) compute(wb, s0 + 4);
        }
    } else {
        for (int s0 = 0; s0 < NB; s0 += 4) {
            load(wa, s0);
#ifdef GCN_ACC_GEMM
            {   // measurement builds: cycles spent waiting for the weight rows (and whatever was requested before them)
                const unsigned long long t0 = clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (wait_acc != nullptr) *wait_acc += clock64() - t0;
            }
#endif
            compute(wa, s0);
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] = a[s] + b[s];
}
// ns (wavefront-uniform) = tiles in use, 1 .. WNS; o[s] for s >= ns is left untouched
template <int PIPE>
__device__ __forceinline__ void wide_gemm_n(const float (*tile)[WLS], int ns, rsrc_t rW, int H, int NB, int nb, int rs, int q,
                                            f32x4 (&o)[WNS], unsigned long long* wait_acc = nullptr) {
    if (ns >= 3) wide_gemm_ns<PIPE, 3>(tile, rW, H, NB, nb, rs, q, o, wait_acc);
    else if (ns == 2) wide_gemm_ns<PIPE, 2>(tile, rW, H, NB, nb, rs, q, o, wait_acc);
    else wide_gemm_ns<PIPE, 1>(tile, rW, H, NB, nb, rs, q, o, wait_acc);
}
// zero columns 16 NB .. 16 ceil4(NB) - 1 of a 16 WNS-row LDS tile (never written afterwards)
__device__ __forceinline__ void wide_zero_pad(float (*tile)[WLS], int NB, int t) {
    const int c0 = 16 * NB, c1 = 16 * ((NB + 3) & ~3);
    for (int e = t; e < 16 * WNS * (c1 - c0); e += 256) tile[e / (c1 - c0)][c0 + e % (c1 - c0)] = 0.f;
}
// sum over the 16 rows of a block of two per-lane partial sums, added by the one lane that owns the four columns into
// per-workgroup LDS accumulators (same lane, same address every time: no atomics, fixed order)
__device__ __forceinline__ void wide_acc_cols(float (*acc)[GCN_WIDE_MAX_H], int bcol, bool own, f32x4 t1, f32x4 t2) {
    t1 = red16x4(t1); t2 = red16x4(t2);
    if (own) {
        st4(&acc[0][bcol], ld4(&acc[0][bcol]) + t1);
        st4(&acc[1][bcol], ld4(&acc[1][bcol]) + t2);
    }
}

// Phase accounting (measurement builds, -DGCN_PHASE_ACC=1 forward / =2 backward edge kernel): every wavefront adds up
// the shader-clock cycles it spends in each phase of its steps over the whole launch; g_phase_acc[workgroup][wavefront][slot]
// (up to twelve wavefronts per workgroup).
#ifdef GCN_PHASE_ACC
__device__ unsigned long long g_phase_acc[1024][12][8];
#define GCN_ACC_DECL unsigned long long pa_t_ = clock64(), pa_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GCN_ACC(slot) do { const unsigned long long n_ = clock64(); pa_[slot] += n_ - pa_t_; pa_t_ = n_; } while (0)
#define GCN_ACC_END(w, lane) do { if ((lane) == 0 && blockIdx.x < 1024) for (int k_ = 0; k_ < 8; ++k_) g_phase_acc[blockIdx.x][w][k_] = pa_[k_]; } while (0)
#endif
#if defined(GCN_PHASE_ACC) && GCN_PHASE_ACC == 1
#define ACCF_DECL GCN_ACC_DECL
#define ACCF(slot) GCN_ACC(slot)
#define ACCF_END(w, lane) GCN_ACC_END(w, lane)
#else
#define ACCF_DECL do { } while (0)
#define ACCF(slot) do { } while (0)
#define ACCF_END(w, lane) do { } while (0)
#endif
#if defined(GCN_PHASE_ACC) && GCN_PHASE_ACC == 2
#define ACCB_DECL GCN_ACC_DECL
#define ACCB(slot) GCN_ACC(slot)
#define ACCB_END(w, lane) GCN_ACC_END(w, lane)
#else
#define ACCB_DECL do { } while (0)
#define ACCB(slot) do { } while (0)
#define ACCB_END(w, lane) do { } while (0)
#endif
// The edge kernels of this hidden-size range are in kernels_wide_bf.hpp (products on the bf16 matrix pipe).

// ------------------------------------------------------------------------------------------------
// Node backward (NodeBwdArgs): one workgroup per 16-row tile (every workgroup of the grid writes a partial row).
// The three products share one LDS tile, filled three times.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_node_bwd_w(NodeBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[16][WLS];
    __shared__ __attribute__((aligned(16))) float cc[10][GCN_WIDE_MAX_H];   // 0..5 this BN, 6..9 layer below
    const WideIdx ix = wide_idx();
    const int H = a.H, NB = (H + 15) / 16, HP = 16 * NB;
    const bool do_next = a.next.tmp != nullptr;
    clds_fill(cc[0], a.mean, H, HP); clds_fill(cc[1], a.rstd, H, HP); clds_fill(cc[2], a.gamma, H, HP);
    clds_fill(cc[3], a.beta, H, HP); clds_fill(cc[4], a.bstats, H, HP); clds_fill(cc[5], a.bstats + H, H, HP);
    clds_fill(cc[6], do_next ? a.next.mean : nullptr, H, HP); clds_fill(cc[7], do_next ? a.next.rstd : nullptr, H, HP);
    clds_fill(cc[8], do_next ? a.next.gamma : nullptr, H, HP); clds_fill(cc[9], do_next ? a.next.beta : nullptr, H, HP);
    const DropCfg dcx = drop_cfg(a.rng, a.sid_x, a.p, a.training);
    const DropCfg ndc = drop_cfg(a.rng, a.next.drop_sid, a.p, a.training && do_next);
    const rsrc_t rWu = make_rsrc(a.WnUt, (uint32_t)H * H * 4), rWn = make_rsrc(a.WnVt, (uint32_t)H * H * 4),
                 rWe = make_rsrc(a.WeVt, (uint32_t)H * H * 4);
    f32x4 na1[WKB], na2[WKB];
#pragma unroll
    for (int k = 0; k < WKB; ++k) { na1[k] = f4zero(); na2[k] = f4zero(); }
    const int ntiles = (a.rows + 15) / 16;
    __syncthreads();
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int rowa = tl * 16 + ix.arow;
        const int rowb = tl * 16 + ix.rs;
        f32x4 acc[WKB];
        // pass 0: dUx = dx_tmp, pass 1: dVn, pass 2: dVe
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            for (int g = ix.g0; g < HP / 4; g += 16) {
                const int acol = 4 * g;
                f32x4 v = f4zero();
                if (rowa < a.rows && acol < H) {
                    const size_t off = (size_t)rowa * H + acol;
                    if (pass == 0) {
                        const f32x4 dxo = ld4(a.dx + off), xt = ld4(a.x_tmp + off);
                        f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (dcx.on) dm = drop_mul4(dcx, drop_rowhash(dcx, (uint32_t)rowa), g);
                        const f32x4 mu = ld4(&cc[0][acol]), rsd = ld4(&cc[1][acol]), ga = ld4(&cc[2][acol]), be = ld4(&cc[3][acol]);
                        const f32x4 m1 = ld4(&cc[4][acol]), m2 = ld4(&cc[5][acol]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = (xt[r] - mu[r]) * rsd[r];
                            const float bn = xh * ga[r] + be[r];
                            const float d = dxo[r] * dm[r];
                            const float dbn = bn > 0.f ? d : 0.f;
                            v[r] = ga[r] * rsd[r] * (dbn - m1[r] - xh * m2[r]);
                        }
                        st4(a.dxtmp_out + off, v);
                    } else if (pass == 1) {
                        for (int k = 0; k < a.nit; ++k) v += ld4(a.colVn + (size_t)k * a.rows * H + off);
                        st4(a.dVn_out + off, v);
                    } else {
                        for (int k = 0; k < a.nit; ++k) v += ld4(a.colVe + (size_t)k * a.rows * H + off);
                        for (int k = 0; k < a.njt; ++k) v += ld4(a.rowpart + (size_t)k * a.rows * H + off);
                        st4(a.dVe_out + off, v);
                    }
                }
                st4(&tile[ix.arow][acol], v);
            }
            __syncthreads();
            const rsrc_t rW = pass == 0 ? rWu : (pass == 1 ? rWn : rWe);
#pragma unroll
            for (int k = 0; k < WKB; ++k) {
                const int nb = ix.w + 4 * k;
                if (nb < NB) {
                    const f32x4 o = wide_gemm<true>(tile, rW, H, NB, nb, ix.rs, ix.q);
                    acc[k] = pass == 0 ? o : acc[k] + o;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < WKB; ++k) {
            const int nb = ix.w + 4 * k;
            const int bcol = 16 * nb + 4 * ix.q;
            if (nb < NB && rowb < a.rows && bcol < H) {
                const size_t off = (size_t)rowb * H + bcol;
                f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                if (dcx.on) dm = drop_mul4(dcx, drop_rowhash(dcx, (uint32_t)rowb), bcol >> 2);
                const f32x4 res = acc[k] + ld4(a.dx + off) * dm;       // + dpre (residual branch)
                st4(a.dx + off, res);
                if (do_next) {
                    const f32x4 ntp = ld4(a.next.tmp + off);
                    const f32x4 nmu = ld4(&cc[6][bcol]), nrs = ld4(&cc[7][bcol]), nga = ld4(&cc[8][bcol]), nbe = ld4(&cc[9][bcol]);
                    f32x4 ndm = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (ndc.on) ndm = drop_mul4(ndc, drop_rowhash(ndc, (uint32_t)rowb), bcol >> 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (ntp[r] - nmu[r]) * nrs[r];
                        const float bn = xh * nga[r] + nbe[r];
                        const float d = bn > 0.f ? res[r] * ndm[r] : 0.f;
                        na1[k][r] += d;
                        na2[k][r] += d * xh;
                    }
                }
            }
        }
        __syncthreads();      // the in-place dx rows of this tile are complete before another pass 0 could read them
    }
    if (do_next) {
#pragma unroll
        for (int k = 0; k < WKB; ++k) {
            const int nb = ix.w + 4 * k;
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                const f32x4 t1 = red16x4(na1[k]), t2 = red16x4(na2[k]);
                if (ix.rs == 0 && bcol < H) {
                    st4(a.next.partial + (size_t)blockIdx.x * 2 * H + bcol, t1);
                    st4(a.next.partial + (size_t)blockIdx.x * 2 * H + H + bcol, t2);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient products (TnProblems, slab layout as k_gemm_tn2), one workgroup per row slice computing the whole
// H x H slab: 8 wavefronts, wavefront
// (wr, wc) = (w / 2, w % 2) holds the 5 x 10 blocks of 16 x 16 outputs with dW rows 80 wr .. and columns 160 wc ..
// in registers (200 accumulators), so every row of X and Y is read from memory once per product and each LDS operand
// feeds 5 or 10 MFMAs.  Rows are staged in chunks of 16 through two LDS buffers (one barrier per chunk) while the
// next chunk's loads are in flight.  grid (slices, 1, problems), 512 threads.
// ------------------------------------------------------------------------------------------------
constexpr int TNW_LT = GCN_WIDE_MAX_H + 16;        // row stride 16 mod 32: conflict-free ds_read_b32
struct TnWideLds {
    float tx[2][16][TNW_LT];
    float ty[2][16][TNW_LT];
};
__global__ __launch_bounds__(512) void k_gemm_tn_w(int H, TnProblems pr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tnw_raw[];
    TnWideLds& L = *reinterpret_cast<TnWideLds*>(tnw_raw);
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int NB = (H + 15) / 16, H4 = GCN_WIDE_MAX_H / 4;
    const int prob = blockIdx.z, slice = blockIdx.x, nslices = pr.nslices[prob];
    if (slice >= nslices) return;
    const size_t rows = pr.rows[prob];
    const float* __restrict__ X = pr.X[prob];
    const float* __restrict__ Y = pr.Y[prob];
    const size_t ntiles = (rows + 15) / 16;
    const size_t per = (ntiles + nslices - 1) / nslices;
    const size_t t0 = (size_t)slice * per, t1 = min(ntiles, t0 + per);
    f32x4 acc[5][10];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[i][k] = f4zero();
    float bsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    // staging: float4 number g of a chunk is (row g / 80, columns 4 (g % 80) ..); a thread owns g = t, t + 512, t + 1024.
    // Both operands are addressed through buffer descriptors that cover exactly this slice's rows (rows past the end
    // and columns past H read as zeros).  The six float4 of the next chunk travel in three pairs, each fetched before
    // one quarter of the chunk's MFMAs and stored to the other LDS buffer after it: 8 staging registers.
    const size_t row0 = t0 * 16;
    const uint32_t srows = t0 < t1 ? (uint32_t)min(rows - row0, (t1 - t0) * 16) : 0u;
    const rsrc_t rX = make_rsrc(X + row0 * H, srows * (uint32_t)H * 4u), rY = make_rsrc(Y + row0 * H, srows * (uint32_t)H * 4u);
    uint32_t voff[3];
    int loff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int g = t + 512 * u, r = g / H4, c = 4 * (g - r * H4);
        voff[u] = (g < 16 * H4 && c < H) ? (uint32_t)(r * H + c) * 4u : kOob;
        loff[u] = g < 16 * H4 ? r * TNW_LT + c : -1;
    }
    const uint32_t chunk_bytes = 16u * (uint32_t)H * 4u;
    auto put = [&](float (*dst)[TNW_LT], int u, const f32x4 v) {
        if (loff[u] >= 0) st4(&dst[0][0] + loff[u], v);
    };
    // blocks of this wavefront that exist at this H (wavefront-uniform, kept in scalar registers)
    const int ni = __builtin_amdgcn_readfirstlane(max(0, min(5, NB - 5 * wr)));
    const int nk = __builtin_amdgcn_readfirstlane(max(0, min(10, NB - 10 * wc)));
    // The chunk loop for a wavefront with NI x NK blocks; G: test every block against (ni, nk) at run time instead.
    auto chunk_loop = [&](auto NI_, auto NK_, auto G_) {
        constexpr int NI = decltype(NI_)::value, NK = decltype(NK_)::value;
        constexpr bool G = decltype(G_)::value;
        auto mfma_step = [&](int buf, int st) {
            float av[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                av[i] = L.tx[buf][4 * st + q][16 * (5 * wr + i) + rs];
                bsum[i] += av[i];
            }
            if constexpr (G) {
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if (k < nk) {
                        const float bv = L.ty[buf][4 * st + q][16 * (10 * wc + k) + rs];
#pragma unroll
                        for (int i = 0; i < NI; ++i)
                            if (i < ni) acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[i][k], 0, 0, 0);
                    }
                }
            } else {
                // the Y operand of block k + 1 is requested before the MFMAs of block k; the scheduling barriers keep
                // that order (left alone, the scheduler hoists every LDS read of the step and runs out of registers)
                float bv = L.ty[buf][4 * st + q][16 * (10 * wc) + rs];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    float bn = 0.f;
                    if (k + 1 < NK) bn = L.ty[buf][4 * st + q][16 * (10 * wc + k + 1) + rs];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NI; ++i) acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[i][k], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    bv = bn;
                }
            }
        };
        int buf = 0;
        for (size_t tile = t0; tile < t1; ++tile) {
            const bool more = tile + 1 < t1;
            const uint32_t nx = (uint32_t)(tile + 1 - t0) * chunk_bytes;
            f32x4 s0 = f4zero(), s1 = f4zero();
            if (more) { s0 = bld4(rX, voff[0] + nx); s1 = bld4(rX, voff[1] + nx); }
            mfma_step(buf, 0);
            if (more) { put(L.tx[buf ^ 1], 0, s0); put(L.tx[buf ^ 1], 1, s1); s0 = bld4(rX, voff[2] + nx); s1 = bld4(rY, voff[0] + nx); }
            mfma_step(buf, 1);
            if (more) { put(L.tx[buf ^ 1], 2, s0); put(L.ty[buf ^ 1], 0, s1); s0 = bld4(rY, voff[1] + nx); s1 = bld4(rY, voff[2] + nx); }
            mfma_step(buf, 2);
            if (more) { put(L.ty[buf ^ 1], 1, s0); put(L.ty[buf ^ 1], 2, s1); }
            mfma_step(buf, 3);
            __syncthreads();
            buf ^= 1;
        }
    };
    if (t0 < t1) {
#pragma unroll
        for (int u = 0; u < 3; ++u) { put(L.tx[0], u, bld4(rX, voff[u])); put(L.ty[0], u, bld4(rY, voff[u])); }
    }
    __syncthreads();
    using std::integral_constant;
    chunk_loop(integral_constant<int, 5>{}, integral_constant<int, 10>{}, integral_constant<bool, true>{});
    float* out = pr.out[prob] + (size_t)slice * pr.out_stride[prob];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int k = 0; k < 10; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = 16 * (5 * wr + i) + 4 * q + r, kc = 16 * (10 * wc + k) + rs;
                if (nn < H && kc < H) out[(size_t)nn * H + kc] = acc[i][k][r];
            }
    if (wc == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float bs = redq(bsum[i]);
            const int n = 16 * (5 * wr + i) + rs;
            if (q == 0 && n < H) out[(size_t)H * H + n] = bs;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Head (edge MLP with at most one hidden layer, loss, edge error), forward.  Same outputs as k_head_fwd;
// loss_partial has one row per workgroup.  Steps of up to WNS 16-row tiles, as in the edge kernels.
// ------------------------------------------------------------------------------------------------
// phase A of both head kernels: 16 rows of the head's input from EdgeSrc into LDS (and optionally to e_stash)
__device__ __forceinline__ void head_src_tile(float (*tile)[WLS], const EdgeSrc& S, const float (*cl)[GCN_WIDE_MAX_H],
                                              const DropCfg& dc, size_t tile0, size_t rows, int H, int HP, int C,
                                              float* e_stash, const WideIdx& ix) {
    const size_t row = tile0 + ix.arow;
    const float capv = (S.mode == 2 && row < rows) ? S.cap[row / C] : 0.f;
    for (int g = ix.g0; g < HP / 4; g += 16) {
        const int acol = 4 * g;
        f32x4 val = f4zero();
        if (row < rows && acol < H) {
            const size_t off = row * H + acol;
            if (S.mode == 0) val = ld4(S.e_in + off);
            else if (S.mode == 2) val = ld4(&cl[CL_WE][acol]) * capv;
            else {
                val = bn_relu_res4(ld4(S.prev_tmp + off), ld4(S.prev_in + off), &cl[CL_PMEAN][acol], &cl[CL_PRSTD][acol],
                                   &cl[CL_PGAMMA][acol], &cl[CL_PBETA][acol]);
                if (dc.on) val *= drop_mul4(dc, drop_rowhash(dc, (uint32_t)row), g);
            }
            if (e_stash != nullptr) st4(e_stash + off, val);
        }
        st4(&tile[ix.arow][acol], val);
    }
}
// Per row of up to WNS tiles: the two logits from per-lane partial dot products (lane (rs, q) of every wavefront holds
// a part of row rs of each tile).  Two barriers; after the second, no wavefront still reads what the partials came from.
__device__ __forceinline__ void head_row_sums(float (&p0)[WNS], float (&p1)[WNS], float (*red)[4][16][2], const WideIdx& ix) {
#pragma unroll
    for (int s = 0; s < WNS; ++s) {
        const float a = redq(p0[s]), b = redq(p1[s]);
        if (ix.q == 0) { red[s][ix.w][ix.rs][0] = a; red[s][ix.w][ix.rs][1] = b; }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < WNS; ++s) {
        p0[s] = red[s][0][ix.rs][0] + red[s][1][ix.rs][0] + red[s][2][ix.rs][0] + red[s][3][ix.rs][0];
        p1[s] = red[s][0][ix.rs][1] + red[s][1][ix.rs][1] + red[s][2][ix.rs][1] + red[s][3][ix.rs][1];
    }
    __syncthreads();
}

template <int NHID>
__global__ __launch_bounds__(256, 2) void k_head_fwd_w(
    size_t rows, int V, int C, int H, int voc, HeadParams hp, EdgeSrc src, const float* __restrict__ cap, int mask_invalid,
    const long long* __restrict__ y_edges, const float* __restrict__ edge_cw, float* __restrict__ logits,
    float* __restrict__ loss_partial, int* __restrict__ counts, const unsigned long long* __restrict__ rng, int stream_id0,
    float p, int training) {
    static_assert(NHID <= 1, "the wide head supports at most one hidden layer");
    __shared__ __attribute__((aligned(16))) float tile[16 * WNS][WLS];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GCN_WIDE_MAX_H];
    __shared__ float red[WNS][4][16][2];
    const WideIdx ix = wide_idx();
    const int NB = (H + 15) / 16, HP = 16 * NB;
    clds_fill(cl[CL_PMEAN], src.mode == 1 ? src.pmean : nullptr, H, HP); clds_fill(cl[CL_PRSTD], src.mode == 1 ? src.prstd : nullptr, H, HP);
    clds_fill(cl[CL_PGAMMA], src.mode == 1 ? src.pgamma : nullptr, H, HP); clds_fill(cl[CL_PBETA], src.mode == 1 ? src.pbeta : nullptr, H, HP);
    clds_fill(cl[CL_WE], src.mode == 2 ? src.we : nullptr, H, HP);
    wide_zero_pad(tile, NB, ix.t);
    const DropCfg sdc = drop_cfg(rng, src.drop_sid, src.drop_p, src.drop_on && src.mode == 1);
    const DropCfg hdc = drop_cfg(rng, stream_id0, p, training);
    const rsrc_t rW = make_rsrc(NHID ? hp.hidW[0] : hp.outW, (uint32_t)H * H * 4);
    const size_t ntiles = (rows + 15) / 16, nsteps = (ntiles + WNS - 1) / WNS;
    const int BC_stride = V * V * C;
    const float b2_0 = hp.outB[0], b2_1 = voc > 1 ? hp.outB[1] : 0.f;
    const float cw0 = edge_cw ? edge_cw[0] : 0.f, cw1 = edge_cw ? edge_cw[1] : 0.f;
    float acc_wn = 0.f, acc_w = 0.f;
    __syncthreads();
    for (size_t stp = blockIdx.x; stp < nsteps; stp += gridDim.x) {
        const int ns = (int)min((size_t)WNS, ntiles - stp * WNS);
#pragma unroll
        for (int s = 0; s < WNS; ++s)
            if (s < ns) head_src_tile(tile + 16 * s, src, cl, sdc, (stp * WNS + s) * 16, rows, H, HP, C, nullptr, ix);
        __syncthreads();
        float p0[WNS], p1[WNS];
#pragma unroll
        for (int s = 0; s < WNS; ++s) { p0[s] = 0.f; p1[s] = 0.f; }
#pragma unroll 1
        for (int kk = 0; kk < WKB; ++kk) {
            const int nb = ix.w + 4 * kk;
            if (nb < NB) {                                   // wavefront-uniform: every lane takes part in the MFMAs
                const int bcol = 16 * nb + 4 * ix.q;
                const bool cok = bcol < H;
                f32x4 h[WNS];
                if (NHID) {
                    wide_gemm_n<0>(tile, ns, rW, H, NB, nb, ix.rs, ix.q, h);
                } else {
#pragma unroll
                    for (int s = 0; s < WNS; ++s) h[s] = ld4(&tile[16 * s + ix.rs][bcol]);
                }
                if (cok) {
                    const f32x4 w0 = ld4(hp.outW + bcol), w1 = voc > 1 ? ld4(hp.outW + H + bcol) : f4zero();
                    const f32x4 hb = NHID ? ld4(hp.hidB[0] + bcol) : f4zero();
#pragma unroll
                    for (int s = 0; s < WNS; ++s) {
                        if (s < ns) {
                            f32x4 hv = h[s];
                            if (NHID) {
                                hv += hb;
#pragma unroll
                                for (int r = 0; r < 4; ++r) hv[r] = hv[r] > 0.f ? hv[r] : 0.f;
                                if (hdc.on) hv *= drop_mul4(hdc, drop_rowhash(hdc, (uint32_t)((stp * WNS + s) * 16 + ix.rs)), bcol >> 2);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) { p0[s] += hv[r] * w0[r]; p1[s] += hv[r] * w1[r]; }
                        }
                    }
                }
            }
        }
        head_row_sums(p0, p1, red, ix);
#pragma unroll
        for (int s = 0; s < WNS; ++s) {
            const size_t row = (stp * WNS + s) * 16 + ix.rs;
            float l0 = p0[s] + b2_0, l1 = p1[s] + b2_1;
            if (s < ns && row < rows && ix.w == 0 && ix.q == 0) {
                if (mask_invalid && cap[row / C] == 0.f) { l0 = -1e9f; l1 = -1e9f; }
                if (voc == 1) {
                    logits[row] = l0;
                } else {
                    logits[row * 2] = l0;
                    logits[row * 2 + 1] = l1;
                    if (y_edges != nullptr) {
                        const int y = (int)y_edges[row];
                        const float mx = fmaxf(l0, l1);
                        const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
                        const float w = y ? cw1 : cw0;
                        acc_wn += -w * ((y ? l1 : l0) - lse);
                        acc_w += w;
                        const int pred = l1 > l0 ? 1 : 0;
                        if (y + pred > 0) {
                            const int b = (int)(row / BC_stride), c = (int)(row % C);
                            const int BC = (int)(rows / BC_stride) * C;
                            atomicAdd(&counts[BC + b * C + c], 1);
                            if (pred == y) atomicAdd(&counts[b * C + c], 1);
                        }
                    }
                }
            }
        }
    }
    if (loss_partial != nullptr && ix.w == 0) {
        acc_wn = red16(acc_wn);
        acc_w = red16(acc_w);
        if (ix.t == 0) {
            loss_partial[(size_t)blockIdx.x * 2] = acc_wn;
            loss_partial[(size_t)blockIdx.x * 2 + 1] = acc_w;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Head backward (same outputs as k_head_bwd, NHID <= 1).  wpartial has one row per workgroup:
// [voc][H] output-layer weight gradient followed by voc bias gradients, row stride (voc*H + voc + 3) & ~3.
// hidWt: the transposed hidden weight (needed only with NHID = 1).
// Per step of up to WNS tiles: input rows to LDS; first product -> hidden pre-activations (kept in registers, 60 per
// lane); logits -> d logits; dpre written over the input rows in LDS (every wavefront is past the first product by
// then); second product -> de and the BatchNorm-backward sums of the last conv layer.  Per-channel sums in LDS.
// ------------------------------------------------------------------------------------------------
template <int NHID>
__global__ __launch_bounds__(256, 2) void k_head_bwd_w(
    size_t rows, int V, int C, int H, int voc, HeadParams hp, const float* __restrict__ hidWt, EdgeSrc src, NextBn next,
    float* __restrict__ e_stash, const float* __restrict__ cap, int mask_invalid, const long long* __restrict__ y_edges,
    const float* __restrict__ edge_cw, const float* __restrict__ loss_out, const float* __restrict__ dloss,
    const float* __restrict__ dy_pred, float* __restrict__ de, float* __restrict__ dpre_stash,
    float* __restrict__ wpartial, const unsigned long long* __restrict__ rng, int stream_id0, float p, int training, ZeroJobs z) {
    static_assert(NHID <= 1, "the wide head supports at most one hidden layer");
    __shared__ __attribute__((aligned(16))) float tile[16 * WNS][WLS];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GCN_WIDE_MAX_H];
    __shared__ __attribute__((aligned(16))) float gacc[2][GCN_WIDE_MAX_H];     // d outW rows 0 and 1
    __shared__ __attribute__((aligned(16))) float nacc[2][GCN_WIDE_MAX_H];     // BatchNorm-backward sums of the last conv layer
    __shared__ float red[WNS][4][16][2];
    zero_fill(z);
    const WideIdx ix = wide_idx();
    const int NB = (H + 15) / 16, HP = 16 * NB;
    clds_fill(cl[CL_PMEAN], src.mode == 1 ? src.pmean : nullptr, H, HP); clds_fill(cl[CL_PRSTD], src.mode == 1 ? src.prstd : nullptr, H, HP);
    clds_fill(cl[CL_PGAMMA], src.mode == 1 ? src.pgamma : nullptr, H, HP); clds_fill(cl[CL_PBETA], src.mode == 1 ? src.pbeta : nullptr, H, HP);
    clds_fill(cl[CL_WE], src.mode == 2 ? src.we : nullptr, H, HP);
    for (int k = ix.t; k < 2 * GCN_WIDE_MAX_H; k += 256) { (&gacc[0][0])[k] = 0.f; (&nacc[0][0])[k] = 0.f; }
    wide_zero_pad(tile, NB, ix.t);
    const DropCfg sdc = drop_cfg(rng, src.drop_sid, src.drop_p, src.drop_on && src.mode == 1);
    const DropCfg hdc = drop_cfg(rng, stream_id0, p, training);
    const bool do_next = next.tmp != nullptr;
    const rsrc_t rW = make_rsrc(NHID ? hp.hidW[0] : hp.outW, (uint32_t)H * H * 4);
    const rsrc_t rWt = make_rsrc(NHID ? hidWt : hp.outW, (uint32_t)H * H * 4);
    const size_t ntiles = (rows + 15) / 16, nsteps = (ntiles + WNS - 1) / WNS;
    const float b2_0 = hp.outB[0], b2_1 = voc > 1 ? hp.outB[1] : 0.f;
    const float cw0 = edge_cw ? edge_cw[0] : 0.f, cw1 = edge_cw ? edge_cw[1] : 0.f;
    const bool has_loss = (y_edges != nullptr) && voc == 2;
    const float gl = has_loss ? (dloss ? dloss[0] : 1.f) / loss_out[1] : 0.f;
    float gb0 = 0.f, gb1 = 0.f;
    __syncthreads();
    for (size_t stp = blockIdx.x; stp < nsteps; stp += gridDim.x) {
        const int ns = (int)min((size_t)WNS, ntiles - stp * WNS);
#pragma unroll
        for (int s = 0; s < WNS; ++s)
            if (s < ns) head_src_tile(tile + 16 * s, src, cl, sdc, (stp * WNS + s) * 16, rows, H, HP, C, e_stash, ix);
        __syncthreads();
        // ---- first product: hidden pre-activations (with bias), partial logits
        f32x4 pre[WNS][WKB];
        float p0[WNS], p1[WNS];
#pragma unroll
        for (int s = 0; s < WNS; ++s) { p0[s] = 0.f; p1[s] = 0.f; }
#pragma unroll 1
        for (int kk = 0; kk < WKB; ++kk) {
            const int nb = ix.w + 4 * kk;
            f32x4 o[WNS];
#pragma unroll
            for (int s = 0; s < WNS; ++s) o[s] = f4zero();
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                if (NHID) {
                    wide_gemm_n<0>(tile, ns, rW, H, NB, nb, ix.rs, ix.q, o);
                } else {
#pragma unroll
                    for (int s = 0; s < WNS; ++s) o[s] = ld4(&tile[16 * s + ix.rs][bcol]);
                }
                if (bcol < H) {
                    const f32x4 w0 = ld4(hp.outW + bcol), w1 = voc > 1 ? ld4(hp.outW + H + bcol) : f4zero();
                    const f32x4 hb = NHID ? ld4(hp.hidB[0] + bcol) : f4zero();
#pragma unroll
                    for (int s = 0; s < WNS; ++s) {
                        if (s < ns) {
                            o[s] += hb;
                            f32x4 hv = o[s];
                            if (NHID) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) hv[r] = hv[r] > 0.f ? hv[r] : 0.f;
                                if (hdc.on) hv *= drop_mul4(hdc, drop_rowhash(hdc, (uint32_t)((stp * WNS + s) * 16 + ix.rs)), bcol >> 2);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) { p0[s] += hv[r] * w0[r]; p1[s] += hv[r] * w1[r]; }
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < WNS; ++s) { pre[s][0] = o[s]; rot(pre[s]); }
        }
        head_row_sums(p0, p1, red, ix);       // its second barrier: every wavefront is done reading the input rows
        // ---- d logits of this lane's rows (every lane of a row computes the same two values)
        float d0[WNS], d1[WNS];
        bool rvalid[WNS];
#pragma unroll
        for (int s = 0; s < WNS; ++s) {
            const size_t row = (stp * WNS + s) * 16 + ix.rs;
            rvalid[s] = s < ns && row < rows;
            d0[s] = 0.f; d1[s] = 0.f;
            if (rvalid[s]) {
                const bool masked = mask_invalid && cap[row / C] == 0.f;
                if (!masked) {
                    if (voc == 1) {
                        d0[s] = dy_pred ? dy_pred[row] : 0.f;
                    } else {
                        if (has_loss) {
                            const float l0 = p0[s] + b2_0, l1 = p1[s] + b2_1;
                            const int y = (int)y_edges[row];
                            const float mx = fmaxf(l0, l1);
                            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), inv = 1.f / (e0 + e1);
                            const float w = (y ? cw1 : cw0) * gl;
                            d0[s] = w * (e0 * inv - (y == 0 ? 1.f : 0.f));
                            d1[s] = w * (e1 * inv - (y == 1 ? 1.f : 0.f));
                        }
                        if (dy_pred) { d0[s] += dy_pred[row * 2]; d1[s] += dy_pred[row * 2 + 1]; }
                    }
                }
            }
            if (ix.w == 0 && ix.q == 0) { gb0 += d0[s]; gb1 += d1[s]; }
        }
        // ---- output layer backward: d outW += d logit x h ; dh = d0 w0 + d1 w1 ; dpre = dh * relu' * dropout -> LDS
#pragma unroll
        for (int k = 0; k < WKB; ++k) {
            const int nb = ix.w + 4 * k;
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                const bool cok = bcol < H;
                const f32x4 w0 = cok ? ld4(hp.outW + bcol) : f4zero(), w1 = (cok && voc > 1) ? ld4(hp.outW + H + bcol) : f4zero();
                f32x4 t0 = f4zero(), t1 = f4zero();
#pragma unroll
                for (int s = 0; s < WNS; ++s) {
                    if (s < ns) {
                        f32x4 dp = f4zero();
                        if (cok) {
                            const size_t row = (stp * WNS + s) * 16 + ix.rs;
                            const f32x4 pr = pre[s][k];
                            f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                            if (NHID && hdc.on) dm = drop_mul4(hdc, drop_rowhash(hdc, (uint32_t)row), bcol >> 2);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float hv = NHID ? (pr[r] > 0.f ? pr[r] : 0.f) * dm[r] : pr[r];
                                const float dmul = NHID ? (pr[r] > 0.f ? dm[r] : 0.f) : 1.f;
                                t0[r] += d0[s] * hv;
                                t1[r] += d1[s] * hv;
                                dp[r] = (d0[s] * w0[r] + d1[s] * w1[r]) * dmul;
                            }
                            if (NHID && rvalid[s]) st4(dpre_stash + row * H + bcol, dp);
                        }
                        st4(&tile[16 * s + ix.rs][bcol], dp);
                    }
                }
                wide_acc_cols(gacc, bcol, ix.rs == 0 && cok, t0, t1);
            }
        }
        __syncthreads();
        // ---- hidden layer backward: de = dpre W_0 (or dpre itself without a hidden layer); statistics of the last conv layer
#pragma unroll 1
        for (int kk = 0; kk < WKB; ++kk) {
            const int nb = ix.w + 4 * kk;
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                const bool cok = bcol < H;
                f32x4 ntp[WNS];
#pragma unroll
                for (int s = 0; s < WNS; ++s) {
                    const size_t row = (stp * WNS + s) * 16 + ix.rs;
                    ntp[s] = (do_next && rvalid[s] && cok) ? ld4(next.tmp + row * H + bcol) : f4zero();
                }
                f32x4 dv[WNS];
                if (NHID) {
                    wide_gemm_n<0>(tile, ns, rWt, H, NB, nb, ix.rs, ix.q, dv);
                } else {
#pragma unroll
                    for (int s = 0; s < WNS; ++s) dv[s] = ld4(&tile[16 * s + ix.rs][bcol]);
                }
                f32x4 t1 = f4zero(), t2 = f4zero();
#pragma unroll
                for (int s = 0; s < WNS; ++s) {
                    if (rvalid[s] && cok) {
                        const size_t row = (stp * WNS + s) * 16 + ix.rs;
                        st4(de + row * H + bcol, dv[s]);
                        if (do_next) {
                            const f32x4 nmu = ld4(&cl[CL_PMEAN][bcol]), nrs = ld4(&cl[CL_PRSTD][bcol]);
                            const f32x4 nga = ld4(&cl[CL_PGAMMA][bcol]), nbe = ld4(&cl[CL_PBETA][bcol]);
                            f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                            if (sdc.on) dm = drop_mul4(sdc, drop_rowhash(sdc, (uint32_t)row), bcol >> 2);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float xh = (ntp[s][r] - nmu[r]) * nrs[r];
                                const float bn = xh * nga[r] + nbe[r];
                                const float d = bn > 0.f ? dv[s][r] * dm[r] : 0.f;
                                t1[r] += d;
                                t2[r] += d * xh;
                            }
                        }
                    }
                }
                if (do_next) wide_acc_cols(nacc, bcol, ix.rs == 0 && cok, t1, t2);
            }
        }
        __syncthreads();
    }
    const int wstride = (voc * H + voc + 3) & ~3;
    float* wp = wpartial + (size_t)blockIdx.x * wstride;
#pragma unroll
    for (int k = 0; k < WKB; ++k) {
        const int nb = ix.w + 4 * k;
        const int bcol = 16 * nb + 4 * ix.q;
        if (nb < NB && ix.rs == 0 && bcol < H) {       // the lane that accumulated these columns
            st4(wp + bcol, ld4(&gacc[0][bcol]));
            if (voc > 1) st4(wp + H + bcol, ld4(&gacc[1][bcol]));
            if (do_next) {
                st4(next.partial + (size_t)blockIdx.x * 2 * H + bcol, ld4(&nacc[0][bcol]));
                st4(next.partial + (size_t)blockIdx.x * 2 * H + H + bcol, ld4(&nacc[1][bcol]));
            }
        }
    }
    if (ix.w == 0) {
        gb0 = red16(gb0); gb1 = red16(gb1);      // lanes with q != 0 hold zeros
        if (ix.t == 0) { wp[voc * H] = gb0; if (voc > 1) wp[voc * H + 1] = gb1; }
    }
}

// ------------------------------------------------------------------------------------------------
// Edge MLPs with more than one hidden layer (MLP, gcn_layers.py:135-165): the last hidden layer and the output layer
// run in the fused head kernels above, every hidden layer before them is a launch of its own.
//   forward : hout = dropout(relu(src W^T + b))          (src: EdgeSrc, i.e. the fused apply of the last conv layer
//                                                          for the first hidden layer, a stored activation after it)
//   backward: dpre = dh * [hout > 0] * dropout multiplier  -> dpre_stash (operand of the weight gradient)
//             dh  <- dpre W                                 (in place: a step reads its rows whole before it writes)
//             and, for the first hidden layer, the BatchNorm-backward sums of the last conv layer as in k_head_bwd_w.
// Steps of up to WNS 16-row tiles.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_mlp_hidden_fwd_w(
    size_t rows, int C, int H, const float* __restrict__ W, const float* __restrict__ bias, EdgeSrc src,
    float* __restrict__ e_stash, float* __restrict__ hout, const unsigned long long* __restrict__ rng, int stream_id,
    float p, int training) {
    __shared__ __attribute__((aligned(16))) float tile[16 * WNS][WLS];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GCN_WIDE_MAX_H];
    const WideIdx ix = wide_idx();
    const int NB = (H + 15) / 16, HP = 16 * NB;
    clds_fill(cl[CL_PMEAN], src.mode == 1 ? src.pmean : nullptr, H, HP); clds_fill(cl[CL_PRSTD], src.mode == 1 ? src.prstd : nullptr, H, HP);
    clds_fill(cl[CL_PGAMMA], src.mode == 1 ? src.pgamma : nullptr, H, HP); clds_fill(cl[CL_PBETA], src.mode == 1 ? src.pbeta : nullptr, H, HP);
    clds_fill(cl[CL_WE], src.mode == 2 ? src.we : nullptr, H, HP);
    wide_zero_pad(tile, NB, ix.t);
    const DropCfg sdc = drop_cfg(rng, src.drop_sid, src.drop_p, src.drop_on && src.mode == 1);
    const DropCfg hdc = drop_cfg(rng, stream_id, p, training);
    const rsrc_t rW = make_rsrc(W, (uint32_t)H * H * 4);
    const size_t ntiles = (rows + 15) / 16, nsteps = (ntiles + WNS - 1) / WNS;
    __syncthreads();
    for (size_t stp = blockIdx.x; stp < nsteps; stp += gridDim.x) {
        const int ns = (int)min((size_t)WNS, ntiles - stp * WNS);
#pragma unroll
        for (int s = 0; s < WNS; ++s)
            if (s < ns) head_src_tile(tile + 16 * s, src, cl, sdc, (stp * WNS + s) * 16, rows, H, HP, C, e_stash, ix);
        __syncthreads();
#pragma unroll 1
        for (int kk = 0; kk < WKB; ++kk) {
            const int nb = ix.w + 4 * kk;
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                f32x4 h[WNS];
                wide_gemm_n<0>(tile, ns, rW, H, NB, nb, ix.rs, ix.q, h);
                if (bcol < H) {
                    const f32x4 hb = ld4(bias + bcol);
#pragma unroll
                    for (int s = 0; s < WNS; ++s) {
                        const size_t row = (stp * WNS + s) * 16 + ix.rs;
                        if (s < ns && row < rows) {
                            f32x4 hv = h[s] + hb;
#pragma unroll
                            for (int r = 0; r < 4; ++r) hv[r] = hv[r] > 0.f ? hv[r] : 0.f;
                            if (hdc.on) hv *= drop_mul4(hdc, drop_rowhash(hdc, (uint32_t)row), bcol >> 2);
                            st4(hout + row * H + bcol, hv);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256, 2) void k_mlp_hidden_bwd_w(
    size_t rows, int H, const float* __restrict__ Wt, const float* __restrict__ hout, float* __restrict__ dh,
    float* __restrict__ dpre_stash, NextBn next, float next_drop_p, int next_drop_on,
    const unsigned long long* __restrict__ rng, int stream_id, float p, int training) {
    __shared__ __attribute__((aligned(16))) float tile[16 * WNS][WLS];
    __shared__ __attribute__((aligned(16))) float cl[4][GCN_WIDE_MAX_H];
    __shared__ __attribute__((aligned(16))) float nacc[2][GCN_WIDE_MAX_H];     // BatchNorm-backward sums of the last conv layer
    const WideIdx ix = wide_idx();
    const int NB = (H + 15) / 16, HP = 16 * NB;
    const bool do_next = next.tmp != nullptr;
    clds_fill(cl[0], do_next ? next.mean : nullptr, H, HP); clds_fill(cl[1], do_next ? next.rstd : nullptr, H, HP);
    clds_fill(cl[2], do_next ? next.gamma : nullptr, H, HP); clds_fill(cl[3], do_next ? next.beta : nullptr, H, HP);
    for (int k = ix.t; k < 2 * GCN_WIDE_MAX_H; k += 256) (&nacc[0][0])[k] = 0.f;
    wide_zero_pad(tile, NB, ix.t);
    const DropCfg hdc = drop_cfg(rng, stream_id, p, training);
    const DropCfg ndc = drop_cfg(rng, next.drop_sid, next_drop_p, next_drop_on && do_next);
    const rsrc_t rWt = make_rsrc(Wt, (uint32_t)H * H * 4);
    const size_t ntiles = (rows + 15) / 16, nsteps = (ntiles + WNS - 1) / WNS;
    __syncthreads();
    for (size_t stp = blockIdx.x; stp < nsteps; stp += gridDim.x) {
        const int ns = (int)min((size_t)WNS, ntiles - stp * WNS);
        // ---- phase A: dpre of the step's rows -> dpre_stash and LDS
#pragma unroll
        for (int s = 0; s < WNS; ++s) {
            if (s >= ns) break;
            const size_t row = (stp * WNS + s) * 16 + ix.arow;
            const uint32_t rh = hdc.on ? drop_rowhash(hdc, (uint32_t)row) : 0u;
            for (int g = ix.g0; g < HP / 4; g += 16) {
                const int acol = 4 * g;
                f32x4 dp = f4zero();
                if (row < rows && acol < H) {
                    const size_t off = row * H + acol;
                    const f32x4 hv = ld4(hout + off), dv = ld4(dh + off);
                    f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (hdc.on) dm = drop_mul4(hdc, rh, g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[r] = hv[r] > 0.f ? dv[r] * dm[r] : 0.f;
                    st4(dpre_stash + off, dp);
                }
                st4(&tile[16 * s + ix.arow][acol], dp);
            }
        }
        __syncthreads();
        // ---- phase B: dh <- dpre W; sums for the BatchNorm backward of the conv layer below (first hidden layer only)
#pragma unroll 1
        for (int kk = 0; kk < WKB; ++kk) {
            const int nb = ix.w + 4 * kk;
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                const bool cok = bcol < H;
                f32x4 ntp[WNS];
                bool rvalid[WNS];
#pragma unroll
                for (int s = 0; s < WNS; ++s) {
                    const size_t row = (stp * WNS + s) * 16 + ix.rs;
                    rvalid[s] = s < ns && row < rows;
                    ntp[s] = (do_next && rvalid[s] && cok) ? ld4(next.tmp + row * H + bcol) : f4zero();
                }
                f32x4 dv[WNS];
                wide_gemm_n<0>(tile, ns, rWt, H, NB, nb, ix.rs, ix.q, dv);
                f32x4 t1 = f4zero(), t2 = f4zero();
#pragma unroll
                for (int s = 0; s < WNS; ++s) {
                    if (rvalid[s] && cok) {
                        const size_t row = (stp * WNS + s) * 16 + ix.rs;
                        st4(dh + row * H + bcol, dv[s]);
                        if (do_next) {
                            const f32x4 nmu = ld4(&cl[0][bcol]), nrs = ld4(&cl[1][bcol]);
                            const f32x4 nga = ld4(&cl[2][bcol]), nbe = ld4(&cl[3][bcol]);
                            f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                            if (ndc.on) dm = drop_mul4(ndc, drop_rowhash(ndc, (uint32_t)row), bcol >> 2);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float xh = (ntp[s][r] - nmu[r]) * nrs[r];
                                const float bn = xh * nga[r] + nbe[r];
                                const float d = bn > 0.f ? dv[s][r] * dm[r] : 0.f;
                                t1[r] += d;
                                t2[r] += d * xh;
                            }
                        }
                    }
                }
                if (do_next) wide_acc_cols(nacc, bcol, ix.rs == 0 && cok, t1, t2);
            }
        }
        __syncthreads();
    }
    if (do_next) {
#pragma unroll
        for (int k = 0; k < WKB; ++k) {
            const int nb = ix.w + 4 * k;
            const int bcol = 16 * nb + 4 * ix.q;
            if (nb < NB && ix.rs == 0 && bcol < H) {       // the lane that accumulated these columns
                st4(next.partial + (size_t)blockIdx.x * 2 * H + bcol, ld4(&nacc[0][bcol]));
                st4(next.partial + (size_t)blockIdx.x * 2 * H + H + bcol, ld4(&nacc[1][bcol]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Per-channel sums over rows for any H, one partial row per workgroup:
//   BNSTATS: partial[block][2][H] = sum(dbn), sum(dbn * xhat) of a BatchNorm backward (as k_bn_bwd_stats)
//   else   : partial[block][H]    = sum_row g[row][h] * scale[row / C]         (edge-embedding gradient)
// Thread (lane r, column group g): r = t / (H/4) walks rows r, r + R, ...; lanes are combined through LDS.
// ------------------------------------------------------------------------------------------------
template <bool BNSTATS>
__global__ __launch_bounds__(256) void k_colsum_w(size_t rows, int H, int C, const float* __restrict__ g, const float* __restrict__ tmp,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* __restrict__ scale, float* __restrict__ partial,
                                                  const unsigned long long* __restrict__ rng, int stream_id, float p, int training) {
    __shared__ __attribute__((aligned(16))) float acc[2][GCN_WIDE_MAX_H];
    const int G = H / 4;
    const int R = 256 / G;                       // row lanes (G <= 80)
    const int t = threadIdx.x, r = t / G, cg = t - r * G, col = 4 * cg;
    for (int k = t; k < 2 * GCN_WIDE_MAX_H; k += 256) (&acc[0][0])[k] = 0.f;
    __syncthreads();
    const DropCfg dc = drop_cfg(rng, stream_id, p, training && BNSTATS);
    const size_t per = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * per, r1 = min(rows, r0 + per);
    f32x4 a1 = f4zero(), a2 = f4zero();
    if (r < R) {
        f32x4 mu = f4zero(), rsd = f4zero(), ga = f4zero(), be = f4zero();
        if (BNSTATS) { mu = ld4(mean + col); rsd = ld4(rstd + col); ga = ld4(gamma + col); be = ld4(beta + col); }
        for (size_t row = r0 + r; row < r1; row += R) {
            const f32x4 gv = ld4(g + row * H + col);
            if (BNSTATS) {
                const f32x4 tp = ld4(tmp + row * H + col);
                f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                if (dc.on) dm = drop_mul4(dc, drop_rowhash(dc, (uint32_t)row), cg);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (tp[k] - mu[k]) * rsd[k];
                    const float bn = xh * ga[k] + be[k];
                    const float d = bn > 0.f ? gv[k] * dm[k] : 0.f;
                    a1[k] += d;
                    a2[k] += d * xh;
                }
            } else {
                a1 += gv * scale[row / C];
            }
        }
    }
    // combine the row lanes in lane order (deterministic): lane r adds in turn
    for (int rr = 0; rr < R; ++rr) {
        if (r == rr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[0][col + k] += a1[k]; acc[1][col + k] += a2[k]; }
        }
        __syncthreads();
    }
    const int nout = BNSTATS ? 2 * H : H;
    for (int k = t; k < nout; k += 256) partial[(size_t)blockIdx.x * nout + k] = acc[k / H][k % H];
}
