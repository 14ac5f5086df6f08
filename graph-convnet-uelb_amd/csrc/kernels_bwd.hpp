// Backward kernels of the residual gated graph-conv stack (gfx950).
// The reference has no hand-written backward: these kernels restate what autograd derives from
// src/gcn/models/gcn_layers.py:117-129 and src/gcn/models/model_utils.py:51-78.
#pragma once
#include "common.hpp"
#include "kernels_fwd.hpp"

// ------------------------------------------------------------------------------------------------
// BatchNorm backward, reduction pass.  For out = dropout(relu(BN(tmp)) + resid):
//   dpre = dnew * dropmask ; dbn = dpre * [BN(tmp) > 0] ; xhat = (tmp - mean) * rstd
// accumulates per channel sum(dbn) and sum(dbn * xhat) into partial[nblocks][2][H].
// Used for the edge tensor (R_e rows) and the node tensor (R_n rows).
// ------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(GCN_BLOCK) void k_bn_bwd_stats(
    size_t rows, int H, const float* __restrict__ dnew, const float* __restrict__ tmp,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ partial,
    const unsigned long long* __restrict__ rng, int stream_id, float p, int training) {
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * GCN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * GCN_WAVES_PER_BLOCK;
    const DropCfg dc = drop_cfg(rng, stream_id, p, training);
    const ChanVec<MT> mu = chan_load<MT>(mean, H, q), rsd = chan_load<MT>(rstd, H, q);
    const ChanVec<MT> ga = chan_load<MT>(gamma, H, q), be = chan_load<MT>(beta, H, q);
    Tile<MT> a1, a2;
    tile_zero(a1); tile_zero(a2);
    const size_t ntiles = (rows + 15) / 16;
    for (size_t tile = wave; tile < ntiles; tile += nwaves) {
        const size_t row = tile * 16 + rs;
        const bool valid = row < rows;
        const Tile<MT> dn = tile_load<MT>(valid ? dnew + row * H : nullptr, H, q);
        const Tile<MT> tp = tile_load<MT>(valid ? tmp + row * H : nullptr, H, q);
        const uint32_t rh = dc.on ? drop_rowhash(dc, (uint32_t)row) : 0u;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = (tp.v[m][r] - mu.v[m][r]) * rsd.v[m][r];
                const float bn = xh * ga.v[m][r] + be.v[m][r];
                float d = dn.v[m][r];
                if (dc.on) d *= drop_mul(dc, rh, 16 * m + 4 * q + r);
                d = (bn > 0.f && valid) ? d : 0.f;
                a1.v[m][r] += d;
                a2.v[m][r] += d * xh;
            }
    }
    tile_red16(a1);
    tile_red16(a2);
    // workgroup partial = sum of its 4 wavefronts, in wavefront order (LDS), one slab per workgroup
    __shared__ float red[GCN_WAVES_PER_BLOCK][2][16 * MT];
    const int wv = threadIdx.x >> 6;
    if (rs == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            *reinterpret_cast<f32x4*>(&red[wv][0][16 * m + 4 * q]) = a1.v[m];
            *reinterpret_cast<f32x4*>(&red[wv][1][16 * m + 4 * q]) = a2.v[m];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * H; t += blockDim.x) {
        const int which = t / H, col = t - which * H;
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < GCN_WAVES_PER_BLOCK; ++w2) v += red[w2][which][col];
        partial[(size_t)blockIdx.x * 2 * H + t] = v;
    }
}

// partial [nP][2][H] -> bstats [2][H] = mean(dbn), mean(dbn*xhat); dgamma = sum(dbn*xhat), dbeta = sum(dbn).
// grid = (H/4, 2): blockIdx.y = 0 edges, 1 nodes.
struct BnBwdFin {
    const float* partial[2];
    int nP[2];
    double n[2];
    float* bstats[2];
    float* dgamma[2];
    float* dbeta[2];
};
// ------------------------------------------------------------------------------------------------
// Weight-gradient GEMM with the reduction over rows: dW[n][k] = sum_row X[row][n] * Y[row][k] and
// db[n] = sum_row X[row][n], split over gridDim.x row slices.  Up to 3 (X, Y) problems per launch
// (blockIdx.z).  Each wavefront owns one 16-wide block of n (block m = blockIdx.y*4 + wave) and all
// k; operands are read straight from global memory in MFMA order (A[i = n][kk = row], B[kk = row][j = k]).
// partial layout: [slice][prob][H*H + H].
// ------------------------------------------------------------------------------------------------
struct TnProblems {
    const float* X[4];
    const float* Y[4];
    // used by k_gemm_tn2 only: rows, row slices and output of each problem; slice s of problem p writes the
    // slab out[p] + s * out_stride[p]
    size_t rows[4];
    int nslices[4];
    float* out[4];
    size_t out_stride[4];
};

// Sum of nP partial slabs of ncols floats (ncols % 4 == 0) -> up to 8 destination tensors: column range
// [off[d], off[d+1]) of the slab goes to dst[d].  One workgroup per 16 columns: thread (pg, cg) adds
// slabs pg, pg + 64, ... of float4 column group cg in fp64, then a fixed LDS tree over the 64 pg's, so
// the result does not depend on scheduling.
struct ReduceDst {
    float* dst[8];
    float* dst2[8];      // optional second copy of the same columns, multiplied by scale2 (nullptr = none)
    double scale2;
    int off[9];
    int n;
};
struct ReduceJob {
    const float* partial;
    int nP, ncols;
    ReduceDst rd;
};
struct ReduceJobs {
    ReduceJob job[4];
    int n;
};
// Columns per workgroup of a slab reduction: few partial rows -> wide column blocks (fewer, fuller workgroups),
// many rows -> narrow blocks with more threads along the rows.  Host and device use the same rule.
__host__ __device__ inline int reduce_cols(int nP) { return nP <= 64 ? 64 : (nP <= 256 ? 32 : 16); }
// Sum of the nP partial rows of one column block of one job, in a fixed order, accumulated in fp64.
// Needs all 256 threads of the workgroup.
__device__ __forceinline__ void reduce_slab_body(const ReduceJob& J, int colblock, double* lds /* [256 * 4] */) {
    const float* __restrict__ partial = J.partial;
    const int nP = J.nP, ncols = J.ncols;
    const int bcols = reduce_cols(nP), ncg = bcols >> 2, npg = 256 / ncg;     // column groups of 4, row groups
    const int cg = threadIdx.x % ncg, pg = threadIdx.x / ncg;
    const int col = colblock * bcols + cg * 4;
    if (colblock * bcols >= ncols) return;          // uniform per workgroup
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (col < ncols) {
        const float* src = partial + col;
        int p = pg;
        for (; p + 3 * npg < nP; p += 4 * npg) {       // four loads in flight, added in row order
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)(p + k * npg) * ncols);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0 += v[k][0]; a1 += v[k][1]; a2 += v[k][2]; a3 += v[k][3]; }
        }
        for (; p < nP; p += npg) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * ncols);
            a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
        }
    }
    double* my = lds + pg * bcols + cg * 4;
    my[0] = a0; my[1] = a1; my[2] = a2; my[3] = a3;
    __syncthreads();
    for (int sft = npg >> 1; sft > 0; sft >>= 1) {
        if (pg < sft) {
#pragma unroll
            for (int k = 0; k < 4; ++k) my[k] += my[sft * bcols + k];
        }
        __syncthreads();
    }
    if (pg == 0 && col < ncols) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cc = col + k;
            for (int d = 0; d < J.rd.n; ++d)
                if (cc >= J.rd.off[d] && cc < J.rd.off[d + 1]) {
                    J.rd.dst[d][cc - J.rd.off[d]] = (float)my[k];
                    if (J.rd.dst2[d] != nullptr) J.rd.dst2[d][cc - J.rd.off[d]] = (float)(my[k] * J.rd.scale2);
                    break;
                }
        }
    }
}
// grid = (largest number of column blocks of a job, number of jobs)
__global__ __launch_bounds__(GCN_BLOCK) void k_reduce_slabs(ReduceJobs jobs) {
    __shared__ double lds[64 * 16];
    reduce_slab_body(jobs.job[blockIdx.y], blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------------
// Head backward.  Recomputes the hidden activations of the edge MLP, forms
//   dlogit[v] = gl * w[y] * (softmax_v - [v == y]) / sum_w  (+ dy_pred[row][v])      (loss_edges backward)
// with zero gradient on masked (-1e9) edges, walks back through the NHID hidden layers, writes
// de = d loss / d e_L, the pre-activation gradients dpre_k (for the weight-gradient GEMM) and, for
// k >= 1, the layer inputs h_{k-1}; accumulates d outW / d outB per wavefront.
// wpartial layout: [nwaves][wstride], wstride = voc*H + voc rounded up to a multiple of 4 floats.
// ------------------------------------------------------------------------------------------------
template <int MT, int NHID>
__global__ __launch_bounds__(GCN_BLOCK) void k_head_bwd(
    size_t rows, int V, int C, int H, int voc, HeadParams hp, EdgeSrc src, NextBn next, float* __restrict__ e_stash,
    const float* __restrict__ cap, int mask_invalid, const long long* __restrict__ y_edges,
    const float* __restrict__ edge_cw, const float* __restrict__ loss_out /* [1] = sum w */,
    const float* __restrict__ dloss, const float* __restrict__ dy_pred,
    float* __restrict__ de, float* __restrict__ dpre_stash /* [NHID][rows*H] */,
    float* __restrict__ h_stash /* [NHID-1][rows*H] */, float* __restrict__ wpartial,
    const unsigned long long* __restrict__ rng, int stream_id0, float p, int training, ZeroJobs z) {
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][16 * MT];
    __shared__ __attribute__((aligned(16))) float nred[GCN_WAVES_PER_BLOCK][2][16 * MT];
    zero_fill(z);
    // with one hidden layer its weight fragments (as stored, and transposed for the way back) live in LDS
    constexpr bool WLDS = NHID == 1;
    __shared__ f32x4 wl[WLDS ? 2 : 1][WLDS ? MT * MT * 64 : 1];
    edge_src_fill_consts<MT>(cl, src, H);
    if constexpr (WLDS) { wlds_fill<MT, false>(wl[0], hp.hidW[0], H); wlds_fill<MT, true>(wl[1], hp.hidW[0], H); }
    for (int t = threadIdx.x; t < GCN_WAVES_PER_BLOCK * 2 * 16 * MT; t += blockDim.x) (&nred[0][0][0])[t] = 0.f;
    __syncthreads();
    const DropCfg sdc = drop_cfg(rng, src.drop_sid, src.drop_p, src.drop_on && src.mode == 1);
    const bool do_next = next.tmp != nullptr;
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * GCN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * GCN_WAVES_PER_BLOCK;
    const size_t ntiles = (rows + 15) / 16;
    ChanVec<MT> w2[2];
    w2[0] = chan_load<MT>(hp.outW, H, q);
    w2[1] = chan_load<MT>(voc > 1 ? hp.outW + H : nullptr, H, q);
    const float b2_0 = hp.outB[0], b2_1 = voc > 1 ? hp.outB[1] : 0.f;
    const float cw0 = edge_cw ? edge_cw[0] : 0.f, cw1 = edge_cw ? edge_cw[1] : 0.f;
    const bool has_loss = (y_edges != nullptr) && voc == 2;
    const float gl = has_loss ? (dloss ? dloss[0] : 1.f) / loss_out[1] : 0.f;
    Tile<MT> dw2a, dw2b;          // per-lane partial d outW rows 0 / 1
    tile_zero(dw2a); tile_zero(dw2b);
    float db2a = 0.f, db2b = 0.f;
    for (size_t tile = wave; tile < ntiles; tile += nwaves) {
        const size_t row = tile * 16 + rs;
        const bool valid = row < rows;
        Tile<MT> h[NHID + 1];
        Tile<MT> dmul[NHID > 0 ? NHID : 1];   // relu' * dropout multiplier per hidden layer
        h[0] = edge_src_tile<MT>(src, cl, sdc, row, valid, H, C, q, nullptr, nullptr, nullptr);
        if (e_stash != nullptr) tile_store(valid ? e_stash + row * H : nullptr, H, q, h[0]);   // operand of the d W_0 GEMM
#pragma unroll
        for (int k = 0; k < NHID; ++k) {
            Tile<MT> o;
            if constexpr (WLDS) {
                o = tile_gemm_lds<MT>(h[k], wl[0] + lds_opaque_zero(), lane);
            } else {
                WFrag<MT> wf;
                wfrag_load_nt(wf, hp.hidW[k], H, rs, q);
                o = tile_gemm(h[k], wf);
            }
            const ChanVec<MT> bv = chan_load<MT>(hp.hidB[k], H, q);
            const DropCfg dc = drop_cfg(rng, stream_id0 + k, p, training);
            const uint32_t rh = dc.on ? drop_rowhash(dc, (uint32_t)row) : 0u;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = o.v[m][r] + bv.v[m][r];
                    float mul = v > 0.f ? 1.f : 0.f;
                    if (dc.on) mul *= drop_mul(dc, rh, 16 * m + 4 * q + r);
                    dmul[k].v[m][r] = mul;
                    h[k + 1].v[m][r] = v * mul;
                }
            if (k + 1 < NHID) tile_store(valid ? h_stash + ((size_t)k * rows + row) * H : nullptr, H, q, h[k + 1]);
        }
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                l0 += h[NHID].v[m][r] * w2[0].v[m][r];
                l1 += h[NHID].v[m][r] * w2[1].v[m][r];
            }
        l0 = redq(l0) + b2_0;
        l1 = redq(l1) + b2_1;
        float d0 = 0.f, d1 = 0.f;
        if (valid) {
            const bool masked = mask_invalid && cap[row / C] == 0.f;
            if (!masked) {
                if (has_loss) {
                    const int y = (int)y_edges[row];
                    const float mx = fmaxf(l0, l1);
                    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
                    const float inv = 1.f / (e0 + e1);
                    const float w = (y ? cw1 : cw0) * gl;
                    d0 = w * (e0 * inv - (y == 0 ? 1.f : 0.f));
                    d1 = w * (e1 * inv - (y == 1 ? 1.f : 0.f));
                }
                if (dy_pred != nullptr) {
                    if (voc == 1) d0 += dy_pred[row];
                    else { d0 += dy_pred[row * 2]; d1 += dy_pred[row * 2 + 1]; }
                }
            }
        }
        if (q == 0) { db2a += d0; db2b += d1; }
        Tile<MT> dh;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dw2a.v[m][r] += d0 * h[NHID].v[m][r];
                dw2b.v[m][r] += d1 * h[NHID].v[m][r];
                dh.v[m][r] = d0 * w2[0].v[m][r] + d1 * w2[1].v[m][r];
            }
#pragma unroll
        for (int k = NHID - 1; k >= 0; --k) {
#pragma unroll
            for (int m = 0; m < MT; ++m) dh.v[m] *= dmul[k].v[m];
            tile_store(valid ? dpre_stash + ((size_t)k * rows + row) * H : nullptr, H, q, dh);
            if constexpr (WLDS) {
                dh = tile_gemm_lds<MT>(dh, wl[1] + lds_opaque_zero(), lane);
            } else {
                WFrag<MT> wf;
                wfrag_load_tr(wf, hp.hidW[k], H, rs, q);
                dh = tile_gemm(dh, wf);
            }
        }
        tile_store(valid ? de + row * H : nullptr, H, q, dh);
        // BatchNorm-backward statistics of the last conv layer, whose apply was fused into this kernel's input
        if (do_next) next_bn_accumulate<MT>(next, cl, nred[threadIdx.x >> 6], sdc, dh, row, valid, H, rs, q);
    }
    if (do_next) {
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * H; t += blockDim.x) {
            const int which = t / H, col = t - which * H;
            float v = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < GCN_WAVES_PER_BLOCK; ++w2) v += nred[w2][which][col];
            next.partial[(size_t)blockIdx.x * 2 * H + t] = v;
        }
    }
    tile_red16(dw2a);
    tile_red16(dw2b);
    db2a = red16(db2a);
    db2b = red16(db2b);
    float* wp = wpartial + (size_t)wave * (size_t)((voc * H + voc + 3) & ~3);
    if (rs == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int col = 16 * m + 4 * q;
            if (col < H) {
                *reinterpret_cast<f32x4*>(wp + col) = dw2a.v[m];
                if (voc > 1) *reinterpret_cast<f32x4*>(wp + H + col) = dw2b.v[m];
            }
        }
    }
    if (lane == 0) {
        wp[voc * H] = db2a;
        if (voc > 1) wp[voc * H + 1] = db2b;
    }
}

// ------------------------------------------------------------------------------------------------
// Embedding backward.  d w_e[h] = sum_rows cap[row / C] * de0[row][h]   (per-wavefront partials [nwaves][H])
// ------------------------------------------------------------------------------------------------
template <int MT>
__device__ __forceinline__ void embed_bwd_edges_body(size_t rows, int C, int H, const float* __restrict__ cap,
                                                     const float* __restrict__ de0, float* __restrict__ partial, int bid, int nblk) {
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4;
    const int wave = bid * GCN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int nwaves = nblk * GCN_WAVES_PER_BLOCK;
    Tile<MT> acc;
    tile_zero(acc);
    const size_t ntiles = (rows + 15) / 16;
    for (size_t tile = wave; tile < ntiles; tile += nwaves) {
        const size_t row = tile * 16 + rs;
        const bool valid = row < rows;
        const float cv = valid ? cap[row / C] : 0.f;
        if (cv != 0.f) {
            const Tile<MT> d = tile_load<MT>(de0 + row * H, H, q);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc.v[m] += d.v[m] * cv;
        }
    }
    tile_red16(acc);
    // the four wavefronts of the workgroup are added in a fixed order: one partial row per workgroup
    __shared__ __attribute__((aligned(16))) float wsum[GCN_WAVES_PER_BLOCK][16 * MT];
    if (rs == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>(&wsum[threadIdx.x >> 6][16 * m + 4 * q]) = acc.v[m];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float s = wsum[0][c];
        for (int w = 1; w < GCN_WAVES_PER_BLOCK; ++w) s += wsum[w][c];
        partial[(size_t)bid * H + c] = s;
    }
}

template <int MT>
__global__ __launch_bounds__(GCN_BLOCK) void k_embed_bwd_edges(size_t rows, int C, int H, const float* __restrict__ cap,
                                                              const float* __restrict__ de0, float* __restrict__ partial) {
    embed_bwd_edges_body<MT>(rows, C, H, cap, de0, partial, (int)blockIdx.x, (int)gridDim.x);
}

// Node embedding backward in two small launches.
//   k_embed_bwd_rowsum: per (b, i): g1[bi][h] = sum_c dx0[bi,c,h], g2[bi][h] = sum_c demand[b,c] * dx0[bi,c,H/2+h]
//                       and cnt[bi][v] = #{c : x_nodes[bi,c] == v}
//   k_embed_bwd_nodes : dEmb[v][h] = (1/C) * sum_bi cnt[bi][v] * g1[bi][h];  d w_c[h] = sum_bi g2[bi][h]
// (x0[b,i,c,:H/2] is the mean over c' of Emb[x_nodes[b,i,c']], gcn_model.py:78-84).  Fixed summation
// order, so reproducible.
struct EmbedRowsumArgs {
    int V, C, H, n_emb;
    const long long* x_nodes; const float* demand; const float* dx0;
    float* g; int* cnt;
};
// one workgroup per (b, i); four independent chains of loads per sum (c = k, k + 4, ... added in that order, then
// the four chains in order)
__device__ __forceinline__ void embed_bwd_rowsum_body(const EmbedRowsumArgs& a, int bi) {
    const int C = a.C, H = a.H, b = bi / a.V, hh = H / 2;
    const float* d0 = a.dx0 + (size_t)bi * C * H;
    const float* dem = a.demand + (size_t)b * C;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        int c = 0;
        for (; c + 4 <= C; c += 4) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = d0[(size_t)(c + k) * H + h] * (h < hh ? 1.f : dem[c + k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += v[k];
        }
        for (int k = 0; c < C; ++c, ++k) s[k] += d0[(size_t)c * H + h] * (h < hh ? 1.f : dem[c]);
        a.g[(size_t)bi * H + h] = (s[0] + s[1]) + (s[2] + s[3]);
    }
    const long long* xn = a.x_nodes + (size_t)bi * C;
    for (int v = threadIdx.x; v < a.n_emb; v += blockDim.x) {
        int n = 0;
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            long long idx = xn[c];
            idx = idx < 0 ? 0 : (idx >= a.n_emb ? a.n_emb - 1 : idx);
            n += (idx == v);
        }
        a.cnt[(size_t)bi * a.n_emb + v] = n;
    }
}
__global__ void k_embed_bwd_rowsum(EmbedRowsumArgs a) { embed_bwd_rowsum_body(a, (int)blockIdx.x); }
__device__ __forceinline__ void embed_bwd_nodes_body(int nbi, int C, int H, int n_emb, const float* __restrict__ g,
                                                     const int* __restrict__ cnt, float* __restrict__ demb,
                                                     float* __restrict__ dwc, int v, double* lds_d /* [blockDim.x] */) {
    const int hh = H / 2;
    const int parts = blockDim.x / hh;
    const int h = threadIdx.x % hh, part = threadIdx.x / hh;
    double acc = 0;
    if (part < parts) {
        for (int bi = part; bi < nbi; bi += parts) {
            if (v < n_emb) {
                const int n = cnt[(size_t)bi * n_emb + v];
                if (n) acc += (double)n * (double)g[(size_t)bi * H + h];
            } else {
                acc += (double)g[(size_t)bi * H + hh + h];
            }
        }
    }
    lds_d[threadIdx.x] = (part < parts) ? acc : 0.0;
    __syncthreads();
    if (part == 0) {
        double sacc = 0;
        for (int k = 0; k < parts; ++k) sacc += lds_d[k * hh + h];
        if (v < n_emb) demb[(size_t)v * hh + h] = (float)(sacc / (double)C);
        else dwc[h] = (float)sacc;
    }
}
__global__ __launch_bounds__(1024) void k_embed_bwd_nodes(int nbi, int C, int H, int n_emb, const float* __restrict__ g,
                                                              const int* __restrict__ cnt, float* __restrict__ demb,
                                                              float* __restrict__ dwc) {
    extern __shared__ double lds_d[];     // [blockDim.x]
    embed_bwd_nodes_body(nbi, C, H, n_emb, g, cnt, demb, dwc, (int)blockIdx.x, lds_d);
}
// The same next to slab reductions, which need nothing from it: workgroups 0 .. n_emb finish the node embedding,
// the others run reduce jobs (first 256 threads; the rest of such a workgroup leaves at once).
// rseg[k] .. rseg[k + 1]: workgroups of reduce job k, counted from n_emb + 1.
struct FinalMap {
    int rseg[5];
};
__global__ __launch_bounds__(1024) void k_bwd_final(int nbi, int C, int H, int n_emb, const float* __restrict__ g,
                                                        const int* __restrict__ cnt, float* __restrict__ demb,
                                                        float* __restrict__ dwc, ReduceJobs jobs, FinalMap map) {
    extern __shared__ double lds_d[];     // [max(blockDim.x, 1024)]
    const int bid = blockIdx.x;
    if (bid <= n_emb) { embed_bwd_nodes_body(nbi, C, H, n_emb, g, cnt, demb, dwc, bid, lds_d); return; }
    if (threadIdx.x >= GCN_BLOCK) return;
    const int r = bid - (n_emb + 1);
    int k = 0;
    while (k < 3 && r >= map.rseg[k + 1]) ++k;
    reduce_slab_body(jobs.job[k], r - map.rseg[k], lds_d);
}

// ------------------------------------------------------------------------------------------------
// Stand-alone loss_edges / edge_error on logits [rows][2] (for callers that hold predictions already;
// inside the model both are fused into k_head_fwd).  loss_edges: src/gcn/models/model_utils.py:51-78,
// edge_error: :255-308.  One thread per row, per-workgroup partial sums (fixed order), integer counters.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GCN_BLOCK) void k_loss_edges_fwd(size_t rows, int VVC, int C, const float* __restrict__ logits,
                                                             const long long* __restrict__ y_edges,
                                                             const float* __restrict__ edge_cw,
                                                             float* __restrict__ loss_partial /* [nblocks][2], nullable */,
                                                             int* __restrict__ counts /* [2][B*C], nullable */, int BC) {
    __shared__ double lds[GCN_BLOCK * 2];
    double a = 0, w = 0;
    const float cw0 = edge_cw ? edge_cw[0] : 0.f, cw1 = edge_cw ? edge_cw[1] : 0.f;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        const float l0 = logits[r * 2], l1 = logits[r * 2 + 1];
        const int y = (int)y_edges[r];
        if (loss_partial != nullptr) {
            const float mx = fmaxf(l0, l1);
            const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
            const float cw = y ? cw1 : cw0;
            a += (double)(-cw * ((y ? l1 : l0) - lse));
            w += (double)cw;
        }
        if (counts != nullptr) {
            const int pred = l1 > l0 ? 1 : 0;
            if (y + pred > 0) {
                const int b = (int)(r / VVC), c = (int)(r % C);
                atomicAdd(&counts[BC + b * C + c], 1);
                if (pred == y) atomicAdd(&counts[b * C + c], 1);
            }
        }
    }
    if (loss_partial == nullptr) return;
    lds[threadIdx.x * 2] = a; lds[threadIdx.x * 2 + 1] = w;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { lds[threadIdx.x * 2] += lds[(threadIdx.x + s) * 2]; lds[threadIdx.x * 2 + 1] += lds[(threadIdx.x + s) * 2 + 1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { loss_partial[(size_t)blockIdx.x * 2] = (float)lds[0]; loss_partial[(size_t)blockIdx.x * 2 + 1] = (float)lds[1]; }
}

// dlogits[row][v] = dloss * w[y] * (softmax_v - [v == y]) / sum_w
__global__ void k_loss_edges_bwd(size_t rows, const float* __restrict__ logits, const long long* __restrict__ y_edges,
                                 const float* __restrict__ edge_cw, const float* __restrict__ loss_out,
                                 const float* __restrict__ dloss, float* __restrict__ dlogits) {
    const float gl = (dloss ? dloss[0] : 1.f) / loss_out[1];
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        const float l0 = logits[r * 2], l1 = logits[r * 2 + 1];
        const int y = (int)y_edges[r];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float inv = 1.f / (e0 + e1);
        const float w = (y ? edge_cw[1] : edge_cw[0]) * gl;
        dlogits[r * 2] = w * (e0 * inv - (y == 0 ? 1.f : 0.f));
        dlogits[r * 2 + 1] = w * (e1 * inv - (y == 1 ? 1.f : 0.f));
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-tensor Adam (torch.optim.Adam arithmetic, see gcn_adam_step): blockIdx.y = tensor of the chunk,
// grid-stride over its elements.
// ------------------------------------------------------------------------------------------------
constexpr int ADAM_CHUNK = 64;
struct AdamChunk {
    float* p[ADAM_CHUNK];
    const float* g[ADAM_CHUNK];
    float* m[ADAM_CHUNK];
    float* v[ADAM_CHUNK];
    long long numel[ADAM_CHUNK];
    int n;
};
__global__ __launch_bounds__(256) void k_adam(AdamChunk c, float step_size, float b1, float b2, float eps, float wd,
                                              float inv_sqrt_bc2) {
    const int k = blockIdx.y;
    float* __restrict__ p = c.p[k];
    const float* __restrict__ g = c.g[k];
    float* __restrict__ m = c.m[k];
    float* __restrict__ v = c.v[k];
    const long long n = c.numel[k];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] + wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}
