// Fused forward kernels: the BatchNorm-apply / ReLU / residual / dropout of layer l-1 (or the edge
// embedding) is folded into the kernel that consumes it in layer l, weights and per-channel constants
// live in LDS, row reductions use DPP.
#pragma once
#include "common.hpp"

// Where a 16-row edge tile comes from.
//   mode 0 (plain) : read e_in
//   mode 1 (apply) : e = dropout(relu(BN(prev_tmp)) + prev_in)      (src/gcn/models/gcn_layers.py:122-127 of
//                    the previous layer); written to e_out if non-null
//   mode 2 (embed) : e = cap[b,i,j] * w_e                            (src/gcn/models/gcn_model.py:80);
//                    written to e_out if non-null
struct EdgeSrc {
    int mode;
    const float* e_in;
    const float* prev_tmp;
    const float* prev_in;
    const float* cap;
    float* e_out;
    // BatchNorm of the producing layer (mode 1) and its dropout stream
    const float* pmean; const float* prstd; const float* pgamma; const float* pbeta;
    const float* we;            // mode 2: edges_values_embedding.weight [H]
    int drop_sid; float drop_p; int drop_on;
};

// LDS rows of per-channel constants used by edge_src_tile
enum { CL_PMEAN = 0, CL_PRSTD = 1, CL_PGAMMA = 2, CL_PBETA = 3, CL_WE = 4, CL_ROWS = 5 };

template <int MT>
__device__ __forceinline__ void edge_src_fill_consts(float (*cl)[16 * MT], const EdgeSrc& S, int H) {
    constexpr int HP = 16 * MT;
    clds_fill(cl[CL_PMEAN], S.mode == 1 ? S.pmean : nullptr, H, HP);
    clds_fill(cl[CL_PRSTD], S.mode == 1 ? S.prstd : nullptr, H, HP);
    clds_fill(cl[CL_PGAMMA], S.mode == 1 ? S.pgamma : nullptr, H, HP);
    clds_fill(cl[CL_PBETA], S.mode == 1 ? S.pbeta : nullptr, H, HP);
    clds_fill(cl[CL_WE], S.mode == 2 ? S.we : nullptr, H, HP);
}

// Returns the tile of edge row `erow` (lane's row; !valid -> zeros).  If XH / DMR are non-null (mode 1
// only) they receive xhat = (prev_tmp - mean) * rstd and the backward multiplier dropmul * [BN > 0];
// DM receives dropmul alone.
template <int MT>
__device__ __forceinline__ Tile<MT> edge_src_tile(const EdgeSrc& S, float (*cl)[16 * MT], const DropCfg& dc,
                                                  size_t erow, bool valid, int H, int C, int q, Tile<MT>* XH,
                                                  Tile<MT>* DMR, Tile<MT>* DM) {
    const int z = lds_opaque_zero();
    Tile<MT> out;
    if (S.mode == 0) {
        out = tile_load<MT>(valid ? S.e_in + erow * H : nullptr, H, q);
    } else if (S.mode == 2) {
        const float cv = valid ? S.cap[erow / C] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) out.v[m] = clds_get(cl[CL_WE] + z, m, q) * cv;
        if (S.e_out != nullptr) tile_store(valid ? S.e_out + erow * H : nullptr, H, q, out);
    } else {
        const Tile<MT> tp = tile_load<MT>(valid ? S.prev_tmp + erow * H : nullptr, H, q);
        const Tile<MT> rs_ = tile_load<MT>(valid ? S.prev_in + erow * H : nullptr, H, q);
        const uint32_t rh = dc.on ? drop_rowhash(dc, (uint32_t)erow) : 0u;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x4 mu = clds_get(cl[CL_PMEAN] + z, m, q), rsd = clds_get(cl[CL_PRSTD] + z, m, q);
            const f32x4 ga = clds_get(cl[CL_PGAMMA] + z, m, q), be = clds_get(cl[CL_PBETA] + z, m, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = (tp.v[m][r] - mu[r]) * rsd[r];
                const float bn = xh * ga[r] + be[r];
                const float dm = dc.on ? drop_mul(dc, rh, 16 * m + 4 * q + r) : 1.f;
                float v = ((bn > 0.f ? bn : 0.f) + rs_.v[m][r]) * dm;
                out.v[m][r] = valid ? v : 0.f;
                if (XH) XH->v[m][r] = xh;
                if (DMR) DMR->v[m][r] = (bn > 0.f && valid) ? dm : 0.f;
                if (DM) DM->v[m][r] = valid ? dm : 0.f;
            }
        }
        if (S.e_out != nullptr) tile_store(valid ? S.e_out + erow * H : nullptr, H, q, out);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// Edge pass of layer l.  One wavefront per (b, i, c) walking the 16-row j tiles.  Per tile:
//   e_in  = <EdgeSrc>                                           (fused apply / embedding / plain read)
//   e_tmp = e_in W_U^T + b_U + Ve[b,i,c] + Ve[b,j,c]            (EdgeFeatures.forward, gcn_layers.py:89-93)
//   gate  = sigmoid(e_tmp) ; num += gate * Vn[b,j,c] ; den += gate   (:120, NodeFeatures.forward :65-69)
// then x_tmp[b,i,c] = Ux + num / (1e-20 + den) ("mean") or Ux + num ("sum") and the per-workgroup sums of
// e_tmp, e_tmp^2, x_tmp, x_tmp^2 per channel for the two BatchNorms: partial [nblocks][4][H].
// ------------------------------------------------------------------------------------------------
struct EdgeFwdArgs {
    int B, V, C, H, mean_agg;
    EdgeSrc src;
    const float* WU; const float* bU;
    const float* Ve; const float* Vn; const float* Ux;
    float* e_tmp; float* x_tmp; float* Aout; float* Dout; float* partial;
    const unsigned long long* rng;
    int probe;      // timing probes only: bit k drops one stream of memory traffic (results are then wrong)
};

template <int MT>
__global__ __launch_bounds__(GCN_BLOCK, 2) void k_edge_fwd2(EdgeFwdArgs a) {
    constexpr int HP = 16 * MT;
    __shared__ f32x4 wl[MT * MT * 64];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS + 1][HP];      // + b_U
    __shared__ __attribute__((aligned(16))) float red[GCN_WAVES_PER_BLOCK][4][HP];
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * GCN_WAVES_PER_BLOCK + wv;
    const int nwaves = gridDim.x * GCN_WAVES_PER_BLOCK;
    const int H = a.H, V = a.V, C = a.C;
    wlds_fill<MT, false>(wl, a.WU, H);
    edge_src_fill_consts<MT>(cl, a.src, H);
    clds_fill(cl[CL_ROWS], a.bU, H, HP);
    for (int t = threadIdx.x; t < GCN_WAVES_PER_BLOCK * 4 * HP; t += blockDim.x) (&red[0][0][0])[t] = 0.f;
    __syncthreads();
    const DropCfg dc = drop_cfg(a.rng, a.src.drop_sid, a.src.drop_p, a.src.drop_on && a.src.mode == 1);
    const int units = a.B * V * C;
    const int njt = (V + 15) / 16;
    for (int u = wave; u < units; u += nwaves) {
        const int b = u / (V * C);
        const int rem = u - b * V * C;
        const int i = rem / C;
        const int c = rem - i * C;
        const size_t nrow_i = (size_t)(b * V + i) * C + c;
        ChanVec<MT> vei = chan_load<MT>(a.Ve + nrow_i * H, H, q);
        const int zu = lds_opaque_zero();
#pragma unroll
        for (int m = 0; m < MT; ++m) vei.v[m] += clds_get(cl[CL_ROWS] + zu, m, q);
        Tile<MT> num, den, s1, s2;
        tile_zero(num); tile_zero(den); tile_zero(s1); tile_zero(s2);
        for (int jt = 0; jt < njt; ++jt) {
            const int j = jt * 16 + rs;
            const bool valid = j < V;
            const float vf = valid ? 1.f : 0.f;
            const size_t erow = ((size_t)(b * V + i) * V + j) * C + c;
            const size_t nrow_j = (size_t)(b * V + j) * C + c;
            const Tile<MT> vej = tile_load<MT>(valid ? a.Ve + nrow_j * H : nullptr, H, q);
            const Tile<MT> vnj = tile_load<MT>(valid ? a.Vn + nrow_j * H : nullptr, H, q);
            const Tile<MT> in = edge_src_tile<MT>(a.src, cl, dc, erow, valid, H, C, q, nullptr, nullptr, nullptr);
            Tile<MT> et = tile_gemm_lds<MT>(in, wl + lds_opaque_zero(), lane);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                et.v[m] += vei.v[m] + vej.v[m];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = et.v[m][r];
                    const float g = sigmoidf_(x) * vf;
                    num.v[m][r] += g * vnj.v[m][r];
                    den.v[m][r] += g;
                    s1.v[m][r] += x * vf;
                    s2.v[m][r] += x * x * vf;
                }
            }
            tile_store(valid ? a.e_tmp + erow * H : nullptr, H, q, et);
        }
        tile_red16(num); tile_red16(den); tile_red16(s1); tile_red16(s2);
        if (rs == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int col = 16 * m + 4 * q;
                if (col < H) {
                    const f32x4 ux = *reinterpret_cast<const f32x4*>(a.Ux + nrow_i * H + col);
                    f32x4 D, A;
                    if (a.mean_agg) {
                        D = den.v[m] + 1e-20f;
                        A = num.v[m] / D;
                    } else {
                        D = f32x4{1.f, 1.f, 1.f, 1.f};
                        A = num.v[m];
                    }
                    const f32x4 xt = ux + A;
                    *reinterpret_cast<f32x4*>(a.x_tmp + nrow_i * H + col) = xt;
                    *reinterpret_cast<f32x4*>(a.Aout + nrow_i * H + col) = A;
                    *reinterpret_cast<f32x4*>(a.Dout + nrow_i * H + col) = D;
                    // this lane is the only writer of these LDS words of its wavefront's slab
                    f32x4* r0 = reinterpret_cast<f32x4*>(&red[wv][0][col]);
                    f32x4* r1 = reinterpret_cast<f32x4*>(&red[wv][1][col]);
                    f32x4* r2 = reinterpret_cast<f32x4*>(&red[wv][2][col]);
                    f32x4* r3 = reinterpret_cast<f32x4*>(&red[wv][3][col]);
                    *r0 += s1.v[m]; *r1 += s2.v[m]; *r2 += xt; *r3 += xt * xt;
                }
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * H; t += blockDim.x) {
        const int which = t / H, col = t - which * H;
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < GCN_WAVES_PER_BLOCK; ++w2) v += red[w2][which][col];
        a.partial[(size_t)blockIdx.x * 4 * H + t] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Node side of layer l.  blockIdx.y selects one of the three linears (0: Ux = x W_nU^T + b, 1: Vn =
// x W_nV^T + b, NodeFeatures.forward gcn_layers.py:63-64; 2: Ve = x W_eV^T + b, EdgeFeatures.forward :90);
// one wavefront per 16-row tile.  With `apply` the input x is first formed from the previous layer:
// x = dropout(relu(BN(prev_tmp)) + prev_in) (gcn_layers.py:123-128) and written to x_out by the
// blockIdx.y == 0 workgroups.
// ------------------------------------------------------------------------------------------------
struct NodeFwdArgs {
    int rows, H, apply;
    const float* x_in;                       // apply == 0
    const float* prev_tmp; const float* prev_in; float* x_out;
    const float* pmean; const float* prstd; const float* pgamma; const float* pbeta;
    int drop_sid; float drop_p; int drop_on;
    const float* W[3]; const float* bias[3]; float* y[3];
    const unsigned long long* rng;
};

template <int MT>
__global__ __launch_bounds__(GCN_BLOCK) void k_node_fwd2(NodeFwdArgs a) {
    constexpr int HP = 16 * MT;
    __shared__ f32x4 wl[MT * MT * 64];
    __shared__ __attribute__((aligned(16))) float cl[5][HP];
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4;
    const int which = blockIdx.y;
    const int H = a.H;
    wlds_fill<MT, false>(wl, a.W[which], H);
    clds_fill(cl[0], a.apply ? a.pmean : nullptr, H, HP);
    clds_fill(cl[1], a.apply ? a.prstd : nullptr, H, HP);
    clds_fill(cl[2], a.apply ? a.pgamma : nullptr, H, HP);
    clds_fill(cl[3], a.apply ? a.pbeta : nullptr, H, HP);
    clds_fill(cl[4], a.bias[which], H, HP);
    __syncthreads();
    const DropCfg dc = drop_cfg(a.rng, a.drop_sid, a.drop_p, a.drop_on && a.apply);
    const int wave = blockIdx.x * GCN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * GCN_WAVES_PER_BLOCK;
    const int ntiles = (a.rows + 15) / 16;
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int row = tile * 16 + rs;
        const bool valid = row < a.rows;
        const int zt = lds_opaque_zero();
        Tile<MT> in;
        if (!a.apply) {
            in = tile_load<MT>(valid ? a.x_in + (size_t)row * H : nullptr, H, q);
        } else {
            const Tile<MT> tp = tile_load<MT>(valid ? a.prev_tmp + (size_t)row * H : nullptr, H, q);
            const Tile<MT> rsd_ = tile_load<MT>(valid ? a.prev_in + (size_t)row * H : nullptr, H, q);
            const uint32_t rh = dc.on ? drop_rowhash(dc, (uint32_t)row) : 0u;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 mu = clds_get(cl[0] + zt, m, q), rsd = clds_get(cl[1] + zt, m, q);
                const f32x4 ga = clds_get(cl[2] + zt, m, q), be = clds_get(cl[3] + zt, m, q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float bn = (tp.v[m][r] - mu[r]) * rsd[r] * ga[r] + be[r];
                    float v = (bn > 0.f ? bn : 0.f) + rsd_.v[m][r];
                    if (dc.on) v *= drop_mul(dc, rh, 16 * m + 4 * q + r);
                    in.v[m][r] = valid ? v : 0.f;
                }
            }
            if (which == 0 && a.x_out != nullptr) tile_store(valid ? a.x_out + (size_t)row * H : nullptr, H, q, in);
        }
        Tile<MT> out = tile_gemm_lds<MT>(in, wl + zt, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) out.v[m] += clds_get(cl[4] + zt, m, q);
        tile_store(valid ? a.y[which] + (size_t)row * H : nullptr, H, q, out);
    }
}

// ================================================================================================
// Backward
// ================================================================================================

// dst[m] = transpose(src[m]) for up to 64 H x H matrices, one workgroup each.  Run once per backward so
// that every "X * W" product can read W^T with the same coalesced fragment loads as the forward.
struct TransposeList {
    const float* src[64];
    float* dst[64];
};
__global__ __launch_bounds__(GCN_BLOCK) void k_transpose_weights(TransposeList tl, int H) {
    const float* __restrict__ s = tl.src[blockIdx.x];
    float* __restrict__ d = tl.dst[blockIdx.x];
    for (int t = threadIdx.x; t < H * H; t += blockDim.x) {
        const int r = t / H, c = t - r * H;
        d[(size_t)c * H + r] = s[t];
    }
}

// BatchNorm-backward statistics of the layer below, fused into the kernel that produces that layer's
// output gradient: given the gradient tile g (= d out of layer l-1) it adds, per channel,
// sum(dbn) and sum(dbn * xhat) with dbn = g * dropmul * [BN > 0] into the wavefront's LDS slab.
struct NextBn {
    const float* tmp;        // pre-BN activations of layer l-1 (nullptr = disabled)
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int drop_sid;
    float* partial;          // [nblocks][2][H]
};

template <int MT>
__device__ __forceinline__ void next_bn_accumulate(const NextBn& nb, float (*ncl)[16 * MT], float (*nred)[16 * MT],
                                                   const DropCfg& ndc, const Tile<MT>& g, size_t row, bool valid, int H,
                                                   int rs, int q) {
    const Tile<MT> tp = tile_load<MT>(valid ? nb.tmp + row * H : nullptr, H, q);
    const uint32_t rh = ndc.on ? drop_rowhash(ndc, (uint32_t)row) : 0u;
    const int z = lds_opaque_zero();
    Tile<MT> a1, a2;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const f32x4 mu = clds_get(ncl[0] + z, m, q), rsd = clds_get(ncl[1] + z, m, q);
        const f32x4 ga = clds_get(ncl[2] + z, m, q), be = clds_get(ncl[3] + z, m, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (tp.v[m][r] - mu[r]) * rsd[r];
            const float bn = xh * ga[r] + be[r];
            float d = g.v[m][r];
            if (ndc.on) d *= drop_mul(ndc, rh, 16 * m + 4 * q + r);
            d = (bn > 0.f && valid) ? d : 0.f;
            a1.v[m][r] = d;
            a2.v[m][r] = d * xh;
        }
    }
    tile_red16(a1);
    tile_red16(a2);
    if (rs == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            *reinterpret_cast<f32x4*>(&nred[0][16 * m + 4 * q]) += a1.v[m];
            *reinterpret_cast<f32x4*>(&nred[1][16 * m + 4 * q]) += a2.v[m];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Edge backward of layer l.  One wavefront per (b, c, i-chunk, j-tile); for each i of the chunk and the
// 16 rows (b, i, j, c):
//   dpre = de_out * dropmul ; dbn = dpre * [BN > 0] ; g = sigmoid(e_tmp)
//   det  = gamma*rstd*(dbn - m1 - xhat*m2) + qv[b,i,c] * (Vn[b,j,c] - A[b,i,c]) * g * (1 - g)
//   de_in = dpre + det * W_U                                            (in place over de)
// with qv[b,i,c] = dx_tmp / D formed on the fly from the node-side BatchNorm backward
//   dx_tmp = gamma_x*rstd_x*(dbn_x - m1_x - xhat_x*m2_x).
// Writes det (for the weight-gradient GEMM), row sums over j (rowpart[jt][b,i,c,:]), column sums over the
// chunk's i (colVe[it][b,j,c,:] = sum_i det, colVn[it][b,j,c,:] = sum_i qv*g) and, if enabled, the
// BatchNorm-backward statistics of layer l-1 computed from de_in.
// ------------------------------------------------------------------------------------------------
struct EdgeBwdArgs {
    int B, V, C, H, IT, mean_agg;
    float* de;
    const float* e_tmp; const float* WUt;
    const float* Vn; const float* Aagg; const float* Dn;
    const float* dx; const float* x_tmp;
    const float* stats;                                   // [4][H] mean_e, rstd_e, mean_x, rstd_x
    const float* gamma_e; const float* beta_e; const float* gamma_x; const float* beta_x;
    const float* bstats_e; const float* bstats_x;         // [2][H] each
    float* det_out; float* rowpart; float* colVe; float* colVn;
    const unsigned long long* rng; int sid_e, sid_x; float p; int training;
    NextBn next;
};

template <int MT>
__global__ __launch_bounds__(GCN_BLOCK, 2) void k_edge_bwd_main2(EdgeBwdArgs a) {
    constexpr int HP = 16 * MT;
    __shared__ f32x4 wl[MT * MT * 64];
    __shared__ __attribute__((aligned(16))) float ce[6][HP];     // edges: mean, rstd, gamma, beta, m1, m2
    __shared__ __attribute__((aligned(16))) float cx[6][HP];     // nodes: same
    __shared__ __attribute__((aligned(16))) float ncl[4][HP];    // layer l-1 edges: mean, rstd, gamma, beta
    __shared__ __attribute__((aligned(16))) float nred[GCN_WAVES_PER_BLOCK][2][HP];
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * GCN_WAVES_PER_BLOCK + wv;
    const int nwaves = gridDim.x * GCN_WAVES_PER_BLOCK;
    const int H = a.H, V = a.V, C = a.C, IT = a.IT;
    const bool do_next = a.next.tmp != nullptr;
    wlds_fill<MT, false>(wl, a.WUt, H);
    clds_fill(ce[0], a.stats, H, HP); clds_fill(ce[1], a.stats + H, H, HP);
    clds_fill(ce[2], a.gamma_e, H, HP); clds_fill(ce[3], a.beta_e, H, HP);
    clds_fill(ce[4], a.bstats_e, H, HP); clds_fill(ce[5], a.bstats_e + H, H, HP);
    clds_fill(cx[0], a.stats + 2 * H, H, HP); clds_fill(cx[1], a.stats + 3 * H, H, HP);
    clds_fill(cx[2], a.gamma_x, H, HP); clds_fill(cx[3], a.beta_x, H, HP);
    clds_fill(cx[4], a.bstats_x, H, HP); clds_fill(cx[5], a.bstats_x + H, H, HP);
    clds_fill(ncl[0], do_next ? a.next.mean : nullptr, H, HP); clds_fill(ncl[1], do_next ? a.next.rstd : nullptr, H, HP);
    clds_fill(ncl[2], do_next ? a.next.gamma : nullptr, H, HP); clds_fill(ncl[3], do_next ? a.next.beta : nullptr, H, HP);
    for (int t = threadIdx.x; t < GCN_WAVES_PER_BLOCK * 2 * HP; t += blockDim.x) (&nred[0][0][0])[t] = 0.f;
    __syncthreads();
    const DropCfg dce = drop_cfg(a.rng, a.sid_e, a.p, a.training);
    const DropCfg dcx = drop_cfg(a.rng, a.sid_x, a.p, a.training);
    const DropCfg ndc = drop_cfg(a.rng, a.next.drop_sid, a.p, a.training && do_next);
    const int njt = (V + 15) / 16;
    const int nit = (V + IT - 1) / IT;
    const int units = a.B * nit * njt * C;
    const size_t Rn = (size_t)a.B * V * C;
    for (int u = wave; u < units; u += nwaves) {
        int t = u;
        const int c = t % C; t /= C;
        const int jt = t % njt; t /= njt;
        const int it = t % nit;
        const int b = t / nit;
        const int j = jt * 16 + rs;
        const bool valid = j < V;
        const size_t nrow_j = (size_t)(b * V + j) * C + c;
        const Tile<MT> vnj = tile_load<MT>(valid ? a.Vn + nrow_j * H : nullptr, H, q);
        Tile<MT> cve, cvn;
        tile_zero(cve); tile_zero(cvn);
        const int i_end = min(V, (it + 1) * IT);
        for (int i = it * IT; i < i_end; ++i) {
            const size_t nrow_i = (size_t)(b * V + i) * C + c;
            // node-side BatchNorm backward for row (b, i, c): qv = dx_tmp / D, ai = A (or 0 for "sum")
            ChanVec<MT> qi, ai;
            {
                const ChanVec<MT> dxn = chan_load<MT>(a.dx + nrow_i * H, H, q);
                const ChanVec<MT> xt = chan_load<MT>(a.x_tmp + nrow_i * H, H, q);
                const ChanVec<MT> Dv = chan_load<MT>(a.mean_agg ? a.Dn + nrow_i * H : nullptr, H, q);
                ai = chan_load<MT>(a.mean_agg ? a.Aagg + nrow_i * H : nullptr, H, q);
                const uint32_t rhx = dcx.on ? drop_rowhash(dcx, (uint32_t)nrow_i) : 0u;
                const int z = lds_opaque_zero();
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 mu = clds_get(cx[0] + z, m, q), rsd = clds_get(cx[1] + z, m, q), ga = clds_get(cx[2] + z, m, q);
                    const f32x4 be = clds_get(cx[3] + z, m, q), m1 = clds_get(cx[4] + z, m, q), m2 = clds_get(cx[5] + z, m, q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (xt.v[m][r] - mu[r]) * rsd[r];
                        const float bn = xh * ga[r] + be[r];
                        float d = dxn.v[m][r];
                        if (dcx.on) d *= drop_mul(dcx, rhx, 16 * m + 4 * q + r);
                        const float dbn = bn > 0.f ? d : 0.f;
                        const float dxt = ga[r] * rsd[r] * (dbn - m1[r] - xh * m2[r]);
                        const float dv = Dv.v[m][r];       // 0 only in the zero-padded columns
                        qi.v[m][r] = a.mean_agg ? (dv != 0.f ? dxt / dv : 0.f) : dxt;
                    }
                }
            }
            const size_t erow = ((size_t)(b * V + i) * V + j) * C + c;
            Tile<MT> dn = tile_load<MT>(valid ? a.de + erow * H : nullptr, H, q);
            const Tile<MT> tp = tile_load<MT>(valid ? a.e_tmp + erow * H : nullptr, H, q);
            const uint32_t rh = dce.on ? drop_rowhash(dce, (uint32_t)erow) : 0u;
            const int ze = lds_opaque_zero();
            Tile<MT> det;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 mu = clds_get(ce[0] + ze, m, q), rsd = clds_get(ce[1] + ze, m, q), ga = clds_get(ce[2] + ze, m, q);
                const f32x4 be = clds_get(ce[3] + ze, m, q), m1 = clds_get(ce[4] + ze, m, q), m2 = clds_get(ce[5] + ze, m, q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = tp.v[m][r];
                    const float xh = (x - mu[r]) * rsd[r];
                    const float bn = xh * ga[r] + be[r];
                    float d = dn.v[m][r];
                    if (dce.on) d *= drop_mul(dce, rh, 16 * m + 4 * q + r);
                    dn.v[m][r] = d;                                   // dpre
                    const float dbn = bn > 0.f ? d : 0.f;
                    const float g = sigmoidf_(x);
                    float v = ga[r] * rsd[r] * (dbn - m1[r] - xh * m2[r]);
                    v += qi.v[m][r] * (vnj.v[m][r] - ai.v[m][r]) * g * (1.f - g);
                    v = valid ? v : 0.f;
                    det.v[m][r] = v;
                    cve.v[m][r] += v;
                    cvn.v[m][r] += valid ? qi.v[m][r] * g : 0.f;
                }
            }
            tile_store(valid ? a.det_out + erow * H : nullptr, H, q, det);
            Tile<MT> din = tile_gemm_lds<MT>(det, wl + ze, lane);
#pragma unroll
            for (int m = 0; m < MT; ++m) din.v[m] += dn.v[m];
            tile_store(valid ? a.de + erow * H : nullptr, H, q, din);
            if (do_next) next_bn_accumulate<MT>(a.next, ncl, nred[wv], ndc, din, erow, valid, H, rs, q);
            tile_red16(det);   // row sums over the tile's j
            if (rs == 0) {
                float* rp = a.rowpart + ((size_t)jt * Rn + nrow_i) * H;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int col = 16 * m + 4 * q;
                    if (col < H) *reinterpret_cast<f32x4*>(rp + col) = det.v[m];
                }
            }
        }
        tile_store(valid ? a.colVe + ((size_t)it * Rn + nrow_j) * H : nullptr, H, q, cve);
        tile_store(valid ? a.colVn + ((size_t)it * Rn + nrow_j) * H : nullptr, H, q, cvn);
    }
    if (do_next) {
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * H; t += blockDim.x) {
            const int which = t / H, col = t - which * H;
            float v = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < GCN_WAVES_PER_BLOCK; ++w2) v += nred[w2][which][col];
            a.next.partial[(size_t)blockIdx.x * 2 * H + t] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Node backward of layer l.  One wavefront per 16-row tile of node rows:
//   dpre = dx_out * dropmul ; dx_tmp = gamma*rstd*(dbn - m1 - xhat*m2)      (BatchNormNode backward)
//   dVe  = sum_jt rowpart + sum_it colVe ; dVn = sum_it colVn ; dUx = dx_tmp
//   dx_in = dpre + dUx W_nU + dVn W_nV + dVe W_eV                            (in place over dx)
// dx_tmp, dVe, dVn are written out for the weight-gradient GEMMs; optionally the BatchNorm-backward
// statistics of layer l-1's node BatchNorm are accumulated from dx_in.
// ------------------------------------------------------------------------------------------------
struct NodeBwdArgs {
    int rows, H, njt, nit;
    const float* rowpart; const float* colVe; const float* colVn;
    float* dx; const float* x_tmp;
    const float* mean; const float* rstd; const float* gamma; const float* beta; const float* bstats;
    const float* WnUt; const float* WnVt; const float* WeVt;
    float* dxtmp_out; float* dVe_out; float* dVn_out;
    const unsigned long long* rng; int sid_x; float p; int training;
    NextBn next;
};

template <int MT>
__global__ __launch_bounds__(GCN_BLOCK) void k_node_bwd2(NodeBwdArgs a) {
    constexpr int HP = 16 * MT;
    __shared__ f32x4 wl[3][MT * MT * 64];
    __shared__ __attribute__((aligned(16))) float cx[6][HP];
    __shared__ __attribute__((aligned(16))) float ncl[4][HP];
    __shared__ __attribute__((aligned(16))) float nred[GCN_WAVES_PER_BLOCK][2][HP];
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
    const int H = a.H;
    const bool do_next = a.next.tmp != nullptr;
    wlds_fill<MT, false>(wl[0], a.WnUt, H);
    wlds_fill<MT, false>(wl[1], a.WnVt, H);
    wlds_fill<MT, false>(wl[2], a.WeVt, H);
    clds_fill(cx[0], a.mean, H, HP); clds_fill(cx[1], a.rstd, H, HP); clds_fill(cx[2], a.gamma, H, HP);
    clds_fill(cx[3], a.beta, H, HP); clds_fill(cx[4], a.bstats, H, HP); clds_fill(cx[5], a.bstats + H, H, HP);
    clds_fill(ncl[0], do_next ? a.next.mean : nullptr, H, HP); clds_fill(ncl[1], do_next ? a.next.rstd : nullptr, H, HP);
    clds_fill(ncl[2], do_next ? a.next.gamma : nullptr, H, HP); clds_fill(ncl[3], do_next ? a.next.beta : nullptr, H, HP);
    for (int t = threadIdx.x; t < GCN_WAVES_PER_BLOCK * 2 * HP; t += blockDim.x) (&nred[0][0][0])[t] = 0.f;
    __syncthreads();
    const DropCfg dcx = drop_cfg(a.rng, a.sid_x, a.p, a.training);
    const DropCfg ndc = drop_cfg(a.rng, a.next.drop_sid, a.p, a.training && do_next);
    const int wave = blockIdx.x * GCN_WAVES_PER_BLOCK + wv;
    const int nwaves = gridDim.x * GCN_WAVES_PER_BLOCK;
    const int ntiles = (a.rows + 15) / 16;
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int row = tile * 16 + rs;
        const bool valid = row < a.rows;
        Tile<MT> acc = tile_load<MT>(valid ? a.dx + (size_t)row * H : nullptr, H, q);      // d x_out
        const Tile<MT> xt = tile_load<MT>(valid ? a.x_tmp + (size_t)row * H : nullptr, H, q);
        const uint32_t rh = dcx.on ? drop_rowhash(dcx, (uint32_t)row) : 0u;
        const int z = lds_opaque_zero();
        Tile<MT> dux;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x4 mu = clds_get(cx[0] + z, m, q), rsd = clds_get(cx[1] + z, m, q), ga = clds_get(cx[2] + z, m, q);
            const f32x4 be = clds_get(cx[3] + z, m, q), m1 = clds_get(cx[4] + z, m, q), m2 = clds_get(cx[5] + z, m, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = (xt.v[m][r] - mu[r]) * rsd[r];
                const float bn = xh * ga[r] + be[r];
                float d = acc.v[m][r];
                if (dcx.on) d *= drop_mul(dcx, rh, 16 * m + 4 * q + r);
                acc.v[m][r] = d;                                           // dpre (residual branch)
                const float dbn = bn > 0.f ? d : 0.f;
                dux.v[m][r] = valid ? ga[r] * rsd[r] * (dbn - m1[r] - xh * m2[r]) : 0.f;
            }
        }
        tile_store(valid ? a.dxtmp_out + (size_t)row * H : nullptr, H, q, dux);
        {
            const Tile<MT> o = tile_gemm_lds<MT>(dux, wl[0] + z, lane);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc.v[m] += o.v[m];
        }
        Tile<MT> dv;
        tile_zero(dv);
        for (int k = 0; k < a.nit; ++k) {
            const Tile<MT> t2 = tile_load<MT>(valid ? a.colVn + ((size_t)k * a.rows + row) * H : nullptr, H, q);
#pragma unroll
            for (int m = 0; m < MT; ++m) dv.v[m] += t2.v[m];
        }
        tile_store(valid ? a.dVn_out + (size_t)row * H : nullptr, H, q, dv);
        {
            const Tile<MT> o = tile_gemm_lds<MT>(dv, wl[1] + z, lane);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc.v[m] += o.v[m];
        }
        tile_zero(dv);
        for (int k = 0; k < a.njt; ++k) {
            const Tile<MT> t2 = tile_load<MT>(valid ? a.rowpart + ((size_t)k * a.rows + row) * H : nullptr, H, q);
#pragma unroll
            for (int m = 0; m < MT; ++m) dv.v[m] += t2.v[m];
        }
        for (int k = 0; k < a.nit; ++k) {
            const Tile<MT> t2 = tile_load<MT>(valid ? a.colVe + ((size_t)k * a.rows + row) * H : nullptr, H, q);
#pragma unroll
            for (int m = 0; m < MT; ++m) dv.v[m] += t2.v[m];
        }
        tile_store(valid ? a.dVe_out + (size_t)row * H : nullptr, H, q, dv);
        {
            const Tile<MT> o = tile_gemm_lds<MT>(dv, wl[2] + z, lane);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc.v[m] += o.v[m];
        }
        tile_store(valid ? a.dx + (size_t)row * H : nullptr, H, q, acc);
        if (do_next) next_bn_accumulate<MT>(a.next, ncl, nred[wv], ndc, acc, (size_t)row, valid, H, rs, q);
    }
    if (do_next) {
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * H; t += blockDim.x) {
            const int which = t / H, col = t - which * H;
            float v = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < GCN_WAVES_PER_BLOCK; ++w2) v += nred[w2][which][col];
            a.next.partial[(size_t)blockIdx.x * 2 * H + t] = v;
        }
    }
}
