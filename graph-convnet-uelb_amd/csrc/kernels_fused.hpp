// Argument blocks and shared device helpers of the layer kernels (kernels_strip.hpp) and the head kernels:
// the BatchNorm-apply / ReLU / residual / dropout of layer l-1 (or the edge embedding) is folded into the
// kernel that consumes it in layer l; the math of each kernel is stated above its argument block.
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
    float* const dst[5] = {cl[CL_PMEAN], cl[CL_PRSTD], cl[CL_PGAMMA], cl[CL_PBETA], cl[CL_WE]};
    const float* const src[5] = {S.mode == 1 ? S.pmean : nullptr, S.mode == 1 ? S.prstd : nullptr, S.mode == 1 ? S.pgamma : nullptr,
                                 S.mode == 1 ? S.pbeta : nullptr, S.mode == 2 ? S.we : nullptr};
    clds_fill_batch<5>(dst, src, H, HP);
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
// Edge pass of layer l (k_edge_fwd3).  For every (b, i, c) and each of its 16-row j tiles:
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
    const u32x4* WUs;           // above 64 channels: bf16 planes of WU in fragment order (k_split_weights)
    const float* Ve; const float* Vn; const float* Ux;
    float* e_tmp; float* x_tmp; float* Aout; float* Dout; float* partial;
    const unsigned long long* rng;
    int probe;      // timing probes only: bit k drops one stream of memory traffic (results are then wrong)
};

// ------------------------------------------------------------------------------------------------
// Node side of layer l (k_node_fwd3): the three linears y[0] = Ux = x W_nU^T + b, y[1] = Vn = x W_nV^T + b
// (NodeFeatures.forward gcn_layers.py:63-64) and y[2] = Ve = x W_eV^T + b (EdgeFeatures.forward :90) of every
// node row.  With `apply` the input x is first formed from the previous layer:
// x = dropout(relu(BN(prev_tmp)) + prev_in) (gcn_layers.py:123-128) and written to x_out.
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
// Edge backward of layer l (k_edge_bwd3).  Units (b, c, i-chunk, j-tile); for each i of the chunk and the
// 16 rows (b, i, j, c):
//   dpre = de_out * dropmul ; dbn = dpre * [BN > 0] ; g = sigmoid(e_tmp)
//   det  = gamma*rstd*(dbn - m1 - xhat*m2) + qv[b,i,c] * (Vn[b,j,c] - A[b,i,c]) * g * (1 - g)
//   de_in = dpre + det * W_U                                            (in place over de)
// with qv[b,i,c] = dx_tmp / D formed on the fly from the node-side BatchNorm backward
//   dx_tmp = gamma_x*rstd_x*(dbn_x - m1_x - xhat_x*m2_x).
// Writes row sums over j (rowpart[jt][b,i,c,:]), column sums over the chunk's i (colVe[it][b,j,c,:] = sum_i det,
// colVn[it][b,j,c,:] = sum_i qv*g) and, if enabled, the BatchNorm-backward statistics of layer l-1 computed from
// de_in.  The weight gradient of edge_feat.U needs det^T e_in: the strip kernel (H <= 64) forms its workgroup's share
// of that product from the det tile it holds in LDS and the matching tile of the layer's saved edge input, and
// writes one H x H + H slab per workgroup at the end (wg_out; det itself never goes to memory); the wide kernel
// writes det (det_out) for a separate product kernel.
// ------------------------------------------------------------------------------------------------
struct EdgeBwdArgs {
    int B, V, C, H, IT, mean_agg;
    float* de;
    const float* e_tmp; const float* WUt;      // edge_feat.U: transposed copy (wide kernels) or as stored (strip kernels)
    const u32x4* WUs;                          // above 64 channels: bf16 planes of WU^T in fragment order (k_split_weights)
    const float* Vn; const float* Aagg; const float* Dn;
    const float* dx; const float* x_tmp;
    const float* stats;                                   // [4][H] mean_e, rstd_e, mean_x, rstd_x
    const float* gamma_e; const float* beta_e; const float* gamma_x; const float* beta_x;
    const float* bstats_e; const float* bstats_x;         // [2][H] each
    float* det_out; float* rowpart; float* colVe; float* colVn;
    const unsigned long long* rng; int sid_e, sid_x; float p; int training;
    NextBn next;
    const float* e_in;       // strip kernel: the layer's edge input (operand of the fused weight-gradient product)
    float* wg_out;           // strip kernel: [gridDim.x][H*H + H] partial d edge_feat.U.weight / .bias
    int jt_lo, jt_n;         // wide kernels: column tiles of this launch (k_edge_bwd_ws / k_edge_bwd_wp share the work of a layer)
    int part_off;            // wide kernels: first row of next.partial this launch writes
};

// ------------------------------------------------------------------------------------------------
// Node backward of layer l (k_node_bwd3).  Per 16-row tile of node rows:
//   dpre = dx_out * dropmul ; dx_tmp = gamma*rstd*(dbn - m1 - xhat*m2)      (BatchNormNode backward)
//   dVe  = sum_jt rowpart + sum_it colVe ; dVn = sum_it colVn ; dUx = dx_tmp
//   dx_in = dpre + dUx W_nU + dVn W_nV + dVe W_eV                            (in place over dx)
// dx_tmp, dVe, dVn are written out for the weight-gradient GEMMs; optionally the BatchNorm-backward
// statistics of layer l-1's node BatchNorm are accumulated from dx_in.
// ------------------------------------------------------------------------------------------------
// Finish of a cross-workgroup column sum inside the kernel that produced the partial rows: the last workgroup to
// arrive (device-scope counter) adds all rows in a fixed order and writes the results.  Which workgroup does the
// adding depends on timing, what it adds and in which order does not: results are bit-reproducible.
// counters: one int, zero before the launch and zero again after it.
struct LastBlockFinish {
    int* counters;
    float* dbeta; float* dgamma; float* bstats;      // BatchNorm-backward results of the layer below
    double inv_n;
};

struct NodeBwdArgs {
    int rows, H, njt, nit;
    const float* rowpart; const float* colVe; const float* colVn;
    float* dx; const float* x_tmp;
    const float* mean; const float* rstd; const float* gamma; const float* beta; const float* bstats;
    const float* WnUt; const float* WnVt; const float* WeVt;   // transposed copies (wide kernels) or as stored (strip kernels)
    float* dxtmp_out; float* dVe_out; float* dVn_out;
    const unsigned long long* rng; int sid_x; float p; int training;
    NextBn next;
    LastBlockFinish fin;       // counters != nullptr: the sums of next.partial are finished by this kernel
};

