// Forward kernels of the residual gated graph-conv stack (gfx950).
// Shapes: R_e = B*V*V*C edge rows laid out [B][V_i][V_j][C][H], R_n = B*V*C node rows [B][V][C][H].
#pragma once
#include "common.hpp"
#include "kernels_fused.hpp"

// one workgroup per (b, i); thread h < H/2 handles channel h of both halves
__global__ void k_embed_nodes(int V, int C, int H, int n_emb, const long long* __restrict__ x_nodes,
                              const float* __restrict__ demand, const float* __restrict__ emb,
                              const float* __restrict__ wc, float* __restrict__ x0, ZeroJobs z,
                              unsigned long long* __restrict__ rng_advance) {
    zero_fill(z);
    // first kernel of a forward pass and not a reader of the dropout state: the place to start the next step's masks
    if (rng_advance != nullptr && blockIdx.x == 0 && threadIdx.x == 0) rng_advance[1] += 1ull;
    const int bi = blockIdx.x;          // b*V + i
    const int b = bi / V;
    const int hh = H / 2;
    for (int h = threadIdx.x; h < hh; h += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            long long idx = x_nodes[(size_t)bi * C + c];
            idx = idx < 0 ? 0 : (idx >= n_emb ? n_emb - 1 : idx);
            s += emb[(size_t)idx * hh + h];
        }
        const float mean = s / (float)C;
        const float w = wc[h];
        for (int c = 0; c < C; ++c) {
            float* row = x0 + ((size_t)bi * C + c) * H;
            row[h] = mean;
            row[hh + h] = demand[(size_t)b * C + c] * w;
        }
    }
}

// partial [nP][4][H] (sum e, sum e^2, sum x, sum x^2) -> stats [4][H] = mean_e, rstd_e, mean_x, rstd_x.
// Biased variance, eps 1e-5 (BatchNorm{2,3}d defaults, gcn_layers.py:10,30).  grid = 2 * H/4: one workgroup per
// four columns of one BatchNorm; sum and sum of squares are added in the same pass, in fp64 and in a fixed
// order: thread t adds rows t, t + 256, ..., a butterfly adds the lanes of each wavefront, thread 0 the wavefronts.
__global__ __launch_bounds__(GCN_BLOCK) void k_bn_stats_finalize(const float* __restrict__ partial, int nP, int H,
                                                                double n_e, double n_x, float* __restrict__ stats) {
    __shared__ double lds[GCN_WAVES_PER_BLOCK][8];
    const int h4 = H / 4;
    const int which = blockIdx.x / h4;            // 0 = edges, 1 = nodes
    const int col = (blockIdx.x - which * h4) * 4;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* p1 = partial + (2 * which) * H + col;
    for (int p = threadIdx.x; p < nP; p += GCN_BLOCK) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(p1 + (size_t)p * 4 * H);
        const f32x4 v = *reinterpret_cast<const f32x4*>(p1 + (size_t)p * 4 * H + H);
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] += u[k]; a[4 + k] += v[k]; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += __shfl_xor(a[k], off, 64);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) lds[threadIdx.x >> 6][k] = a[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        double s1 = 0, s2 = 0;
        for (int w = 0; w < GCN_WAVES_PER_BLOCK; ++w) { s1 += lds[w][k]; s2 += lds[w][4 + k]; }
        const double n = which == 0 ? n_e : n_x;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0) var = 0;
        stats[(2 * which) * H + col + k] = (float)mean;
        stats[(2 * which + 1) * H + col + k] = (float)(1.0 / sqrt(var + (double)GCN_BN_EPS));
    }
}

// ------------------------------------------------------------------------------------------------
// out = dropout(relu(BN(tmp)) + resid)   (gcn_layers.py:122-128), rows x H, elementwise float4.
// ------------------------------------------------------------------------------------------------
__global__ void k_bn_apply(size_t rows, int H, const float* __restrict__ tmp, const float* __restrict__ resid,
                           const float* __restrict__ mean, const float* __restrict__ rstd,
                           const float* __restrict__ gamma, const float* __restrict__ beta,
                           float* __restrict__ out, const unsigned long long* __restrict__ rng, int stream_id,
                           float p, int training) {
    const DropCfg dc = drop_cfg(rng, stream_id, p, training);
    const int h4 = H / 4;
    const size_t n4 = rows * (size_t)h4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / h4;
        const int col = (int)(i - row * h4) * 4;
        const f32x4 t = *reinterpret_cast<const f32x4*>(tmp + row * H + col);
        const f32x4 r = *reinterpret_cast<const f32x4*>(resid + row * H + col);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + col);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + col);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + col);
        const f32x4 be = *reinterpret_cast<const f32x4*>(beta + col);
        f32x4 y = (t - mu) * rs * ga + be;
        const uint32_t rh = dc.on ? drop_rowhash(dc, (uint32_t)row) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = (y[k] > 0.f ? y[k] : 0.f) + r[k];
            if (dc.on) v *= drop_mul(dc, rh, col + k);
            y[k] = v;
        }
        *reinterpret_cast<f32x4*>(out + row * H + col) = y;
    }
}

// mask[row][h] = 1 iff the ReLU behind a BatchNorm is open at that element, decided by the expression every
// kernel of the library uses for it: (tmp - mean) * rstd * gamma + beta > 0 (a subtraction, a product, one fused
// multiply-add).  For tests that hand the library's own ReLU decisions to a reference implementation.
__global__ void k_bn_relu_pattern(size_t rows, int H, const float* __restrict__ tmp, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, unsigned char* __restrict__ mask) {
    const int h4 = H / 4;
    const size_t n4 = rows * (size_t)h4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / h4;
        const int col = (int)(i - row * h4) * 4;
        const f32x4 t = *reinterpret_cast<const f32x4*>(tmp + row * H + col);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + col);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + col);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + col);
        const f32x4 be = *reinterpret_cast<const f32x4*>(beta + col);
        uchar4 m;
        m.x = ((t[0] - mu[0]) * rs[0] * ga[0] + be[0]) > 0.f ? 1 : 0;
        m.y = ((t[1] - mu[1]) * rs[1] * ga[1] + be[1]) > 0.f ? 1 : 0;
        m.z = ((t[2] - mu[2]) * rs[2] * ga[2] + be[2]) > 0.f ? 1 : 0;
        m.w = ((t[3] - mu[3]) * rs[3] * ga[3] + be[3]) > 0.f ? 1 : 0;
        *reinterpret_cast<uchar4*>(mask + row * H + col) = m;
    }
}

// ------------------------------------------------------------------------------------------------
// Edge MLP head + loss partials.  MLP.forward (gcn_layers.py:167-176) with NHID = mlp_layers - 1
// hidden Linear(H,H)+ReLU+Dropout blocks, then Linear(H, voc); optional -1e9 masking of zero-capacity
// edges (gcn_model.py:97-110); loss_edges (model_utils.py:51-78): log_softmax over the 2 classes and
// class-weighted NLL partial sums (sum w*nll, sum w); edge_error counters (model_utils.py:255-308)
// per (b, c): #(pred == y and mask), #mask with mask = (y + pred) > 0.
// The input tile comes through EdgeSrc (plain read, or the fused apply of the last conv layer).
// hidW/hidB: NHID pointers each.  loss_partial: [nwaves][2].  counts: [2][B*C] int32 (zeroed by caller).
// ------------------------------------------------------------------------------------------------
struct HeadParams {
    const float* hidW[3];
    const float* hidB[3];
    const float* outW;   // [voc][H]
    const float* outB;   // [voc]
};

template <int MT, int NHID>
__global__ __launch_bounds__(GCN_BLOCK) void k_head_fwd(
    size_t rows, int V, int C, int H, int voc, HeadParams hp, EdgeSrc src,
    const float* __restrict__ cap, int mask_invalid, const long long* __restrict__ y_edges,
    const float* __restrict__ edge_cw, float* __restrict__ logits, float* __restrict__ loss_partial,
    int* __restrict__ counts, const unsigned long long* __restrict__ rng, int stream_id0, float p, int training) {
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][16 * MT];
    // with one hidden layer its weight fragments live in LDS; deeper heads read them from memory per tile
    constexpr bool WLDS = NHID == 1;
    __shared__ f32x4 wl[WLDS ? MT * MT * 64 : 1];
    edge_src_fill_consts<MT>(cl, src, H);
    if constexpr (WLDS) wlds_fill<MT, false>(wl, hp.hidW[0], H);
    __syncthreads();
    const DropCfg sdc = drop_cfg(rng, src.drop_sid, src.drop_p, src.drop_on && src.mode == 1);
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * GCN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * GCN_WAVES_PER_BLOCK;
    const size_t ntiles = (rows + 15) / 16;
    const int BC_stride = V * V * C;
    ChanVec<MT> w2[2];
    w2[0] = chan_load<MT>(hp.outW, H, q);
    w2[1] = chan_load<MT>(voc > 1 ? hp.outW + H : nullptr, H, q);
    const float b2_0 = hp.outB[0], b2_1 = voc > 1 ? hp.outB[1] : 0.f;
    const float cw0 = edge_cw ? edge_cw[0] : 0.f, cw1 = edge_cw ? edge_cw[1] : 0.f;
    float acc_wn = 0.f, acc_w = 0.f;
    for (size_t tile = wave; tile < ntiles; tile += nwaves) {
        const size_t row = tile * 16 + rs;
        const bool valid = row < rows;
        Tile<MT> h = edge_src_tile<MT>(src, cl, sdc, row, valid, H, C, q, nullptr, nullptr, nullptr);
#pragma unroll
        for (int k = 0; k < NHID; ++k) {
            Tile<MT> o;
            if constexpr (WLDS) {
                o = tile_gemm_lds<MT>(h, wl + lds_opaque_zero(), lane);
            } else {
                WFrag<MT> wf;
                wfrag_load_nt(wf, hp.hidW[k], H, rs, q);
                o = tile_gemm(h, wf);
            }
            const ChanVec<MT> bv = chan_load<MT>(hp.hidB[k], H, q);
            const DropCfg dc = drop_cfg(rng, stream_id0 + k, p, training);
            const uint32_t rh = dc.on ? drop_rowhash(dc, (uint32_t)row) : 0u;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = o.v[m][r] + bv.v[m][r];
                    v = v > 0.f ? v : 0.f;
                    if (dc.on) v *= drop_mul(dc, rh, 16 * m + 4 * q + r);
                    h.v[m][r] = v;
                }
        }
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                l0 += h.v[m][r] * w2[0].v[m][r];
                l1 += h.v[m][r] * w2[1].v[m][r];
            }
        l0 = redq(l0) + b2_0;
        l1 = redq(l1) + b2_1;
        if (valid && q == 0) {
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
                    const float logp = (y ? l1 : l0) - lse;
                    const float w = y ? cw1 : cw0;
                    acc_wn += -w * logp;
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
    if (loss_partial != nullptr) {
        // lanes with q != 0 hold zeros; sum the 16 row slots of quarter 0
        acc_wn = red16(acc_wn);
        acc_w = red16(acc_w);
        if (lane == 0) {
            loss_partial[(size_t)wave * 2] = acc_wn;
            loss_partial[(size_t)wave * 2 + 1] = acc_w;
        }
    }
}

// loss = sum(w*nll) / sum(w); edge error = 100 * (1 - mean_{b,c} correct/mask) (0/0 -> NaN as in the
// reference).  out[0] = loss, out[1] = sum w, out[2] = edge error %, flags[b*C+c] = acc < 1.
// counts: nrep replicas of [2][BC] (correct, mask) counters, added here (the head spreads its increments over
// the replicas so that they do not queue up on a dozen cache lines).
__global__ __launch_bounds__(GCN_BLOCK) void k_loss_finalize(const float* __restrict__ loss_partial, int nP,
                                                            const int* __restrict__ counts, int BC, int nrep,
                                                            float* __restrict__ out, unsigned char* __restrict__ flags) {
    __shared__ double lds[GCN_WAVES_PER_BLOCK][3];
    double a = 0, w = 0;
    for (int p = threadIdx.x; p < nP; p += GCN_BLOCK) { a += loss_partial[(size_t)p * 2]; w += loss_partial[(size_t)p * 2 + 1]; }
    double accsum = 0;
    for (int k = threadIdx.x; k < BC; k += GCN_BLOCK) {
        int correct = 0, mask = 0;
#pragma unroll 4
        for (int r = 0; r < nrep; ++r) { correct += counts[(size_t)r * 2 * BC + k]; mask += counts[(size_t)r * 2 * BC + BC + k]; }
        const float acc = (float)correct / (float)mask;
        flags[k] = acc < 1.0f ? 1 : 0;
        accsum += (double)acc;
    }
    // fixed order: butterfly over the lanes of each wavefront, then the wavefronts in order
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off, 64); w += __shfl_xor(w, off, 64); accsum += __shfl_xor(accsum, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6][0] = a; lds[threadIdx.x >> 6][1] = w; lds[threadIdx.x >> 6][2] = accsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot_a = 0, tot_w = 0, tot_acc = 0;
        for (int k = 0; k < GCN_WAVES_PER_BLOCK; ++k) { tot_a += lds[k][0]; tot_w += lds[k][1]; tot_acc += lds[k][2]; }
        out[0] = (float)(tot_a / tot_w);
        out[1] = (float)tot_w;
        out[2] = (float)(100.0 * (1.0 - tot_acc / (double)BC));
    }
}
