// C ABI of libgcn_uelb_hip.so: argument checks, workspace carving and kernel launches.
// The entry points and what they replace in the reference are documented in include/gcn_uelb_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/gcn_uelb_hip.h"
#include "beam.hpp"
#include "kernels_bwd.hpp"
#include "kernels_fwd.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- optional per-kernel timing with HIP events (used by bench.py; off by default) -----------------
// gcn_prof_begin(substr) arms it: every later launch whose name contains substr is bracketed by two
// events on its own stream.  gcn_prof_end() synchronises those events and returns the summed time.
struct ProfState {
    bool on = false;
    char match[64] = "";
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    hipEvent_t pending = nullptr;
};
ProfState g_prof;

inline void prof_pre(const char* name, hipStream_t st) {
    if (!g_prof.on || strstr(name, g_prof.match) == nullptr) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    (void)hipEventRecord(a, st);
    g_prof.ev.emplace_back(a, b);
    g_prof.pending = b;
}
inline void prof_post(hipStream_t st) {
    if (g_prof.pending) { (void)hipEventRecord(g_prof.pending, st); g_prof.pending = nullptr; }
}

#define GCN_PROF_PRE(name) prof_pre(name, (hipStream_t)GCN_STREAM)
#define GCN_CHECK_LAUNCH(name)                                                                   \
    do {                                                                                         \
        prof_post((hipStream_t)GCN_STREAM);                                                      \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return fail(-3, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

#define GCN_STREAM st

struct Dims {
    int B, V, C, H;
    size_t Re, Rn;   // edge rows, node rows
    int njt;         // 16-row j tiles per (b, i, c)
    int IT, nit;     // i-chunk length / count of the edge backward pass
    int MT;
};

Dims make_dims(int B, int V, int C, int H) {
    Dims d;
    d.B = B; d.V = V; d.C = C; d.H = H;
    d.Rn = (size_t)B * V * C;
    d.Re = d.Rn * V;
    d.njt = (V + 15) / 16;
    // enough (b, c, i-chunk, j-tile) units to give every SIMD of the chip a wavefront
    const long long fine = (long long)B * C * d.njt * V;
    d.IT = (int)std::max<long long>(1, std::min<long long>(V, fine / 1024));
    d.nit = (V + d.IT - 1) / d.IT;
    d.MT = (H + 15) / 16;
    return d;
}

int check_dims(int B, int V, int C, int H) {
    if (B <= 0 || V <= 0 || C <= 0 || H <= 0) return fail(-1, "non-positive dimension B=%d V=%d C=%d H=%d", B, V, C, H);
    if (H % 4 != 0) return fail(-1, "hidden_dim %d must be a multiple of 4", H);
    if (H > GCN_MAX_H) return fail(-1, "hidden_dim %d > %d is not supported by this build", H, GCN_MAX_H);
    if ((double)B * V * V * C * H >= 4.0e9 * 16) return fail(-1, "problem too large");
    if ((double)B * V * V * C >= 4.0e9) return fail(-1, "more than 2^32 edge rows");
    return 0;
}

inline int wave_blocks(size_t units, int cap = 512) {
    return (int)std::max<size_t>(1, std::min<size_t>((units + GCN_WAVES_PER_BLOCK - 1) / GCN_WAVES_PER_BLOCK, (size_t)cap));
}
inline int elem_blocks(size_t n4) { return (int)std::max<size_t>(1, std::min<size_t>((n4 + 255) / 256, 4096)); }

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Bump {
    char* base;
    size_t cap, off;
    Bump(void* p, size_t c) : base((char*)p), cap(c), off(0) {}
    float* f(size_t nfloats) { return (float*)raw(nfloats * sizeof(float)); }
    void* raw(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    }
    bool ok() const { return off <= cap; }
};

// ---- layer-level buffer plans -------------------------------------------------------------------
struct LayerSaved {
    float *e_tmp, *x_tmp, *A, *D, *Vn, *stats;
};
size_t layer_saved_floats(const Dims& d) { return d.Re * d.H + 4 * d.Rn * d.H + 4 * (size_t)d.H; }
LayerSaved carve_saved(const Dims& d, float* p) {
    LayerSaved s;
    s.e_tmp = p; p += d.Re * d.H;
    s.x_tmp = p; p += d.Rn * d.H;
    s.A = p; p += d.Rn * d.H;
    s.D = p; p += d.Rn * d.H;
    s.Vn = p; p += d.Rn * d.H;
    s.stats = p;
    return s;
}

struct LayerFwdWs {
    float *Ux, *Ve, *partial;
    int nblk;
};
LayerFwdWs carve_fwd_ws(const Dims& d, Bump& b) {
    LayerFwdWs w;
    w.nblk = wave_blocks(d.Rn);
    w.Ux = b.f(d.Rn * d.H);
    w.Ve = b.f(d.Rn * d.H);
    w.partial = b.f((size_t)w.nblk * GCN_WAVES_PER_BLOCK * 4 * d.H);
    return w;
}

int tn_slices(size_t rows) {
    const size_t ntiles = (rows + 15) / 16;
    return (int)std::max<size_t>(1, std::min<size_t>(ntiles / 4, 256));
}

struct LayerBwdWs {
    float *det, *bpart_e, *bpart_x, *bstats_e, *bstats_x, *dxtmp, *qv, *rowpart, *colVe, *colVn, *dVe, *dVn;
    float *tn_e, *tn_n;
    int nblk_e, nblk_x, S_e, S_n;
};
LayerBwdWs carve_bwd_ws(const Dims& d, Bump& b) {
    LayerBwdWs w;
    const size_t nh = d.Rn * d.H;
    w.nblk_e = wave_blocks((d.Re + 15) / 16);
    w.nblk_x = wave_blocks((d.Rn + 15) / 16);
    w.S_e = tn_slices(d.Re);
    w.S_n = tn_slices(d.Rn);
    w.det = b.f(d.Re * d.H);
    w.bpart_e = b.f((size_t)w.nblk_e * GCN_WAVES_PER_BLOCK * 2 * d.H);
    w.bpart_x = b.f((size_t)w.nblk_x * GCN_WAVES_PER_BLOCK * 2 * d.H);
    w.bstats_e = b.f(2 * d.H);
    w.bstats_x = b.f(2 * d.H);
    w.dxtmp = b.f(nh);
    w.qv = b.f(nh);
    w.rowpart = b.f((size_t)d.njt * nh);
    w.colVe = b.f((size_t)d.nit * nh);
    w.colVn = b.f((size_t)d.nit * nh);
    w.dVe = b.f(nh);
    w.dVn = b.f(nh);
    const size_t slab = (size_t)d.H * d.H + d.H;
    w.tn_e = b.f((size_t)w.S_e * slab);
    w.tn_n = b.f((size_t)w.S_n * 3 * slab);
    return w;
}

// ---- launches -------------------------------------------------------------------------------------
template <int MT>
int layer_fwd_t(const Dims& d, int agg, float p, int training, const unsigned long long* rng, int sid,
                const float* x_in, const float* e_in, const float* const* P, float* x_out, float* e_out,
                const LayerSaved& S, const LayerFwdWs& W, hipStream_t st) {
    const int H = d.H;
    GCN_PROF_PRE("k_node_gemm3");
    k_node_gemm3<MT><<<wave_blocks((d.Rn + 15) / 16), GCN_BLOCK, 0, st>>>(
        (int)d.Rn, H, x_in, P[0], P[1], W.Ux, P[2], P[3], S.Vn, P[6], P[7], W.Ve);
    GCN_CHECK_LAUNCH("k_node_gemm3");
    GCN_PROF_PRE("k_edge_fwd");
    k_edge_fwd<MT><<<W.nblk, GCN_BLOCK, 0, st>>>(d.B, d.V, d.C, H, agg, e_in, P[4], P[5], W.Ve, S.Vn, W.Ux,
                                                 S.e_tmp, S.x_tmp, S.A, S.D, W.partial);
    GCN_CHECK_LAUNCH("k_edge_fwd");
    GCN_PROF_PRE("k_bn_stats_finalize");
    k_bn_stats_finalize<<<2 * (H / 4), GCN_BLOCK, 0, st>>>(W.partial, W.nblk * GCN_WAVES_PER_BLOCK, H, (double)d.Re,
                                                           (double)d.Rn, S.stats);
    GCN_CHECK_LAUNCH("k_bn_stats_finalize");
    GCN_PROF_PRE("k_bn_apply(edges)");
    k_bn_apply<<<elem_blocks(d.Re * (H / 4)), 256, 0, st>>>(d.Re, H, S.e_tmp, e_in, S.stats, S.stats + H, P[10], P[11],
                                                            e_out, rng, 2 * sid, p, training);
    GCN_CHECK_LAUNCH("k_bn_apply(edges)");
    if (x_out != nullptr) {
        GCN_PROF_PRE("k_bn_apply(nodes)");
        k_bn_apply<<<elem_blocks(d.Rn * (H / 4)), 256, 0, st>>>(d.Rn, H, S.x_tmp, x_in, S.stats + 2 * H, S.stats + 3 * H,
                                                                P[8], P[9], x_out, rng, 2 * sid + 1, p, training);
        GCN_CHECK_LAUNCH("k_bn_apply(nodes)");
    }
    return 0;
}

template <int MT>
int layer_bwd_t(const Dims& d, int agg, float p, int training, const unsigned long long* rng, int sid,
                const float* x_in, const float* e_in, const float* const* P, const LayerSaved& S, float* dx,
                float* de, float* const* G, const LayerBwdWs& W, hipStream_t st) {
    const int H = d.H;
    GCN_PROF_PRE("k_bn_bwd_stats(edges)");
    k_bn_bwd_stats<MT><<<W.nblk_e, GCN_BLOCK, 0, st>>>(d.Re, H, de, S.e_tmp, S.stats, S.stats + H, P[10], P[11],
                                                       W.bpart_e, rng, 2 * sid, p, training);
    GCN_CHECK_LAUNCH("k_bn_bwd_stats(edges)");
    GCN_PROF_PRE("k_bn_bwd_stats(nodes)");
    k_bn_bwd_stats<MT><<<W.nblk_x, GCN_BLOCK, 0, st>>>(d.Rn, H, dx, S.x_tmp, S.stats + 2 * H, S.stats + 3 * H, P[8], P[9],
                                                       W.bpart_x, rng, 2 * sid + 1, p, training);
    GCN_CHECK_LAUNCH("k_bn_bwd_stats(nodes)");
    GCN_PROF_PRE("k_bn_bwd_finalize(edges)");
    k_bn_bwd_finalize<<<H / 4, GCN_BLOCK, 0, st>>>(W.bpart_e, W.nblk_e * GCN_WAVES_PER_BLOCK, H, (double)d.Re, W.bstats_e,
                                                   G[10], G[11]);
    GCN_CHECK_LAUNCH("k_bn_bwd_finalize(edges)");
    GCN_PROF_PRE("k_bn_bwd_finalize(nodes)");
    k_bn_bwd_finalize<<<H / 4, GCN_BLOCK, 0, st>>>(W.bpart_x, W.nblk_x * GCN_WAVES_PER_BLOCK, H, (double)d.Rn, W.bstats_x,
                                                   G[8], G[9]);
    GCN_CHECK_LAUNCH("k_bn_bwd_finalize(nodes)");
    GCN_PROF_PRE("k_node_bwd_pre");
    k_node_bwd_pre<<<elem_blocks(d.Rn * (H / 4)), 256, 0, st>>>(d.Rn, H, dx, S.x_tmp, S.D, S.stats + 2 * H, S.stats + 3 * H,
                                                                P[8], P[9], W.bstats_x, dx, W.dxtmp, W.qv, rng, 2 * sid + 1,
                                                                p, training);
    GCN_CHECK_LAUNCH("k_node_bwd_pre");
    const size_t units = (size_t)d.B * d.nit * d.njt * d.C;
    GCN_PROF_PRE("k_edge_bwd_main");
    k_edge_bwd_main<MT><<<wave_blocks(units), GCN_BLOCK, 0, st>>>(d.B, d.V, d.C, H, d.IT, agg, de, S.e_tmp, P[4], S.Vn, S.A, W.qv,
                                                                  S.stats, P[10], P[11], W.bstats_e, W.det, W.rowpart,
                                                                  W.colVe, W.colVn, rng, 2 * sid, p, training);
    GCN_CHECK_LAUNCH("k_edge_bwd_main");
    GCN_PROF_PRE("k_node_bwd");
    k_node_bwd<MT><<<wave_blocks((d.Rn + 15) / 16), GCN_BLOCK, 0, st>>>((int)d.Rn, H, d.njt, d.nit, W.rowpart, W.colVe,
                                                                       W.colVn, W.dxtmp, P[0], P[2], P[6], dx, W.dVe, W.dVn);
    GCN_CHECK_LAUNCH("k_node_bwd");
    const int slab = H * H + H;
    {
        TnProblems pr = {};
        pr.X[0] = W.det; pr.Y[0] = e_in;
        GCN_PROF_PRE("k_gemm_tn(edges)");
        k_gemm_tn<MT><<<dim3(W.S_e, (MT + 3) / 4, 1), GCN_BLOCK, 0, st>>>(d.Re, H, 1, pr, W.tn_e);
        GCN_CHECK_LAUNCH("k_gemm_tn(edges)");
        ReduceDst rd = {};
        rd.n = 2; rd.dst[0] = G[4]; rd.dst[1] = G[5];
        rd.off[0] = 0; rd.off[1] = H * H; rd.off[2] = slab;
        GCN_PROF_PRE("k_reduce_slabs(edges)");
        k_reduce_slabs<<<(slab + 255) / 256, 256, 0, st>>>(W.tn_e, W.S_e, slab, rd);
        GCN_CHECK_LAUNCH("k_reduce_slabs(edges)");
    }
    {
        TnProblems pr = {};
        GCN_PROF_PRE("k_gemm_tn(nodes)");
        pr.X[0] = W.dxtmp; pr.Y[0] = x_in;   // node_feat.U
        pr.X[1] = W.dVn; pr.Y[1] = x_in;     // node_feat.V
        pr.X[2] = W.dVe; pr.Y[2] = x_in;     // edge_feat.V
        k_gemm_tn<MT><<<dim3(W.S_n, (MT + 3) / 4, 3), GCN_BLOCK, 0, st>>>(d.Rn, H, 3, pr, W.tn_n);
        GCN_CHECK_LAUNCH("k_gemm_tn(nodes)");
        ReduceDst rd = {};
        rd.n = 6;
        rd.dst[0] = G[0]; rd.dst[1] = G[1]; rd.dst[2] = G[2]; rd.dst[3] = G[3]; rd.dst[4] = G[6]; rd.dst[5] = G[7];
        for (int k = 0; k < 3; ++k) { rd.off[2 * k] = k * slab; rd.off[2 * k + 1] = k * slab + H * H; }
        rd.off[6] = 3 * slab;
        GCN_PROF_PRE("k_reduce_slabs(nodes)");
        k_reduce_slabs<<<(3 * slab + 255) / 256, 256, 0, st>>>(W.tn_n, W.S_n, 3 * slab, rd);
        GCN_CHECK_LAUNCH("k_reduce_slabs(nodes)");
    }
    return 0;
}

#define DISPATCH_MT(MTV, CALL)                                   \
    switch (MTV) {                                               \
        case 1: { constexpr int MT = 1; CALL; } break;           \
        case 2: { constexpr int MT = 2; CALL; } break;           \
        case 3: { constexpr int MT = 3; CALL; } break;           \
        case 4: { constexpr int MT = 4; CALL; } break;           \
        default: return fail(-1, "unsupported hidden_dim tile count %d", MTV); \
    }

// ---- model-level plans -----------------------------------------------------------------------------
struct ModelPlan {
    Dims d;
    int L, NHID, voc, nparams;
    size_t x_act, e_act, lsaved;    // float offsets inside `saved`
    size_t lsaved_each, total_saved;
};
ModelPlan make_plan(const gcn_model_desc* m) {
    ModelPlan p;
    p.d = make_dims(m->B, m->V, m->C, m->H);
    p.L = m->L; p.NHID = m->mlp_layers - 1; p.voc = m->voc_out;
    p.nparams = 3 + GCN_LAYER_NPARAMS * m->L + 2 * m->mlp_layers;
    const size_t nh = p.d.Rn * p.d.H, eh = p.d.Re * p.d.H;
    p.lsaved_each = layer_saved_floats(p.d);
    p.x_act = 0;
    p.e_act = p.x_act + (size_t)(p.L + 1) * nh;
    p.lsaved = p.e_act + (size_t)(p.L + 1) * eh;
    p.total_saved = p.lsaved + (size_t)p.L * p.lsaved_each;
    return p;
}
int check_model(const gcn_model_desc* m) {
    if (m == nullptr) return fail(-1, "null model descriptor");
    if (int rc = check_dims(m->B, m->V, m->C, m->H)) return rc;
    if (m->L < 1) return fail(-1, "num_layers %d < 1", m->L);
    if (m->mlp_layers < 1 || m->mlp_layers > GCN_MAX_MLP_LAYERS) return fail(-1, "mlp_layers %d outside 1..%d", m->mlp_layers, GCN_MAX_MLP_LAYERS);
    if (m->voc_out != 1 && m->voc_out != 2) return fail(-1, "voc_edges_out %d must be 1 or 2", m->voc_out);
    if (m->H % 2 != 0) return fail(-1, "hidden_dim must be even");
    if (m->n_emb < 1) return fail(-1, "n_emb %d < 1", m->n_emb);
    if (m->has_labels && m->voc_out != 2) return fail(-1, "loss_edges needs voc_edges_out == 2");
    return 0;
}

struct ModelFwdWs {
    LayerFwdWs lw;
    float* loss_partial;
    int head_blk;
};
ModelFwdWs carve_model_fwd(const ModelPlan& p, Bump& b) {
    ModelFwdWs w;
    w.lw = carve_fwd_ws(p.d, b);
    w.head_blk = wave_blocks((p.d.Re + 15) / 16);
    w.loss_partial = b.f((size_t)w.head_blk * GCN_WAVES_PER_BLOCK * 2);
    return w;
}
struct ModelBwdWs {
    LayerBwdWs lw;
    float *de, *dx, *dpre_stash, *h_stash, *wpartial, *tn_h, *embpart;
    int head_blk, S_h, emb_blk;
};
ModelBwdWs carve_model_bwd(const ModelPlan& p, Bump& b) {
    ModelBwdWs w;
    const size_t eh = p.d.Re * p.d.H;
    w.lw = carve_bwd_ws(p.d, b);
    w.de = b.f(eh);
    w.dx = b.f(p.d.Rn * p.d.H);
    // with one hidden layer the pre-activation stash aliases the layers' det buffer (the head finishes
    // before layer L-1 starts); deeper MLPs get NHID contiguous slabs of their own
    w.dpre_stash = p.NHID > 1 ? b.f((size_t)p.NHID * eh) : nullptr;
    w.h_stash = p.NHID > 1 ? b.f((size_t)(p.NHID - 1) * eh) : nullptr;
    w.head_blk = wave_blocks((p.d.Re + 15) / 16);
    w.wpartial = b.f((size_t)w.head_blk * GCN_WAVES_PER_BLOCK * (size_t)((p.voc * p.d.H + p.voc + 3) & ~3));
    w.S_h = tn_slices(p.d.Re);
    w.tn_h = b.f((size_t)w.S_h * std::max(1, p.NHID) * ((size_t)p.d.H * p.d.H + p.d.H));
    w.emb_blk = wave_blocks((p.d.Re + 15) / 16);
    w.embpart = b.f((size_t)w.emb_blk * GCN_WAVES_PER_BLOCK * p.d.H);
    return w;
}

template <int MT, int NHID>
void launch_head_fwd(const ModelPlan& p, const gcn_model_desc* m, const HeadParams& hp, const float* eL, const float* cap,
                     const long long* y_edges, const float* edge_cw, float* y_pred, float* loss_partial, int* counts,
                     const unsigned long long* rng, int blocks, hipStream_t st) {
    k_head_fwd<MT, NHID><<<blocks, GCN_BLOCK, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, eL, cap, m->mask_invalid_edges,
                                                        m->has_labels ? y_edges : nullptr, edge_cw, y_pred,
                                                        m->has_labels ? loss_partial : nullptr, counts, rng, 1000, m->p_mlp,
                                                        m->training);
}
template <int MT, int NHID>
void launch_head_bwd(const ModelPlan& p, const gcn_model_desc* m, const HeadParams& hp, const float* eL, const float* cap,
                     const long long* y_edges, const float* edge_cw, const float* loss_out, const float* dloss,
                     const float* dy_pred, float* de, float* dpre, float* hst, float* wpartial,
                     const unsigned long long* rng, int blocks, hipStream_t st) {
    k_head_bwd<MT, NHID><<<blocks, GCN_BLOCK, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, eL, cap, m->mask_invalid_edges,
                                                        m->has_labels ? y_edges : nullptr, edge_cw, loss_out, dloss, dy_pred, de,
                                                        dpre, hst, wpartial, rng, 1000, m->p_mlp, m->training);
}

#define DISPATCH_NHID(NH, CALL)                                  \
    switch (NH) {                                                \
        case 0: { constexpr int NHID = 0; CALL; } break;         \
        case 1: { constexpr int NHID = 1; CALL; } break;         \
        case 2: { constexpr int NHID = 2; CALL; } break;         \
        case 3: { constexpr int NHID = 3; CALL; } break;         \
        default: return fail(-1, "unsupported mlp_layers %d", NH + 1); \
    }

HeadParams head_params(const ModelPlan& p, const float* const* P) {
    HeadParams hp = {};
    const int base = 3 + GCN_LAYER_NPARAMS * p.L;
    for (int k = 0; k < p.NHID; ++k) { hp.hidW[k] = P[base + 2 * k]; hp.hidB[k] = P[base + 2 * k + 1]; }
    hp.outW = P[base + 2 * p.NHID];
    hp.outB = P[base + 2 * p.NHID + 1];
    return hp;
}

}  // namespace

extern "C" {

int gcn_abi_version(void) { return GCN_ABI_VERSION; }

int gcn_prof_begin(const char* kernel_substr) {
    if (!kernel_substr) return fail(-1, "gcn_prof_begin: null name");
    for (auto& e : g_prof.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    g_prof.ev.clear();
    g_prof.pending = nullptr;
    strncpy(g_prof.match, kernel_substr, sizeof(g_prof.match) - 1);
    g_prof.match[sizeof(g_prof.match) - 1] = 0;
    g_prof.on = true;
    return 0;
}

int gcn_prof_end(double* total_ms, int* launches) {
    g_prof.on = false;
    double tot = 0;
    int n = 0;
    for (auto& e : g_prof.ev) {
        float ms = 0.f;
        if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { tot += ms; ++n; }
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    g_prof.ev.clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return 0;
}
const char* gcn_last_error(void) { return g_err; }

size_t gcn_layer_saved_floats(int B, int V, int C, int H) {
    if (check_dims(B, V, C, H)) return 0;
    return layer_saved_floats(make_dims(B, V, C, H));
}
size_t gcn_layer_ws_bytes(int B, int V, int C, int H) {
    if (check_dims(B, V, C, H)) return 0;
    const Dims d = make_dims(B, V, C, H);
    Bump f(nullptr, 0), b(nullptr, 0);
    carve_fwd_ws(d, f);
    carve_bwd_ws(d, b);
    return std::max(f.off, b.off);
}

int gcn_layer_fwd(int B, int V, int C, int H, int aggregation, float p_drop, int training,
                  const unsigned long long* rng, int stream_id, const float* x_in, const float* e_in,
                  const float* const* params, float* x_out, float* e_out, float* saved, void* ws, size_t ws_bytes,
                  void* stream) {
    if (int rc = check_dims(B, V, C, H)) return rc;
    if (!x_in || !e_in || !params || !x_out || !e_out || !saved || !ws) return fail(-1, "gcn_layer_fwd: null pointer argument");
    if (training && p_drop > 0.f && !rng) return fail(-1, "gcn_layer_fwd: rng state required when dropout is active");
    const Dims d = make_dims(B, V, C, H);
    Bump b(ws, ws_bytes);
    const LayerFwdWs W = carve_fwd_ws(d, b);
    if (!b.ok()) return fail(-2, "gcn_layer_fwd: workspace too small (%zu < %zu bytes)", ws_bytes, b.off);
    const LayerSaved S = carve_saved(d, saved);
    DISPATCH_MT(d.MT, return layer_fwd_t<MT>(d, aggregation, p_drop, training, rng, stream_id, x_in, e_in, params, x_out,
                                             e_out, S, W, (hipStream_t)stream));
    return 0;
}

int gcn_layer_bwd(int B, int V, int C, int H, int aggregation, float p_drop, int training,
                  const unsigned long long* rng, int stream_id, const float* x_in, const float* e_in,
                  const float* const* params, const float* saved, float* dx, float* de, float* const* grads, void* ws,
                  size_t ws_bytes, void* stream) {
    if (int rc = check_dims(B, V, C, H)) return rc;
    if (!x_in || !e_in || !params || !saved || !dx || !de || !grads || !ws) return fail(-1, "gcn_layer_bwd: null pointer argument");
    if (training && p_drop > 0.f && !rng) return fail(-1, "gcn_layer_bwd: rng state required when dropout is active");
    const Dims d = make_dims(B, V, C, H);
    Bump b(ws, ws_bytes);
    const LayerBwdWs W = carve_bwd_ws(d, b);
    if (!b.ok()) return fail(-2, "gcn_layer_bwd: workspace too small (%zu < %zu bytes)", ws_bytes, b.off);
    const LayerSaved S = carve_saved(d, const_cast<float*>(saved));
    DISPATCH_MT(d.MT, return layer_bwd_t<MT>(d, aggregation, p_drop, training, rng, stream_id, x_in, e_in, params, S, dx, de,
                                             grads, W, (hipStream_t)stream));
    return 0;
}

int gcn_model_num_params(const gcn_model_desc* m) {
    if (check_model(m)) return -1;
    return 3 + GCN_LAYER_NPARAMS * m->L + 2 * m->mlp_layers;
}
size_t gcn_model_saved_floats(const gcn_model_desc* m) {
    if (check_model(m)) return 0;
    return make_plan(m).total_saved;
}
size_t gcn_model_ws_bytes(const gcn_model_desc* m) {
    if (check_model(m)) return 0;
    const ModelPlan p = make_plan(m);
    Bump f(nullptr, 0), b(nullptr, 0);
    carve_model_fwd(p, f);
    carve_model_bwd(p, b);
    return std::max(f.off, b.off);
}

int gcn_model_fwd(const gcn_model_desc* m, const float* cap, const float* demand, const long long* x_nodes,
                  const long long* y_edges, const float* edge_cw, const float* const* P,
                  const unsigned long long* rng, float* y_pred, float* loss_out, int* err_counts,
                  unsigned char* err_flags, float* saved, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_model(m)) return rc;
    if (!cap || !demand || !x_nodes || !P || !y_pred || !saved || !ws) return fail(-1, "gcn_model_fwd: null pointer argument");
    if (m->has_labels && (!y_edges || !edge_cw || !loss_out || !err_counts || !err_flags))
        return fail(-1, "gcn_model_fwd: has_labels needs y_edges, edge_cw, loss_out, err_counts, err_flags");
    if (m->training && (m->p_layer > 0.f || m->p_mlp > 0.f) && !rng) return fail(-1, "gcn_model_fwd: rng state required when dropout is active");
    const ModelPlan p = make_plan(m);
    const Dims& d = p.d;
    hipStream_t st = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    const ModelFwdWs W = carve_model_fwd(p, b);
    if (!b.ok()) return fail(-2, "gcn_model_fwd: workspace too small (%zu < %zu bytes)", ws_bytes, b.off);
    const int H = d.H;
    const size_t nh = d.Rn * H, eh = d.Re * H;
    float* x_act = saved + p.x_act;
    float* e_act = saved + p.e_act;
    GCN_PROF_PRE("k_embed_edges");
    k_embed_edges<<<elem_blocks(d.Re * (H / 4)), 256, 0, st>>>(d.Re * (size_t)(H / 4), d.C, H, cap, P[2], e_act);
    GCN_CHECK_LAUNCH("k_embed_edges");
    GCN_PROF_PRE("k_embed_nodes");
    k_embed_nodes<<<d.B * d.V, 64, 0, st>>>(d.V, d.C, H, m->n_emb, x_nodes, demand, P[0], P[1], x_act);
    GCN_CHECK_LAUNCH("k_embed_nodes");
    for (int l = 0; l < p.L; ++l) {
        const LayerSaved S = carve_saved(d, saved + p.lsaved + (size_t)l * p.lsaved_each);
        const float* const* LP = P + 3 + GCN_LAYER_NPARAMS * l;
        int rc = 0;
        DISPATCH_MT(d.MT, rc = layer_fwd_t<MT>(d, m->aggregation, m->p_layer, m->training, rng, l, x_act + (size_t)l * nh,
                                               e_act + (size_t)l * eh, LP, x_act + (size_t)(l + 1) * nh,
                                               e_act + (size_t)(l + 1) * eh, S, W.lw, st));
        if (rc) return rc;
    }
    const HeadParams hp = head_params(p, P);
    if (m->has_labels) {
        if (hipMemsetAsync(err_counts, 0, sizeof(int) * 2 * (size_t)d.B * d.C, st) != hipSuccess) return fail(-3, "memset failed");
    }
    GCN_PROF_PRE("k_head_fwd");
    DISPATCH_MT(d.MT, DISPATCH_NHID(p.NHID, (launch_head_fwd<MT, NHID>(p, m, hp, e_act + (size_t)p.L * eh, cap, y_edges, edge_cw,
                                                                      y_pred, W.loss_partial, err_counts, rng, W.head_blk, st))));
    GCN_CHECK_LAUNCH("k_head_fwd");
    if (m->has_labels) {
        GCN_PROF_PRE("k_loss_finalize");
        k_loss_finalize<<<1, GCN_BLOCK, 0, st>>>(W.loss_partial, W.head_blk * GCN_WAVES_PER_BLOCK, err_counts, d.B * d.C, loss_out,
                                                 err_flags);
        GCN_CHECK_LAUNCH("k_loss_finalize");
    }
    return 0;
}

int gcn_model_bwd(const gcn_model_desc* m, const float* cap, const float* demand, const long long* x_nodes,
                  const long long* y_edges, const float* edge_cw, const float* const* P,
                  const unsigned long long* rng, const float* saved, const float* loss_out, const float* dloss,
                  const float* dy_pred, float* const* G, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_model(m)) return rc;
    if (!cap || !demand || !x_nodes || !P || !saved || !G || !ws) return fail(-1, "gcn_model_bwd: null pointer argument");
    if (m->has_labels && (!y_edges || !edge_cw || !loss_out)) return fail(-1, "gcn_model_bwd: has_labels needs y_edges, edge_cw, loss_out");
    if (m->training && (m->p_layer > 0.f || m->p_mlp > 0.f) && !rng) return fail(-1, "gcn_model_bwd: rng state required when dropout is active");
    const ModelPlan p = make_plan(m);
    const Dims& d = p.d;
    hipStream_t st = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    const ModelBwdWs W = carve_model_bwd(p, b);
    if (!b.ok()) return fail(-2, "gcn_model_bwd: workspace too small (%zu < %zu bytes)", ws_bytes, b.off);
    const int H = d.H;
    const size_t nh = d.Rn * H, eh = d.Re * H;
    const float* x_act = saved + p.x_act;
    const float* e_act = saved + p.e_act;
    const HeadParams hp = head_params(p, P);
    const int hbase = 3 + GCN_LAYER_NPARAMS * p.L;
    const int slab = H * H + H;
    GCN_PROF_PRE("k_head_bwd");
    float* dpre_all = p.NHID == 1 ? W.lw.det : W.dpre_stash;   // [NHID][Re*H], see carve_model_bwd
    DISPATCH_MT(d.MT, DISPATCH_NHID(p.NHID, (launch_head_bwd<MT, NHID>(p, m, hp, e_act + (size_t)p.L * eh, cap, y_edges, edge_cw,
                                                                      loss_out, dloss, dy_pred, W.de, dpre_all, W.h_stash,
                                                                      W.wpartial, rng, W.head_blk, st))));
    GCN_CHECK_LAUNCH("k_head_bwd");
    {
        const int wcols = (p.voc * H + p.voc + 3) & ~3;   // per-wavefront stride of wpartial
        ReduceDst rd = {};
        rd.n = 2; rd.dst[0] = G[hbase + 2 * p.NHID]; rd.dst[1] = G[hbase + 2 * p.NHID + 1];
        rd.off[0] = 0; rd.off[1] = p.voc * H; rd.off[2] = p.voc * H + p.voc;
        GCN_PROF_PRE("k_reduce_slabs(head out)");
        k_reduce_slabs<<<(wcols + 255) / 256, 256, 0, st>>>(W.wpartial, W.head_blk * GCN_WAVES_PER_BLOCK, wcols, rd);
        GCN_CHECK_LAUNCH("k_reduce_slabs(head out)");
    }
    if (p.NHID > 0) {
        TnProblems pr = {};
        for (int k = 0; k < p.NHID; ++k) {
            pr.X[k] = dpre_all + (size_t)k * eh;
            pr.Y[k] = k == 0 ? e_act + (size_t)p.L * eh : W.h_stash + (size_t)(k - 1) * eh;
        }
        GCN_PROF_PRE("k_gemm_tn(head)");
        DISPATCH_MT(d.MT, (k_gemm_tn<MT><<<dim3(W.S_h, (MT + 3) / 4, p.NHID), GCN_BLOCK, 0, st>>>(d.Re, H, p.NHID, pr, W.tn_h)));
        GCN_CHECK_LAUNCH("k_gemm_tn(head)");
        ReduceDst rd = {};
        rd.n = 2 * p.NHID;
        for (int k = 0; k < p.NHID; ++k) {
            rd.dst[2 * k] = G[hbase + 2 * k]; rd.dst[2 * k + 1] = G[hbase + 2 * k + 1];
            rd.off[2 * k] = k * slab; rd.off[2 * k + 1] = k * slab + H * H;
        }
        rd.off[2 * p.NHID] = p.NHID * slab;
        GCN_PROF_PRE("k_reduce_slabs(head hidden)");
        k_reduce_slabs<<<(p.NHID * slab + 255) / 256, 256, 0, st>>>(W.tn_h, W.S_h, p.NHID * slab, rd);
        GCN_CHECK_LAUNCH("k_reduce_slabs(head hidden)");
    }
    // the node output of the last layer feeds nothing (gcn_model.py:89 uses e only): its gradient is zero
    if (hipMemsetAsync(W.dx, 0, nh * sizeof(float), st) != hipSuccess) return fail(-3, "memset failed");
    for (int l = p.L - 1; l >= 0; --l) {
        const LayerSaved S = carve_saved(d, const_cast<float*>(saved) + p.lsaved + (size_t)l * p.lsaved_each);
        const float* const* LP = P + 3 + GCN_LAYER_NPARAMS * l;
        float* const* LG = G + 3 + GCN_LAYER_NPARAMS * l;
        int rc = 0;
        DISPATCH_MT(d.MT, rc = layer_bwd_t<MT>(d, m->aggregation, m->p_layer, m->training, rng, l, x_act + (size_t)l * nh,
                                               e_act + (size_t)l * eh, LP, S, W.dx, W.de, LG, W.lw, st));
        if (rc) return rc;
    }
    // embeddings
    GCN_PROF_PRE("k_embed_bwd_edges");
    DISPATCH_MT(d.MT, (k_embed_bwd_edges<MT><<<W.emb_blk, GCN_BLOCK, 0, st>>>(d.Re, d.C, H, cap, W.de, W.embpart)));
    GCN_CHECK_LAUNCH("k_embed_bwd_edges");
    {
        ReduceDst rd = {};
        rd.n = 1; rd.dst[0] = G[2]; rd.off[0] = 0; rd.off[1] = H;
        GCN_PROF_PRE("k_reduce_slabs(edge embedding)");
        k_reduce_slabs<<<(H + 255) / 256, 256, 0, st>>>(W.embpart, W.emb_blk * GCN_WAVES_PER_BLOCK, H, rd);
        GCN_CHECK_LAUNCH("k_reduce_slabs(edge embedding)");
    }
    {
        const int hh = H / 2;
        const int threads = std::max(64, std::min(256, (256 / hh) * hh >= hh ? 256 : hh));
        GCN_PROF_PRE("k_embed_bwd_nodes");
        k_embed_bwd_nodes<<<m->n_emb + 1, threads, threads * sizeof(double), st>>>(d.B, d.V, d.C, H, m->n_emb, x_nodes, demand,
                                                                                W.dx, G[0], G[1]);
        GCN_CHECK_LAUNCH("k_embed_bwd_nodes");
    }
    return 0;
}


// ---- beam search ---------------------------------------------------------------------------------
size_t gcn_beam_ws_bytes(int B, int V, int C, int K) {
    (void)C; (void)K;
    if (B <= 0 || V <= 0) return 0;
    return align_up((size_t)B * V * V * sizeof(float));
}

int gcn_beam_search(int B, int V, int C, int K, int algorithm, int mode_strict, int max_iter, const float* y,
                    const float* cap, const long long* comm, const int* perms, int n_perms, const int* perm_start,
                    const unsigned char* run_mask, int* paths, int* path_len, int* feasible, int* attempts, void* ws,
                    size_t ws_bytes, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0 || K <= 0) return fail(-1, "gcn_beam_search: non-positive dimension B=%d V=%d C=%d K=%d", B, V, C, K);
    if (V > 65535) return fail(-1, "gcn_beam_search: V=%d exceeds 65535", V);
    if (algorithm < 0 || algorithm > 3) return fail(-1, "gcn_beam_search: unknown algorithm id %d", algorithm);
    if (max_iter < 1) return fail(-1, "gcn_beam_search: max_iter %d < 1", max_iter);
    if (!y || !cap || !comm || !paths || !path_len || !feasible || !attempts) return fail(-1, "gcn_beam_search: null pointer argument");
    const bool needs_perms = algorithm == BEAM_ALGO_STANDARD || algorithm == BEAM_ALGO_GREEDY;
    if (needs_perms && (!perms || !perm_start || n_perms < 1)) return fail(-1, "gcn_beam_search: this algorithm needs permutations");
    const bool strict = mode_strict && algorithm != BEAM_ALGO_UNCONSTRAINED;
    int cap_in_lds = strict && ((size_t)V * V * sizeof(float) <= 32768) ? 1 : 0;
    if (strict && !cap_in_lds) {
        if (!ws || ws_bytes < gcn_beam_ws_bytes(B, V, C, K)) return fail(-2, "gcn_beam_search: workspace too small");
    }
    const size_t lds = beam_lds_bytes(V, K, cap_in_lds);
    if (lds > 160 * 1024) return fail(-1, "gcn_beam_search: beam state of %zu bytes exceeds the 160 KiB LDS (V=%d, K=%d)", lds, V, K);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_beam_search), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(-3, "gcn_beam_search: cannot raise the dynamic LDS limit to %zu bytes", lds);
    }
    const int jobs = algorithm == BEAM_ALGO_UNCONSTRAINED ? B * C : B;
    hipStream_t st = (hipStream_t)stream;
    GCN_PROF_PRE("k_beam_search");
    k_beam_search<<<jobs, 64, lds, st>>>(B, V, C, K, algorithm, strict ? 1 : 0, max_iter, y, cap, comm, perms,
                                                          n_perms, perm_start, run_mask, (float*)ws, cap_in_lds, paths, path_len,
                                                          feasible, attempts);
    GCN_CHECK_LAUNCH("k_beam_search");
    return 0;
}

int gcn_beam_scores(int B, int V, int C, int five_d, const float* in, float* y, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0 || !in || !y) return fail(-1, "gcn_beam_scores: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (five_d) {
        const size_t rows = (size_t)B * V * V * C;
        k_beam_scores_2class<<<elem_blocks(rows), 256, 0, st>>>(rows, in, y);
    } else {
        k_beam_scores_over_j<<<elem_blocks((size_t)B * V * C), 256, 0, st>>>(B, V, C, in, y);
    GCN_PROF_PRE("k_beam_scores");
    }
    GCN_CHECK_LAUNCH("k_beam_scores");
    return 0;
}

}  // extern "C"
