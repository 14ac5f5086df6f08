// C ABI of libgcn_uelb_hip.so: argument checks, workspace carving and kernel launches.
// The entry points and what they replace in the reference are documented in include/gcn_uelb_hip.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/gcn_uelb_hip.h"
#include "beam.hpp"
#include "kernels_bwd.hpp"
#include "kernels_fwd.hpp"
#include "kernels_strip.hpp"
#include "kernels_strip2.hpp"
#include "kernels_wide.hpp"
#include "kernels_tn_bf.hpp"
#include "sampler.hpp"
#include "dataset.hpp"

namespace {

thread_local char g_err[512] = "";

// Measurement overrides (grid sizes, chunk lengths, probes) exist only in builds made with -DGCN_TUNING
// (tools/build_tuning_lib.sh); the product library reads no environment variable.
#ifdef GCN_TUNING
inline const char* tune_env(const char* name) { return getenv(name); }
#else
inline const char* tune_env(const char*) { return nullptr; }
#endif

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
struct ProfEvents {
    hipEvent_t e0, e1, e2;      // e0, e1: an empty bracket recorded right before the launch; e1, e2: the launch
};
struct ProfState {
    bool on = false;
    char match[64] = "";
    std::vector<ProfEvents> ev;
    hipEvent_t pending = nullptr;
};
ProfState g_prof;

inline void prof_pre(const char* name, hipStream_t st) {
    if (!g_prof.on || strstr(name, g_prof.match) == nullptr) return;
    ProfEvents p;
    if (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess || hipEventCreate(&p.e2) != hipSuccess) return;
    (void)hipEventRecord(p.e0, st);
    (void)hipEventRecord(p.e1, st);
    g_prof.ev.push_back(p);
    g_prof.pending = p.e2;
}
// For launches made with hipExtLaunchKernelGGL: the two events receive the kernel's own begin and end time
// stamps (what a kernel trace reports), so no bracket cost is involved.  e0 == nullptr marks such an entry.
inline void prof_ext(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    *start = nullptr; *stop = nullptr;
    if (!g_prof.on || strstr(name, g_prof.match) == nullptr) return;
    ProfEvents p;
    p.e0 = nullptr;
    if (hipEventCreate(&p.e1) != hipSuccess || hipEventCreate(&p.e2) != hipSuccess) return;
    g_prof.ev.push_back(p);
    *start = p.e1; *stop = p.e2;
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

// launch of one of the main kernels, timed by its own time stamps when the profiler is armed for its name
#define GCN_LAUNCH_TIMED_LDS(name, kern, grid, block, lds, ...)                                   \
    do {                                                                                         \
        hipEvent_t es_ = nullptr, ee_ = nullptr;                                                 \
        prof_ext(name, &es_, &ee_);                                                              \
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, (hipStream_t)GCN_STREAM, es_, ee_, 0, __VA_ARGS__); \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return fail(-3, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)
#define GCN_LAUNCH_TIMED(name, kern, grid, block, ...) GCN_LAUNCH_TIMED_LDS(name, kern, grid, block, 0, __VA_ARGS__)

struct Dims {
    int B, V, C, H;
    size_t Re, Rn;   // edge rows, node rows
    int njt;         // 16-row j tiles per (b, i, c)
    int IT, nit;     // i-chunk length / count of the edge backward pass
    int grid_f, grid_b;   // workgroups of the forward / backward edge kernel (upper bounds: no workgroup is left without work)
    int MT;
    bool two_blocks;      // H of 2 or 4 column blocks: the edge kernels with two column blocks per wavefront (kernels_strip2.hpp)
    bool fuse_wg;         // the edge backward kernel also forms the edge weight-gradient product (no det tensor in memory)
};

Dims make_dims(int B, int V, int C, int H) {
    Dims d;
    d.B = B; d.V = V; d.C = C; d.H = H;
    d.Rn = (size_t)B * V * C;
    d.Re = d.Rn * V;
    d.njt = (V + 15) / 16;
    // Row chunks of the edge backward pass: about 1024 (b, c, i-chunk, j-tile) units, chunks of equal length (68 + 32
    // rows at V = 100 would leave half of the workgroups with twice the work).
    const long long fine = (long long)B * C * d.njt * V;
    const int target = (int)std::max<long long>(1, std::min<long long>(V, (fine + 512) / 1024));
    auto balanced = [&](int it) { const int n = (V + it - 1) / it; return (V + n - 1) / n; };
    d.IT = balanced(target);
    d.grid_f = d.grid_b = 512;                              // two workgroups per CU
    d.MT = (H + 15) / 16;
    // Edge kernels with two column blocks per wavefront and the edge weight-gradient product inside the backward kernel
    // (kernels_strip2.hpp): from about 300,000 edge rows on, where the det tensor (written by the edge kernel, read
    // back with the layer input by a product kernel) is HBM traffic worth removing: 200 nodes x 100 commodities runs 5 %
    // faster with them and moves two edge tensors less per layer.  Below that the launch floor and the first steps of
    // every workgroup dominate and the kernels with one wavefront per column block are faster (same-box A/B of whole
    // steps: nsfnet_10 at 20 / 80 / 120 graphs 3 / 2 / 1 % slower with two blocks, at 160 / 200 / 320 / 640 graphs
    // 2 / 8 / 8 / 10 % faster; default2 at 20 graphs 6 % slower, at 40 / 60 / 80 graphs 9 / 9 / 4 % faster; nsfnet_20
    // at 20 graphs 11 % slower): the crossover lies between 235,000 and 314,000 rows.
    d.two_blocks = (d.MT == 2 || d.MT == 4) && d.Re >= 300000;
    if (const char* e = tune_env("GCN_TUNE_TWOBLOCKS")) d.two_blocks = (d.MT == 2 || d.MT == 4) && atoi(e) != 0;
    d.fuse_wg = d.two_blocks;
    if (const char* e = tune_env("GCN_TUNE_FUSE")) d.fuse_wg = d.two_blocks && atoi(e) != 0;
    if (H <= 64) {
        // Strip path: the edge kernels are issue bound, so a launch lasts as long as the busiest CU has 16-row steps to
        // issue: (workgroups on that CU) x (units per workgroup) x (steps per unit).  Grid (two or three workgroups per
        // CU) and chunk length (within a factor of two of the target) are the pair with the fewest such steps; on a
        // tie, three per CU once there are many steps per workgroup, and the chunk length nearest to the target.
        // (Measured against the fixed choice 512 / target: default2 8 rows instead of 10, 5 % faster; nsfnet_20 2 rows
        // instead of 5 and 512 forward workgroups instead of 768, 3 %; nsfnet_10 unchanged at 512 / 512 / 3 rows.)
        const int pref = fine >= 5000 ? 768 : 512;
        auto busiest = [](long long units, int grid, long long steps_per_unit) {
            const long long per = (units + grid - 1) / grid, active = (units + per - 1) / per;
            return ((active + 255) / 256) * per * steps_per_unit;
        };
        const int grids[2] = {pref, pref == 768 ? 512 : 768};
        long long best = -1;
        for (int g : grids)
            for (int it = std::max(1, target / 2); it <= std::min(V, 2 * target); ++it) {
                if (balanced(it) != it) continue;
                const long long cost = busiest((long long)B * ((V + it - 1) / it) * d.njt * C, g, it);
                if (best < 0 || cost < best || (cost == best && g == d.grid_b && std::abs(it - target) < std::abs(d.IT - target))) {
                    best = cost; d.IT = it; d.grid_b = g;
                }
            }
        d.grid_f = busiest((long long)d.Rn, grids[1], d.njt) < busiest((long long)d.Rn, grids[0], d.njt) ? grids[1] : grids[0];
        if (d.two_blocks) {
            // Workgroups of two wavefronts, four per CU (measured: 1024 and 2048 alike, 1536 5 % slower).  The chunk
            // length is the one that leaves the busiest CU the fewest steps at that grid, every chunk count n weighted
            // by the traffic of its column partial sums (2 n node tensors written and read against 5 V node tensors
            // of edge traffic).  Measured against the choice above: 200 nodes x 100 commodities 25 rows instead of
            // 100 (867 -> 1015 workgroups with equal ranges), 113.4 -> 108.7 ms per step; nsfnet_10 at 320 graphs 7
            // rows instead of 14, 6.17 -> 5.80 ms.
            d.grid_f = d.grid_b = 1024;
            double bestc = -1.0;
            for (int it = 1; it <= V; ++it) {
                if (balanced(it) != it) continue;
                const int n = (V + it - 1) / it;
                const double c = (double)busiest((long long)B * n * d.njt * C, d.grid_b, it) * (1.0 + 0.8 * n / V);
                if (bestc < 0 || c < bestc) { bestc = c; d.IT = it; }
            }
        }
    } else {
        d.grid_f = d.grid_b = 256;                          // one 512-thread workgroup per CU (kernels_wide_bf.hpp)
        // Row chunks of the wide backward kernel: a workgroup takes a contiguous range of (b, chunk, column tile, c) units of
        // ceil(IT / SNS) steps each, and the launch lasts as long as the workgroup with the most units: the chunk count n
        // (chunks of equal length, a multiple of SNS rows) with the fewest steps on that workgroup, a unit's fixed work
        // (column sums written, node rows fetched) counted as 0.15 of a step, and the node backward kernel's reads of the
        // 2 n column-sum tensors as 0.6 % of the launch each (measured: + 44 us at n = 5 against n = 2).  100 nodes x 50 commodities, 2 graphs:
        // 5 chunks of 20 rows (3,000 units, 12 x 5 steps) instead of 2 of 52 / 48 (1,200 units, 5 x 13 steps): 2.61 ->
        // 2.48 ms per layer, 190.7 -> 184.0 ms per step.
        const int njt_main = (d.njt > 1 && V - 16 * (d.njt - 1) <= 8) ? d.njt - 1 : d.njt;   // a ragged last tile has its own launch
        double bestc = -1.0;
        for (int n = 1; n <= (V + SNS - 1) / SNS; ++n) {
            const int it = std::min(V, ((V + n - 1) / n + SNS - 1) / SNS * SNS), nn = (V + it - 1) / it;
            const long long units = (long long)B * nn * njt_main * C, per = (units + d.grid_b - 1) / d.grid_b;
            const double c = (double)per * ((it + SNS - 1) / SNS + 0.15) * (1.0 + 0.6 * nn / V);   // + the node kernel's 2 n column-sum slots
            if (bestc < 0 || c < bestc) { bestc = c; d.IT = it; }
        }
    }
    if (const char* e = tune_env("GCN_TUNE_IT")) d.IT = std::max(1, std::min(V, atoi(e)));
    d.nit = (V + d.IT - 1) / d.IT;
    return d;
}

int check_dims(int B, int V, int C, int H) {
    if (B <= 0 || V <= 0 || C <= 0 || H <= 0) return fail(-1, "non-positive dimension B=%d V=%d C=%d H=%d", B, V, C, H);
    if (H % 4 != 0) return fail(-1, "hidden_dim %d must be a multiple of 4", H);
    if (H > GCN_MAX_H) return fail(-1, "hidden_dim %d > %d is not supported by this build", H, GCN_MAX_H);
    if ((double)B * V * V * C * H >= 4.0e9 * 16) return fail(-1, "problem too large");
    if ((double)B * V * V * C >= 4.0e9) return fail(-1, "more than 2^32 edge rows");
    if ((double)B * V * C * H * 4.0 >= 2147483648.0) return fail(-1, "node tensors of 2 GiB or more are not supported");
    if (16.0 * C * H * 4.0 >= 2147483648.0) return fail(-1, "a 16-row edge tile of 2 GiB or more is not supported");
    return 0;
}

constexpr int kMaxHeadBlocks = 2048;   // as many rows as the BatchNorm-backward partial buffers hold
inline int wave_blocks(size_t units, int cap = 512) {
    return (int)std::max<size_t>(1, std::min<size_t>((units + GCN_WAVES_PER_BLOCK - 1) / GCN_WAVES_PER_BLOCK, (size_t)cap));
}
// Workgroups of the head kernels (one 16-row tile per step, grid-stride).  Forward strip kernel: four per CU (its step
// ends in a serial stretch of one wavefront, softmax and loss of 16 rows, which more resident workgroups cover:
// 4.49 -> 3.15 ms at 200 nodes x 100 commodities, 194 -> 140 us for 320 NSFNet graphs, 18.0 -> 16.9 us for 20); the
// backward kernel (165 registers) and the wide kernels stay at two per CU (no gain from more).
inline int head_blocks(size_t tiles, bool forward, int H, bool two_blocks = false) {
    int cap = (forward && H <= 64) ? 1024 : 512;
    if (forward && two_blocks) cap = 1536;        // k_head_fwd4: workgroups of half the size, six per CU (three wavefronts per SIMD)
    if (!forward && two_blocks && H <= 64) cap = 768;   // k_head_bwd3 at three wavefronts per SIMD: three workgroups per CU
    if (const char* e = tune_env(forward ? "GCN_TUNE_GRID_H" : "GCN_TUNE_GRID_HB")) cap = std::max(1, std::min(atoi(e), kMaxHeadBlocks));
    return wave_blocks(tiles, cap);
}
// grid of the one-workgroup-per-unit kernels (grid-stride beyond the cap)
inline int strip_blocks(size_t units, int cap = 2048) { return (int)std::max<size_t>(1, std::min<size_t>(units, (size_t)cap)); }
// Grid of the strip node kernels (one 16-row tile per workgroup and trip, grid-stride beyond the cap).  A workgroup's
// prologue (three weight matrices as MFMA fragments, BatchNorm constants) costs as much as a few tiles, and every workgroup
// of the backward kernel adds a row of partial sums to reduce: one workgroup per CU up to 2048 tiles, two from there on
// (nsfnet_10 at 80 graphs, 700 tiles: 1.870 ms per step on 700 workgroups, 1.841 on 512, 1.808 on 384, 1.786 on 256, 1.833 on
// 128; at 320 graphs, 2800 tiles: 5.79 on 2048, 5.68 on 1024, 5.62 on 768, 5.59 on 512, 5.63 on 384, 5.68 on 256).
// the same for the node kernels of the wide path (64 < H <= 320)
inline int wide_node_blocks(size_t tiles, int cap = 2048) {
    if (const char* e = tune_env("GCN_TUNE_GRID_NW")) cap = std::max(1, std::min(atoi(e), 2048));
    return strip_blocks(tiles, cap);
}
inline int node_blocks(size_t tiles, int cap = 2048) {
    cap = std::min(cap, tiles >= 2048 ? 512 : 256);
    if (const char* e = tune_env("GCN_TUNE_GRID_N")) cap = std::max(1, std::min(atoi(e), 2048));
    return strip_blocks(tiles, cap);
}
// grid of a kernel whose workgroups each take a contiguous range of units: no workgroup is left without work
inline int range_blocks(size_t units, int cap) {
    const size_t per = (units + cap - 1) / cap;
    return (int)std::max<size_t>(1, (units + per - 1) / std::max<size_t>(per, 1));
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
    u32x4* wsplit;      // above 64 channels: bf16 planes of edge_feat.U.weight (kernels_wide_bf.hpp)
};
LayerFwdWs carve_fwd_ws(const Dims& d, Bump& b) {
    LayerFwdWs w;
    w.nblk = range_blocks(d.Rn, d.grid_f);
    if (d.MT <= 4) w.nblk = d.grid_f;                       // the strip forward kernels deal their units by XCD: a multiple of 8 workgroups
    if (const char* e = tune_env("GCN_TUNE_GRID_F")) w.nblk = d.MT <= 4 ? std::max(8, std::min(atoi(e), 2048) & ~7) : range_blocks(d.Rn, std::min(atoi(e), 2048));
    w.Ux = b.f(d.Rn * d.H);
    w.Ve = b.f(d.Rn * d.H);
    w.partial = b.f((size_t)w.nblk * 4 * d.H);
    w.wsplit = d.MT > 4 ? (u32x4*)b.raw(split_weight_bytes(d.H)) : nullptr;
    return w;
}

// workgroups of one slab reduction, and the grid x of a launch that runs several
int red_blocks(const ReduceJob& j) { const int c = reduce_cols(j.nP); return (j.ncols + c - 1) / c; }
int red_grid(const ReduceJobs& jobs) {
    int g = 1;
    for (int k = 0; k < jobs.n; ++k) g = std::max(g, red_blocks(jobs.job[k]));
    return g;
}

// Row slices of a weight-gradient product (one partial H x H slab each, added up afterwards).  Strip path: the cost is
// (rows per slice) + (slices to add up), so the count grows with the square root of the rows; 3.8 sqrt(tiles) matches the
// measured optima (192 / 256 / 384 slices at 2,450 / 4,900 / 10,000 row tiles).  Wide path: one 512-thread workgroup per
// slice, one slice per CU.
int tn_slices(size_t rows, int H) {
    const size_t ntiles = (rows + 15) / 16;
    if (const char* e = tune_env("GCN_TUNE_SLICES")) return (int)std::max<size_t>(1, std::min<size_t>(ntiles / 4, (size_t)atoi(e)));
    // wide kernels: one workgroup per CU, and a workgroup clears 132 KiB of LDS and writes a slab of H x H + H floats
    // whatever its share of rows: at least 8 chunks of 16 rows each
    if (H > 64) return (int)std::max<size_t>(1, std::min<size_t>(ntiles / 8, 256));
    const size_t s = ((size_t)(3.8 * std::sqrt((double)ntiles)) + 8) / 16 * 16;
    return (int)std::max<size_t>(1, std::min<size_t>(ntiles / 4, std::min<size_t>(std::max<size_t>(s, 16), 768)));   // 768: flat beyond at 10^6 row tiles
}

constexpr int kMaxPartBlocks = 2048;   // largest grid that writes rows of the BatchNorm-backward partial buffers

// workgroups of the edge backward kernel (each writes one row of BatchNorm-backward partial sums and, on the strip
// path, one slab of the edge weight gradient)
int edge_bwd_blocks(const Dims& d) {
    const size_t units = (size_t)d.B * d.nit * d.njt * d.C;
    int blocks = range_blocks(units, d.grid_b);
    if (const char* e = tune_env("GCN_TUNE_GRID_B")) blocks = range_blocks(units, std::min(atoi(e), kMaxPartBlocks));
    return blocks;
}

struct LayerBwdWs {
    float *det, *bpart_e, *bpart_x, *bstats_e, *bstats_x, *dxtmp, *rowpart, *colVe, *colVn, *dVe, *dVn;
    float *tn_e, *tn_n;
    int S_e, S_n;       // partial slabs of the edge / node weight-gradient products
    u32x4* wsplit;      // above 64 channels: bf16 planes of edge_feat.U.weight^T (kernels_wide_bf.hpp)
};
LayerBwdWs carve_bwd_ws(const Dims& d, Bump& b) {
    LayerBwdWs w;
    const size_t nh = d.Rn * d.H;
    // strip path: the edge product is part of the edge backward kernel, one slab per workgroup of that kernel
    w.S_e = d.fuse_wg ? edge_bwd_blocks(d) : tn_slices(d.Re, d.H);
    w.S_n = tn_slices(d.Rn, d.H);
    w.det = b.f(d.Re * d.H);
    w.bpart_e = b.f((size_t)kMaxPartBlocks * 2 * d.H);
    w.bpart_x = b.f((size_t)kMaxPartBlocks * 2 * d.H);
    w.bstats_e = b.f(2 * d.H);
    w.bstats_x = b.f(2 * d.H);
    w.dxtmp = b.f(nh);
    w.rowpart = b.f((size_t)d.njt * nh);
    w.colVe = b.f((size_t)d.nit * nh);
    w.colVn = b.f((size_t)d.nit * nh);
    w.dVe = b.f(nh);
    w.dVn = b.f(nh);
    const size_t slab = (size_t)d.H * d.H + d.H;
    w.tn_e = b.f((size_t)w.S_e * slab);
    w.tn_n = b.f((size_t)w.S_n * 3 * slab);
    w.wsplit = d.MT > 4 ? (u32x4*)b.raw(split_weight_bytes(d.H)) : nullptr;
    return w;
}

// ---- launches -------------------------------------------------------------------------------------
// What precedes a layer's kernels: nothing (inputs given), the previous layer (its BatchNorm apply is fused
// into this layer's loads) or the embedding.
struct PrevLayer {
    int kind;                       // 0 none, 1 previous layer, 2 embedding (edges only; nodes are given)
    const float *e_tmp, *e_in, *x_tmp, *x_in, *stats;
    const float *gamma_x, *beta_x, *gamma_e, *beta_e;
    int sid;                        // dropout stream base of that layer
    const float *cap, *we;
};

EdgeSrc make_edge_src(const PrevLayer& pv, const float* e_in, float* e_out, int H, float p, int training) {
    EdgeSrc s = {};
    if (pv.kind == 1) {
        s.mode = 1; s.prev_tmp = pv.e_tmp; s.prev_in = pv.e_in; s.e_out = e_out;
        s.pmean = pv.stats; s.prstd = pv.stats + H; s.pgamma = pv.gamma_e; s.pbeta = pv.beta_e;
        s.drop_sid = 2 * pv.sid; s.drop_p = p; s.drop_on = training;
    } else if (pv.kind == 2) {
        s.mode = 2; s.cap = pv.cap; s.we = pv.we; s.e_out = e_out;
    } else {
        s.mode = 0; s.e_in = e_in;
    }
    return s;
}

// bf16 planes of n H x H matrices (at most 64 per launch), optionally of their transposes
int launch_split_weights(const float* const* src, u32x4* const* dst, int n, int H, bool transposed, hipStream_t st) {
    const int frags = ((H + 15) / 16) * split_nkb(H);
    for (int i0 = 0; i0 < n; i0 += 64) {
        SplitList sl = {};
        const int m = std::min(64, n - i0);
        for (int k = 0; k < m; ++k) { sl.src[k] = src[i0 + k]; sl.dst[k] = dst[i0 + k]; }
        sl.transposed = transposed ? 1 : 0;
        GCN_PROF_PRE("k_split_weights");
        k_split_weights<<<dim3((frags + 3) / 4, m, 1), 256, 0, st>>>(sl, H);
        GCN_CHECK_LAUNCH("k_split_weights");
    }
    return 0;
}
// the edge kernels of kernels_wide_bf.hpp keep their step's rows in 3 SPB bytes of dynamic LDS: more than a kernel
// gets without asking (set per call: the attribute belongs to the current device)
template <typename K>
int reserve_split_lds(K kern, const char* name) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SPB) != hipSuccess)
        return fail(-3, "%s: cannot reserve %d bytes of LDS", name, 3 * SPB);
    return 0;
}

// forward of one layer up to the BatchNorm statistics (the apply is left to the consumer)
template <int MT>
int layer_fwd_core(const Dims& d, int agg, float p, int training, const unsigned long long* rng, const PrevLayer& pv,
                   const float* x_in, float* x_in_out, const float* e_in, float* e_in_out, const float* const* P,
                   const LayerSaved& S, const LayerFwdWs& W, hipStream_t st) {
    const int H = d.H;
    {
        NodeFwdArgs a = {};
        a.rows = (int)d.Rn; a.H = H; a.apply = pv.kind == 1;
        a.x_in = x_in; a.prev_tmp = pv.x_tmp; a.prev_in = pv.x_in; a.x_out = x_in_out;
        if (a.apply) { a.pmean = pv.stats + 2 * H; a.prstd = pv.stats + 3 * H; a.pgamma = pv.gamma_x; a.pbeta = pv.beta_x; }
        a.drop_sid = 2 * pv.sid + 1; a.drop_p = p; a.drop_on = training;
        a.W[0] = P[0]; a.bias[0] = P[1]; a.y[0] = W.Ux;
        a.W[1] = P[2]; a.bias[1] = P[3]; a.y[1] = S.Vn;
        a.W[2] = P[6]; a.bias[2] = P[7]; a.y[2] = W.Ve;
        a.rng = rng;
        if constexpr (MT == 0) GCN_LAUNCH_TIMED("k_node_fwd_w", k_node_fwd_w, wide_node_blocks((d.Rn + 15) / 16), 256, a);
        else GCN_LAUNCH_TIMED("k_node_fwd3", (k_node_fwd3<MT>), node_blocks((d.Rn + 15) / 16), 64 * MT, a);
    }
    {
        EdgeFwdArgs a = {};
        a.B = d.B; a.V = d.V; a.C = d.C; a.H = H; a.mean_agg = agg;
        a.src = make_edge_src(pv, e_in, e_in_out, H, p, training);
        a.WU = P[4]; a.bU = P[5]; a.Ve = W.Ve; a.Vn = S.Vn; a.Ux = W.Ux;
        a.e_tmp = S.e_tmp; a.x_tmp = S.x_tmp; a.Aout = S.A; a.Dout = S.D; a.partial = W.partial; a.rng = rng;
        if (const char* e = tune_env("GCN_TUNE_PROBE")) a.probe = atoi(e);
        const bool one = d.V <= 16;
        if constexpr (MT == 0) {
            a.WUs = W.wsplit;
            if (int rc = reserve_split_lds(k_edge_fwd_ws, "k_edge_fwd_ws")) return rc;
            GCN_LAUNCH_TIMED_LDS("k_edge_fwd_ws", k_edge_fwd_ws, W.nblk, SFT, 3 * SPB, a);
        } else {
            bool launched = false;
            if constexpr (MT == 2 || MT == 4) {
                if (d.two_blocks) {          // two column blocks per wavefront
                    switch (a.src.mode * 2 + (one ? 1 : 0)) {
                        case 0: GCN_LAUNCH_TIMED("k_edge_fwd4", (k_edge_fwd4<MT, 0, false>), W.nblk, 32 * MT, a); break;
                        case 1: GCN_LAUNCH_TIMED("k_edge_fwd4", (k_edge_fwd4<MT, 0, true>), W.nblk, 32 * MT, a); break;
                        case 2: GCN_LAUNCH_TIMED("k_edge_fwd4", (k_edge_fwd4<MT, 1, false>), W.nblk, 32 * MT, a); break;
                        case 3: GCN_LAUNCH_TIMED("k_edge_fwd4", (k_edge_fwd4<MT, 1, true>), W.nblk, 32 * MT, a); break;
                        case 4: GCN_LAUNCH_TIMED("k_edge_fwd4", (k_edge_fwd4<MT, 2, false>), W.nblk, 32 * MT, a); break;
                        default: GCN_LAUNCH_TIMED("k_edge_fwd4", (k_edge_fwd4<MT, 2, true>), W.nblk, 32 * MT, a); break;
                    }
                    launched = true;
                }
            }
            if (!launched) {
                switch (a.src.mode * 2 + (one ? 1 : 0)) {
                    case 0: GCN_LAUNCH_TIMED("k_edge_fwd3", (k_edge_fwd3<MT, 0, false>), W.nblk, 64 * MT, a); break;
                    case 1: GCN_LAUNCH_TIMED("k_edge_fwd3", (k_edge_fwd3<MT, 0, true>), W.nblk, 64 * MT, a); break;
                    case 2: GCN_LAUNCH_TIMED("k_edge_fwd3", (k_edge_fwd3<MT, 1, false>), W.nblk, 64 * MT, a); break;
                    case 3: GCN_LAUNCH_TIMED("k_edge_fwd3", (k_edge_fwd3<MT, 1, true>), W.nblk, 64 * MT, a); break;
                    case 4: GCN_LAUNCH_TIMED("k_edge_fwd3", (k_edge_fwd3<MT, 2, false>), W.nblk, 64 * MT, a); break;
                    default: GCN_LAUNCH_TIMED("k_edge_fwd3", (k_edge_fwd3<MT, 2, true>), W.nblk, 64 * MT, a); break;
                }
            }
        }
    }
    GCN_PROF_PRE("k_bn_stats_finalize");
    k_bn_stats_finalize<<<2 * (H / 4), GCN_BLOCK, 0, st>>>(W.partial, W.nblk, H, (double)d.Re, (double)d.Rn, S.stats);
    GCN_CHECK_LAUNCH("k_bn_stats_finalize");
    return 0;
}

// standalone layer: plain inputs, explicit apply kernels for both outputs
template <int MT>
int layer_fwd_t(const Dims& d, int agg, float p, int training, const unsigned long long* rng, int sid,
                const float* x_in, const float* e_in, const float* const* P, float* x_out, float* e_out,
                const LayerSaved& S, const LayerFwdWs& W, hipStream_t st) {
    const int H = d.H;
    PrevLayer none = {};
    if constexpr (MT == 0) {
        if (int rc = launch_split_weights(&P[4], &W.wsplit, 1, H, false, st)) return rc;
    }
    if (int rc = layer_fwd_core<MT>(d, agg, p, training, rng, none, x_in, nullptr, e_in, nullptr, P, S, W, st)) return rc;
    GCN_PROF_PRE("k_bn_apply(edges)");
    k_bn_apply<<<elem_blocks(d.Re * (H / 4)), 256, 0, st>>>(d.Re, H, S.e_tmp, e_in, S.stats, S.stats + H, P[10], P[11],
                                                            e_out, rng, 2 * sid, p, training);
    GCN_CHECK_LAUNCH("k_bn_apply(edges)");
    GCN_PROF_PRE("k_bn_apply(nodes)");
    k_bn_apply<<<elem_blocks(d.Rn * (H / 4)), 256, 0, st>>>(d.Rn, H, S.x_tmp, x_in, S.stats + 2 * H, S.stats + 3 * H,
                                                            P[8], P[9], x_out, rng, 2 * sid + 1, p, training);
    GCN_CHECK_LAUNCH("k_bn_apply(nodes)");
    return 0;
}

// BatchNorm-backward statistics of the layer below (fused into this layer's backward kernels)
struct NextLayer {
    bool on;
    const float *e_tmp, *x_tmp, *stats, *gamma_x, *beta_x, *gamma_e, *beta_e;
    int sid;
    float* const* G;                // gradient pointers of that layer (its BatchNorm gradients are finished here)
};

// Reduce jobs that finish a layer's BatchNorm backward: partial [nP][2][H] -> dbeta, dgamma and
// bstats = (mean(dbn), mean(dbn * xhat)) for the edge (which = 0) or node (1) BatchNorm.
void bn_finish_job(ReduceJob& j, const float* partial, int nP, int H, double n, float* dbeta, float* dgamma, float* bstats) {
    j.partial = partial; j.nP = nP; j.ncols = 2 * H;
    j.rd.n = 2;
    j.rd.dst[0] = dbeta; j.rd.dst[1] = dgamma;
    j.rd.dst2[0] = bstats; j.rd.dst2[1] = bstats + H;
    j.rd.scale2 = 1.0 / n;
    j.rd.off[0] = 0; j.rd.off[1] = H; j.rd.off[2] = 2 * H;
}

// edge kernel of a layer's backward data path
template <int MT>
int launch_edge_bwd(const Dims& d, int agg, float p, int training, const unsigned long long* rng, int sid,
                    const float* const* P, const float* const* wT, const LayerSaved& S, float* dx, float* de,
                    const float* e_in, float* tn_e, const LayerBwdWs& W, int& npart_e, const NextLayer& nx, hipStream_t st) {
    const int H = d.H;
    EdgeBwdArgs a = {};
    a.B = d.B; a.V = d.V; a.C = d.C; a.H = H; a.IT = d.IT; a.mean_agg = agg;
    a.de = de; a.e_tmp = S.e_tmp; a.WUt = wT[2];
    a.Vn = S.Vn; a.Aagg = S.A; a.Dn = S.D; a.dx = dx; a.x_tmp = S.x_tmp; a.stats = S.stats;
    a.gamma_e = P[10]; a.beta_e = P[11]; a.gamma_x = P[8]; a.beta_x = P[9];
    a.bstats_e = W.bstats_e; a.bstats_x = W.bstats_x;
    a.det_out = W.det; a.rowpart = W.rowpart; a.colVe = W.colVe; a.colVn = W.colVn;
    a.e_in = e_in; a.wg_out = tn_e;
    a.rng = rng; a.sid_e = 2 * sid; a.sid_x = 2 * sid + 1; a.p = p; a.training = training;
    if (nx.on) {
        a.next.tmp = nx.e_tmp; a.next.mean = nx.stats; a.next.rstd = nx.stats + H; a.next.gamma = nx.gamma_e;
        a.next.beta = nx.beta_e; a.next.drop_sid = 2 * nx.sid; a.next.partial = W.bpart_e;
    }
    const int blocks = edge_bwd_blocks(d);
    if constexpr (MT == 0) {
        a.WUs = W.wsplit;
        // a last column tile with at most 8 of its 16 columns in use goes to the kernel that packs several rows i into a tile
        const int njt = (d.V + 15) / 16, vlast = d.V - 16 * (njt - 1);
        const bool packed_last = vlast <= 8;
        int blocks_main = 0, blocks_last = 0;
        if (!packed_last || njt > 1) {
            a.jt_lo = 0; a.jt_n = packed_last ? njt - 1 : njt; a.part_off = 0;
            blocks_main = range_blocks((size_t)d.B * d.nit * a.jt_n * d.C, d.grid_b);
            if (const char* e = tune_env("GCN_TUNE_GRID_B")) blocks_main = range_blocks((size_t)d.B * d.nit * a.jt_n * d.C, std::min(atoi(e), kMaxPartBlocks / 2));
            if (int rc = reserve_split_lds(k_edge_bwd_ws, "k_edge_bwd_ws")) return rc;
            GCN_LAUNCH_TIMED_LDS("k_edge_bwd_ws", k_edge_bwd_ws, blocks_main, STH, 3 * SPB, a);
        }
        if (packed_last) {
            a.jt_lo = njt - 1; a.jt_n = 1; a.part_off = blocks_main;
            blocks_last = range_blocks((size_t)d.B * d.nit * d.C, d.grid_b);
            if (int rc = reserve_split_lds(k_edge_bwd_wp, "k_edge_bwd_wp")) return rc;
            GCN_LAUNCH_TIMED_LDS("k_edge_bwd_wp", k_edge_bwd_wp, blocks_last, STH, 3 * SPB, a);
        }
        if (nx.on) npart_e = blocks_main + blocks_last;
        return 0;
    } else {
        bool launched = false;
        if constexpr (MT == 2 || MT == 4) {
            if (d.two_blocks) {
                if (d.fuse_wg) GCN_LAUNCH_TIMED("k_edge_bwd4", (k_edge_bwd4<MT, true>), blocks, 32 * MT, a);
                else GCN_LAUNCH_TIMED("k_edge_bwd4", (k_edge_bwd4<MT, false>), blocks, 32 * MT, a);
                launched = true;
            }
        }
        if (!launched) GCN_LAUNCH_TIMED("k_edge_bwd3", (k_edge_bwd3<MT>), blocks, 64 * MT, a);
    }
    if (nx.on) npart_e = blocks;
    return 0;
}

// arguments of the node kernel of a layer's backward data path; dx_tmp / dVn / dVe go to the given buffers
NodeBwdArgs node_bwd_args(const Dims& d, float p, int training, const unsigned long long* rng, int sid, const float* const* P,
                          const float* const* wT, const LayerSaved& S, float* dx, const LayerBwdWs& W, float* dxtmp, float* dVn,
                          float* dVe, const NextLayer& nx) {
    const int H = d.H;
    NodeBwdArgs a = {};
    a.rows = (int)d.Rn; a.H = H; a.njt = d.njt; a.nit = d.nit;
    a.rowpart = W.rowpart; a.colVe = W.colVe; a.colVn = W.colVn; a.dx = dx; a.x_tmp = S.x_tmp;
    a.mean = S.stats + 2 * H; a.rstd = S.stats + 3 * H; a.gamma = P[8]; a.beta = P[9]; a.bstats = W.bstats_x;
    a.WnUt = wT[0]; a.WnVt = wT[1]; a.WeVt = wT[3];
    a.dxtmp_out = dxtmp; a.dVe_out = dVe; a.dVn_out = dVn;
    a.rng = rng; a.sid_x = 2 * sid + 1; a.p = p; a.training = training;
    if (nx.on) {
        a.next.tmp = nx.x_tmp; a.next.mean = nx.stats + 2 * H; a.next.rstd = nx.stats + 3 * H; a.next.gamma = nx.gamma_x;
        a.next.beta = nx.beta_x; a.next.drop_sid = 2 * nx.sid + 1; a.next.partial = W.bpart_x;
    }
    return a;
}

// Backward of one layer, data path: edge kernel, node kernel.  bstats_e / bstats_x and this layer's BatchNorm
// gradients must be finished (with fin_first they are reduced here from npart_e / npart_x partial rows).  On
// return, if nx.on, bpart_e / bpart_x hold npart_e / npart_x partial rows of the layer below.
// wT: transposed node_feat.U, node_feat.V, edge_feat.U, edge_feat.V weights of this layer.
template <int MT>
int layer_bwd_data(const Dims& d, int agg, float p, int training, const unsigned long long* rng, int sid,
                   const float* const* P, const float* const* wT, const LayerSaved& S,
                   float* dx, float* de, const float* e_in, float* tn_e, float* const* G, const LayerBwdWs& W, int& npart_e,
                   int& npart_x, const NextLayer& nx, bool fin_first, hipStream_t st) {
    const int H = d.H;
    if (fin_first) {
        ReduceJobs jobs = {};
        jobs.n = 2;
        bn_finish_job(jobs.job[0], W.bpart_e, npart_e, H, (double)d.Re, G[11], G[10], W.bstats_e);
        bn_finish_job(jobs.job[1], W.bpart_x, npart_x, H, (double)d.Rn, G[9], G[8], W.bstats_x);
        GCN_PROF_PRE("k_reduce_slabs(bn)");
        k_reduce_slabs<<<dim3(red_grid(jobs), 2, 1), GCN_BLOCK, 0, st>>>(jobs);
        GCN_CHECK_LAUNCH("k_reduce_slabs(bn)");
    }
    if (int rc = launch_edge_bwd<MT>(d, agg, p, training, rng, sid, P, wT, S, dx, de, e_in, tn_e, W, npart_e, nx, st)) return rc;
    {
        const int blocks = MT == 0 ? wide_node_blocks((d.Rn + 15) / 16, kMaxPartBlocks) : strip_blocks((d.Rn + 15) / 16, kMaxPartBlocks);
        const NodeBwdArgs a = node_bwd_args(d, p, training, rng, sid, P, wT, S, dx, W, W.dxtmp, W.dVn, W.dVe, nx);
        if constexpr (MT == 0) GCN_LAUNCH_TIMED("k_node_bwd_w", k_node_bwd_w, blocks, 256, a);
        else GCN_LAUNCH_TIMED("k_node_bwd3", (k_node_bwd3<MT>), blocks, 64 * MT, a);
        if (nx.on) npart_x = blocks;
    }
    return 0;
}

// the four weight-gradient products of a layer; partial slabs go to tn_e / tn_n
void layer_tn_problems(TnProblems& pr, const Dims& d, const LayerBwdWs& W, const float* x_in, const float* e_in, float* tn_e,
                       float* tn_n) {
    const size_t slab = (size_t)d.H * d.H + d.H;
    pr.X[0] = W.det; pr.Y[0] = e_in;     // edge_feat.U
    pr.X[1] = W.dxtmp; pr.Y[1] = x_in;   // node_feat.U
    pr.X[2] = W.dVn; pr.Y[2] = x_in;     // node_feat.V
    pr.X[3] = W.dVe; pr.Y[3] = x_in;     // edge_feat.V
    pr.rows[0] = d.Re; pr.nslices[0] = d.fuse_wg ? 0 : W.S_e; pr.out[0] = tn_e; pr.out_stride[0] = slab;   // fused: done by the edge kernel
    for (int k = 1; k < 4; ++k) {
        pr.rows[k] = d.Rn; pr.nslices[k] = W.S_n; pr.out[k] = tn_n + (size_t)(k - 1) * slab; pr.out_stride[k] = 3 * slab;
    }
}
// launch of weight-gradient products (nprob problems, at most max_slices row slices each)
template <int MT>
int launch_tn(int H, const TnProblems& pr, int nprob, int max_slices, hipStream_t st) {
    GCN_PROF_PRE("k_gemm_tn");
    if constexpr (MT == 0) {
        // 132 KiB (86 KiB) of LDS per workgroup: above the 64 KiB a kernel gets without asking (set per call: the attribute
        // belongs to the current device)
        if (H <= TS_MAX_H) {         // bf16 pieces on the matrix pipe; needs a free column of the padded H for the bias sums
            const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_ws),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS);
            if (attr != hipSuccess) return fail(-3, "k_gemm_tn_ws: cannot reserve %d bytes of LDS", TS_LDS);
            k_gemm_tn_ws<<<dim3(max_slices, 1, nprob), TS_TH, TS_LDS, st>>>(H, pr);
        } else {
            const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_w),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TnWideLds));
            if (attr != hipSuccess) return fail(-3, "k_gemm_tn_w: cannot reserve %zu bytes of LDS", sizeof(TnWideLds));
            k_gemm_tn_w<<<dim3(max_slices, 1, nprob), 512, sizeof(TnWideLds), st>>>(H, pr);
        }
    }
    else k_gemm_tn2<MT><<<dim3(max_slices, 1, nprob), 64 * MT, 0, st>>>(H, pr);
    GCN_CHECK_LAUNCH("k_gemm_tn");
    return 0;
}

// the two reduce jobs that turn those slabs into the layer's weight and bias gradients
void layer_wgrad_jobs(ReduceJob& je, ReduceJob& jn, const Dims& d, const LayerBwdWs& W, const float* tn_e, const float* tn_n,
                      float* const* G) {
    const int H = d.H, slab = H * H + H;
    je.partial = tn_e; je.nP = W.S_e; je.ncols = slab;
    je.rd.n = 2; je.rd.dst[0] = G[4]; je.rd.dst[1] = G[5];
    je.rd.off[0] = 0; je.rd.off[1] = H * H; je.rd.off[2] = slab;
    jn.partial = tn_n; jn.nP = W.S_n; jn.ncols = 3 * slab;
    ReduceDst& rd = jn.rd;
    rd.n = 6;
    rd.dst[0] = G[0]; rd.dst[1] = G[1]; rd.dst[2] = G[2]; rd.dst[3] = G[3]; rd.dst[4] = G[6]; rd.dst[5] = G[7];
    for (int k = 0; k < 3; ++k) { rd.off[2 * k] = k * slab; rd.off[2 * k + 1] = k * slab + H * H; }
    rd.off[6] = 3 * slab;
}

template <int MT>
int transpose_layer_weights(const float* const* P, float* wTbuf, const float** wT, int H, hipStream_t st) {
    TransposeList tl = {};
    const int idx[4] = {0, 2, 4, 6};     // node_feat.U, node_feat.V, edge_feat.U, edge_feat.V weights
    for (int k = 0; k < 4; ++k) { tl.src[k] = P[idx[k]]; tl.dst[k] = wTbuf + (size_t)k * H * H; wT[k] = tl.dst[k]; }
    GCN_PROF_PRE("k_transpose_weights");
    k_transpose_weights<<<4, GCN_BLOCK, 0, st>>>(tl, H);
    GCN_CHECK_LAUNCH("k_transpose_weights");
    return 0;
}

// standalone layer backward: own statistics passes, no layer below
template <int MT>
int layer_bwd_t(const Dims& d, int agg, float p, int training, const unsigned long long* rng, int sid,
                const float* x_in, const float* e_in, const float* const* P, const LayerSaved& S, float* dx,
                float* de, float* const* G, const LayerBwdWs& W, float* wTbuf, hipStream_t st) {
    const int H = d.H;
    const float* wT[4] = {P[0], P[2], P[4], P[6]};        // the strip kernels read the weights transposed themselves
    if constexpr (MT == 0) {
        if (int rc = transpose_layer_weights<MT>(P, wTbuf, wT, H, st)) return rc;
        if (int rc = launch_split_weights(&P[4], &W.wsplit, 1, H, true, st)) return rc;
    }
    const int nblk_e = wave_blocks((d.Re + 15) / 16), nblk_x = wave_blocks((d.Rn + 15) / 16);
    GCN_PROF_PRE("k_bn_bwd_stats(edges)");
    if constexpr (MT == 0)
        k_colsum_w<true><<<nblk_e, 256, 0, st>>>(d.Re, H, d.C, de, S.e_tmp, S.stats, S.stats + H, P[10], P[11], nullptr, W.bpart_e,
                                                 rng, 2 * sid, p, training);
    else
        k_bn_bwd_stats<MT><<<nblk_e, GCN_BLOCK, 0, st>>>(d.Re, H, de, S.e_tmp, S.stats, S.stats + H, P[10], P[11], W.bpart_e, rng,
                                                         2 * sid, p, training);
    GCN_CHECK_LAUNCH("k_bn_bwd_stats(edges)");
    GCN_PROF_PRE("k_bn_bwd_stats(nodes)");
    if constexpr (MT == 0)
        k_colsum_w<true><<<nblk_x, 256, 0, st>>>(d.Rn, H, d.C, dx, S.x_tmp, S.stats + 2 * H, S.stats + 3 * H, P[8], P[9], nullptr,
                                                 W.bpart_x, rng, 2 * sid + 1, p, training);
    else
        k_bn_bwd_stats<MT><<<nblk_x, GCN_BLOCK, 0, st>>>(d.Rn, H, dx, S.x_tmp, S.stats + 2 * H, S.stats + 3 * H, P[8], P[9],
                                                         W.bpart_x, rng, 2 * sid + 1, p, training);
    GCN_CHECK_LAUNCH("k_bn_bwd_stats(nodes)");
    int npe = nblk_e, npx = nblk_x;
    NextLayer none = {};
    if (int rc = layer_bwd_data<MT>(d, agg, p, training, rng, sid, P, wT, S, dx, de, e_in, W.tn_e, G, W, npe, npx, none, true, st)) return rc;
    TnProblems pr = {};
    layer_tn_problems(pr, d, W, x_in, e_in, W.tn_e, W.tn_n);
    if (int rc = launch_tn<MT>(H, pr, 4, std::max(pr.nslices[0], W.S_n), st)) return rc;
    ReduceJobs jobs = {};
    jobs.n = 2;
    layer_wgrad_jobs(jobs.job[0], jobs.job[1], d, W, W.tn_e, W.tn_n, G);
    GCN_PROF_PRE("k_reduce_slabs(layer)");
    k_reduce_slabs<<<dim3(red_grid(jobs), 2, 1), GCN_BLOCK, 0, st>>>(jobs);
    GCN_CHECK_LAUNCH("k_reduce_slabs(layer)");
    return 0;
}

#define DISPATCH_MT(MTV, CALL)                                   \
    switch (MTV) {                                               \
        case 1: { constexpr int MT = 1; CALL; } break;           \
        case 2: { constexpr int MT = 2; CALL; } break;           \
        case 3: { constexpr int MT = 3; CALL; } break;           \
        case 4: { constexpr int MT = 4; CALL; } break;           \
        default: { constexpr int MT = 0; CALL; } break;          /* hidden_dim > 64: the wide kernels */ \
    }

// ---- model-level plans -----------------------------------------------------------------------------
struct ModelPlan {
    Dims d;
    bool whead;
    int L, NHID, voc, nparams, n_emb;
    size_t x_act, e_act, lsaved;    // float offsets inside `saved`
    size_t lsaved_each, total_saved;
};
ModelPlan make_plan(const gcn_model_desc* m) {
    ModelPlan p;
    p.d = make_dims(m->B, m->V, m->C, m->H);
    p.L = m->L; p.NHID = m->mlp_layers - 1; p.voc = m->voc_out; p.n_emb = m->n_emb;
    // the workgroup-per-tile head kernels (one partial row per workgroup): k_head_*_w above H = 64, k_head_*3 up to
    // one hidden layer below it; deeper heads at H <= 64 use the wavefront-per-tile kernels
    p.whead = p.d.MT > 4 || (p.NHID <= 1 && tune_env("GCN_TUNE_OLDHEAD") == nullptr);
    p.nparams = 3 + GCN_LAYER_NPARAMS * m->L + 2 * m->mlp_layers;
    const size_t nh = p.d.Rn * p.d.H, eh = p.d.Re * p.d.H;
    p.lsaved_each = layer_saved_floats(p.d);
    p.x_act = 0;                                   // x_act[l] = node input of layer l, l = 0..L-1
    p.e_act = p.x_act + (size_t)p.L * nh;          // e_act[l] = edge input of layer l
    p.lsaved = p.e_act + (size_t)p.L * eh;
    p.total_saved = p.lsaved + (size_t)p.L * p.lsaved_each;
    return p;
}
int check_model(const gcn_model_desc* m) {
    if (m == nullptr) return fail(-1, "null model descriptor");
    if (int rc = check_dims(m->B, m->V, m->C, m->H)) return rc;
    if (m->L < 1) return fail(-1, "num_layers %d < 1", m->L);
    if (m->L > 64) return fail(-1, "num_layers %d > 64 is not supported by this build", m->L);
    if (m->mlp_layers < 1 || m->mlp_layers > GCN_MAX_MLP_LAYERS) return fail(-1, "mlp_layers %d outside 1..%d", m->mlp_layers, GCN_MAX_MLP_LAYERS);
    if (m->voc_out != 1 && m->voc_out != 2) return fail(-1, "voc_edges_out %d must be 1 or 2", m->voc_out);
    if (m->H % 2 != 0) return fail(-1, "hidden_dim must be even");
    if (m->n_emb < 1) return fail(-1, "n_emb %d < 1", m->n_emb);
    if (m->has_labels && m->voc_out != 2) return fail(-1, "loss_edges needs voc_edges_out == 2");
    return 0;
}

constexpr int kCountReps = 16;         // replicas of the edge-error counters the strip head spreads its increments over
struct ModelFwdWs {
    LayerFwdWs lw;
    float* loss_partial;
    int* count_rep;                 // [kCountReps][2][B*C]
    int head_blk;
    float* hid[2];                  // above 64 channels, MLPs with several hidden layers: outputs of the hidden layers before the last
    unsigned char* wsplit_all;      // above 64 channels: bf16 planes of every layer's edge_feat.U.weight
};
ModelFwdWs carve_model_fwd(const ModelPlan& p, Bump& b) {
    ModelFwdWs w;
    w.lw = carve_fwd_ws(p.d, b);
    w.head_blk = head_blocks((p.d.Re + 15) / 16, true, p.d.H, p.d.two_blocks);
    w.loss_partial = b.f((size_t)w.head_blk * GCN_WAVES_PER_BLOCK * 2);
    w.count_rep = (int*)b.raw((size_t)kCountReps * 2 * p.d.B * p.d.C * sizeof(int));
    const bool deep_wide = p.d.MT > 4 && p.NHID > 1;
    w.hid[0] = deep_wide ? b.f(p.d.Re * p.d.H) : nullptr;
    w.hid[1] = deep_wide && p.NHID > 2 ? b.f(p.d.Re * p.d.H) : nullptr;
    w.wsplit_all = p.d.MT > 4 ? (unsigned char*)b.raw((size_t)p.L * align_up(split_weight_bytes(p.d.H))) : nullptr;
    return w;
}
struct ModelBwdWs {
    LayerBwdWs lw;
    float *tn_e2, *tn_n2;           // second set of product slabs: layer l's products are written while layer l+1's are reduced
    float *nd2[3];                  // second set of dx_tmp / dVn / dVe: layer l's node backward writes while layer l+1's products read
    int* fin_counters;              // arrival counter of the in-kernel BatchNorm-backward finish (zero between launches)
    float *de, *dx, *eL, *dpre_stash, *h_stash, *wpartial, *tn_h, *embpart, *emb_g, *wT;
    unsigned char* wsplit_all;      // above 64 channels: bf16 planes of every layer's edge_feat.U.weight^T
    int* emb_cnt;
    int head_blk, S_h, emb_blk;
    bool head_fused;                // the first MLP weight gradient is formed inside the head kernel (strip path, one hidden layer)
};
ModelBwdWs carve_model_bwd(const ModelPlan& p, Bump& b) {
    ModelBwdWs w;
    const size_t eh = p.d.Re * p.d.H;
    w.lw = carve_bwd_ws(p.d, b);
    {
        const size_t slab = (size_t)p.d.H * p.d.H + p.d.H;
        w.tn_e2 = b.f((size_t)w.lw.S_e * slab);
        w.tn_n2 = b.f((size_t)w.lw.S_n * 3 * slab);
        for (int k = 0; k < 3; ++k) w.nd2[k] = b.f(p.d.Rn * p.d.H);
        w.fin_counters = (int*)b.raw(4 * sizeof(int));
    }
    w.de = b.f(eh);
    w.dx = b.f(p.d.Rn * p.d.H);
    w.head_fused = p.whead && p.NHID == 1 && p.d.MT >= 1 && p.d.MT <= 4;
    if (tune_env("GCN_TUNE_NOHEADFUSE")) w.head_fused = false;
    // output of the last conv layer, operand of the first MLP weight gradient (not stored when the head kernel forms that product)
    w.eL = (p.NHID > 0 && !w.head_fused) ? b.f(eh) : nullptr;
    // with one hidden layer the pre-activation stash aliases the layers' det buffer (the head finishes
    // before layer L-1 starts); deeper MLPs get NHID contiguous slabs of their own
    w.dpre_stash = p.NHID > 1 ? b.f((size_t)p.NHID * eh) : nullptr;
    w.h_stash = p.NHID > 1 ? b.f((size_t)(p.NHID - 1) * eh) : nullptr;
    w.head_blk = head_blocks((p.d.Re + 15) / 16, false, p.d.H, p.d.two_blocks);
    w.wpartial = b.f((size_t)w.head_blk * GCN_WAVES_PER_BLOCK * (size_t)((p.voc * p.d.H + p.voc + 3) & ~3));
    w.S_h = tn_slices(p.d.Re, p.d.H);
    w.tn_h = b.f((size_t)std::max(w.S_h, w.head_fused ? w.head_blk : 0) * std::max(1, p.NHID) * ((size_t)p.d.H * p.d.H + p.d.H));
    w.emb_blk = wave_blocks((p.d.Re + 15) / 16);
    w.embpart = b.f((size_t)w.emb_blk * GCN_WAVES_PER_BLOCK * p.d.H);
    w.emb_g = b.f((size_t)p.d.B * p.d.V * p.d.H);
    w.emb_cnt = (int*)b.raw((size_t)p.d.B * p.d.V * (size_t)p.n_emb * sizeof(int));
    w.wT = b.f((size_t)(4 * p.L + p.NHID) * p.d.H * p.d.H);
    w.wsplit_all = p.d.MT > 4 ? (unsigned char*)b.raw((size_t)p.L * align_up(split_weight_bytes(p.d.H))) : nullptr;
    return w;
}

template <int MT, int NHID>
void launch_head_fwd(const ModelPlan& p, const gcn_model_desc* m, const HeadParams& hp, const EdgeSrc& src, const float* cap,
                     const long long* y_edges, const float* edge_cw, float* y_pred, float* loss_partial, int* counts,
                     int count_reps, const unsigned long long* rng, int blocks, hipStream_t st) {
    if (p.whead) {
        if constexpr (NHID <= 1) {
            if constexpr (MT == 0)
                k_head_fwd_w<NHID><<<blocks, 256, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, src, cap, m->mask_invalid_edges,
                                                           m->has_labels ? y_edges : nullptr, edge_cw, y_pred,
                                                           m->has_labels ? loss_partial : nullptr, counts, rng, 1000, m->p_mlp, m->training);
            else {
                bool launched = false;
                if constexpr (MT == 2 || MT == 4) {
                    if (p.d.two_blocks) {        // two column blocks per wavefront (from 300,000 edge rows on, as the edge kernels)
                        k_head_fwd4<MT, NHID><<<blocks, 32 * MT, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, src, cap, m->mask_invalid_edges,
                                                                          m->has_labels ? y_edges : nullptr, edge_cw, y_pred,
                                                                          m->has_labels ? loss_partial : nullptr, counts, count_reps, rng,
                                                                          1000, m->p_mlp, m->training);
                        launched = true;
                    }
                }
                if (!launched)
                    k_head_fwd3<MT, NHID><<<blocks, 64 * MT, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, src, cap, m->mask_invalid_edges,
                                                                      m->has_labels ? y_edges : nullptr, edge_cw, y_pred,
                                                                      m->has_labels ? loss_partial : nullptr, counts, count_reps, rng, 1000,
                                                                      m->p_mlp, m->training);
            }
        }
    } else if constexpr (MT != 0)
    k_head_fwd<MT, NHID><<<blocks, GCN_BLOCK, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, src, cap, m->mask_invalid_edges,
                                                        m->has_labels ? y_edges : nullptr, edge_cw, y_pred,
                                                        m->has_labels ? loss_partial : nullptr, counts, rng, 1000, m->p_mlp,
                                                        m->training);
}
template <int MT, int NHID>
void launch_head_bwd(const ModelPlan& p, const gcn_model_desc* m, const HeadParams& hp, const EdgeSrc& src, const NextBn& next,
                     float* e_stash, const float* cap, const long long* y_edges, const float* edge_cw, const float* loss_out,
                     const float* dloss, const float* dy_pred, float* de, float* dpre, float* hst, float* wpartial,
                     const unsigned long long* rng, int blocks, const ZeroJobs& z, const float* hidWt, float* wg_out, hipStream_t st) {
    if (p.whead) {
        if constexpr (NHID <= 1) {
            if constexpr (MT == 0)
                k_head_bwd_w<NHID><<<blocks, 256, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, hidWt, src, next, e_stash, cap,
                                                           m->mask_invalid_edges, m->has_labels ? y_edges : nullptr, edge_cw, loss_out,
                                                           dloss, dy_pred, de, dpre, wpartial, rng, 1000, m->p_mlp, m->training, z);
            else
                k_head_bwd3<MT, NHID><<<blocks, 64 * MT, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, hidWt, src, next, e_stash, cap,
                                                                  m->mask_invalid_edges, m->has_labels ? y_edges : nullptr, edge_cw,
                                                                  loss_out, dloss, dy_pred, de, dpre, wpartial, rng, 1000, m->p_mlp,
                                                                  m->training, z, wg_out);
        }
    } else if constexpr (MT != 0)
    k_head_bwd<MT, NHID><<<blocks, GCN_BLOCK, 0, st>>>(p.d.Re, p.d.V, p.d.C, p.d.H, p.voc, hp, src, next, e_stash, cap,
                                                        m->mask_invalid_edges, m->has_labels ? y_edges : nullptr, edge_cw, loss_out,
                                                        dloss, dy_pred, de, dpre, hst, wpartial, rng, 1000, m->p_mlp, m->training, z);
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

PrevLayer prev_of(const ModelPlan& p, const float* saved, const float* const* P, int l) {
    // description of layer l as the producer of layer l+1's (or the head's) inputs
    const Dims& d = p.d;
    const size_t nh = d.Rn * d.H, eh = d.Re * d.H;
    const LayerSaved S = carve_saved(d, const_cast<float*>(saved) + p.lsaved + (size_t)l * p.lsaved_each);
    const float* const* LP = P + 3 + GCN_LAYER_NPARAMS * l;
    PrevLayer pv = {};
    pv.kind = 1;
    pv.e_tmp = S.e_tmp; pv.x_tmp = S.x_tmp; pv.stats = S.stats;
    pv.e_in = saved + p.e_act + (size_t)l * eh;
    pv.x_in = saved + p.x_act + (size_t)l * nh;
    pv.gamma_x = LP[8]; pv.beta_x = LP[9]; pv.gamma_e = LP[10]; pv.beta_e = LP[11];
    pv.sid = l;
    return pv;
}

}  // namespace

extern "C" {

int gcn_abi_version(void) { return GCN_ABI_VERSION; }
#ifdef GCN_PHASE_TRACE
extern "C" int gcn_phase_trace_read(unsigned long long* host_out /* [512] */) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_trace), 512 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef GCN_PHASE_ACC
extern "C" int gcn_phase_acc_read(unsigned long long* host_out /* [1024][12][8] */) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_acc), sizeof(unsigned long long) * 1024 * 12 * 8) == hipSuccess ? 0 : -1;
}
#endif
const char* gcn_last_error(void) { return g_err; }

static void prof_clear() {
    for (auto& e : g_prof.ev) { if (e.e0) (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); (void)hipEventDestroy(e.e2); }
    g_prof.ev.clear();
    g_prof.pending = nullptr;
}

int gcn_prof_begin(const char* kernel_substr) {
    if (!kernel_substr) return fail(-1, "gcn_prof_begin: null name");
    prof_clear();
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
        float ms = 0.f, empty = 0.f;
        if (hipEventSynchronize(e.e2) == hipSuccess && hipEventElapsedTime(&ms, e.e1, e.e2) == hipSuccess &&
            (e.e0 == nullptr || hipEventElapsedTime(&empty, e.e0, e.e1) == hipSuccess)) {
            tot += (double)ms - (double)empty;       // the event bracket's own cost, measured beside every launch
            ++n;
        }
    }
    prof_clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return 0;
}

int gcn_layer_plan(int B, int V, int C, int H, int out[8]) {
    if (int rc = check_dims(B, V, C, H)) return rc;
    if (!out) return fail(-1, "gcn_layer_plan: null output");
    const Dims d = make_dims(B, V, C, H);
    Bump f(nullptr, 0);
    const LayerFwdWs fw = carve_fwd_ws(d, f);
    const size_t units = (size_t)d.B * d.nit * d.njt * d.C;
    out[0] = d.IT; out[1] = d.nit; out[2] = d.njt; out[3] = fw.nblk; out[4] = range_blocks(units, d.grid_b);
    out[5] = d.MT > 4 ? 1 : 0; out[6] = d.two_blocks ? 1 : 0;
    out[7] = (d.MT > 4 && V - 16 * (d.njt - 1) <= 8) ? 1 : 0;       // the last column tile runs in k_edge_bwd_wp
    if (out[7]) out[4] = d.njt > 1 ? range_blocks((size_t)d.B * d.nit * (d.njt - 1) * d.C, d.grid_b) : 0;
    return 0;
}

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
    b.f((size_t)4 * H * H);
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
    float* wTbuf = b.f((size_t)4 * H * H);
    if (!b.ok()) return fail(-2, "gcn_layer_bwd: workspace too small (%zu < %zu bytes)", ws_bytes, b.off);
    const LayerSaved S = carve_saved(d, const_cast<float*>(saved));
    DISPATCH_MT(d.MT, return layer_bwd_t<MT>(d, aggregation, p_drop, training, rng, stream_id, x_in, e_in, params, S, dx, de,
                                             grads, W, wTbuf, (hipStream_t)stream));
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

int gcn_model_saved_layout(const gcn_model_desc* m, int layer, size_t out[8]) {
    if (int rc = check_model(m)) return rc;
    if (!out) return fail(-1, "gcn_model_saved_layout: null output");
    if (layer < 0 || layer >= m->L) return fail(-1, "gcn_model_saved_layout: layer %d outside 0..%d", layer, m->L - 1);
    const ModelPlan p = make_plan(m);
    const size_t nh = p.d.Rn * p.d.H, eh = p.d.Re * p.d.H;
    const size_t base = p.lsaved + (size_t)layer * p.lsaved_each;
    out[0] = base;                       // the order of carve_saved: e_tmp, x_tmp, A, D, Vn, stats
    out[1] = base + eh;
    out[2] = base + eh + 4 * nh;
    out[3] = p.e_act + (size_t)layer * eh;
    out[4] = p.x_act + (size_t)layer * nh;
    out[5] = 0; out[6] = 0; out[7] = 0;
    return 0;
}

int gcn_bn_relu_pattern(size_t rows, int H, const float* tmp, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, unsigned char* mask, void* stream) {
    if (rows == 0 || H <= 0 || H % 4 != 0) return fail(-1, "gcn_bn_relu_pattern: rows and H must be positive, H a multiple of 4");
    if (!tmp || !mean || !rstd || !gamma || !beta || !mask) return fail(-1, "gcn_bn_relu_pattern: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    GCN_PROF_PRE("k_bn_relu_pattern");
    k_bn_relu_pattern<<<elem_blocks(rows * (size_t)(H / 4)), 256, 0, st>>>(rows, H, tmp, mean, rstd, gamma, beta, mask);
    GCN_CHECK_LAUNCH("k_bn_relu_pattern");
    return 0;
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
    // the per-(graph, commodity) counters of edge_error start at zero; the strip head (H <= 64, at most one hidden
    // layer) counts into replicas in the workspace, the other heads into the caller's err_counts
    const bool rep = p.whead && d.MT <= 4;
    int* counts = rep ? W.count_rep : err_counts;
    const int nrep = rep ? kCountReps : 1;
    ZeroJobs ez = {};
    if (m->has_labels) { ez.p[0] = reinterpret_cast<float*>(counts); ez.n[0] = (size_t)nrep * 2 * d.B * d.C; }
    GCN_PROF_PRE("k_embed_nodes");
    k_embed_nodes<<<d.B * d.V, 64, 0, st>>>(d.V, d.C, H, m->n_emb, x_nodes, demand, P[0], P[1], x_act, ez,
                                            (m->advance_rng && rng) ? const_cast<unsigned long long*>(rng) : nullptr);
    GCN_CHECK_LAUNCH("k_embed_nodes");
    const size_t wsplit_stride = align_up(split_weight_bytes(H));
    if (d.MT > 4) {
        std::vector<const float*> src;
        std::vector<u32x4*> dst;
        for (int l = 0; l < p.L; ++l) { src.push_back(P[3 + GCN_LAYER_NPARAMS * l + 4]); dst.push_back((u32x4*)(W.wsplit_all + (size_t)l * wsplit_stride)); }
        if (int rc = launch_split_weights(src.data(), dst.data(), p.L, H, false, st)) return rc;
    }
    for (int l = 0; l < p.L; ++l) {
        const LayerSaved S = carve_saved(d, saved + p.lsaved + (size_t)l * p.lsaved_each);
        const float* const* LP = P + 3 + GCN_LAYER_NPARAMS * l;
        LayerFwdWs lw = W.lw;
        if (d.MT > 4) lw.wsplit = (u32x4*)(W.wsplit_all + (size_t)l * wsplit_stride);
        PrevLayer pv = {};
        if (l == 0) { pv.kind = 2; pv.cap = cap; pv.we = P[2]; }
        else pv = prev_of(p, saved, P, l - 1);
        int rc = 0;
        // layer 0: x_act[0] was written by the embedding; later layers write their own input while applying layer l-1
        DISPATCH_MT(d.MT, rc = layer_fwd_core<MT>(d, m->aggregation, m->p_layer, m->training, rng, pv, x_act + (size_t)l * nh,
                                                  l == 0 ? nullptr : x_act + (size_t)l * nh, nullptr, e_act + (size_t)l * eh, LP, S,
                                                  lw, st));
        if (rc) return rc;
    }
    const HeadParams hp = head_params(p, P);
    const PrevLayer last = prev_of(p, saved, P, p.L - 1);
    const EdgeSrc src = make_edge_src(last, nullptr, nullptr, H, m->p_layer, m->training);
    if (d.MT > 4 && p.NHID > 1) {
        // hidden layers 0 .. NHID-2 one launch each, then the fused head on the last hidden layer and the output layer
        EdgeSrc cur = src;
        for (int k = 0; k + 1 < p.NHID; ++k) {
            float* out = W.hid[k & 1];
            GCN_PROF_PRE("k_mlp_hidden_fwd_w");
            k_mlp_hidden_fwd_w<<<W.head_blk, 256, 0, st>>>(d.Re, d.C, H, hp.hidW[k], hp.hidB[k], cur, nullptr, out, rng, 1000 + k,
                                                           m->p_mlp, m->training);
            GCN_CHECK_LAUNCH("k_mlp_hidden_fwd_w");
            cur = EdgeSrc{};
            cur.mode = 0; cur.e_in = out;
        }
        HeadParams hl = {};
        hl.hidW[0] = hp.hidW[p.NHID - 1]; hl.hidB[0] = hp.hidB[p.NHID - 1]; hl.outW = hp.outW; hl.outB = hp.outB;
        GCN_PROF_PRE("k_head_fwd");
        k_head_fwd_w<1><<<W.head_blk, 256, 0, st>>>(d.Re, d.V, d.C, H, p.voc, hl, cur, cap, m->mask_invalid_edges,
                                                    m->has_labels ? y_edges : nullptr, edge_cw, y_pred,
                                                    m->has_labels ? W.loss_partial : nullptr, counts, rng, 1000 + p.NHID - 1,
                                                    m->p_mlp, m->training);
        GCN_CHECK_LAUNCH("k_head_fwd");
    } else {
        GCN_PROF_PRE("k_head_fwd");
        DISPATCH_MT(d.MT, DISPATCH_NHID(p.NHID, (launch_head_fwd<MT, NHID>(p, m, hp, src, cap, y_edges, edge_cw, y_pred,
                                                                          W.loss_partial, counts, nrep, rng, W.head_blk, st))));
        GCN_CHECK_LAUNCH("k_head_fwd");
    }
    if (m->has_labels) {
        GCN_PROF_PRE("k_loss_finalize");
        k_loss_finalize<<<1, GCN_BLOCK, 0, st>>>(W.loss_partial, p.whead ? W.head_blk : W.head_blk * GCN_WAVES_PER_BLOCK, counts, d.B * d.C, nrep,
                                                 loss_out, err_flags);
        GCN_CHECK_LAUNCH("k_loss_finalize");
    }
    return 0;
}

int gcn_model_bwd(const gcn_model_desc* m, const float* cap, const float* demand, const long long* x_nodes,
                  const long long* y_edges, const float* edge_cw, const float* const* P,
                  const unsigned long long* rng, const float* saved, const float* loss_out, const float* dloss,
                  const float* dy_pred, float* const* G, void* ws, size_t ws_bytes, void* stream) {
    return gcn_model_bwd_ev(m, cap, demand, x_nodes, y_edges, edge_cw, P, rng, saved, loss_out, dloss, dy_pred, G, ws, ws_bytes,
                            stream, nullptr);
}

int gcn_model_bwd_ev(const gcn_model_desc* m, const float* cap, const float* demand, const long long* x_nodes,
                     const long long* y_edges, const float* edge_cw, const float* const* P,
                     const unsigned long long* rng, const float* saved, const float* loss_out, const float* dloss,
                     const float* dy_pred, float* const* G, void* ws, size_t ws_bytes, void* stream, void* const* done_events) {
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
    // done_events[k] (may be null, as may the array): recorded on the stream once every gradient of conv layer k
    // (k < L) or of the edge MLP (k = L) is final
    auto mark_done = [&](int k) -> int {
        if (done_events != nullptr && done_events[k] != nullptr && hipEventRecord((hipEvent_t)done_events[k], st) != hipSuccess)
            return fail(-3, "gcn_model_bwd_ev: cannot record the completion event of part %d", k);
        return 0;
    };
    const int H = d.H;
    const size_t nh = d.Rn * H, eh = d.Re * H, hh2 = (size_t)H * H;
    const float* x_act = saved + p.x_act;
    const float* e_act = saved + p.e_act;
    const HeadParams hp = head_params(p, P);
    const int hbase = 3 + GCN_LAYER_NPARAMS * p.L;
    const int slab = H * H + H;
    const size_t wsplit_stride = align_up(split_weight_bytes(H));
    // wide kernels only: transposed copies of every H x H weight used in an "X * W" product of the backward pass
    // (64 per launch); the strip kernels read the stored weights transposed
    if (d.MT > 4) {
        std::vector<std::pair<const float*, float*>> tr;
        const int idx[4] = {0, 2, 4, 6};
        for (int l = 0; l < p.L; ++l)
            for (int k = 0; k < 4; ++k) tr.emplace_back(P[3 + GCN_LAYER_NPARAMS * l + idx[k]], W.wT + (size_t)(4 * l + k) * hh2);
        for (int k = 0; k < p.NHID; ++k) tr.emplace_back(hp.hidW[k], W.wT + (size_t)(4 * p.L + k) * hh2);   // the head reads W_k^T
        for (size_t i0 = 0; i0 < tr.size(); i0 += 64) {
            TransposeList tl = {};
            const int n = (int)std::min<size_t>(64, tr.size() - i0);
            for (int k = 0; k < n; ++k) { tl.src[k] = tr[i0 + k].first; tl.dst[k] = tr[i0 + k].second; }
            GCN_PROF_PRE("k_transpose_weights");
            k_transpose_weights<<<n, GCN_BLOCK, 0, st>>>(tl, H);
            GCN_CHECK_LAUNCH("k_transpose_weights");
        }
        std::vector<const float*> src;
        std::vector<u32x4*> dst;
        for (int l = 0; l < p.L; ++l) { src.push_back(P[3 + GCN_LAYER_NPARAMS * l + 4]); dst.push_back((u32x4*)(W.wsplit_all + (size_t)l * wsplit_stride)); }
        if (int rc = launch_split_weights(src.data(), dst.data(), p.L, H, true, st)) return rc;
    }
    float* dpre_all = p.NHID == 1 ? W.lw.det : W.dpre_stash;   // [NHID][Re*H], see carve_model_bwd
    const PrevLayer last = prev_of(p, saved, P, p.L - 1);
    const EdgeSrc src = make_edge_src(last, nullptr, nullptr, H, m->p_layer, m->training);
    NextBn hnext = {};
    hnext.tmp = last.e_tmp; hnext.mean = last.stats; hnext.rstd = last.stats + H; hnext.gamma = last.gamma_e; hnext.beta = last.beta_e;
    hnext.drop_sid = 2 * last.sid; hnext.partial = W.lw.bpart_e;
    // the node output of the last layer feeds nothing (gcn_model.py:89 uses e only): its gradient and the
    // BatchNorm-backward sums derived from it are zero (cleared by the head kernel on the side)
    ZeroJobs hz = {};
    hz.p[0] = W.dx; hz.n[0] = nh;
    hz.p[1] = W.lw.bpart_x; hz.n[1] = 2 * (size_t)H;
    hz.p[2] = reinterpret_cast<float*>(W.fin_counters); hz.n[2] = 4;
    if (d.MT > 4 && p.NHID > 1) {
        // (1) the hidden layers before the last once more (h_stash[k] = output of hidden layer k; the first also
        //     stores the head's input), (2) the fused head backward on the last hidden layer and the output layer,
        //     (3) the earlier hidden layers backwards, in place in de; the first one adds the BatchNorm-backward sums
        //     of the last conv layer
        EdgeSrc cur = src;
        for (int k = 0; k + 1 < p.NHID; ++k) {
            float* out = W.h_stash + (size_t)k * eh;
            GCN_PROF_PRE("k_mlp_hidden_fwd_w");
            k_mlp_hidden_fwd_w<<<W.head_blk, 256, 0, st>>>(d.Re, d.C, H, hp.hidW[k], hp.hidB[k], cur, k == 0 ? W.eL : nullptr, out, rng,
                                                           1000 + k, m->p_mlp, m->training);
            GCN_CHECK_LAUNCH("k_mlp_hidden_fwd_w");
            cur = EdgeSrc{};
            cur.mode = 0; cur.e_in = out;
        }
        HeadParams hl = {};
        hl.hidW[0] = hp.hidW[p.NHID - 1]; hl.hidB[0] = hp.hidB[p.NHID - 1]; hl.outW = hp.outW; hl.outB = hp.outB;
        GCN_PROF_PRE("k_head_bwd");
        k_head_bwd_w<1><<<W.head_blk, 256, 0, st>>>(d.Re, d.V, d.C, H, p.voc, hl, W.wT + (size_t)(4 * p.L + p.NHID - 1) * hh2, cur,
                                                    NextBn{}, nullptr, cap, m->mask_invalid_edges, m->has_labels ? y_edges : nullptr,
                                                    edge_cw, loss_out, dloss, dy_pred, W.de, dpre_all + (size_t)(p.NHID - 1) * eh,
                                                    W.wpartial, rng, 1000 + p.NHID - 1, m->p_mlp, m->training, hz);
        GCN_CHECK_LAUNCH("k_head_bwd");
        for (int k = p.NHID - 2; k >= 0; --k) {
            NextBn nb = {};
            if (k == 0) { nb = hnext; nb.drop_sid = src.drop_sid; }
            GCN_PROF_PRE("k_mlp_hidden_bwd_w");
            k_mlp_hidden_bwd_w<<<W.head_blk, 256, 0, st>>>(d.Re, H, W.wT + (size_t)(4 * p.L + k) * hh2, W.h_stash + (size_t)k * eh, W.de,
                                                           dpre_all + (size_t)k * eh, nb, src.drop_p, src.drop_on, rng, 1000 + k,
                                                           m->p_mlp, m->training);
            GCN_CHECK_LAUNCH("k_mlp_hidden_bwd_w");
        }
    } else {
        GCN_PROF_PRE("k_head_bwd");
        // strip path with one hidden layer: its weight-gradient product is part of the head kernel (one slab per workgroup
        // in tn_h) and neither the head's input nor the pre-activation gradient is stored
        DISPATCH_MT(d.MT, DISPATCH_NHID(p.NHID, (launch_head_bwd<MT, NHID>(p, m, hp, src, hnext, W.head_fused ? nullptr : W.eL, cap, y_edges,
                                                                          edge_cw, loss_out, dloss, dy_pred, W.de,
                                                                          W.head_fused ? nullptr : dpre_all, W.h_stash, W.wpartial, rng,
                                                                          W.head_blk, hz, W.wT + (size_t)4 * p.L * hh2,
                                                                          W.head_fused ? W.tn_h : nullptr, st))));
        GCN_CHECK_LAUNCH("k_head_bwd");
    }
    int npart_e = W.head_blk, npart_x = 1;
    // hidden-layer weight-gradient products of the head and the reduce jobs that are ready now: output-layer
    // gradients and the BatchNorm backward of the top layer.  The slabs of the head's products are reduced by the
    // top layer's k_wgrad launch (H <= 64) or right here (wider).
    TnProblems hpr = {};
    for (int k = 0; k < p.NHID; ++k) {
        hpr.X[k] = dpre_all + (size_t)k * eh;
        hpr.Y[k] = k == 0 ? W.eL : W.h_stash + (size_t)(k - 1) * eh;
        hpr.rows[k] = d.Re; hpr.nslices[k] = W.head_fused ? 0 : W.S_h; hpr.out[k] = W.tn_h + (size_t)k * slab; hpr.out_stride[k] = (size_t)p.NHID * slab;
    }
    ReduceJob head_tn_job = {};
    bool head_tn_pending = false;
    if (p.NHID > 0) {
        head_tn_job.partial = W.tn_h; head_tn_job.nP = W.head_fused ? W.head_blk : W.S_h; head_tn_job.ncols = p.NHID * slab;
        ReduceDst& rd = head_tn_job.rd;
        rd.n = 2 * p.NHID;
        for (int k = 0; k < p.NHID; ++k) {
            rd.dst[2 * k] = G[hbase + 2 * k]; rd.dst[2 * k + 1] = G[hbase + 2 * k + 1];
            rd.off[2 * k] = k * slab; rd.off[2 * k + 1] = k * slab + H * H;
        }
        rd.off[2 * p.NHID] = p.NHID * slab;
    }
    {
        const int wcols = (p.voc * H + p.voc + 3) & ~3;   // row stride of wpartial
        ReduceJobs jobs = {};
        jobs.job[0].partial = W.wpartial; jobs.job[0].nP = p.whead ? W.head_blk : W.head_blk * GCN_WAVES_PER_BLOCK; jobs.job[0].ncols = wcols;
        jobs.job[0].rd.n = 2; jobs.job[0].rd.dst[0] = G[hbase + 2 * p.NHID]; jobs.job[0].rd.dst[1] = G[hbase + 2 * p.NHID + 1];
        jobs.job[0].rd.off[0] = 0; jobs.job[0].rd.off[1] = p.voc * H; jobs.job[0].rd.off[2] = p.voc * H + p.voc;
        jobs.n = 1;
        float* const* GL = G + 3 + GCN_LAYER_NPARAMS * (p.L - 1);
        bn_finish_job(jobs.job[jobs.n++], W.lw.bpart_e, npart_e, H, (double)d.Re, GL[11], GL[10], W.lw.bstats_e);
        bn_finish_job(jobs.job[jobs.n++], W.lw.bpart_x, npart_x, H, (double)d.Rn, GL[9], GL[8], W.lw.bstats_x);
        if (d.MT <= 4) {
            WgradMap map = {};
            int nb = 0;
            for (int k = 0; k < 4; ++k) { map.seg[k] = nb; nb += k < p.NHID ? hpr.nslices[k] : 0; }
            for (int k = 0; k < 4; ++k) { map.seg[4 + k] = nb; if (k < jobs.n) nb += red_blocks(jobs.job[k]); }
            map.seg[8] = nb;
            DISPATCH_MT(d.MT, if constexpr (MT != 0) GCN_LAUNCH_TIMED("k_wgrad(head)", (k_wgrad<MT>), nb, GCN_BLOCK, H, hpr, jobs, map));
            head_tn_pending = p.NHID > 0;
        } else {
            if (p.NHID > 0) {
                if (int rc = launch_tn<0>(H, hpr, p.NHID, W.S_h, st)) return rc;
                jobs.job[jobs.n++] = head_tn_job;
            }
            GCN_PROF_PRE("k_reduce_slabs(head)");
            k_reduce_slabs<<<dim3(red_grid(jobs), jobs.n, 1), GCN_BLOCK, 0, st>>>(jobs);
            GCN_CHECK_LAUNCH("k_reduce_slabs(head)");
            if (int rc = mark_done(p.L)) return rc;          // the edge MLP's gradients are final
        }
    }
    if (d.MT > 4) {
        for (int l = p.L - 1; l >= 0; --l) {
            const LayerSaved S = carve_saved(d, const_cast<float*>(saved) + p.lsaved + (size_t)l * p.lsaved_each);
            const float* const* LP = P + 3 + GCN_LAYER_NPARAMS * l;
            float* const* LG = G + 3 + GCN_LAYER_NPARAMS * l;
            const float* wT[4];
            for (int k = 0; k < 4; ++k) wT[k] = W.wT + (size_t)(4 * l + k) * hh2;
            NextLayer nx = {};
            if (l > 0) {
                const PrevLayer below = prev_of(p, saved, P, l - 1);
                nx.on = true; nx.e_tmp = below.e_tmp; nx.x_tmp = below.x_tmp; nx.stats = below.stats;
                nx.gamma_x = below.gamma_x; nx.beta_x = below.beta_x; nx.gamma_e = below.gamma_e; nx.beta_e = below.beta_e; nx.sid = l - 1;
                nx.G = G + 3 + GCN_LAYER_NPARAMS * (l - 1);
            }
            int rc = 0;
            LayerBwdWs lw = W.lw;
            lw.wsplit = (u32x4*)(W.wsplit_all + (size_t)l * wsplit_stride);
            DISPATCH_MT(d.MT, rc = layer_bwd_data<MT>(d, m->aggregation, m->p_layer, m->training, rng, l, LP, wT, S, W.dx, W.de,
                                                      e_act + (size_t)l * eh, nullptr, LG, lw, npart_e, npart_x, nx, false, st));
            if (rc) return rc;
            // one launch: this layer's weight-gradient products, the reduction of the layer above's products and the
            // BatchNorm-backward finish of the layer below (its partial sums were just written)
            float* tn_e = (l & 1) ? W.tn_e2 : W.lw.tn_e;
            float* tn_n = (l & 1) ? W.tn_n2 : W.lw.tn_n;
            TnProblems pr = {};
            layer_tn_problems(pr, d, W.lw, x_act + (size_t)l * nh, e_act + (size_t)l * eh, tn_e, tn_n);
            ReduceJobs jobs = {};
            WgradMap map = {};
            int nb = 0;
            for (int k = 0; k < 4; ++k) { map.seg[k] = nb; nb += pr.nslices[k]; }
            int nj = 0;
            if (l + 1 < p.L) {
                float* const* UG = G + 3 + GCN_LAYER_NPARAMS * (l + 1);
                layer_wgrad_jobs(jobs.job[0], jobs.job[1], d, W.lw, (l & 1) ? W.lw.tn_e : W.tn_e2, (l & 1) ? W.lw.tn_n : W.tn_n2, UG);
                nj = 2;
            }
            if (nx.on) {
                bn_finish_job(jobs.job[nj], W.lw.bpart_e, npart_e, H, (double)d.Re, nx.G[11], nx.G[10], W.lw.bstats_e);
                bn_finish_job(jobs.job[nj + 1], W.lw.bpart_x, npart_x, H, (double)d.Rn, nx.G[9], nx.G[8], W.lw.bstats_x);
                nj += 2;
            }
            if (head_tn_pending) { jobs.job[nj++] = head_tn_job; head_tn_pending = false; }    // only in the top layer's launch: nj <= 3
            jobs.n = nj;
            for (int k = 0; k < 4; ++k) { map.seg[4 + k] = nb; if (k < nj) nb += red_blocks(jobs.job[k]); }
            map.seg[8] = nb;
            if (d.MT > 4) {
                // wide path: the products and the ready reduce jobs as two launches
                if (int rc2 = launch_tn<0>(H, pr, 4, std::max(W.lw.S_e, W.lw.S_n), st)) return rc2;
                if (nj > 0) {
                    GCN_PROF_PRE("k_reduce_slabs(layer)");
                    k_reduce_slabs<<<dim3(red_grid(jobs), nj, 1), GCN_BLOCK, 0, st>>>(jobs);
                    GCN_CHECK_LAUNCH("k_reduce_slabs(layer)");
                }
                // layer l + 1 is final now: its products were just reduced, its BatchNorm gradients one launch earlier
                if (l + 1 < p.L) { if (int rc3 = mark_done(l + 1)) return rc3; }
            } else {
                DISPATCH_MT(d.MT, if constexpr (MT != 0) GCN_LAUNCH_TIMED("k_wgrad", (k_wgrad<MT>), nb, GCN_BLOCK, H, pr, jobs, map));
            }
        }
    } else {
        // H <= 64: two launches per layer.  After the edge kernel of layer l, one launch runs the node kernel of layer l
        // (which also finishes the BatchNorm backward of layer l-1's node BatchNorm) next to the products and
        // reductions that are ready: the edge product of layer l, the node products of layer l+1 (their operands
        // come from layer l+1's node kernel, hence the second set of node buffers), the slabs written by the
        // launch before, and the BatchNorm-backward sums the edge kernel just produced.
        auto ndbuf = [&](int l, int k) { return (l & 1) ? W.nd2[k] : (k == 0 ? W.lw.dxtmp : k == 1 ? W.lw.dVn : W.lw.dVe); };
        auto tn_e_of = [&](int l) { return (l & 1) ? W.tn_e2 : W.lw.tn_e; };
        auto tn_n_of = [&](int l) { return (l & 1) ? W.tn_n2 : W.lw.tn_n; };
        auto node_problems = [&](TnProblems& pr, int slot0, int l) {       // node_feat.U, node_feat.V, edge_feat.V of layer l
            for (int k = 0; k < 3; ++k) {
                pr.X[slot0 + k] = ndbuf(l, k); pr.Y[slot0 + k] = x_act + (size_t)l * nh;
                pr.rows[slot0 + k] = d.Rn; pr.nslices[slot0 + k] = W.lw.S_n;
                pr.out[slot0 + k] = tn_n_of(l) + (size_t)k * slab; pr.out_stride[slot0 + k] = 3 * slab;
            }
        };
        for (int l = p.L - 1; l >= 0; --l) {
            const LayerSaved S = carve_saved(d, const_cast<float*>(saved) + p.lsaved + (size_t)l * p.lsaved_each);
            const float* const* LP = P + 3 + GCN_LAYER_NPARAMS * l;
            const float* wT[4] = {LP[0], LP[2], LP[4], LP[6]};      // as stored: node_feat.U, node_feat.V, edge_feat.U, edge_feat.V
            NextLayer nx = {};
            if (l > 0) {
                const PrevLayer below = prev_of(p, saved, P, l - 1);
                nx.on = true; nx.e_tmp = below.e_tmp; nx.x_tmp = below.x_tmp; nx.stats = below.stats;
                nx.gamma_x = below.gamma_x; nx.beta_x = below.beta_x; nx.gamma_e = below.gamma_e; nx.beta_e = below.beta_e; nx.sid = l - 1;
                nx.G = G + 3 + GCN_LAYER_NPARAMS * (l - 1);
            }
            int rc = 0;
            DISPATCH_MT(d.MT, if constexpr (MT != 0) rc = launch_edge_bwd<MT>(d, m->aggregation, m->p_layer, m->training, rng, l, LP, wT, S,
                                                                              W.dx, W.de, e_act + (size_t)l * eh, tn_e_of(l), W.lw,
                                                                              npart_e, nx, st));
            if (rc) return rc;
            NodeBwdArgs na = node_bwd_args(d, m->p_layer, m->training, rng, l, LP, wT, S, W.dx, W.lw, ndbuf(l, 0), ndbuf(l, 1),
                                           ndbuf(l, 2), nx);
            if (nx.on) {
                na.fin.counters = W.fin_counters;
                na.fin.dbeta = nx.G[9]; na.fin.dgamma = nx.G[8]; na.fin.bstats = W.lw.bstats_x; na.fin.inv_n = 1.0 / (double)d.Rn;
            }
            TnProblems pr = {};
            if (!d.fuse_wg) {                 // fused: slot 0 stays empty, the edge product of layer l was part of its edge kernel
                pr.X[0] = W.lw.det; pr.Y[0] = e_act + (size_t)l * eh; pr.rows[0] = d.Re; pr.nslices[0] = W.lw.S_e;
                pr.out[0] = tn_e_of(l); pr.out_stride[0] = slab;
            }
            if (l + 1 < p.L) node_problems(pr, 1, l + 1);
            ReduceJobs jobs = {};
            ReduceJob unused = {};
            int nj = 0;
            if (l + 1 < p.L) layer_wgrad_jobs(jobs.job[nj++], unused, d, W.lw, tn_e_of(l + 1), tn_n_of(l + 1), G + 3 + GCN_LAYER_NPARAMS * (l + 1));
            if (l + 2 < p.L) layer_wgrad_jobs(unused, jobs.job[nj++], d, W.lw, tn_e_of(l + 2), tn_n_of(l + 2), G + 3 + GCN_LAYER_NPARAMS * (l + 2));
            if (nx.on) bn_finish_job(jobs.job[nj++], W.lw.bpart_e, npart_e, H, (double)d.Re, nx.G[11], nx.G[10], W.lw.bstats_e);
            if (head_tn_pending) { jobs.job[nj++] = head_tn_job; head_tn_pending = false; }    // top layer only: nj <= 2 there
            jobs.n = nj;
            MidMap map = {};
            int nb = node_blocks((d.Rn + 15) / 16, kMaxPartBlocks);
            if (const char* e = tune_env("GCN_TUNE_MIDPROBE")) {     // timing probe: segments of the launch dropped (results are then wrong)
                const int mk = atoi(e);
                if (mk & 1) nb = 0;
                if (mk & 2) pr.nslices[0] = 0;
                if (mk & 4) pr.nslices[1] = pr.nslices[2] = pr.nslices[3] = 0;
                if (mk & 8) { nj = 0; jobs.n = 0; }
            }
            map.seg[1] = nb;
            for (int k = 0; k < 4; ++k) { nb += pr.nslices[k]; map.seg[2 + k] = nb; }
            for (int k = 0; k < 4; ++k) { if (k < nj) nb += red_blocks(jobs.job[k]); map.seg[6 + k] = nb; }
            DISPATCH_MT(d.MT, if constexpr (MT != 0) GCN_LAUNCH_TIMED("k_bwd_mid", (k_bwd_mid<MT>), nb, GCN_BLOCK, na, H, pr, jobs, map));
        }
        // what is left after layer 0: its node products, and the slabs of the last launch
        {
            TnProblems pr = {};
            node_problems(pr, 0, 0);
            ReduceJobs jobs = {};
            ReduceJob unused = {};
            int nj = 0;
            layer_wgrad_jobs(jobs.job[nj++], unused, d, W.lw, tn_e_of(0), tn_n_of(0), G + 3);
            if (p.L > 1) layer_wgrad_jobs(unused, jobs.job[nj++], d, W.lw, tn_e_of(1), tn_n_of(1), G + 3 + GCN_LAYER_NPARAMS);
            jobs.n = nj;
            TailMap map = {};
            int nb = 0;
            for (int k = 0; k < 4; ++k) { map.seg[k] = nb; nb += pr.nslices[k]; }
            for (int k = 0; k < 4; ++k) { map.seg[4 + k] = nb; if (k < nj) nb += red_blocks(jobs.job[k]); }
            map.seg[8] = nb; nb += W.emb_blk;
            map.seg[9] = nb; nb += d.B * d.V;
            map.seg[10] = nb;
            const EmbedEdgesArgs ee = {d.Re, d.C, cap, W.de, W.embpart};
            const EmbedRowsumArgs er = {d.V, d.C, H, m->n_emb, x_nodes, demand, W.dx, W.emb_g, W.emb_cnt};
            DISPATCH_MT(d.MT, if constexpr (MT != 0) GCN_LAUNCH_TIMED("k_bwd_tail", (k_bwd_tail<MT>), nb, GCN_BLOCK, H, pr, jobs, ee, er, map));
        }
    }
    // embeddings (H <= 64: partial sums and row sums were part of the tail launch)
    if (d.MT > 4) {
        GCN_PROF_PRE("k_embed_bwd_edges");
        k_colsum_w<false><<<W.emb_blk, 256, 0, st>>>(d.Re, H, d.C, W.de, nullptr, nullptr, nullptr, nullptr, nullptr, cap, W.embpart,
                                                     nullptr, 0, 0.f, 0);
        GCN_CHECK_LAUNCH("k_embed_bwd_edges");
    }
    {
        // one reduce launch: products of layer 0 (slab set 0) and the edge-embedding gradient
        ReduceJobs jobs = {};
        ReduceJob unused = {};
        int nj = 0;
        if (d.MT > 4) { layer_wgrad_jobs(jobs.job[0], jobs.job[1], d, W.lw, W.lw.tn_e, W.lw.tn_n, G + 3); nj = 2; }
        else layer_wgrad_jobs(unused, jobs.job[nj++], d, W.lw, W.lw.tn_e, W.lw.tn_n, G + 3);       // node products of layer 0
        ReduceJob& je = jobs.job[nj++];
        je.partial = W.embpart; je.nP = W.emb_blk; je.ncols = H;
        je.rd.n = 1; je.rd.dst[0] = G[2]; je.rd.off[0] = 0; je.rd.off[1] = H;
        jobs.n = nj;
        // many short chains of dependent loads in the node-embedding finish: wide workgroups (hh * parts threads)
        const int eb_threads = std::max(GCN_BLOCK, std::min(1024, (1024 / std::max(H / 2, 1)) * (H / 2)));
        if (d.MT > 4) {
            GCN_PROF_PRE("k_reduce_slabs(layer 0, edge embedding)");
            k_reduce_slabs<<<dim3(red_grid(jobs), nj, 1), GCN_BLOCK, 0, st>>>(jobs);
            GCN_CHECK_LAUNCH("k_reduce_slabs(layer 0, edge embedding)");
            GCN_PROF_PRE("k_embed_bwd_rowsum");
            const EmbedRowsumArgs er = {d.V, d.C, H, m->n_emb, x_nodes, demand, W.dx, W.emb_g, W.emb_cnt};
            k_embed_bwd_rowsum<<<d.B * d.V, 64, 0, st>>>(er);
            GCN_CHECK_LAUNCH("k_embed_bwd_rowsum");
            GCN_PROF_PRE("k_embed_bwd_nodes");
            k_embed_bwd_nodes<<<m->n_emb + 1, eb_threads, eb_threads * sizeof(double), st>>>(d.B * d.V, d.C, H, m->n_emb, W.emb_g,
                                                                                        W.emb_cnt, G[0], G[1]);
            GCN_CHECK_LAUNCH("k_embed_bwd_nodes");
        } else {
            // H <= 64: the last reductions and the node-embedding finish need nothing from each other: one launch
            FinalMap map = {};
            int nb = 0;
            for (int k = 0; k < 4; ++k) { map.rseg[k] = nb; if (k < nj) nb += red_blocks(jobs.job[k]); }
            map.rseg[4] = nb;
            GCN_PROF_PRE("k_bwd_final");
            k_bwd_final<<<m->n_emb + 1 + nb, eb_threads, std::max(eb_threads, 1024) * sizeof(double), st>>>(
                d.B * d.V, d.C, H, m->n_emb, W.emb_g, W.emb_cnt, G[0], G[1], jobs, map);
            GCN_CHECK_LAUNCH("k_bwd_final");
        }
    }
    // Whatever was not marked on the way is final now: layer 0 (and the embeddings) on every path; up to 64 channels
    // the gradients of all layers are finished by launches shared between layers, so every part is marked here.
    if (done_events != nullptr) {
        if (d.MT > 4) { if (int rc = mark_done(0)) return rc; }
        else for (int k = 0; k <= p.L; ++k) { if (int rc = mark_done(k)) return rc; }
    }
    return 0;
}

// ---- stand-alone loss / edge error -----------------------------------------------------------------
size_t gcn_loss_ws_bytes(int B, int V, int C) {
    if (B <= 0 || V <= 0 || C <= 0) return 0;
    return align_up((size_t)kMaxPartBlocks * 2 * sizeof(float)) + align_up(2 * (size_t)B * C * sizeof(int));
}

int gcn_loss_edges_fwd(int B, int V, int C, const float* logits, const long long* y_edges, const float* edge_cw,
                       float* loss_out, unsigned char* err_flags, void* ws, size_t ws_bytes, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0) return fail(-1, "gcn_loss_edges_fwd: non-positive dimension");
    if (!logits || !y_edges || !loss_out || !err_flags || !ws) return fail(-1, "gcn_loss_edges_fwd: null pointer argument");
    if (ws_bytes < gcn_loss_ws_bytes(B, V, C)) return fail(-2, "gcn_loss_edges_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Bump b(ws, ws_bytes);
    float* part = b.f((size_t)kMaxPartBlocks * 2);
    int* counts = (int*)b.raw(2 * (size_t)B * C * sizeof(int));
    const size_t rows = (size_t)B * V * V * C;
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>((rows + GCN_BLOCK - 1) / GCN_BLOCK, kMaxPartBlocks));
    if (hipMemsetAsync(counts, 0, 2 * (size_t)B * C * sizeof(int), st) != hipSuccess) return fail(-3, "memset failed");
    GCN_PROF_PRE("k_loss_edges_fwd");
    k_loss_edges_fwd<<<blocks, GCN_BLOCK, 0, st>>>(rows, V * V * C, C, logits, y_edges, edge_cw, edge_cw ? part : nullptr, counts, B * C);
    GCN_CHECK_LAUNCH("k_loss_edges_fwd");
    if (!edge_cw) {
        // edge error only: give the finalizer zero-weight partials (loss comes out as NaN and is ignored)
        if (hipMemsetAsync(part, 0, (size_t)blocks * 2 * sizeof(float), st) != hipSuccess) return fail(-3, "memset failed");
    }
    GCN_PROF_PRE("k_loss_finalize");
    k_loss_finalize<<<1, GCN_BLOCK, 0, st>>>(part, blocks, counts, B * C, 1, loss_out, err_flags);
    GCN_CHECK_LAUNCH("k_loss_finalize");
    return 0;
}

int gcn_loss_edges_bwd(int B, int V, int C, const float* logits, const long long* y_edges, const float* edge_cw,
                       const float* loss_out, const float* dloss, float* dlogits, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0) return fail(-1, "gcn_loss_edges_bwd: non-positive dimension");
    if (!logits || !y_edges || !edge_cw || !loss_out || !dlogits) return fail(-1, "gcn_loss_edges_bwd: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)B * V * V * C;
    GCN_PROF_PRE("k_loss_edges_bwd");
    k_loss_edges_bwd<<<elem_blocks(rows), 256, 0, st>>>(rows, logits, y_edges, edge_cw, loss_out, dloss, dlogits);
    GCN_CHECK_LAUNCH("k_loss_edges_bwd");
    return 0;
}

// ---- beam search ---------------------------------------------------------------------------------
// workspace: working capacities [B][V][V] (strict mode when they do not fit LDS), then the speculative
// per-commodity results: lengths [B][C] and paths [B][C][max(V, 10)], then (large K x V only) the adjacency lists:
// row pointers [B][V + 1], columns and rows [B][nnzc] (beam_adj_capacity), and the scores gathered per list entry [B][C][nnzc]
static size_t beam_ws_base_bytes(int B, int V, int C) {
    const int PL = V > 10 ? V : 10;
    return align_up((size_t)B * V * V * sizeof(float)) + align_up((size_t)B * C * sizeof(int)) +
           align_up((size_t)B * C * PL * sizeof(int));
}
// ... then the state of the ordered walk between rounds: [B][4] ints and the working capacities per list entry [B][nnzc]
static size_t beam_ws_lists_bytes(int B, int V, int C, size_t nnzc) {
    return align_up((size_t)B * (V + 1) * sizeof(int)) + 2 * align_up((size_t)B * nnzc * sizeof(unsigned short)) +
           align_up((size_t)B * C * nnzc * sizeof(float));
}
size_t gcn_beam_ws_bytes(int B, int V, int C, int K) {
    if (B <= 0 || V <= 0 || C <= 0) return 0;
    const size_t nnzc = K > 0 ? (size_t)beam_adj_capacity(V, K) : 0;
    return beam_ws_base_bytes(B, V, C) + (nnzc ? beam_ws_lists_bytes(B, V, C, nnzc) + align_up((size_t)B * 4 * sizeof(int)) +
                                                 align_up((size_t)B * nnzc * sizeof(float)) : 0);
}
// Rounds of the ordered walk (k_beam_search): worth their launches when a search is long (large graphs: the working
// capacities do not fit LDS) and there are many commodities to speculate on.  One round per four commodities, at most 32.
static int beam_rounds(int V, int C, bool strict, int cap_in_lds, int nnzc, bool have_spec) {
    if (!strict || cap_in_lds || !nnzc || !have_spec || C < 16) return 0;
    return std::min(32, C / 4);
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
        if (!ws || ws_bytes < align_up((size_t)B * V * V * sizeof(float))) return fail(-2, "gcn_beam_search: workspace too small");
    }
    // adjacency lists (large K x V): only with the full workspace, and only if they fit LDS beside the beam state
    hipStream_t st = (hipStream_t)stream;
    int nnzc = beam_adj_capacity(V, K);
    if (!ws || ws_bytes < gcn_beam_ws_bytes(B, V, C, K) || beam_lds_bytes(V, K, cap_in_lds, nnzc) > 96 * 1024) nnzc = 0;
    int* adj_ptr = nullptr;
    unsigned short *adj_col = nullptr, *adj_row = nullptr;
    float* y_e = nullptr;
    if (nnzc) {
        char* base = (char*)ws + beam_ws_base_bytes(B, V, C);
        adj_ptr = (int*)base; base += align_up((size_t)B * (V + 1) * sizeof(int));
        adj_col = (unsigned short*)base; base += align_up((size_t)B * nnzc * sizeof(unsigned short));
        adj_row = (unsigned short*)base; base += align_up((size_t)B * nnzc * sizeof(unsigned short));
        y_e = (float*)base;
        GCN_PROF_PRE("k_beam_adjacency");
        k_beam_adjacency<<<B, 256, 0, st>>>(V, nnzc, cap, adj_ptr, adj_col, adj_row);
        GCN_CHECK_LAUNCH("k_beam_adjacency");
        GCN_PROF_PRE("k_beam_gather_y");
        k_beam_gather_y<<<dim3(B, 64, 1), 256, 0, st>>>(V, C, nnzc, y, adj_ptr, adj_col, adj_row, y_e);
        GCN_CHECK_LAUNCH("k_beam_gather_y");
    }
    const size_t lds = beam_lds_bytes(V, K, cap_in_lds, nnzc);
    if (lds > 160 * 1024) return fail(-1, "gcn_beam_search: beam state of %zu bytes exceeds the 160 KiB LDS (V=%d, K=%d)", lds, V, K);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_beam_search), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(-3, "gcn_beam_search: cannot raise the dynamic LDS limit to %zu bytes", lds);
    }
    const int jobs = algorithm == BEAM_ALGO_UNCONSTRAINED ? B * C : B;
    // speculative pass over all (graph, commodity) pairs, when the workspace has room for its results
    int *spec_len = nullptr, *spec_paths = nullptr;
    if (algorithm != BEAM_ALGO_UNCONSTRAINED && ws && ws_bytes >= beam_ws_base_bytes(B, V, C)) {
        char* base = (char*)ws + align_up((size_t)B * V * V * sizeof(float));
        spec_len = (int*)base;
        spec_paths = (int*)(base + align_up((size_t)B * C * sizeof(int)));
        int spec_cap_lds = (size_t)V * V * sizeof(float) <= 32768 ? 1 : 0;
        if (beam_lds_bytes(V, K, spec_cap_lds, nnzc) > 160 * 1024) spec_cap_lds = 0;
        const size_t slds = beam_lds_bytes(V, K, spec_cap_lds, nnzc);
        if (slds > 48 * 1024) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_beam_spec), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds) != hipSuccess)
                return fail(-3, "gcn_beam_search: cannot raise the dynamic LDS limit to %zu bytes", slds);
        }
        GCN_PROF_PRE("k_beam_spec");
        k_beam_spec<<<B * C, 64, slds, st>>>(B, V, C, K, algorithm, y, cap, comm, run_mask, spec_cap_lds, spec_paths, spec_len,
                                             adj_ptr, adj_col, y_e, nnzc, nullptr, nullptr, nullptr, perms, n_perms, perm_start);
        GCN_CHECK_LAUNCH("k_beam_spec");
        const int rounds = beam_rounds(V, C, strict, cap_in_lds, nnzc, true);
        if (rounds > 0) {
            // the ordered walk in rounds: walk until a commodity's predicate differs from the snapshot, search all the
            // remaining commodities against the capacities reached, walk on (see k_beam_search); the last launch
            // finishes whatever is left one commodity after the other
            int* state = (int*)((char*)y_e + align_up((size_t)B * C * nnzc * sizeof(float)));
            float* work_e = (float*)((char*)state + align_up((size_t)B * 4 * sizeof(int)));
            for (int r = 0; r <= rounds; ++r) {
                const int mode = r == 0 ? 1 : (r < rounds ? 2 : 3);
                if (r > 0) {
                    GCN_PROF_PRE("k_beam_spec");
                    k_beam_spec<<<B * C, 64, slds, st>>>(B, V, C, K, algorithm, y, cap, comm, run_mask, spec_cap_lds, spec_paths,
                                                         spec_len, adj_ptr, adj_col, y_e, nnzc, state, work_e, (const float*)ws,
                                                         perms, n_perms, perm_start);
                    GCN_CHECK_LAUNCH("k_beam_spec");
                }
                GCN_PROF_PRE("k_beam_search");
                k_beam_search<<<jobs, 64, lds, st>>>(B, V, C, K, algorithm, 1, max_iter, y, cap, comm, perms, n_perms, perm_start,
                                                     run_mask, (float*)ws, cap_in_lds, paths, path_len, feasible, attempts,
                                                     spec_paths, spec_len, adj_ptr, adj_col, adj_row, y_e, nnzc, mode, state, work_e);
                GCN_CHECK_LAUNCH("k_beam_search");
            }
            return 0;
        }
    }
    GCN_PROF_PRE("k_beam_search");
    k_beam_search<<<jobs, 64, lds, st>>>(B, V, C, K, algorithm, strict ? 1 : 0, max_iter, y, cap, comm, perms,
                                                          n_perms, perm_start, run_mask, (float*)ws, cap_in_lds, paths, path_len,
                                                          feasible, attempts, spec_paths, spec_len, adj_ptr, adj_col, adj_row, y_e, nnzc,
                                                          0, nullptr, nullptr);
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
    }
    GCN_CHECK_LAUNCH("k_beam_scores");
    return 0;
}


// usage_ws: B*V*V int32 of scratch.  max_lf: float [B].  complete: uint8 [B*C].
int gcn_load_factor(int B, int V, int C, int PL, const int* paths, const int* path_len, const float* cap,
                    const long long* comm, int* usage_ws, float* max_lf, unsigned char* complete, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0 || PL <= 0) return fail(-1, "gcn_load_factor: non-positive dimension");
    if (!paths || !path_len || !cap || !comm || !usage_ws || !max_lf || !complete) return fail(-1, "gcn_load_factor: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(usage_ws, 0, (size_t)B * V * V * sizeof(int), st) != hipSuccess) return fail(-3, "memset failed");
    GCN_PROF_PRE("k_path_usage");
    k_path_usage<<<elem_blocks((size_t)B * C * PL), 256, 0, st>>>(B, V, C, PL, paths, path_len, comm, usage_ws, complete);
    GCN_CHECK_LAUNCH("k_path_usage");
    GCN_PROF_PRE("k_max_load_factor");
    k_max_load_factor<<<B, 256, 0, st>>>(V, usage_ws, cap, max_lf);
    GCN_CHECK_LAUNCH("k_max_load_factor");
    return 0;
}

// ---- load factor without any allocation: the usage scratch is a fixed array of the code object ----------
// In host code the array's name is only a handle for hipGetSymbolAddress; every use of its memory goes through the
// per-device address looked up once per device (each GPU holds its own copy at its own address).
static __device__ __align__(256) int scratch[GCN_STATIC_USAGE_INTS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
int* g_scratch_ptr[kMaxDevices] = {};

int static_scratch(int** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return fail(-3, "gcn_load_factor_static: no current device");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr)
            return fail(-3, "gcn_load_factor_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<int*>(p);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

int gcn_load_factor_static(int B, int V, int C, int PL, const int* paths, const int* path_len, const float* cap,
                           const long long* comm, float* max_lf, unsigned char* complete, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0 || PL <= 0) return fail(-1, "gcn_load_factor_static: non-positive dimension");
    if (!paths || !path_len || !cap || !comm || !max_lf || !complete) return fail(-1, "gcn_load_factor_static: null pointer argument");
    const size_t n = (size_t)B * V * V;
    if (n > (size_t)GCN_STATIC_USAGE_INTS)
        return fail(-2, "gcn_load_factor_static: B*V*V = %zu exceeds the static scratch of %u int32 (use gcn_load_factor with a caller buffer)",
                    n, (unsigned)GCN_STATIC_USAGE_INTS);
    int* usage = nullptr;
    if (int rc = static_scratch(&usage)) return rc;
    // the array keeps the previous call's counts: gcn_load_factor clears the part this call uses
    return gcn_load_factor(B, V, C, PL, paths, path_len, cap, comm, usage, max_lf, complete, stream);
}

namespace {
int sampler_check(const char* who, int B, int V, int C, float temperature) {
    if (B <= 0 || V <= 0 || C <= 0) return fail(-1, "%s: non-positive dimension", who);
    if (V > SMP_MAX_V) return fail(-2, "%s: V = %d exceeds the sampler's limit of %d nodes", who, V, SMP_MAX_V);
    if (!(temperature > 0.f)) return fail(-1, "%s: temperature must be positive", who);
    if ((size_t)B * V * V * C >= ((size_t)1 << 31)) return fail(-2, "%s: logits tensor too large for 32-bit row indices", who);
    return 0;
}
}  // namespace

int gcn_path_sample(int B, int V, int C, float temperature, float top_p, float eps, const float* y, const float* cap,
                    const long long* comm, const unsigned long long* rng, int given_paths, int* paths, int* path_len,
                    float* logp, float* step_entropy, int* n_entropy, void* stream) {
    if (int rc = sampler_check("gcn_path_sample", B, V, C, temperature)) return rc;
    if (!y || !cap || !comm || !paths || !path_len || !logp || !step_entropy || !n_entropy)
        return fail(-1, "gcn_path_sample: null pointer argument");
    if (!given_paths && !rng) return fail(-1, "gcn_path_sample: rng is required when paths are drawn");
    SampleArgs a = {};
    a.B = B; a.V = V; a.C = C; a.PL = std::max(V, 10);
    a.temperature = temperature; a.top_p = top_p; a.eps = eps;
    a.y = y; a.cap = cap; a.comm = comm; a.rng = rng;
    a.paths = paths; a.path_len = path_len; a.logp = logp; a.ent = step_entropy; a.n_ent = n_entropy;
    hipStream_t st = (hipStream_t)stream;
    GCN_PROF_PRE("k_path_sample");
    if (given_paths) k_path_sample<SMP_MODE_EVAL><<<B * C, 64, 0, st>>>(a);
    else k_path_sample<SMP_MODE_SAMPLE><<<B * C, 64, 0, st>>>(a);
    GCN_CHECK_LAUNCH("k_path_sample");
    return 0;
}

int gcn_path_sample_bwd(int B, int V, int C, float temperature, float eps, const float* y, const float* cap,
                        const long long* comm, const int* paths, const int* path_len, const float* dlogp, float* dy,
                        void* stream) {
    if (int rc = sampler_check("gcn_path_sample_bwd", B, V, C, temperature)) return rc;
    if (!y || !cap || !comm || !paths || !path_len || !dlogp || !dy) return fail(-1, "gcn_path_sample_bwd: null pointer argument");
    SampleArgs a = {};
    a.B = B; a.V = V; a.C = C; a.PL = std::max(V, 10);
    a.temperature = temperature; a.top_p = 1.f; a.eps = eps;
    a.y = y; a.cap = cap; a.comm = comm;
    a.paths = const_cast<int*>(paths); a.path_len = const_cast<int*>(path_len);
    a.dlogp = dlogp; a.dy = dy;
    hipStream_t st = (hipStream_t)stream;
    GCN_PROF_PRE("k_path_sample_bwd");
    k_path_sample<SMP_MODE_BWD><<<B * C, 64, 0, st>>>(a);
    GCN_CHECK_LAUNCH("k_path_sample_bwd");
    return 0;
}

int gcn_reachability(int B, int V, const float* cap, unsigned char* reach, void* stream) {
    if (B <= 0 || V <= 0 || !cap || !reach) return fail(-1, "gcn_reachability: bad argument");
    if (V > SMP_MAX_V) return fail(-2, "gcn_reachability: V = %d exceeds the limit of %d nodes", V, SMP_MAX_V);
    hipStream_t st = (hipStream_t)stream;
    GCN_PROF_PRE("k_reachability");
    k_reachability<<<B * V, 64, 0, st>>>(V, cap, reach);
    GCN_CHECK_LAUNCH("k_reachability");
    return 0;
}

int gcn_entropy_grid_fwd(int B, int V, int C, float temperature, const float* y, float* row_ws, float* out, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0 || !y || !row_ws || !out) return fail(-1, "gcn_entropy_grid_fwd: bad argument");
    if (!(temperature > 0.f)) return fail(-1, "gcn_entropy_grid_fwd: temperature must be positive");
    if ((size_t)B * V * V * C >= ((size_t)1 << 31)) return fail(-2, "gcn_entropy_grid_fwd: logits tensor too large");
    const int rows = B * V * C;
    hipStream_t st = (hipStream_t)stream;
    GCN_PROF_PRE("k_entropy_rows");
    k_entropy_rows<<<(rows + 255) / 256, 256, 0, st>>>(rows, V, C, temperature, y, row_ws);
    GCN_CHECK_LAUNCH("k_entropy_rows");
    GCN_PROF_PRE("k_entropy_finish");
    k_entropy_finish<<<1, 256, 0, st>>>(rows, row_ws, out);
    GCN_CHECK_LAUNCH("k_entropy_finish");
    return 0;
}

int gcn_entropy_grid_bwd(int B, int V, int C, float temperature, const float* y, const float* row_ws, const float* out,
                         const float* dout, float* dy, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0 || !y || !row_ws || !out || !dy) return fail(-1, "gcn_entropy_grid_bwd: bad argument");
    if (!(temperature > 0.f)) return fail(-1, "gcn_entropy_grid_bwd: temperature must be positive");
    if ((size_t)B * V * V * C >= ((size_t)1 << 31)) return fail(-2, "gcn_entropy_grid_bwd: logits tensor too large");
    const int rows = B * V * C;
    hipStream_t st = (hipStream_t)stream;
    GCN_PROF_PRE("k_entropy_bwd");
    k_entropy_bwd<<<(rows + 255) / 256, 256, 0, st>>>(rows, V, C, temperature, y, row_ws, out, dout, dy);
    GCN_CHECK_LAUNCH("k_entropy_bwd");
    return 0;
}

int gcn_unpack_batch(int B, int V, int C, const unsigned char* edges_u8, const int* comm_i32, const short* flow_i16,
                     long long* x_edges, long long* x_nodes, long long* y_edges, long long* comm, float* demand,
                     short* succ_ws, void* stream) {
    if (B <= 0 || V <= 0 || C <= 0) return fail(-1, "gcn_unpack_batch: non-positive dimension");
    if (V > 32767) return fail(-2, "gcn_unpack_batch: V = %d does not fit the 16-bit node order", V);
    if (!edges_u8 || !comm_i32 || !flow_i16 || !x_edges || !x_nodes || !y_edges || !comm || !demand || !succ_ws)
        return fail(-1, "gcn_unpack_batch: null pointer argument");
    // every output must be no larger than y_edges, whose index range the fill kernel walks: V >= 3 / C covers comm
    if ((size_t)V * V < 3) return fail(-2, "gcn_unpack_batch: needs V >= 2");
    if ((size_t)B * C * V >= ((size_t)1 << 31)) return fail(-2, "gcn_unpack_batch: batch too large");
    hipStream_t st = (hipStream_t)stream;
    const int n = B * C * V;
    GCN_PROF_PRE("k_unpack_succ");
    k_unpack_succ<<<(n + 255) / 256, 256, 0, st>>>(n, V, flow_i16, succ_ws);
    GCN_CHECK_LAUNCH("k_unpack_succ");
    UnpackArgs a = {B, V, C, edges_u8, comm_i32, succ_ws, x_edges, x_nodes, y_edges, comm, demand};
    const size_t ny = (size_t)B * V * V * C;
    GCN_PROF_PRE("k_unpack_fill");
    k_unpack_fill<<<(int)std::min<size_t>((ny + 255) / 256, 4096), 256, 0, st>>>(a);
    GCN_CHECK_LAUNCH("k_unpack_fill");
    return 0;
}

int gcn_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const long long* numel, float lr, float beta1, float beta2, float eps,
                  float weight_decay, long long step, void* stream) {
    if (n < 0 || step < 1) return fail(-1, "gcn_adam_step: n %d < 0 or step %lld < 1", n, step);
    if (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)) return fail(-1, "gcn_adam_step: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
    for (int i0 = 0; i0 < n; i0 += ADAM_CHUNK) {
        AdamChunk c = {};
        c.n = std::min(ADAM_CHUNK, n - i0);
        long long maxn = 0;
        for (int k = 0; k < c.n; ++k) {
            if (numel[i0 + k] < 0 || (numel[i0 + k] > 0 && (!params[i0 + k] || !grads[i0 + k] || !exp_avg[i0 + k] || !exp_avg_sq[i0 + k])))
                return fail(-1, "gcn_adam_step: tensor %d has a null pointer or a negative size", i0 + k);
            c.p[k] = params[i0 + k]; c.g[k] = grads[i0 + k]; c.m[k] = exp_avg[i0 + k]; c.v[k] = exp_avg_sq[i0 + k];
            c.numel[k] = numel[i0 + k];
            maxn = std::max(maxn, numel[i0 + k]);
        }
        if (maxn == 0) continue;
        const int bx = (int)std::max<long long>(1, std::min<long long>((maxn + 1023) / 1024, 256));
        GCN_PROF_PRE("k_adam");
        k_adam<<<dim3(bx, c.n, 1), 256, 0, st>>>(c, step_size, beta1, beta2, eps, weight_decay, inv_sqrt_bc2);
        GCN_CHECK_LAUNCH("k_adam");
    }
    return 0;
}

}  // extern "C"
