// Stochastic path sampling for the REINFORCE training path on the GPU (gfx950): one wavefront per
// (graph, commodity), lane l owns the next-node candidates l, l + 64, ...
//
// Restates src/gcn/algorithms/path_sampler.py of the reference:
//   * edges with cap <= 0 and self loops are invalid: their logits become -1e20 before the softmax over the
//     next node j of logits / temperature, and their probability is forced to exactly 0 (:86, :106-114);
//   * a node may be entered only if it is not on the path yet and the destination is reachable from it over
//     valid edges (:305-318; reachability = breadth-first search over cap > 0, :220-260);
//   * if the masked row sums to <= 1e-20 the step falls back to a uniform choice among the nodes that are
//     valid, unvisited and reachable, adding log(1 / n) to the log-probability, or ends the path with
//     log(1e-8) if there is none (:326-372).  (The reference's "sample from the model's probabilities"
//     branch, :349-356, needs those same probabilities to sum to more than 1e-10 and can therefore never be
//     taken once their sum is <= 1e-20.);
//   * otherwise the row is normalised, mixed with the uniform distribution over the same node set
//     ((1 - eps) p + eps u, renormalised with + 1e-8, :374-391), its entropy -sum p log(p + 1e-8) is recorded
//     (:394-395), the next node is drawn from the top-p nucleus of the row (:407-426) and
//     log(p[next] + 1e-8) is added to the log-probability (:400);
//   * a path ends at the destination, when stuck, or after 2 V steps (:292-296).
// The draw itself uses a counter-based uniform per (seed, call, graph, commodity, step) instead of torch's
// generator: u picks, in descending-probability order (ties: lower node first), the first nucleus entry
// whose cumulative mass exceeds u * (nucleus mass).  The backward pass differentiates the log-probability
// with respect to the logits analytically (what autograd derives for :111-400).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define SMP_CPL 4                    // candidates per lane
#define SMP_MAX_V (64 * SMP_CPL)     // largest supported number of nodes

#define SMP_MODE_SAMPLE 0            // draw paths
#define SMP_MODE_EVAL 1              // log-probability and entropies of given paths
#define SMP_MODE_BWD 2               // d log-probability / d logits of given paths

struct SampleArgs {
    int B, V, C, PL;
    float temperature, top_p, eps;
    const float* y;                  // [B,V,V,C] logits
    const float* cap;                // [B,V,V]
    const long long* comm;           // [B,C,3]
    const unsigned long long* rng;   // {seed, call counter}
    int* paths;                      // [B,C,PL]   (read in EVAL / BWD)
    int* path_len;                   // [B,C]
    float* logp;                     // [B,C]
    float* ent;                      // [B,C,PL] per-step entropies
    int* n_ent;                      // [B,C]
    const float* dlogp;              // [B]   upstream gradient of the per-graph log-probability
    float* dy;                       // [B,V,V,C], accumulated into
};

__device__ __forceinline__ float smp_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float smp_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int smp_maxi(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ uint32_t smp_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// uniform in [0, 1) with 24 random bits
__device__ __forceinline__ float smp_uniform(unsigned long long seed, unsigned long long call, uint32_t bc, uint32_t step) {
    uint32_t h = smp_mix((uint32_t)seed ^ 0x9E3779B9U);
    h = smp_mix(h ^ (uint32_t)(seed >> 32));
    h = smp_mix(h ^ (uint32_t)call);
    h = smp_mix(h ^ (uint32_t)(call >> 32) ^ 0x85EBCA6BU);
    h = smp_mix(h + bc * 0x9E3779B9U);
    h = smp_mix(h ^ (step * 0xC2B2AE35U + 0x165667B1U));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// adjacency bit rows (cap > 0, no self loops) of graph b into LDS: adj[r * SMP_CPL + w] bit l = edge r -> l + 64 w
__device__ __forceinline__ void smp_build_adj(const float* __restrict__ capb, int V, int lane, unsigned long long* adj) {
    for (int r = 0; r < V; ++r) {
#pragma unroll
        for (int i = 0; i < SMP_CPL; ++i) {
            const int j = lane + 64 * i;
            const bool bit = j < V && j != r && capb[(size_t)r * V + j] > 0.f;
            const unsigned long long w = __ballot(bit);
            if (lane == 0) adj[r * SMP_CPL + i] = w;
        }
    }
    __syncthreads();
}

// nodes from which dst is reachable (dst included), as SMP_CPL wavefront-uniform words
__device__ __forceinline__ void smp_reach(const unsigned long long* adj, int V, int lane, int dst, unsigned long long* R) {
#pragma unroll
    for (int i = 0; i < SMP_CPL; ++i) R[i] = (dst >> 6) == i ? 1ull << (dst & 63) : 0ull;
    for (int it = 0; it < V; ++it) {
        unsigned long long N[SMP_CPL];
        bool any = false;
#pragma unroll
        for (int i = 0; i < SMP_CPL; ++i) {
            const int j = lane + 64 * i;
            bool hit = false;
            if (j < V && !((R[i] >> lane) & 1)) {
                unsigned long long acc = 0;
#pragma unroll
                for (int w = 0; w < SMP_CPL; ++w) acc |= adj[j * SMP_CPL + w] & R[w];
                hit = acc != 0;
            }
            N[i] = __ballot(hit);
        }
#pragma unroll
        for (int i = 0; i < SMP_CPL; ++i) { any = any || N[i] != 0; R[i] |= N[i]; }
        if (!any) break;
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_path_sample(SampleArgs a) {
    __shared__ unsigned long long adj[SMP_MAX_V * SMP_CPL];
    __shared__ float arr[SMP_MAX_V];
    const int lane = threadIdx.x, bc = blockIdx.x, b = bc / a.C, c = bc % a.C, V = a.V;
    int* path = a.paths + (size_t)bc * a.PL;
    const int src = (int)a.comm[(size_t)bc * 3 + 0], dst = (int)a.comm[(size_t)bc * 3 + 1];
    if (src < 0 || src >= V || dst < 0 || dst >= V) {       // malformed commodity: empty path
        if (MODE != SMP_MODE_BWD && lane == 0) {
            a.logp[bc] = 0.f; a.n_ent[bc] = 0;
            if (MODE == SMP_MODE_SAMPLE) { a.path_len[bc] = 0; for (int k = 0; k < a.PL; ++k) path[k] = -1; }
        }
        return;
    }
    smp_build_adj(a.cap + (size_t)b * V * V, V, lane, adj);
    unsigned long long R[SMP_CPL], Vis[SMP_CPL];
    smp_reach(adj, V, lane, dst, R);
#pragma unroll
    for (int i = 0; i < SMP_CPL; ++i) Vis[i] = (src >> 6) == i ? 1ull << (src & 63) : 0ull;

    int plen = MODE == SMP_MODE_SAMPLE ? 0 : min(a.path_len[bc], a.PL);
    unsigned long long seed = 0, call = 0;
    if (MODE == SMP_MODE_SAMPLE) { seed = a.rng[0]; call = a.rng[1]; }
    const float c1 = (float)(1.0 - (double)a.eps);
    const float gup = MODE == SMP_MODE_BWD ? a.dlogp[b] : 0.f;
    int u = src, len = 1, nent = 0;
    float logp = 0.f;
    if (MODE == SMP_MODE_SAMPLE && lane == 0) path[0] = src;
    if (MODE != SMP_MODE_SAMPLE && (plen < 1 || path[0] != src)) len = plen = 0, u = dst;   // nothing to evaluate

    for (int step = 0; step < 2 * V; ++step) {
        if (u == dst) break;
        if (MODE == SMP_MODE_BWD && len >= plen) break;
        const float* yrow = a.y + ((size_t)(b * V + u) * V) * a.C + c;
        bool av[SMP_CPL], sv[SMP_CPL];
        float z[SMP_CPL], o[SMP_CPL], os[SMP_CPL];
        unsigned long long S[SMP_CPL];
        float zmax = -INFINITY;
        int nvalid = 0, ns = 0;
#pragma unroll
        for (int i = 0; i < SMP_CPL; ++i) {
            const int j = lane + 64 * i;
            const unsigned long long aw = adj[u * SMP_CPL + i];
            av[i] = j < V && ((aw >> lane) & 1);
            sv[i] = av[i] && !((Vis[i] >> lane) & 1) && ((R[i] >> lane) & 1);
            z[i] = av[i] ? yrow[(size_t)j * a.C] / a.temperature : -INFINITY;
            zmax = fmaxf(zmax, z[i]);
            nvalid += __popcll(aw);
            S[i] = __ballot(sv[i]);
            ns += __popcll(S[i]);
        }
        zmax = smp_max(zmax);
        float se = 0.f;
#pragma unroll
        for (int i = 0; i < SMP_CPL; ++i) { o[i] = av[i] ? expf(z[i] - zmax) : 0.f; se += o[i]; }
        se = smp_sum(se);
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < SMP_CPL; ++i) { o[i] = (nvalid > 0 && sv[i]) ? o[i] / se : 0.f; tot += o[i]; }
        tot = smp_sum(tot);
        const float u01 = MODE == SMP_MODE_SAMPLE ? smp_uniform(seed, call, (uint32_t)bc, (uint32_t)step) : 0.f;
        int next = -1;
        if (!(tot > 1e-20f)) {
            // fallback: uniform over the valid, unvisited, reachable nodes; none = stuck
            if (ns == 0) { logp += (float)log(1e-8); break; }
            if (MODE == SMP_MODE_SAMPLE) {
                const int k = min((int)(u01 * (float)ns), ns - 1);
                int before = 0, mine = -1;
#pragma unroll
                for (int i = 0; i < SMP_CPL; ++i) {
                    const int idx = before + __popcll(S[i] & ((1ull << lane) - 1ull));
                    if (sv[i] && idx == k) mine = lane + 64 * i;
                    before += __popcll(S[i]);
                }
                next = smp_maxi(mine);
            } else {
                if (len >= plen) break;
                next = path[len];
            }
            if (next < 0 || next >= V || len >= a.PL) break;
            logp += (float)log(1.0 / (double)ns);
        } else {
            float den = 1.f;
            bool mixed = false;
#pragma unroll
            for (int i = 0; i < SMP_CPL; ++i) { o[i] = o[i] / tot; os[i] = o[i]; }
            if (a.eps > 0.f && ns > 0) {
                const float un = 1.0f / ((float)ns + 1e-8f);
                float so = 0.f;
#pragma unroll
                for (int i = 0; i < SMP_CPL; ++i) { o[i] = c1 * o[i] + a.eps * (sv[i] ? un : 0.f); so += o[i]; }
                so = smp_sum(so);
                den = so + 1e-8f;
#pragma unroll
                for (int i = 0; i < SMP_CPL; ++i) o[i] = o[i] / den;
                mixed = true;
            }
            float H = 0.f;
#pragma unroll
            for (int i = 0; i < SMP_CPL; ++i) H += o[i] * logf(o[i] + 1e-8f);
            H = -smp_sum(H);
            if (MODE == SMP_MODE_SAMPLE) {
                float pp[SMP_CPL], sp = 0.f;
#pragma unroll
                for (int i = 0; i < SMP_CPL; ++i) { pp[i] = o[i] > 1e-8f ? o[i] : 0.f; sp += pp[i]; }
                sp = smp_sum(sp);
                if (!(sp > 0.f)) {
                    next = min((int)(u01 * (float)V), V - 1);       // reference: uniform over all nodes (:410-412)
                } else {
#pragma unroll
                    for (int i = 0; i < SMP_CPL; ++i) {
                        pp[i] = pp[i] / sp;
                        if (lane + 64 * i < V) arr[lane + 64 * i] = pp[i];
                    }
                    __syncthreads();
                    float cb[SMP_CPL];
                    int rk[SMP_CPL];
#pragma unroll
                    for (int i = 0; i < SMP_CPL; ++i) { cb[i] = 0.f; rk[i] = 0; }
                    for (int k = 0; k < V; ++k) {
                        const float q = arr[k];
#pragma unroll
                        for (int i = 0; i < SMP_CPL; ++i) {
                            const bool before = q > pp[i] || (q == pp[i] && k < lane + 64 * i);
                            cb[i] += before ? q : 0.f;
                            rk[i] += before ? 1 : 0;
                        }
                    }
                    __syncthreads();
                    bool kept[SMP_CPL];
                    float mass = 0.f;
#pragma unroll
                    for (int i = 0; i < SMP_CPL; ++i) { kept[i] = pp[i] > 0.f && cb[i] <= a.top_p; mass += kept[i] ? pp[i] : 0.f; }
                    mass = smp_sum(mass);
                    const float t = u01 * mass;
                    int key = -1;
#pragma unroll
                    for (int i = 0; i < SMP_CPL; ++i) if (kept[i] && cb[i] <= t) key = max(key, rk[i]);
                    key = smp_maxi(key);
                    int mine = -1;
#pragma unroll
                    for (int i = 0; i < SMP_CPL; ++i) if (kept[i] && rk[i] == key) mine = lane + 64 * i;
                    next = smp_maxi(mine);
                }
            } else {
                if (len >= plen) break;
                next = path[len];
            }
            if (next < 0 || next >= V || len >= a.PL) break;
            float on = 0.f, osn = 0.f;
#pragma unroll
            for (int i = 0; i < SMP_CPL; ++i) if (lane + 64 * i == next) { on = o[i]; osn = os[i]; }
            on = smp_sum(on);
            osn = smp_sum(osn);
            logp += logf(on + 1e-8f);
            if (MODE != SMP_MODE_BWD) {
                if (lane == 0) a.ent[(size_t)bc * a.PL + nent] = H;
                ++nent;
            } else {
                // d log(q_next + 1e-8) / d logit_i = coef * (delta(i, next) - softmax_i) over the support
                const float coef = gup * (mixed ? c1 / den / (on + 1e-8f) * osn : osn / (osn + 1e-8f)) / a.temperature;
                float* drow = a.dy + ((size_t)(b * V + u) * V) * a.C + c;
#pragma unroll
                for (int i = 0; i < SMP_CPL; ++i) {
                    const int j = lane + 64 * i;
                    if (sv[i]) drow[(size_t)j * a.C] += coef * ((j == next ? 1.f : 0.f) - os[i]);
                }
            }
        }
        if (MODE == SMP_MODE_SAMPLE && lane == 0) path[len] = next;
        ++len;
#pragma unroll
        for (int i = 0; i < SMP_CPL; ++i) if ((next >> 6) == i) Vis[i] |= 1ull << (next & 63);
        u = next;
    }
    if (MODE == SMP_MODE_BWD) return;
    if (MODE == SMP_MODE_SAMPLE) {
        for (int k = len + lane; k < a.PL; k += 64) path[k] = -1;
        if (lane == 0) a.path_len[bc] = len;
    }
    if (lane == 0) { a.logp[bc] = logp; a.n_ent[bc] = nent; }
}

// reach[b][i][dst] = 1 iff dst can be reached from i over edges with cap > 0 (i == dst included): the
// reference's PathSampler.reachability (:220-260).  One wavefront per (graph, destination).
__global__ __launch_bounds__(64) void k_reachability(int V, const float* __restrict__ cap, unsigned char* __restrict__ reach) {
    __shared__ unsigned long long adj[SMP_MAX_V * SMP_CPL];
    const int lane = threadIdx.x, b = blockIdx.x / V, dst = blockIdx.x % V;
    smp_build_adj(cap + (size_t)b * V * V, V, lane, adj);
    unsigned long long R[SMP_CPL];
    smp_reach(adj, V, lane, dst, R);
#pragma unroll
    for (int i = 0; i < SMP_CPL; ++i) {
        const int j = lane + 64 * i;
        if (j < V) reach[((size_t)b * V + j) * V + dst] = (unsigned char)((R[i] >> lane) & 1);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Entropy of the next-node distribution of every (graph, node, commodity) row, the exploration bonus of the
// REINFORCE loss (src/gcn/training/reinforcement_strategy.py:460-476, :494):
//   p = softmax(y / T over j), H_row = -sum_j p log p, multi_row = (#{j: p > 1e-8} >= 2),
//   entropy = mean of H_row over the rows with multi_row (over all rows if there is none).
// k_entropy_rows: one thread per row (rows are contiguous in c, so a wavefront reads contiguous logits for each
// j).  row_ws[r] = H_row, row_ws[rows + r] = multi_row.  k_entropy_finish reduces in a fixed order:
// out = {entropy, number of rows averaged, 1 if only multi rows were averaged, 0}.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_entropy_rows(int rows, int V, int C, float temperature, const float* __restrict__ y,
                                                      float* __restrict__ row_ws) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const int bi = r / C, c = r % C;
    const float* yr = y + (size_t)bi * V * C + c;
    float zmax = -INFINITY;
    for (int j = 0; j < V; ++j) zmax = fmaxf(zmax, yr[(size_t)j * C] / temperature);
    float se = 0.f;
    for (int j = 0; j < V; ++j) se += expf(yr[(size_t)j * C] / temperature - zmax);
    const float lse = logf(se);
    float H = 0.f;
    int n = 0;
    for (int j = 0; j < V; ++j) {
        const float lp = yr[(size_t)j * C] / temperature - zmax - lse;
        const float p = expf(yr[(size_t)j * C] / temperature - zmax) / se;
        H -= p * lp;
        n += p > 1e-8f ? 1 : 0;
    }
    row_ws[r] = H;
    row_ws[rows + r] = n >= 2 ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void k_entropy_finish(int rows, const float* __restrict__ row_ws, float* __restrict__ out) {
    __shared__ double sm[256], sa[256];
    __shared__ int cm[256];
    double m = 0.0, a = 0.0;
    int n = 0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const float H = row_ws[r];
        a += H;
        if (row_ws[rows + r] != 0.f) { m += H; ++n; }
    }
    sm[threadIdx.x] = m; sa[threadIdx.x] = a; cm[threadIdx.x] = n;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sm[threadIdx.x] += sm[threadIdx.x + s]; sa[threadIdx.x] += sa[threadIdx.x + s]; cm[threadIdx.x] += cm[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const bool multi = cm[0] > 0;
        const int cnt = multi ? cm[0] : rows;
        out[0] = (float)((multi ? sm[0] : sa[0]) / (double)cnt);
        out[1] = (float)cnt; out[2] = multi ? 1.f : 0.f; out[3] = 0.f;
    }
}

// dy[b,i,j,c] += dout / count * (-p_j (log p_j + H_row)) / T for the rows that were averaged
__global__ __launch_bounds__(256) void k_entropy_bwd(int rows, int V, int C, float temperature, const float* __restrict__ y,
                                                     const float* __restrict__ row_ws, const float* __restrict__ out,
                                                     const float* __restrict__ dout, float* __restrict__ dy) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    if (out[2] != 0.f && row_ws[rows + r] == 0.f) return;
    const int bi = r / C, c = r % C;
    const float* yr = y + (size_t)bi * V * C + c;
    float* dr = dy + (size_t)bi * V * C + c;
    const float g = (dout ? dout[0] : 1.f) / out[1] / temperature, H = row_ws[r];
    float zmax = -INFINITY;
    for (int j = 0; j < V; ++j) zmax = fmaxf(zmax, yr[(size_t)j * C] / temperature);
    float se = 0.f;
    for (int j = 0; j < V; ++j) se += expf(yr[(size_t)j * C] / temperature - zmax);
    const float lse = logf(se);
    for (int j = 0; j < V; ++j) {
        const float z = yr[(size_t)j * C] / temperature - zmax;
        const float p = expf(z) / se;
        dr[(size_t)j * C] += -g * p * ((z - lse) + H);
    }
}
