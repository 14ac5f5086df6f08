// Capacity-aware beam search on the GPU (gfx950): one wavefront per graph (or per (graph, commodity)
// for the unconstrained algorithm), beams and candidates resident in LDS.
//
// Restates src/gcn/algorithms/beamsearch_uelb.py of the reference so that the decoded paths are
// bit-identical for the same log-probabilities y:
//   * scores are fp32 and are added left to right along the path (:158-159);
//   * each iteration keeps the first K entries of the queue sorted by score descending with
//     insertion order breaking ties (Python's stable sorted(..., reverse=True), :144) -- here K rounds
//     of a wavefront arg-max over (score, candidate index);
//   * queue entries that sit on the target occupy beam slots and become result candidates (:148-150);
//     the result is the earliest found path with the strictly highest score (:173);
//   * a neighbour is skipped if it is already on the path (:153), if the edge has zero capacity at the
//     start of this commodity (:155) or, except for "unconstrained", if demand > capacity (:161; paths
//     are simple, so the per-path remaining capacity of edge (cur, n) equals that entry capacity);
//   * strict mode subtracts the demand along the chosen path before the next commodity (:123-124);
//   * standard / greedy process commodities in the order of a permutation drawn per attempt (:99, :299),
//     retry up to max_iter times and fall back to [0..9] per commodity (:131-133, :179-181).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BEAM_ALGO_STANDARD 0
#define BEAM_ALGO_DETERMINISTIC 1
#define BEAM_ALGO_GREEDY 2
#define BEAM_ALGO_UNCONSTRAINED 3
#define BEAM_CPL 4              // candidates per lane kept in registers when K * V <= 64 * BEAM_CPL

struct BeamCand {
    float score;
    int idx;      // candidate index k*V + n, or -1 if none
};

__device__ __forceinline__ bool beam_better(const BeamCand& a, const BeamCand& b) {
    if (a.idx < 0) return false;
    if (b.idx < 0) return true;
    if (a.score > b.score) return true;
    if (a.score < b.score) return false;
    return a.idx < b.idx;
}

__device__ __forceinline__ BeamCand beam_wave_best(BeamCand c) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        BeamCand o;
        o.score = __shfl_xor(c.score, off, 64);
        o.idx = __shfl_xor(c.idx, off, 64);
        if (beam_better(o, c)) c = o;
    }
    return c;
}

// The same arg-max without LDS permutes: rotations inside each row of 16 lanes (DPP), then the four row
// winners are read as scalars.  Every lane returns the wavefront's best candidate.
template <int ROT>
__device__ __forceinline__ BeamCand beam_dpp_ror(const BeamCand& c) {
    BeamCand o;
    o.score = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.score), 0x120 + ROT, 0xf, 0xf, false));
    o.idx = __builtin_amdgcn_update_dpp(0, c.idx, 0x120 + ROT, 0xf, 0xf, false);
    return o;
}
__device__ __forceinline__ BeamCand beam_wave_best_dpp(BeamCand c) {
    BeamCand o;
    o = beam_dpp_ror<8>(c); if (beam_better(o, c)) c = o;
    o = beam_dpp_ror<4>(c); if (beam_better(o, c)) c = o;
    o = beam_dpp_ror<2>(c); if (beam_better(o, c)) c = o;
    o = beam_dpp_ror<1>(c); if (beam_better(o, c)) c = o;
    BeamCand best;
    best.score = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.score), 0));
    best.idx = __builtin_amdgcn_readlane(c.idx, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        o.score = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.score), 16 * row));
        o.idx = __builtin_amdgcn_readlane(c.idx, 16 * row);
        if (beam_better(o, best)) best = o;
    }
    return best;
}

// LDS carve (bytes), all regions 16-byte aligned:
//   cand_score float[K*V] | cand_ok u8[K*V] | path u16[2][K][V] | visited u32[2][K][VW] |
//   beam_node int[2][K] | beam_score float[2][K] | best_path u16[V] | wcap float[V*V] (optional) |
//   adjacency (optional): off int[K+1] | adj_ptr int[V+1] | adj_col u16[nnzc]
struct BeamLds {
    float* cand_score;
    unsigned char* cand_ok;
    unsigned short* path;
    unsigned int* visited;
    int* beam_node;
    float* beam_score;
    unsigned short* best_path;
    float* wcap;
    int* off;                   // first compact candidate of every beam (adjacency lists in use)
    int* adj_ptr;               // CSR of the graph's edges with non-zero initial capacity, columns ascending; nullptr = not in use
    unsigned short* adj_col;
};

__host__ __device__ inline size_t beam_align16(size_t x) { return (x + 15) & ~(size_t)15; }

// Adjacency lists pay when the dense candidate grid K x V does not fit the registers of a wavefront; they hold up to
// 24 edges per node on average (denser graphs, or more than 2048 nodes, keep the dense scan).
__host__ __device__ inline int beam_adj_capacity(int V, int K) {
    return (K * V > 64 * BEAM_CPL && K <= 64 && V <= 2048) ? 24 * V : 0;
}
__host__ __device__ inline size_t beam_lds_bytes(int V, int K, int cap_in_lds, int nnzc = 0) {
    const int VW = (V + 31) / 32;
    size_t s = 0;
    s += beam_align16((size_t)K * V * 4);
    s += beam_align16((size_t)K * V);
    s += beam_align16((size_t)2 * K * V * 2);
    s += beam_align16((size_t)2 * K * VW * 4);
    s += beam_align16((size_t)2 * K * 4);
    s += beam_align16((size_t)2 * K * 4);
    s += beam_align16((size_t)V * 2);
    if (cap_in_lds) s += beam_align16((size_t)V * V * 4);
    if (nnzc) s += beam_align16((size_t)(K + 1) * 4) + beam_align16((size_t)(V + 1) * 4) + beam_align16((size_t)nnzc * 2);
    return s;
}

__device__ inline BeamLds beam_carve(char* base, int V, int K, int cap_in_lds, int nnzc = 0) {
    const int VW = (V + 31) / 32;
    BeamLds l;
    l.cand_score = (float*)base; base += beam_align16((size_t)K * V * 4);
    l.cand_ok = (unsigned char*)base; base += beam_align16((size_t)K * V);
    l.path = (unsigned short*)base; base += beam_align16((size_t)2 * K * V * 2);
    l.visited = (unsigned int*)base; base += beam_align16((size_t)2 * K * VW * 4);
    l.beam_node = (int*)base; base += beam_align16((size_t)2 * K * 4);
    l.beam_score = (float*)base; base += beam_align16((size_t)2 * K * 4);
    l.best_path = (unsigned short*)base; base += beam_align16((size_t)V * 2);
    l.wcap = cap_in_lds ? (float*)base : nullptr;
    if (cap_in_lds) base += beam_align16((size_t)V * V * 4);
    l.off = nullptr; l.adj_ptr = nullptr; l.adj_col = nullptr;
    if (nnzc) {
        l.off = (int*)base; base += beam_align16((size_t)(K + 1) * 4);
        l.adj_ptr = (int*)base; base += beam_align16((size_t)(V + 1) * 4);
        l.adj_col = (unsigned short*)base;
    }
    return l;
}

// The graph's adjacency lists from global memory into LDS; returns the number of edges, or -1 when the lists are not
// available (no workspace, or the graph has more edges than the lists hold): the caller then clears L.adj_ptr.
__device__ inline int beam_load_adjacency(BeamLds& L, const int* __restrict__ gptr, const unsigned short* __restrict__ gcol, int V) {
    const int lane = threadIdx.x;
    const int nnz = (L.adj_ptr != nullptr && gptr != nullptr) ? gptr[V] : -1;
    if (nnz < 0) { L.adj_ptr = nullptr; L.adj_col = nullptr; return -1; }
    for (int t = lane; t <= V; t += 64) L.adj_ptr[t] = gptr[t];
    for (int t = lane; t < nnz; t += 64) L.adj_col[t] = gcol[t];
    __syncthreads();
    return nnz;
}

// CSR of the edges with non-zero capacity, one workgroup per graph: adj_ptr [B][V + 1] (entry V = number of edges, or
// -1 if they exceed nnzc), adj_col / adj_row [B][nnzc] with the columns of every row ascending.  V <= 2048.
__global__ __launch_bounds__(256) void k_beam_adjacency(int V, int nnzc, const float* __restrict__ cap, int* __restrict__ adj_ptr,
                                                       unsigned short* __restrict__ adj_col, unsigned short* __restrict__ adj_row) {
    __shared__ int cnt[2049];
    const int b = blockIdx.x;
    const float* cb = cap + (size_t)b * V * V;
    for (int r = threadIdx.x; r < V; r += blockDim.x) {
        int n = 0;
        for (int j = 0; j < V; ++j) n += cb[(size_t)r * V + j] != 0.f ? 1 : 0;
        cnt[r + 1] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[0] = 0;
        for (int r = 0; r < V; ++r) cnt[r + 1] += cnt[r];
    }
    __syncthreads();
    const int total = cnt[V];
    int* pb = adj_ptr + (size_t)b * (V + 1);
    for (int r = threadIdx.x; r < V; r += blockDim.x) pb[r] = cnt[r];
    if (threadIdx.x == 0) pb[V] = total <= nnzc ? total : -1;
    if (total > nnzc) return;
    for (int r = threadIdx.x; r < V; r += blockDim.x) {
        int o = cnt[r];
        for (int j = 0; j < V; ++j)
            if (cb[(size_t)r * V + j] != 0.f) {
                adj_col[(size_t)b * nnzc + o] = (unsigned short)j;
                adj_row[(size_t)b * nnzc + o] = (unsigned short)r;
                ++o;
            }
    }
}

// One commodity.  capm: [V][V] capacities valid at the start of this commodity (LDS or global).
// yb: y + b*V*V*C + c, element (i, j) at yb[(i*V + j)*C].  Returns the length of the best path
// (written to L.best_path), 0 if none.  All 64 lanes must call it.
__device__ inline int beam_commodity(const BeamLds& L, const float* capm, const float* __restrict__ yb, int V, int C,
                                     int K, int src, int dst, float demand, int check_demand) {
    const int lane = threadIdx.x;
    const int VW = (V + 31) / 32;
    int gen = 0, nb = 1, len = 1;
    bool have_best = false;
    float best_score = 0.f;
    int best_len = 0;
    if (lane == 0) {
        L.beam_node[0] = src;
        L.beam_score[0] = 0.f;
        L.path[0] = (unsigned short)src;
    }
    for (int w = lane; w < VW; w += 64) L.visited[w] = (w == (src >> 5)) ? (1u << (src & 31)) : 0u;
    __syncthreads();
    while (nb > 0) {
        const unsigned short* gpath = L.path + (size_t)gen * K * V;
        const unsigned int* gvis = L.visited + (size_t)gen * K * VW;
        const int* gnode = L.beam_node + gen * K;
        const float* gscore = L.beam_score + gen * K;
        // beams on the target become result candidates, in rank order, strict > keeps the earliest
        for (int r = 0; r < nb; ++r) {
            if (gnode[r] == dst) {
                const float s = gscore[r];
                if (!have_best || s > best_score) {
                    have_best = true; best_score = s; best_len = len;
                    for (int t = lane; t < len; t += 64) L.best_path[t] = gpath[(size_t)r * V + t];
                }
            }
        }
        // expansion: candidate (k, n) in beam-rank order, n ascending
        const int ncand = nb * V;
        const int ngen = gen ^ 1;
        unsigned short* npath = L.path + (size_t)ngen * K * V;
        unsigned int* nvis = L.visited + (size_t)ngen * K * VW;
        int nnb = 0;
        // With adjacency lists the candidates of a step are the out-edges of the beams' nodes, in the same (beam rank,
        // neighbour ascending) order as the dense grid and with the same index k * V + n for the tie-break: every other
        // entry of the grid fails the capacity test anyway.  total = their number.
        int total = -1;
        if (L.adj_ptr != nullptr) {
            int inc = 0;
            if (lane < nb) { const int node = gnode[lane]; inc = node == dst ? 0 : L.adj_ptr[node + 1] - L.adj_ptr[node]; }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
            if (lane < nb) L.off[lane + 1] = inc;
            if (lane == 0) L.off[0] = 0;
            __syncthreads();
            total = L.off[nb];
            if (total == 0) break;
        }
        if (K * V <= 64 * BEAM_CPL || (total >= 0 && total <= 64 * BEAM_CPL)) {
            // candidates stay in registers (a lane owns four of them); no LDS traffic and no barrier per round
            float cs[BEAM_CPL];
            bool ck[BEAM_CPL];
            int cidx[BEAM_CPL];
            int any = 0;
#pragma unroll
            for (int j = 0; j < BEAM_CPL; ++j) {
                const int ci = lane + 64 * j;
                bool ok = ci < (total >= 0 ? total : ncand);
                float sc = 0.f;
                int idx = -1;
                if (ok) {
                    int k, n;
                    if (total >= 0) {
                        k = 0;
                        while (k + 1 < nb && ci >= L.off[k + 1]) ++k;
                        n = L.adj_col[L.adj_ptr[gnode[k]] + ci - L.off[k]];
                    } else {
                        k = ci / V; n = ci - k * V;
                    }
                    idx = k * V + n;
                    const int node = gnode[k];
                    ok = node != dst && !((gvis[(size_t)k * VW + (n >> 5)] >> (n & 31)) & 1u);
                    if (ok) {
                        const float cv = capm[node * V + n];
                        ok = (cv != 0.f) && (!check_demand || demand <= cv);
                        if (ok) sc = gscore[k] + yb[((size_t)node * V + n) * C];
                    }
                }
                cs[j] = sc; ck[j] = ok; cidx[j] = idx;
                any |= ok ? 1 : 0;
            }
            if (!__any(any)) break;
            for (int r = 0; r < K; ++r) {
                BeamCand c;
                c.idx = -1; c.score = 0.f;
#pragma unroll
                for (int j = 0; j < BEAM_CPL; ++j) {
                    if (ck[j]) {
                        BeamCand o;
                        o.idx = cidx[j]; o.score = cs[j];
                        if (beam_better(o, c)) c = o;
                    }
                }
                c = beam_wave_best_dpp(c);
                if (c.idx < 0) break;
#pragma unroll
                for (int j = 0; j < BEAM_CPL; ++j)
                    if (cidx[j] == c.idx) ck[j] = false;
                const int k = c.idx / V, n = c.idx - k * V;
                if (lane == 0) {
                    L.beam_node[ngen * K + r] = n;
                    L.beam_score[ngen * K + r] = c.score;
                    npath[(size_t)r * V + len] = (unsigned short)n;
                }
                for (int t = lane; t < len; t += 64) npath[(size_t)r * V + t] = gpath[(size_t)k * V + t];
                for (int w = lane; w < VW; w += 64)
                    nvis[(size_t)r * VW + w] = gvis[(size_t)k * VW + w] | ((w == (n >> 5)) ? (1u << (n & 31)) : 0u);
                nnb = r + 1;
            }
        } else {
            int any = 0;
            for (int idx = lane; idx < ncand; idx += 64) {
                const int k = idx / V, n = idx - k * V;
                const int node = gnode[k];
                bool ok = node != dst;
                ok = ok && !((gvis[(size_t)k * VW + (n >> 5)] >> (n & 31)) & 1u);
                float sc = 0.f;
                if (ok) {
                    const float cv = capm[node * V + n];
                    ok = (cv != 0.f) && (!check_demand || demand <= cv);
                    if (ok) sc = gscore[k] + yb[((size_t)node * V + n) * C];
                }
                L.cand_score[idx] = sc;
                L.cand_ok[idx] = ok ? 1 : 0;
                any |= ok ? 1 : 0;
            }
            __syncthreads();
            if (!__any(any)) break;
            // stable top-K: K rounds of arg-max over (score desc, idx asc)
            for (int r = 0; r < K; ++r) {
                BeamCand c;
                c.idx = -1; c.score = 0.f;
                for (int idx = lane; idx < ncand; idx += 64) {
                    if (L.cand_ok[idx]) {
                        BeamCand o;
                        o.idx = idx; o.score = L.cand_score[idx];
                        if (beam_better(o, c)) c = o;
                    }
                }
                c = beam_wave_best(c);
                if (c.idx < 0) break;
                const int k = c.idx / V, n = c.idx - k * V;
                if (lane == 0) {
                    L.cand_ok[c.idx] = 0;
                    L.beam_node[ngen * K + r] = n;
                    L.beam_score[ngen * K + r] = c.score;
                    npath[(size_t)r * V + len] = (unsigned short)n;
                }
                for (int t = lane; t < len; t += 64) npath[(size_t)r * V + t] = gpath[(size_t)k * V + t];
                for (int w = lane; w < VW; w += 64)
                    nvis[(size_t)r * VW + w] = gvis[(size_t)k * VW + w] | ((w == (n >> 5)) ? (1u << (n & 31)) : 0u);
                nnb = r + 1;
                __syncthreads();
            }
        }
        gen = ngen; nb = nnb; len += 1;
        __syncthreads();
    }
    __syncthreads();
    return have_best ? best_len : 0;
}

// jobs: graphs (algorithms 0..2) or graph x commodity (algorithm 3).  blockDim = 64.
// perms: [n_perms][C] int32; perm_start[b]: index of graph b's first permutation.
// run_mask (nullable): graphs with run_mask[b] == 0 are skipped (outputs untouched).
// capws: [B][V][V] floats of global scratch, used as the working capacity in strict mode when the
// matrix does not fit LDS.  paths: [B][C][PL] int32 padded with -1, PL = max(V, 10).
__global__ __launch_bounds__(64) void k_beam_search(
    int B, int V, int C, int K, int algo, int strict, int max_iter, const float* __restrict__ y,
    const float* __restrict__ cap, const long long* __restrict__ comm, const int* __restrict__ perms, int n_perms,
    const int* __restrict__ perm_start, const unsigned char* __restrict__ run_mask, float* __restrict__ capws,
    int cap_in_lds, int* __restrict__ paths, int* __restrict__ path_len, int* __restrict__ feasible,
    int* __restrict__ attempts, const int* __restrict__ spec_paths, const int* __restrict__ spec_len,
    const int* __restrict__ adj_ptr, const unsigned short* __restrict__ adj_col, const unsigned short* __restrict__ adj_row,
    int nnzc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int PL = V > 10 ? V : 10;
    BeamLds L = beam_carve(smem, V, K, cap_in_lds && strict && algo != BEAM_ALGO_UNCONSTRAINED, nnzc);
    if (algo == BEAM_ALGO_UNCONSTRAINED) {
        const int b = blockIdx.x / C, c = blockIdx.x - b * C;
        if (run_mask && !run_mask[b]) return;
        beam_load_adjacency(L, adj_ptr ? adj_ptr + (size_t)b * (V + 1) : nullptr, adj_col ? adj_col + (size_t)b * nnzc : nullptr, V);
        const long long* cm = comm + ((size_t)b * C + c) * 3;
        const int len = beam_commodity(L, cap + (size_t)b * V * V, y + (size_t)b * V * V * C + c, V, C, K, (int)cm[0],
                                       (int)cm[1], 0.f, 0);
        int* pb = paths + ((size_t)b * C + c) * PL;
        for (int t = lane; t < PL; t += 64) pb[t] = t < len ? (int)L.best_path[t] : -1;
        if (lane == 0) {
            path_len[(size_t)b * C + c] = len;
            if (c == 0) { feasible[b] = 1; attempts[b] = 0; }
        }
        return;
    }
    const int b = blockIdx.x;
    if (run_mask && !run_mask[b]) return;
    if (algo == BEAM_ALGO_GREEDY) K = 1;
    const float* cap_b = cap + (size_t)b * V * V;
    const int nnz = beam_load_adjacency(L, adj_ptr ? adj_ptr + (size_t)b * (V + 1) : nullptr, adj_col ? adj_col + (size_t)b * nnzc : nullptr, V);
    const unsigned short* rows_b = adj_row ? adj_row + (size_t)b * nnzc : nullptr;
    float* wcap = strict ? (L.wcap ? L.wcap : capws + (size_t)b * V * V) : nullptr;
    const float* yb = y + (size_t)b * V * V * C;
    const long long* cm = comm + (size_t)b * C * 3;
    int* pb = paths + (size_t)b * C * PL;
    int* plb = path_len + (size_t)b * C;
    int used = 0, ok = 0;
    for (int count = 0; count < max_iter; ++count) {
        const int* perm = nullptr;
        if (algo != BEAM_ALGO_DETERMINISTIC) {
            const int pi = perm_start[b] + count;
            if (pi >= n_perms) break;          // host supplies max_iter permutations per graph; defensive
            perm = perms + (size_t)pi * C;
        }
        ++used;
        if (strict) {
            for (int t = lane; t < V * V; t += 64) wcap[t] = cap_b[t];
        }
        __syncthreads();
        ok = 1;
        for (int pos = 0; pos < C; ++pos) {
            const int c = perm ? perm[pos] : pos;
            const int s = (int)cm[c * 3], d = (int)cm[c * 3 + 1];
            const float demand = (float)cm[c * 3 + 2];
            // A commodity's search depends on the capacities only through the edge predicate
            // (cap != 0 and demand <= cap).  If that predicate is the same on the working capacities as on the
            // initial ones, the search k_beam_spec ran against the initial capacities is this search exactly.
            bool reuse = spec_len != nullptr;
            if (reuse && strict) {
                int diff = 0;
                if (nnz >= 0) {         // edges without initial capacity are never used: only the listed ones can differ
                    for (int e = lane; e < nnz; e += 64) {
                        const int t = (int)rows_b[e] * V + (int)L.adj_col[e];
                        const float c0 = cap_b[t], c1 = wcap[t];
                        diff |= ((c0 != 0.f) && (demand <= c0)) != ((c1 != 0.f) && (demand <= c1)) ? 1 : 0;
                    }
                } else {
                    for (int t = lane; t < V * V; t += 64) {
                        const float c0 = cap_b[t], c1 = wcap[t];
                        diff |= ((c0 != 0.f) && (demand <= c0)) != ((c1 != 0.f) && (demand <= c1)) ? 1 : 0;
                    }
                }
                reuse = !__any(diff);
            }
            int len;
            if (reuse) {
                len = spec_len[(size_t)b * C + c];
                for (int t = lane; t < len; t += 64) L.best_path[t] = (unsigned short)spec_paths[((size_t)b * C + c) * PL + t];
                __syncthreads();
            } else {
                len = beam_commodity(L, strict ? wcap : cap_b, yb + c, V, C, K, s, d, demand, 1);
            }
            if (len == 0) { ok = 0; break; }
            for (int t = lane; t < PL; t += 64) pb[(size_t)c * PL + t] = t < len ? (int)L.best_path[t] : -1;
            if (lane == 0) plb[c] = len;
            if (strict) {
                for (int t = lane; t + 1 < len; t += 64) {
                    const int u = L.best_path[t], v = L.best_path[t + 1];
                    wcap[u * V + v] -= demand;      // a simple path uses each directed edge once
                }
            }
            __syncthreads();
        }
        if (ok) break;
        if (algo == BEAM_ALGO_DETERMINISTIC) { used = max_iter; break; }   // identical retries, identical result
    }
    if (!ok) {
        for (int t = lane; t < C * PL; t += 64) {
            const int col = t % PL;
            pb[t] = col < 10 ? col : -1;
        }
        for (int c = lane; c < C; c += 64) plb[c] = 10;
    }
    if (lane == 0) { feasible[b] = ok; attempts[b] = used; }
}

// y from the network output, BeamSearchAlgorithm.__init__ (:14-21).
// five_d != 0: logits [rows][2] -> y[row] = log_softmax(.)[1]; thread per row.
// Every (graph, commodity) searched against the graph's initial capacities, all in parallel: one wavefront
// per job.  k_beam_search then walks the commodities in order and keeps a result whenever the capacity
// updates so far leave the commodity's edge predicate unchanged (the common case), re-searching otherwise.
__global__ __launch_bounds__(64) void k_beam_spec(
    int B, int V, int C, int K, int algo, const float* __restrict__ y, const float* __restrict__ cap,
    const long long* __restrict__ comm, const unsigned char* __restrict__ run_mask, int cap_in_lds,
    int* __restrict__ spec_paths, int* __restrict__ spec_len, const int* __restrict__ adj_ptr,
    const unsigned short* __restrict__ adj_col, int nnzc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int PL = V > 10 ? V : 10;
    BeamLds L = beam_carve(smem, V, K, cap_in_lds, nnzc);
    const int b = blockIdx.x / C, c = blockIdx.x - b * C;
    if (run_mask && !run_mask[b]) return;
    if (algo == BEAM_ALGO_GREEDY) K = 1;
    const float* cap_b = cap + (size_t)b * V * V;
    beam_load_adjacency(L, adj_ptr ? adj_ptr + (size_t)b * (V + 1) : nullptr, adj_col ? adj_col + (size_t)b * nnzc : nullptr, V);
    if (L.wcap) {
        for (int t = lane; t < V * V; t += 64) L.wcap[t] = cap_b[t];
        __syncthreads();
    }
    const long long* cm = comm + ((size_t)b * C + c) * 3;
    const int len = beam_commodity(L, L.wcap ? L.wcap : cap_b, y + (size_t)b * V * V * C + c, V, C, K, (int)cm[0], (int)cm[1],
                                   (float)cm[2], 1);
    for (int t = lane; t < len; t += 64) spec_paths[((size_t)b * C + c) * PL + t] = (int)L.best_path[t];
    if (lane == 0) spec_len[(size_t)b * C + c] = len;
}

__global__ void k_beam_scores_2class(size_t rows, const float* __restrict__ logits, float* __restrict__ y) {
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        const float l0 = logits[r * 2], l1 = logits[r * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float lse = logf(expf(l0 - mx) + expf(l1 - mx));
        float v = (l1 - mx) - lse;
        if (v == 0.f) v = -1e-20f;
        y[r] = v;
    }
}
// scores [B][V_i][V_j][C] -> log_softmax over j; thread per (b, i, c).
__global__ void k_beam_scores_over_j(int B, int V, int C, const float* __restrict__ s, float* __restrict__ y) {
    const size_t n = (size_t)B * V * C;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t bi = t / C;
        const int c = (int)(t - bi * C);
        const float* p = s + bi * V * C + c;
        float mx = -INFINITY;
        for (int j = 0; j < V; ++j) mx = fmaxf(mx, p[(size_t)j * C]);
        float sum = 0.f;
        for (int j = 0; j < V; ++j) sum += expf(p[(size_t)j * C] - mx);
        const float lse = logf(sum);
        float* o = y + bi * V * C + c;
        for (int j = 0; j < V; ++j) {
            float v = (p[(size_t)j * C] - mx) - lse;
            if (v == 0.f) v = -1e-20f;
            o[(size_t)j * C] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Load factor of decoded paths, kept on the GPU right behind the beam search.
// Restates mean_feasible_load_factor (src/common/graph/graph_utils.py:90-192): the demand of every
// commodity is added along the edges of its path (integers, so the order of the atomic adds does not
// matter), load = usage / capacity, +inf where a zero-capacity edge carries demand, 0 where an unused
// edge has zero capacity; max_lf[b] is the maximum over the graph's edges.  Also reports, per commodity,
// whether its path ends on the destination (Evaluator._compute_completion_and_approx,
// src/gcn/train/evaluator.py:45-52).
// ------------------------------------------------------------------------------------------------
__global__ void k_path_usage(int B, int V, int C, int PL, const int* __restrict__ paths, const int* __restrict__ path_len,
                             const long long* __restrict__ comm, int* __restrict__ usage, unsigned char* __restrict__ complete) {
    const size_t n = (size_t)B * C * PL;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = t / PL;
        const int k = (int)(t - bc * PL);
        const int len = path_len[bc];
        const int b = (int)(bc / C);
        const int* p = paths + bc * PL;
        if (k == 0) complete[bc] = (len > 0 && p[len - 1] == (int)comm[bc * 3 + 1]) ? 1 : 0;
        if (k + 1 < len) {
            const int u = p[k], v = p[k + 1];
            if (u >= 0 && u < V && v >= 0 && v < V) atomicAdd(&usage[((size_t)b * V + u) * V + v], (int)comm[bc * 3 + 2]);
        }
    }
}
// one workgroup per graph
__global__ __launch_bounds__(256) void k_max_load_factor(int V, const int* __restrict__ usage, const float* __restrict__ cap,
                                                        float* __restrict__ max_lf) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    float m = -INFINITY;
    for (int t = threadIdx.x; t < V * V; t += blockDim.x) {
        const int u = usage[(size_t)b * V * V + t];
        const float c = cap[(size_t)b * V * V + t];
        const float lf = c == 0.f ? (u == 0 ? 0.f : INFINITY) : (float)u / c;
        m = fmaxf(m, lf);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) max_lf[b] = red[0];
}
