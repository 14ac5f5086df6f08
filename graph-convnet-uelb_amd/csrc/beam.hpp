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
//   cand_score float[K*V] | cand_ok u8[K*V] | bp_node u16[V][K] | bp_par u16[V][K] | visited u32[2][K][VW] |
//   beam_node int[2][K] | beam_score float[2][K] | best_path u16[V] | wcap float[V*V] (optional) |
//   adjacency (optional): adj_ptr int[V+1] | adj_col u16[nnzc] | cap_e float[nnzc] | y_e float[nnzc]
struct BeamLds {
    float* cand_score;
    unsigned char* cand_ok;
    unsigned short* bp_node;    // back pointers: node at position t of the beam of rank r at depth t ...
    unsigned short* bp_par;     // ... and the rank of the beam it extends
    unsigned int* visited;
    int* beam_node;
    float* beam_score;
    unsigned short* best_path;
    float* wcap;
    int* adj_ptr;               // CSR of the graph's edges with non-zero initial capacity, columns ascending; nullptr = not in use
    unsigned short* adj_col;
    float* cap_e;               // capacities per list entry (the working capacities in strict mode)
    float* y_e;                 // scores of the current commodity per list entry
};

__host__ __device__ inline size_t beam_align16(size_t x) { return (x + 15) & ~(size_t)15; }

// Adjacency lists hold up to 24 edges per node on average (denser graphs, or more than 2048 nodes, keep the dense
// scan).  They pay at every size: at 14 nodes the K x V grid fits the registers, but a step over the beams' out-edges
// (K x 3 of them) is a third of the instructions of a step over the grid (0.25 -> 0.125 ms for 20 NSFNet graphs).
__host__ __device__ inline int beam_adj_capacity(int V, int K) {
    return (K <= 64 && V <= 2048) ? 24 * V : 0;
}
__host__ __device__ inline size_t beam_lds_bytes(int V, int K, int cap_in_lds, int nnzc = 0) {
    const int VW = (V + 31) / 32;
    size_t s = 0;
    s += beam_align16((size_t)K * V * 4);
    s += beam_align16((size_t)K * V);
    s += beam_align16((size_t)K * V * 2);
    s += beam_align16((size_t)K * V * 2);
    s += beam_align16((size_t)2 * K * VW * 4);
    s += beam_align16((size_t)2 * K * 4);
    s += beam_align16((size_t)2 * K * 4);
    s += beam_align16((size_t)V * 2);
    if (cap_in_lds) s += beam_align16((size_t)V * V * 4);
    if (nnzc) s += beam_align16((size_t)(V + 1) * 4) + beam_align16((size_t)nnzc * 2) + 2 * beam_align16((size_t)nnzc * 4);
    return s;
}

__device__ inline BeamLds beam_carve(char* base, int V, int K, int cap_in_lds, int nnzc = 0) {
    const int VW = (V + 31) / 32;
    BeamLds l;
    l.cand_score = (float*)base; base += beam_align16((size_t)K * V * 4);
    l.cand_ok = (unsigned char*)base; base += beam_align16((size_t)K * V);
    l.bp_node = (unsigned short*)base; base += beam_align16((size_t)K * V * 2);
    l.bp_par = (unsigned short*)base; base += beam_align16((size_t)K * V * 2);
    l.visited = (unsigned int*)base; base += beam_align16((size_t)2 * K * VW * 4);
    l.beam_node = (int*)base; base += beam_align16((size_t)2 * K * 4);
    l.beam_score = (float*)base; base += beam_align16((size_t)2 * K * 4);
    l.best_path = (unsigned short*)base; base += beam_align16((size_t)V * 2);
    l.wcap = cap_in_lds ? (float*)base : nullptr;
    if (cap_in_lds) base += beam_align16((size_t)V * V * 4);
    l.adj_ptr = nullptr; l.adj_col = nullptr; l.cap_e = nullptr; l.y_e = nullptr;
    if (nnzc) {
        l.adj_ptr = (int*)base; base += beam_align16((size_t)(V + 1) * 4);
        l.adj_col = (unsigned short*)base; base += beam_align16((size_t)nnzc * 2);
        l.cap_e = (float*)base; base += beam_align16((size_t)nnzc * 4);
        l.y_e = (float*)base;
    }
    return l;
}

// The graph's adjacency lists from global memory into LDS; returns the number of edges, or -1 when the lists are not
// available (no workspace, or the graph has more edges than the lists hold): the caller then clears L.adj_ptr.
__device__ inline int beam_load_adjacency(BeamLds& L, const int* __restrict__ gptr, const unsigned short* __restrict__ gcol, int V) {
    const int lane = threadIdx.x;
    const int nnz = (L.adj_ptr != nullptr && gptr != nullptr) ? gptr[V] : -1;
    if (nnz < 0) { L.adj_ptr = nullptr; L.adj_col = nullptr; L.cap_e = nullptr; L.y_e = nullptr; return -1; }
    for (int t = lane; t <= V; t += 64) L.adj_ptr[t] = gptr[t];
    for (int t = lane; t < nnz; t += 64) L.adj_col[t] = gcol[t];
    __syncthreads();
    return nnz;
}

// Capacities per list entry from a dense [V][V] matrix (lanes over rows).
__device__ inline void beam_fill_edge_caps(const BeamLds& L, const float* __restrict__ capm, int V) {
    if (L.cap_e == nullptr) return;
    for (int r = threadIdx.x; r < V; r += 64)
        for (int e = L.adj_ptr[r]; e < L.adj_ptr[r + 1]; ++e) L.cap_e[e] = capm[(size_t)r * V + L.adj_col[e]];
    __syncthreads();
}

// Scores of one commodity per list entry from the gathered table ye_bc [nnz] into LDS.  Returns (to every lane) whether
// all of them are <= 0, as log-probabilities are: a path's score then never rises when the path is extended.
__device__ inline int beam_load_edge_scores(const BeamLds& L, const float* __restrict__ ye_bc, int nnz) {
    if (L.y_e == nullptr) return 0;
    int pos = 0;
    for (int e = threadIdx.x; e < nnz; e += 64) {
        const float v = ye_bc[e];
        L.y_e[e] = v;
        pos |= !(v <= 0.f) ? 1 : 0;          // a NaN counts as "may rise"
    }
    __syncthreads();
    return __any(pos) ? 0 : 1;
}

// y_e [B][C][nnzc]: the scores y[b][row][col][c] of every list entry, gathered once so that a search reads its scores
// from LDS instead of one scattered cache line per candidate and step.  Grid (B, chunks); consecutive threads take
// consecutive commodities of one entry (contiguous in y).
__global__ __launch_bounds__(256) void k_beam_gather_y(int V, int C, int nnzc, const float* __restrict__ y, const int* __restrict__ adj_ptr,
                                                      const unsigned short* __restrict__ adj_col, const unsigned short* __restrict__ adj_row,
                                                      float* __restrict__ y_e) {
    const int b = blockIdx.x;
    const int nnz = adj_ptr[(size_t)b * (V + 1) + V];
    if (nnz < 0) return;
    const size_t n = (size_t)nnz * C;
    for (size_t t = (size_t)blockIdx.y * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.y * blockDim.x) {
        const int e = (int)(t / C), c = (int)(t - (size_t)e * C);
        const int row = adj_row[(size_t)b * nnzc + e], col = adj_col[(size_t)b * nnzc + e];
        y_e[((size_t)b * C + c) * nnzc + e] = y[(((size_t)b * V + row) * V + col) * C + c];
    }
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

// Inclusive prefix sum over the lanes of a wavefront (DPP shifts inside each row of 16 lanes, then the row totals).
__device__ __forceinline__ int beam_wave_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1, lanes shifted in read as 0
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31), t2 = __builtin_amdgcn_readlane(x, 47);
    const int row = (int)threadIdx.x >> 4;
    return x + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}

// One commodity.  capm: [V][V] capacities valid at the start of this commodity (LDS or global); cap_e (nullable): the
// same capacities per adjacency-list entry, in LDS.  yb: y + b*V*V*C + c, element (i, j) at yb[(i*V + j)*C].
// Returns the length of the best path (written to L.best_path), 0 if none.  All 64 lanes must call it.
// Paths are kept as back pointers: bp_node[t][r] is the node at position t of the beam of rank r at that depth and
// bp_par[t][r] the rank of the beam it extends; a path is written out only when a beam on the target becomes the result.
// monotone != 0: every edge score of this commodity is <= 0 (beam_load_edge_scores).  Scores are then non-increasing
// along a path (fp32 addition of a non-positive term never rounds upwards), so once no beam off the target has a score
// above the best result's, no later result can be strictly better and the search ends with the result the reference's
// run-until-the-queue-is-empty loop (:141-169) returns - typically a few steps past the shortest path instead of a walk
// over up to V nodes.
__device__ inline int beam_commodity(const BeamLds& L, const float* capm, const float* cap_e, const float* __restrict__ yb, int V,
                                     int C, int K, int src, int dst, float demand, int check_demand, int monotone = 0) {
    const int lane = threadIdx.x;
    const int VW = (V + 31) / 32;
    int gen = 0, nb = 1, len = 1;
    bool have_best = false;
    float best_score = 0.f;
    int best_len = 0;
    if (lane == 0) {
        L.beam_node[0] = src;
        L.beam_score[0] = 0.f;
    }
    for (int w = lane; w < VW; w += 64) L.visited[w] = (w == (src >> 5)) ? (1u << (src & 31)) : 0u;
    __syncthreads();
    while (nb > 0) {
        const unsigned int* gvis = L.visited + (size_t)gen * K * VW;
        const int* gnode = L.beam_node + gen * K;
        const float* gscore = L.beam_score + gen * K;
        // this lane's beam (lanes beyond the queue hold none); K <= 64 whenever the lists or the register path are in use
        const int my_node = lane < nb ? gnode[lane] : -1;
        // beams on the target become result candidates, in rank order, strict > keeps the earliest
        if (K > 64 || __any(my_node == dst)) {
            for (int r = 0; r < nb; ++r) {
                if (gnode[r] == dst) {
                    const float s = gscore[r];
                    if (!have_best || s > best_score) {
                        have_best = true; best_score = s; best_len = len;
                        if (lane == 0) {
                            int rr = r;
                            for (int t = len - 1; t > 0; --t) {
                                L.best_path[t] = L.bp_node[(size_t)t * K + rr];
                                rr = L.bp_par[(size_t)t * K + rr];
                            }
                            L.best_path[0] = (unsigned short)src;
                        }
                    }
                }
            }
        }
        if (monotone && have_best && K <= 64) {
            const float my_score = lane < nb ? gscore[lane] : 0.f;
            if (!__any(my_node >= 0 && my_node != dst && my_score > best_score)) break;
        }
        // expansion: candidate (k, n) in beam-rank order, n ascending
        const int ncand = nb * V;
        const int ngen = gen ^ 1;
        unsigned int* nvis = L.visited + (size_t)ngen * K * VW;
        int nnb = 0;
        // the beam of rank r of the next depth: node n reached from the beam of rank k
        auto commit = [&](int r, int k, int n, float score) {
            if (lane == 0) {
                L.beam_node[ngen * K + r] = n;
                L.beam_score[ngen * K + r] = score;
                L.bp_node[(size_t)len * K + r] = (unsigned short)n;
                L.bp_par[(size_t)len * K + r] = (unsigned short)k;
            }
            for (int w = lane; w < VW; w += 64)
                nvis[(size_t)r * VW + w] = gvis[(size_t)k * VW + w] | ((w == (n >> 5)) ? (1u << (n & 31)) : 0u);
        };
        // With adjacency lists the candidates of a step are the out-edges of the beams' nodes, in the same (beam rank,
        // neighbour ascending) order as the dense grid and with the same index k * V + n for the tie-break: every other
        // entry of the grid fails the capacity test anyway.  total = their number; lane k holds the running count
        // through beam k (inc) and the start of its list (my_ptr).
        int total = -1, inc = 0, my_ptr = 0;
        if (L.adj_ptr != nullptr) {
            if (my_node >= 0 && my_node != dst) { my_ptr = L.adj_ptr[my_node]; inc = L.adj_ptr[my_node + 1] - my_ptr; }
            inc = beam_wave_scan(inc);
            total = __builtin_amdgcn_readlane(inc, 63);
            if (total == 0) break;
        }
        if (total >= 0 && total <= 64) {
            // At most one candidate per lane (the usual case on sparse graphs: K beams x a few out-edges).  A lone
            // wavefront pays for every instruction it issues, so instead of K rounds of arg-max every candidate counts
            // the candidates that beat it (score descending, index ascending: the order of the stable sort): that count
            // is its rank in the next queue, and the first K ranks write their beams side by side.
            int k = 0, e = 0, node = 0, prev = 0;
            for (int kb = 0; kb < nb; ++kb) {
                const int upto = __builtin_amdgcn_readlane(inc, kb), ptr = __builtin_amdgcn_readlane(my_ptr, kb);
                const int nd = __builtin_amdgcn_readlane(my_node, kb);
                if (lane >= prev && lane < upto) { k = kb; e = ptr + lane - prev; node = nd; }
                prev = upto;
            }
            bool ok = lane < total;
            int n = 0, idx = 0x7fffffff;
            float sc = 0.f;
            if (ok) {
                n = L.adj_col[e];
                idx = k * V + n;
                const float yv = L.y_e != nullptr ? L.y_e[e] : yb[((size_t)node * V + n) * C];
                const float cv = cap_e != nullptr ? cap_e[e] : capm[node * V + n];
                ok = !((gvis[(size_t)k * VW + (n >> 5)] >> (n & 31)) & 1u) && (cv != 0.f) && (!check_demand || demand <= cv);
                sc = gscore[k] + yv;
            }
            unsigned long long valid = __ballot(ok);
            if (valid == 0ull) break;
            nnb = min(K, (int)__popcll(valid));
            int rank = 0;
            while (valid) {
                const int l = __builtin_ctzll(valid);
                valid &= valid - 1;
                const float os = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), l));
                const int oi = __builtin_amdgcn_readlane(idx, l);
                rank += (os > sc || (os == sc && oi < idx)) ? 1 : 0;
            }
            if (ok && rank < K) {
                L.beam_node[ngen * K + rank] = n;
                L.beam_score[ngen * K + rank] = sc;
                L.bp_node[(size_t)len * K + rank] = (unsigned short)n;
                L.bp_par[(size_t)len * K + rank] = (unsigned short)k;
                for (int w = 0; w < VW; ++w)
                    nvis[(size_t)rank * VW + w] = gvis[(size_t)k * VW + w] | ((w == (n >> 5)) ? (1u << (n & 31)) : 0u);
            }
        } else if (K * V <= 64 * BEAM_CPL || (total >= 0 && total <= 64 * BEAM_CPL)) {
            // candidates stay in registers (a lane owns four of them); no LDS traffic and no barrier per round
            float cs[BEAM_CPL];
            bool ck[BEAM_CPL];
            int cidx[BEAM_CPL];
            int any = 0;
            int ck_k[BEAM_CPL], ck_e[BEAM_CPL], ck_node[BEAM_CPL];
            if (total >= 0) {
                // beam of every candidate: the first whose running count exceeds its position (uniform loop, scalar broadcasts)
#pragma unroll
                for (int j = 0; j < BEAM_CPL; ++j) { ck_k[j] = 0; ck_e[j] = 0; ck_node[j] = 0; }
                int prev = 0;
                for (int k = 0; k < nb; ++k) {
                    const int upto = __builtin_amdgcn_readlane(inc, k), ptr = __builtin_amdgcn_readlane(my_ptr, k);
                    const int nd = __builtin_amdgcn_readlane(my_node, k);
#pragma unroll
                    for (int j = 0; j < BEAM_CPL; ++j) {
                        const int ci = lane + 64 * j;
                        if (ci >= prev && ci < upto) { ck_k[j] = k; ck_e[j] = ptr + ci - prev; ck_node[j] = nd; }
                    }
                    prev = upto;
                }
            }
#pragma unroll
            for (int j = 0; j < BEAM_CPL; ++j) {
                const int ci = lane + 64 * j;
                bool ok = ci < (total >= 0 ? total : ncand);
                float sc = 0.f;
                int idx = -1;
                if (ok) {
                    int k, n, node;
                    if (total >= 0) {
                        k = ck_k[j]; node = ck_node[j];
                        n = L.adj_col[ck_e[j]];
                    } else {
                        k = ci / V; n = ci - k * V;
                        node = gnode[k];
                    }
                    idx = k * V + n;
                    // requested beside the capacity, used only if the edge is usable
                    const float yv = (total >= 0 && L.y_e != nullptr) ? L.y_e[ck_e[j]] : yb[((size_t)node * V + n) * C];
                    const float cv = (total >= 0 && cap_e != nullptr) ? cap_e[ck_e[j]] : capm[node * V + n];
                    ok = node != dst && !((gvis[(size_t)k * VW + (n >> 5)] >> (n & 31)) & 1u);
                    ok = ok && (cv != 0.f) && (!check_demand || demand <= cv);
                    if (ok) sc = gscore[k] + yv;
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
                commit(r, k, n, c.score);
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
                if (lane == 0) L.cand_ok[c.idx] = 0;
                commit(r, k, n, c.score);
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
//
// Rounds (round_mode != 0; strict mode with adjacency lists and the working capacities in global memory): the ordered
// walk is split over several launches with a speculative pass between them.  A launch keeps first-pass results while
// the commodity's edge predicate is the same on the working capacities as on the snapshot the first pass searched
// against; at the first commodity for which it differs, modes 1 and 2 save the graph's state (position, attempt,
// working capacities per list entry in work_e; the dense matrix lives in capws) and return, and the next speculative
// pass searches every remaining commodity against exactly those capacities, all in parallel - its result for the
// commodity that stopped the walk is therefore final, and so is every following one until the updates change a
// predicate again.  Mode 3 (last launch) searches such commodities itself, one after the other.  Same results as the
// single ordered walk.  state: [B][4] = {position, attempt, finished, attempts counted}.
//   round_mode 0: single launch (no state); 1: first round (snapshot = initial capacities); 2: later round; 3: last.
__global__ __launch_bounds__(64) void k_beam_search(
    int B, int V, int C, int K, int algo, int strict, int max_iter, const float* __restrict__ y,
    const float* __restrict__ cap, const long long* __restrict__ comm, const int* __restrict__ perms, int n_perms,
    const int* __restrict__ perm_start, const unsigned char* __restrict__ run_mask, float* __restrict__ capws,
    int cap_in_lds, int* __restrict__ paths, int* __restrict__ path_len, int* __restrict__ feasible,
    int* __restrict__ attempts, const int* __restrict__ spec_paths, const int* __restrict__ spec_len,
    const int* __restrict__ adj_ptr, const unsigned short* __restrict__ adj_col, const unsigned short* __restrict__ adj_row,
    const float* __restrict__ y_e, int nnzc, int round_mode, int* __restrict__ state, float* __restrict__ work_e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int PL = V > 10 ? V : 10;
    BeamLds L = beam_carve(smem, V, K, cap_in_lds && strict && algo != BEAM_ALGO_UNCONSTRAINED, nnzc);
    if (algo == BEAM_ALGO_UNCONSTRAINED) {
        const int b = blockIdx.x / C, c = blockIdx.x - b * C;
        if (run_mask && !run_mask[b]) return;
        const int nnz_u = beam_load_adjacency(L, adj_ptr ? adj_ptr + (size_t)b * (V + 1) : nullptr, adj_col ? adj_col + (size_t)b * nnzc : nullptr, V);
        beam_fill_edge_caps(L, cap + (size_t)b * V * V, V);
        const int mono = nnz_u >= 0 ? beam_load_edge_scores(L, y_e + ((size_t)b * C + c) * nnzc, nnz_u) : 0;
        const long long* cm = comm + ((size_t)b * C + c) * 3;
        const int len = beam_commodity(L, cap + (size_t)b * V * V, L.cap_e, y + (size_t)b * V * V * C + c, V, C, K, (int)cm[0],
                                       (int)cm[1], 0.f, 0, mono);
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
    int* stb = state ? state + (size_t)b * 4 : nullptr;
    if (round_mode >= 2 && stb[2]) return;                 // finished in an earlier round
    if (algo == BEAM_ALGO_GREEDY) K = 1;
    const float* cap_b = cap + (size_t)b * V * V;
    const int nnz = beam_load_adjacency(L, adj_ptr ? adj_ptr + (size_t)b * (V + 1) : nullptr, adj_col ? adj_col + (size_t)b * nnzc : nullptr, V);
    const unsigned short* rows_b = adj_row ? adj_row + (size_t)b * nnzc : nullptr;
    float* wcap = strict ? (L.wcap ? L.wcap : capws + (size_t)b * V * V) : nullptr;
    float* we_b = work_e ? work_e + (size_t)b * nnzc : nullptr;
    const float* yb = y + (size_t)b * V * V * C;
    const long long* cm = comm + (size_t)b * C * 3;
    int* pb = paths + (size_t)b * C * PL;
    int* plb = path_len + (size_t)b * C;
    // a graph without adjacency lists (denser than they hold) is walked to the end in the first round
    const bool may_stop = (round_mode == 1 || round_mode == 2) && nnz >= 0;
    const int count0 = round_mode >= 2 ? stb[1] : 0;
    const int pos0 = round_mode >= 2 ? stb[0] : 0;
    int used = round_mode >= 2 ? stb[3] : 0, ok = 0;
    for (int count = count0; count < max_iter; ++count) {
        const int* perm = nullptr;
        if (algo != BEAM_ALGO_DETERMINISTIC) {
            const int pi = perm_start[b] + count;
            if (pi >= n_perms) break;          // host supplies max_iter permutations per graph; defensive
            perm = perms + (size_t)pi * C;
        }
        // resumed: this attempt was begun by an earlier launch (its capacities are in capws / work_e, and the first
        // pass before this launch searched against them)
        const bool resumed = round_mode >= 2 && count == count0;
        if (!resumed) {
            ++used;
            if (strict) {
                for (int t = lane; t < V * V; t += 64) wcap[t] = cap_b[t];
            }
        }
        __syncthreads();
        if (resumed) {
            for (int e = lane; e < nnz; e += 64) L.cap_e[e] = we_b[e];
            __syncthreads();
        } else if (strict || count == 0) {
            beam_fill_edge_caps(L, cap_b, V);      // the working capacities, per list entry
        }
        // first-pass results are usable while they were searched against this attempt's snapshot: the initial
        // capacities (modes 0 and 1, any attempt) or the saved capacities of the resumed attempt
        const bool spec_ok = spec_len != nullptr && (round_mode <= 1 || resumed);
        ok = 1;
        for (int pos = resumed ? pos0 : 0; pos < C; ++pos) {
            const int c = perm ? perm[pos] : pos;
            const int s = (int)cm[c * 3], d = (int)cm[c * 3 + 1];
            const float demand = (float)cm[c * 3 + 2];
            // A commodity's search depends on the capacities only through the edge predicate
            // (cap != 0 and demand <= cap).  If that predicate is the same on the working capacities as on the
            // snapshot, the search the first pass ran against the snapshot is this search exactly.
            bool reuse = spec_ok;
            if (reuse && strict) {
                int diff = 0;
                if (nnz >= 0) {         // edges without initial capacity are never used: only the listed ones can differ
                    for (int e = lane; e < nnz; e += 64) {
                        const float c0 = resumed ? we_b[e] : cap_b[(int)rows_b[e] * V + (int)L.adj_col[e]], c1 = L.cap_e[e];
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
            if (!reuse && may_stop) {
                // save the attempt and leave this commodity (and the rest) to the next first pass
                for (int e = lane; e < nnz; e += 64) we_b[e] = L.cap_e[e];
                if (lane == 0) { stb[0] = pos; stb[1] = count; stb[2] = 0; stb[3] = used; }
                return;
            }
            int len;
            if (reuse) {
                len = spec_len[(size_t)b * C + c];
                for (int t = lane; t < len; t += 64) L.best_path[t] = (unsigned short)spec_paths[((size_t)b * C + c) * PL + t];
                __syncthreads();
            } else {
                const int mono = nnz >= 0 ? beam_load_edge_scores(L, y_e + ((size_t)b * C + c) * nnzc, nnz) : 0;
                len = beam_commodity(L, strict ? wcap : cap_b, L.cap_e, yb + c, V, C, K, s, d, demand, 1, mono);
            }
            if (len == 0) { ok = 0; break; }
            for (int t = lane; t < PL; t += 64) pb[(size_t)c * PL + t] = t < len ? (int)L.best_path[t] : -1;
            if (lane == 0) plb[c] = len;
            if (strict) {
                for (int t = lane; t + 1 < len; t += 64) {
                    const int u = L.best_path[t], v = L.best_path[t + 1];
                    wcap[u * V + v] -= demand;      // a simple path uses each directed edge once
                    if (L.cap_e != nullptr)
                        for (int e = L.adj_ptr[u]; e < L.adj_ptr[u + 1]; ++e)
                            if (L.adj_col[e] == v) L.cap_e[e] -= demand;
                }
            }
            __syncthreads();
        }
        if (ok) break;
        if (algo == BEAM_ALGO_DETERMINISTIC) { used = max_iter; break; }   // identical retries, identical result
        if (round_mode == 2 && count + 1 < max_iter && perm_start[b] + count + 1 < n_perms) {
            // The next attempt starts from the initial capacities, which the first-pass results at hand were not
            // searched against: set the attempt up and leave it to the next round.
            for (int t = lane; t < V * V; t += 64) wcap[t] = cap_b[t];
            __syncthreads();
            beam_fill_edge_caps(L, cap_b, V);
            for (int e = lane; e < nnz; e += 64) we_b[e] = L.cap_e[e];
            if (lane == 0) { stb[0] = 0; stb[1] = count + 1; stb[2] = 0; stb[3] = used + 1; }
            return;
        }
    }
    if (!ok) {
        for (int t = lane; t < C * PL; t += 64) {
            const int col = t % PL;
            pb[t] = col < 10 ? col : -1;
        }
        for (int c = lane; c < C; c += 64) plb[c] = 10;
    }
    if (lane == 0) {
        feasible[b] = ok; attempts[b] = used;
        if (stb) stb[2] = 1;
    }
}

// Every (graph, commodity) searched against the graph's initial capacities, all in parallel: one wavefront
// per job.  k_beam_search then walks the commodities in order and keeps a result whenever the capacity
// updates so far leave the commodity's edge predicate unchanged (the common case), re-searching otherwise.
// state / work_e / capws (rounds after the first, see k_beam_search): job (b, p) is position p of graph b's current
// attempt; positions already walked and finished graphs return at once, the others search against the saved working
// capacities.  With state == nullptr job (b, c) searches commodity c against the initial capacities.
__global__ __launch_bounds__(64) void k_beam_spec(
    int B, int V, int C, int K, int algo, const float* __restrict__ y, const float* __restrict__ cap,
    const long long* __restrict__ comm, const unsigned char* __restrict__ run_mask, int cap_in_lds,
    int* __restrict__ spec_paths, int* __restrict__ spec_len, const int* __restrict__ adj_ptr,
    const unsigned short* __restrict__ adj_col, const float* __restrict__ y_e, int nnzc,
    const int* __restrict__ state, const float* __restrict__ work_e, const float* __restrict__ capws,
    const int* __restrict__ perms, int n_perms, const int* __restrict__ perm_start) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int PL = V > 10 ? V : 10;
    const int b = blockIdx.x / C;
    int c = blockIdx.x - b * C;
    if (run_mask && !run_mask[b]) return;
    if (state != nullptr) {
        const int* stb = state + (size_t)b * 4;
        if (stb[2] || c < stb[0]) return;
        if (algo != BEAM_ALGO_DETERMINISTIC) {
            const int pi = perm_start[b] + stb[1];
            if (pi >= n_perms) return;
            c = perms[(size_t)pi * C + c];
        }
    }
    BeamLds L = beam_carve(smem, V, K, cap_in_lds, nnzc);
    if (algo == BEAM_ALGO_GREEDY) K = 1;
    const float* cap_b = cap + (size_t)b * V * V;
    const int nnz = beam_load_adjacency(L, adj_ptr ? adj_ptr + (size_t)b * (V + 1) : nullptr, adj_col ? adj_col + (size_t)b * nnzc : nullptr, V);
    const float* dense = cap_b;
    if (state != nullptr) {
        if (nnz < 0) return;                   // such a graph finished in the first round
        for (int e = lane; e < nnz; e += 64) L.cap_e[e] = work_e[(size_t)b * nnzc + e];
        __syncthreads();
        dense = capws + (size_t)b * V * V;
    } else {
        beam_fill_edge_caps(L, cap_b, V);
    }
    const int mono = nnz >= 0 ? beam_load_edge_scores(L, y_e + ((size_t)b * C + c) * nnzc, nnz) : 0;
    if (L.wcap) {
        for (int t = lane; t < V * V; t += 64) L.wcap[t] = dense[t];
        __syncthreads();
    }
    const long long* cm = comm + ((size_t)b * C + c) * 3;
    const int len = beam_commodity(L, L.wcap ? L.wcap : dense, L.cap_e, y + (size_t)b * V * V * C + c, V, C, K, (int)cm[0], (int)cm[1],
                                   (float)cm[2], 1, mono);
    for (int t = lane; t < len; t += 64) spec_paths[((size_t)b * C + c) * PL + t] = (int)L.best_path[t];
    if (lane == 0) spec_len[(size_t)b * C + c] = len;
}

// y from the network output, BeamSearchAlgorithm.__init__ (:14-21).
// five_d != 0: logits [rows][2] -> y[row] = log_softmax(.)[1]; thread per row.
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
