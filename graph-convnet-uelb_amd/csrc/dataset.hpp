// Expansion of a packed mini-batch into the dense tensors the model consumes (gfx950).
//
// The reference builds these arrays on the host for every batch and copies them to the device one by one
// (src/common/data_management/dataset_reader.py:117-145, src/gcn/training/base_strategy.py:89-102); the largest,
// the int64 edge labels [B,V,V,C], is a pure function of the per-commodity node visiting order [B,C,V].
// Here only the compact form crosses the bus and the dense tensors are written by the GPU:
//   x_nodes[b,v,c]   = 2 if v == dst_c, else 1 if v == src_c, else 0                     (:117-123)
//   y_edges[b,u,v,c] = 1 iff u and v have positive order and v follows u in the order-sorted sequence of
//                      commodity c (ties: lower node first)                               (:132-145)
//   x_edges = adjacency as int64, commodities as int64, demand = commodities[..., 2] as fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// succ[b,c,u] = successor of u on commodity c's path, -1 if none.  One thread per (b, c, u).
__global__ __launch_bounds__(256) void k_unpack_succ(int n, int V, const short* __restrict__ flow, short* __restrict__ succ) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int u = t % V;
    const short* ord = flow + (size_t)(t / V) * V;
    const int o = ord[u];
    int best = -1, best_o = 0;
    if (o > 0) {
        for (int v = 0; v < V; ++v) {
            const int ov = ord[v];
            const bool after = ov > o || (ov == o && v > u);
            if (after && (best < 0 || ov < best_o)) { best = v; best_o = ov; }     // ascending v: first of equal orders wins
        }
    }
    succ[t] = (short)best;
}

struct UnpackArgs {
    int B, V, C;
    const unsigned char* edges_u8;   // [B,V,V]
    const int* comm_i32;             // [B,C,3]
    const short* succ;               // [B,C,V]
    long long* x_edges;              // [B,V,V]
    long long* x_nodes;              // [B,V,C]
    long long* y_edges;              // [B,V,V,C]
    long long* comm;                 // [B,C,3]
    float* demand;                   // [B,C]
};

__global__ __launch_bounds__(256) void k_unpack_fill(UnpackArgs a) {
    const size_t V = a.V, C = a.C;
    const size_t ny = (size_t)a.B * V * V * C, ne = (size_t)a.B * V * V, nn = (size_t)a.B * V * C, nc = (size_t)a.B * C * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ny; i += (size_t)gridDim.x * 256) {
        const size_t c = i % C, v = (i / C) % V, u = (i / (C * V)) % V, b = i / (C * V * V);
        a.y_edges[i] = a.succ[(b * C + c) * V + u] == (short)v ? 1 : 0;
        if (i < ne) a.x_edges[i] = a.edges_u8[i];
        if (i < nn) {
            const size_t cc = i % C, vv = (i / C) % V, bb = i / (C * V);
            const int* cm = a.comm_i32 + (bb * C + cc) * 3;
            a.x_nodes[i] = (int)vv == cm[1] ? 2 : ((int)vv == cm[0] ? 1 : 0);
        }
        if (i < nc) {
            a.comm[i] = a.comm_i32[i];
            if (i % 3 == 2) a.demand[i / 3] = (float)a.comm_i32[i];
        }
    }
}
