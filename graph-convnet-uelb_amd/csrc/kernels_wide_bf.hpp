// Edge kernels for hidden sizes above 64 with the H x H products on the bf16 matrix pipe at fp32 accuracy.
//
// An fp32 value x is cut into three bf16 pieces x = x1 + x2 + x3, each piece the round-to-nearest bf16 of what the
// pieces before it left over.  The cut is exact (8 + 8 + 8 significant bits), so
//     a w = a1 w1 + (a1 w2 + a2 w1) + (a2 w2 + a1 w3 + a3 w1) + (a2 w3 + a3 w2 + a3 w3),
// and the last group is below 2^-26 |a w|: it is dropped.  Every bf16 x bf16 product is exact in fp32 and the sums
// are accumulated in fp32 by v_mfma_f32_16x16x32_bf16, so a product formed this way is as accurate as an fp32 FMA
// chain (measured against a float64 product: 4.3e-7 of the largest output against 5.1e-7 for the fp32 MFMA loop at
// K = 300).  Six bf16 MFMAs of 16 x 16 x 32 take 96 cycles of the matrix pipe where the fp32 form
// (v_mfma_f32_16x16x4_f32) needs 8 x 32 = 256 for the same block.
//
// Layout.  Activations: a step's rows in LDS as three planes of bf16 (row stride 640 B; the 16-byte chunk c of row r
// sits at chunk c ^ 2 ((r >> 1) & 3), which makes the 128-bit operand reads of a wavefront conflict-free).
// Weights: cut once per pass by k_split_weights into fragment order, plane p, column block nb, k block kb, lane:
// lane (rs, q) finds W[16 nb + rs][32 kb + 8 q .. + 7] as one 16-byte load, a wavefront reads 1 KiB contiguous.
#pragma once
#include "kernels_wide.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SROWB = GCN_WIDE_MAX_H * 2;         // bytes per LDS row of one plane (320 bf16)
constexpr int SNS = 4;                            // 16-row tiles per step of the edge kernels
constexpr int SPB = 16 * SNS * SROWB;             // bytes per LDS plane
constexpr int STH = 512;                          // threads per workgroup of the edge kernels: 8 wavefronts, one workgroup per CU
constexpr int SKA = (GCN_WIDE_MAX_H / 4 + 31) / 32;   // float4 column groups per staging thread
constexpr int SKB8 = (WNB + 7) / 8;               // column blocks per wavefront

// Thread roles in the edge kernels.  Staging: thread (arow, g0) = (t / 32, t % 32) walks the float4 column groups
// g0, g0 + 32, ... of row arow of every tile of the step.  Product: wavefront w = t / 64 takes the column blocks
// w, w + 8, ... against all tiles of the step, lane (rs, q) as everywhere.
struct SplitIdx {
    int t, lane, rs, q, w, arow, g0;
};
__device__ __forceinline__ SplitIdx split_idx() {
    SplitIdx x;
    x.t = threadIdx.x; x.lane = x.t & 63; x.rs = x.lane & 15; x.q = x.lane >> 4; x.w = x.t >> 6;
    x.arow = x.t >> 5; x.g0 = x.t & 31;
    return x;
}
// x[0] <- x[1] <- ... <- x[SKB8-1] <- x[0]: after SKB8 calls every entry is back in place
__device__ __forceinline__ void rot8(f32x4 (&x)[SKB8]) {
    const f32x4 t = x[0];
#pragma unroll
    for (int k = 0; k + 1 < SKB8; ++k) x[k] = x[k + 1];
    x[SKB8 - 1] = t;
}

__host__ __device__ inline int split_nkb(int H) { return (H + 31) / 32; }
// bytes of the fragment-ordered planes of one H x H matrix
__host__ __device__ inline size_t split_weight_bytes(int H) { return (size_t)3 * ((H + 15) / 16) * split_nkb(H) * 64 * 16; }

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// two fp32 values -> three packed bf16 pairs (low half = first value)
struct Split3 { uint32_t p[3]; };
__device__ __forceinline__ Split3 split2(float x, float y) {
    Split3 r;
    r.p[0] = cvt_pk_bf16(x, y);
    x -= __builtin_bit_cast(float, r.p[0] << 16); y -= __builtin_bit_cast(float, r.p[0] & 0xffff0000u);
    r.p[1] = cvt_pk_bf16(x, y);
    x -= __builtin_bit_cast(float, r.p[1] << 16); y -= __builtin_bit_cast(float, r.p[1] & 0xffff0000u);
    r.p[2] = cvt_pk_bf16(x, y);
    return r;
}

// Up to 64 matrices per launch: dst[m] = fragment-ordered planes of src[m] (transposed != 0: of src[m]^T, i.e. the
// operand of an "X * W" product).  grid (ceil(NB * NKB / 4), matrices), 256 threads: one fragment per wavefront.
struct SplitList {
    const float* src[64];
    u32x4* dst[64];
    int transposed;
};
__global__ __launch_bounds__(256) void k_split_weights(SplitList sl, int H) {
    const float* __restrict__ W = sl.src[blockIdx.y];
    u32x4* __restrict__ Wt = sl.dst[blockIdx.y];
    const int NB = (H + 15) / 16, NKB = split_nkb(H);
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4;
    const int frag = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frag >= NB * NKB) return;
    const int nb = frag / NKB, kb = frag - nb * NKB;
    const int n = 16 * nb + rs;
    u32x4 o[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 32 * kb + 8 * q + 2 * e;
        float x = 0.f, y = 0.f;
        if (n < H) {
            if (k < H) x = sl.transposed ? W[(size_t)k * H + n] : W[(size_t)n * H + k];
            if (k + 1 < H) y = sl.transposed ? W[(size_t)(k + 1) * H + n] : W[(size_t)n * H + k + 1];
        }
        const Split3 sp = split2(x, y);
        o[0][e] = sp.p[0]; o[1][e] = sp.p[1]; o[2][e] = sp.p[2];
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) Wt[((size_t)p * NB * NKB + frag) * 64 + lane] = o[p];
}

// byte offset of the 16-byte chunk (row, chunk) inside a plane
__device__ __forceinline__ int split_off(int row, int chunk) { return row * SROWB + ((chunk ^ (2 * ((row >> 1) & 3))) << 4); }
// columns 4 g .. 4 g + 3 of a row: cut and stored to the three planes (8 bytes each)
__device__ __forceinline__ void split_put4(unsigned char* planes, int row, int g, const f32x4 v) {
    const Split3 a = split2(v[0], v[1]), b = split2(v[2], v[3]);
    unsigned char* p = planes + split_off(row, g >> 1) + 8 * (g & 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2*>(p + k * SPB) = uint2{a.p[k], b.p[k]};
}
// the same four columns read back: the three pieces add up to the stored fp32 value exactly
__device__ __forceinline__ f32x4 split_get4(const unsigned char* planes, int row, int g) {
    const unsigned char* p = planes + split_off(row, g >> 1) + 8 * (g & 1);
    uint2 u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] = *reinterpret_cast<const uint2*>(p + k * SPB);
    f32x4 v;
    auto lo = [](uint32_t w) { return __builtin_bit_cast(float, w << 16); };
    auto hi = [](uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); };
    v[0] = (lo(u[0].x) + lo(u[1].x)) + lo(u[2].x);
    v[1] = (hi(u[0].x) + hi(u[1].x)) + hi(u[2].x);
    v[2] = (lo(u[0].y) + lo(u[1].y)) + lo(u[2].y);
    v[3] = (hi(u[0].y) + hi(u[1].y)) + hi(u[2].y);
    return v;
}
__device__ __forceinline__ void split_zero_planes(unsigned char* planes, int t) {
    for (int e = t * 16; e < 3 * SPB; e += STH * 16) *reinterpret_cast<u32x4*>(planes + e) = u32x4{0u, 0u, 0u, 0u};
}

// Output columns 16 nb + 4 q .. + 3 of rows 16 s + rs (s < NS) of planes * W^T.  The weight fragments of k block
// kb + 1 are requested before the MFMAs of k block kb; the LDS operands of a tile (three 128-bit reads) are requested
// while the six MFMAs of the tile before it run (two register sets of three); the small products are added first.
template <int NS>
__device__ __forceinline__ void split_gemm_ns(const unsigned char* planes, const u32x4* __restrict__ Wt, int NB, int NKB, int nb,
                                              int lane, int rs, int q, f32x4 (&o)[SNS]) {
    f32x4 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = f4zero();
    const size_t pstride = (size_t)NB * NKB * 64;
    const u32x4* wb = Wt + (size_t)nb * NKB * 64 + lane;
    u32x4 w[3], wn[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) w[p] = wb[p * pstride];
    auto read_tile = [&](u32x4 (&a)[3], int s, int kb) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            a[p] = *reinterpret_cast<const u32x4*>(planes + p * SPB + split_off(16 * s + rs, 4 * min(kb, NKB - 1) + q));
    };
    u32x4 a[2][3];
    read_tile(a[0], 0, 0);
    // two k blocks per trip so that the register set a tile sits in is known at compile time
    for (int kb0 = 0; kb0 < NKB; kb0 += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int kb = kb0 + d;
            if (kb < NKB) {
                const int kn = min(kb + 1, NKB - 1);
#pragma unroll
                for (int p = 0; p < 3; ++p) wn[p] = wb[p * pstride + (size_t)kn * 64];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int cur = (NS * d + s) & 1;
                    if (s + 1 < NS) read_tile(a[cur ^ 1], s + 1, kb);
                    else read_tile(a[cur ^ 1], 0, kb + 1);
                    __builtin_amdgcn_sched_barrier(0);
#define GCN_SPLIT_MM(pw, pa) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[pw]), __builtin_bit_cast(bf16x8, a[cur][pa]), acc[s], 0, 0, 0)
                    GCN_SPLIT_MM(2, 0); GCN_SPLIT_MM(0, 2); GCN_SPLIT_MM(1, 1); GCN_SPLIT_MM(1, 0); GCN_SPLIT_MM(0, 1); GCN_SPLIT_MM(0, 0);
#undef GCN_SPLIT_MM
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) w[p] = wn[p];
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] = acc[s];
}
// ns (wavefront-uniform) = tiles in use, 1 .. SNS; o[s] for s >= ns is left untouched
__device__ __forceinline__ void split_gemm_n(const unsigned char* planes, int ns, const u32x4* __restrict__ Wt, int NB, int NKB, int nb,
                                             int lane, int rs, int q, f32x4 (&o)[SNS]) {
    if (ns >= 4) split_gemm_ns<4>(planes, Wt, NB, NKB, nb, lane, rs, q, o);
    else if (ns == 3) split_gemm_ns<3>(planes, Wt, NB, NKB, nb, lane, rs, q, o);
    else if (ns == 2) split_gemm_ns<2>(planes, Wt, NB, NKB, nb, lane, rs, q, o);
    else split_gemm_ns<1>(planes, Wt, NB, NKB, nb, lane, rs, q, o);
}

// ------------------------------------------------------------------------------------------------
// Edge forward (EdgeFwdArgs, a.WUs = planes of edge_feat.U.weight).  A workgroup takes a contiguous range of (b, i, c)
// units; the rows j of its units, one unit after the other, form one sequence that is walked 64 rows (SNS tiles) per
// step, so that every weight fragment fetched from L2 feeds 24 MFMAs (at 32 rows per step the product phase ran at the
// rate the L2s deliver the weight planes, 18 TB/s over the chip).  From V = 64 on a step runs on across the end of a unit
// (it then holds rows of two units: no ragged last tile per unit, V = 100 takes 1.56 steps per unit instead of 2);
// below that a step ends with its unit.  One 768-thread workgroup per CU, two roles:
//   wavefronts 0..7 (two per SIMD, which hide each other's waits for weight fragments and LDS operands) multiply:
//     column blocks w, w + 8, ... of the step's rows, epilogue, sums;
//   wavefronts 8..11 (one per SIMD) stage: while step n is multiplied they fetch the rows of step n + 1 from memory,
//     form the layer input (fused BatchNorm apply / embedding), write it out and keep it in registers (80 per lane);
//     once step n's product is done they cut the values into the bf16 planes in LDS.
// A wavefront's vector-memory operations retire in issue order, so a wavefront that waits for weight fragments from
// L2 every few hundred cycles cannot also have rows from HBM in flight; split by role, the HBM stream runs under the
// whole product phase (with every wavefront doing both in turn, the chip alternated between an HBM-bound phase with
// the matrix pipe idle and a product phase with HBM idle: 39 % + 42 % of the kernel).
// partial layout [gridDim.x][4][H].  Dynamic LDS: 3 SPB bytes.
// ------------------------------------------------------------------------------------------------
#ifndef GCN_FMW
#define GCN_FMW 8      // measurement builds may set 4 (one multiplying wavefront per SIMD, 256 registers, no scratch: 201 ms per step against 184-187 at H = 300)
#endif
constexpr int FMW = GCN_FMW;                 // multiplying wavefronts of the forward kernel
constexpr int FMT = 64 * FMW;                // their threads
constexpr int FKB = (WNB + FMW - 1) / FMW;   // column blocks per multiplying wavefront
constexpr int SFT = FMT + 256;               // threads of the forward kernel: multiplying + 256 staging
// A step of the row sequence: it starts at row j0 of unit u and holds n rows (rows past the end of unit u belong to
// unit u + 1: at most one boundary per step).
struct FwdStep {
    int u, j0, n;
};
__device__ __forceinline__ FwdStep fwd_step_at(int u, int j0, int u1, int V, bool span) {
    FwdStep st;
    st.u = u; st.j0 = j0;
    const long long avail = span ? (long long)(u1 - u) * V - j0 : (long long)(V - j0);
    st.n = (int)(avail < 16 * SNS ? avail : 16 * SNS);
    return st;
}
__device__ __forceinline__ void fwd_step_next(int& u, int& j0, int n, int V) {
    j0 += n;
    if (j0 >= V) { j0 -= V; ++u; }
}
__global__ __launch_bounds__(SFT) void k_edge_fwd_ws(EdgeFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char planes[];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GCN_WIDE_MAX_H];
    __shared__ __attribute__((aligned(16))) float pxs[2][GCN_WIDE_MAX_H];      // sums of x_tmp, x_tmp^2 over this workgroup's units
    __shared__ __attribute__((aligned(16))) float ess[2][GCN_WIDE_MAX_H];      // sums of e_tmp, e_tmp^2 over this workgroup's rows
    __shared__ __attribute__((aligned(16))) float agg[2][2][GCN_WIDE_MAX_H];   // per unit in progress (slot u & 1): sum_j g Vn_j, sum_j g
    const int H = a.H, V = a.V, C = a.C, NB = (H + 15) / 16, HP = 16 * NB, NKB = split_nkb(H);
    const EdgeSrc& S = a.src;
    clds_fill(cl[CL_PMEAN], S.mode == 1 ? S.pmean : nullptr, H, HP); clds_fill(cl[CL_PRSTD], S.mode == 1 ? S.prstd : nullptr, H, HP);
    clds_fill(cl[CL_PGAMMA], S.mode == 1 ? S.pgamma : nullptr, H, HP); clds_fill(cl[CL_PBETA], S.mode == 1 ? S.pbeta : nullptr, H, HP);
    clds_fill(cl[CL_WE], S.mode == 2 ? S.we : nullptr, H, HP);
    for (int k = threadIdx.x; k < 2 * GCN_WIDE_MAX_H; k += SFT) { (&pxs[0][0])[k] = 0.f; (&ess[0][0])[k] = 0.f; }
    for (int k = threadIdx.x; k < 4 * GCN_WIDE_MAX_H; k += SFT) (&agg[0][0][0])[k] = 0.f;
    for (int e = threadIdx.x * 16; e < 3 * SPB; e += SFT * 16) *reinterpret_cast<u32x4*>(planes + e) = u32x4{0u, 0u, 0u, 0u};
    const int units = a.B * V * C;
    const int per = (units + (int)gridDim.x - 1) / (int)gridDim.x;
    const int u0 = (int)blockIdx.x * per, u1 = min(units, u0 + per);
    const bool span = V >= 16 * SNS;
    __syncthreads();
    if (threadIdx.x >= FMT) {
        // ---------------------------------------------------------------- staging wavefronts
        const int pt = threadIdx.x - FMT, arow = pt >> 4, g0 = pt & 15;
        const DropCfg dc = drop_cfg(a.rng, S.drop_sid, S.drop_p, S.drop_on && S.mode == 1);
        f32x4 val[SNS][WKA];
        // rows 16 s + arow of a step: layer input -> val (zeros outside the step), written to e_out
        auto produce = [&](const FwdStep& st) {
#pragma unroll
            for (int s = 0; s < SNS; ++s) {
#pragma unroll
                for (int k = 0; k < WKA; ++k) val[s][k] = f4zero();
                if (16 * s >= st.n) continue;
                const int off = 16 * s + arow;
                const bool jok = off < st.n;
                int j = st.j0 + min(off, st.n - 1), u = st.u;
                if (j >= V) { j -= V; ++u; }
                const int bi = u / C, c = u - bi * C;
                const size_t erow = ((size_t)bi * V + j) * C + c;
                if (S.mode == 1) {
                    f32x4 tp[WKA], ti[WKA];
#pragma unroll
                    for (int k = 0; k < WKA; ++k) {
                        const int acol = 4 * (g0 + 16 * k);
                        const size_t o = (jok && acol < H) ? erow * H + acol : 0;
                        tp[k] = ld4(S.prev_tmp + o); ti[k] = ld4(S.prev_in + o);
                    }
                    const uint32_t rh = dc.on ? drop_rowhash(dc, (uint32_t)erow) : 0u;
#pragma unroll
                    for (int k = 0; k < WKA; ++k) {
                        const int g = g0 + 16 * k, acol = 4 * g;
                        if (jok && acol < H) {
                            f32x4 v = bn_relu_res4(tp[k], ti[k], &cl[CL_PMEAN][acol], &cl[CL_PRSTD][acol], &cl[CL_PGAMMA][acol], &cl[CL_PBETA][acol]);
                            if (dc.on) v *= drop_mul4(dc, rh, g);
                            if (S.e_out != nullptr) st4(S.e_out + erow * H + acol, v);
                            val[s][k] = v;
                        }
                    }
                } else {
                    const float capv = (S.mode == 2 && jok) ? S.cap[(size_t)bi * V + j] : 0.f;
#pragma unroll
                    for (int k = 0; k < WKA; ++k) {
                        const int acol = 4 * (g0 + 16 * k);
                        if (jok && acol < H) {
                            const size_t o = erow * H + acol;
                            f32x4 v;
                            if (S.mode == 0) v = ld4(S.e_in + o);
                            else {
                                v = ld4(&cl[CL_WE][acol]) * capv;
                                if (S.e_out != nullptr) st4(S.e_out + o, v);
                            }
                            val[s][k] = v;
                        }
                    }
                }
            }
        };
        int u = u0, j0 = 0;
        if (u < u1) produce(fwd_step_at(u, j0, u1, V, span));
        ACCF_DECL;
        while (u < u1) {
            const FwdStep st = fwd_step_at(u, j0, u1, V, span);
#pragma unroll
            for (int s = 0; s < SNS; ++s) {
                if (16 * s >= st.n) break;
#pragma unroll
                for (int k = 0; k < WKA; ++k) {
                    const int g = g0 + 16 * k;
                    if (4 * g < HP) split_put4(planes, 16 * s + arow, g, val[s][k]);
                }
            }
            ACCF(0);
            __syncthreads();                    // the planes of this step are complete
            ACCF(1);
            fwd_step_next(u, j0, st.n, V);
            if (u < u1) produce(fwd_step_at(u, j0, u1, V, span));
            ACCF(2);
            __syncthreads();                    // the product of this step is done: the planes may be overwritten
            ACCF(5);
        }
        ACCF_END(threadIdx.x >> 6, threadIdx.x & 63);
        return;
    }
    // -------------------------------------------------------------------- multiplying wavefronts
    const SplitIdx ix = split_idx();
    ACCF_DECL;
    int u = u0, j0 = 0;
    while (u < u1) {
        const FwdStep st = fwd_step_at(u, j0, u1, V, span);
        const int ns = (st.n + 15) >> 4;
        // this lane's rows of the step: row j (of unit u, or of unit u + 1 past the end of unit u)
        int jr[SNS];
        bool rok[SNS], inb[SNS];
#pragma unroll
        for (int s = 0; s < SNS; ++s) {
            const int off = 16 * s + ix.rs;
            rok[s] = off < st.n;
            int j = st.j0 + min(off, st.n - 1);
            inb[s] = j >= V;
            jr[s] = inb[s] ? j - V : j;
        }
        const bool two = st.j0 + st.n > V;              // the step holds rows of unit u + 1 as well
        const int biA = u / C, cA = u - biA * C, bA = biA / V;
        const int uB = two ? u + 1 : u;
        const int biB = uB / C, cB = uB - biB * C, bB = biB / V;
        ACCF(0);
        __syncthreads();
        ACCF(1);
#pragma unroll 1
        for (int kk = 0; kk < FKB; ++kk) {
            const int nb = ix.w + FMW * kk;
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                // epilogue operands are requested before the product so that their latency hides under it
                const bool cok = bcol < H;
                f32x4 vej[SNS], vnj2[SNS];
                const f32x4 bu = ld4(a.bU + (cok ? bcol : 0));
                const f32x4 veiA = ld4(a.Ve + (cok ? (size_t)u * H + bcol : 0)) + bu;
                const f32x4 veiB = ld4(a.Ve + (cok ? (size_t)uB * H + bcol : 0)) + bu;
#pragma unroll
                for (int s = 0; s < SNS; ++s) {
                    const size_t nrow = (size_t)((inb[s] ? bB : bA) * V + jr[s]) * C + (inb[s] ? cB : cA);
                    const size_t o = (cok && rok[s]) ? nrow * H + bcol : 0;
                    vej[s] = ld4(a.Ve + o); vnj2[s] = ld4(a.Vn + o);
                }
                f32x4 et2[SNS];
                split_gemm_n(planes, ns, a.WUs, NB, NKB, nb, ix.lane, ix.rs, ix.q, et2);
                ACCF(2);
                f32x4 t1 = f4zero(), t2 = f4zero(), nmA = f4zero(), dnA = f4zero(), nmB = f4zero(), dnB = f4zero();
                if (cok) {
#pragma unroll
                    for (int s = 0; s < SNS; ++s) {
                        if (s < ns && rok[s]) {
                            const size_t erow = ((size_t)(inb[s] ? biB : biA) * V + jr[s]) * C + (inb[s] ? cB : cA);
                            const f32x4 et = et2[s] + (inb[s] ? veiB : veiA) + vej[s];
                            const f32x4 vnj = vnj2[s];
                            st4(a.e_tmp + erow * H + bcol, et);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float g = sigmoidf_(et[r]);
                                const float gv = g * vnj[r];
                                if (inb[s]) { nmB[r] += gv; dnB[r] += g; }
                                else { nmA[r] += gv; dnA[r] += g; }
                                t1[r] += et[r];
                                t2[r] += et[r] * et[r];
                            }
                        }
                    }
                }
                // sums over the step's rows, added by the lane that owns the four columns (same lane, same
                // address every time: fixed order, no atomics)
                wide_acc_cols(ess, bcol, ix.rs == 0 && cok, t1, t2);
                wide_acc_cols(agg[u & 1], bcol, ix.rs == 0 && cok, nmA, dnA);
                if (two) wide_acc_cols(agg[uB & 1], bcol, ix.rs == 0 && cok, nmB, dnB);
                ACCF(3);
            }
        }
        // ---- unit u ends with this step: the lane that accumulated these columns finishes it
        if (st.j0 + st.n >= V) {
#pragma unroll
            for (int k = 0; k < FKB; ++k) {
                const int nb = ix.w + FMW * k;
                if (nb < NB) {
                    const int bcol = 16 * nb + 4 * ix.q;
                    if (ix.rs == 0 && bcol < H) {
                        float (*ag)[GCN_WIDE_MAX_H] = agg[u & 1];
                        const f32x4 nm = ld4(&ag[0][bcol]), dn = ld4(&ag[1][bcol]);
                        st4(&ag[0][bcol], f4zero()); st4(&ag[1][bcol], f4zero());
                        f32x4 D = f32x4{1.f, 1.f, 1.f, 1.f}, A = nm;
                        if (a.mean_agg) {
                            D = dn + 1e-20f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) A[r] = nm[r] * __builtin_amdgcn_rcpf(D[r]);
                        }
                        const f32x4 xt = ld4(a.Ux + (size_t)u * H + bcol) + A;
                        st4(a.x_tmp + (size_t)u * H + bcol, xt);
                        st4(a.Aout + (size_t)u * H + bcol, A);
                        st4(a.Dout + (size_t)u * H + bcol, D);
                        // this lane is the only one that ever touches these four columns of pxs
                        st4(&pxs[0][bcol], ld4(&pxs[0][bcol]) + xt);
                        st4(&pxs[1][bcol], ld4(&pxs[1][bcol]) + xt * xt);
                    }
                }
            }
        }
        ACCF(4);
        __syncthreads();
        ACCF(5);
        fwd_step_next(u, j0, st.n, V);
    }
#pragma unroll
    for (int k = 0; k < FKB; ++k) {
        const int nb = ix.w + FMW * k;
        if (nb < NB) {
            const int bcol = 16 * nb + 4 * ix.q;
            if (ix.rs == 0 && bcol < H) {       // the lane that accumulated these columns
                float* p = a.partial + (size_t)blockIdx.x * 4 * H;
                st4(p + bcol, ld4(&ess[0][bcol])); st4(p + H + bcol, ld4(&ess[1][bcol]));
                st4(p + 2 * H + bcol, ld4(&pxs[0][bcol])); st4(p + 3 * H + bcol, ld4(&pxs[1][bcol]));
            }
        }
    }
    ACCF(6);
    ACCF_END(ix.w, ix.lane);
}

// ------------------------------------------------------------------------------------------------
// Edge backward (EdgeBwdArgs, a.WUs = planes of edge_feat.U.weight^T).  Contiguous ranges of (b, it, jt, c) units, c
// fastest; a unit walks its rows i SNS at a time (LDS rows 16 s .. 16 s + 15: the 16 columns j of row i + s), so the
// column partial sums of a staging thread cover all of them.  The residual-branch term dpre = de * dropout mask is
// re-read in the product phase by the thread that overwrites it.  Dynamic LDS: 3 SPB bytes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(STH) void k_edge_bwd_ws(EdgeBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char planes[];
    __shared__ __attribute__((aligned(16))) float cc[6][GCN_WIDE_MAX_H];    // constants of the edge BatchNorm backward
    __shared__ __attribute__((aligned(16))) float qa[SNS][2][GCN_WIDE_MAX_H]; // per row of the step: qi, aggregate A
    __shared__ __attribute__((aligned(16))) float nss[2][GCN_WIDE_MAX_H];     // BN-backward sums of the layer below over this workgroup's rows
    const SplitIdx ix = split_idx();
    const int H = a.H, V = a.V, C = a.C, IT = a.IT, NB = (H + 15) / 16, HP = 16 * NB, NKB = split_nkb(H);
    const bool do_next = a.next.tmp != nullptr;
    const bool mean_agg = a.mean_agg != 0;
    clds_fill(cc[0], a.stats, H, HP); clds_fill(cc[1], a.stats + H, H, HP);
    clds_fill(cc[2], a.gamma_e, H, HP); clds_fill(cc[3], a.beta_e, H, HP);
    clds_fill(cc[4], a.bstats_e, H, HP); clds_fill(cc[5], a.bstats_e + H, H, HP);
    const DropCfg dce = drop_cfg(a.rng, a.sid_e, a.p, a.training);
    const DropCfg dcx = drop_cfg(a.rng, a.sid_x, a.p, a.training);
    const DropCfg ndc = drop_cfg(a.rng, a.next.drop_sid, a.p, a.training && do_next);
    split_zero_planes(planes, ix.t);
    for (int k = ix.t; k < 2 * GCN_WIDE_MAX_H; k += STH) (&nss[0][0])[k] = 0.f;
    const int njt = a.jt_n;                            // column tiles jt_lo .. jt_lo + jt_n - 1 (the ragged last one may go to k_edge_bwd_wp)
    const int nit = (V + IT - 1) / IT;
    const int units = a.B * nit * njt * C;
    const size_t Rn = (size_t)a.B * V * C;
    // balanced ranges: every workgroup gets floor or ceil of units / gridDim.x
    const int u0 = (int)((long long)blockIdx.x * units / (int)gridDim.x), u1 = (int)((long long)(blockIdx.x + 1) * units / (int)gridDim.x);
    __syncthreads();
    ACCB_DECL;
    for (int u = u0; u < u1; ++u) {
        int tt = u;
        const int c = tt % C; tt /= C;
        const int jt = a.jt_lo + tt % njt; tt /= njt;
        const int it = tt % nit;
        const int b = tt / nit;
        const int ja = jt * 16 + ix.arow;
        const bool jaok = ja < V;
        const size_t nrow_ja = (size_t)(b * V + min(ja, V - 1)) * C + c;
        f32x4 cve[SKA], cvn[SKA];
#pragma unroll
        for (int k = 0; k < SKA; ++k) { cve[k] = f4zero(); cvn[k] = f4zero(); }
        const int i_end = min(V, (it + 1) * IT);
        for (int i = it * IT; i < i_end; i += SNS) {
            const int ns = min(SNS, i_end - i);
            // Requested before anything else of the step: the Vn row of column j (it serves every row i of the step) and
            // the two streams of the first row; the streams of the later rows follow when their turn comes.
            f32x4 vnv[SKA], dnv[SKA], tpv[SKA];
            auto load_streams = [&](int s) {
                const size_t erow = ((size_t)(b * V + i + s) * V + min(ja, V - 1)) * C + c;
#pragma unroll
                for (int k = 0; k < SKA; ++k) {
                    const int acol = 4 * (ix.g0 + 32 * k);
                    const size_t off = (jaok && acol < H) ? erow * H + acol : 0;
                    dnv[k] = ld4(a.de + off); tpv[k] = ld4(a.e_tmp + off);
                }
            };
#pragma unroll
            for (int k = 0; k < SKA; ++k) {
                const int acol = 4 * (ix.g0 + 32 * k);
                vnv[k] = ld4(a.Vn + ((jaok && acol < H) ? nrow_ja * H + acol : 0));
            }
            load_streams(0);
            // ---- node side of the step, once per (row i, column): qi = d x_tmp / D and the aggregate A of row i
            if (ix.t < SNS * (HP / 4)) {
                const int s = ix.t / (HP / 4), g = ix.t - s * (HP / 4), acol = 4 * g;
                f32x4 qv = f4zero(), ai = f4zero();
                if (s < ns && acol < H) {
                    const size_t nrow_i = (size_t)(b * V + i + s) * C + c;
                    const f32x4 dxn = ld4(a.dx + nrow_i * H + acol), xt = ld4(a.x_tmp + nrow_i * H + acol);
                    f32x4 Dv = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (mean_agg) { Dv = ld4(a.Dn + nrow_i * H + acol); ai = ld4(a.Aagg + nrow_i * H + acol); }
                    f32x4 dmx = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (dcx.on) dmx = drop_mul4(dcx, drop_rowhash(dcx, (uint32_t)nrow_i), g);
                    const f32x4 xmu = ld4(a.stats + 2 * H + acol), xrs = ld4(a.stats + 3 * H + acol), xga = ld4(a.gamma_x + acol);
                    const f32x4 xbe = ld4(a.beta_x + acol), xm1 = ld4(a.bstats_x + acol), xm2 = ld4(a.bstats_x + H + acol);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xhx = (xt[r] - xmu[r]) * xrs[r];
                        const float bnx = xhx * xga[r] + xbe[r];
                        const float dxd = dxn[r] * dmx[r];
                        const float dbnx = bnx > 0.f ? dxd : 0.f;
                        const float dxt = xga[r] * xrs[r] * (dbnx - xm1[r] - xhx * xm2[r]);
                        qv[r] = mean_agg ? dxt * __builtin_amdgcn_rcpf(Dv[r]) : dxt;
                    }
                }
                st4(&qa[s][0][acol], qv);
                st4(&qa[s][1][acol], ai);
            }
            ACCB(0);
            __syncthreads();
            ACCB(1);
            // ---- phase A
#pragma unroll
            for (int s = 0; s < SNS; ++s) {
                if (s >= ns) break;
                const size_t erow = ((size_t)(b * V + i + s) * V + min(ja, V - 1)) * C + c;
                const int lrow = 16 * s + ix.arow;
                if (s > 0) load_streams(s);
                const uint32_t rhe = dce.on ? drop_rowhash(dce, (uint32_t)erow) : 0u;
#pragma unroll
                for (int k = 0; k < SKA; ++k) {
                    const int g = ix.g0 + 32 * k;
                    const int acol = 4 * g;
                    if (acol < HP) {
                        f32x4 det = f4zero();
                        if (jaok && acol < H) {
                            const f32x4 qv = ld4(&qa[s][0][acol]), ai = ld4(&qa[s][1][acol]);
                            const f32x4 dn = dnv[k], tp = tpv[k];
                            const f32x4 vnj = vnv[k];
                            f32x4 dme = f32x4{1.f, 1.f, 1.f, 1.f};
                            if (dce.on) dme = drop_mul4(dce, rhe, g);
                            const f32x4 emu = ld4(&cc[0][acol]), ers = ld4(&cc[1][acol]), ega = ld4(&cc[2][acol]);
                            const f32x4 ebe = ld4(&cc[3][acol]), em1 = ld4(&cc[4][acol]), em2 = ld4(&cc[5][acol]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float qi = qv[r];
                                const float x = tp[r];
                                const float xh = (x - emu[r]) * ers[r];
                                const float bn = xh * ega[r] + ebe[r];
                                const float d = dn[r] * dme[r];
                                const float dbn = bn > 0.f ? d : 0.f;
                                const float gt = sigmoidf_(x);
                                const float v = ega[r] * ers[r] * (dbn - em1[r] - xh * em2[r]) + qi * (vnj[r] - ai[r]) * gt * (1.f - gt);
                                det[r] = v;
                                cve[k][r] += v;
                                cvn[k][r] += qi * gt;
                            }
                            st4(a.det_out + erow * H + acol, det);
                        }
                        split_put4(planes, lrow, g, det);
                    }
                }
            }
            ACCB(2);
            __syncthreads();
            ACCB(3);
            {   // ---- phase B.  One copy of the product loop: the per-block accumulators rotate through slot 0.
                const int jb = jt * 16 + ix.rs;
                const bool jbok = jb < V;
                const size_t erow_b0 = ((size_t)(b * V + i) * V + min(jb, V - 1)) * C + c;
                const size_t nrow_i0 = (size_t)(b * V + i) * C + c;
#pragma unroll 1
                for (int kk = 0; kk < SKB8; ++kk) {
                    const int nb = ix.w + 8 * kk;
                    if (nb < NB) {
                        const int bcol = 16 * nb + 4 * ix.q;
                        const bool valid_b = jbok && bcol < H;
                        // this thread's elements of de (read here, overwritten below) and of the next layer's pre-BN tensor
                        f32x4 dold[SNS], ntp[SNS];
#pragma unroll
                        for (int s = 0; s < SNS; ++s) {
                            const size_t off = (valid_b && s < ns) ? (erow_b0 + (size_t)s * V * C) * H + bcol : 0;
                            dold[s] = ld4(a.de + off);
                            ntp[s] = do_next ? ld4(a.next.tmp + off) : f4zero();
                        }
                        f32x4 nmu = f4zero(), nrs = f4zero(), nga = f4zero(), nbe = f4zero();
                        if (do_next) {
                            const int cb = bcol < H ? bcol : 0;
                            nmu = ld4(a.next.mean + cb); nrs = ld4(a.next.rstd + cb); nga = ld4(a.next.gamma + cb); nbe = ld4(a.next.beta + cb);
                        }
                        f32x4 o2[SNS];
                        split_gemm_n(planes, ns, a.WUs, NB, NKB, nb, ix.lane, ix.rs, ix.q, o2);
                        ACCB(4);
                        f32x4 t1 = f4zero(), t2 = f4zero();
#pragma unroll
                        for (int s = 0; s < SNS; ++s) {
                            if (s >= ns) break;
                            const size_t erow_b = erow_b0 + (size_t)s * V * C;
                            f32x4 rsum = split_get4(planes, 16 * s + ix.rs, bcol >> 2);
                            if (valid_b) {
                                f32x4 dpre = dold[s];
                                if (dce.on) dpre *= drop_mul4(dce, drop_rowhash(dce, (uint32_t)erow_b), bcol >> 2);
                                const f32x4 din = o2[s] + dpre;
                                st4(a.de + erow_b * H + bcol, din);
                                if (do_next) {
                                    f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                                    if (ndc.on) dm = drop_mul4(ndc, drop_rowhash(ndc, (uint32_t)erow_b), bcol >> 2);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float xh = (ntp[s][r] - nmu[r]) * nrs[r];
                                        const float bn = xh * nga[r] + nbe[r];
                                        const float d = bn > 0.f ? din[r] * dm[r] : 0.f;
                                        t1[r] += d;
                                        t2[r] += d * xh;
                                    }
                                }
                            }
                            rsum = red16x4(rsum);
                            if (ix.rs == 0 && bcol < H) st4(a.rowpart + ((size_t)jt * Rn + nrow_i0 + (size_t)s * C) * H + bcol, rsum);
                        }
                        if (do_next) wide_acc_cols(nss, bcol, ix.rs == 0 && bcol < H, t1, t2);
                        ACCB(5);
                    }
                }
            }
            ACCB(6);
            __syncthreads();
            ACCB(7);
        }
        if (jaok) {
#pragma unroll
            for (int k = 0; k < SKA; ++k) {
                const int acol = 4 * (ix.g0 + 32 * k);
                if (acol < H) {
                    st4(a.colVe + ((size_t)it * Rn + nrow_ja) * H + acol, cve[k]);
                    st4(a.colVn + ((size_t)it * Rn + nrow_ja) * H + acol, cvn[k]);
                }
            }
        }
    }
    if (do_next) {
#pragma unroll
        for (int k = 0; k < SKB8; ++k) {
            const int nb = ix.w + 8 * k;
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                if (ix.rs == 0 && bcol < H) {       // the lane that accumulated these columns
                    st4(a.next.partial + (size_t)(a.part_off + blockIdx.x) * 2 * H + bcol, ld4(&nss[0][bcol]));
                    st4(a.next.partial + (size_t)(a.part_off + blockIdx.x) * 2 * H + H + bcol, ld4(&nss[1][bcol]));
                }
            }
        }
    }
    ACCB_END(ix.w, ix.lane);
}

// ------------------------------------------------------------------------------------------------
// Edge backward of the ragged last column tile (EdgeBwdArgs with jt_lo = that tile, jt_n = 1; same math and outputs as
// k_edge_bwd_ws for its units (b, it, jt_lo, c)).  When the node count is no multiple of 16 and the last tile has at
// most 8 of its 16 columns j in use, a 16-row tile in LDS holds P rows i of 16 / P columns each (P = 2, or 4 with at
// most 4 columns) and a step takes 8 rows i (P = 2) or 10 (P = 4): at 100 nodes the seventh column tile has 4 columns, and walked like the
// others it costs a seventh of the kernel's time for 4 % of its rows (2.85 ms per launch against 2.48 at 96 nodes per
// million rows).  A kernel of its own so that the layout arithmetic does not cost the main kernel registers.
// Dynamic LDS: 3 SPB bytes.
// ------------------------------------------------------------------------------------------------
constexpr int BQR = 10;         // most rows i per step of a packed unit = rows of node-side terms kept in LDS (what fits beside the planes)
// sum over the 16 / P lanes of a tile that belong to the same row i (every lane receives its group's sum); P = 2, 4.
// Cross-lane moves through the LDS crossbar: packed tiles are the exception, and the order of the additions is fixed.
__device__ __forceinline__ f32x4 red_group4(f32x4 v, int P) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y = v[r];
        y += __shfl_xor(y, 1, 64);
        y += __shfl_xor(y, 2, 64);
        if (P == 2) y += __shfl_xor(y, 4, 64);
        v[r] = y;
    }
    return v;
}
__global__ __launch_bounds__(STH) void k_edge_bwd_wp(EdgeBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char planes[];
    __shared__ __attribute__((aligned(16))) float cc[6][GCN_WIDE_MAX_H];    // constants of the edge BatchNorm backward
    __shared__ __attribute__((aligned(16))) float qa[BQR][2][GCN_WIDE_MAX_H]; // per row i of the step: qi, aggregate A
    __shared__ __attribute__((aligned(16))) float nss[2][GCN_WIDE_MAX_H];     // BN-backward sums of the layer below over this workgroup's rows
    const SplitIdx ix = split_idx();
    const int H = a.H, V = a.V, C = a.C, IT = a.IT, NB = (H + 15) / 16, HP = 16 * NB, NKB = split_nkb(H);
    const bool do_next = a.next.tmp != nullptr;
    const bool mean_agg = a.mean_agg != 0;
    clds_fill(cc[0], a.stats, H, HP); clds_fill(cc[1], a.stats + H, H, HP);
    clds_fill(cc[2], a.gamma_e, H, HP); clds_fill(cc[3], a.beta_e, H, HP);
    clds_fill(cc[4], a.bstats_e, H, HP); clds_fill(cc[5], a.bstats_e + H, H, HP);
    const DropCfg dce = drop_cfg(a.rng, a.sid_e, a.p, a.training);
    const DropCfg dcx = drop_cfg(a.rng, a.sid_x, a.p, a.training);
    const DropCfg ndc = drop_cfg(a.rng, a.next.drop_sid, a.p, a.training && do_next);
    split_zero_planes(planes, ix.t);
    for (int k = ix.t; k < 2 * GCN_WIDE_MAX_H; k += STH) (&nss[0][0])[k] = 0.f;
    const int nit = (V + IT - 1) / IT;
    const size_t Rn = (size_t)a.B * V * C;
    // all units of this launch are of column tile jt_lo: balanced ranges
    const int units = a.B * nit * C;
    const int u0 = (int)((long long)blockIdx.x * units / (int)gridDim.x), u1 = (int)((long long)(blockIdx.x + 1) * units / (int)gridDim.x);
    __syncthreads();
    ACCB_DECL;
    const int jt = a.jt_lo;
    const int vj = min(16, V - 16 * jt);
    const int lg = vj <= 4 ? 2 : 3, JW = 1 << lg, P = 16 >> lg;                          // columns j per tile row group, rows i per tile
    const int srows = P == 4 ? BQR : 8;                                                     // rows i per step: at most SNS tiles of P rows
    for (int u = u0; u < u1; ++u) {
        int tt = u;
        const int c = tt % C; tt /= C;
        const int it = tt % nit;
        const int b = tt / nit;
        const int adi = ix.arow >> lg, adj = ix.arow & (JW - 1);                            // staging thread: row i and column j inside a tile
        const int ja = jt * 16 + adj;
        const bool jaok = ja < V;
        const size_t nrow_ja = (size_t)(b * V + min(ja, V - 1)) * C + c;
        f32x4 cve[SKA], cvn[SKA];
#pragma unroll
        for (int k = 0; k < SKA; ++k) { cve[k] = f4zero(); cvn[k] = f4zero(); }
        const int i_end = min(V, (it + 1) * IT);
        for (int i = it * IT; i < i_end; i += srows) {
            const int rows = min(srows, i_end - i);
            const int ns = (rows + P - 1) >> (4 - lg);                                      // tiles in use
            // Requested before anything else of the step: the Vn row of column j (it serves every row i of the step) and
            // the two streams of the first tile; the streams of the later tiles follow when their turn comes.
            f32x4 vnv[SKA], dnv[SKA], tpv[SKA];
            auto load_streams = [&](int s) {
                const int r = s * P + adi;
                const bool ok = jaok && r < rows;
                const size_t erow = ((size_t)(b * V + i + min(r, rows - 1)) * V + min(ja, V - 1)) * C + c;
#pragma unroll
                for (int k = 0; k < SKA; ++k) {
                    const int acol = 4 * (ix.g0 + 32 * k);
                    const size_t off = (ok && acol < H) ? erow * H + acol : 0;
                    dnv[k] = ld4(a.de + off); tpv[k] = ld4(a.e_tmp + off);
                }
            };
#pragma unroll
            for (int k = 0; k < SKA; ++k) {
                const int acol = 4 * (ix.g0 + 32 * k);
                vnv[k] = ld4(a.Vn + ((jaok && acol < H) ? nrow_ja * H + acol : 0));
            }
            load_streams(0);
            // ---- node side of the step, once per (row i, column): qi = d x_tmp / D and the aggregate A of row i
            for (int e = ix.t; e < srows * (HP / 4); e += STH) {
                const int r = e / (HP / 4), g = e - r * (HP / 4), acol = 4 * g;
                f32x4 qv = f4zero(), ai = f4zero();
                if (r < rows && acol < H) {
                    const size_t nrow_i = (size_t)(b * V + i + r) * C + c;
                    const f32x4 dxn = ld4(a.dx + nrow_i * H + acol), xt = ld4(a.x_tmp + nrow_i * H + acol);
                    f32x4 Dv = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (mean_agg) { Dv = ld4(a.Dn + nrow_i * H + acol); ai = ld4(a.Aagg + nrow_i * H + acol); }
                    f32x4 dmx = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (dcx.on) dmx = drop_mul4(dcx, drop_rowhash(dcx, (uint32_t)nrow_i), g);
                    const f32x4 xmu = ld4(a.stats + 2 * H + acol), xrs = ld4(a.stats + 3 * H + acol), xga = ld4(a.gamma_x + acol);
                    const f32x4 xbe = ld4(a.beta_x + acol), xm1 = ld4(a.bstats_x + acol), xm2 = ld4(a.bstats_x + H + acol);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float xhx = (xt[rr] - xmu[rr]) * xrs[rr];
                        const float bnx = xhx * xga[rr] + xbe[rr];
                        const float dxd = dxn[rr] * dmx[rr];
                        const float dbnx = bnx > 0.f ? dxd : 0.f;
                        const float dxt = xga[rr] * xrs[rr] * (dbnx - xm1[rr] - xhx * xm2[rr]);
                        qv[rr] = mean_agg ? dxt * __builtin_amdgcn_rcpf(Dv[rr]) : dxt;
                    }
                }
                st4(&qa[r][0][acol], qv);
                st4(&qa[r][1][acol], ai);
            }
            ACCB(0);
            __syncthreads();
            ACCB(1);
            // ---- phase A
#pragma unroll
            for (int s = 0; s < SNS; ++s) {
                if (s >= ns) break;
                const int r = s * P + adi;
                const bool rok = jaok && r < rows;
                const int rq = min(r, BQR - 1);
                const size_t erow = ((size_t)(b * V + i + min(r, rows - 1)) * V + min(ja, V - 1)) * C + c;
                const int lrow = 16 * s + ix.arow;
                if (s > 0) load_streams(s);
                const uint32_t rhe = dce.on ? drop_rowhash(dce, (uint32_t)erow) : 0u;
#pragma unroll
                for (int k = 0; k < SKA; ++k) {
                    const int g = ix.g0 + 32 * k;
                    const int acol = 4 * g;
                    if (acol < HP) {
                        f32x4 det = f4zero();
                        if (rok && acol < H) {
                            const f32x4 qv = ld4(&qa[rq][0][acol]), ai = ld4(&qa[rq][1][acol]);
                            const f32x4 dn = dnv[k], tp = tpv[k];
                            const f32x4 vnj = vnv[k];
                            f32x4 dme = f32x4{1.f, 1.f, 1.f, 1.f};
                            if (dce.on) dme = drop_mul4(dce, rhe, g);
                            const f32x4 emu = ld4(&cc[0][acol]), ers = ld4(&cc[1][acol]), ega = ld4(&cc[2][acol]);
                            const f32x4 ebe = ld4(&cc[3][acol]), em1 = ld4(&cc[4][acol]), em2 = ld4(&cc[5][acol]);
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const float qi = qv[rr];
                                const float x = tp[rr];
                                const float xh = (x - emu[rr]) * ers[rr];
                                const float bn = xh * ega[rr] + ebe[rr];
                                const float d = dn[rr] * dme[rr];
                                const float dbn = bn > 0.f ? d : 0.f;
                                const float gt = sigmoidf_(x);
                                const float v = ega[rr] * ers[rr] * (dbn - em1[rr] - xh * em2[rr]) + qi * (vnj[rr] - ai[rr]) * gt * (1.f - gt);
                                det[rr] = v;
                                cve[k][rr] += v;
                                cvn[k][rr] += qi * gt;
                            }
                            st4(a.det_out + erow * H + acol, det);
                        }
                        split_put4(planes, lrow, g, det);
                    }
                }
            }
            ACCB(2);
            __syncthreads();
            ACCB(3);
            {   // ---- phase B.  One copy of the product loop: the per-block accumulators rotate through slot 0.
                const int bdi = ix.rs >> lg, bdj = ix.rs & (JW - 1);                        // this lane's row i and column j inside a tile
                const int jb = jt * 16 + bdj;
                const bool jbok = jb < V;
                const size_t ecol_b = (size_t)min(jb, V - 1) * C + c;
#pragma unroll 1
                for (int kk = 0; kk < SKB8; ++kk) {
                    const int nb = ix.w + 8 * kk;
                    if (nb < NB) {
                        const int bcol = 16 * nb + 4 * ix.q;
                        const bool valid_b = jbok && bcol < H;
                        // this thread's elements of de (read here, overwritten below) and of the next layer's pre-BN tensor
                        f32x4 dold[SNS], ntp[SNS];
#pragma unroll
                        for (int s = 0; s < SNS; ++s) {
                            const int r = s * P + bdi;
                            const size_t off = (valid_b && r < rows) ? ((size_t)(b * V + i + r) * V * C + ecol_b) * H + bcol : 0;
                            dold[s] = ld4(a.de + off);
                            ntp[s] = do_next ? ld4(a.next.tmp + off) : f4zero();
                        }
                        f32x4 nmu = f4zero(), nrs = f4zero(), nga = f4zero(), nbe = f4zero();
                        if (do_next) {
                            const int cb = bcol < H ? bcol : 0;
                            nmu = ld4(a.next.mean + cb); nrs = ld4(a.next.rstd + cb); nga = ld4(a.next.gamma + cb); nbe = ld4(a.next.beta + cb);
                        }
                        f32x4 o2[SNS];
                        split_gemm_n(planes, ns, a.WUs, NB, NKB, nb, ix.lane, ix.rs, ix.q, o2);
                        ACCB(4);
                        f32x4 t1 = f4zero(), t2 = f4zero();
#pragma unroll
                        for (int s = 0; s < SNS; ++s) {
                            if (s >= ns) break;
                            const int r = s * P + bdi;
                            const bool rok = r < rows;
                            const size_t erow_b = (size_t)(b * V + i + min(r, rows - 1)) * V * C + ecol_b;
                            f32x4 rsum = split_get4(planes, 16 * s + ix.rs, bcol >> 2);
                            if (valid_b && rok) {
                                f32x4 dpre = dold[s];
                                if (dce.on) dpre *= drop_mul4(dce, drop_rowhash(dce, (uint32_t)erow_b), bcol >> 2);
                                const f32x4 din = o2[s] + dpre;
                                st4(a.de + erow_b * H + bcol, din);
                                if (do_next) {
                                    f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                                    if (ndc.on) dm = drop_mul4(ndc, drop_rowhash(ndc, (uint32_t)erow_b), bcol >> 2);
#pragma unroll
                                    for (int rr = 0; rr < 4; ++rr) {
                                        const float xh = (ntp[s][rr] - nmu[rr]) * nrs[rr];
                                        const float bn = xh * nga[rr] + nbe[rr];
                                        const float d = bn > 0.f ? din[rr] * dm[rr] : 0.f;
                                        t1[rr] += d;
                                        t2[rr] += d * xh;
                                    }
                                }
                            }
                            // sum over the columns j of row i (rows and columns outside the tensor hold zeros in the tile)
                            rsum = red_group4(rsum, P);
                            if (bdj == 0 && rok && bcol < H)
                                st4(a.rowpart + ((size_t)jt * Rn + (size_t)(b * V + i + r) * C + c) * H + bcol, rsum);
                        }
                        if (do_next) wide_acc_cols(nss, bcol, ix.rs == 0 && bcol < H, t1, t2);
                        ACCB(5);
                    }
                }
            }
            ACCB(6);
            __syncthreads();
            ACCB(7);
        }
        {
            // packed tiles: the P staging threads of a column j (one per row group) add their sums up through LDS, in the
            // order of the row groups (the planes are free between units; plane 0 is cleared again afterwards, its padding
            // columns have to read as zeros)
            float* scr = reinterpret_cast<float*>(planes);                                  // [16 tile rows][2][GCN_WIDE_MAX_H]
#pragma unroll
            for (int k = 0; k < SKA; ++k) {
                const int acol = 4 * (ix.g0 + 32 * k);
                if (acol < HP) {
                    st4(scr + (size_t)(ix.arow * 2 + 0) * GCN_WIDE_MAX_H + acol, cve[k]);
                    st4(scr + (size_t)(ix.arow * 2 + 1) * GCN_WIDE_MAX_H + acol, cvn[k]);
                }
            }
            __syncthreads();
            if (adi == 0 && jaok) {
#pragma unroll
                for (int k = 0; k < SKA; ++k) {
                    const int acol = 4 * (ix.g0 + 32 * k);
                    if (acol < H) {
                        f32x4 se = cve[k], sn = cvn[k];
                        for (int d = 1; d < P; ++d) {
                            se += ld4(scr + (size_t)((d * JW + adj) * 2 + 0) * GCN_WIDE_MAX_H + acol);
                            sn += ld4(scr + (size_t)((d * JW + adj) * 2 + 1) * GCN_WIDE_MAX_H + acol);
                        }
                        st4(a.colVe + ((size_t)it * Rn + nrow_ja) * H + acol, se);
                        st4(a.colVn + ((size_t)it * Rn + nrow_ja) * H + acol, sn);
                    }
                }
            }
            __syncthreads();
            for (int e = ix.t * 16; e < 16 * 2 * GCN_WIDE_MAX_H * 4; e += STH * 16) *reinterpret_cast<u32x4*>(planes + e) = u32x4{0u, 0u, 0u, 0u};
            __syncthreads();
        }
    }
    if (do_next) {
#pragma unroll
        for (int k = 0; k < SKB8; ++k) {
            const int nb = ix.w + 8 * k;
            if (nb < NB) {
                const int bcol = 16 * nb + 4 * ix.q;
                if (ix.rs == 0 && bcol < H) {       // the lane that accumulated these columns
                    st4(a.next.partial + (size_t)(a.part_off + blockIdx.x) * 2 * H + bcol, ld4(&nss[0][bcol]));
                    st4(a.next.partial + (size_t)(a.part_off + blockIdx.x) * 2 * H + H + bcol, ld4(&nss[1][bcol]));
                }
            }
        }
    }
    ACCB_END(ix.w, ix.lane);
}
