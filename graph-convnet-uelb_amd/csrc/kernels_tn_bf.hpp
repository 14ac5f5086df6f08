// Weight-gradient products for hidden sizes above 64 on the bf16 matrix pipe at fp32 accuracy.
//
// dW[n][k] = sum over rows of X[row][n] Y[row][k] and db[n] = sum over rows of X[row][n] (TnProblems, slab layout as
// k_gemm_tn2), one workgroup per row slice computing the whole H x H slab.  Both operands are cut into three bf16
// pieces as in the edge kernels (x = x1 + x2 + x3 exactly; six bf16 products per block, the three below 2^-26 dropped).
//
// The sum runs over rows, so both MFMA operands want "8 consecutive rows of one column" per lane.  Rows are staged in
// their natural [row][column] order as bf16 planes and read with ds_read_b64_tr_b16, which hands a lane the four rows
// of its column: two such reads form one operand of v_mfma_f32_32x32x16_bf16.  That instruction sums over 16 rows, so a
// chunk of 16 rows is one k step and two chunks fit in LDS (132 KiB): the rows of chunk c + 1 are fetched while chunk
// c is multiplied.  (v_mfma_f32_16x16x32_bf16 sums over 32 rows: two 32-row chunks of six planes do not fit.)
//
// 256 threads: four wavefronts, one per SIMD, wavefront (wr, wc) = (w / 2, w % 2) holds the 5 x 5 blocks of 32 x 32
// outputs with dW rows 160 wr .. and columns 160 wc .. (400 accumulator registers of the 512 a lone wavefront on a
// SIMD may use), so every row of X and Y is read from memory once per product.  Everything else a chunk needs is
// issued in the shadow of its 150 MFMAs: 180 transposed LDS reads, and the twelve float4 per lane of the next chunk,
// which are requested at the start of the chunk and cut into planes one float4 at a time between the later MFMAs.
//
// LDS row stride 704 B (320 bf16 + 64 B): 176 dwords = 48 mod 64, so the four rows x 64 B a half-wavefront takes in
// one transposed read fall on 64 distinct banks.  The 64 pad bytes of a row take the stores of lanes without a
// column (a store without a branch keeps the MFMA sequence one basic block).
#pragma once
#include "kernels_wide_bf.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TS_ROWB = 704;                    // bytes per LDS row of one plane
constexpr int TS_ROWS = 16;                     // rows per chunk
constexpr int TS_PLANE = TS_ROWS * TS_ROWB;     // 11,264
constexpr int TS_BUF = 6 * TS_PLANE;            // planes 0..2: X, 3..5: Y
constexpr int TS_LDS = 2 * TS_BUF;              // 135,168
constexpr int TS_TH = 256;
constexpr int TS_C4 = GCN_WIDE_MAX_H / 4;       // float4 column groups of a row (80)
constexpr int TS_NU = 6;                        // rows per staging thread and chunk: r3, r3 + 3, ..., r3 + 15 (< 16)
static_assert(GCN_WIDE_MAX_H == 320, "k_gemm_tn_ws: 10 x 10 blocks of 32");

// one MFMA operand: rows 8 h .. 8 h + 7 of the lane's column, from the plane at p (lane part of the address included)
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * TS_ROWB));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// acc += a * b with the accumulator block in the accumulation registers (A) or in ordinary vector registers (V).
// Written as assembly because the compiler gives every MFMA of a kernel the same form: with builtins all 25 blocks of a
// wavefront would have to sit in the 256 accumulation registers, and the nine that do not fit are copied in and out
// around every use.  volatile: the MFMAs keep their order among themselves and against the no-ops after the loop.
#define GCN_TN_MFMA_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define GCN_TN_MFMA_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
constexpr int TS_NA = 16;                       // blocks of a wavefront held in accumulation registers (the other 9: vector registers)
constexpr int TS_ONES = GCN_WIDE_MAX_H - 1;     // column of Y that holds 1.0: dW[n][TS_ONES] = db[n]
constexpr int TS_MAX_H = GCN_WIDE_MAX_H - 4;    // the float4 group of the ones column must not be a staged one

#ifndef TN_ABL
#define TN_ABL 0
#endif
__global__ __launch_bounds__(TS_TH) void k_gemm_tn_ws(int H, TnProblems pr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ts_raw[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int prob = blockIdx.z, slice = blockIdx.x, nslices = pr.nslices[prob];
    if (slice >= nslices) return;
    const size_t rows = pr.rows[prob];
    const size_t ntiles = (rows + 15) / 16;
    const size_t per = (ntiles + nslices - 1) / nslices;
    const size_t t0 = (size_t)slice * per, t1 = min(ntiles, t0 + per);
    const size_t row0 = t0 * 16;
    const uint32_t srows = t0 < t1 ? (uint32_t)min(rows - row0, (t1 - t0) * 16) : 0u;
    const rsrc_t rX = make_rsrc(pr.X[prob] + row0 * H, srows * (uint32_t)H * 4u), rY = make_rsrc(pr.Y[prob] + row0 * H, srows * (uint32_t)H * 4u);

    // staging: thread (r3, c4) = (t / 80, t % 80), t < 240, takes columns 4 c4 .. of rows r3 + 3 u of a chunk
    const int c4 = t % TS_C4, r3 = t / TS_C4;
    const bool has_col = t < 3 * TS_C4 && 4 * c4 < H;
    const uint32_t vo = has_col ? (uint32_t)(r3 * H + 4 * c4) * 4u : kOob;        // u = 0 .. 4
    const uint32_t vo5 = (has_col && r3 == 0) ? vo : kOob;                        // u = 5: row 15
    const uint32_t ustep = 3u * (uint32_t)H * 4u, chunk_bytes = (uint32_t)TS_ROWS * (uint32_t)H * 4u;
    const int dummy = 640 + 8 * (t & 7);                                          // pad bytes of row 0
    const int lw = has_col ? r3 * TS_ROWB + 8 * c4 : dummy;
    const int lw5 = (has_col && r3 == 0) ? lw : dummy;
    // operand reads: lane = 32 h + 16 g + 4 qq + pp supplies row 8 h + qq, columns 16 g + 4 pp .. of the block
    const int h = lane >> 5, g = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
    const int rd = (8 * h + qq) * TS_ROWB + 32 * g + 8 * pp;

    for (int e = t * 16; e < TS_LDS; e += TS_TH * 16) *reinterpret_cast<u32x4*>(ts_raw + e) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    // the ones column of Y (first piece 1.0, the other pieces 0), all rows of both buffers
    if (t < 2 * TS_ROWS) *reinterpret_cast<unsigned short*>(ts_raw + (t >> 4) * TS_BUF + 3 * TS_PLANE + (t & 15) * TS_ROWB + 2 * TS_ONES) = 0x3f80;

    f32x16 accA[TS_NA], accV[25 - TS_NA];
#pragma unroll
    for (int n = 0; n < 25; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { if (n < TS_NA) accA[n][r] = 0.f; else accV[n - TS_NA][r] = 0.f; }
    f32x4 sx[TS_NU], sy[TS_NU];
    auto fetch = [&](uint32_t nx) {
#pragma unroll
        for (int u = 0; u < TS_NU; ++u) sx[u] = bld4(rX, (u < 5 ? vo : vo5) + __builtin_amdgcn_readfirstlane(nx + (uint32_t)u * ustep));
#pragma unroll
        for (int u = 0; u < TS_NU; ++u) sy[u] = bld4(rY, (u < 5 ? vo : vo5) + __builtin_amdgcn_readfirstlane(nx + (uint32_t)u * ustep));
    };
    // piece 0..5: X rows, 6..11: Y rows
    auto stash = [&](unsigned char* buf, int piece) {
        const int u = piece % TS_NU, y = piece / TS_NU;
        const f32x4 v = y ? sy[u] : sx[u];
        const Split3 a = split2(v[0], v[1]), b = split2(v[2], v[3]);
        unsigned char* p = buf + (u < 5 ? lw : lw5) + u * 3 * TS_ROWB + y * 3 * TS_PLANE;
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2*>(p + k * TS_PLANE) = uint2{a.p[k], b.p[k]};
    };
    if (t0 < t1) {
        fetch(0u);
#pragma unroll
        for (int piece = 0; piece < 2 * TS_NU; ++piece) stash(ts_raw, piece);
    }
    __syncthreads();
    int cur = 0;
    const int nchunks = __builtin_amdgcn_readfirstlane(t0 < t1 ? (int)(t1 - t0) : 0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const unsigned char* rb = ts_raw + cur * TS_BUF + rd;
        unsigned char* nb = ts_raw + (cur ^ 1) * TS_BUF;
        // Pairs n = 5 kb + i of (Y block kb, X block i), each six MFMAs on one accumulator block.  The operands of pair
        // n + 1 are read while pair n is multiplied (two register sets each); from pair 13 on, one float4 of the next
        // chunk is cut and stored per pair, in three parts between the pair's MFMAs.  The scheduling barriers pin that
        // order: left alone, the compiler moves the cuts up to the fetch and waits there for memory.
        bf16x8 a[2][3], b[2][3];
        auto load_a = [&](bf16x8 (&x)[3], int i) {
#pragma unroll
            for (int p = 0; p < 3; ++p) x[p] = tn_frag(rb + p * TS_PLANE + 64 * (5 * wr + i));
        };
        auto load_b = [&](bf16x8 (&x)[3], int kb) {
#pragma unroll
            for (int p = 0; p < 3; ++p) x[p] = tn_frag(rb + (3 + p) * TS_PLANE + 64 * (5 * wc + kb));
        };
        load_b(b[0], 0);
        load_a(a[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        // the next chunk's rows, requested while the first operands travel; rows past the slice read as zeros, so the
        // last chunk's fetch (of nothing) needs no branch
        if (!(TN_ABL & 4)) fetch((uint32_t)(chunk + 1) * chunk_bytes);
#pragma unroll
        for (int n = 0; n < 25; ++n) {
            const int kb = n / 5, sa = (TN_ABL & 1) ? 0 : (n & 1), sb = (TN_ABL & 1) ? 0 : (kb & 1);
            if (n + 1 < 25 && !(TN_ABL & 1)) {
                load_a(a[sa ^ 1], (n + 1) % 5);
                if ((n + 1) % 5 == 0) load_b(b[sb ^ 1], kb + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#define GCN_TN_MM(pa, pb) do { if (n < TS_NA) GCN_TN_MFMA_A(accA[n < TS_NA ? n : 0], a[sa][pa], b[sb][pb]); else GCN_TN_MFMA_V(accV[n >= TS_NA ? n - TS_NA : 0], a[sa][pa], b[sb][pb]); } while (0)
            if (n >= 13 && !(TN_ABL & 2)) {
                const int piece = n - 13, u = piece % TS_NU, y = piece / TS_NU;
                const f32x4 v = y ? sy[u] : sx[u];
                unsigned char* p = nb + (u < 5 ? lw : lw5) + u * 3 * TS_ROWB + y * 3 * TS_PLANE;
                GCN_TN_MM(2, 0);
                const Split3 c = split2(v[0], v[1]);
                __builtin_amdgcn_sched_barrier(0);
                GCN_TN_MM(0, 2);
                const Split3 d = split2(v[2], v[3]);
                __builtin_amdgcn_sched_barrier(0);
                GCN_TN_MM(1, 1);
#pragma unroll
                for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2*>(p + k * TS_PLANE) = uint2{c.p[k], d.p[k]};
                __builtin_amdgcn_sched_barrier(0);
                GCN_TN_MM(1, 0); GCN_TN_MM(0, 1); GCN_TN_MM(0, 0);
            } else {
                GCN_TN_MM(2, 0); GCN_TN_MM(0, 2); GCN_TN_MM(1, 1); GCN_TN_MM(1, 0); GCN_TN_MM(0, 1); GCN_TN_MM(0, 0);
            }
#undef GCN_TN_MM
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        cur ^= 1;
    }
    // the last MFMAs' results must have landed before other instructions read the accumulators (the compiler cannot
    // see that the statements above are MFMAs and inserts no wait states of its own)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float* out = pr.out[prob] + (size_t)slice * pr.out_stride[prob];
#pragma unroll
    for (int kb = 0; kb < 5; ++kb)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int nblk = 5 * kb + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * (5 * wr + i) + 8 * (r >> 2) + 4 * h + (r & 3), kc = 32 * (5 * wc + kb) + (lane & 31);
                const float v = nblk < TS_NA ? accA[nblk < TS_NA ? nblk : 0][r] : accV[nblk >= TS_NA ? nblk - TS_NA : 0][r];
                if (n < H && kc < H) out[(size_t)n * H + kc] = v;
                if (n < H && kc == TS_ONES) out[(size_t)H * H + n] = v;
            }
        }
}
