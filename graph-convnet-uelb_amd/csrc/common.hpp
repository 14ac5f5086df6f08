// Shared device helpers for the gfx950 GCN kernels.
//
// Register layout used by every row-tile kernel ("tile layout"):
//   a wavefront (64 lanes) owns a tile of 16 rows x HP columns, HP = 16*MT >= H.
//   lane l = (rs, q) with rs = l & 15 (row slot) and q = l >> 4 (column quarter);
//   Tile<MT>.v[m] holds columns 16*m + 4*q .. 16*m + 4*q + 3 of row rs.
// With this layout the lane's own registers are at the same time
//   * the B operand of v_mfma_f32_16x16x4_f32 (B[k = q][j = rs]) for reduction index
//     k = 16*s + 4*q + t at step (s, t), and
//   * the C/D fragment (D[i = 4*q + r][j = rs]) of output block m,
// so Y = X * W^T maps tile -> tile with no LDS traffic and no cross-lane movement, and every
// elementwise epilogue (bias, gate, BatchNorm, residual) is lane-local.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GCN_WAVE 64
#define GCN_BLOCK 256           // 4 wavefronts per workgroup
#define GCN_WAVES_PER_BLOCK 4
#define GCN_BN_EPS 1e-5f

template <int MT>
struct Tile {
    f32x4 v[MT];
};

// Weight fragments (A operand): w[m][s] covers output block m, reduction block s;
// component t multiplies reduction index 16*s + 4*q + t.
template <int MT>
struct WFrag {
    f32x4 w[MT][MT];
};

__device__ __forceinline__ f32x4 f4zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

template <int MT>
__device__ __forceinline__ void tile_zero(Tile<MT>& t) {
#pragma unroll
    for (int m = 0; m < MT; ++m) t.v[m] = f4zero();
}

// row == nullptr (row slot outside the problem) or column >= H loads zeros.
template <int MT>
__device__ __forceinline__ Tile<MT> tile_load(const float* __restrict__ row, int H, int q) {
    Tile<MT> t;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int col = 16 * m + 4 * q;
        t.v[m] = (row != nullptr && col < H) ? *reinterpret_cast<const f32x4*>(row + col) : f4zero();
    }
    return t;
}

template <int MT>
__device__ __forceinline__ void tile_store(float* __restrict__ row, int H, int q, const Tile<MT>& t) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int col = 16 * m + 4 * q;
        if (row != nullptr && col < H) *reinterpret_cast<f32x4*>(row + col) = t.v[m];
    }
}

// Fragments of W ([H out][H in], row-major, as torch.nn.Linear stores it) for Y = X * W^T.
template <int MT>
__device__ __forceinline__ void wfrag_load_nt(WFrag<MT>& f, const float* __restrict__ W, int H, int rs, int q) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int s = 0; s < MT; ++s) {
            const int n = 16 * m + rs, k = 16 * s + 4 * q;
            f.w[m][s] = (n < H && k < H) ? *reinterpret_cast<const f32x4*>(W + (size_t)n * H + k) : f4zero();
        }
}

// Fragments of W^T for Y = X * W (backward through the same Linear): element [k][n] := W[n][k].
template <int MT>
__device__ __forceinline__ void wfrag_load_tr(WFrag<MT>& f, const float* __restrict__ W, int H, int rs, int q) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int s = 0; s < MT; ++s) {
            const int kout = 16 * m + rs;
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n = 16 * s + 4 * q + t;
                v[t] = (kout < H && n < H) ? W[(size_t)n * H + kout] : 0.f;
            }
            f.w[m][s] = v;
        }
}

// out[row][16m + 4q + r] = sum_k in[row][k] * Wf[16m + ..][k]   (exact f32 FMA chain per output)
template <int MT>
__device__ __forceinline__ Tile<MT> tile_gemm(const Tile<MT>& in, const WFrag<MT>& f) {
    Tile<MT> out;
    tile_zero(out);
#pragma unroll
    for (int s = 0; s < MT; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                out.v[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[m][s][t], in.v[s][t], out.v[m], 0, 0, 0);
    return out;
}

// Sum over the 16 row slots with DPP row rotations (one VALU op per step, no LDS crossbar);
// every lane of the 16-lane group ends up with the total.
template <int N>
__device__ __forceinline__ float dpp_row_ror(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float red16(float x) {
    x += dpp_row_ror<8>(x);
    x += dpp_row_ror<4>(x);
    x += dpp_row_ror<2>(x);
    x += dpp_row_ror<1>(x);
    return x;
}
template <int MT>
__device__ __forceinline__ void tile_red16(Tile<MT>& t) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) t.v[m][r] = red16(t.v[m][r]);
}
// Sum over the 4 column quarters (lanes with equal rs).
__device__ __forceinline__ float redq(float x) {
    x += __shfl_xor(x, 16, GCN_WAVE);
    x += __shfl_xor(x, 32, GCN_WAVE);
    return x;
}

// sigmoid through the hardware exp2 / rcp (each ~1 ulp): |error| < 3e-7, far inside the 1e-4 parity budget
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Buffers a kernel clears on the side for later kernels of the same stream (saves separate fill launches).
struct ZeroJobs {
    float* p[3];
    size_t n[3];      // 32-bit words
};
__device__ __forceinline__ void zero_fill(const ZeroJobs& z) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < z.n[k]; i += (size_t)gridDim.x * blockDim.x) z.p[k][i] = 0.f;
}

// ---------------------------------------------------------------- counter-based dropout
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// One key per (seed, step, stream id); rng points at two 64-bit words {seed, step} in device memory
// so that a captured HIP graph sees a new step on every replay.  Element (row, col) is kept iff the
// 16-bit field (col & 1) of hash(row, col >> 1) is >= p * 2^16: one hash serves two neighbouring columns.
struct DropCfg {
    uint32_t key;
    uint32_t thresh;   // keep iff field >= thresh, thresh = p * 2^16
    float scale;       // 1 / (1 - p)
    int on;
};
__device__ __forceinline__ DropCfg drop_cfg(const unsigned long long* __restrict__ rng, int stream_id, float p, int training) {
    DropCfg d;
    d.on = (training && p > 0.f) ? 1 : 0;
    d.key = 0; d.thresh = 0; d.scale = 1.f;
    if (d.on) {
        const unsigned long long seed = rng[0], step = rng[1];
        uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9U);
        k = mix32(k ^ (uint32_t)(seed >> 32));
        k = mix32(k ^ (uint32_t)step);
        k = mix32(k ^ (uint32_t)(step >> 32) ^ ((uint32_t)stream_id * 0x85EBCA6BU));
        d.key = k;
        d.thresh = (p >= 1.f) ? 0x10000U : (uint32_t)(p * 65536.0f);
        d.scale = (p >= 1.f) ? 0.f : 1.0f / (1.0f - p);
    }
    return d;
}
__device__ __forceinline__ uint32_t drop_rowhash(const DropCfg& d, uint32_t row) { return mix32(row ^ d.key); }
// rowhash is already a full avalanche of (row, key); one more multiply round decorrelates the column pairs
__device__ __forceinline__ uint32_t drop_pairhash(uint32_t rowhash, int colpair) {
    uint32_t x = rowhash + (uint32_t)colpair * 0x9E3779B9U;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
    return x;
}
// multiplier applied to element (row, col): 0 if dropped, 1/(1-p) if kept
__device__ __forceinline__ float drop_mul(const DropCfg& d, uint32_t rowhash, int col) {
    const uint32_t h = drop_pairhash(rowhash, col >> 1);
    const uint32_t f = (col & 1) ? (h >> 16) : (h & 0xFFFFU);
    return (f >= d.thresh) ? d.scale : 0.f;
}
// multipliers of the four columns 4g .. 4g+3 (two hashes)
__device__ __forceinline__ f32x4 drop_mul4(const DropCfg& d, uint32_t rowhash, int g) {
    const uint32_t h0 = drop_pairhash(rowhash, 2 * g), h1 = drop_pairhash(rowhash, 2 * g + 1);
    f32x4 m;
    m[0] = ((h0 & 0xFFFFU) >= d.thresh) ? d.scale : 0.f;
    m[1] = ((h0 >> 16) >= d.thresh) ? d.scale : 0.f;
    m[2] = ((h1 & 0xFFFFU) >= d.thresh) ? d.scale : 0.f;
    m[3] = ((h1 >> 16) >= d.thresh) ? d.scale : 0.f;
    return m;
}

// Per-channel BatchNorm constants for the lane's columns.
template <int MT>
struct ChanVec {
    f32x4 v[MT];
};
template <int MT>
__device__ __forceinline__ ChanVec<MT> chan_load(const float* __restrict__ p, int H, int q) {
    ChanVec<MT> c;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int col = 16 * m + 4 * q;
        c.v[m] = (p != nullptr && col < H) ? *reinterpret_cast<const f32x4*>(p + col) : f4zero();
    }
    return c;
}

// ---------------------------------------------------------------- weights / channel constants in LDS
// Fragment-linear image of a weight matrix: fragment (m, s) occupies 64 consecutive float4, one per lane,
// so a wavefront reads it with one conflict-free ds_read_b128.  MT*MT*1 KiB per matrix.
template <int MT, bool TRANSPOSED>
__device__ __forceinline__ void wlds_fill(f32x4* __restrict__ dst, const float* __restrict__ W, int H) {
    const int lane = threadIdx.x & 63, rs = lane & 15, q = lane >> 4;
    const int wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    for (int f = wv; f < MT * MT; f += nwv) {
        const int m = f / MT, s = f - m * MT;
        f32x4 v;
        if (!TRANSPOSED) {
            const int n = 16 * m + rs, k = 16 * s + 4 * q;
            v = (n < H && k < H) ? *reinterpret_cast<const f32x4*>(W + (size_t)n * H + k) : f4zero();
        } else {
            const int kout = 16 * m + rs;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n = 16 * s + 4 * q + t;
                v[t] = (kout < H && n < H) ? W[(size_t)n * H + kout] : 0.f;
            }
        }
        dst[f * 64 + lane] = v;
    }
}
template <int MT>
__device__ __forceinline__ Tile<MT> tile_gemm_lds(const Tile<MT>& in, const f32x4* __restrict__ wl, int lane) {
    Tile<MT> out;
    tile_zero(out);
#pragma unroll
    for (int s = 0; s < MT; ++s) {
        f32x4 w[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) w[m] = wl[(m * MT + s) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                out.v[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[m][t], in.v[s][t], out.v[m], 0, 0, 0);
    }
    return out;
}
// Returns 0 through an opaque register.  Adding it to an LDS index inside a loop keeps the compiler from
// hoisting loop-invariant LDS reads (weights, channel constants) into registers for the whole kernel,
// which would cost hundreds of VGPRs and the occupancy the LDS staging is meant to buy.
__device__ __forceinline__ int lds_opaque_zero() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}

__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* __restrict__ p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Buffer addressing: a 128-bit descriptor (base, byte count) built from workgroup-uniform values plus a
// 32-bit per-lane byte offset.  The hardware range-checks the offset: loads beyond the byte count return
// 0 and stores are dropped, so ragged tiles need neither clamped addresses nor predicated (branched-around)
// memory instructions, and every memory instruction in a loop body is issued unconditionally.
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOob = 0x80000000u;        // an offset beyond any descriptor used here (all are < 2 GiB)
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t p = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 bld4(rsrc_t r, uint32_t voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ float bld1(rsrc_t r, uint32_t voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ void bst4(rsrc_t r, uint32_t voff, const f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}

// Channel constants: nvec vectors of H floats copied into LDS rows of HP = 16*MT floats (zero padded).
__device__ __forceinline__ void clds_fill(float* __restrict__ dst, const float* __restrict__ src, int H, int HP) {
    for (int t = threadIdx.x; t < HP; t += blockDim.x) dst[t] = (src != nullptr && t < H) ? src[t] : 0.f;
}
// N such vectors at once.  One clds_fill after another is a chain of N memory round trips in the prologue of every
// workgroup (load, wait, LDS store, next vector): here all loads are requested before the first is stored.  Range-checked
// buffer loads make a null source or an element beyond H read as 0 without a branch.  ITERS = ceil(HP / blockDim.x).
template <int N, int ITERS = 1>
__device__ __forceinline__ void clds_fill_batch(float* const (&dst)[N], const float* const (&src)[N], int H, int HP) {
    float v[N][ITERS];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const rsrc_t r = make_rsrc(src[k], src[k] != nullptr ? (uint32_t)H * 4u : 0u);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) v[k][it] = bld1(r, (uint32_t)(threadIdx.x + it * blockDim.x) * 4u);
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int t = threadIdx.x + it * blockDim.x;
            if (t < HP) dst[k][t] = v[k][it];
        }
}
__device__ __forceinline__ f32x4 clds_get(const float* __restrict__ row, int m, int q) {
    return *reinterpret_cast<const f32x4*>(row + 16 * m + 4 * q);
}
