// "Strip" kernels: one workgroup per 16-row tile, one wavefront per 16-column block.
//
// Phase A (all threads, elementwise, coalesced): thread (arow, ag) owns the float4 at row arow, columns
//   4*ag .. 4*ag+3 of the tile; it forms the tile (fused BatchNorm apply / backward elementwise math),
//   writes whatever must go to HBM as whole rows, and drops the GEMM operand into an LDS tile.
// Phase B (per wavefront w): reads the whole 16 x H operand tile from LDS in MFMA B-operand order, multiplies by
//   the 16 weight rows of output column block w held in registers (MT float4 = 16 MFMAs for H = 64) and
//   finishes the 4 output values per lane.
// Small register footprint (high occupancy) and short per-wavefront dependency chains: the shapes of
// this workload (tens of thousands of rows, H = 64) are latency bound, not bandwidth bound.
// LDS tiles are double buffered, one barrier per tile.
#pragma once
#include "common.hpp"
#include "kernels_fused.hpp"
#include "kernels_bwd.hpp"

template <int MT>
struct StripGeom {
    static constexpr int HP = 16 * MT;
    static constexpr int LS = HP + 4;          // LDS row stride (floats) of operand tiles
    static constexpr int NT = 64 * MT;         // threads per workgroup
    static constexpr int G = 4 * MT;           // float4 column groups per row
};

// out[16w + 4q + r] for row rs = sum_k tile[rs][k] * Wrow[16w + ..][k]; two accumulation chains hide the
// 40-cycle dependent latency of v_mfma_f32_16x16x4_f32.
template <int MT>
__device__ __forceinline__ f32x4 strip_gemm(const float (*tile)[StripGeom<MT>::LS], const f32x4 (&wf)[MT], int rs, int q) {
    f32x4 acc0 = f4zero(), acc1 = f4zero();
#pragma unroll
    for (int s = 0; s < MT; ++s) {
        const f32x4 in = *reinterpret_cast<const f32x4*>(&tile[rs][16 * s + 4 * q]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((t & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][t], in[t], acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][t], in[t], acc1, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

// The same product with the weight fragments of this lane in LDS (wl[s * 64], read inside the loop: 4 MT registers
// less than fragments held for the whole kernel, which is what decides between two and three wavefronts per SIMD in
// the head backward kernel).
template <int MT>
__device__ __forceinline__ f32x4 strip_gemm_lw(const float (*tile)[StripGeom<MT>::LS], const f32x4* __restrict__ wl, int rs, int q) {
    f32x4 acc0 = f4zero(), acc1 = f4zero();
#pragma unroll
    for (int s = 0; s < MT; ++s) {
        const f32x4 in = *reinterpret_cast<const f32x4*>(&tile[rs][16 * s + 4 * q]);
        const f32x4 wf = wl[s * 64];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((t & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t], in[t], acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t], in[t], acc1, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

// weight rows 16w + rs of W ([H][H] row-major), columns 16s + 4q ..: the A operand of column block w
template <int MT>
__device__ __forceinline__ void strip_wfrag(f32x4 (&wf)[MT], const float* __restrict__ W, int H, int w, int rs, int q) {
    const rsrc_t r = make_rsrc(W, (uint32_t)H * H * 4);
#pragma unroll
    for (int s = 0; s < MT; ++s) {
        const int n = 16 * w + rs, k = 16 * s + 4 * q;
        wf[s] = bld4(r, (n < H && k < H) ? (uint32_t)(n * H + k) * 4 : kOob);
    }
}

// The same fragment of the TRANSPOSE of W, read from W as stored: element t of wf[s] is W[16s + 4q + t][16w + rs]
// (four 32-bit loads, the 16 lanes of a row read 64 contiguous bytes).  The backward kernels multiply by W^T; the
// weights are a few KiB and sit in L2, so no transposed copies are made.
template <int MT>
__device__ __forceinline__ void strip_wfrag_t(f32x4 (&wf)[MT], const float* __restrict__ W, int H, int w, int rs, int q) {
    const rsrc_t r = make_rsrc(W, (uint32_t)H * H * 4);
#pragma unroll
    for (int s = 0; s < MT; ++s) {
        const int n = 16 * w + rs, k = 16 * s + 4 * q;
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[s][t] = bld1(r, (n < H && k + t < H) ? (uint32_t)((k + t) * H + n) * 4 : kOob);
    }
}

// Sum over the 16 lanes of a row (all 16 lanes receive the sum) of four values at once with DPP-fused adds:
// four independent chains interleaved, so a result is read by a DPP operand no sooner than three
// instructions after it is written (the hardware needs two); the leading s_nop covers the producer of the inputs.
#define GCN_DPP4(CTRL)                                                   \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ f32x4 red16x4(f32x4 v) {
    float a = v[0], b = v[1], c = v[2], d = v[3];
    asm volatile("s_nop 1\n\t" GCN_DPP4("row_ror:8") GCN_DPP4("row_ror:4") GCN_DPP4("row_ror:2") GCN_DPP4("row_ror:1")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    return f32x4{a, b, c, d};
}

// ------------------------------------------------------------------------------------------------
// Edge forward of layer l (math: see EdgeFwdArgs; MODE = EdgeSrc::mode).  Persistent
// workgroups walk steps = ((b, i, c) unit, 16-row j tile).  The loop body is straight-line: loads use
// clamped (always valid) coordinates and results are masked, so the compiler can keep the loads of step
// n+1 in flight across the MFMA phase of step n (vector-memory operations retire in issue order: the
// epilogue operands of step n are requested before the phase-A operands of step n+1).
// partial layout [gridDim.x][4][H], written by the lanes that own each column.
// ------------------------------------------------------------------------------------------------
#ifdef GCN_PHASE_TRACE
// Build-time option for measurements only: wavefront 0 of workgroup 0 stamps the shader clock at phase boundaries.
__device__ unsigned long long g_phase_trace[512];
#define GCN_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0 && tstep < 16) g_phase_trace[tstep * 8 + (slot)] = clock64(); } while (0)
#else
#define GCN_STAMP(slot) do { } while (0)
#endif
// ONE_TILE: V <= 16, every step ends its unit (no branch around the unit-end code).
template <int MT, int MODE, bool ONE_TILE>
__global__ __launch_bounds__(64 * MT, 3) void k_edge_fwd3(EdgeFwdArgs a) {
    using GM = StripGeom<MT>;
    __shared__ __attribute__((aligned(16))) float tile[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GM::HP];
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;
    const int H = a.H, V = a.V, C = a.C;
    const bool acol_ok = acol < H;
    const int bcol = 16 * w + 4 * q;
    const bool bcol_ok = bcol < H;
    const bool lead = rs == 0 && bcol_ok;
    const uint32_t rowb = (uint32_t)C * H * 4;                                   // bytes between rows j and j+1
    const uint32_t voffA = acol_ok ? (uint32_t)arow * rowb + acol * 4 : kOob;     // phase-A lane offset in a tile
    const uint32_t voffB = bcol_ok ? (uint32_t)rs * rowb + bcol * 4 : kOob;       // phase-B lane offset in a tile
    const uint32_t voffN = bcol_ok ? (uint32_t)bcol * 4 : kOob;                   // lane offset in one node row
    const uint32_t voffL = lead ? (uint32_t)bcol * 4 : kOob;                      // same, leading lanes only
    const EdgeSrc& S = a.src;
#ifdef GCN_TUNING
    const int probe = a.probe;         // measurement builds only: bit k drops one stream of memory traffic
#else
    constexpr int probe = 0;
#endif
    edge_src_fill_consts<MT>(cl, S, H);
    const DropCfg dc = drop_cfg(a.rng, S.drop_sid, S.drop_p, S.drop_on && MODE == 1);
    f32x4 psx1 = f4zero(), psx2 = f4zero();
    const int units = a.B * V * C;
    const int njt = ONE_TILE ? 1 : (V + 15) / 16;
    __syncthreads();
    // Units u = (b * V + i) * C + c (= node row of (b, i, c)) are taken in the order (b, c, i), an eighth of them per XCD
    // (workgroups are dealt to the XCDs round robin by index; the host launches a multiple of 8), the workgroups of an
    // XCD side by side, as in k_edge_fwd4: the node rows Ve_j, Vn_j of a (b, c), read again by every unit of it, then
    // come from that XCD's L2 instead of being fetched by all eight (a third of the kernel's reads at 20 NSFNet graphs:
    // 14.7 -> 13.2 us per launch).  Row indices are 32-bit with one widening multiply by the row bytes.
    // units in (b, c, i) order, an eighth per XCD, the workgroups of an XCD side by side (see k_edge_fwd4)
    const int nslots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int per_xcd = (units + 7) >> 3;
    const int o_hi = min(units, (xcd + 1) * per_xcd);
    int o = xcd * per_xcd + slot;
    const bool none = o >= o_hi;
    if (none) {                                        // uniform: no units for this workgroup
        if (lead) {
            float* p = a.partial + (size_t)blockIdx.x * 4 * H;
            st4(p + bcol, psx1); st4(p + H + bcol, psx1); st4(p + 2 * H + bcol, psx1); st4(p + 3 * H + bcol, psx1);
        }
        return;
    }
    const uint32_t hb = (uint32_t)H * 4;               // bytes of one row
    auto rowp = [&](const float* base, uint32_t row) { return (const float*)((const char*)base + (uint64_t)row * hb); };
    auto rowpw = [&](float* base, uint32_t row) { return (float*)((char*)base + (uint64_t)row * hb); };
    int jt = 0, buf = 0;
    int b, c, i;
    { const int bc = o / V; i = o - bc * V; b = bc / C; c = bc - b * C; }
    int bi = b * V + i, u = bi * C + c;
    f32x4 ra = f4zero(), rb = f4zero();
    float rcap = 0.f;
    auto tile_bytes = [&](int jt_) { return (uint32_t)(min(16, V - jt_ * 16) - 1) * rowb + hb; };
    auto load_raw = [&](int bi_, int c_, int jt_) {
        const uint32_t erow0 = (uint32_t)(bi_ * V + jt_ * 16) * C + c_;
        const uint32_t nb = (probe & 1) ? 0u : tile_bytes(jt_);
        if constexpr (MODE == 0) ra = bld4(make_rsrc(rowp(S.e_in, erow0), nb), voffA);
        if constexpr (MODE == 1) { ra = bld4(make_rsrc(rowp(S.prev_tmp, erow0), nb), voffA); rb = bld4(make_rsrc(rowp(S.prev_in, erow0), nb), voffA); }
        if constexpr (MODE == 2)
            rcap = bld1(make_rsrc(S.cap + (uint32_t)(bi_ * V + jt_ * 16), (uint32_t)min(16, V - jt_ * 16) * 4), (uint32_t)arow * 4);
    };
    // The first step's phase-A operands are requested before every other prologue load: when the loop
    // is entered at least as many younger memory operations are outstanding as on the loop's back edge, and
    // the wait at the top of the loop is the steady-state count rather than "everything".
    load_raw(bi, c, 0);
    __builtin_amdgcn_sched_barrier(0);       // keep the instruction scheduler from moving these past the loads below
    f32x4 wf[MT];
    strip_wfrag<MT>(wf, a.WU, H, w, rs, q);
    const f32x4 bu = bld4(make_rsrc(a.bU, hb), voffN);
    f32x4 vei = bld4(make_rsrc(rowp(a.Ve, u), hb), voffN);
    f32x4 uxv = bld4(make_rsrc(rowp(a.Ux, u), hb), voffN);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 num = f4zero(), den = f4zero(), s1 = f4zero(), s2 = f4zero();
#ifdef GCN_PHASE_TRACE
    int tstep = 0;
#endif
    while (true) {
        GCN_STAMP(0);
        const uint32_t erow0 = (uint32_t)(bi * V + jt * 16) * C + c;
        const uint32_t nbytes = tile_bytes(jt);
        // ---- phase A of the current step
        {
            const int j = jt * 16 + arow;
            const bool valid = j < V && acol_ok;
            f32x4 val;
            if constexpr (MODE == 0) val = ra;
            if constexpr (MODE == 2) val = ld4(&cl[CL_WE][acol]) * rcap;
            if constexpr (MODE == 1) {
                const f32x4 pm = ld4(&cl[CL_PMEAN][acol]), pr = ld4(&cl[CL_PRSTD][acol]);
                const f32x4 pg = ld4(&cl[CL_PGAMMA][acol]), pb = ld4(&cl[CL_PBETA][acol]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float bn = (ra[k] - pm[k]) * pr[k] * pg[k] + pb[k];
                    val[k] = (bn > 0.f ? bn : 0.f) + rb[k];
                }
                if (dc.on) val *= drop_mul4(dc, drop_rowhash(dc, erow0 + (uint32_t)arow * C), ag);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) val[k] = valid ? val[k] : 0.f;
            if constexpr (MODE != 0) bst4(make_rsrc(rowpw(S.e_out, erow0), (probe & 2) ? 0u : nbytes), voffA, val);
            st4(&tile[buf][arow][acol], val);
        }
        // ---- next step's coordinates
        GCN_STAMP(1);
        const bool unit_end = ONE_TILE || jt + 1 == njt;
        int nu = u, njt2 = jt + 1, nbi = bi, nc = c, nb = b, ni = i;
        int no = o;
        bool more = true;
        if (unit_end) {
            no = o + nslots;
            more = no < o_hi;
            if (more) {
                const int bc = no / V;
                ni = no - bc * V; nb = bc / C; nc = bc - nb * C;
                nbi = nb * V + ni; nu = nbi * C + nc; njt2 = 0;
            } else {
                njt2 = jt; no = o;                                              // last step: reload something valid
            }
        }
        // ---- loads, oldest needed first: next unit's node rows, this step's epilogue rows, next step's phase A
        const bool valid = jt * 16 + rs < V && bcol_ok;
        const f32x4 vei_n = bld4(make_rsrc(rowp(a.Ve, nu), hb), voffN);
        const f32x4 uxv_n = bld4(make_rsrc(rowp(a.Ux, nu), hb), voffN);
        const uint32_t n0 = (uint32_t)(b * V + jt * 16) * C + c;                 // node row (b, 16 jt, c)
        const f32x4 vej = bld4(make_rsrc(rowp(a.Ve, n0), (probe & 8) ? 0u : nbytes), voffB);
        const f32x4 vnj = bld4(make_rsrc(rowp(a.Vn, n0), (probe & 8) ? 0u : nbytes), voffB);
        load_raw(nbi, nc, njt2);
        GCN_STAMP(2);
        __syncthreads();
        GCN_STAMP(3);
        // ---- phase B
        f32x4 et = strip_gemm<MT>(tile[buf], wf, rs, q);
#ifdef GCN_PHASE_TRACE
        asm volatile("" : "+v"(et));
        GCN_STAMP(4);
#endif
        et += (vei + bu) + vej;
#ifdef GCN_PHASE_TRACE
        asm volatile("" : "+v"(et));
        GCN_STAMP(5);
#endif
        bst4(make_rsrc(rowpw(a.e_tmp, erow0), (probe & 4) ? 0u : nbytes), voffB, et);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = valid ? et[r] : 0.f;
            const float g = valid ? sigmoidf_(et[r]) : 0.f;
            num[r] += g * vnj[r];
            den[r] += g;
            s1[r] += x;
            s2[r] += x * x;
        }
        buf ^= 1;
        if (unit_end) {
            // ---- unit end: aggregate over j, node update, statistics
            num = red16x4(num); den = red16x4(den);
            f32x4 D = f32x4{1.f, 1.f, 1.f, 1.f}, A = num;
            if (a.mean_agg) {
                D = den + 1e-20f;
#pragma unroll
                for (int r = 0; r < 4; ++r) A[r] = num[r] * __builtin_amdgcn_rcpf(D[r]);      // 1 ulp reciprocal
            }
            const f32x4 xt = uxv + A;
            bst4(make_rsrc(rowpw(a.x_tmp, u), hb), voffL, xt);
            bst4(make_rsrc(rowpw(a.Aout, u), hb), voffL, A);
            bst4(make_rsrc(rowpw(a.Dout, u), hb), voffL, D);
            psx1 += xt; psx2 += xt * xt;
            num = f4zero(); den = f4zero();
        }
        GCN_STAMP(6);
#ifdef GCN_PHASE_TRACE
        ++tstep;
#endif
        if (!more) break;
        u = nu; jt = njt2; bi = nbi; c = nc; b = nb; i = ni; vei = vei_n; uxv = uxv_n;
        o = no;
    }
    s1 = red16x4(s1); s2 = red16x4(s2);        // edge statistics: per-lane sums over all steps, reduced once
    if (lead) {
        float* p = a.partial + (size_t)blockIdx.x * 4 * H;
        st4(p + bcol, s1); st4(p + H + bcol, s2); st4(p + 2 * H + bcol, psx1); st4(p + 3 * H + bcol, psx2);
    }
}

// ------------------------------------------------------------------------------------------------
// Node forward of layer l (math: see NodeFwdArgs): one workgroup per 16-row tile;
// each wavefront produces its column block of all three linears.
// ------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(64 * MT) void k_node_fwd3(NodeFwdArgs a) {
    using GM = StripGeom<MT>;
    __shared__ __attribute__((aligned(16))) float tile[2][16][GM::LS];
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;
    const int H = a.H;
    const bool acol_ok = acol < H;
    const int bcol = 16 * w + 4 * q;
    const bool bcol_ok = bcol < H;
    f32x4 pm = f4zero(), pr = f4zero(), pg = f4zero(), pb = f4zero();
    if (a.apply && acol_ok) { pm = ld4(a.pmean + acol); pr = ld4(a.prstd + acol); pg = ld4(a.pgamma + acol); pb = ld4(a.pbeta + acol); }
    f32x4 wf[3][MT];
    f32x4 bias[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        strip_wfrag<MT>(wf[k], a.W[k], H, w, rs, q);
        bias[k] = bcol_ok ? ld4(a.bias[k] + bcol) : f4zero();
    }
    const DropCfg dc = drop_cfg(a.rng, a.drop_sid, a.drop_p, a.drop_on && a.apply);
    const int ntiles = (a.rows + 15) / 16;
    int buf = 0;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        {
            const int row = tl * 16 + arow;
            const bool valid = row < a.rows && acol_ok;
            f32x4 val = f4zero();
            if (valid) {
                if (!a.apply) {
                    val = ld4(a.x_in + (size_t)row * H + acol);
                } else {
                    const f32x4 tp = ld4(a.prev_tmp + (size_t)row * H + acol);
                    const f32x4 rin = ld4(a.prev_in + (size_t)row * H + acol);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float bn = (tp[r] - pm[r]) * pr[r] * pg[r] + pb[r];
                        val[r] = (bn > 0.f ? bn : 0.f) + rin[r];
                    }
                    if (dc.on) val *= drop_mul4(dc, drop_rowhash(dc, (uint32_t)row), ag);
                    if (a.x_out != nullptr) st4(a.x_out + (size_t)row * H + acol, val);
                }
            }
            if (acol < GM::HP) st4(&tile[buf][arow][acol], val);
        }
        __syncthreads();
        const int row = tl * 16 + rs;
        const bool valid = row < a.rows && bcol_ok;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const f32x4 o = strip_gemm<MT>(tile[buf], wf[k], rs, q) + bias[k];
            if (valid) st4(a.y[k] + (size_t)row * H + bcol, o);
        }
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Edge backward of layer l (math: see EdgeBwdArgs).  Persistent workgroups walk
// steps = ((b, c, i-chunk, j-tile) unit, i of the chunk); straight-line loop body with clamped loads as in
// k_edge_fwd3.
// ------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(64 * MT, 3) void k_edge_bwd3(EdgeBwdArgs a) {
    using GM = StripGeom<MT>;
    constexpr int HP = GM::HP;
    __shared__ __attribute__((aligned(16))) float tdet[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float tpre[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float ce[6][HP];     // edges: mean, rstd, gamma, beta, m1, m2
    __shared__ __attribute__((aligned(16))) float cx[6][HP];     // nodes: same
    __shared__ __attribute__((aligned(16))) float ncl[4][HP];    // layer l-1 edges: mean, rstd, gamma, beta
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;
    const int H = a.H, V = a.V, C = a.C, IT = a.IT;
    const bool acol_ok = acol < H;
    const int bcol = 16 * w + 4 * q;
    const bool bcol_ok = bcol < H;
    const bool lead = rs == 0 && bcol_ok;
    const uint32_t rowb = (uint32_t)C * H * 4;                                   // bytes between rows j and j+1
    const uint32_t voffA = acol_ok ? (uint32_t)arow * rowb + acol * 4 : kOob;     // phase-A lane offset in a tile
    const uint32_t voffB = bcol_ok ? (uint32_t)rs * rowb + bcol * 4 : kOob;       // phase-B lane offset in a tile
    const uint32_t voffNA = acol_ok ? (uint32_t)acol * 4 : kOob;                  // phase-A lane offset in one node row
    const uint32_t voffL = lead ? (uint32_t)bcol * 4 : kOob;                      // leading lanes, one node row
    const bool do_next = a.next.tmp != nullptr;
    const bool mean_agg = a.mean_agg != 0;
    const float* __restrict__ ntmp = do_next ? a.next.tmp : a.e_tmp;     // always a readable edge tensor
    {
        float* const dst[16] = {ce[0], ce[1], ce[2], ce[3], ce[4], ce[5], cx[0], cx[1], cx[2], cx[3], cx[4], cx[5], ncl[0], ncl[1], ncl[2], ncl[3]};
        const float* const src[16] = {a.stats, a.stats + H, a.gamma_e, a.beta_e, a.bstats_e, a.bstats_e + H,
                                      a.stats + 2 * H, a.stats + 3 * H, a.gamma_x, a.beta_x, a.bstats_x, a.bstats_x + H,
                                      do_next ? a.next.mean : nullptr, do_next ? a.next.rstd : nullptr,
                                      do_next ? a.next.gamma : nullptr, do_next ? a.next.beta : nullptr};
        clds_fill_batch<16>(dst, src, H, HP);
    }
    const DropCfg dce = drop_cfg(a.rng, a.sid_e, a.p, a.training);
    const DropCfg dcx = drop_cfg(a.rng, a.sid_x, a.p, a.training);
    const DropCfg ndc = drop_cfg(a.rng, a.next.drop_sid, a.p, a.training && do_next);
    f32x4 na1 = f4zero(), na2 = f4zero();
    const int njt = (V + 15) / 16;
    const int nit = (V + IT - 1) / IT;
    const int units = a.B * nit * njt * C;
    const uint32_t Rn = (uint32_t)a.B * V * C;
    __syncthreads();
    // contiguous range of units ((b, c, it, jt), jt fastest) per workgroup, so that the units of a workgroup share their
    // commodity's node rows (with c fastest every unit brought in another Vn tile: default2 59.5 -> 57.5 us per launch);
    // counters instead of divisions, 32-bit row indices with one widening multiply by the row bytes
    const int per = (units + (int)gridDim.x - 1) / (int)gridDim.x;
    const int u0 = (int)blockIdx.x * per, u1 = min(units, u0 + per);
    if (u0 >= u1) {                                    // uniform: no units for this workgroup
        if (do_next && lead) {
            st4(a.next.partial + (size_t)blockIdx.x * 2 * H + bcol, na1);
            st4(a.next.partial + (size_t)blockIdx.x * 2 * H + H + bcol, na1);
        }
        return;
    }
    const uint32_t hb = (uint32_t)H * 4;               // bytes of one row
    auto rowp = [&](const float* base, uint32_t row) { return (const float*)((const char*)base + (uint64_t)row * hb); };
    auto rowpw = [&](float* base, uint32_t row) { return (float*)((char*)base + (uint64_t)row * hb); };
    auto tile_bytes = [&](int jt_) { return (uint32_t)(min(16, V - jt_ * 16) - 1) * rowb + hb; };
    int u = u0, buf = 0;
    int c, jt, it, b;
    {
        int tt = u;
        jt = tt % njt; tt /= njt;
        it = tt % nit; tt /= nit;
        c = tt % C;
        b = tt / C;
    }
    f32x4 r_dn, r_tp, r_dxn, r_xt, r_Dv, r_ai, vnj;
    auto load_raw = [&](int b_, int c_, int jt_, int i_) {
        const uint32_t bVi = (uint32_t)b_ * V + i_;
        const uint32_t nrow_i = bVi * C + c_;
        const uint32_t erow0 = (bVi * V + jt_ * 16) * C + c_;
        const uint32_t nb = tile_bytes(jt_);
        r_dn = bld4(make_rsrc(rowp(a.de, erow0), nb), voffA);
        r_tp = bld4(make_rsrc(rowp(a.e_tmp, erow0), nb), voffA);
        r_dxn = bld4(make_rsrc(rowp(a.dx, nrow_i), hb), voffNA);
        r_xt = bld4(make_rsrc(rowp(a.x_tmp, nrow_i), hb), voffNA);
        r_Dv = bld4(make_rsrc(rowp(a.Dn, nrow_i), hb), voffNA);
        r_ai = bld4(make_rsrc(rowp(a.Aagg, nrow_i), hb), voffNA);
    };
    auto load_vnj = [&](int b_, int c_, int jt_) {
        vnj = bld4(make_rsrc(rowp(a.Vn, ((uint32_t)b_ * V + jt_ * 16) * C + c_), tile_bytes(jt_)), voffA);
    };
    int i = it * IT, i_end = min(V, (it + 1) * IT);
    // first step's operands before every other prologue load (see k_edge_fwd3)
    load_raw(b, c, jt, i);
    load_vnj(b, c, jt);
    __builtin_amdgcn_sched_barrier(0);       // keep the instruction scheduler from moving these past the loads below
    f32x4 wf[MT];
    strip_wfrag_t<MT>(wf, a.WUt, H, w, rs, q);          // a.WUt: edge_feat.U weight as stored
    __builtin_amdgcn_sched_barrier(0);
    f32x4 cve = f4zero(), cvn = f4zero();
    while (true) {
        const uint32_t bVi = (uint32_t)b * V + i;
        const uint32_t nrow_i = bVi * C + c;
        const uint32_t erow0 = (bVi * V + jt * 16) * C + c;
        const uint32_t nbytes = tile_bytes(jt);
        const int ja = jt * 16 + arow;
        const bool valid_a = ja < V && acol_ok;
        // ---- phase A
        {
            const f32x4 xmu = ld4(&cx[0][acol]), xrs = ld4(&cx[1][acol]), xga = ld4(&cx[2][acol]);
            const f32x4 xbe = ld4(&cx[3][acol]), xm1 = ld4(&cx[4][acol]), xm2 = ld4(&cx[5][acol]);
            const f32x4 emu = ld4(&ce[0][acol]), ers = ld4(&ce[1][acol]), ega = ld4(&ce[2][acol]);
            const f32x4 ebe = ld4(&ce[3][acol]), em1 = ld4(&ce[4][acol]), em2 = ld4(&ce[5][acol]);
            f32x4 dmx = f32x4{1.f, 1.f, 1.f, 1.f}, dme = f32x4{1.f, 1.f, 1.f, 1.f};
            if (dcx.on) {
                dmx = drop_mul4(dcx, drop_rowhash(dcx, nrow_i), ag);
                dme = drop_mul4(dce, drop_rowhash(dce, erow0 + (uint32_t)arow * C), ag);
            }
            f32x4 det, dpre;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // node-side BatchNorm backward of row (b, i, c): qi = dx_tmp / D
                const float xhx = (r_xt[r] - xmu[r]) * xrs[r];
                const float bnx = xhx * xga[r] + xbe[r];
                const float dxd = r_dxn[r] * dmx[r];
                const float dbnx = bnx > 0.f ? dxd : 0.f;
                const float dxt = xga[r] * xrs[r] * (dbnx - xm1[r] - xhx * xm2[r]);
                const float qi = mean_agg ? dxt * __builtin_amdgcn_rcpf(r_Dv[r]) : dxt;
                const float aiv = mean_agg ? r_ai[r] : 0.f;
                const float x = r_tp[r];
                const float xh = (x - emu[r]) * ers[r];
                const float bn = xh * ega[r] + ebe[r];
                const float d = r_dn[r] * dme[r];
                const float dbn = bn > 0.f ? d : 0.f;
                const float g = sigmoidf_(x);
                const float v = ega[r] * ers[r] * (dbn - em1[r] - xh * em2[r]) + qi * (vnj[r] - aiv) * g * (1.f - g);
                det[r] = valid_a ? v : 0.f;
                dpre[r] = valid_a ? d : 0.f;
                cve[r] += det[r];
                cvn[r] += valid_a ? qi * g : 0.f;
            }
            bst4(make_rsrc(rowpw(a.det_out, erow0), nbytes), voffA, det);
            st4(&tdet[buf][arow][acol], det);
            st4(&tpre[buf][arow][acol], dpre);
        }
        // ---- next step's coordinates
        const bool unit_end = i + 1 >= i_end;
        int nu = u, ni = i + 1, nc = c, njt2 = jt, nit2 = it, nb = b, ni_end = i_end;
        if (unit_end) {
            nu = u + 1; njt2 = jt + 1;
            if (njt2 == njt) {
                njt2 = 0; nit2 = it + 1;
                if (nit2 == nit) { nit2 = 0; nc = c + 1; if (nc == C) { nc = 0; nb = b + 1; } }
            }
            ni = nit2 * IT; ni_end = min(V, (nit2 + 1) * IT);
        }
        const bool more = nu < u1;
        if (!more) { nu = u; ni = i; nc = c; njt2 = jt; nit2 = it; nb = b; ni_end = i_end; }   // last step: reload something valid
        // ---- operand of this step's statistics, the finished unit's column sums, the next step's phase-A operands
        const bool valid_b = jt * 16 + rs < V && bcol_ok;
        const f32x4 ntp = bld4(make_rsrc(rowp(ntmp, erow0), nbytes), voffB);
        {
            // dropped by the range check (zero records) while the unit continues
            const uint32_t colrow = (uint32_t)it * Rn + ((uint32_t)b * V + jt * 16) * C + c;
            const uint32_t nbu = unit_end ? nbytes : 0u;
            bst4(make_rsrc(rowpw(a.colVe, colrow), nbu), voffA, cve);
            bst4(make_rsrc(rowpw(a.colVn, colrow), nbu), voffA, cvn);
#pragma unroll
            for (int r = 0; r < 4; ++r) { cve[r] = unit_end ? 0.f : cve[r]; cvn[r] = unit_end ? 0.f : cvn[r]; }
        }
        load_raw(nb, nc, njt2, ni);
        load_vnj(nb, nc, njt2);
        __syncthreads();
        // ---- phase B
        const f32x4 din = strip_gemm<MT>(tdet[buf], wf, rs, q) + ld4(&tpre[buf][rs][bcol]);
        f32x4 rsum = ld4(&tdet[buf][rs][bcol]);
        bst4(make_rsrc(rowpw(a.de, erow0), nbytes), voffB, din);
        {
            const f32x4 nmu = ld4(&ncl[0][bcol]), nrs = ld4(&ncl[1][bcol]), nga = ld4(&ncl[2][bcol]), nbe = ld4(&ncl[3][bcol]);
            f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
            if (ndc.on) dm = drop_mul4(ndc, drop_rowhash(ndc, erow0 + (uint32_t)rs * C), bcol >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = (ntp[r] - nmu[r]) * nrs[r];
                const float bn = xh * nga[r] + nbe[r];
                const float d = (bn > 0.f && valid_b) ? din[r] * dm[r] : 0.f;
                na1[r] += d;
                na2[r] += d * xh;
            }
        }
        rsum = red16x4(rsum);      // invalid rows hold zeros in the tile
        bst4(make_rsrc(rowpw(a.rowpart, (uint32_t)jt * Rn + nrow_i), hb), voffL, rsum);
        buf ^= 1;
        if (!more) break;
        u = nu; i = ni; b = nb; c = nc; jt = njt2; it = nit2; i_end = ni_end;
    }
    if (do_next) {
        na1 = red16x4(na1); na2 = red16x4(na2);
        if (lead) {
            st4(a.next.partial + (size_t)blockIdx.x * 2 * H + bcol, na1);
            st4(a.next.partial + (size_t)blockIdx.x * 2 * H + H + bcol, na2);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Node backward of layer l (math: see NodeBwdArgs): one workgroup per 16-row tile.
// ------------------------------------------------------------------------------------------------
// Device-coherent store of four floats (one 32-bit store each, performed at device scope): for the few values
// that another workgroup of the same launch reads.  A device-scope release fence would do for ordinary stores too,
// but it writes back the whole L2 of the XCD, which costs far more than the kernel around it.
__device__ __forceinline__ void st4_dev(float* p, const f32x4& v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) __hip_atomic_store(p + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Called by the 64 * MT threads of every workgroup after workgroup bid of nb has stored its row of `partial`
// ([nb][ncols], ncols <= 32 * MT, a multiple of 4) with st4_dev.  Returns true in the last workgroup to arrive, with
// the column sums in tot[0 .. ncols) (LDS, 8 * 32 * MT doubles of scratch).  The sums are formed in a fixed order
// (eight interleaved row groups per column, then a tree over the groups), so which workgroup arrives last does
// not matter.  Ordering: every value handed between workgroups is stored and loaded at device scope (sc1,
// written through).  Each wavefront waits for its own stores to be acknowledged with an explicit s_waitcnt vmcnt(0)
// (a workgroup-scope fence emits no such wait on gfx950), the barrier then orders all of them before the
// device-scope increment of the arrival counter, and the reader takes its ticket before it loads.
// The loads of one thread are issued in batches of twelve: up to 96 rows the tail costs one ticket and one round of loads.
template <int MT>
__device__ __forceinline__ bool last_block_totals(const LastBlockFinish& f, const float* partial, int ncols, double* tot, int bid, int nb) {
    constexpr int CG = 8 * MT;                 // column groups of four
    int* ticket = reinterpret_cast<int*>(tot);
    const int t = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's st4_dev stores have reached memory
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (t == 0) *ticket = __hip_atomic_fetch_add(f.counters, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = *ticket == nb - 1;
    __syncthreads();
    if (!last) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int cg = t % CG, rg = t / CG;        // rg in 0..7
    const rsrc_t r = make_rsrc(partial, (uint32_t)nb * (uint32_t)ncols * 4u);
    const uint32_t rowb = (uint32_t)ncols * 4u;
    const uint32_t base = cg * 4 < ncols ? (uint32_t)cg * 16u : kOob;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    constexpr int NB = 12;                      // rows of this thread's group per batch of loads
    for (int r0 = rg; r0 < nb; r0 += 8 * NB) {
        f32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int row = r0 + 8 * k;
            v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, row < nb ? base + (uint32_t)row * rowb : kOob, 0, 16 /* sc1: device scope */));
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) { a0 += (double)v[k][0]; a1 += (double)v[k][1]; a2 += (double)v[k][2]; a3 += (double)v[k][3]; }
    }
    __syncthreads();                           // the ticket word is part of tot
    double* my = tot + rg * (4 * CG) + cg * 4;
    my[0] = a0; my[1] = a1; my[2] = a2; my[3] = a3;
    __syncthreads();
    for (int sft = 4; sft > 0; sft >>= 1) {
        if (rg < sft) {
#pragma unroll
            for (int k = 0; k < 4; ++k) my[k] += my[sft * (4 * CG) + k];
        }
        __syncthreads();
    }
    if (t == 0) __hip_atomic_store(f.counters, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <int MT>
struct NodeBwdLds {
    using GM = StripGeom<MT>;
    float tu[2][16][GM::LS];     // dUx = dx_tmp
    float tn[2][16][GM::LS];     // dVn
    float te[2][16][GM::LS];     // dVe
    float tp[2][16][GM::LS];     // dpre (residual branch)
};
// Workgroup bid of nblk; uses the first 64 * MT threads of the workgroup.
template <int MT>
__device__ __forceinline__ void node_bwd_body(const NodeBwdArgs& a, int bid, int nblk, NodeBwdLds<MT>& L) {
    using GM = StripGeom<MT>;
    auto& tu = L.tu; auto& tn = L.tn; auto& te = L.te; auto& tp = L.tp;
    const int t = threadIdx.x;
    if (t >= 64 * MT) return;                         // whole wavefronts: they take no part in the barriers below
    const int lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;
    const int H = a.H;
    const bool acol_ok = acol < H;
    const int bcol = 16 * w + 4 * q;
    const bool bcol_ok = bcol < H;
    const bool do_next = a.next.tmp != nullptr;
    f32x4 xmu = f4zero(), xrs = f4zero(), xga = f4zero(), xbe = f4zero(), xm1 = f4zero(), xm2 = f4zero();
    if (acol_ok) {
        xmu = ld4(a.mean + acol); xrs = ld4(a.rstd + acol); xga = ld4(a.gamma + acol); xbe = ld4(a.beta + acol);
        xm1 = ld4(a.bstats + acol); xm2 = ld4(a.bstats + H + acol);
    }
    f32x4 wfu[MT], wfn[MT], wfe[MT];
    strip_wfrag_t<MT>(wfu, a.WnUt, H, w, rs, q);        // the three weights as stored
    strip_wfrag_t<MT>(wfn, a.WnVt, H, w, rs, q);
    strip_wfrag_t<MT>(wfe, a.WeVt, H, w, rs, q);
    f32x4 nmu = f4zero(), nrs = f4zero(), nga = f4zero(), nbe = f4zero();
    if (do_next && bcol_ok) { nmu = ld4(a.next.mean + bcol); nrs = ld4(a.next.rstd + bcol); nga = ld4(a.next.gamma + bcol); nbe = ld4(a.next.beta + bcol); }
    const DropCfg dcx = drop_cfg(a.rng, a.sid_x, a.p, a.training);
    const DropCfg ndc = drop_cfg(a.rng, a.next.drop_sid, a.p, a.training && do_next);
    f32x4 na1 = f4zero(), na2 = f4zero();
    const int ntiles = (a.rows + 15) / 16;
    int buf = 0;
    for (int tl = bid; tl < ntiles; tl += nblk) {
        {
            const int row = tl * 16 + arow;
            const bool valid = row < a.rows && acol_ok;
            f32x4 dux = f4zero(), dvn = f4zero(), dve = f4zero(), dpre = f4zero();
            if (valid) {
                const size_t off = (size_t)row * H + acol;
                const f32x4 dxo = ld4(a.dx + off);
                const f32x4 xt = ld4(a.x_tmp + off);
                f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                if (dcx.on) dm = drop_mul4(dcx, drop_rowhash(dcx, (uint32_t)row), ag);
                for (int k = 0; k < a.nit; ++k) {
                    dvn += ld4(a.colVn + (size_t)k * a.rows * H + off);
                    dve += ld4(a.colVe + (size_t)k * a.rows * H + off);
                }
                for (int k = 0; k < a.njt; ++k) dve += ld4(a.rowpart + (size_t)k * a.rows * H + off);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (xt[r] - xmu[r]) * xrs[r];
                    const float bn = xh * xga[r] + xbe[r];
                    const float d = dxo[r] * dm[r];
                    dpre[r] = d;
                    const float dbn = bn > 0.f ? d : 0.f;
                    dux[r] = xga[r] * xrs[r] * (dbn - xm1[r] - xh * xm2[r]);
                }
                st4(a.dxtmp_out + off, dux);
                st4(a.dVn_out + off, dvn);
                st4(a.dVe_out + off, dve);
            }
            if (acol < GM::HP) {
                st4(&tu[buf][arow][acol], dux); st4(&tn[buf][arow][acol], dvn);
                st4(&te[buf][arow][acol], dve); st4(&tp[buf][arow][acol], dpre);
            }
        }
        const int row = tl * 16 + rs;
        const bool valid = row < a.rows && bcol_ok;
        const f32x4 ntp = (do_next && valid) ? ld4(a.next.tmp + (size_t)row * H + bcol) : f4zero();
        __syncthreads();
        f32x4 acc = strip_gemm<MT>(tu[buf], wfu, rs, q);
        acc += strip_gemm<MT>(tn[buf], wfn, rs, q);
        acc += strip_gemm<MT>(te[buf], wfe, rs, q);
        if (bcol < GM::HP) acc += *reinterpret_cast<const f32x4*>(&tp[buf][rs][bcol]);
        if (valid) {
            st4(a.dx + (size_t)row * H + bcol, acc);
            if (do_next) {
                f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                if (ndc.on) dm = drop_mul4(ndc, drop_rowhash(ndc, (uint32_t)row), bcol >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (ntp[r] - nmu[r]) * nrs[r];
                    const float bn = xh * nga[r] + nbe[r];
                    const float d = bn > 0.f ? acc[r] * dm[r] : 0.f;
                    na1[r] += d;
                    na2[r] += d * xh;
                }
            }
        }
        buf ^= 1;
    }
    if (do_next) {
        na1 = red16x4(na1); na2 = red16x4(na2);
        const bool fin = a.fin.counters != nullptr;
        if (rs == 0 && bcol_ok) {
            float* p = a.next.partial + (size_t)bid * 2 * H + bcol;
            if (fin) { st4_dev(p, na1); st4_dev(p + H, na2); }
            else { st4(p, na1); st4(p + H, na2); }
        }
        if (fin) {
            // the last workgroup to arrive finishes the BatchNorm backward of the layer below: d beta = sum dbn,
            // d gamma = sum dbn * xhat, and their means for the data path
            double* tot = reinterpret_cast<double*>(&L.tu[0][0][0]);     // tiles are done (barrier inside)
            if (last_block_totals<MT>(a.fin, a.next.partial, 2 * H, tot, bid, nblk) && t < 2 * H) {
                const double v = tot[t];
                if (t < H) a.fin.dbeta[t] = (float)v; else a.fin.dgamma[t - H] = (float)v;
                a.fin.bstats[t] = (float)(v * a.fin.inv_n);
            }
        }
    }
}

template <int MT>
__global__ __launch_bounds__(64 * MT) void k_node_bwd3(NodeBwdArgs a) {
    __shared__ __attribute__((aligned(16))) NodeBwdLds<MT> L;
    node_bwd_body<MT>(a, (int)blockIdx.x, (int)gridDim.x, L);
}


// ------------------------------------------------------------------------------------------------
// Weight-gradient GEMM dW[n][k] = sum_row X[row][n] * Y[row][k], db[n] = sum_row X[row][n] (each (problem,
// slice) writes a slab of H*H + H floats).  One workgroup per (row slice, problem); per 16-row tile
// the workgroup stages X and Y rows coalesced into LDS, wavefront w then accumulates the 16 x H block of
// dW rows 16w.. with MFMA operands read from LDS (A[i = n][kk = row], B[kk = row][j = k]).
// ------------------------------------------------------------------------------------------------
template <int MT>
struct TnLds {
    static constexpr int LT = 16 * MT + 16;      // row stride 16 mod 32: conflict-free ds_read_b32
    float tx[2][16][LT];
    float ty[2][16][LT];
};
// One (problem, slice) of the weight-gradient products; uses the first 64 * MT threads of the workgroup.
template <int MT>
__device__ __forceinline__ void gemm_tn_body(int H, const TnProblems& pr, int prob, int slice, TnLds<MT>& L) {
    constexpr int G = 4 * MT;
    const int t = threadIdx.x;
    if (t >= 64 * MT) return;                         // whole wavefronts: they take no part in the barriers below
    const int lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / G, ag = t - arow * G, acol = 4 * ag;
    const bool acol_ok = acol < H;
    const int nslices = pr.nslices[prob];
    if (slice >= nslices) return;                     // uniform per workgroup
    const size_t rows = pr.rows[prob];
    const float* __restrict__ X = pr.X[prob];
    const float* __restrict__ Y = pr.Y[prob];
    const size_t ntiles = (rows + 15) / 16;
    const size_t per = (ntiles + nslices - 1) / nslices;
    const size_t t0 = (size_t)slice * per, t1 = min(ntiles, t0 + per);
    f32x4 acc[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k) acc[k] = f4zero();
    float bsum = 0.f;
    f32x4 xv = f4zero(), yv = f4zero();
    auto fetch = [&](size_t tile) {
        const size_t row = tile * 16 + arow;
        const bool valid = row < rows && acol_ok;
        xv = valid ? ld4(X + row * H + acol) : f4zero();
        yv = valid ? ld4(Y + row * H + acol) : f4zero();
    };
    if (t0 < t1) fetch(t0);
    int buf = 0;
    for (size_t tile = t0; tile < t1; ++tile) {
        st4(&L.tx[buf][arow][acol], xv);
        st4(&L.ty[buf][arow][acol], yv);
        if (tile + 1 < t1) fetch(tile + 1);          // next tile's rows are in flight during this tile's MFMAs
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float av = L.tx[buf][4 * st + q][16 * w + rs];
            bsum += av;
#pragma unroll
            for (int k = 0; k < MT; ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, L.ty[buf][4 * st + q][16 * k + rs], acc[k], 0, 0, 0);
        }
        buf ^= 1;
    }
    float* out = pr.out[prob] + (size_t)slice * pr.out_stride[prob];
#pragma unroll
    for (int k = 0; k < MT; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = 16 * w + 4 * q + r, kc = 16 * k + rs;
            if (nn < H && kc < H) out[(size_t)nn * H + kc] = acc[k][r];
        }
    bsum = redq(bsum);
    const int n = 16 * w + rs;
    if (q == 0 && n < H) out[(size_t)H * H + n] = bsum;
}

template <int MT>
__global__ __launch_bounds__(64 * MT) void k_gemm_tn2(int H, TnProblems pr) {
    __shared__ __attribute__((aligned(16))) TnLds<MT> L;
    gemm_tn_body<MT>(H, pr, blockIdx.z, blockIdx.x, L);
}

// One launch for the work between two layers' backward data paths: the weight-gradient products of the layer
// just finished (pr), and reduce jobs that are ready at that point (the previous layer's product slabs, the
// BatchNorm-backward sums of the layer below).  1-D grid of 256-thread workgroups; seg[k] .. seg[k+1] are the
// workgroups of segment k: segments 0..3 = products (one workgroup per slice), 4..7 = reduce jobs (one per 16 columns).
struct WgradMap {
    int seg[9];
};
template <int MT>
__global__ __launch_bounds__(GCN_BLOCK) void k_wgrad(int H, TnProblems pr, ReduceJobs jobs, WgradMap map) {
    __shared__ __attribute__((aligned(16))) TnLds<MT> L;
    __shared__ double lds[64 * 16];
    const int bid = blockIdx.x;
    int k = 0;
    while (k < 7 && bid >= map.seg[k + 1]) ++k;
    const int idx = bid - map.seg[k];
    if (k < 4) gemm_tn_body<MT>(H, pr, k, idx, L);
    else reduce_slab_body(jobs.job[k - 4], idx, lds);
}

// Tail of the backward pass in one launch: the last products and slab reductions (as k_wgrad) next to the two
// embedding kernels that are ready at the same point.  seg[0..3] products, seg[4..7] reduce jobs, seg[8] partial sums
// of the edge-embedding gradient, seg[9] row sums of the node-embedding gradient, seg[10] = grid.
struct TailMap {
    int seg[11];
};
struct EmbedEdgesArgs {
    size_t rows; int C;
    const float* cap; const float* de0; float* partial;
};
template <int MT>
__global__ __launch_bounds__(GCN_BLOCK) void k_bwd_tail(int H, TnProblems pr, ReduceJobs jobs, EmbedEdgesArgs ee, EmbedRowsumArgs er, TailMap map) {
    __shared__ __attribute__((aligned(16))) TnLds<MT> L;
    __shared__ double lds[64 * 16];
    const int bid = blockIdx.x;
    int k = 0;
    while (k < 9 && bid >= map.seg[k + 1]) ++k;
    const int idx = bid - map.seg[k];
    if (k < 4) gemm_tn_body<MT>(H, pr, k, idx, L);
    else if (k < 8) reduce_slab_body(jobs.job[k - 4], idx, lds);
    else if (k == 8) embed_bwd_edges_body<MT>(ee.rows, ee.C, H, ee.cap, ee.de0, ee.partial, idx, map.seg[9] - map.seg[8]);
    else embed_bwd_rowsum_body(er, idx);
}

// The launch between two layers' edge backward kernels: the node backward of layer l (on the critical path: the
// edge kernel of layer l-1 needs its dx and, through the in-kernel finish, the BatchNorm-backward means of layer
// l-1's node BatchNorm), next to it the weight-gradient products that are ready (edge product of layer l, node
// products of layer l+1) and the reduce jobs that are ready.  1-D grid of 256-thread workgroups:
// seg[0] .. seg[1] node backward, seg[1 + k] .. products k = 0..3, seg[5 + k] .. reduce jobs k = 0..3.
struct MidMap {
    int seg[10];
};
template <int MT>
union MidLds {
    NodeBwdLds<MT> nb;
    struct { TnLds<MT> tn; double red[64 * 16]; } wg;
};
template <int MT>
__global__ __launch_bounds__(GCN_BLOCK, 3) void k_bwd_mid(NodeBwdArgs na, int H, TnProblems pr, ReduceJobs jobs, MidMap map) {
    __shared__ __attribute__((aligned(16))) MidLds<MT> L;
    const int bid = blockIdx.x;
    int k = 0;
    while (k < 8 && bid >= map.seg[k + 1]) ++k;
    const int idx = bid - map.seg[k];
    if (k == 0) node_bwd_body<MT>(na, idx, map.seg[1], L.nb);
    else if (k < 5) gemm_tn_body<MT>(H, pr, k - 1, idx, L.wg.tn);
    else reduce_slab_body(jobs.job[k - 5], idx, L.wg.red);
}

// ------------------------------------------------------------------------------------------------
// Head (edge MLP with at most one hidden layer, masking, loss, edge error) in the strip layout, H <= 64.
// Forward: same outputs as k_head_fwd; loss_partial has one row per workgroup.
// Phase A builds the head's input tile from EdgeSrc; phase B: wavefront w computes hidden columns 16 w ..,
// their share of the two output dot products, and the workgroup adds the four shares through LDS.
// ------------------------------------------------------------------------------------------------
// The head's input at (row, acol .. acol + 3), in two steps so that the loads of the next tile can be in flight
// while the current one is processed: head_load issues them (clamped in-range addresses, no predication),
// head_finish forms the value (zero outside the tensor).
struct HeadRaw {
    f32x4 a, b;
    float cv;
};
__device__ __forceinline__ HeadRaw head_load(const EdgeSrc& S, size_t row, size_t rows, int H, int C, int acol, bool acol_ok) {
    HeadRaw r;
    r.a = f4zero(); r.b = f4zero(); r.cv = 0.f;
    const size_t rowc = row < rows ? row : rows - 1;
    const size_t off = rowc * H + (acol_ok ? acol : 0);
    if (S.mode == 0) r.a = ld4(S.e_in + off);
    else if (S.mode == 2) r.cv = S.cap[rowc / C];
    else { r.a = ld4(S.prev_tmp + off); r.b = ld4(S.prev_in + off); }
    return r;
}
template <int MT>
__device__ __forceinline__ f32x4 head_finish(const EdgeSrc& S, const HeadRaw& r, const float (*cl)[16 * MT], const DropCfg& dc,
                                             size_t row, bool valid, int acol, int ag) {
    if (!valid) return f4zero();
    if (S.mode == 0) return r.a;
    if (S.mode == 2) return ld4(&cl[CL_WE][acol]) * r.cv;
    const f32x4 pm = ld4(&cl[CL_PMEAN][acol]), pr = ld4(&cl[CL_PRSTD][acol]);
    const f32x4 pg = ld4(&cl[CL_PGAMMA][acol]), pb = ld4(&cl[CL_PBETA][acol]);
    f32x4 val;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float bn = (r.a[k] - pm[k]) * pr[k] * pg[k] + pb[k];
        val[k] = (bn > 0.f ? bn : 0.f) + r.b[k];
    }
    if (dc.on) val *= drop_mul4(dc, drop_rowhash(dc, (uint32_t)row), ag);
    return val;
}

template <int MT, int NHID>
__global__ __launch_bounds__(64 * MT) void k_head_fwd3(
    size_t rows, int V, int C, int H, int voc, HeadParams hp, EdgeSrc src, const float* __restrict__ cap, int mask_invalid,
    const long long* __restrict__ y_edges, const float* __restrict__ edge_cw, float* __restrict__ logits,
    float* __restrict__ loss_partial, int* __restrict__ counts, int count_reps, const unsigned long long* __restrict__ rng,
    int stream_id0, float p, int training) {
    static_assert(NHID <= 1, "at most one hidden layer");
    using GM = StripGeom<MT>;
    __shared__ __attribute__((aligned(16))) float tile[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GM::HP];
    __shared__ float red[2][MT][16][2];
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;
    const bool acol_ok = acol < H;
    const int bcol = 16 * w + 4 * q;
    const bool bcol_ok = bcol < H;
    edge_src_fill_consts<MT>(cl, src, H);
    const DropCfg sdc = drop_cfg(rng, src.drop_sid, src.drop_p, src.drop_on && src.mode == 1);
    const DropCfg hdc = drop_cfg(rng, stream_id0, p, training);
    f32x4 wf[MT];
    if (NHID) strip_wfrag<MT>(wf, hp.hidW[0], H, w, rs, q);
    const f32x4 b1 = (NHID && bcol_ok) ? ld4(hp.hidB[0] + bcol) : f4zero();
    const f32x4 w0 = bcol_ok ? ld4(hp.outW + bcol) : f4zero();
    const f32x4 w1 = (bcol_ok && voc > 1) ? ld4(hp.outW + H + bcol) : f4zero();
    const float b2_0 = hp.outB[0], b2_1 = voc > 1 ? hp.outB[1] : 0.f;
    const float cw0 = edge_cw ? edge_cw[0] : 0.f, cw1 = edge_cw ? edge_cw[1] : 0.f;
    const size_t ntiles = (rows + 15) / 16;
    const int BC_stride = V * V * C;
    float acc_wn = 0.f, acc_w = 0.f;
    int buf = 0;
    __syncthreads();
    HeadRaw raw = head_load(src, (size_t)blockIdx.x * 16 + arow, rows, H, C, acol, acol_ok);
    for (size_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        // operands of this tile's epilogue first, then the next tile's input: vector-memory operations retire in
        // issue order, and the epilogue must not wait for the prefetch
        const size_t row = tl * 16 + rs;
        const size_t rowc = row < rows ? row : rows - 1;
        const float capv = mask_invalid ? cap[rowc / C] : 1.f;
        const int yv = y_edges != nullptr ? (int)y_edges[rowc] : 0;
        {
            const size_t rowa = tl * 16 + arow;
            st4(&tile[buf][arow][acol], head_finish<MT>(src, raw, cl, sdc, rowa, rowa < rows && acol_ok, acol, ag));
            if (tl + gridDim.x < ntiles) raw = head_load(src, (tl + gridDim.x) * 16 + arow, rows, H, C, acol, acol_ok);
        }
        __syncthreads();
        f32x4 h;
        if (NHID) {
            h = strip_gemm<MT>(tile[buf], wf, rs, q) + b1;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = h[r] > 0.f ? h[r] : 0.f;
            if (hdc.on) h *= drop_mul4(hdc, drop_rowhash(hdc, (uint32_t)row), bcol >> 2);
        } else {
            h = ld4(&tile[buf][rs][bcol]);
        }
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p0 += h[r] * w0[r]; p1 += h[r] * w1[r]; }
        p0 = redq(p0); p1 = redq(p1);
        if (q == 0) { red[buf][w][rs][0] = p0; red[buf][w][rs][1] = p1; }
        __syncthreads();
        if (w == 0 && q == 0 && row < rows) {
            float l0 = b2_0, l1 = b2_1;
#pragma unroll
            for (int k = 0; k < MT; ++k) { l0 += red[buf][k][rs][0]; l1 += red[buf][k][rs][1]; }
            if (mask_invalid && capv == 0.f) { l0 = -1e9f; l1 = -1e9f; }
            if (voc == 1) {
                logits[row] = l0;
            } else {
                logits[row * 2] = l0;
                logits[row * 2 + 1] = l1;
                if (y_edges != nullptr) {
                    const int y = yv;
                    const float mx = fmaxf(l0, l1);
                    const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
                    const float wgt = y ? cw1 : cw0;
                    acc_wn += -wgt * ((y ? l1 : l0) - lse);
                    acc_w += wgt;
                    const int pred = l1 > l0 ? 1 : 0;
                    if (y + pred > 0 && counts != nullptr) {
                        const int b = (int)(row / BC_stride), c = (int)(row % C);
                        const int BC = (int)(rows / BC_stride) * C;
                        int* cnt = counts + (size_t)(blockIdx.x % count_reps) * 2 * BC;     // this workgroup's replica
                        atomicAdd(&cnt[BC + b * C + c], 1);
                        if (pred == y) atomicAdd(&cnt[b * C + c], 1);
                    }
                }
            }
        }
        buf ^= 1;
    }
    if (loss_partial != nullptr && w == 0) {
        acc_wn = red16(acc_wn);      // lanes with q != 0 hold zeros
        acc_w = red16(acc_w);
        if (t == 0) {
            loss_partial[(size_t)blockIdx.x * 2] = acc_wn;
            loss_partial[(size_t)blockIdx.x * 2 + 1] = acc_w;
        }
    }
}

// Backward (same outputs as k_head_bwd, NHID <= 1); wpartial has one row per workgroup, hidWt is the transposed
// hidden weight.  wg_out (nullable, NHID = 1): the workgroup also forms its share of the hidden layer's weight gradient
// dpre^T e (one H x H + H slab per workgroup, rows = output channels, then the bias sums) from the two tiles it has in
// LDS anyway; the caller then passes null stashes and neither the head's input nor dpre goes to memory.
template <int MT, int NHID>
__global__ __launch_bounds__(64 * MT, 3) void k_head_bwd3(
    size_t rows, int V, int C, int H, int voc, HeadParams hp, const float* __restrict__ hidWt, EdgeSrc src, NextBn next,
    float* __restrict__ e_stash, const float* __restrict__ cap, int mask_invalid, const long long* __restrict__ y_edges,
    const float* __restrict__ edge_cw, const float* __restrict__ loss_out, const float* __restrict__ dloss,
    const float* __restrict__ dy_pred, float* __restrict__ de, float* __restrict__ dpre_stash,
    float* __restrict__ wpartial, const unsigned long long* __restrict__ rng, int stream_id0, float p, int training, ZeroJobs z,
    float* __restrict__ wg_out) {
    static_assert(NHID <= 1, "at most one hidden layer");
    using GM = StripGeom<MT>;
    __shared__ __attribute__((aligned(16))) float tile[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float tdp[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GM::HP];
    __shared__ float red[2][MT][16][2];
    __shared__ __attribute__((aligned(16))) f32x4 wtl[NHID ? MT * MT * 64 : 1];   // fragments of the transposed hidden weight, [wavefront][s][lane]
    zero_fill(z);
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;
    const bool acol_ok = acol < H;
    const int bcol = 16 * w + 4 * q;
    const bool bcol_ok = bcol < H;
    edge_src_fill_consts<MT>(cl, src, H);
    const DropCfg sdc = drop_cfg(rng, src.drop_sid, src.drop_p, src.drop_on && src.mode == 1);
    const DropCfg hdc = drop_cfg(rng, stream_id0, p, training);
    const bool do_next = next.tmp != nullptr;
    f32x4 wf[MT];
    if (NHID) {
        f32x4 wft[MT];
        strip_wfrag_t<MT>(wft, hp.hidW[0], H, w, rs, q);
#pragma unroll
        for (int s = 0; s < MT; ++s) wtl[(w * MT + s) * 64 + lane] = wft[s];
        strip_wfrag<MT>(wf, hp.hidW[0], H, w, rs, q);
    }
    const f32x4 b1 = (NHID && bcol_ok) ? ld4(hp.hidB[0] + bcol) : f4zero();
    const f32x4 w0 = bcol_ok ? ld4(hp.outW + bcol) : f4zero();
    const f32x4 w1 = (bcol_ok && voc > 1) ? ld4(hp.outW + H + bcol) : f4zero();
    const float b2_0 = hp.outB[0], b2_1 = voc > 1 ? hp.outB[1] : 0.f;
    const float cw0 = edge_cw ? edge_cw[0] : 0.f, cw1 = edge_cw ? edge_cw[1] : 0.f;
    const bool has_loss = (y_edges != nullptr) && voc == 2;
    const float gl = has_loss ? (dloss ? dloss[0] : 1.f) / loss_out[1] : 0.f;
    f32x4 gw0 = f4zero(), gw1 = f4zero(), na1 = f4zero(), na2 = f4zero();
    float gb0 = 0.f, gb1 = 0.f;
    const bool fuse = NHID && wg_out != nullptr;
    f32x4 wacc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wacc[m] = f4zero();
    float bsum = 0.f;
    const size_t ntiles = (rows + 15) / 16;
    int buf = 0;
    __syncthreads();
    HeadRaw raw = head_load(src, (size_t)blockIdx.x * 16 + arow, rows, H, C, acol, acol_ok);
    for (size_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        // operands of this tile's epilogue first, then the next tile's input (see k_head_fwd3)
        const size_t row = tl * 16 + rs;
        const bool rvalid = row < rows;
        const size_t rowc = rvalid ? row : rows - 1;
        const float capv = mask_invalid ? cap[rowc / C] : 1.f;
        const int yv = has_loss ? (int)y_edges[rowc] : 0;
        float dy0 = 0.f, dy1 = 0.f;
        if (dy_pred != nullptr) {
            if (voc == 1) dy0 = dy_pred[rowc];
            else { dy0 = dy_pred[rowc * 2]; dy1 = dy_pred[rowc * 2 + 1]; }
        }
        const f32x4 ntp = do_next ? ld4(next.tmp + rowc * H + (bcol_ok ? bcol : 0)) : f4zero();
        {
            const size_t rowa = tl * 16 + arow;
            const bool valid = rowa < rows && acol_ok;
            const f32x4 val = head_finish<MT>(src, raw, cl, sdc, rowa, valid, acol, ag);
            if (valid && e_stash != nullptr) st4(e_stash + rowa * H + acol, val);
            st4(&tile[buf][arow][acol], val);
            if (tl + gridDim.x < ntiles) raw = head_load(src, (tl + gridDim.x) * 16 + arow, rows, H, C, acol, acol_ok);
        }
        __syncthreads();
        f32x4 h, dmul;
        if (NHID) {
            const f32x4 pre = strip_gemm<MT>(tile[buf], wf, rs, q) + b1;
            f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
            if (hdc.on) dm = drop_mul4(hdc, drop_rowhash(hdc, (uint32_t)row), bcol >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dmul[r] = pre[r] > 0.f ? dm[r] : 0.f;
                h[r] = (pre[r] > 0.f ? pre[r] : 0.f) * dm[r];
            }
        } else {
            h = ld4(&tile[buf][rs][bcol]);
            dmul = f32x4{1.f, 1.f, 1.f, 1.f};
        }
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p0 += h[r] * w0[r]; p1 += h[r] * w1[r]; }
        p0 = redq(p0); p1 = redq(p1);
        if (q == 0) { red[buf][w][rs][0] = p0; red[buf][w][rs][1] = p1; }
        __syncthreads();
        float d0 = 0.f, d1 = 0.f;
        if (rvalid) {
            const bool masked = mask_invalid && capv == 0.f;
            if (!masked) {
                if (has_loss) {
                    float l0 = b2_0, l1 = b2_1;
#pragma unroll
                    for (int k = 0; k < MT; ++k) { l0 += red[buf][k][rs][0]; l1 += red[buf][k][rs][1]; }
                    const int y = yv;
                    const float mx = fmaxf(l0, l1);
                    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), inv = 1.f / (e0 + e1);
                    const float wgt = (y ? cw1 : cw0) * gl;
                    d0 = wgt * (e0 * inv - (y == 0 ? 1.f : 0.f));
                    d1 = wgt * (e1 * inv - (y == 1 ? 1.f : 0.f));
                }
                d0 += dy0;
                d1 += dy1;
            }
        }
        if (w == 0 && q == 0) { gb0 += d0; gb1 += d1; }
        f32x4 dp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gw0[r] += d0 * h[r];
            gw1[r] += d1 * h[r];
            dp[r] = (d0 * w0[r] + d1 * w1[r]) * dmul[r];
        }
        if (NHID && dpre_stash != nullptr && rvalid && bcol_ok) st4(dpre_stash + row * H + bcol, dp);
        f32x4 dv = dp;
        if (NHID) {
            st4(&tdp[buf][rs][bcol], dp);
            __syncthreads();
            dv = strip_gemm_lw<MT>(tdp[buf], wtl + w * MT * 64 + lane + lds_opaque_zero(), rs, q);
            if (fuse) {
                // dW[n][kc] += sum over the tile's rows of dpre[row][n] * e[row][kc] (A[i = n][kk = row], B[kk = row][j = kc]);
                // rows beyond the tensor hold zeros in both tiles
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const float av = tdp[buf][4 * st + q][16 * w + rs];
                    bsum += av;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        wacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tile[buf][4 * st + q][16 * m + rs], wacc[m], 0, 0, 0);
                }
            }
        }
        if (rvalid && bcol_ok) {
            st4(de + row * H + bcol, dv);
            if (do_next) {
                const f32x4 nmu = ld4(&cl[CL_PMEAN][bcol]), nrs = ld4(&cl[CL_PRSTD][bcol]);
                const f32x4 nga = ld4(&cl[CL_PGAMMA][bcol]), nbe = ld4(&cl[CL_PBETA][bcol]);
                f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                if (sdc.on) dm = drop_mul4(sdc, drop_rowhash(sdc, (uint32_t)row), bcol >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (ntp[r] - nmu[r]) * nrs[r];
                    const float bn = xh * nga[r] + nbe[r];
                    const float d = bn > 0.f ? dv[r] * dm[r] : 0.f;
                    na1[r] += d;
                    na2[r] += d * xh;
                }
            }
        }
        buf ^= 1;
    }
    const int wstride = (voc * H + voc + 3) & ~3;
    float* wp = wpartial + (size_t)blockIdx.x * wstride;
    gw0 = red16x4(gw0); gw1 = red16x4(gw1); na1 = red16x4(na1); na2 = red16x4(na2);
    if (rs == 0 && bcol_ok) {
        st4(wp + bcol, gw0);
        if (voc > 1) st4(wp + H + bcol, gw1);
        if (do_next) {
            st4(next.partial + (size_t)blockIdx.x * 2 * H + bcol, na1);
            st4(next.partial + (size_t)blockIdx.x * 2 * H + H + bcol, na2);
        }
    }
    if (w == 0) {
        gb0 = red16(gb0); gb1 = red16(gb1);      // lanes with q != 0 hold zeros
        if (t == 0) { wp[voc * H] = gb0; if (voc > 1) wp[voc * H + 1] = gb1; }
    }
    if (fuse) {
        float* out = wg_out + (size_t)blockIdx.x * ((size_t)H * H + H);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = 16 * w + 4 * q + r, kc = 16 * m + rs;
                if (nn < H && kc < H) out[(size_t)nn * H + kc] = wacc[m][r];
            }
        const float bs = redq(bsum);
        const int n = 16 * w + rs;
        if (q == 0 && n < H) out[(size_t)H * H + n] = bs;
    }
}
