// Edge kernels of the strip layout with TWO 16-column output blocks per wavefront (hidden sizes of 2 or 4 column
// blocks: 17..32 and 49..64 channels).  Same tiles, same math and same memory layout as k_edge_fwd3 / k_edge_bwd3 of
// kernels_strip.hpp; what changes is how a 16-row tile is shared out:
//   * a workgroup has MT / 2 wavefronts; wavefront w owns column blocks w and w + MT / 2, multiplies the operand tile
//     it reads from LDS once by both blocks' weight rows (two independent MFMA chains per block) and finishes
//     8 outputs per lane;
//   * in the elementwise phase a thread owns one float4 column group of rows r and r + 8, so everything that depends
//     on the column alone (BatchNorm constants, the node-side backward terms, dropout column hashes) is formed once
//     for two rows.
// Per tile the instruction count drops by about a third against four wavefronts with one block each (coordinates,
// descriptors, barriers and per-column work are paid by half as many wavefronts), and twice as many workgroups fit a
// CU at the same number of wavefronts per SIMD.
#pragma once
#include "kernels_strip.hpp"

template <int MT>
struct Strip2Geom {
    static_assert(MT == 2 || MT == 4, "two column blocks per wavefront need an even number of blocks");
    static constexpr int HP = 16 * MT;
    static constexpr int LS = HP + 4;          // LDS row stride (floats) of operand tiles: conflict-free ds_read_b128
    static constexpr int NW = MT / 2;          // wavefronts per workgroup
    static constexpr int NT = 64 * NW;         // threads per workgroup = 8 rows x G column groups
    static constexpr int G = 4 * MT;           // float4 column groups per row
};

// Both column blocks of one wavefront: out[k][..] for row rs = sum_c tile[rs][c] * W[16 cb_k + ..][c].
// One LDS read of the operand serves both blocks; two accumulation chains per block.
template <int MT>
__device__ __forceinline__ void strip_gemm2(const float (*tile)[Strip2Geom<MT>::LS], const f32x4 (&wf0)[MT], const f32x4 (&wf1)[MT],
                                            int rs, int q, f32x4& o0, f32x4& o1) {
    f32x4 a00 = f4zero(), a01 = f4zero(), a10 = f4zero(), a11 = f4zero();
#pragma unroll
    for (int s = 0; s < MT; ++s) {
        const f32x4 in = *reinterpret_cast<const f32x4*>(&tile[rs][16 * s + 4 * q]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((t & 1) == 0) {
                a00 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[s][t], in[t], a00, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[s][t], in[t], a10, 0, 0, 0);
            } else {
                a01 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[s][t], in[t], a01, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[s][t], in[t], a11, 0, 0, 0);
            }
        }
    }
    o0 = a00 + a01;
    o1 = a10 + a11;
}

// ------------------------------------------------------------------------------------------------
// Edge forward of layer l (math: see EdgeFwdArgs; MODE = EdgeSrc::mode; ONE_TILE: V <= 16).
// partial layout [gridDim.x][4][H], written by the lanes that own each column.
// ------------------------------------------------------------------------------------------------
template <int MT, int MODE, bool ONE_TILE>
__global__ __launch_bounds__(32 * MT, 2) void k_edge_fwd4(EdgeFwdArgs a) {
    using GM = Strip2Geom<MT>;
    __shared__ __attribute__((aligned(16))) float tile[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GM::HP];
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;        // rows arow and arow + 8
    const int H = a.H, V = a.V, C = a.C;
    const bool acol_ok = acol < H;
    const uint32_t rowb = (uint32_t)C * H * 4;                                   // bytes between rows j and j+1
    const uint32_t hb = (uint32_t)H * 4;                                         // bytes of one row
    const uint32_t voffA0 = acol_ok ? (uint32_t)arow * rowb + acol * 4 : kOob;   // phase-A lane offsets in a tile
    const uint32_t voffA1 = acol_ok ? (uint32_t)(arow + 8) * rowb + acol * 4 : kOob;
    int bcol[2];
    bool bcol_ok[2];
    uint32_t voffB[2], voffN[2], voffL[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        bcol[k] = 16 * (w + k * GM::NW) + 4 * q;
        bcol_ok[k] = bcol[k] < H;
        voffB[k] = bcol_ok[k] ? (uint32_t)rs * rowb + bcol[k] * 4 : kOob;       // phase-B lane offset in a tile
        voffN[k] = bcol_ok[k] ? (uint32_t)bcol[k] * 4 : kOob;                   // lane offset in one node row
        voffL[k] = (rs == 0 && bcol_ok[k]) ? (uint32_t)bcol[k] * 4 : kOob;      // same, leading lanes only
    }
    const EdgeSrc& S = a.src;
    edge_src_fill_consts<MT>(cl, S, H);
    const DropCfg dc = drop_cfg(a.rng, S.drop_sid, S.drop_p, S.drop_on && MODE == 1);
    f32x4 psx1[2] = {f4zero(), f4zero()}, psx2[2] = {f4zero(), f4zero()};
    const int units = a.B * V * C;
    const int njt = ONE_TILE ? 1 : (V + 15) / 16;
    __syncthreads();
    // Units u = (b * V + i) * C + c (= node row of (b, i, c)) are taken in the order (b, c, i), i fastest, an eighth of
    // them per XCD (workgroups are dealt to the 8 XCDs round robin by index; the host launches a multiple of 8), and the
    // workgroups of an XCD walk their eighth side by side (workgroup `slot` takes positions slot, slot + nslots, ...):
    // at any time they all work on rows i of the same one or two (b, c), whose node rows Ve_j, Vn_j (2 V H floats, read
    // again by every unit) then come from that XCD's L2.  With contiguous ranges in (b, i, c) order every workgroup
    // walks all C commodities of its rows and the node rows (2 Rn H floats in use at once) are fetched from memory for
    // every unit: a third of the kernel's traffic at 200 nodes x 100 commodities.
    const int nslots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int per_xcd = (units + 7) >> 3;
    const int o_hi = min(units, (xcd + 1) * per_xcd);
    int o = xcd * per_xcd + slot;
    if (o >= o_hi) {                                   // uniform: no units for this workgroup
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (rs == 0 && bcol_ok[k]) {
                float* p = a.partial + (size_t)blockIdx.x * 4 * H + bcol[k];
                st4(p, f4zero()); st4(p + H, f4zero()); st4(p + 2 * H, f4zero()); st4(p + 3 * H, f4zero());
            }
        return;
    }
    auto rowp = [&](const float* base, uint32_t row) { return (const float*)((const char*)base + (uint64_t)row * hb); };
    auto rowpw = [&](float* base, uint32_t row) { return (float*)((char*)base + (uint64_t)row * hb); };
    int jt = 0, buf = 0;
    int b, c, i;
    { const int bc = o / V; i = o - bc * V; b = bc / C; c = bc - b * C; }
    int bi = b * V + i, u = bi * C + c;                // bi = b * V + i
    f32x4 ra[2] = {f4zero(), f4zero()}, rb[2] = {f4zero(), f4zero()};
    float rcap[2] = {0.f, 0.f};
    auto tile_bytes = [&](int jt_) { return (uint32_t)(min(16, V - jt_ * 16) - 1) * rowb + hb; };
    auto load_raw = [&](int bi_, int c_, int jt_) {
        const uint32_t erow0 = (uint32_t)(bi_ * V + jt_ * 16) * C + c_;
        const uint32_t nb = tile_bytes(jt_);
        if constexpr (MODE == 0) {
            const rsrc_t r = make_rsrc(rowp(S.e_in, erow0), nb);
            ra[0] = bld4(r, voffA0); ra[1] = bld4(r, voffA1);
        }
        if constexpr (MODE == 1) {
            const rsrc_t r1 = make_rsrc(rowp(S.prev_tmp, erow0), nb), r2 = make_rsrc(rowp(S.prev_in, erow0), nb);
            ra[0] = bld4(r1, voffA0); ra[1] = bld4(r1, voffA1);
            rb[0] = bld4(r2, voffA0); rb[1] = bld4(r2, voffA1);
        }
        if constexpr (MODE == 2) {
            const rsrc_t r = make_rsrc(S.cap + (uint32_t)(bi_ * V + jt_ * 16), (uint32_t)min(16, V - jt_ * 16) * 4);
            rcap[0] = bld1(r, (uint32_t)arow * 4); rcap[1] = bld1(r, (uint32_t)(arow + 8) * 4);
        }
    };
    // the first step's phase-A operands are requested before every other prologue load (see k_edge_fwd3)
    load_raw(bi, c, 0);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 wf0[MT], wf1[MT];
    strip_wfrag<MT>(wf0, a.WU, H, w, rs, q);
    strip_wfrag<MT>(wf1, a.WU, H, w + GM::NW, rs, q);
    f32x4 bu[2], vei[2], uxv[2];
    {
        const rsrc_t rbu = make_rsrc(a.bU, hb), rve = make_rsrc(rowp(a.Ve, u), hb), rux = make_rsrc(rowp(a.Ux, u), hb);
#pragma unroll
        for (int k = 0; k < 2; ++k) { bu[k] = bld4(rbu, voffN[k]); vei[k] = bld4(rve, voffN[k]); uxv[k] = bld4(rux, voffN[k]); }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 num[2] = {f4zero(), f4zero()}, den[2] = {f4zero(), f4zero()}, s1[2] = {f4zero(), f4zero()}, s2[2] = {f4zero(), f4zero()};
    while (true) {
        const uint32_t erow0 = (uint32_t)(bi * V + jt * 16) * C + c;
        const uint32_t nbytes = tile_bytes(jt);
        // ---- phase A of the current step: rows arow and arow + 8, column group ag
        {
            f32x4 val[2];
            if constexpr (MODE == 0) { val[0] = ra[0]; val[1] = ra[1]; }
            if constexpr (MODE == 2) {
                const f32x4 we = ld4(&cl[CL_WE][acol]);
                val[0] = we * rcap[0]; val[1] = we * rcap[1];
            }
            if constexpr (MODE == 1) {
                const f32x4 pm = ld4(&cl[CL_PMEAN][acol]), pr = ld4(&cl[CL_PRSTD][acol]);
                const f32x4 pg = ld4(&cl[CL_PGAMMA][acol]), pb = ld4(&cl[CL_PBETA][acol]);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float bn = (ra[r][k] - pm[k]) * pr[k] * pg[k] + pb[k];
                        val[r][k] = (bn > 0.f ? bn : 0.f) + rb[r][k];
                    }
                    if (dc.on) val[r] *= drop_mul4(dc, drop_rowhash(dc, erow0 + (uint32_t)(arow + 8 * r) * C), ag);
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool valid = jt * 16 + arow + 8 * r < V && acol_ok;
#pragma unroll
                for (int k = 0; k < 4; ++k) val[r][k] = valid ? val[r][k] : 0.f;
            }
            if constexpr (MODE != 0) {
                const rsrc_t ro = make_rsrc(rowpw(S.e_out, erow0), nbytes);
                bst4(ro, voffA0, val[0]); bst4(ro, voffA1, val[1]);
            }
            st4(&tile[buf][arow][acol], val[0]);
            st4(&tile[buf][arow + 8][acol], val[1]);
        }
        // ---- next step's coordinates
        const bool unit_end = ONE_TILE || jt + 1 == njt;
        int nu = u, njt2 = jt + 1, nbi = bi, nc = c, nb = b, ni = i, no = o;
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
        const bool rvalid = jt * 16 + rs < V;
        f32x4 vei_n[2], uxv_n[2], vej[2], vnj[2];
        {
            const rsrc_t rve = make_rsrc(rowp(a.Ve, nu), hb), rux = make_rsrc(rowp(a.Ux, nu), hb);
            const uint32_t n0 = (uint32_t)(b * V + jt * 16) * C + c;                 // node row (b, 16 jt, c)
            const rsrc_t rvj = make_rsrc(rowp(a.Ve, n0), nbytes), rnj = make_rsrc(rowp(a.Vn, n0), nbytes);
#pragma unroll
            for (int k = 0; k < 2; ++k) { vei_n[k] = bld4(rve, voffN[k]); uxv_n[k] = bld4(rux, voffN[k]); }
#pragma unroll
            for (int k = 0; k < 2; ++k) { vej[k] = bld4(rvj, voffB[k]); vnj[k] = bld4(rnj, voffB[k]); }
        }
        load_raw(nbi, nc, njt2);
        __syncthreads();
        // ---- phase B
        f32x4 et[2];
        strip_gemm2<MT>(tile[buf], wf0, wf1, rs, q, et[0], et[1]);
        {
            const rsrc_t rt = make_rsrc(rowpw(a.e_tmp, erow0), nbytes);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                et[k] += (vei[k] + bu[k]) + vej[k];
                bst4(rt, voffB[k], et[k]);
                const bool valid = rvalid && bcol_ok[k];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = valid ? et[k][r] : 0.f;
                    const float g = valid ? sigmoidf_(et[k][r]) : 0.f;
                    num[k][r] += g * vnj[k][r];
                    den[k][r] += g;
                    s1[k][r] += x;
                    s2[k][r] += x * x;
                }
            }
        }
        buf ^= 1;
        if (unit_end) {
            // ---- unit end: aggregate over j, node update, statistics
            const rsrc_t rx = make_rsrc(rowpw(a.x_tmp, u), hb), rA = make_rsrc(rowpw(a.Aout, u), hb), rD = make_rsrc(rowpw(a.Dout, u), hb);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                num[k] = red16x4(num[k]); den[k] = red16x4(den[k]);
                f32x4 D = f32x4{1.f, 1.f, 1.f, 1.f}, A = num[k];
                if (a.mean_agg) {
                    D = den[k] + 1e-20f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) A[r] = num[k][r] * __builtin_amdgcn_rcpf(D[r]);      // 1 ulp reciprocal
                }
                const f32x4 xt = uxv[k] + A;
                bst4(rx, voffL[k], xt);
                bst4(rA, voffL[k], A);
                bst4(rD, voffL[k], D);
                psx1[k] += xt; psx2[k] += xt * xt;
                num[k] = f4zero(); den[k] = f4zero();
            }
        }
        if (!more) break;
        u = nu; jt = njt2; bi = nbi; c = nc; b = nb; i = ni; o = no;
#pragma unroll
        for (int k = 0; k < 2; ++k) { vei[k] = vei_n[k]; uxv[k] = uxv_n[k]; }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        s1[k] = red16x4(s1[k]); s2[k] = red16x4(s2[k]);        // edge statistics: per-lane sums over all steps, reduced once
        if (rs == 0 && bcol_ok[k]) {
            float* p = a.partial + (size_t)blockIdx.x * 4 * H + bcol[k];
            st4(p, s1[k]); st4(p + H, s2[k]); st4(p + 2 * H, psx1[k]); st4(p + 3 * H, psx2[k]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Edge backward of layer l (math: see EdgeBwdArgs).  FUSE_WG: the workgroup also forms its share of the edge
// weight gradient det^T e_in (one H x H + H slab per workgroup, a.wg_out) and det never goes to memory; otherwise det
// is written to a.det_out for a separate product kernel.
// ------------------------------------------------------------------------------------------------
template <int MT, bool FUSE_WG>
__global__ __launch_bounds__(32 * MT, 2) void k_edge_bwd4(EdgeBwdArgs a) {
    using GM = Strip2Geom<MT>;
    constexpr int HP = GM::HP;
    __shared__ __attribute__((aligned(16))) float tdet[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float tpre[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float tein[FUSE_WG ? 2 : 1][FUSE_WG ? 16 : 1][GM::LS];
    __shared__ __attribute__((aligned(16))) float ce[6][HP];     // edges: mean, rstd, gamma, beta, m1, m2
    __shared__ __attribute__((aligned(16))) float cx[6][HP];     // nodes: same
    __shared__ __attribute__((aligned(16))) float ncl[4][HP];    // layer l-1 edges: mean, rstd, gamma, beta
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;        // rows arow and arow + 8
    const int H = a.H, V = a.V, C = a.C, IT = a.IT;
    const bool acol_ok = acol < H;
    const uint32_t rowb = (uint32_t)C * H * 4;                                   // bytes between rows j and j+1
    const uint32_t hb = (uint32_t)H * 4;                                         // bytes of one row
    const uint32_t voffA0 = acol_ok ? (uint32_t)arow * rowb + acol * 4 : kOob;
    const uint32_t voffA1 = acol_ok ? (uint32_t)(arow + 8) * rowb + acol * 4 : kOob;
    const uint32_t voffNA = acol_ok ? (uint32_t)acol * 4 : kOob;                 // phase-A lane offset in one node row
    int bcol[2];
    bool bcol_ok[2];
    uint32_t voffB[2], voffL[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        bcol[k] = 16 * (w + k * GM::NW) + 4 * q;
        bcol_ok[k] = bcol[k] < H;
        voffB[k] = bcol_ok[k] ? (uint32_t)rs * rowb + bcol[k] * 4 : kOob;
        voffL[k] = (rs == 0 && bcol_ok[k]) ? (uint32_t)bcol[k] * 4 : kOob;
    }
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
    f32x4 na1[2] = {f4zero(), f4zero()}, na2[2] = {f4zero(), f4zero()};
    const int njt = (V + 15) / 16;
    const int nit = (V + IT - 1) / IT;
    const int units = a.B * nit * njt * C;
    const uint32_t Rn = (uint32_t)a.B * V * C;
    const size_t slab = (size_t)H * H + H;
    __syncthreads();
    // contiguous range of units ((b, it, jt, c), c fastest) per workgroup
    const int per = (units + (int)gridDim.x - 1) / (int)gridDim.x;
    const int u0 = (int)blockIdx.x * per, u1 = min(units, u0 + per);
    if (u0 >= u1) {                                    // uniform: no units for this workgroup
        if (do_next) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (rs == 0 && bcol_ok[k]) {
                    st4(a.next.partial + (size_t)blockIdx.x * 2 * H + bcol[k], f4zero());
                    st4(a.next.partial + (size_t)blockIdx.x * 2 * H + H + bcol[k], f4zero());
                }
        }
        if constexpr (FUSE_WG) {
            float* out = a.wg_out + (size_t)blockIdx.x * slab;
            for (int k = t; k < (int)slab; k += GM::NT) out[k] = 0.f;
        }
        return;
    }
    auto rowp = [&](const float* base, uint32_t row) { return (const float*)((const char*)base + (uint64_t)row * hb); };
    auto rowpw = [&](float* base, uint32_t row) { return (float*)((char*)base + (uint64_t)row * hb); };
    auto tile_bytes = [&](int jt_) { return (uint32_t)(min(16, V - jt_ * 16) - 1) * rowb + hb; };
    int u = u0, buf = 0;
    int c, jt, it, b;
    {
        int tt = u;
        c = tt % C; tt /= C;
        jt = tt % njt; tt /= njt;
        it = tt % nit;
        b = tt / nit;
    }
    f32x4 r_dn[2], r_tp[2], r_ei[2] = {f4zero(), f4zero()}, r_dxn, r_xt, r_Dv, r_ai, vnj[2];
    auto load_raw = [&](int b_, int c_, int jt_, int i_) {
        const uint32_t bVi = (uint32_t)b_ * V + i_;
        const uint32_t nrow_i = bVi * C + c_;
        const uint32_t erow0 = (bVi * V + jt_ * 16) * C + c_;
        const uint32_t nb = tile_bytes(jt_);
        const rsrc_t rd = make_rsrc(rowp(a.de, erow0), nb), rt = make_rsrc(rowp(a.e_tmp, erow0), nb);
        r_dn[0] = bld4(rd, voffA0); r_dn[1] = bld4(rd, voffA1);
        r_tp[0] = bld4(rt, voffA0); r_tp[1] = bld4(rt, voffA1);
        if constexpr (FUSE_WG) {
            const rsrc_t re = make_rsrc(rowp(a.e_in, erow0), nb);      // zeros beyond the tile (range check)
            r_ei[0] = bld4(re, voffA0); r_ei[1] = bld4(re, voffA1);
        }
        r_dxn = bld4(make_rsrc(rowp(a.dx, nrow_i), hb), voffNA);
        r_xt = bld4(make_rsrc(rowp(a.x_tmp, nrow_i), hb), voffNA);
        r_Dv = bld4(make_rsrc(rowp(a.Dn, nrow_i), hb), voffNA);
        r_ai = bld4(make_rsrc(rowp(a.Aagg, nrow_i), hb), voffNA);
    };
    auto load_vnj = [&](int b_, int c_, int jt_) {
        const rsrc_t r = make_rsrc(rowp(a.Vn, ((uint32_t)b_ * V + jt_ * 16) * C + c_), tile_bytes(jt_));
        vnj[0] = bld4(r, voffA0); vnj[1] = bld4(r, voffA1);
    };
    int i = it * IT, i_end = min(V, (it + 1) * IT);
    // first step's operands before every other prologue load (see k_edge_fwd3)
    load_raw(b, c, jt, i);
    load_vnj(b, c, jt);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 wf0[MT], wf1[MT];
    strip_wfrag_t<MT>(wf0, a.WUt, H, w, rs, q);          // a.WUt: edge_feat.U weight as stored
    strip_wfrag_t<MT>(wf1, a.WUt, H, w + GM::NW, rs, q);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 cve[2] = {f4zero(), f4zero()}, cvn[2] = {f4zero(), f4zero()};
    // this workgroup's share of d edge_feat.U.weight: wavefront w holds rows 16 cb .. 16 cb + 15 of its two column
    // blocks (MT accumulators of 16 x 16 each) and the matching entries of the bias gradient
    f32x4 wacc[FUSE_WG ? 2 : 1][FUSE_WG ? MT : 1];
#pragma unroll
    for (int k = 0; k < (FUSE_WG ? 2 : 1); ++k)
#pragma unroll
        for (int m = 0; m < (FUSE_WG ? MT : 1); ++m) wacc[k][m] = f4zero();
    float bsum[2] = {0.f, 0.f};
    while (true) {
        const uint32_t bVi = (uint32_t)b * V + i;
        const uint32_t nrow_i = bVi * C + c;
        const uint32_t erow0 = (bVi * V + jt * 16) * C + c;
        const uint32_t nbytes = tile_bytes(jt);
        // ---- phase A: rows arow and arow + 8 of the tile, column group ag
        {
            const f32x4 xmu = ld4(&cx[0][acol]), xrs = ld4(&cx[1][acol]), xga = ld4(&cx[2][acol]);
            const f32x4 xbe = ld4(&cx[3][acol]), xm1 = ld4(&cx[4][acol]), xm2 = ld4(&cx[5][acol]);
            const f32x4 emu = ld4(&ce[0][acol]), ers = ld4(&ce[1][acol]), ega = ld4(&ce[2][acol]);
            const f32x4 ebe = ld4(&ce[3][acol]), em1 = ld4(&ce[4][acol]), em2 = ld4(&ce[5][acol]);
            f32x4 dmx = f32x4{1.f, 1.f, 1.f, 1.f};
            if (dcx.on) dmx = drop_mul4(dcx, drop_rowhash(dcx, nrow_i), ag);
            // node-side BatchNorm backward of row (b, i, c), once for both rows: qi = dx_tmp / D
            f32x4 qi, aiv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xhx = (r_xt[r] - xmu[r]) * xrs[r];
                const float bnx = xhx * xga[r] + xbe[r];
                const float dxd = r_dxn[r] * dmx[r];
                const float dbnx = bnx > 0.f ? dxd : 0.f;
                const float dxt = xga[r] * xrs[r] * (dbnx - xm1[r] - xhx * xm2[r]);
                qi[r] = mean_agg ? dxt * __builtin_amdgcn_rcpf(r_Dv[r]) : dxt;
                aiv[r] = mean_agg ? r_ai[r] : 0.f;
            }
            f32x4 det[2], dpre[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const bool valid_a = jt * 16 + arow + 8 * rr < V && acol_ok;
                f32x4 dme = f32x4{1.f, 1.f, 1.f, 1.f};
                if (dce.on) dme = drop_mul4(dce, drop_rowhash(dce, erow0 + (uint32_t)(arow + 8 * rr) * C), ag);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = r_tp[rr][r];
                    const float xh = (x - emu[r]) * ers[r];
                    const float bn = xh * ega[r] + ebe[r];
                    const float d = r_dn[rr][r] * dme[r];
                    const float dbn = bn > 0.f ? d : 0.f;
                    const float g = sigmoidf_(x);
                    const float v = ega[r] * ers[r] * (dbn - em1[r] - xh * em2[r]) + qi[r] * (vnj[rr][r] - aiv[r]) * g * (1.f - g);
                    det[rr][r] = valid_a ? v : 0.f;
                    dpre[rr][r] = valid_a ? d : 0.f;
                    cve[rr][r] += det[rr][r];
                    cvn[rr][r] += valid_a ? qi[r] * g : 0.f;
                }
            }
            if constexpr (!FUSE_WG) {
                const rsrc_t ro = make_rsrc(rowpw(a.det_out, erow0), nbytes);
                bst4(ro, voffA0, det[0]); bst4(ro, voffA1, det[1]);
            }
            st4(&tdet[buf][arow][acol], det[0]); st4(&tdet[buf][arow + 8][acol], det[1]);
            st4(&tpre[buf][arow][acol], dpre[0]); st4(&tpre[buf][arow + 8][acol], dpre[1]);
            if constexpr (FUSE_WG) { st4(&tein[buf][arow][acol], r_ei[0]); st4(&tein[buf][arow + 8][acol], r_ei[1]); }
        }
        // ---- next step's coordinates
        const bool unit_end = i + 1 >= i_end;
        int nu = u, ni = i + 1, nc = c, njt2 = jt, nit2 = it, nb = b, ni_end = i_end;
        if (unit_end) {
            nu = u + 1; nc = c + 1;
            if (nc == C) {
                nc = 0; njt2 = jt + 1;
                if (njt2 == njt) { njt2 = 0; nit2 = it + 1; if (nit2 == nit) { nit2 = 0; nb = b + 1; } }
            }
            ni = nit2 * IT; ni_end = min(V, (nit2 + 1) * IT);
        }
        const bool more = nu < u1;
        if (!more) { nu = u; ni = i; nc = c; njt2 = jt; nit2 = it; nb = b; ni_end = i_end; }   // last step: reload something valid
        // ---- operand of this step's statistics, the finished unit's column sums, the next step's phase-A operands
        const bool rvalid = jt * 16 + rs < V;
        f32x4 ntp[2];
        {
            const rsrc_t rn = make_rsrc(rowp(ntmp, erow0), nbytes);
            ntp[0] = bld4(rn, voffB[0]); ntp[1] = bld4(rn, voffB[1]);
        }
        {
            // dropped by the range check (zero records) while the unit continues
            const uint32_t colrow = (uint32_t)it * Rn + ((uint32_t)b * V + jt * 16) * C + c;
            const uint32_t nbu = unit_end ? nbytes : 0u;
            const rsrc_t r1 = make_rsrc(rowpw(a.colVe, colrow), nbu), r2 = make_rsrc(rowpw(a.colVn, colrow), nbu);
            bst4(r1, voffA0, cve[0]); bst4(r1, voffA1, cve[1]);
            bst4(r2, voffA0, cvn[0]); bst4(r2, voffA1, cvn[1]);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int r = 0; r < 4; ++r) { cve[rr][r] = unit_end ? 0.f : cve[rr][r]; cvn[rr][r] = unit_end ? 0.f : cvn[rr][r]; }
        }
        load_raw(nb, nc, njt2, ni);
        load_vnj(nb, nc, njt2);
        __syncthreads();
        // ---- phase B: both column blocks of this wavefront
        f32x4 din[2];
        strip_gemm2<MT>(tdet[buf], wf0, wf1, rs, q, din[0], din[1]);
        {
            const rsrc_t rde = make_rsrc(rowpw(a.de, erow0), nbytes);
            const rsrc_t rrp = make_rsrc(rowpw(a.rowpart, (uint32_t)jt * Rn + nrow_i), hb);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int bc = bcol_ok[k] ? bcol[k] : 0;                 // in-range LDS column for lanes beyond H (values unused)
                din[k] += ld4(&tpre[buf][rs][bc]);
                f32x4 rsum = ld4(&tdet[buf][rs][bc]);
                bst4(rde, voffB[k], din[k]);
                const f32x4 nmu = ld4(&ncl[0][bc]), nrs = ld4(&ncl[1][bc]), nga = ld4(&ncl[2][bc]), nbe = ld4(&ncl[3][bc]);
                f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                if (ndc.on) dm = drop_mul4(ndc, drop_rowhash(ndc, erow0 + (uint32_t)rs * C), bc >> 2);
                const bool valid_b = rvalid && bcol_ok[k];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (ntp[k][r] - nmu[r]) * nrs[r];
                    const float bn = xh * nga[r] + nbe[r];
                    const float d = (bn > 0.f && valid_b) ? din[k][r] * dm[r] : 0.f;
                    na1[k][r] += d;
                    na2[k][r] += d * xh;
                }
                rsum = red16x4(rsum);      // invalid rows hold zeros in the tile
                bst4(rrp, voffL[k], rsum);
            }
        }
        if constexpr (FUSE_WG) {
            // d W[n][kc] += sum over the tile's rows of det[row][n] * e_in[row][kc] (A[i = n][kk = row], B[kk = row][j = kc]);
            // the e_in operand is read once for both blocks; rows beyond the graph hold zeros in the det tile
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float av0 = tdet[buf][4 * st + q][16 * w + rs];
                const float av1 = tdet[buf][4 * st + q][16 * (w + GM::NW) + rs];
                bsum[0] += av0; bsum[1] += av1;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float bv = tein[buf][4 * st + q][16 * m + rs];
                    wacc[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv, wacc[0][m], 0, 0, 0);
                    wacc[1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv, wacc[1][m], 0, 0, 0);
                }
            }
        }
        buf ^= 1;
        if (!more) break;
        u = nu; i = ni; b = nb; c = nc; jt = njt2; it = nit2; i_end = ni_end;
    }
    if (do_next) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            na1[k] = red16x4(na1[k]); na2[k] = red16x4(na2[k]);
            if (rs == 0 && bcol_ok[k]) {
                st4(a.next.partial + (size_t)blockIdx.x * 2 * H + bcol[k], na1[k]);
                st4(a.next.partial + (size_t)blockIdx.x * 2 * H + H + bcol[k], na2[k]);
            }
        }
    }
    if constexpr (FUSE_WG) {
        float* out = a.wg_out + (size_t)blockIdx.x * slab;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int cb = w + k * GM::NW;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nn = 16 * cb + 4 * q + r, kc = 16 * m + rs;
                    if (nn < H && kc < H) out[(size_t)nn * H + kc] = wacc[k][m][r];
                }
            const float bs = redq(bsum[k]);
            const int n = 16 * cb + rs;
            if (q == 0 && n < H) out[(size_t)H * H + n] = bs;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Head forward (edge MLP with at most one hidden layer, loss, edge error) with two column blocks per wavefront: the same
// outputs, argument list and per-workgroup partial rows as k_head_fwd3 of kernels_strip.hpp, MT / 2 wavefronts per
// workgroup.  A thread stages rows arow and arow + 8 of its float4 column group; wavefront w multiplies the tile by the
// weight rows of column blocks w and w + MT / 2 from one LDS read of the operand and finishes eight outputs per lane,
// so the per-tile bookkeeping (coordinates, barriers, row reductions, the loss of 16 rows) is paid by half as many
// wavefronts: 3.16 -> 2.81 ms at 200 nodes x 100 commodities on a grid of six workgroups per CU (three wavefronts per
// SIMD, what its 163 registers allow; larger grids leave a second round of workgroups).  The backward kernel was built
// the same way and is not here: with both products' weight rows, the fused weight-gradient accumulators and the sums
// for three outputs per column it needs more than 256 registers per lane (one wavefront per SIMD, 8.9 ms against the
// 7.2 ms of k_head_bwd3; capped at 256 it spills and is slower still).
// ------------------------------------------------------------------------------------------------
template <int MT, int NHID>
__global__ __launch_bounds__(32 * MT) void k_head_fwd4(
    size_t rows, int V, int C, int H, int voc, HeadParams hp, EdgeSrc src, const float* __restrict__ cap, int mask_invalid,
    const long long* __restrict__ y_edges, const float* __restrict__ edge_cw, float* __restrict__ logits,
    float* __restrict__ loss_partial, int* __restrict__ counts, int count_reps, const unsigned long long* __restrict__ rng,
    int stream_id0, float p, int training) {
    static_assert(NHID <= 1, "at most one hidden layer");
    using GM = Strip2Geom<MT>;
    __shared__ __attribute__((aligned(16))) float tile[2][16][GM::LS];
    __shared__ __attribute__((aligned(16))) float cl[CL_ROWS][GM::HP];
    __shared__ float red[2][GM::NW][16][2];
    const int t = threadIdx.x, lane = t & 63, rs = lane & 15, q = lane >> 4, w = t >> 6;
    const int arow = t / GM::G, ag = t - arow * GM::G, acol = 4 * ag;        // rows arow and arow + 8
    const bool acol_ok = acol < H;
    int bcol[2];
    bool bcol_ok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { bcol[k] = 16 * (w + k * GM::NW) + 4 * q; bcol_ok[k] = bcol[k] < H; }
    edge_src_fill_consts<MT>(cl, src, H);
    const DropCfg sdc = drop_cfg(rng, src.drop_sid, src.drop_p, src.drop_on && src.mode == 1);
    const DropCfg hdc = drop_cfg(rng, stream_id0, p, training);
    f32x4 wf0[MT], wf1[MT];
    if (NHID) { strip_wfrag<MT>(wf0, hp.hidW[0], H, w, rs, q); strip_wfrag<MT>(wf1, hp.hidW[0], H, w + GM::NW, rs, q); }
    f32x4 b1[2], w0[2], w1[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        b1[k] = (NHID && bcol_ok[k]) ? ld4(hp.hidB[0] + bcol[k]) : f4zero();
        w0[k] = bcol_ok[k] ? ld4(hp.outW + bcol[k]) : f4zero();
        w1[k] = (bcol_ok[k] && voc > 1) ? ld4(hp.outW + H + bcol[k]) : f4zero();
    }
    const float b2_0 = hp.outB[0], b2_1 = voc > 1 ? hp.outB[1] : 0.f;
    const float cw0 = edge_cw ? edge_cw[0] : 0.f, cw1 = edge_cw ? edge_cw[1] : 0.f;
    const size_t ntiles = (rows + 15) / 16;
    const int BC_stride = V * V * C;
    float acc_wn = 0.f, acc_w = 0.f;
    int buf = 0;
    __syncthreads();
    HeadRaw raw0 = head_load(src, (size_t)blockIdx.x * 16 + arow, rows, H, C, acol, acol_ok);
    HeadRaw raw1 = head_load(src, (size_t)blockIdx.x * 16 + arow + 8, rows, H, C, acol, acol_ok);
    for (size_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        // operands of this tile's epilogue first, then the next tile's input (vector-memory operations retire in issue order)
        const size_t row = tl * 16 + rs;
        const size_t rowc = row < rows ? row : rows - 1;
        const float capv = mask_invalid ? cap[rowc / C] : 1.f;
        const int yv = y_edges != nullptr ? (int)y_edges[rowc] : 0;
        {
            const size_t rowa = tl * 16 + arow;
            st4(&tile[buf][arow][acol], head_finish<MT>(src, raw0, cl, sdc, rowa, rowa < rows && acol_ok, acol, ag));
            st4(&tile[buf][arow + 8][acol], head_finish<MT>(src, raw1, cl, sdc, rowa + 8, rowa + 8 < rows && acol_ok, acol, ag));
            if (tl + gridDim.x < ntiles) {
                raw0 = head_load(src, (tl + gridDim.x) * 16 + arow, rows, H, C, acol, acol_ok);
                raw1 = head_load(src, (tl + gridDim.x) * 16 + arow + 8, rows, H, C, acol, acol_ok);
            }
        }
        __syncthreads();
        f32x4 h[2];
        if (NHID) {
            strip_gemm2<MT>(tile[buf], wf0, wf1, rs, q, h[0], h[1]);
            const uint32_t rh = hdc.on ? drop_rowhash(hdc, (uint32_t)row) : 0u;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                h[k] += b1[k];
#pragma unroll
                for (int r = 0; r < 4; ++r) h[k][r] = h[k][r] > 0.f ? h[k][r] : 0.f;
                if (hdc.on) h[k] *= drop_mul4(hdc, rh, bcol[k] >> 2);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) h[k] = ld4(&tile[buf][rs][bcol_ok[k] ? bcol[k] : 0]);
        }
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) { p0 += h[k][r] * w0[k][r]; p1 += h[k][r] * w1[k][r]; }
        p0 = redq(p0); p1 = redq(p1);
        if (q == 0) { red[buf][w][rs][0] = p0; red[buf][w][rs][1] = p1; }
        __syncthreads();
        if (w == 0 && q == 0 && row < rows) {
            float l0 = b2_0, l1 = b2_1;
#pragma unroll
            for (int k = 0; k < GM::NW; ++k) { l0 += red[buf][k][rs][0]; l1 += red[buf][k][rs][1]; }
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
