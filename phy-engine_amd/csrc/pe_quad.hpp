// pe_quad.hpp -- lane-group ("quad") numeric factorisation of the WAVE fronts of the split schedule.
//
// Replaces, for the small fronts below the cooperative part of the assembly tree, the per-instance path of pe_front.hpp
// (front_factor<WaveTeam>) -- i.e. the same slice of Eigen's SparseLU compute()+solve() (circuits/circuit.h:1516-1518) -- with an
// execution that shares ALL index and control work between instances, the lever the round-2 counters pointed at (every one of the
// 1 024 instances of a sweep walks the same elimination tree):
//
//   * one wavefront = FOUR instances x one list of wave fronts; lane = 16 q + r: instance q of the quad, row r of a row set;
//   * the whole front lives in REGISTERS: a[s][c] = entry (row r + 16 s, column c) -- fronts of order <= 32 (two row sets), <= 16
//     pivots (all in row set 0), the right-hand side of the fused forward substitution as one more column g[s];
//   * every index is wavefront-uniform (scalar loads from the quad plan of pe_symbolic.cpp): which columns a child contributes to,
//     where a column of the factor / of the update matrix goes; the only per-lane index data are one byte per own entry of A and
//     one byte per child and row set;
//   * the pivot row reaches the 16 lanes of its instance by DPP row_newbcast (v_mov_b64_dpp: one instruction per column for all
//     four instances, no LDS, no v_readlane, no SGPR round trip); rows at or above the pivot take a zero multiplier;
//   * no LDS image of the front: every load of a front is independent of every other (one memory round trip per batch instead of
//     index -> value -> LDS chains), lanes without a contribution read a zero region parked behind each instance's arena instead
//     of branching;
//   * (behind PE_QUAD_LDS_STACK, off: the update matrix of a front whose parent follows in the same list can stay on a small LDS
//     stack of the wavefront -- measured slower, see the switch below).
//
// Output layout = front_factor's: U11 (with the scaled multipliers below its diagonal) + U12 in the factor store, L21 only when a
// later launch reuses the factors (V.keep_l21), update matrix + update vector in the arena slot, forward-substituted pivots in w.
//
// Written against an execution model X so that tests/emu can run the SAME code on the host with 64-wide vector types
// (test infrastructure; the product instantiates only the device model of pe_kernels.hip):
//   X::vd / vi / vu / vm   double / int / unsigned / predicate of one lane (device) or of all 64 lanes (emulation)
//   X::lane()              lane index 0..63
//   X::bcast(v, k)         value of lane 16 (lane / 16) + k, k a compile-time constant after unrolling
//   X::ld / ld_u32x4 / ld_i32 (uniform base, per-lane byte offset), X::st(base, offset, value) inside X::when(mask, body)
//   X::lds_ld(byte address) / lds_st(byte address, value) inside X::when / lds_fence(): the wavefront's LDS (update-matrix stack)
//   X::st_if(all, mask, base, offset, value): inside X::when -- store for every lane of the region if `all` (uniform), else where mask
//   X::fence(): the wavefront's earlier global stores are visible to its later loads (other lanes included)
//   X::sel(mask, a, b), X::rcp(d), X::fma(a, b, c), X::bad(piv), X::flag(ptr, index, bits, mask)
#pragma once
#include "pe_device.hpp"

// Measured and NOT kept (round 3, profiles/README.md): parking the update matrix of a front whose parent follows in the same list on
// an LDS stack of the wavefront instead of the arena.  Only 35 % of the wave fronts' update-matrix doubles qualify on the M10k tree
// (the others belong to subtree roots under cooperative parents), and the second copy of the child loop costs registers: 441 spilled
// VGPRs at two wavefronts per SIMD, 3.1 ms per launch against 1.75.  The code stays behind this switch as the A/B's other arm.
#ifndef PE_QUAD_LDS_STACK
    #define PE_QUAD_LDS_STACK 0
#endif
// columns per wavefront-uniform skip test of the child loop (4 = one word of the child's column bytes, 8 = round 3's first version)
#ifndef PE_QUAD_BACK_LAZY_FENCE
    #define PE_QUAD_BACK_LAZY_FENCE 1
#endif
#ifndef PE_QUAD_ELIM_GUARD
    #define PE_QUAD_ELIM_GUARD 1
#endif
#ifndef PE_QUAD_COLS
    #define PE_QUAD_COLS 8
#endif
#if defined(__HIPCC__)
    #define PEQ_DEV __device__ __forceinline__
#else
    #define PEQ_DEV inline
#endif

namespace pe
{
    template <class X>
    struct QuadCtx
    {
        char const* baseA;             // aval / factor / arena / w of the quad's FIRST instance; the others sit at 32-bit byte offsets from it
        char* baseF;
        char* baseR;
        char* baseW;
        typename X::vu offA, offF, offR, offW;
        typename X::vu ldsq, ldsz;     // LDS byte address of this lane's instance stack, and of the stack's zero slot
        typename X::vm valid;          // the lane's instance exists and is active (stores only)
        typename X::vi r;              // lane & 15
    };

    // this lane's index data of one front (pe_symbolic.hpp, q_lane): per row set the 16 RS table bytes of its row, then 16 child bytes
    template <class X>
    struct QuadLaneIdx
    {
        typename X::vu w[2][12];
    };
    template <class X>
    PEQ_DEV void quad_lane_idx(QuadLaneIdx<X>& L, unsigned char const* base, int rs, typename X::vi r)
    {
        // (requested one front ahead: the loads fly behind the current front's arithmetic)
        if(rs == 1)
        {
            typename X::vu const off = X::to_u(r * 32);
            X::ld_u32x4(base, off, &L.w[0][0]);
            X::ld_u32x4(base, off + 16u, &L.w[0][4]);
#pragma unroll
            for(int j = 8; j < 12; ++j) L.w[0][j] = typename X::vu(0u);
#pragma unroll
            for(int j = 0; j < 12; ++j) L.w[1][j] = typename X::vu(0u);
        }
        else
        {
#pragma unroll
            for(int s2 = 0; s2 < 2; ++s2)
            {
                typename X::vu const off = X::to_u((r + 16 * s2) * 48);
#pragma unroll
                for(int j = 0; j < 3; ++j) X::ld_u32x4(base, off + 16u * j, &L.w[s2][4 * j]);
            }
        }
    }

    // One wave front of four instances.  RS = row sets of 16 rows; columns 0 .. 16 RS - 1.  `blk`: the front's block of the list's
    // program, `L`: this lane's index data.  Returns the lanes that met a bad pivot.
    template <class X, int RS>
    PEQ_DEV typename X::vm quad_front(DevView const& V, QuadCtx<X> const& cx, int const* blk, QuadLaneIdx<X> const& L, bool clk, long long (&clkv)[6])
    {
        long long const ck0 = clk ? X::clock() : 0;
        using vd = typename X::vd;
        using vi = typename X::vi;
        using vu = typename X::vu;
        using vm = typename X::vm;
        constexpr int M = 16 * RS;
        int const m = blk[0], c0 = blk[2], e0 = blk[3], sl = blk[13];
        // (developer timing knobs, PHY_ENGINE_HIP_QUAD bits 1..4: skip the elimination / the stores / the children / the own entries
        //  -- wrong results)
        int const p = (V.quad & 2) ? 0 : blk[1], nch = (V.quad & 8) ? 0 : blk[4];
        int const u = m - p;
        long long const lptr = static_cast<long long>(static_cast<unsigned>(blk[6])) | (static_cast<long long>(blk[7]) << 32);
        long long const sptr = static_cast<long long>(static_cast<unsigned>(blk[8])) | (static_cast<long long>(blk[9]) << 32);
        vi const r = cx.r;
        long long const ck1 = clk ? X::clock() : 0;  // (the header's scalar loads have landed: m is used above)

        vd a[RS][M];
        vd g[RS];
        // ---- own entries of A.  Loads are UNCONDITIONAL inside a group of 8 columns (a lane without an entry re-reads the front's
        // first entry and drops it): no branch per cell, so the loads of a group -- and, registers permitting, of the next groups --
        // are in flight together.  Row set 1 holds rows >= 16 >= p: only its pivot COLUMNS can carry entries of A.
        // (addresses: one wavefront-uniform 64-bit base per array for the whole kernel + a 32-bit offset per lane -- the uniform part
        //  of an offset is added on the vector side, one v_add per access, instead of a 64-bit scalar pointer per column)
        {
            vu const oa = cx.offA + static_cast<unsigned>(e0 - 1) * 8u;  // table byte k >= 1 -> entry e0 + k - 1
#pragma unroll
            for(int s2 = 0; s2 < RS; ++s2)
            {
#pragma unroll
                for(int C0 = 0; C0 < M; C0 += 8)
                {
                    if(C0 < (s2 == 0 ? m : p) && !(V.quad & 16))
                    {
#pragma unroll
                        for(int C = C0; C < C0 + 8; ++C)
                        {
                            vu const k = (L.w[s2][C >> 2] >> (8 * (C & 3))) & 255u;
                            vm const has = k != 0u;
                            vd const v = X::ld(cx.baseA, oa + (X::sel(has, k, vu(1u)) << 3));
                            a[s2][C] = X::sel(has, v, vd(0.0));
                        }
                    }
                    else
                    {
#pragma unroll
                        for(int C = C0; C < C0 + 8; ++C) a[s2][C] = vd(0.0);
                    }
                }
            }
            // right-hand side of the fused forward substitution: pivot rows from w (children's update vectors below)
            vm const piv_row = r < p;
            g[0] = X::sel(piv_row, X::ld(cx.baseW, cx.offW + static_cast<unsigned>(c0) * 8u + (X::to_u(X::sel(piv_row, r, vi(0))) << 3)), vd(0.0));
            if constexpr(RS > 1) g[1] = vd(0.0);
        }
        // ---- children: column C of the front takes column cj(C) of the child's update matrix (uniform), this lane its row ci (a byte
        // per child and row set).
        //  * child in the ARENA (global memory): no branch per column -- a column the child does not touch, and a lane whose row it does
        //    not touch, read the zero region behind the arena: every load of every child is independent of all the others;
        //  * child on the wavefront's LDS stack (its parent follows closely in the same list): plain LDS reads, untouched columns are
        //    skipped (a wavefront-uniform branch: LDS latency needs no batching), untouched rows read the stack's zero slot.
        vu const zoff = cx.offR + static_cast<unsigned>(V.q_zero_off * 8);
#pragma unroll
        for(int grp = 0; grp < 4; ++grp)  // (the child bytes of a row: four to a register)
        {
            if(4 * grp < nch)
            {
                int const e1 = nch < 4 * grp + 4 ? nch : 4 * grp + 4;
                for(int e = 4 * grp; e < e1; ++e)
                {
                    int const* cb = blk + 16 + e * 32;
                    int const sp = cb[0], uc = cb[1], csl = cb[18];
                    unsigned cm[M / 4];
#pragma unroll
                    for(int j = 0; j < M / 4; ++j) cm[j] = static_cast<unsigned>(cb[2 + j]);
                    unsigned const ucb = static_cast<unsigned>(uc) * 8u;
#if PE_QUAD_LDS_STACK
                    if(csl >= 0)
                    {
                        vu lrow[RS];
                        vm hasr[RS];
#pragma unroll
                        for(int s2 = 0; s2 < RS; ++s2)
                        {
                            vu const ci = (L.w[s2][M / 4 + grp] >> (8 * (e & 3))) & 255u;
                            hasr[s2] = ci != 0u;
                            lrow[s2] = cx.ldsq + ((ci + static_cast<unsigned>(csl - 1)) << 3);
                        }
#pragma unroll
                        for(int C = 0; C < M; ++C)
                        {
                            unsigned const cj = (cm[C >> 2] >> (8 * (C & 3))) & 255u;
                            if(cj)
                            {
                                unsigned const shift = (cj - 1u) * ucb;
#pragma unroll
                                for(int s2 = 0; s2 < RS; ++s2) a[s2][C] = a[s2][C] + X::lds_ld(X::sel(hasr[s2], lrow[s2] + shift, cx.ldsz));
                            }
                        }
                        unsigned const vshift = static_cast<unsigned>(uc) * ucb;
#pragma unroll
                        for(int s2 = 0; s2 < RS; ++s2) g[s2] = g[s2] + X::lds_ld(X::sel(hasr[s2], lrow[s2] + vshift, cx.ldsz));
                    }
                    else
#else
                    (void)csl;
#endif
                    {
                        vu voff[RS];
#pragma unroll
                        for(int s2 = 0; s2 < RS; ++s2)
                        {
                            vu const ci = (L.w[s2][M / 4 + grp] >> (8 * (e & 3))) & 255u;
                            voff[s2] = cx.offR + (X::sel(ci != 0u, ci + static_cast<unsigned>(sp - 1), vu(static_cast<unsigned>(V.q_zero_off))) << 3);
                        }
                        // (one wavefront-uniform skip per group of PE_QUAD_COLS columns the child does not touch at all.  Measured and not
                        //  kept, profiles/r03_ab_runs.log ab8 / ab9: a second skip per row set none of the child's rows lands in -- the
                        //  condition inside the column loop costs the launch pair +9 %: every branch ends a batch of loads in flight)
#pragma unroll
                        for(int C0 = 0; C0 < M; C0 += PE_QUAD_COLS)
                        {
                            unsigned touched = 0u;
#pragma unroll
                            for(int j = 0; j < PE_QUAD_COLS / 4; ++j)
                                if((C0 >> 2) + j < M / 4) touched |= cm[(C0 >> 2) + j];
                            if(touched != 0u)
                            {
#pragma unroll
                                for(int C = C0; C < C0 + PE_QUAD_COLS && C < M; ++C)
                                {
                                    unsigned const cj = (cm[C >> 2] >> (8 * (C & 3))) & 255u;
                                    unsigned const shift = (cj ? cj - 1u : 0u) * ucb;
#pragma unroll
                                    for(int s2 = 0; s2 < RS; ++s2) a[s2][C] = a[s2][C] + X::ld(cx.baseR, (cj ? voff[s2] : zoff) + shift);
                                }
                            }
                        }
                        unsigned const vshift = static_cast<unsigned>(uc) * ucb;  // the child's update vector sits behind its update matrix
#pragma unroll
                        for(int s2 = 0; s2 < RS; ++s2) g[s2] = g[s2] + X::ld(cx.baseR, voff[s2] + vshift);
                    }
                }
            }
        }
        long long const ck2 = clk ? X::clock(g[0]) : 0;  // (assembled: the clock is read behind the last add of the right-hand side)
        // ---- right-looking elimination of the p pivots (rows of set 0): same operation order as block_step_t of pe_kernels.hip
        vm bad = X::none();
#pragma unroll
        for(int kk = 0; kk < 16; ++kk)
        {
            // (guards instead of early exits: a `break` would turn the constant trip count into min(p, 16) and the loop would no
            //  longer unroll -- the register arrays must be indexed by constants)
            if(kk < p)
            {
                vd const piv = X::bcast(a[0][kk], kk);
                bad = bad | X::bad(piv);
                vd const rp = X::rcp(piv);
                vd lm[RS];
                {
                    vd const l = a[0][kk] * rp;
                    vm const below = r > kk;
                    lm[0] = X::sel(below, l, vd(0.0));
                    a[0][kk] = X::sel(below, l, a[0][kk]);
                }
                if constexpr(RS > 1)
                {
                    lm[1] = a[1][kk] * rp;  // rows 16.. are below every pivot (rows >= m hold zeros)
                    a[1][kk] = lm[1];
                }
#pragma unroll
                for(int C0 = 0; C0 < M; C0 += 4)
                {
                    // (C0 + 3 > kk folds at compile time; the run-time guard skips the groups of 4 columns beyond the front's order -- they
                    //  hold zeros and are never stored.  PE_QUAD_ELIM_GUARD 0: no guard, those FMAs run on the zeros)
                    if(C0 + 3 > kk && (!PE_QUAD_ELIM_GUARD || C0 < m))
                    {
#pragma unroll
                        for(int C = C0; C < C0 + 4; ++C)
                        {
                            if(C > kk)
                            {
                                vd const row = X::bcast(a[0][C], kk);
#pragma unroll
                                for(int s2 = 0; s2 < RS; ++s2) a[s2][C] = X::fma(-lm[s2], row, a[s2][C]);
                            }
                        }
                    }
                }
                vd const grow = X::bcast(g[0], kk);
#pragma unroll
                for(int s2 = 0; s2 < RS; ++s2) g[s2] = X::fma(-lm[s2], grow, g[s2]);
            }
        }
        long long const ck3 = clk ? X::clock(g[0]) : 0;
        // ---- stores (layout of front_factor): L panel m x p (ld m) -- its top p x p block always, the rows below only when a later
        // launch runs a separate forward pass --, U panel p x u (ld p) behind it, update matrix u x u (ld u) + update vector (arena
        // slot or LDS stack), w.  One exec region per group of rows, wavefront-uniform column guards inside.
        if(!(V.quad & 4))
        {
            unsigned const pf = static_cast<unsigned>(lptr) * 8u, mb = static_cast<unsigned>(m) * 8u, pb = static_cast<unsigned>(p) * 8u, ub = static_cast<unsigned>(u) * 8u;
            unsigned const pu = pf + static_cast<unsigned>(m) * pb;
            unsigned const ps = static_cast<unsigned>(sptr) * 8u;
            vu const fo = cx.offF + (X::to_u(r) << 3);
            X::when(cx.valid & (r < p),
                    [&]
                    {
#pragma unroll
                        for(int C = 0; C < M; ++C)
                        {
                            // (top block: without a separate forward pass nobody reads the multipliers below the diagonal of U11 --
                            //  the backward passes take rows <= column only -- so they are not written: 15 % of this kernel's bytes)
                            if(C < p) X::st_if(V.keep_l21 != 0 || C >= 15, r <= C, cx.baseF, fo + (pf + static_cast<unsigned>(C) * mb), a[0][C]);
                            else if(C < m)
                                X::st(cx.baseF, fo + (pu + static_cast<unsigned>(C - p) * pb), a[0][C]);
                        }
                        X::st(cx.baseW, cx.offW + static_cast<unsigned>(c0) * 8u + (X::to_u(r) << 3), g[0]);
                    });
#pragma unroll
            for(int s2 = 0; s2 < RS; ++s2)
            {
                vi const R = r + 16 * s2;
                vm const srow = (R >= p) & (R < m);
#if PE_QUAD_LDS_STACK
                if(sl >= 0)
                {
                    // (LDS stack: every lane of the quad writes, valid instance or not -- the parent reads whatever sits there)
                    vu const so = cx.ldsq + ((X::to_u(R - p) + static_cast<unsigned>(sl)) << 3);
                    X::when(srow,
                            [&]
                            {
#pragma unroll
                                for(int C = 0; C < M; ++C)
                                    if(C >= p && C < m) X::lds_st(so + static_cast<unsigned>(C - p) * ub, a[s2][C]);
                                X::lds_st(so + static_cast<unsigned>(u) * ub, g[s2]);
                            });
                }
                else
#else
                (void)sl;
#endif
                {
                    vu const so = cx.offR + (X::to_u(R - p) << 3) + ps;
                    X::when(cx.valid & srow,
                            [&]
                            {
#pragma unroll
                                for(int C = 0; C < M; ++C)
                                    if(C >= p && C < m) X::st(cx.baseR, so + static_cast<unsigned>(C - p) * ub, a[s2][C]);
                                X::st(cx.baseR, so + static_cast<unsigned>(u) * ub, g[s2]);
                            });
                }
                if(V.keep_l21)  // L21: the multipliers below the pivot block, read by a separate forward pass only
                {
                    vu const lo = cx.offF + (X::to_u(R) << 3) + pf;
                    X::when(cx.valid & srow,
                            [&]
                            {
#pragma unroll
                                for(int C = 0; C < 16; ++C)
                                    if(C < p) X::st(cx.baseF, lo + static_cast<unsigned>(C) * mb, a[s2][C]);
                            });
                }
            }
#if PE_QUAD_LDS_STACK
            if(sl >= 0) X::lds_fence();  // (the parent's lanes read what other lanes wrote)
#endif
        }
        if(clk)
        {
            long long const ck4 = X::clock();
            clkv[0] += ck1 - ck0;
            clkv[1] += ck2 - ck1;
            clkv[2] += ck3 - ck2;
            clkv[3] += ck4 - ck3;
            clkv[4] += ck4 - ck0;
            clkv[5] += 1;
        }
        return bad;
    }

    // =====================================================================================================================
    // MID fronts (order <= 64, f_kind 3): the same lanes-and-registers scheme with up to four row sets, columns in BLOCKS so that the
    // registers hold one block at a time: first columns 0..15 -- they hold every pivot column (p <= 16) -- with a right-looking
    // elimination inside the block, then columns 16.. in chunks of 8, LEFT-looking: a column's final values depend only on its own
    // initial values and on the multipliers, which by then sit in the first block's registers (a[s][kk], rows at or above pivot kk
    // zeroed).  Live: a[RS][16] + one chunk x[RS][8] -- 192 VGPRs at four row sets where the whole front would take 520.
    // =====================================================================================================================
    // this lane's index data of a MID front: the table bytes of its row in set 0 (every column), of its rows in sets 1.. (the first
    // 16 columns: only pivot columns carry own entries there), and the 16 child bytes of each of its rows
    template <class X>
    struct QuadLaneIdxM
    {
        typename X::vu t0[16], tp[3][4], ci[4][4];
    };
    template <class X>
    PEQ_DEV void quad_lane_idx_mid(QuadLaneIdxM<X>& L, unsigned char const* base, int rs, typename X::vi r)
    {
        using vu = typename X::vu;
        unsigned const rec = 16u * static_cast<unsigned>(rs) + 16u;
#pragma unroll
        for(int j = 0; j < 4; ++j)
        {
            if(j < rs) X::ld_u32x4(base, X::to_u(r) * rec + 16u * j, &L.t0[4 * j]);
            else
            {
#pragma unroll
                for(int k = 0; k < 4; ++k) L.t0[4 * j + k] = vu(0u);
            }
        }
        X::ld_u32x4(base, X::to_u(r) * rec + 16u * static_cast<unsigned>(rs), &L.ci[0][0]);
#pragma unroll
        for(int s2 = 1; s2 < 4; ++s2)
        {
            if(s2 < rs)
            {
                vu const off = X::to_u(r + 16 * s2) * rec;
                X::ld_u32x4(base, off, &L.tp[s2 - 1][0]);
                X::ld_u32x4(base, off + 16u * static_cast<unsigned>(rs), &L.ci[s2][0]);
            }
            else
            {
#pragma unroll
                for(int k = 0; k < 4; ++k)
                {
                    L.tp[s2 - 1][k] = vu(0u);
                    L.ci[s2][k] = vu(0u);
                }
            }
        }
    }

    // Columns [CB, CB + W) of a front, W = 16 (the first block) or 8: own entries of A + the children's contributions into x[s][C - CB];
    // RHS: also the right-hand-side column g[s].  Same branch-free loads as quad_front (zero region for untouched rows / columns).
    template <class X, int RS, int CB, int W, bool RHS>
    PEQ_DEV void quad_assemble(DevView const& V, QuadCtx<X> const& cx, int const* blk, QuadLaneIdxM<X> const& L, int m, int p, int nch, typename X::vd (&x)[RS][W],
                               typename X::vd* g)
    {
        using vd = typename X::vd;
        using vi = typename X::vi;
        using vu = typename X::vu;
        using vm = typename X::vm;
        int const c0 = blk[2], e0 = blk[3];
        vi const r = cx.r;
        vu const oa = cx.offA + static_cast<unsigned>(e0 - 1) * 8u;  // table byte k >= 1 -> entry e0 + k - 1
#pragma unroll
        for(int s2 = 0; s2 < RS; ++s2)
        {
#pragma unroll
            for(int G = 0; G < W; G += 8)
            {
                bool live = false;
                if constexpr(CB == 0) live = G < (s2 == 0 ? m : p);  // (row sets >= 1 hold rows >= 16 >= p: only their pivot columns carry entries)
                else
                    live = s2 == 0 && CB + G < m;
                if(live)
                {
#pragma unroll
                    for(int c = G; c < G + 8; ++c)
                    {
                        int const C = CB + c;
                        vu const word = s2 == 0 ? L.t0[C >> 2] : L.tp[s2 > 0 ? s2 - 1 : 0][(C >> 2) & 3];
                        vu const k = (word >> (8 * (C & 3))) & 255u;
                        vm const has = k != 0u;
                        vd const v = X::ld(cx.baseA, oa + (X::sel(has, k, vu(1u)) << 3));
                        x[s2][c] = X::sel(has, v, vd(0.0));
                    }
                }
                else
                {
#pragma unroll
                    for(int c = G; c < G + 8; ++c) x[s2][c] = vd(0.0);
                }
            }
        }
        if constexpr(RHS)
        {
            vm const piv_row = r < p;
            g[0] = X::sel(piv_row, X::ld(cx.baseW, cx.offW + static_cast<unsigned>(c0) * 8u + (X::to_u(X::sel(piv_row, r, vi(0))) << 3)), vd(0.0));
#pragma unroll
            for(int s2 = 1; s2 < RS; ++s2) g[s2] = vd(0.0);
        }
        vu const zoff = cx.offR + static_cast<unsigned>(V.q_zero_off * 8);
#pragma unroll
        for(int grp = 0; grp < 4; ++grp)  // (the child bytes of a row: four to a register)
        {
            if(4 * grp < nch)
            {
                int const e1 = nch < 4 * grp + 4 ? nch : 4 * grp + 4;
                for(int e = 4 * grp; e < e1; ++e)
                {
                    int const* cb = blk + 16 + e * 32;
                    int const sp = cb[0], uc = cb[1];
                    unsigned cm[W / 4];
#pragma unroll
                    for(int j = 0; j < W / 4; ++j) cm[j] = static_cast<unsigned>(cb[2 + CB / 4 + j]);
                    vu voff[RS];
#pragma unroll
                    for(int s2 = 0; s2 < RS; ++s2)
                    {
                        vu const ci = (L.ci[s2][grp] >> (8 * (e & 3))) & 255u;
                        voff[s2] = cx.offR + (X::sel(ci != 0u, ci + static_cast<unsigned>(sp - 1), vu(static_cast<unsigned>(V.q_zero_off))) << 3);
                    }
                    unsigned const ucb = static_cast<unsigned>(uc) * 8u;
#pragma unroll
                    for(int G = 0; G < W; G += 8)
                    {
                        if((cm[G >> 2] | cm[(G >> 2) + 1]) != 0u)  // (a group of 8 columns the child does not touch at all: skipped)
                        {
#pragma unroll
                            for(int c = G; c < G + 8; ++c)
                            {
                                unsigned const cj = (cm[c >> 2] >> (8 * (c & 3))) & 255u;
                                unsigned const shift = (cj ? cj - 1u : 0u) * ucb;
#pragma unroll
                                for(int s2 = 0; s2 < RS; ++s2) x[s2][c] = x[s2][c] + X::ld(cx.baseR, (cj ? voff[s2] : zoff) + shift);
                            }
                        }
                    }
                    if constexpr(RHS)
                    {
                        unsigned const vshift = static_cast<unsigned>(uc) * ucb;  // the child's update vector sits behind its update matrix
#pragma unroll
                        for(int s2 = 0; s2 < RS; ++s2) g[s2] = g[s2] + X::ld(cx.baseR, voff[s2] + vshift);
                    }
                }
            }
        }
    }

    // stores of columns [CB, CB + W) (layout of front_factor): L panel m x p (ld m) -- its top p x p block always, the rows below only
    // when a later launch runs a separate forward pass (V.keep_l21) --, U panel p x u (ld p) behind it, update matrix u x u (ld u).
    template <class X, int RS, int CB, int W>
    PEQ_DEV void quad_store(DevView const& V, QuadCtx<X> const& cx, int m, int p, long long lptr, long long sptr, typename X::vd const (&x)[RS][W])
    {
        using vi = typename X::vi;
        using vu = typename X::vu;
        int const u = m - p;
        vi const r = cx.r;
        unsigned const pf = static_cast<unsigned>(lptr) * 8u, mb = static_cast<unsigned>(m) * 8u, pb = static_cast<unsigned>(p) * 8u, ub = static_cast<unsigned>(u) * 8u;
        unsigned const pu = pf + static_cast<unsigned>(m) * pb;
        unsigned const ps = static_cast<unsigned>(sptr) * 8u;
        vu const fo = cx.offF + (X::to_u(r) << 3);
        X::when(cx.valid & (r < p),
                [&]
                {
#pragma unroll
                    for(int c = 0; c < W; ++c)
                    {
                        int const C = CB + c;
                        if(C < p) X::st(cx.baseF, fo + (pf + static_cast<unsigned>(C) * mb), x[0][c]);
                        else if(C < m)
                            X::st(cx.baseF, fo + (pu + static_cast<unsigned>(C - p) * pb), x[0][c]);
                    }
                });
#pragma unroll
        for(int s2 = 0; s2 < RS; ++s2)
        {
            vi const R = r + 16 * s2;
            vu const so = cx.offR + (X::to_u(R - p) << 3) + ps;
            X::when(cx.valid & (R >= p) & (R < m),
                    [&]
                    {
#pragma unroll
                        for(int c = 0; c < W; ++c)
                        {
                            int const C = CB + c;
                            if(C >= p && C < m) X::st(cx.baseR, so + static_cast<unsigned>(C - p) * ub, x[s2][c]);
                        }
                        if(CB == 0 && V.keep_l21)  // L21: the multipliers below the pivot block, read by a separate forward pass only
                        {
                            vu const lo = cx.offF + (X::to_u(R) << 3) + pf;
#pragma unroll
                            for(int c = 0; c < (W < 16 ? W : 16); ++c)
                                if(c < p) X::st(cx.baseF, lo + static_cast<unsigned>(c) * mb, x[s2][c]);
                        }
                    });
        }
    }

    // columns [CB, CB + 8), CB >= 16, of a front whose first block is done: assemble, apply the p pivots with the multipliers a[s][kk],
    // store; then the next chunk
    template <class X, int RS, int CB>
    PEQ_DEV void quad_chunks(DevView const& V, QuadCtx<X> const& cx, int const* blk, QuadLaneIdxM<X> const& L, int m, int p, int nch, long long lptr, long long sptr,
                             typename X::vd const (&a)[RS][16])
    {
        using vd = typename X::vd;
        if constexpr(CB < 16 * RS)
        {
            if(CB < m)
            {
                vd x[RS][8];
                quad_assemble<X, RS, CB, 8, false>(V, cx, blk, L, m, p, nch, x, nullptr);
#pragma unroll
                for(int kk = 0; kk < 16; ++kk)
                {
                    if(kk < p)
                    {
#pragma unroll
                        for(int G = 0; G < 8; G += 4)
                        {
                            if(CB + G < m)
                            {
#pragma unroll
                                for(int c = G; c < G + 4; ++c)
                                {
                                    vd const row = X::bcast(x[0][c], kk);
#pragma unroll
                                    for(int s2 = 0; s2 < RS; ++s2) x[s2][c] = X::fma(-a[s2][kk], row, x[s2][c]);
                                }
                            }
                        }
                    }
                }
                quad_store<X, RS, CB, 8>(V, cx, m, p, lptr, sptr, x);
                quad_chunks<X, RS, CB + 8>(V, cx, blk, L, m, p, nch, lptr, sptr, a);
            }
        }
    }

    template <class X, int RS>
    PEQ_DEV typename X::vm quad_front_mid(DevView const& V, QuadCtx<X> const& cx, int const* blk, QuadLaneIdxM<X> const& L)
    {
        using vd = typename X::vd;
        using vi = typename X::vi;
        using vu = typename X::vu;
        using vm = typename X::vm;
        int const m = blk[0], c0 = blk[2];
        int const p = (V.quad & 2) ? 0 : blk[1], nch = (V.quad & 8) ? 0 : blk[4];  // (developer timing knobs, as quad_front)
        int const u = m - p;
        long long const lptr = static_cast<long long>(static_cast<unsigned>(blk[6])) | (static_cast<long long>(blk[7]) << 32);
        long long const sptr = static_cast<long long>(static_cast<unsigned>(blk[8])) | (static_cast<long long>(blk[9]) << 32);
        vi const r = cx.r;
        vd a[RS][16];
        vd g[RS];
        quad_assemble<X, RS, 0, 16, true>(V, cx, blk, L, m, p, nch, a, g);
        vm bad = X::none();
#pragma unroll
        for(int kk = 0; kk < 16; ++kk)
        {
            if(kk < p)  // (guards, not early exits: the loop must unroll -- the register arrays are indexed by constants)
            {
                vd const piv = X::bcast(a[0][kk], kk);
                bad = bad | X::bad(piv);
                vd const rp = X::rcp(piv);
                vd lm[RS];
                {
                    vd const l = a[0][kk] * rp;
                    vm const below = r > kk;
                    lm[0] = X::sel(below, l, vd(0.0));
                    a[0][kk] = X::sel(below, l, a[0][kk]);
                }
#pragma unroll
                for(int s2 = 1; s2 < RS; ++s2)
                {
                    lm[s2] = a[s2][kk] * rp;  // rows 16.. are below every pivot (rows >= m hold zeros)
                    a[s2][kk] = lm[s2];
                }
#pragma unroll
                for(int C0 = 0; C0 < 16; C0 += 4)
                {
                    if(C0 + 3 > kk && C0 < m)
                    {
#pragma unroll
                        for(int C = C0; C < C0 + 4; ++C)
                        {
                            if(C > kk)
                            {
                                vd const row = X::bcast(a[0][C], kk);
#pragma unroll
                                for(int s2 = 0; s2 < RS; ++s2) a[s2][C] = X::fma(-lm[s2], row, a[s2][C]);
                            }
                        }
                    }
                }
                vd const grow = X::bcast(g[0], kk);
#pragma unroll
                for(int s2 = 0; s2 < RS; ++s2) g[s2] = X::fma(-lm[s2], grow, g[s2]);
            }
        }
        quad_store<X, RS, 0, 16>(V, cx, m, p, lptr, sptr, a);
        {
            // update vector behind the update matrix, forward-substituted pivots to w
            unsigned const ub = static_cast<unsigned>(u) * 8u;
#pragma unroll
            for(int s2 = 0; s2 < RS; ++s2)
            {
                vi const R = r + 16 * s2;
                vu const so = cx.offR + (X::to_u(R - p) << 3) + static_cast<unsigned>(sptr) * 8u;
                X::when(cx.valid & (R >= p) & (R < m), [&] { X::st(cx.baseR, so + static_cast<unsigned>(u) * ub, g[s2]); });
            }
            X::when(cx.valid & (r < p), [&] { X::st(cx.baseW, cx.offW + static_cast<unsigned>(c0) * 8u + (X::to_u(r) << 3), g[0]); });
        }
        if constexpr(RS > 1)
        {
            // the multipliers of the chunks below: rows at or above pivot kk take none
#pragma unroll
            for(int kk = 0; kk < 16; ++kk) a[0][kk] = X::sel(r > kk, a[0][kk], vd(0.0));
            quad_chunks<X, RS, 16>(V, cx, blk, L, m, p, nch, lptr, sptr, a);
        }
        return bad;
    }

    // One wavefront: quad `quad` (four instances of V.q_list) x wave-front list `list` (0 .. n_parts * n_waves - 1).
    template <class X>
    PEQ_DEV void quad_factor_list(DevView const& V, int quad, int list)
    {
        using vi = typename X::vi;
        using vu = typename X::vu;
        using vm = typename X::vm;
        int const* ql = V.q_list + static_cast<long long>(quad) * 4;
        int const b0 = ql[0];
        if(b0 < 0) return;  // an empty quad: a captured launch sequence keeps the grid of the full sweep, converged instances leave holes at the end
        vi const lane = X::lane();
        vi const q = lane >> 4;
        vi const b = X::ld_i32(ql, X::to_u(q) << 2);
        QuadCtx<X> cx;
        cx.valid = b >= 0;
        vi const bb = X::sel(cx.valid, b, vi(b0));
        vu const d = X::to_u(bb - b0);  // the host packs a quad so that (d + 1) * stride < 2^32 for every array
        cx.r = lane & 15;
        cx.offA = d * static_cast<unsigned>(V.nnzA * 8ll);
        cx.offF = d * static_cast<unsigned>(V.factor_doubles * 8ll);
        cx.offR = d * static_cast<unsigned>(V.arena_doubles * 8ll);
        cx.offW = d * static_cast<unsigned>(V.rows * 8ll);
        cx.baseA = reinterpret_cast<char const*>(V.aval + static_cast<long long>(b0) * V.nnzA);
        cx.baseF = reinterpret_cast<char*>(V.factor + static_cast<long long>(b0) * V.factor_doubles);
        cx.baseR = reinterpret_cast<char*>(V.arena + static_cast<long long>(b0) * V.arena_doubles);
        cx.baseW = reinterpret_cast<char*>(V.w + static_cast<long long>(b0) * V.rows);
        // LDS of the wavefront: [instance q][q_lds_stride doubles], slot 0 of every instance stack holds a zero
        cx.ldsq = X::to_u(q) * static_cast<unsigned>(V.q_lds_stride * 8) + 8u;
        cx.ldsz = X::to_u(q) * static_cast<unsigned>(V.q_lds_stride * 8);
#if PE_QUAD_LDS_STACK
        if(V.q_lds_stride > 0)
        {
            X::when(cx.r == 0, [&] { X::lds_st(cx.ldsz, typename X::vd(0.0)); });
            X::lds_fence();
        }
#endif
        int const* lp = V.q_lists + 2 * list;
        int const* blk = V.q_prog + lp[0];
        int const nfr = lp[1];
        vm bad = X::none();
        long long clkv[6] = {0, 0, 0, 0, 0, 0};
        bool const clk = V.prof && list == 0 && (V.quad & 32);  // developer phase clocks: PHY_ENGINE_HIP_QUAD bit 5
        if(nfr > 0)
        {
            QuadLaneIdx<X> cur;
            quad_lane_idx<X>(cur, V.q_lane + blk[10], blk[5], cx.r);
            for(int i = 0; i < nfr; ++i)
            {
                int const rs = blk[5], nch = blk[4], rs_next = blk[11];
                QuadLaneIdx<X> nxt;  // (the last front of a list loads its own data again: no conditional copy of the register block)
                quad_lane_idx<X>(nxt, V.q_lane + (rs_next ? blk[12] : blk[10]), rs_next ? rs_next : rs, cx.r);
                if(rs == 1) bad = bad | quad_front<X, 1>(V, cx, blk, cur, clk, clkv);
                else
                    bad = bad | quad_front<X, 2>(V, cx, blk, cur, clk, clkv);
                cur = nxt;
                blk += 16 + 32 * nch;
            }
        }
        if(clk) X::prof(V.prof + static_cast<long long>(b0) * PE_PROF + 64, clkv, 6);
        X::flag(V.flags, bb, 4, bad & cx.valid & (cx.r == 0));
    }

    // One wavefront of the MID launch: quad `quad` x list `list` of MID fronts.
    template <class X>
    PEQ_DEV void quad_factor_mid_list(DevView const& V, int quad, int list)
    {
        using vi = typename X::vi;
        using vu = typename X::vu;
        using vm = typename X::vm;
        int const* lp = V.q2_lists + 2 * list;
        int const nfr = lp[1];
        if(nfr <= 0) return;
        int const* ql = V.q_list + static_cast<long long>(quad) * 4;
        int const b0 = ql[0];
        if(b0 < 0) return;  // an empty quad: a captured launch sequence keeps the grid of the full sweep, converged instances leave holes at the end
        vi const lane = X::lane();
        vi const q = lane >> 4;
        vi const b = X::ld_i32(ql, X::to_u(q) << 2);
        QuadCtx<X> cx;
        cx.valid = b >= 0;
        vi const bb = X::sel(cx.valid, b, vi(b0));
        vu const d = X::to_u(bb - b0);
        cx.r = lane & 15;
        cx.offA = d * static_cast<unsigned>(V.nnzA * 8ll);
        cx.offF = d * static_cast<unsigned>(V.factor_doubles * 8ll);
        cx.offR = d * static_cast<unsigned>(V.arena_doubles * 8ll);
        cx.offW = d * static_cast<unsigned>(V.rows * 8ll);
        cx.baseA = reinterpret_cast<char const*>(V.aval + static_cast<long long>(b0) * V.nnzA);
        cx.baseF = reinterpret_cast<char*>(V.factor + static_cast<long long>(b0) * V.factor_doubles);
        cx.baseR = reinterpret_cast<char*>(V.arena + static_cast<long long>(b0) * V.arena_doubles);
        cx.baseW = reinterpret_cast<char*>(V.w + static_cast<long long>(b0) * V.rows);
        cx.ldsq = vu(0u);
        cx.ldsz = vu(0u);
        int const* blk = V.q2_prog + lp[0];
        vm bad = X::none();
        QuadLaneIdxM<X> cur;
        quad_lane_idx_mid<X>(cur, V.q2_lane + blk[10], blk[5], cx.r);
        for(int i = 0; i < nfr; ++i)
        {
            int const rs = blk[5], nch = blk[4], rs_next = blk[11];
            QuadLaneIdxM<X> nxt;
            quad_lane_idx_mid<X>(nxt, V.q2_lane + (rs_next ? blk[12] : blk[10]), rs_next ? rs_next : rs, cx.r);
            if(rs == 1) bad = bad | quad_front_mid<X, 1>(V, cx, blk, cur);
            else if(rs == 2)
                bad = bad | quad_front_mid<X, 2>(V, cx, blk, cur);
            else if(rs == 3)
                bad = bad | quad_front_mid<X, 3>(V, cx, blk, cur);
            else
                bad = bad | quad_front_mid<X, 4>(V, cx, blk, cur);
            cur = nxt;
            blk += 16 + 32 * nch;
        }
        X::flag(V.flags, bb, 4, bad & cx.valid & (cx.r == 0));
    }

    // =====================================================================================================================
    // Backward pass of the quad fronts (k_m2_backward_quads): x_piv = U11^-1 (w_piv - U12 x_anc), parents before children.
    // lane = 16 q + r: instance q, pivot row r (p <= 16: one row set).  The ancestors' unknowns come straight from w (their fronts
    // stored them there: the launch before for cooperative / per-instance parents, earlier in this wavefront's walk otherwise); the
    // panels' loads are requested before the fence that makes the parent's stores visible.  Same summation order as
    // front_backward_lean + tri_upper on a 64-lane team (four interleaved partial sums of U12 x_anc, subtracted in order; the
    // refined quotient of the triangular solve): the results are those of the per-instance path.
    // =====================================================================================================================
#ifndef PE_QUAD_BACK_DPP
    #define PE_QUAD_BACK_DPP 1  // 0: round 3's one load per ancestor term (A/B switch)
#endif
    template <class X>
    PEQ_DEV void quad_backward_list(DevView const& V, int quad, int list)
    {
        using vd = typename X::vd;
        using vi = typename X::vi;
        using vu = typename X::vu;
        using vm = typename X::vm;
        int const* lp = V.q_lists + 2 * list;
        int const nfr = lp[1];
        if(nfr <= 0) return;
        int first = 0;  // fronts of the lists before this one
        for(int L = 0; L < list; ++L) first += V.q_lists[2 * L + 1];
        int const* ql = V.q_list + static_cast<long long>(quad) * 4;
        int const b0 = ql[0];
        if(b0 < 0) return;  // (an empty quad, see quad_factor_list)
        vi const lane = X::lane();
        vi const q = lane >> 4, r = lane & 15;
        vi const b = X::ld_i32(ql, X::to_u(q) << 2);
        vm const valid = b >= 0;
        vi const bb = X::sel(valid, b, vi(b0));
        vu const d = X::to_u(bb - b0);
        vu const offF = d * static_cast<unsigned>(V.factor_doubles * 8ll), offW = d * static_cast<unsigned>(V.rows * 8ll);
        char const* baseF = reinterpret_cast<char const*>(V.factor + static_cast<long long>(b0) * V.factor_doubles);
        char* baseW = reinterpret_cast<char*>(V.w + static_cast<long long>(b0) * V.rows);
        int const* blk = V.q_bprog + static_cast<long long>(first) * 40;
        for(int i = 0; i < nfr; ++i, blk += 40)
        {
            int const m = blk[0], p = blk[1], c0 = blk[2], u = blk[3];
            unsigned const pf = static_cast<unsigned>(static_cast<long long>(static_cast<unsigned>(blk[4])) | (static_cast<long long>(blk[5]) << 32)) * 8u;
            unsigned const mb = static_cast<unsigned>(m) * 8u, pb = static_cast<unsigned>(p) * 8u;
            unsigned const pu = pf + static_cast<unsigned>(m) * pb;
            vm const own = r < p;
            vu const ro = X::to_u(X::sel(own, r, vi(0))) << 3;  // (lanes beyond the pivots re-read row 0 and drop it)
            // U12 row r (p x u, ld p) and U11 column k, rows <= k (top block of the L panel, ld m): independent of w
            vd u12[32], ucol[16];
#pragma unroll
            for(int J = 0; J < 32; J += 4)
            {
                if(J < u)
                {
#pragma unroll
                    for(int j = J; j < J + 4; ++j) u12[j] = X::ld(baseF, offF + ro + (pu + static_cast<unsigned>(j < u ? j : 0) * pb));
                }
                else
                {
#pragma unroll
                    for(int j = J; j < J + 4; ++j) u12[j] = vd(0.0);
                }
            }
#pragma unroll
            for(int K = 0; K < 16; K += 4)
            {
                if(K < p)
                {
#pragma unroll
                    for(int k = K; k < K + 4; ++k) ucol[k] = X::ld(baseF, offF + ro + (pf + static_cast<unsigned>(k < p ? k : 0) * mb));
                }
                else
                {
#pragma unroll
                    for(int k = K; k < K + 4; ++k) ucol[k] = vd(0.0);
                }
            }
#if PE_QUAD_BACK_DPP
            // positions of the ancestors' unknowns this lane fetches: x_anc[r] and x_anc[16 + r] (static table: requested with the panels)
            vu const ia0 = X::to_u(X::ld_i32(blk + 8, X::to_u(X::sel(r < u, r, vi(0))) << 2)) << 3;
            vu const ia1 = X::to_u(X::ld_i32(blk + 8, X::to_u(X::sel(r + 16 < u, r + 16, vi(0))) << 2)) << 3;
#endif
            // the unknowns this front reads are in memory: behind a fence only if this wavefront wrote one of them since its last fence
            // (blk[6], pe_symbolic.cpp) -- else the loads above and below go out together, one memory round trip instead of two
            if(blk[6] || !PE_QUAD_BACK_LAZY_FENCE) X::fence();
            vd const wi = X::ld(baseW, offW + static_cast<unsigned>(c0) * 8u + ro);
            vd acc[4] = {vd(0.0), vd(0.0), vd(0.0), vd(0.0)};
#if PE_QUAD_BACK_DPP
            // x_anc: TWO loads per lane (lane r of an instance takes x_anc[r] and x_anc[16 + r]) and a row broadcast per term, instead of
            // one load per term in which the 16 lanes of an instance read the same address -- 32 of the 81 vector memory instructions of
            // a front; the kernel was bound by the address unit (round 4).  Same products, same order of the four partial sums.
            vd const xlo = X::ld(baseW, offW + ia0), xhi = u > 16 ? X::ld(baseW, offW + ia1) : vd(0.0);
#pragma unroll
            for(int J = 0; J < 32; J += 4)
            {
                if(J < u)
                {
#pragma unroll
                    for(int j = J; j < J + 4; ++j)
                    {
                        vd const xa = X::bcast(j < 16 ? xlo : xhi, j & 15);
                        if(j < u) acc[j & 3] = acc[j & 3] + u12[j] * xa;
                    }
                }
            }
#else
#pragma unroll
            for(int J = 0; J < 32; J += 4)
            {
                if(J < u)
                {
#pragma unroll
                    for(int j = J; j < J + 4; ++j)
                    {
                        // x_anc[j]: one value per instance (every lane of a quad reads the same address)
                        vd const xa = X::ld(baseW, offW + static_cast<unsigned>(blk[8 + (j < u ? j : 0)]) * 8u);
                        if(j < u) acc[j & 3] = acc[j & 3] + u12[j] * xa;
                    }
                }
            }
#endif
            vd ti = wi;
#pragma unroll
            for(int g = 0; g < 4; ++g) ti = ti - acc[g];
            // U11 x = t: lane r owns x_r; the dependent chain runs on row broadcasts
            vd dg = vd(1.0);
#pragma unroll
            for(int k = 0; k < 16; ++k) dg = X::sel(own & (r == k), ucol[k], dg);
            vd const rr = vd(1.0) / dg;
#pragma unroll
            for(int k = 15; k >= 0; --k)
            {
                if(k < p)
                {
                    vd const uc = X::sel(r < k, ucol[k], vd(0.0));
                    vd qv = ti * rr;
                    vd const e = X::fma(-qv, dg, ti);
                    qv = X::fma(e, rr, qv);
                    vd const xk = X::bcast(qv, k);
                    ti = X::sel(r == k, xk, ti);
                    ti = X::sel(r < k, X::fma(-uc, xk, ti), ti);
                }
            }
            X::when(valid & own, [&] { X::st(baseW, offW + static_cast<unsigned>(c0) * 8u + (X::to_u(r) << 3), ti); });
        }
    }
}  // namespace pe
