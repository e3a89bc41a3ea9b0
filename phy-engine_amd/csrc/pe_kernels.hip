// pe_kernels.hip -- gfx950 kernels of the resident transient path.
//
// Mapping (MI355X-first): ONE WORKGROUP = ONE CIRCUIT INSTANCE.  A workgroup owns its instance for a whole
// launch: companion update, device evaluation, MNA gather, multifrontal LU with the dense fronts staged in LDS,
// triangular solves with the front-local vectors in LDS, Newton test -- with workgroup barriers only, no
// inter-workgroup traffic and no host round trip inside a time step.  A Monte-Carlo sweep fills the chip with
// independent instances (256 CUs x k workgroups); symbolic data is shared read-only and stays in L2/MALL.
//
// Two launch schemes share the front code of pe_front.hpp:
//   * resident kernels k_tr_steps<MINW> / k_dc_point<MINW> / k_factor_solve: one workgroup per instance does everything
//     (MINW = 2: <= 256 VGPRs, one 512-thread workgroup per CU; MINW = 4: 128 VGPRs, 256-thread workgroups, 4 per CU);
//   * multi-workgroup schedule k_m2_*: one launch per phase and per top level of the assembly tree, an instance spread over
//     n_parts workgroups + one workgroup per top front; the host drives the Newton loop from one flag word per instance
//     (pe_engine_newton.cpp run_m2_tr / m2_point).  Selected for few instances of a large circuit.
// Small-signal AC runs the same kernels on the real-equivalent 2N system (pe_ac.cpp).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "pe_front.hpp"
#include "pe_kernels.hpp"
#include "pe_quad.hpp"
#include "pe_top_plan.hpp"

namespace pe
{
    // ---- matrix-core / cross-lane primitives shared by the workgroup team and the single-wavefront team
    struct WaveOps
    {
        // a wavefront's own LDS / global writes become visible to its own later reads (other lanes included)
        __device__ __forceinline__ void wave_fence() const
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        __device__ __forceinline__ int lanes() const { return 64; }
        // LDS-only variant of wave_fence(): does not wait for outstanding global stores / loads (s_waitcnt lgkmcnt only)
        __device__ __forceinline__ void wave_fence_lds() const
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        // a value that is the same in every lane of the wavefront but that the compiler cannot prove uniform (anything
        // derived from the wavefront's index): moved to an SGPR so that the loads it indexes become scalar loads and
        // the address arithmetic leaves the vector ALU
        __device__ __forceinline__ int uniform(int v) const { return __builtin_amdgcn_readfirstlane(v); }
        // 1 / d for a pivot: v_rcp_f64 + two Newton steps (no v_div_scale / v_div_fmas / v_div_fixup sequence: pivots are
        // checked finite and non-zero before, and circuit values sit far from the subnormal range)
        __device__ __forceinline__ double rcp(double d) const
        {
            double r = __builtin_amdgcn_rcp(d);
            double e = __builtin_fma(-d, r, 1.0);
            r = __builtin_fma(e, r, r);
            e = __builtin_fma(-d, r, 1.0);
            return __builtin_fma(e, r, r);
        }
        __device__ __forceinline__ long long clock() const { return static_cast<long long>(wall_clock64()); }

        // LU (no pivoting) of a kb x kb block (kb <= 8) held one entry per lane: lane l <-> (row l&7, col l>>3).
        // The dependent chain runs on cross-lane shuffles, not on LDS round trips.  On return the block in memory
        // holds U on and above the diagonal and the SCALED multipliers below it.  Returns 1 on a bad pivot.
        // (the pivots' reciprocals go to rdiag()[0..8): the row solves of panel_solve multiply by them instead of recomputing)
        __device__ __forceinline__ double* rdiag() const
        {
            __shared__ double rd[8];
            return rd;
        }
        // LU (no pivoting) of the KB x KB diagonal block of a cooperative front, on ONE wavefront: lane i < KB holds row i in
        // registers, the pivot row of step kk comes from lane kk by v_readlane (no LDS round trip on the dependent chain; the
        // first version exchanged entries through ds_bpermute: three LDS latencies per pivot).  Same operation order as
        // block_step_t's row part.  On return the block in LDS holds U on and above the diagonal and the SCALED multipliers
        // below it, rdiag()[kk] the pivots' reciprocals.  Returns 1 on a bad pivot.
        template <int KB>
        __device__ __forceinline__ int diag_lu_t(double* blk, int ld, int lane) const
        {
            bool const own = lane < KB;
            double* row = blk + (own ? lane : KB - 1);
            double v[KB];
#pragma unroll
            for(int c = 0; c < KB; ++c) v[c] = row[c * ld];
            int bad = 0;
            double* rd = rdiag();
#pragma unroll
            for(int kk = 0; kk < KB; ++kk)
            {
                double const piv = bcast(v[kk], kk);
                if(piv == 0.0 || !(fabs(piv) <= 1.7976931348623157e308)) bad = 1;
                double const rp = rcp(piv);
                rd[kk] = rp;  // (every lane stores the same value)
                double const l = v[kk] * rp;
                bool const below = lane > kk;
                double const lm = below ? l : 0.0;
#pragma unroll
                for(int c = kk + 1; c < KB; ++c) v[c] = __builtin_fma(-lm, bcast(v[c], kk), v[c]);
                v[kk] = below ? l : v[kk];
            }
            if(own)
            {
#pragma unroll
                for(int c = 0; c < KB; ++c) row[c * ld] = v[c];
            }
            return bad;
        }
        __device__ __forceinline__ int diag_lu8(double* blk, int ld, int kb, int lane) const
        {
            switch(kb)
            {
                case 8: return diag_lu_t<8>(blk, ld, lane);
                case 7: return diag_lu_t<7>(blk, ld, lane);
                case 6: return diag_lu_t<6>(blk, ld, lane);
                case 5: return diag_lu_t<5>(blk, ld, lane);
                case 4: return diag_lu_t<4>(blk, ld, lane);
                case 3: return diag_lu_t<3>(blk, ld, lane);
                case 2: return diag_lu_t<2>(blk, ld, lane);
                default: return diag_lu_t<1>(blk, ld, lane);
            }
        }
        // Rows below a factored KB x KB diagonal block (x U11 = a) and columns right of it (L11 y = a), one thread each, the
        // block read through v_readlane from one entry per lane.  The split between rows and columns falls on a wavefront
        // boundary (a wavefront solves rows OR columns: no divergent double pass); KB is a compile-time constant.
        template <int KB>
        __device__ __forceinline__ void panel_solve_t(double* Lp, int ld, int m, double* Up, int ldu, double* g, int p, int u, int k0, bool fuse, int t0, int T) const
        {
            int const lane = t0 & 63, wv = __builtin_amdgcn_readfirstlane(t0 >> 6), nwv = T >> 6;
            int const nrows = m - k0 - KB, ncolL = p - k0 - KB, ncols = ncolL + u + (fuse ? 1 : 0);
            Blk8 const B8 = blk_load(Lp + k0 + k0 * ld, ld, KB, lane);
            int const row_items = (nrows + 63) >> 6, col_items = (ncols + 63) >> 6;
            double const* rd = rdiag();
            for(int item = wv; item < row_items + col_items; item += nwv)
            {
                double x[KB];
                if(item < row_items)
                {
                    int const i = item * 64 + lane;
                    bool const own = i < nrows;
                    double* row = Lp + (k0 + KB + (own ? i : nrows - 1)) + k0 * ld;
#pragma unroll
                    for(int kk = 0; kk < KB; ++kk) x[kk] = row[kk * ld];
#pragma unroll
                    for(int kk = 0; kk < KB; ++kk)
                    {
                        double acc = x[kk];
#pragma unroll
                        for(int r = 0; r < kk; ++r) acc = __builtin_fma(-x[r], blk_at(B8, r, kk), acc);
                        x[kk] = acc * rd[kk];
                    }
                    if(own)
                    {
#pragma unroll
                        for(int kk = 0; kk < KB; ++kk) row[kk * ld] = x[kk];
                    }
                }
                else
                {
                    int const jj = (item - row_items) * 64 + lane;
                    bool const own = jj < ncols;
                    int const j = own ? jj : ncols - 1;
                    double* col = j < ncolL ? Lp + (k0 + KB + j) * ld + k0 : (j < ncolL + u ? Up + (j - ncolL) * ldu + k0 : g + k0);
#pragma unroll
                    for(int kk = 0; kk < KB; ++kk) x[kk] = col[kk];
#pragma unroll
                    for(int kk = 1; kk < KB; ++kk)
                    {
                        double acc = x[kk];
#pragma unroll
                        for(int r = 0; r < kk; ++r) acc = __builtin_fma(-blk_at(B8, kk, r), x[r], acc);
                        x[kk] = acc;
                    }
                    if(own)
                    {
#pragma unroll
                        for(int kk = 1; kk < KB; ++kk) col[kk] = x[kk];
                    }
                }
            }
        }
        __device__ __forceinline__ void panel_solve(double* Lp, int ld, int m, double* Up, int ldu, double* g, int p, int u, int k0, int kb, bool fuse, int t0, int T) const
        {
            switch(kb)
            {
                case 8: return panel_solve_t<8>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, t0, T);
                case 7: return panel_solve_t<7>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, t0, T);
                case 6: return panel_solve_t<6>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, t0, T);
                case 5: return panel_solve_t<5>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, t0, T);
                case 4: return panel_solve_t<4>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, t0, T);
                case 3: return panel_solve_t<3>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, t0, T);
                case 2: return panel_solve_t<2>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, t0, T);
                default: return panel_solve_t<1>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, t0, T);
            }
        }

        // broadcast lane k's value (k wavefront-uniform): v_readlane, no LDS round trip
        __device__ __forceinline__ double bcast(double v, int k) const
        {
            int const lo = __builtin_amdgcn_readlane(__double2loint(v), k);
            int const hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
            return __hiloint2double(hi, lo);
        }
        // The factored 8 x 8 diagonal block held one entry per lane (lane l <-> row l&7, column l>>3); blk_at(r, c) broadcasts an
        // entry by v_readlane: the solves against the block then need ONE LDS read per lane instead of one per triangle entry
        struct Blk8
        {
            double e;
        };
        __device__ __forceinline__ Blk8 blk_load(double const* blk, int ld, int kb, int lane) const
        {
            int const r = lane & 7, c = lane >> 3;
            return Blk8{(r < kb && c < kb) ? blk[r + c * ld] : (r == c ? 1.0 : 0.0)};
        }
        __device__ __forceinline__ double blk_at(Blk8 const& b, int r, int c) const { return bcast(b.e, r + 8 * c); }
        // One block step (KB <= 8 pivots from k0) of a front that ONE wavefront owns (m <= 64).
        // Rows: lane i holds row k0 + i of the block's columns in registers and the pivot row of step kk comes from lane kk by
        // v_readlane -- the LU of the diagonal block and the solve of the rows below it (x U11 = a) are the same elimination,
        // with no LDS round trip on the dependent chain.  Columns: lane j holds the block rows of one column right of the
        // block (rest of the L panel, the U panel, the right-hand-side column) and solves L11 y = a with the multipliers read
        // from the row lanes' registers.  Same operation order as diag_lu8 + the per-thread solves of front_factor.
        // KB is a compile-time constant (block_step dispatches on the wavefront-uniform kb): the pivot / column loops unroll into
        // straight-line code with no branch per entry; lanes outside the front work on a clamped row / column (valid
        // addresses, results dropped) and rows not below the pivot take a zero multiplier instead of an exec mask.
        template <int KB>
        __device__ __forceinline__ int block_step_t(double* Lp, int ld, int m, double* Up, int ldu, double* g, int p, int u, int k0, bool fuse, int lane) const
        {
            int const nrow = m - k0;  // >= KB >= 1
            bool const own = lane < nrow;
            double* row = Lp + (k0 + (own ? lane : nrow - 1)) + k0 * ld;
            double v[KB];
#pragma unroll
            for(int c = 0; c < KB; ++c) v[c] = row[c * ld];
            int bad = 0;
#pragma unroll
            for(int kk = 0; kk < KB; ++kk)
            {
                double const piv = bcast(v[kk], kk);
                if(piv == 0.0 || !(fabs(piv) <= 1.7976931348623157e308)) bad = 1;
                double const l = v[kk] * rcp(piv);
                bool const below = lane > kk;
                double const lm = below ? l : 0.0;
#pragma unroll
                for(int c = kk + 1; c < KB; ++c) v[c] = __builtin_fma(-lm, bcast(v[c], kk), v[c]);
                v[kk] = below ? l : v[kk];
            }
            if(own)
            {
#pragma unroll
                for(int c = 0; c < KB; ++c) row[c * ld] = v[c];
            }
            int const ncolL = p - k0 - KB, ncols = ncolL + u + (fuse ? 1 : 0);
            if(ncols > 0)
            {
                bool const ownc = lane < ncols;
                int const j = ownc ? lane : ncols - 1;
                double* col = j < ncolL ? Lp + (k0 + KB + j) * ld + k0 : (j < ncolL + u ? Up + (j - ncolL) * ldu + k0 : g + k0);
                double x[KB];
#pragma unroll
                for(int kk = 0; kk < KB; ++kk) x[kk] = col[kk];
#pragma unroll
                for(int kk = 1; kk < KB; ++kk)
                {
                    double acc = x[kk];
#pragma unroll
                    for(int r = 0; r < kk; ++r) acc = __builtin_fma(-bcast(v[r], kk), x[r], acc);
                    x[kk] = acc;
                }
                if(ownc)
                {
#pragma unroll
                    for(int kk = 1; kk < KB; ++kk) col[kk] = x[kk];
                }
            }
            return bad;
        }
        __device__ __forceinline__ int block_step(double* Lp, int ld, int m, double* Up, int ldu, double* g, int p, int u, int k0, int kb, bool fuse,
                                                  int lane) const
        {
            switch(kb)
            {
                case 8: return block_step_t<8>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, lane);
                case 7: return block_step_t<7>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, lane);
                case 6: return block_step_t<6>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, lane);
                case 5: return block_step_t<5>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, lane);
                case 4: return block_step_t<4>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, lane);
                case 3: return block_step_t<3>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, lane);
                case 2: return block_step_t<2>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, lane);
                default: return block_step_t<1>(Lp, ld, m, Up, ldu, g, p, u, k0, fuse, lane);
            }
        }
        // Triangular solves of the triangular-solve phase, p <= 64, one wavefront: lane i owns t[i]; the dependent chain
        // runs on lane broadcasts, the matrix columns (LDS, independent of the chain) are fetched four steps ahead.
        // L y = t: unit lower triangle of the first p columns of Lb (ld); rows p..nrows-1 (nrows <= 64) are the
        // sub-diagonal block, updated by the same sweep (t[p..] -= L21 y)
        __device__ __forceinline__ void tri_lower_unit(double* t, double const* Lb, int ld, int p, int nrows, int lane) const
        {
            bool const own = lane < nrows;
            double ti = own ? t[lane] : 0.0;
            int const last = nrows > p ? p : p - 1;  // pivot steps that still have rows below them
            for(int k0 = 0; k0 < last; k0 += 4)
            {
                double l[4];
#pragma unroll
                for(int q = 0; q < 4; ++q) l[q] = (own && k0 + q < p) ? Lb[lane + (k0 + q) * ld] : 0.0;
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
                    int const k = k0 + q;
                    double const yk = bcast(ti, k & 63);
                    if(lane > k) ti -= l[q] * yk;
                }
            }
            if(own) t[lane] = ti;
        }
        // U x = t: upper triangle of Ub (p x p, ld) with its diagonal.  nu > 0: first t[0..p) -= U12 * t[p..p+nu) with
        // U12 (p x nu, ld p) stored right behind the p x p block (small fronts: everything on this wavefront).
        __device__ __forceinline__ void tri_upper(double* t, double const* Ub, int ld, int p, int nu, int lane) const
        {
            bool const own = lane < p;
            double ti = own ? t[lane] : 0.0;
            if(nu > 0)
            {
                double const* U12 = Ub + p * ld;
                double a0 = 0.0, a1 = 0.0;
                int j = 0;
                for(; j + 1 < nu; j += 2)
                {
                    double const u0 = own ? U12[lane + j * p] : 0.0, u1 = own ? U12[lane + (j + 1) * p] : 0.0;
                    a0 += u0 * t[p + j];
                    a1 += u1 * t[p + j + 1];
                }
                if(j < nu) a0 += (own ? U12[lane + j * p] : 0.0) * t[p + j];
                ti -= a0 + a1;
            }
            double const d = own ? Ub[lane + lane * ld] : 1.0;
            double const r = 1.0 / d;  // off the chain; the quotient below is refined to the correctly rounded t / d
            for(int k0 = p - 1; k0 >= 0; k0 -= 4)
            {
                double uc[4];
#pragma unroll
                for(int q = 0; q < 4; ++q) uc[q] = (k0 - q >= 0 && lane < k0 - q) ? Ub[lane + (k0 - q) * ld] : 0.0;
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
                    int const k = k0 - q;
                    if(k >= 0)
                    {
                        double qv = ti * r;
                        double const e = __builtin_fma(-qv, d, ti);
                        qv = __builtin_fma(e, r, qv);
                        double const xk = bcast(qv, k);
                        if(lane == k) ti = xk;
                        if(lane < k) ti -= uc[q] * xk;
                    }
                }
            }
            if(own) t[lane] = ti;
        }

        // children (<= 64, one per lane) whose staged block mask has both the tile's row block and its column block: bit q set
        __device__ __forceinline__ unsigned long long tile_children(unsigned const* cmk, int nch, int ti, int tj, int lane) const
        {
            unsigned const mk = lane < nch ? cmk[lane] : 0u;
            int const bi = ti < 31 ? ti : 31, bj = tj < 31 ? tj : 31;
            return __ballot((((mk >> bi) & (mk >> bj)) & 1u) != 0u);
        }
        // max over the 64 lanes (every lane gets it)
        __device__ __forceinline__ double wave_max(double v) const
        {
#pragma unroll
            for(int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
            return v;
        }
        // ---- 16 x 16 fp64 tiles on the matrix core: v_mfma_f64_16x16x4_f64.
        // The instruction takes A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15] from lane l and keeps
        // D[(l>>4) + 4r][l&15] in accumulator register r (cdna_hip_programming.md 3, "f64 MFMA does NOT use these maps").
        // The tiles here are held TRANSPOSED (the instruction computes C^T -= B^T A^T, i.e. the operands swap seats):
        // lane l owns C[row = l&15][col = (l>>4) + 4r], so the 16 lanes of a quarter-wave touch 16 consecutive rows of
        // a column-major matrix -- 128-byte segments in HBM (Schur blocks) and conflict-free LDS banks (panels).
        using v4d = __attribute__((ext_vector_type(4))) double;
        struct Acc
        {
            v4d v;
        };
        __device__ __forceinline__ Acc tile_zero() const { return Acc{v4d{0.0, 0.0, 0.0, 0.0}}; }
        __device__ __forceinline__ void tile_add(Acc& a, Acc const& b) const { a.v += b.v; }
        // Tile loads are UNCONDITIONAL: a lane outside the mr x nc part of an edge tile reads the nearest entry inside it (a
        // valid address of the same block) instead of branching around the load, so the four loads of a tile issue back to
        // back.  Such a lane's accumulator entries are never stored (tile_store masks them) and never reach an in-range
        // entry: C is updated element by element, and tile_mulsub zeroes only what would (the k tail).
        __device__ __forceinline__ Acc tile_load(double const* C, int ldc, int mr, int nc, int lane) const
        {
            Acc a;
            int row = lane & 15;
            int const cb = lane >> 4;
            row = row < mr ? row : mr - 1;
            double const* p = C + row;
#pragma unroll
            for(int r = 0; r < 4; ++r)
            {
                int col = cb + 4 * r;
                col = col < nc ? col : nc - 1;
                a.v[r] = p[col * ldc];
            }
            return a;
        }
        __device__ __forceinline__ void tile_store(Acc const& a, double* C, int ldc, int mr, int nc, int lane) const
        {
            int const row = lane & 15, cb = lane >> 4;
            double* p = C + row + cb * ldc;
            if(mr == 16 && nc == 16)  // (wavefront-uniform) interior tile: four plain stores
            {
#pragma unroll
                for(int r = 0; r < 4; ++r) p[4 * r * ldc] = a.v[r];
            }
            else if(row < mr)
            {
#pragma unroll
                for(int r = 0; r < 4; ++r)
                    if(cb + 4 * r < nc) p[4 * r * ldc] = a.v[r];
            }
        }
        // acc -= A(mr x kd, ld lda) * B(kd x nc, ld ldb).  The operands of up to four k-steps are requested before the
        // first MFMA (unconditional loads from clamped rows / columns); only the k tail (kd % 4) is masked.
        __device__ __forceinline__ void tile_mulsub(Acc& a, double const* A, int lda, double const* B, int ldb, int mr, int nc, int kd, int lane) const
        {
            int const ij = lane & 15, kq = lane >> 4;
            double const* pa = A + (ij < mr ? ij : mr - 1) + kq * lda;
            double const* pb = B + kq + (ij < nc ? ij : nc - 1) * ldb;
            int kk = 0;
            for(; kk + 16 <= kd; kk += 16)
            {
                double av[4], bv[4];
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
                    av[q] = pa[(kk + 4 * q) * lda];
                    bv[q] = pb[kk + 4 * q];
                }
#pragma unroll
                for(int q = 0; q < 4; ++q) a.v = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[q], -av[q], a.v, 0, 0, 0);
            }
            if(kk + 8 <= kd)
            {
                double av[2], bv[2];
#pragma unroll
                for(int q = 0; q < 2; ++q)
                {
                    av[q] = pa[(kk + 4 * q) * lda];
                    bv[q] = pb[kk + 4 * q];
                }
#pragma unroll
                for(int q = 0; q < 2; ++q) a.v = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[q], -av[q], a.v, 0, 0, 0);
                kk += 8;
            }
            if(kk + 4 <= kd)
            {
                double const av = pa[kk * lda], bv = pb[kk];
                a.v = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, -av, a.v, 0, 0, 0);
                kk += 4;
            }
            if(kk < kd)
            {
                bool const in = kk + kq < kd;
                int const kc = in ? kq : 0;  // (a masked lane re-reads k = kk)
                double const a0 = A[(ij < mr ? ij : mr - 1) + (kk + kc) * lda], b0 = B[kk + kc + (ij < nc ? ij : nc - 1) * ldb];
                a.v = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? b0 : 0.0, in ? -a0 : 0.0, a.v, 0, 0, 0);
            }
        }
        template <class F>
        __device__ __forceinline__ void tile_foreach(Acc& a, int lane, F&& f) const
        {
            int const row = lane & 15, cb = lane >> 4;
#pragma unroll
            for(int r = 0; r < 4; ++r)
            {
                double t = a.v[r];
                f(row, cb + 4 * r, t);
                a.v[r] = t;
            }
        }
    };

    // one wavefront acting as a team of 64: barriers degenerate to wavefront fences
    struct WaveTeam : WaveOps
    {
        int lane_;
        __device__ __forceinline__ int tid() const { return lane_; }
        __device__ __forceinline__ int size() const { return 64; }
        __device__ __forceinline__ void sync() const { wave_fence(); }
        __device__ __forceinline__ void sync_lds() const { wave_fence_lds(); }
        __device__ __forceinline__ int sync_or(int v) const { return __any(v); }
        __device__ __forceinline__ int n_waves() const { return 1; }
        __device__ __forceinline__ bool single_wave() const { return true; }
        template <class F>
        __device__ __forceinline__ void for_each_wave(F&& body) const
        {
            body(0, lane_, 64);
        }
        __device__ __forceinline__ WaveTeam wave_team(int) const { return *this; }
        __device__ __forceinline__ void team_max4(double (&v)[4], double*) const
        {
#pragma unroll
            for(int k = 0; k < 4; ++k) v[k] = wave_max(v[k]);
        }
    };

    // the whole workgroup
    struct HipTeam : WaveOps
    {
        __device__ __forceinline__ int tid() const { return static_cast<int>(threadIdx.x); }
        __device__ __forceinline__ int size() const { return static_cast<int>(blockDim.x); }
        __device__ __forceinline__ void sync() const { __syncthreads(); }
        // barrier that orders LDS traffic only: global stores stay in flight across it (whoever reads them later does
        // so behind a full sync())
        __device__ __forceinline__ void sync_lds() const
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        __device__ __forceinline__ int sync_or(int v) const { return __syncthreads_or(v); }
        __device__ __forceinline__ int n_waves() const { return static_cast<int>(blockDim.x) >> 6; }
        __device__ __forceinline__ bool single_wave() const { return false; }
        template <class F>
        __device__ __forceinline__ void for_each_wave(F&& body) const
        {
            body(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6), static_cast<int>(threadIdx.x) & 63, 64);
        }
        __device__ __forceinline__ WaveTeam wave_team(int lane) const
        {
            WaveTeam w;
            w.lane_ = lane;
            return w;
        }
        // every thread gets the workgroup's maxima of its four values (scratch: 4 doubles per wavefront of dynamic LDS)
        __device__ __forceinline__ void team_max4(double (&v)[4], double* scratch) const
        {
            int const lane = static_cast<int>(threadIdx.x) & 63, w = static_cast<int>(threadIdx.x) >> 6, nw = static_cast<int>(blockDim.x) >> 6;
            __syncthreads();  // (the scratch region may still be read by the phase before)
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                v[k] = wave_max(v[k]);
                if(lane == 0) scratch[4 * w + k] = v[k];
            }
            __syncthreads();
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                double m = scratch[k];
                for(int q = 1; q < nw; ++q) m = fmax(m, scratch[4 * q + k]);
                v[k] = m;
            }
            __syncthreads();
        }
    };

    // dynamic LDS: V.lds_doubles doubles, carved per phase by pe_front.hpp
    extern __shared__ __attribute__((aligned(16))) double pe_lds[];

    // Two register budgets of the same code: MINW = 2 -> up to 256 VGPRs (one 512-thread workgroup per CU: few
    // instances, lowest latency, no spills); MINW = 4 -> 128 VGPRs (2-4 workgroups per CU: a sweep that oversubscribes
    // the chip hides each workgroup's dependent-latency chains behind the others).
    template <int MINW>
    __global__ void __launch_bounds__(PE_THREADS, MINW) k_tr_steps(DevView V, double dt, int nsteps, int reuse_factor)
    {
        int const b = static_cast<int>(blockIdx.x);
        if(b >= V.batch) return;
        HipTeam tm;
        tr_steps(tm, V, b, dt, nsteps, reuse_factor != 0, pe_lds);
    }

    template <int MINW>
    __global__ void __launch_bounds__(PE_THREADS, MINW) k_dc_point(DevView V, int mode)
    {
        int const b = static_cast<int>(blockIdx.x);
        if(b >= V.batch) return;
        HipTeam tm;
        dc_point(tm, V, b, mode, pe_lds);
    }

    // A x = b with A values / rhs already resident (solve_csr_real seam): factor + solve, instance 0..batch-1
    __global__ void __launch_bounds__(PE_THREADS, 2) k_factor_solve(DevView V, int do_factor)
    {
        int const b = static_cast<int>(blockIdx.x);
        if(b >= V.batch) return;
        HipTeam tm;
        int st = ST_OK;
        if(do_factor)
        {
            permute_rhs(tm, V, b);
            if(!factor_all(tm, V, b, pe_lds, true)) st = ST_SINGULAR;  // forward substitution rides along
        }
        if(st == ST_OK)
        {
            solve_all(tm, V, b, pe_lds, do_factor != 0);
            double const* x = V.x + static_cast<long long>(b) * V.rows;
            int nonfinite = 0;
            for(int r = tm.tid(); r < V.rows; r += tm.size())
                if(!(fabs(x[r]) <= 1.7976931348623157e308)) nonfinite = 1;
            if(tm.sync_or(nonfinite)) st = ST_SINGULAR;
        }
        if(tm.tid() == 0) V.status[b] = st;
    }

    // =====================================================================================================
    // Multi-workgroup mode (one or few instances of a large circuit): the same front code, split at the level-1 cut of
    // the tree into kernels whose boundaries are the only inter-workgroup synchronisation -- no spin-waits, no
    // cross-CU visibility protocol.  grid = (work items, instances).
    // =====================================================================================================
    struct GridTeam  // elementwise phases: G workgroups share one instance's device / slot / row loops
    {
        __device__ __forceinline__ int tid() const { return static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x); }
        __device__ __forceinline__ int size() const { return static_cast<int>(gridDim.x * blockDim.x); }
    };

    __global__ void __launch_bounds__(256) k_m2_companion(DevView V, double dt)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        companion_update(GridTeam{}, V, b, dt);
    }

    // companion_dt_set != 0: first Newton iteration of a transient step -- the trapezoidal companion update of that step (what k_m2_companion
    // does in a launch of its own) runs first, in the same threads: companion_update and eval_devices deal out every device kind by the same
    // thread index, so what eval reads of a device (history, junction voltage) is what THIS thread has just written; no barrier, one launch less
    __global__ void __launch_bounds__(256) k_m2_eval(DevView V, int mode, double t, double last_step, int dynamic_only, int companion_dt_set, double companion_dt)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        GridTeam tm;
        if(companion_dt_set) companion_update(tm, V, b, companion_dt);
        double const* x = V.x + static_cast<long long>(b) * V.rows;
        double* xp = V.xprev + static_cast<long long>(b) * V.rows;
        for(int r = tm.tid(); r < V.rows; r += tm.size()) xp[r] = x[r];
        eval_devices(tm, V, b, mode, t, last_step, dynamic_only != 0);
        if(tm.tid() == 0)
        {
            V.flags[b] = 0;
            if(V.eta_acc)
                for(int k = 0; k < 4; ++k) V.eta_acc[4 * b + k] = 0.0;
        }
    }

    // (also initialises the permuted work vector w = P rhs: k_m2_winit remains for the refinement solve, whose right-hand side is a residual)
    __global__ void __launch_bounds__(256) k_m2_stamp(DevView V, int dynamic_only)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        // dynamic_only: 0 everything, 1 the x-dependent slots and rows, 2 the x-dependent slots of the matrix + the whole right-hand side
        if(dynamic_only) stamp_dynamic_chunk(V, b, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x), static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x), true, dynamic_only == 2);
        else
            stamp_chunk(V, b, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x), static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x), true);
    }

    // LDS guard of the top launches: front s laid out for more LDS than this launch has (V.f_need: image or panels + right-hand-side
    // column, fixed with the launch plan) -> flag bit 3, nothing is touched.  Wavefront-uniform; one scalar load per front.
    __device__ __forceinline__ bool lds_overrun(DevView const& V, int b, int s, int lds_doubles)
    {
        if(V.f_need[s] <= lds_doubles - 2) return false;
        if(threadIdx.x == 0) atomicOr(V.flags + b, 8);
        return true;
    }

    // (the four team kernels of the split schedule come in the two register budgets of the resident kernels: MINW = 2 for few
    // big workgroups, MINW = 4 when several workgroups share a CU)
    template <int MINW>
    __global__ void __launch_bounds__(PE_THREADS, MINW) k_m2_factor_parts(DevView V)
    {
        // x = instance, y = part: workgroups are dispatched part by part (parts in descending cost, pe_symbolic.cpp), so the
        // last ones to start are the cheapest
        int const b = static_cast<int>(blockIdx.x);
        if(!V.active[b]) return;
        HipTeam tm;
        long long const t_start = V.prof ? tm.clock() : 0;
        if(!factor_part(tm, V, b, static_cast<int>(blockIdx.y), pe_lds, true) && tm.tid() == 0) atomicOr(V.flags + b, 4);
        // developer timeline (scripts/wg_timeline.py): start / end (100 MHz wall clock) and placement of the workgroups of parts 0..3 in
        // the LAST launch -- XCC_ID in the high word, HW_ID (wave slot, SIMD, CU, SE) in the low word
        if(V.prof && tm.tid() == 0 && blockIdx.y < 4)
        {
            long long* q = V.prof + b * PE_PROF + 48 + 3 * static_cast<int>(blockIdx.y);
            q[0] = t_start;
            q[1] = tm.clock();
            q[2] = (static_cast<long long>(__builtin_amdgcn_s_getreg(6164)) << 32) | static_cast<unsigned>(__builtin_amdgcn_s_getreg(63492));
        }
    }

    template <int MINW>
    __global__ void __launch_bounds__(PE_THREADS, MINW) k_m2_factor_top(DevView V, int level, int nlev, int lds_doubles)
    {
        // one workgroup per front of `level`; nlev > 1: a run of single-front levels (a chain at the top of the tree) handled by
        // the same workgroup one after the other -- saves a launch per level where a launch is most of the level's time
        // lds_doubles: the dynamic LDS THIS launch was given -- the cap of every front it runs; a front whose layout needs more was put
        // on the wrong launch (flag bit 3: internal error; the host refuses such a plan at load time, upload_symbolic)
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        HipTeam tm;
        for(int l = level; l < level + nlev; ++l)
        {
            int const s = V.top_list[V.top_ptr[l] + static_cast<int>(blockIdx.x)];
            if(lds_overrun(V, b, s, lds_doubles)) return;
            if(!front_factor(tm, V, b, s, pe_lds, lds_doubles - 2, 0, true))
            {
                if(tm.tid() == 0) atomicOr(V.flags + b, 4);
                return;
            }
        }
    }

    // the same with 16 wavefronts per workgroup: few instances leave the GPU idle, a top front then gets a whole CU's wavefront slots
    __global__ void __launch_bounds__(1024) k_m2_factor_top_wide(DevView V, int level, int nlev, int lds_doubles)
    {
        // one workgroup per front of `level`; nlev > 1: a run of single-front levels (a chain at the top of the tree) handled by
        // the same workgroup one after the other -- saves a launch per level where a launch is most of the level's time
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        HipTeam tm;
        // (this workgroup owns its CU's LDS: whole-front layout up to V.lds_top_doubles, and inside a run of single-front levels a
        //  chain link finds its front where the front before it left the Schur block -- ChainState, pe_front.hpp)
        ChainState cs;
        for(int l = level; l < level + nlev; ++l)
        {
            int const s = V.top_list[V.top_ptr[l] + static_cast<int>(blockIdx.x)];
            if(lds_overrun(V, b, s, lds_doubles)) return;
            if(!front_factor<HipTeam, true>(tm, V, b, s, pe_lds, lds_doubles - 2, 0, true, &cs))
            {
                if(tm.tid() == 0) atomicOr(V.flags + b, 4);
                return;
            }
        }
    }

    // ... and with 8: a level of 257..512 workgroups fills half of the wavefront slots with 4-wavefront workgroups; two 8-wavefront
    // workgroups per CU use all of them (128 instances x 4 fronts: the lower top levels of the 8-GPU share of the sweep)
    __global__ void __launch_bounds__(512, 4) k_m2_factor_top_mid(DevView V, int level, int nlev, int lds_doubles)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        HipTeam tm;
        for(int l = level; l < level + nlev; ++l)
        {
            int const s = V.top_list[V.top_ptr[l] + static_cast<int>(blockIdx.x)];
            // (a launch of levels marked 3 -- fronts formed against half a CU's LDS -- has that much: two of these workgroups per CU)
            if(lds_overrun(V, b, s, lds_doubles)) return;
            if(!front_factor(tm, V, b, s, pe_lds, lds_doubles - 2, 0, true))
            {
                if(tm.tid() == 0) atomicOr(V.flags + b, 4);
                return;
            }
        }
    }

    __global__ void __launch_bounds__(256) k_m2_winit(DevView V)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        GridTeam tm;
        double const* rhs = V.rhs + static_cast<long long>(b) * V.rows;
        double* w = V.w + static_cast<long long>(b) * V.rows;
        for(int k = tm.tid(); k < V.rows; k += tm.size()) w[k] = rhs[V.row_src[k]];
    }

    template <int MINW>
    __global__ void __launch_bounds__(PE_THREADS, MINW) k_m2_solve_parts(DevView V, int backward)
    {
        int const b = static_cast<int>(blockIdx.x);  // x = instance, y = part (see k_m2_factor_parts)
        if(!V.active[b]) return;
        HipTeam tm;
        if(backward) backward_part(tm, V, b, static_cast<int>(blockIdx.y), pe_lds);
        else
            forward_part(tm, V, b, static_cast<int>(blockIdx.y), pe_lds);
    }

    // the backward pass of the parts for 4-wavefront workgroups at 8 wavefronts per SIMD (64 VGPRs, lean LDS plan of
    // front_backward_lean): 8 workgroups per CU, i.e. the 4 096 workgroups of the 1 024-instance sweep in two rounds
    __global__ void __launch_bounds__(256, 8) k_m2_backward_parts(DevView V)
    {
        int const b = static_cast<int>(blockIdx.x);
        if(!V.active[b]) return;
        HipTeam tm;
        backward_part(tm, V, b, static_cast<int>(blockIdx.y), pe_lds);
    }

    template <int MINW>
    __global__ void __launch_bounds__(PE_THREADS, MINW) k_m2_solve_top(DevView V, int level, int nlev, int backward)
    {
        // levels level .. level + nlev - 1 (nlev > 1: single-front levels, see k_m2_factor_top); backward walks them downwards
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        HipTeam tm;
        for(int i = 0; i < nlev; ++i)
        {
            int const l = backward ? level + nlev - 1 - i : level + i;
            int const s = V.top_list[V.top_ptr[l] + static_cast<int>(blockIdx.x)];
            if(backward) front_backward(tm, V, b, s, pe_lds, V.max_m, V.lds_top_stage);
            else
                front_forward(tm, V, b, s, pe_lds, V.max_m, V.lds_top_stage);
        }
    }

    __global__ void __launch_bounds__(256) k_m2_finish(DevView V)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        GridTeam tm;
        double const* w = V.w + static_cast<long long>(b) * V.rows;
        double* x = V.x + static_cast<long long>(b) * V.rows;
        double const* xp = V.xprev + static_cast<long long>(b) * V.rows;
        int bits = 0;
        for(int k = tm.tid(); k < V.rows; k += tm.size())
        {
            // x[col_src[k]] = w[k]; the Newton test of that row is done right here (xprev was saved by k_m2_eval)
            int const r = V.col_src[k];
            double const xn = w[k];
            x[r] = xn;
            if(!(fabs(xn) <= 1.7976931348623157e308)) bits |= 1;
            bool const node = r < V.n_nodes;
            double const tol = (node ? V.v_abstol : V.i_abstol) + (node ? V.v_reltol : V.i_reltol) * fmax(fabs(xn), fabs(xp[r]));
            if(!(fabs(xn - xp[r]) <= tol)) bits |= 2;
        }
        if(bits) atomicOr(V.flags + b, bits);
    }

    // ---- residual safety net (pe_front.hpp residual_norms): the four norms of every active instance's last solve, combined over the
    // workgroups by atomic max on the (non-negative) doubles' bit patterns; keep_x != 0 also stores r and a copy of x for refinement
    __global__ void __launch_bounds__(256) k_m2_residual(DevView V, int keep_x)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        // only an iterate that is about to be ACCEPTED is checked: a non-linear instance whose Newton test (k_m2_finish, the launch
        // before this one) still shows a violation iterates again anyway -- one residual pass per instance and time point, not per
        // Newton iteration (its accumulators stay 0: "not above tolerance")
        if(V.nonlinear && !keep_x && (V.flags[b] & 2)) return;
        GridTeam tm;
        double n4[4];
        residual_norms(tm, V, b, keep_x ? V.rres + static_cast<long long>(b) * V.rows : nullptr, n4);
        if(keep_x)
        {
            double const* x = V.x + static_cast<long long>(b) * V.rows;
            double* xs = V.xsave + static_cast<long long>(b) * V.rows;
            for(int r = tm.tid(); r < V.rows; r += tm.size()) xs[r] = x[r];
        }
        WaveOps wo;
#pragma unroll
        for(int k = 0; k < 4; ++k)
        {
            double const m = wo.wave_max(n4[k]);
            if((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(V.eta_acc + 4 * b + k), static_cast<unsigned long long>(__double_as_longlong(m)));
        }
    }
    // x = (solution before the correction solve) + (correction, still permuted in w); clears the instance's accumulators for the re-check
    __global__ void __launch_bounds__(256) k_m2_refine_apply(DevView V)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        GridTeam tm;
        double const* w = V.w + static_cast<long long>(b) * V.rows;
        double const* xs = V.xsave + static_cast<long long>(b) * V.rows;
        double* x = V.x + static_cast<long long>(b) * V.rows;
        for(int k = tm.tid(); k < V.rows; k += tm.size())
        {
            int const r = V.col_src[k];
            x[r] = xs[r] + w[k];
        }
    }
    __global__ void __launch_bounds__(64) k_m2_clear_eta(DevView V)
    {
        int const b = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
        if(b < V.batch && V.active[b])
        {
            V.flags[b] = 0;
            for(int k = 0; k < 4; ++k) V.eta_acc[4 * b + k] = 0.0;
        }
    }
    // the non-finite / Newton test of k_m2_finish on the (refined) x of the active instances
    __global__ void __launch_bounds__(256) k_m2_retest(DevView V)
    {
        int const b = static_cast<int>(blockIdx.y);
        if(!V.active[b]) return;
        GridTeam tm;
        double const* x = V.x + static_cast<long long>(b) * V.rows;
        double const* xp = V.xprev + static_cast<long long>(b) * V.rows;
        int bits = 0;
        for(int r = tm.tid(); r < V.rows; r += tm.size())
        {
            double const xn = x[r];
            if(!(fabs(xn) <= 1.7976931348623157e308)) bits |= 1;
            bool const node = r < V.n_nodes;
            double const tol = (node ? V.v_abstol : V.i_abstol) + (node ? V.v_reltol : V.i_reltol) * fmax(fabs(xn), fabs(xp[r]));
            if(!(fabs(xn - xp[r]) <= tol)) bits |= 2;
        }
        if(bits) atomicOr(V.flags + b, bits);
    }

#ifndef PE_QUAD_WAVES
    #define PE_QUAD_WAVES 2  // wavefronts per SIMD the lane-group kernel is compiled for (<= 256 VGPRs: a front of order 32 alone takes 132)
#endif
    // ---- lane-group kernel of the wave fronts (pe_quad.hpp): the device execution model.  One lane's view; the 64 lanes of the
    // wavefront run in lockstep under wavefront-uniform control flow, so every cross-lane read is a DPP broadcast inside a row of 16.
    extern "C" __device__ double pe_update_dpp_f64(double, double, int, int, int, bool) __asm("llvm.amdgcn.update.dpp.f64");
    struct QuadDev
    {
        using vd = double;
        using vi = int;
        using vu = unsigned;
        using vm = bool;
        static __device__ __forceinline__ vi lane() { return static_cast<int>(threadIdx.x) & 63; }
        static __device__ __forceinline__ vu to_u(vi v) { return static_cast<unsigned>(v); }
        // DPP row_newbcast:K (dpp_ctrl 0x150 + K): lane K of each row of 16 lanes to the whole row -- v_mov_b64_dpp, one instruction
        template <int K>
        static __device__ __forceinline__ vd bc(vd v)
        {
            return pe_update_dpp_f64(v, v, 0x150 + K, 0xf, 0xf, true);
        }
        static __device__ __forceinline__ vd bcast(vd v, int k)  // k is a constant after unrolling: the switch folds
        {
            switch(k)
            {
                case 0: return bc<0>(v);
                case 1: return bc<1>(v);
                case 2: return bc<2>(v);
                case 3: return bc<3>(v);
                case 4: return bc<4>(v);
                case 5: return bc<5>(v);
                case 6: return bc<6>(v);
                case 7: return bc<7>(v);
                case 8: return bc<8>(v);
                case 9: return bc<9>(v);
                case 10: return bc<10>(v);
                case 11: return bc<11>(v);
                case 12: return bc<12>(v);
                case 13: return bc<13>(v);
                case 14: return bc<14>(v);
                default: return bc<15>(v);
            }
        }
        static __device__ __forceinline__ vd ld(char const* base, vu off) { return *reinterpret_cast<double const*>(base + off); }
        static __device__ __forceinline__ void ld_u32x4(unsigned char const* base, vu off, vu* out)
        {
            uint4 const t = *reinterpret_cast<uint4 const*>(base + off);
            out[0] = t.x;
            out[1] = t.y;
            out[2] = t.z;
            out[3] = t.w;
        }
        static __device__ __forceinline__ vi ld_i32(int const* base, vu off) { return *reinterpret_cast<int const*>(reinterpret_cast<char const*>(base) + off); }
        static __device__ __forceinline__ void st(char* base, vu off, vd v) { *reinterpret_cast<double*>(base + off) = v; }
        static __device__ __forceinline__ void st_if(bool all, vm mask, char* base, vu off, vd v)
        {
            if(all || mask) *reinterpret_cast<double*>(base + off) = v;
        }
        template <class F>
        static __device__ __forceinline__ void when(vm mask, F&& body)  // one exec region for all the stores of `body`
        {
            if(mask) body();
        }
        // the wavefront's LDS (one wavefront per workgroup): update-matrix stack of the quad
        static __device__ __forceinline__ vd lds_ld(vu addr) { return *reinterpret_cast<double const*>(reinterpret_cast<char const*>(pe_lds) + addr); }
        static __device__ __forceinline__ void lds_st(vu addr, vd v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(pe_lds) + addr) = v; }
        static __device__ __forceinline__ void lds_fence() { WaveOps{}.wave_fence_lds(); }
        static __device__ __forceinline__ void fence() { WaveOps{}.wave_fence(); }
        template <class T>
        static __device__ __forceinline__ T sel(vm m, T a, T b)
        {
            return m ? a : b;
        }
        static __device__ __forceinline__ vd rcp(vd d) { return WaveOps{}.rcp(d); }
        static __device__ __forceinline__ vd fma(vd a, vd b, vd c) { return __builtin_fma(a, b, c); }
        static __device__ __forceinline__ vm bad(vd piv) { return piv == 0.0 || !(fabs(piv) <= 1.7976931348623157e308); }
        static __device__ __forceinline__ vm none() { return false; }
        static __device__ __forceinline__ long long clock() { return static_cast<long long>(wall_clock64()); }
        static __device__ __forceinline__ long long clock(vd dep)  // read behind the instruction that produced `dep`
        {
            asm volatile("" ::"v"(dep));
            return static_cast<long long>(wall_clock64());
        }
        static __device__ __forceinline__ void prof(long long* dst, long long const* v, int n)
        {
            if((threadIdx.x & 63) == 0)
                for(int k = 0; k < n; ++k) dst[k] += v[k];
        }
        static __device__ __forceinline__ void flag(int* f, vi idx, int bits, vm mask)
        {
            if(mask) atomicOr(f + idx, bits);
        }
    };

    // grid = quads x lists, one wavefront each.  Consecutive workgroups go round-robin to the 8 XCDs: the lists are dealt out so
    // that one XCD works on L / 8 of them -- their shared tables (row tables, child maps, metadata) stay in that XCD's L2.
    __global__ void __launch_bounds__(64, PE_QUAD_WAVES) k_m2_factor_quads(DevView V)
    {
        int const n = static_cast<int>(blockIdx.x), L = V.n_parts * V.n_waves;
        int quad, list;
        if((L & 7) == 0)
        {
            int const lpx = L >> 3;
            list = (n & 7) * lpx + (n >> 3) % lpx;
            quad = n / L;
        }
        else
        {
            list = n % L;
            quad = n / L;
        }
        quad_factor_list<QuadDev>(V, quad, list);
    }

    // backward pass of the quad fronts: after k_m2_backward_parts (cooperative + per-instance wave fronts of the parts)
    __global__ void __launch_bounds__(64, 4) k_m2_backward_quads(DevView V)
    {
        int const n = static_cast<int>(blockIdx.x), L = V.n_parts * V.n_waves;
        int quad, list;
        if((L & 7) == 0)
        {
            int const lpx = L >> 3;
            list = (n & 7) * lpx + (n >> 3) % lpx;
            quad = n / L;
        }
        else
        {
            list = n % L;
            quad = n / L;
        }
        quad_backward_list<QuadDev>(V, quad, list);
    }

    // the MID fronts of the same quads (row sets up to four, columns in blocks): after the wave fronts, before the cooperative parts
#ifndef PE_MID_WAVES
    #define PE_MID_WAVES 2
#endif
    __global__ void __launch_bounds__(64, PE_MID_WAVES) k_m2_factor_mid(DevView V)
    {
        int const n = static_cast<int>(blockIdx.x), L = V.n_parts * V.n_waves;
        int quad, list;
        if((L & 7) == 0)
        {
            int const lpx = L >> 3;
            list = (n & 7) * lpx + (n >> 3) % lpx;
            quad = n / L;
        }
        else
        {
            list = n % L;
            quad = n / L;
        }
        quad_factor_mid_list<QuadDev>(V, quad, list);
    }

    // hipFuncAttributeMaxDynamicSharedMemorySize is a per-function upper bound: raised when a launch needs more than any launch
    // before it, never per launch (a Newton iteration of the split schedule is ~16 launches; engines on several host threads
    // share the table)
    static hipError_t set_lds(void const* fn, size_t bytes)
    {
        // (the attribute is per DEVICE: engines on several GPUs in one process each raise it on their own device)
        static std::mutex mu;
        static std::unordered_map<unsigned long long, size_t> allowed;
        int dev = 0;
        if(hipError_t const e = hipGetDevice(&dev); e != hipSuccess) return e;
        unsigned long long const key = (static_cast<unsigned long long>(reinterpret_cast<uintptr_t>(fn)) * 1000003ull) ^ static_cast<unsigned long long>(dev + 1);
        std::lock_guard<std::mutex> lock(mu);
        auto it = allowed.find(key);
        if(it != allowed.end() && bytes <= it->second) return hipSuccess;
        hipError_t const e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        if(e == hipSuccess) allowed[key] = bytes;
        return e;
    }

    hipError_t launch_tr_steps(hipStream_t st, DevView const& V, double dt, int nsteps, bool reuse)
    {
        size_t const lds = static_cast<size_t>(V.lds_doubles) * sizeof(double);
        auto const fn = V.high_occupancy ? &k_tr_steps<4> : &k_tr_steps<2>;
        hipError_t e = set_lds(reinterpret_cast<void const*>(fn), lds);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(V.batch), dim3(V.n_waves * 64), lds, st, V, dt, nsteps, reuse ? 1 : 0);
        return hipGetLastError();
    }

    hipError_t launch_dc_point(hipStream_t st, DevView const& V, int mode)
    {
        size_t const lds = static_cast<size_t>(V.lds_doubles) * sizeof(double);
        auto const fn = V.high_occupancy ? &k_dc_point<4> : &k_dc_point<2>;
        hipError_t e = set_lds(reinterpret_cast<void const*>(fn), lds);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3(V.batch), dim3(V.n_waves * 64), lds, st, V, mode);
        return hipGetLastError();
    }

    // one Newton iteration of every active instance in the split schedule: stamp -> LU -> solves -> Newton bits.
    // (no launch reads the environment: every knob arrives in the view, set per engine by pe_engine_policy.cpp)
    // workgroups per instance of the elementwise kernels (eval, stamp, winit, finish, companion): ~2048 rows each for large
    // batches; few instances spread over more workgroups (these kernels are gathers: latency-bound at low occupancy)
    static int grid_per_instance(DevView const& V)
    {
        if(V.ew_grid > 0) return V.ew_grid;  // knob EW_GRID of this engine (sweeps): workgroups per instance, as given
        // (round 4: 1 280 workgroups in all -- ten per instance at 128 instances, 1.376 against 1.395 ms per iteration; 256 instances and more
        //  are at rows / 2 048 either way)
        int const by_rows = (V.rows + 2047) / 2048, fine = (V.rows + 255) / 256, want = 1280 / (V.batch > 0 ? V.batch : 1);
        int g = by_rows > (want < fine ? want : fine) ? by_rows : (want < fine ? want : fine);
        return g < 1 ? 1 : (g > 64 ? 64 : g);
    }

    // ev0 / ev1 (may be null): HIP events recorded around the dominant launch (k_m2_factor_parts, or the backward
    // k_m2_solve_parts when the factors are reused) for the per-kernel roofline of bench.py.
    template <int MINW>
    // stamp_mode: 0 full stamp, 1 x-dependent slots / rows only (later Newton iterations of a point), 2 x-dependent matrix slots + full right-hand
    // side (first iteration of a transient step at an unchanged dt: the rest of the matrix is last step's)
    static hipError_t m2_sequence(hipStream_t st, DevView const& V, int mode, double t, double last_step, bool do_factor, hipEvent_t ev0, hipEvent_t ev1, bool refine,
                                  int stamp_mode = 0, bool companion = false, double companion_dt = 0.0)
    {
        bool const have_lists = V.dyn_a && V.dyn_b;
        int const eval_dyn = (stamp_mode == 1 && have_lists) ? 1 : 0, stamp_dyn = have_lists ? stamp_mode : 0;
        size_t const lds = static_cast<size_t>(V.lds_doubles) * sizeof(double);
        size_t const lds_s = static_cast<size_t>(V.lds_solve_doubles) * sizeof(double);
        size_t const lds_st = static_cast<size_t>(V.lds_solve_top_doubles) * sizeof(double);  // (top fronts may carry more pivots: larger staged block)
        {
            hipError_t e = set_lds(reinterpret_cast<void const*>(&k_m2_factor_parts<MINW>), lds);
            if(e == hipSuccess) e = set_lds(reinterpret_cast<void const*>(&k_m2_factor_top<MINW>), lds);
            if(e == hipSuccess) e = set_lds(reinterpret_cast<void const*>(&k_m2_factor_top_wide), static_cast<size_t>(V.lds_top_doubles) * sizeof(double));
            if(e == hipSuccess) e = set_lds(reinterpret_cast<void const*>(&k_m2_factor_top_mid), std::max(lds, static_cast<size_t>(V.lds_mid_doubles) * sizeof(double)));
            if(e == hipSuccess) e = set_lds(reinterpret_cast<void const*>(&k_m2_solve_parts<MINW>), lds_s);
            if(e == hipSuccess) e = set_lds(reinterpret_cast<void const*>(&k_m2_solve_top<MINW>), lds_st);
            if(e != hipSuccess) return e;
        }
        int const B = V.batch, T = V.n_waves * 64;
        int const G = grid_per_instance(V);
        // runs of single-front top levels share a launch
        // (... of the same launch class: a level whose fronts were formed against a CU's whole / half LDS must run on the launch that
        //  has that LDS -- pe_top_plan.hpp)
        auto run = [&](int l) { return top_run(V, l); };
        auto run_down = [&](int l)
        {
            int n = 1;
            if(V.top_cnt[l] == 1)
                while(l - n >= 0 && V.top_cnt[l - n] == 1) ++n;
            return n;
        };
        if(!refine)
        {
            hipLaunchKernelGGL(k_m2_eval, dim3(G, B), dim3(256), 0, st, V, mode, t, last_step, eval_dyn, companion ? 1 : 0, companion_dt);
            hipLaunchKernelGGL(k_m2_stamp, dim3(G, B), dim3(256), 0, st, V, stamp_dyn);  // (+ w = P rhs)
        }
        else  // (refinement: V arrives with rhs = the residual of the solve being corrected; the matrix values are still assembled)
            hipLaunchKernelGGL(k_m2_winit, dim3(G, B), dim3(256), 0, st, V);
        if(do_factor)
        {
            // the factorisation carries the right-hand side along (fused forward substitution): no forward launches
            if(ev0) (void)hipEventRecord(ev0, st);
            // the wave fronts of four instances per wavefront (pe_quad.hpp); k_m2_factor_parts then runs the cooperative fronts only
            if(V.quad && V.n_quads > 0)
            {
                // (knob QUAD_LDS of this engine: an LDS request that limits the wavefronts per CU -- occupancy probe)
                size_t const qlds = std::max(static_cast<size_t>(std::max(0, V.quad_lds_pad)), static_cast<size_t>(V.q_lds_stride) * 32);  // four instance stacks
                if(qlds > 0)
                {
                    hipError_t const e = set_lds(reinterpret_cast<void const*>(&k_m2_factor_quads), qlds);
                    if(e != hipSuccess) return e;
                }
                hipLaunchKernelGGL(k_m2_factor_quads, dim3(V.n_quads * V.n_parts * V.n_waves), dim3(64), qlds, st, V);
                if(V.n_mid > 0) hipLaunchKernelGGL(k_m2_factor_mid, dim3(V.n_quads * V.n_parts * V.n_waves), dim3(64), 0, st, V);
            }
            hipLaunchKernelGGL(k_m2_factor_parts<MINW>, dim3(B, V.n_parts), dim3(T), lds, st, V);
            if(ev1) (void)hipEventRecord(ev1, st);
            // 16 wavefronts per front where a level leaves most CUs without a workgroup anyway: always in the one-workgroup-per-CU
            // geometry (few instances), and on the under-filled levels near the root of a sweep (fronts x instances <= CUs + 25 %);
            // the rule lives in upload_symbolic (the fronts' LDS layout depends on it), the plan in pe_top_plan.hpp
            // (knob MID_TOP of this engine: V.mid_top_limit; the kernels take the launch's LDS as their cap -- never a field of V)
            for_each_top_launch(V, B, MINW == 4 && T == 256, V.mid_top_limit,
                                [&](TopLaunch const& t)
                                {
                                    dim3 const grid(V.top_cnt[t.level], B);
                                    size_t const bytes = static_cast<size_t>(t.lds_doubles) * sizeof(double);
                                    int const cap = static_cast<int>(t.lds_doubles);
                                    if(t.kind == 1) hipLaunchKernelGGL(k_m2_factor_top_wide, grid, dim3(1024), bytes, st, V, t.level, t.nlev, cap);
                                    else if(t.kind != 0)
                                        hipLaunchKernelGGL(k_m2_factor_top_mid, grid, dim3(512), bytes, st, V, t.level, t.nlev, cap);
                                    else
                                        hipLaunchKernelGGL(k_m2_factor_top<MINW>, grid, dim3(T), bytes, st, V, t.level, t.nlev, cap);
                                });
        }
        else
        {
            hipLaunchKernelGGL(k_m2_solve_parts<MINW>, dim3(B, V.n_parts), dim3(T), lds_s, st, V, 0);
            for(int l = 0; l < V.n_top_levels; l += run(l)) hipLaunchKernelGGL(k_m2_solve_top<MINW>, dim3(V.top_cnt[l], B), dim3(T), lds_st, st, V, l, run(l), 0);
        }
        for(int l = V.n_top_levels - 1; l >= 0;)
        {
            int const n = run_down(l);
            hipLaunchKernelGGL(k_m2_solve_top<MINW>, dim3(V.top_cnt[l], B), dim3(T), lds_st, st, V, l - n + 1, n, 1);
            l -= n;
        }
        if(!do_factor && ev0) (void)hipEventRecord(ev0, st);
        // (the backward pass of the parts runs on the lean LDS plan: more workgroups per CU)
        size_t const lds_b = static_cast<size_t>(V.lds_solve_b_doubles) * sizeof(double);
        if(T <= 256)
        {
            hipError_t const e = set_lds(reinterpret_cast<void const*>(&k_m2_backward_parts), lds_b);
            if(e != hipSuccess) return e;
            hipLaunchKernelGGL(k_m2_backward_parts, dim3(B, V.n_parts), dim3(T), lds_b, st, V);
        }
        else
            hipLaunchKernelGGL(k_m2_solve_parts<MINW>, dim3(B, V.n_parts), dim3(T), lds_b, st, V, 1);
        if(V.quad_back && V.n_quads > 0) hipLaunchKernelGGL(k_m2_backward_quads, dim3(V.n_quads * V.n_parts * V.n_waves), dim3(64), 0, st, V);
        if(!do_factor && ev1) (void)hipEventRecord(ev1, st);
        if(!refine)
        {
            hipLaunchKernelGGL(k_m2_finish, dim3(G, B), dim3(256), 0, st, V);
            if(V.residual_tol > 0.0) hipLaunchKernelGGL(k_m2_residual, dim3(G, B), dim3(256), 0, st, V, 0);
        }
        return hipGetLastError();
    }

    hipError_t launch_m2_iteration(hipStream_t st, DevView const& V, int mode, double t, double last_step, bool do_factor, hipEvent_t ev0, hipEvent_t ev1,
                                   int stamp_mode, bool companion, double companion_dt)
    {
        return V.high_occupancy ? m2_sequence<4>(st, V, mode, t, last_step, do_factor, ev0, ev1, false, stamp_mode, companion, companion_dt)
                                : m2_sequence<2>(st, V, mode, t, last_step, do_factor, ev0, ev1, false, stamp_mode, companion, companion_dt);
    }

    // One round of iterative refinement of the active instances' last solve (same matrix values, same pivot order):
    // r = b - A x -> correction solve A d = r (a full refactorisation with r riding along: the fused path keeps no L21) -> x += d,
    // then the norms of the corrected solve and its Newton / finiteness bits.
    hipError_t launch_m2_refine(hipStream_t st, DevView const& V)
    {
        int const B = V.batch, G = grid_per_instance(V);
        hipLaunchKernelGGL(k_m2_residual, dim3(G, B), dim3(256), 0, st, V, 1);
        DevView Vr = V;
        Vr.rhs = V.rres;
        hipError_t const e = V.high_occupancy ? m2_sequence<4>(st, Vr, 0, 0.0, 0.0, true, nullptr, nullptr, true) : m2_sequence<2>(st, Vr, 0, 0.0, 0.0, true, nullptr, nullptr, true);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(k_m2_refine_apply, dim3(G, B), dim3(256), 0, st, V);
        hipLaunchKernelGGL(k_m2_clear_eta, dim3((B + 63) / 64), dim3(64), 0, st, V);
        hipLaunchKernelGGL(k_m2_residual, dim3(G, B), dim3(256), 0, st, V, 0);
        hipLaunchKernelGGL(k_m2_retest, dim3(G, B), dim3(256), 0, st, V);
        return hipGetLastError();
    }

    // ---- results of one Newton iteration handed to the host WITHOUT a copy command: the flag word and the four residual norms of every
    // instance go straight into pinned host memory, then a sequence number (system-scope release) -- the host polls that word instead of
    // issuing hipMemcpyAsync + hipStreamSynchronize (two driver calls and a completion signal: 25-35 us of a 0.5 ms iteration of a single
    // circuit, pe_engine_newton.cpp wait_published).  One small workgroup, last launch of the iteration.
    __global__ void __launch_bounds__(256) k_m2_publish(int const* __restrict__ flags, double const* __restrict__ eta, int batch, int* pub_flags, double* pub_eta,
                                                        unsigned long long* pub_seq, unsigned long long seq)
    {
        for(int b = static_cast<int>(threadIdx.x); b < batch; b += static_cast<int>(blockDim.x)) pub_flags[b] = flags[b];
        if(eta)
            for(int i = static_cast<int>(threadIdx.x); i < 4 * batch; i += static_cast<int>(blockDim.x)) pub_eta[i] = eta[i];
        __threadfence_system();
        __syncthreads();
        if(threadIdx.x == 0) __hip_atomic_store(pub_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    hipError_t launch_m2_publish(hipStream_t st, DevView const& V, int* pub_flags, double* pub_eta, unsigned long long* pub_seq, unsigned long long seq)
    {
        hipLaunchKernelGGL(k_m2_publish, dim3(1), dim3(256), 0, st, V.flags, V.residual_tol > 0.0 ? V.eta_acc : nullptr, V.batch, pub_flags, pub_eta, pub_seq, seq);
        return hipGetLastError();
    }

    // ---- captured launch sequences (hipGraph) of the split schedule.  One Newton iteration of a small sweep is 15-20 short launches; enqueued one
    // by one the host call rate (3-5 us per launch) is of the order of the kernels themselves (3-15 us for the elementwise and top-solve launches
    // of a single circuit) and the stream runs dry between them.  The sequence of an iteration is fixed by (analysis mode, factor / reuse, full /
    // x-dependent stamp, companion update yes / no) and the view V: it is captured ONCE per such key into a graph and replayed with one
    // hipGraphLaunch; what changes from iteration to iteration -- t, last_step, the companion's dt (arguments of k_m2_eval) and the sequence number
    // of k_m2_publish -- is patched into those two nodes (hipGraphExecKernelNodeSetParams).  The instances that still iterate are device data
    // (V.active, the quad list): the captured grids are those of the full sweep, finished instances return at once.
    struct M2GraphEntry
    {
        hipGraph_t graph{};
        hipGraphExec_t exec{};
        hipGraphNode_t eval_node{}, publish_node{};
        DevView V{};
        int mode{}, do_factor{}, dyn{}, companion{};
        int* pub_flags{};
        double* pub_eta{};
        unsigned long long* pub_seq{};
    };
    struct M2GraphCache
    {
        std::vector<M2GraphEntry> entries;
        ~M2GraphCache() { clear(); }
        void clear()
        {
            for(auto& e: entries)
            {
                if(e.exec) (void)hipGraphExecDestroy(e.exec);
                if(e.graph) (void)hipGraphDestroy(e.graph);
            }
            entries.clear();
        }
    };
    M2GraphCache* m2_graphs_create() { return new M2GraphCache; }
    void m2_graphs_destroy(M2GraphCache* c) { delete c; }
    void m2_graphs_clear(M2GraphCache* c)
    {
        if(c) c->clear();
    }

    hipError_t launch_m2_iteration_graph(hipStream_t st, M2GraphCache* cache, DevView const& V, int mode, double t, double last_step, bool do_factor, int stamp_mode,
                                         bool companion, double companion_dt, int* pub_flags, double* pub_eta, unsigned long long* pub_seq, unsigned long long seq)
    {
        int const dyn = (V.dyn_a && V.dyn_b) ? stamp_mode : 0;
        M2GraphEntry* hit = nullptr;
        for(auto& e: cache->entries)
            if(e.mode == mode && e.do_factor == (do_factor ? 1 : 0) && e.dyn == dyn && e.companion == (companion ? 1 : 0) && e.pub_flags == pub_flags && e.pub_eta == pub_eta &&
               e.pub_seq == pub_seq && std::memcmp(&e.V, &V, sizeof(DevView)) == 0)
            {
                hit = &e;
                break;
            }
        if(!hit)
        {
            if(cache->entries.size() >= 16) cache->clear();  // (a view that keeps changing: start over rather than grow)
            M2GraphEntry e;
            std::memcpy(&e.V, &V, sizeof(DevView));  // (bytes, padding included: the key is compared with memcmp)
            e.mode = mode;
            e.do_factor = do_factor ? 1 : 0;
            e.dyn = dyn;
            e.companion = companion ? 1 : 0;
            e.pub_flags = pub_flags;
            e.pub_eta = pub_eta;
            e.pub_seq = pub_seq;
            hipError_t rc = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            if(rc != hipSuccess) return rc;
            rc = V.high_occupancy ? m2_sequence<4>(st, V, mode, t, last_step, do_factor, nullptr, nullptr, false, stamp_mode, companion, companion_dt)
                                  : m2_sequence<2>(st, V, mode, t, last_step, do_factor, nullptr, nullptr, false, stamp_mode, companion, companion_dt);
            hipError_t const rp = launch_m2_publish(st, V, pub_flags, pub_eta, pub_seq, seq);
            hipError_t const re = hipStreamEndCapture(st, &e.graph);
            if(rc != hipSuccess) return rc;
            if(rp != hipSuccess) return rp;
            if(re != hipSuccess) return re;
            rc = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
            if(rc != hipSuccess)
            {
                (void)hipGraphDestroy(e.graph);
                return rc;
            }
            size_t n = 0;
            rc = hipGraphGetNodes(e.graph, nullptr, &n);
            if(rc != hipSuccess) return rc;
            std::vector<hipGraphNode_t> nodes(n);
            rc = hipGraphGetNodes(e.graph, nodes.data(), &n);
            if(rc != hipSuccess) return rc;
            for(hipGraphNode_t nd: nodes)
            {
                hipGraphNodeType ty{};
                if(hipGraphNodeGetType(nd, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
                hipKernelNodeParams kp{};
                if(hipGraphKernelNodeGetParams(nd, &kp) != hipSuccess) continue;
                if(kp.func == reinterpret_cast<void*>(&k_m2_eval)) e.eval_node = nd;
                else if(kp.func == reinterpret_cast<void*>(&k_m2_publish))
                    e.publish_node = nd;
            }
            if(!e.eval_node || !e.publish_node)
            {
                (void)hipGraphExecDestroy(e.exec);
                (void)hipGraphDestroy(e.graph);
                return hipErrorInvalidValue;
            }
            cache->entries.push_back(e);
            hit = &cache->entries.back();
        }
        // this iteration's scalars into the two nodes that take them
        {
            DevView v = V;
            int md = mode, dy = dyn == 1 ? 1 : 0, cs = companion ? 1 : 0;  // (k_m2_eval: only mode 1 skips the x-independent values)
            double tt = t, ls = last_step, cd = companion_dt;
            void* args[] = {&v, &md, &tt, &ls, &dy, &cs, &cd};
            hipKernelNodeParams kp{};
            hipError_t rc = hipGraphKernelNodeGetParams(hit->eval_node, &kp);
            if(rc != hipSuccess) return rc;
            kp.kernelParams = args;
            kp.extra = nullptr;
            rc = hipGraphExecKernelNodeSetParams(hit->exec, hit->eval_node, &kp);
            if(rc != hipSuccess) return rc;
        }
        {
            int const* fl = V.flags;
            double const* eta = V.residual_tol > 0.0 ? V.eta_acc : nullptr;
            int batch = V.batch;
            int* pf = pub_flags;
            double* pe_ = pub_eta;
            unsigned long long* ps = pub_seq;
            unsigned long long sq = seq;
            void* args[] = {&fl, &eta, &batch, &pf, &pe_, &ps, &sq};
            hipKernelNodeParams kp{};
            hipError_t rc = hipGraphKernelNodeGetParams(hit->publish_node, &kp);
            if(rc != hipSuccess) return rc;
            kp.kernelParams = args;
            kp.extra = nullptr;
            rc = hipGraphExecKernelNodeSetParams(hit->exec, hit->publish_node, &kp);
            if(rc != hipSuccess) return rc;
        }
        return hipGraphLaunch(hit->exec, st);
    }

    hipError_t launch_m2_companion(hipStream_t st, DevView const& V, double dt)
    {
        int const G = grid_per_instance(V);
        hipLaunchKernelGGL(k_m2_companion, dim3(G, V.batch), dim3(256), 0, st, V, dt);
        return hipGetLastError();
    }

    // ---- sweep statistics: x[batch][rows] -> {sum, sum of squares, min, max}[rows].  Thread = row (coalesced over rows), grid.y =
    // chunks of instances; a second pass combines the chunks in order (no atomics: bitwise reproducible).
    __global__ void __launch_bounds__(256) k_sweep_stats_partial(double const* __restrict__ x, int rows, int batch, int chunk_len, double* __restrict__ partial)
    {
        int const r = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
        if(r >= rows) return;
        int const c = static_cast<int>(blockIdx.y), b0 = c * chunk_len, b1 = b0 + chunk_len < batch ? b0 + chunk_len : batch;
        double s = 0.0, q = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
        double const* p = x + static_cast<long long>(b0) * rows + r;
        int b = b0;
        for(; b + 4 <= b1; b += 4, p += 4ll * rows)
        {
            double const v0 = p[0], v1 = p[rows], v2 = p[2ll * rows], v3 = p[3ll * rows];
            s += v0; q += v0 * v0; mn = fmin(mn, v0); mx = fmax(mx, v0);
            s += v1; q += v1 * v1; mn = fmin(mn, v1); mx = fmax(mx, v1);
            s += v2; q += v2 * v2; mn = fmin(mn, v2); mx = fmax(mx, v2);
            s += v3; q += v3 * v3; mn = fmin(mn, v3); mx = fmax(mx, v3);
        }
        for(; b < b1; ++b, p += rows)
        {
            double const v = *p;
            s += v; q += v * v; mn = fmin(mn, v); mx = fmax(mx, v);
        }
        double* o = partial + static_cast<long long>(c) * 4 * rows + r;
        o[0] = s;
        o[rows] = q;
        o[2ll * rows] = mn;
        o[3ll * rows] = mx;
    }
    __global__ void __launch_bounds__(256) k_sweep_stats_final(double const* __restrict__ partial, int rows, int n_chunks, double* __restrict__ out)
    {
        int const r = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
        if(r >= rows) return;
        double s = 0.0, q = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
        for(int c = 0; c < n_chunks; ++c)
        {
            double const* p = partial + static_cast<long long>(c) * 4 * rows + r;
            s += p[0];
            q += p[rows];
            mn = fmin(mn, p[2ll * rows]);
            mx = fmax(mx, p[3ll * rows]);
        }
        out[r] = s;
        out[rows + r] = q;
        out[2ll * rows + r] = mn;
        out[3ll * rows + r] = mx;
    }
    hipError_t launch_sweep_statistics(hipStream_t st, DevView const& V, int n_chunks, double* partial, double* out)
    {
        if(V.rows <= 0 || V.batch <= 0) return hipSuccess;
        int const chunk_len = (V.batch + n_chunks - 1) / n_chunks;
        dim3 const grid((V.rows + 255) / 256, n_chunks);
        hipLaunchKernelGGL(k_sweep_stats_partial, grid, dim3(256), 0, st, V.x, V.rows, V.batch, chunk_len, partial);
        hipLaunchKernelGGL(k_sweep_stats_final, dim3(grid.x), dim3(256), 0, st, partial, V.rows, n_chunks, out);
        return hipGetLastError();
    }

    // ---- small-signal AC: iterative refinement of a frequency point entirely on the device (pe_engine_ac.cpp pe_hip_analyze_ac)
    __global__ void __launch_bounds__(256) k_ac_residual(DevView V, double const* __restrict__ xacc, double const* __restrict__ b0, int rhs0, double* worst)
    {
        int const b = static_cast<int>(blockIdx.y);
        double const w = WaveOps{}.wave_max(ac_residual(GridTeam{}, V, b, xacc, b0, rhs0));
        // (non-negative doubles order like their bit patterns; a NaN residual has the largest pattern: the host sees it)
        if((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(worst), static_cast<unsigned long long>(__double_as_longlong(fabs(w))));
    }
    __global__ void __launch_bounds__(256) k_ac_accumulate(DevView V, double* __restrict__ xacc, double* __restrict__ b0, int first)
    {
        size_t const n = static_cast<size_t>(V.batch) * V.rows;
        for(size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        {
            if(first)
            {
                xacc[i] = V.x[i];
                b0[i] = V.rhs[i];
            }
            else
                xacc[i] += V.x[i];
        }
    }
    hipError_t launch_ac_residual(hipStream_t st, DevView const& V, double const* xacc, double const* b0, int rhs0, double* worst)
    {
        hipError_t const e = hipMemsetAsync(worst, 0, sizeof(double), st);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(k_ac_residual, dim3(grid_per_instance(V), V.batch), dim3(256), 0, st, V, xacc, b0, rhs0, worst);
        return hipGetLastError();
    }
    hipError_t launch_ac_accumulate(hipStream_t st, DevView const& V, double* xacc, double* b0, bool first)
    {
        size_t const n = static_cast<size_t>(V.batch) * V.rows;
        int const g = static_cast<int>(std::min<size_t>(1024, (n + 255) / 256));
        hipLaunchKernelGGL(k_ac_accumulate, dim3(g > 0 ? g : 1), dim3(256), 0, st, V, xacc, b0, first ? 1 : 0);
        return hipGetLastError();
    }

    // ---- complex twin of the solver seam (pe_engine_seam.cpp pe_hip_solve_csr_complex): refinement residual of the real-equivalent system
    // kept in CSR order -- r = b0 - A xacc straight into V.rhs (what the correction solve permutes into w), *worst = max componentwise
    // backward error |r_i| / (|b_i| + sum_j |a_ij x_j|)
    __global__ void __launch_bounds__(256) k_csr_residual(DevView V, double const* __restrict__ xacc, double const* __restrict__ b0, double* worst)
    {
        GridTeam tm;
        double w = 0.0;
        for(int r = tm.tid(); r < V.rows; r += tm.size())
        {
            double acc = b0[r], mag = fabs(acc);
            int const e1 = V.csr_rp[r + 1];
            for(int e = V.csr_rp[r]; e < e1; ++e)
            {
                double const t = V.aval[e] * xacc[V.csr_ci[e]];
                acc -= t;
                mag += fabs(t);
            }
            V.rhs[r] = acc;
            w = fmax(w, fabs(acc) / (mag > 0.0 ? mag : 1.0));
        }
        w = WaveOps{}.wave_max(w);
        if((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(worst), static_cast<unsigned long long>(__double_as_longlong(fabs(w))));
    }
    hipError_t launch_csr_residual(hipStream_t st, DevView const& V, double const* xacc, double const* b0, double* worst)
    {
        hipError_t const e = hipMemsetAsync(worst, 0, sizeof(double), st);
        if(e != hipSuccess) return e;
        int const g = std::max(1, std::min(64, (V.rows + 255) / 256));
        hipLaunchKernelGGL(k_csr_residual, dim3(g), dim3(256), 0, st, V, xacc, b0, worst);
        return hipGetLastError();
    }

    // ---- on-box HBM ceiling: device-to-device stream copy, 16 B per lane.  Shape from scripts/copy_sweep.hip (round 3): every
    // workgroup owns a contiguous chunk, eight loads in flight per lane, non-temporal loads and stores -- 5.2-5.3 TB/s on these boxes
    // (grid-stride with four in flight, round 2: 4.4-4.7; hipMemcpyDtoD: 5.0; the guide's float4 copy: 6.29, MI355X_MICROARCH.md:36)
    using v4f_t = __attribute__((ext_vector_type(4))) float;
    __global__ void __launch_bounds__(256) k_stream_copy(v4f_t const* __restrict__ src, v4f_t* __restrict__ dst, size_t n)
    {
        size_t const per = (n + gridDim.x - 1) / gridDim.x;
        size_t const lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
        for(size_t base = lo; base < hi; base += 8 * static_cast<size_t>(blockDim.x))
        {
            v4f_t v[8];
#pragma unroll
            for(int q = 0; q < 8; ++q)
            {
                size_t const i = base + q * blockDim.x + threadIdx.x;
                if(i < hi) v[q] = __builtin_nontemporal_load(src + i);
            }
#pragma unroll
            for(int q = 0; q < 8; ++q)
            {
                size_t const i = base + q * blockDim.x + threadIdx.x;
                if(i < hi) __builtin_nontemporal_store(v[q], dst + i);
            }
        }
    }
    hipError_t launch_stream_copy(hipStream_t st, void const* src, void* dst, size_t bytes)
    {
        hipLaunchKernelGGL(k_stream_copy, dim3(256 * 32), dim3(256), 0, st, static_cast<v4f_t const*>(src), static_cast<v4f_t*>(dst), bytes / 16);
        return hipGetLastError();
    }

    hipError_t launch_factor_solve(hipStream_t st, DevView const& V, bool do_factor)
    {
        size_t const lds = static_cast<size_t>(V.lds_doubles) * sizeof(double);
        hipError_t e = set_lds(reinterpret_cast<void const*>(&k_factor_solve), lds);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(k_factor_solve, dim3(V.batch), dim3(V.n_waves * 64), lds, st, V, do_factor ? 1 : 0);
        return hipGetLastError();
    }
}  // namespace pe
