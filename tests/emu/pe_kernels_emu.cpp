// tests/emu/pe_kernels_emu.cpp -- TEST INFRASTRUCTURE ONLY: runs the team-generic code of pe_front.hpp on the host
// with a one-thread team per instance and one-lane "wavefronts" executed one after the other (see
// hip_shim/hip/hip_runtime_api.h).  Checks indexing and the orchestration; says nothing about races or performance.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pe_front.hpp"
#include "pe_kernels.hpp"
#include "pe_quad.hpp"
#include "pe_top_plan.hpp"

namespace pe
{
    // Test knob of the emulation ONLY (tests/test_host_logic.py: residual safety net): every pivot reciprocal is taken with a relative
    // error PE_EMU_PIVOT_ERROR, i.e. a deliberately inexact LU whose solves leave residuals that iterative refinement must repair.
    static double emu_rcp(double d)
    {
        static double const eps = [] {
            char const* v = std::getenv("PE_EMU_PIVOT_ERROR");
            return v && *v ? std::atof(v) : 0.0;
        }();
        return (1.0 / d) * (1.0 + eps);
    }

}  // namespace pe
#include "quad_emu.hpp"  // (uses emu_rcp)
namespace pe
{
    // the lane-group kernel of the wave fronts, every (quad, list) wavefront one after the other
    static void emu_factor_quads(DevView const& V)
    {
        if(!V.quad) return;
        QuadEmu::lds().assign(static_cast<size_t>(V.q_lds_stride) * 4 + 1, std::nan(""));  // (every wavefront starts with garbage in LDS)
        for(int quad = 0; quad < V.n_quads; ++quad)
            for(int list = 0; list < V.n_parts * V.n_waves; ++list) quad_factor_list<QuadEmu>(V, quad, list);
        if(V.n_mid > 0)
            for(int quad = 0; quad < V.n_quads; ++quad)
                for(int list = 0; list < V.n_parts * V.n_waves; ++list) quad_factor_mid_list<QuadEmu>(V, quad, list);
    }

    static void emu_backward_quads(DevView const& V)
    {
        if(!V.quad_back) return;
        for(int quad = 0; quad < V.n_quads; ++quad)
            for(int list = 0; list < V.n_parts * V.n_waves; ++list) quad_backward_list<QuadEmu>(V, quad, list);
    }

    struct SerialTeam
    {
        int nw;
        SerialTeam wave_team(int) const { return SerialTeam{1}; }
        int tid() const { return 0; }
        int uniform(int v) const { return v; }
        double rcp(double d) const { return emu_rcp(d); }
        void sync_lds() const {}
        int size() const { return 1; }
        void sync() const {}
        int sync_or(int v) const { return v; }
        int lanes() const { return 1; }
        long long clock() const { return 0; }
        void wave_fence() const {}
        template <class F>
        void for_each_wave(F&& body) const
        {
            for(int w = 0; w < nw; ++w) body(w, 0, 1);
        }
        int n_waves() const { return nw; }
        bool single_wave() const { return nw == 1; }
        void team_max4(double (&)[4], double*) const {}  // one thread: its partial maxima are the team's
        struct Blk8
        {
            double const* p;
            int ld;
        };
        Blk8 blk_load(double const* blk, int ld, int, int) const { return Blk8{blk, ld}; }
        double blk_at(Blk8 const& b, int r, int c) const { return b.p[r + c * b.ld]; }
        int block_step(double* Lp, int ld, int m, double* Up, int ldu, double* g, int p, int u, int k0, int kb, bool fuse, int) const
        {
            for(int kk = 0; kk < kb; ++kk)
            {
                double const piv = Lp[(k0 + kk) + (k0 + kk) * ld];
                if(piv == 0.0 || !(std::fabs(piv) <= 1.7976931348623157e308)) return 1;
                double const r = emu_rcp(piv);
                for(int i = k0 + kk + 1; i < m; ++i)
                {
                    double const l = Lp[i + (k0 + kk) * ld] * r;
                    for(int c = kk + 1; c < kb; ++c) Lp[i + (k0 + c) * ld] -= l * Lp[(k0 + kk) + (k0 + c) * ld];
                    Lp[i + (k0 + kk) * ld] = l;
                }
            }
            int const ncolL = p - k0 - kb, ncols = ncolL + u + (fuse ? 1 : 0);
            for(int j = 0; j < ncols; ++j)
            {
                double* col = j < ncolL ? Lp + (k0 + kb + j) * ld + k0 : (j < ncolL + u ? Up + (j - ncolL) * ldu + k0 : g + k0);
                for(int kk = 1; kk < kb; ++kk)
                    for(int r = 0; r < kk; ++r) col[kk] -= Lp[(k0 + kk) + (k0 + r) * ld] * col[r];
            }
            return 0;
        }
        // rows below / columns right of a factored kb x kb diagonal block (serial stand-in of the per-thread solves)
        void panel_solve(double* Lp, int ld, int m, double* Up, int ldu, double* g, int p, int u, int k0, int kb, bool fuse, int, int) const
        {
            double const* blk = Lp + k0 + k0 * ld;
            for(int i = k0 + kb; i < m; ++i)
            {
                double* row = Lp + i + k0 * ld;
                for(int kk = 0; kk < kb; ++kk)
                {
                    double acc = row[kk * ld];
                    for(int r = 0; r < kk; ++r) acc -= row[r * ld] * blk[r + kk * ld];
                    row[kk * ld] = acc * emu_rcp(blk[kk + kk * ld]);
                }
            }
            int const ncolL = p - k0 - kb, ncols = ncolL + u + (fuse ? 1 : 0);
            for(int j = 0; j < ncols; ++j)
            {
                double* col = j < ncolL ? Lp + (k0 + kb + j) * ld + k0 : (j < ncolL + u ? Up + (j - ncolL) * ldu + k0 : g + k0);
                for(int kk = 1; kk < kb; ++kk)
                    for(int r = 0; r < kk; ++r) col[kk] -= blk[kk + r * ld] * col[r];
            }
        }
        int diag_lu8(double* blk, int ld, int kb, int) const
        {
            for(int kk = 0; kk < kb; ++kk)
            {
                double const piv = blk[kk + kk * ld];
                if(piv == 0.0 || !(std::fabs(piv) <= 1.7976931348623157e308)) return 1;
                for(int r = kk + 1; r < kb; ++r)
                {
                    double const l = blk[r + kk * ld] * emu_rcp(piv);
                    for(int c = kk + 1; c < kb; ++c) blk[r + c * ld] -= l * blk[kk + c * ld];
                    blk[r + kk * ld] = l;
                }
            }
            return 0;
        }
        void tri_lower_unit(double* t, double const* Lb, int ld, int p, int nrows, int) const
        {
            for(int k = 0; k < p; ++k)
                for(int i = k + 1; i < nrows; ++i) t[i] -= Lb[i + k * ld] * t[k];
        }
        void tri_upper(double* t, double const* Ub, int ld, int p, int nu, int) const
        {
            double const* U12 = Ub + p * ld;
            for(int j = 0; j < nu; ++j)
                for(int i = 0; i < p; ++i) t[i] -= U12[i + j * p] * t[p + j];
            for(int k = p - 1; k >= 0; --k)
            {
                t[k] = t[k] / Ub[k + k * ld];
                for(int i = 0; i < k; ++i) t[i] -= Ub[i + k * ld] * t[k];
            }
        }
        unsigned long long tile_children(unsigned const* cmk, int nch, int ti, int tj, int) const
        {
            unsigned long long todo = 0;
            int const bi = ti < 31 ? ti : 31, bj = tj < 31 ? tj : 31;
            for(int q = 0; q < nch; ++q)
                if(((cmk[q] >> bi) & (cmk[q] >> bj)) & 1u) todo |= 1ull << q;
            return todo;
        }
        // plain-loop stand-in for the 16 x 16 matrix-core tiles
        struct Acc
        {
            double v[16][16];
        };
        Acc tile_zero() const
        {
            Acc a;
            for(auto& r: a.v)
                for(double& x: r) x = 0.0;
            return a;
        }
        void tile_add(Acc& a, Acc const& b) const
        {
            for(int r = 0; r < 16; ++r)
                for(int c = 0; c < 16; ++c) a.v[r][c] += b.v[r][c];
        }
        Acc tile_load(double const* C, int ldc, int mr, int nc, int) const
        {
            Acc a = tile_zero();
            for(int c = 0; c < nc; ++c)
                for(int r = 0; r < mr; ++r) a.v[r][c] = C[r + c * ldc];
            return a;
        }
        void tile_store(Acc const& a, double* C, int ldc, int mr, int nc, int) const
        {
            for(int c = 0; c < nc; ++c)
                for(int r = 0; r < mr; ++r) C[r + c * ldc] = a.v[r][c];
        }
        void tile_mulsub(Acc& a, double const* A, int lda, double const* B, int ldb, int mr, int nc, int kd, int) const
        {
            for(int k = 0; k < kd; ++k)
                for(int c = 0; c < nc; ++c)
                    for(int r = 0; r < mr; ++r) a.v[r][c] -= A[r + k * lda] * B[k + c * ldb];
        }
        template <class F>
        void tile_foreach(Acc& a, int, F&& f) const
        {
            for(int c = 0; c < 16; ++c)
                for(int r = 0; r < 16; ++r) f(r, c, a.v[r][c]);
        }
    };


    // The top levels as the device runs them (pe_top_plan.hpp: the same plan as m2_sequence): every level of a launch gets that
    // launch's LDS -- front_factor asserts that the front fits -- and a 16-wavefront launch carries its ChainState through its run.
    template <class Team>
    bool emu_factor_top(Team const& tm, DevView const& V, int b, double* mem)
    {
        bool ok = true;
        for_each_top_launch(V, V.batch, V.high_occupancy && V.n_waves == 4, V.mid_top_limit,
                            [&](TopLaunch const& t)
                            {
                                ChainState cs;  // (a run of single-front wide levels is ONE workgroup on the device: k_m2_factor_top_wide)
                                for(int l = t.level; l < t.level + t.nlev; ++l)
                                    for(int i = V.top_ptr[l]; i < V.top_ptr[l + 1]; ++i)
                                    {
                                        if(V.f_need[V.top_list[i]] > static_cast<int>(t.lds_doubles) - 2)  // the device's guard (pe_kernels.hip lds_overrun)
                                        {
                                            V.flags[b] |= 8;
                                            ok = false;
                                            continue;
                                        }
                                        if(!front_factor<Team, true>(tm, V, b, V.top_list[i], mem, static_cast<int>(t.lds_doubles) - 2, 0, true, t.kind == 1 ? &cs : nullptr)) ok = false;
                                    }
                            });
        return ok;
    }
    hipError_t launch_tr_steps(hipStream_t, DevView const& V, double dt, int nsteps, bool reuse)
    {
        std::vector<double> mem(static_cast<size_t>(std::max(V.lds_doubles, V.lds_top_doubles)) + 1);
        for(int b = 0; b < V.batch; ++b) tr_steps(SerialTeam{V.n_waves}, V, b, dt, nsteps, reuse, mem.data());
        return hipSuccess;
    }
    hipError_t launch_dc_point(hipStream_t, DevView const& V, int mode)
    {
        std::vector<double> mem(static_cast<size_t>(std::max(V.lds_doubles, V.lds_top_doubles)) + 1);
        for(int b = 0; b < V.batch; ++b) dc_point(SerialTeam{V.n_waves}, V, b, mode, mem.data());
        return hipSuccess;
    }
    hipError_t launch_m2_companion(hipStream_t, DevView const& V, double dt)
    {
        for(int b = 0; b < V.batch; ++b)
            if(V.active[b]) companion_update(SerialTeam{1}, V, b, dt);
        return hipSuccess;
    }
    hipError_t launch_m2_iteration(hipStream_t, DevView const& V, int mode, double t, double last_step, bool do_factor, hipEvent_t, hipEvent_t, int stamp_mode,
                                   bool companion, double companion_dt)
    {
        bool const have_lists = V.dyn_a && V.dyn_b;
        bool const stamp_dynamic = stamp_mode == 1;
        int const stamp_dyn = have_lists ? stamp_mode : 0;
        std::vector<double> mem(static_cast<size_t>(std::max(V.lds_doubles, V.lds_top_doubles)) + 1);
        SerialTeam tm{V.n_waves};
        for(int b = 0; b < V.batch; ++b)
        {
            if(!V.active[b]) continue;
            double* x = V.x + static_cast<long long>(b) * V.rows;
            double* xp = V.xprev + static_cast<long long>(b) * V.rows;
            double* w = V.w + static_cast<long long>(b) * V.rows;
            double const* rhs = V.rhs + static_cast<long long>(b) * V.rows;
            if(companion) companion_update(SerialTeam{1}, V, b, companion_dt);  // (k_m2_eval: the step's companion update rides along)
            for(int r = 0; r < V.rows; ++r) xp[r] = x[r];
            eval_devices(tm, V, b, mode, t, last_step, stamp_dynamic && V.dyn_a && V.dyn_b);
            V.flags[b] = 0;
            if(V.eta_acc)
                for(int k = 0; k < 4; ++k) V.eta_acc[4 * b + k] = 0.0;
            // (k_m2_stamp as the device runs it: the gathered rows write their entry of w, the others are copied -- in three chunks, like a
            //  grid of three workgroups, so that the chunk arithmetic is exercised)
            for(int k = 0; k < V.rows; ++k) w[k] = std::nan("");
            for(int g = 0; g < 3; ++g)
            {
                if(stamp_dyn) stamp_dynamic_chunk(V, b, g, 3, 0, 1, true, stamp_dyn == 2);
                else
                    stamp_chunk(V, b, g, 3, 0, 1, true);
            }
            (void)rhs;
        }
        if(do_factor) emu_factor_quads(V);  // (a launch of its own on the device, between the stamp and the per-instance parts)
        for(int b = 0; b < V.batch; ++b)
        {
            if(!V.active[b]) continue;
            double* x = V.x + static_cast<long long>(b) * V.rows;
            double* xp = V.xprev + static_cast<long long>(b) * V.rows;
            double* w = V.w + static_cast<long long>(b) * V.rows;
            if(do_factor)
            {
                for(int q = 0; q < V.n_parts; ++q)
                    if(!factor_part(tm, V, b, q, mem.data(), true)) V.flags[b] |= 4;
                if(!emu_factor_top(tm, V, b, mem.data())) V.flags[b] |= 4;
            }
            else
            {
                for(int q = 0; q < V.n_parts; ++q) forward_part(tm, V, b, q, mem.data());
                for(int l = 0; l < V.n_top_levels; ++l)
                    for(int i = V.top_ptr[l]; i < V.top_ptr[l + 1]; ++i) front_forward(tm, V, b, V.top_list[i], mem.data(), V.max_m, V.lds_top_stage);
            }
            for(int l = V.n_top_levels - 1; l >= 0; --l)
                for(int i = V.top_ptr[l]; i < V.top_ptr[l + 1]; ++i) front_backward(tm, V, b, V.top_list[i], mem.data(), V.max_m, V.lds_top_stage);
            for(int q = 0; q < V.n_parts; ++q) backward_part(tm, V, b, q, mem.data());
        }
        emu_backward_quads(V);  // (its own launch on the device, behind the parts' backward pass)
        for(int b = 0; b < V.batch; ++b)
        {
            if(!V.active[b]) continue;
            double* x = V.x + static_cast<long long>(b) * V.rows;
            double* xp = V.xprev + static_cast<long long>(b) * V.rows;
            double* w = V.w + static_cast<long long>(b) * V.rows;
            for(int k = 0; k < V.rows; ++k)
            {
                int const r = V.col_src[k];
                double const xn = w[k];
                x[r] = xn;
                if(!(std::fabs(xn) <= 1.7976931348623157e308)) V.flags[b] |= 1;
                bool const node = r < V.n_nodes;
                double const tol = (node ? V.v_abstol : V.i_abstol) + (node ? V.v_reltol : V.i_reltol) * std::fmax(std::fabs(xn), std::fabs(xp[r]));
                if(!(std::fabs(xn - xp[r]) <= tol)) V.flags[b] |= 2;
            }
            if(V.residual_tol > 0.0 && !(V.nonlinear && (V.flags[b] & 2)))  // (only an iterate about to be accepted, as k_m2_residual)
            {
                double n4[4];
                residual_norms(tm, V, b, nullptr, n4);
                for(int k = 0; k < 4; ++k) V.eta_acc[4 * b + k] = n4[k];
            }
        }
        return hipSuccess;
    }
    hipError_t launch_m2_publish(hipStream_t, DevView const& V, int* pub_flags, double* pub_eta, unsigned long long* pub_seq, unsigned long long seq)
    {
        for(int b = 0; b < V.batch; ++b) pub_flags[b] = V.flags[b];
        if(V.residual_tol > 0.0)
            for(int i = 0; i < 4 * V.batch; ++i) pub_eta[i] = V.eta_acc[i];
        *pub_seq = seq;
        return hipSuccess;
    }

    // (captured launch sequences are a device matter: the emulation runs the plain sequence and the publication)
    struct M2GraphCache
    {
        int launches{};
    };
    M2GraphCache* m2_graphs_create() { return new M2GraphCache; }
    void m2_graphs_destroy(M2GraphCache* c) { delete c; }
    void m2_graphs_clear(M2GraphCache*) {}
    hipError_t launch_m2_iteration_graph(hipStream_t st, M2GraphCache* cache, DevView const& V, int mode, double t, double last_step, bool do_factor, int stamp_mode,
                                         bool companion, double companion_dt, int* pub_flags, double* pub_eta, unsigned long long* pub_seq, unsigned long long seq)
    {
        ++cache->launches;
        hipError_t const rc = launch_m2_iteration(st, V, mode, t, last_step, do_factor, nullptr, nullptr, stamp_mode, companion, companion_dt);
        return rc != hipSuccess ? rc : launch_m2_publish(st, V, pub_flags, pub_eta, pub_seq, seq);
    }
    hipError_t launch_m2_refine(hipStream_t, DevView const& V)
    {
        std::vector<double> mem(static_cast<size_t>(std::max(V.lds_doubles, V.lds_top_doubles)) + 1);
        SerialTeam tm{V.n_waves};
        for(int b = 0; b < V.batch; ++b)
        {
            if(!V.active[b]) continue;
            double* x = V.x + static_cast<long long>(b) * V.rows;
            double const* xp = V.xprev + static_cast<long long>(b) * V.rows;
            double* xs = V.xsave + static_cast<long long>(b) * V.rows;
            double* rr = V.rres + static_cast<long long>(b) * V.rows;
            double* w = V.w + static_cast<long long>(b) * V.rows;
            double n4[4];
            residual_norms(tm, V, b, rr, n4);
            for(int r = 0; r < V.rows; ++r) xs[r] = x[r];
            for(int k = 0; k < V.rows; ++k) w[k] = rr[V.row_src[k]];
        }
        emu_factor_quads(V);
        for(int b = 0; b < V.batch; ++b)
        {
            if(!V.active[b]) continue;
            double* x = V.x + static_cast<long long>(b) * V.rows;
            double const* xp = V.xprev + static_cast<long long>(b) * V.rows;
            double* xs = V.xsave + static_cast<long long>(b) * V.rows;
            double* w = V.w + static_cast<long long>(b) * V.rows;
            double n4[4];
            for(int q = 0; q < V.n_parts; ++q)
                if(!factor_part(tm, V, b, q, mem.data(), true)) V.flags[b] |= 4;
            if(!emu_factor_top(tm, V, b, mem.data())) V.flags[b] |= 4;
            for(int l = V.n_top_levels - 1; l >= 0; --l)
                for(int i = V.top_ptr[l]; i < V.top_ptr[l + 1]; ++i) front_backward(tm, V, b, V.top_list[i], mem.data(), V.max_m, V.lds_top_stage);
            for(int q = 0; q < V.n_parts; ++q) backward_part(tm, V, b, q, mem.data());
        }
        emu_backward_quads(V);
        for(int b = 0; b < V.batch; ++b)
        {
            if(!V.active[b]) continue;
            double* x = V.x + static_cast<long long>(b) * V.rows;
            double const* xp = V.xprev + static_cast<long long>(b) * V.rows;
            double* xs = V.xsave + static_cast<long long>(b) * V.rows;
            double* w = V.w + static_cast<long long>(b) * V.rows;
            double n4[4];
            for(int k = 0; k < V.rows; ++k) x[V.col_src[k]] = xs[V.col_src[k]] + w[k];
            V.flags[b] = 0;
            residual_norms(tm, V, b, nullptr, n4);
            for(int k = 0; k < 4; ++k) V.eta_acc[4 * b + k] = n4[k];
            for(int r = 0; r < V.rows; ++r)
            {
                if(!(std::fabs(x[r]) <= 1.7976931348623157e308)) V.flags[b] |= 1;
                bool const node = r < V.n_nodes;
                double const tol = (node ? V.v_abstol : V.i_abstol) + (node ? V.v_reltol : V.i_reltol) * std::fmax(std::fabs(x[r]), std::fabs(xp[r]));
                if(!(std::fabs(x[r] - xp[r]) <= tol)) V.flags[b] |= 2;
            }
        }
        return hipSuccess;
    }
    hipError_t launch_ac_residual(hipStream_t, DevView const& V, double const* xacc, double const* b0, int rhs0, double* worst)
    {
        double w = 0.0;
        for(int b = 0; b < V.batch; ++b)
        {
            double const wb = ac_residual(SerialTeam{1}, V, b, xacc, b0, rhs0);
            w = (wb > w || wb != wb) ? wb : w;
        }
        *worst = w;
        return hipSuccess;
    }
    hipError_t launch_ac_accumulate(hipStream_t, DevView const& V, double* xacc, double* b0, bool first)
    {
        size_t const n = static_cast<size_t>(V.batch) * V.rows;
        for(size_t i = 0; i < n; ++i)
        {
            if(first)
            {
                xacc[i] = V.x[i];
                b0[i] = V.rhs[i];
            }
            else
                xacc[i] += V.x[i];
        }
        return hipSuccess;
    }
    hipError_t launch_csr_residual(hipStream_t, DevView const& V, double const* xacc, double const* b0, double* worst)
    {
        double w = 0.0;
        for(int r = 0; r < V.rows; ++r)
        {
            double acc = b0[r], mag = std::fabs(acc);
            for(int e = V.csr_rp[r]; e < V.csr_rp[r + 1]; ++e)
            {
                double const t = V.aval[e] * xacc[V.csr_ci[e]];
                acc -= t;
                mag += std::fabs(t);
            }
            V.rhs[r] = acc;
            double const wb = std::fabs(acc) / (mag > 0.0 ? mag : 1.0);
            w = (wb > w || wb != wb) ? wb : w;
        }
        *worst = w;
        return hipSuccess;
    }
    hipError_t launch_stream_copy(hipStream_t, void const* src, void* dst, size_t bytes)
    {
        std::memcpy(dst, src, bytes);
        return hipSuccess;
    }
    hipError_t launch_sweep_statistics(hipStream_t, DevView const& V, int n_chunks, double* partial, double* out)
    {
        (void)partial;
        // the chunked summation order of the device kernels (so that the emulation checks the same arithmetic)
        int const chunk_len = (V.batch + n_chunks - 1) / n_chunks;
        for(int r = 0; r < V.rows; ++r)
        {
            double s = 0.0, q = 0.0, mn = INFINITY, mx = -INFINITY;
            for(int c = 0; c < n_chunks; ++c)
            {
                double cs = 0.0, cq = 0.0;
                for(int b = c * chunk_len; b < V.batch && b < (c + 1) * chunk_len; ++b)
                {
                    double const v = V.x[static_cast<long long>(b) * V.rows + r];
                    cs += v;
                    cq += v * v;
                    mn = std::fmin(mn, v);
                    mx = std::fmax(mx, v);
                }
                s += cs;
                q += cq;
            }
            out[r] = s;
            out[V.rows + r] = q;
            out[2ll * V.rows + r] = mn;
            out[3ll * V.rows + r] = mx;
        }
        return hipSuccess;
    }
    hipError_t launch_factor_solve(hipStream_t, DevView const& V, bool do_factor)
    {
        std::vector<double> mem(static_cast<size_t>(std::max(V.lds_doubles, V.lds_top_doubles)) + 1);
        SerialTeam tm{V.n_waves};
        for(int b = 0; b < V.batch; ++b)
        {
            int st = ST_OK;
            if(do_factor)
            {
                permute_rhs(tm, V, b);
                if(!factor_all(tm, V, b, mem.data(), true)) st = ST_SINGULAR;
            }
            if(st == ST_OK)
            {
                solve_all(tm, V, b, mem.data(), do_factor);
                double const* x = V.x + static_cast<long long>(b) * V.rows;
                for(int r = 0; r < V.rows; ++r)
                    if(!(std::fabs(x[r]) <= 1.7976931348623157e308)) st = ST_SINGULAR;
            }
            V.status[b] = st;
        }
        return hipSuccess;
    }
}  // namespace pe
