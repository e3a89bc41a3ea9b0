// pe_front.hpp -- the per-instance numeric path, written against a "team" (the threads that cooperate on one
// circuit instance).  On the GPU the team is one workgroup (pe_kernels.hip); tests/emu instantiates the same
// code with a one-thread team on the host purely to check indexing before a kernel is ever launched
// (test infrastructure: the product library never instantiates the host team).
//
// Reference semantics restated here (paths relative to the reference tree):
//   companion_update  circult::update_tr_step -> step_changed_tr_define of capacitor.h:106-128,
//                     inductor.h:134-160, PN_junction.h:440-476
//   eval_devices      iterate_{dc,tr,trop}_define of capacitor.h:132-155, inductor.h:164-195, VAC.h:162-179,
//                     PN_junction.h:358-402,478-503 (vlimit :58-109, limexp :10-16)
//   stamp             MNA accumulation of circult::solve_once (circuit.h:1015-1110, mna.h:60-157)
//   factor/solve      replaces Eigen SparseLU compute()+solve() (circuit.h:1516-1518)
//   newton_converged  circult::solve convergence test (circuit.h:921-948)
#pragma once
#include "pe_device.hpp"

#include <cmath>

#if defined(__HIPCC__)
    #define PE_DEV __device__ __forceinline__
#else
    #define PE_DEV inline
#endif

namespace pe
{
    enum : int
    {
        MODE_OP = 0,
        MODE_DC = 1,
        MODE_TR = 4,
        MODE_TROP = 5
    };
    enum : int
    {
        ST_OK = 0,
        ST_SINGULAR = -3,
        ST_NO_CONVERGENCE = -4
    };

    PE_DEV double limexp(double x)
    {
        if(x > 50.0) return exp(50.0) * (1.0 + (x - 50.0));
        if(x < -50.0) return exp(-50.0);
        return exp(x);
    }

    // SPICE3f5 junction limiting with the breakdown mirror (PN_junction.h:58-109)
    PE_DEV double vlimit(double Ud, double Ud_last, double Ute, double Uth, double Bv_eff, bool Bv_set)
    {
        bool flag = false;
        double Ud_0, Ud_1, Ud_f;
        if(Bv_set && Ud < fmin(0.0, -Bv_eff + 10.0 * Ute))
        {
            Ud_0 = -(Ud + Bv_eff);
            Ud_1 = -(Ud_last + Bv_eff);
            flag = true;
        }
        else
        {
            Ud_0 = Ud;
            Ud_1 = Ud_last;
        }
        if(Ud_0 > Uth && fabs(Ud_0 - Ud_1) > 2.0 * Ute)
        {
            if(Ud_1 > 0)
            {
                double const arg = (Ud_0 - Ud_1) / Ute;
                if(arg > 0.0) Ud_f = Ud_1 + Ute * (2.0 + log(arg - 2.0));
                else
                    Ud_f = Ud_1 - Ute * (2.0 + log(2.0 - arg));
            }
            else
                Ud_f = Ute * log(Ud_0 / Ute);
        }
        else
        {
            Ud_f = Ud_0;
            if(Ud_0 < 0.0)
            {
                double const arg = Ud_1 > 0.0 ? -1.0 - Ud_1 : 2.0 * Ud_1 - 1;
                if(Ud_0 < arg) Ud_f = arg;
            }
        }
        return flag ? -(Ud_f + Bv_eff) : Ud_f;
    }

    PE_DEV double volt(double const* x, int row) { return row >= 0 ? x[row] : 0.0; }

    // ------------------------------------------------------------------------------------------------
    // start of a time step: trapezoidal companion models take the previous time point's solution
    // ------------------------------------------------------------------------------------------------
    template <class Team>
    PE_DEV void companion_update(Team const& tm, DevView const& V, int b, double dt)
    {
        double const* x = V.x + static_cast<long long>(b) * V.rows;
        double* dv = V.dv + static_cast<long long>(b) * V.dv_len;
        if(dt > 0.0)
        {
            double* hist = V.c_hist + static_cast<long long>(b) * V.nC;
            double* prevg = V.c_prevg + static_cast<long long>(b) * V.nC;
            double const* cap = V.c_cap + static_cast<long long>(b) * V.nC;
            for(int i = tm.tid(); i < V.nC; i += tm.size())
            {
                double const v_prev = volt(x, V.c_a[i]) - volt(x, V.c_b[i]);
                double const g_new = 2.0 * cap[i] / dt;
                hist[i] = -(g_new + prevg[i]) * v_prev - hist[i];
                prevg[i] = g_new;
            }
        }
        {
            double const* ind = V.l_ind + static_cast<long long>(b) * V.nL;
            for(int i = tm.tid(); i < V.nL; i += tm.size())
            {
                double req = 0.0, ueq = 0.0;
                if(dt > 0.0)
                {
                    double const v_prev = volt(x, V.l_a[i]) - volt(x, V.l_b[i]);
                    double const i_prev = x[V.l_k[i]];
                    req = 2.0 * ind[i] / dt;
                    ueq = -v_prev - req * i_prev;
                }
                dv[V.dv_lr + i] = -req;
                dv[V.dv_lu + i] = ueq;
            }
        }
        {
            double* udl = V.d_udlast + static_cast<long long>(b) * V.nD;
            double* geq = V.d_geq + static_cast<long long>(b) * V.nD;
            double* hist = V.d_hist + static_cast<long long>(b) * V.nD;
            double* prevg = V.d_prevg + static_cast<long long>(b) * V.nD;
            double const* par = V.d_par + static_cast<long long>(b) * V.nD * DP_NCOL;
            for(int i = tm.tid(); i < V.nD; i += tm.size())
            {
                double const vd = volt(x, V.d_a[i]) - volt(x, V.d_c[i]);
                udl[i] = vd;
                double const tt = par[i * DP_NCOL + DP_TT];
                double const cd = tt * geq[i];
                if(!(dt > 0.0) || !(tt > 0.0) || !(geq[i] > 0.0) || !(cd > 0.0))
                {
                    hist[i] = 0.0;
                    prevg[i] = 0.0;
                }
                else
                {
                    double const g_new = 2.0 * cd / dt;
                    hist[i] = -(g_new + prevg[i]) * vd - hist[i];
                    prevg[i] = g_new;
                }
            }
        }
    }

    // ------------------------------------------------------------------------------------------------
    // device evaluation: fills the dynamic part of dv for this Newton iteration
    // ------------------------------------------------------------------------------------------------
    template <class Team>
    PE_DEV void eval_devices(Team const& tm, DevView const& V, int b, int mode, double t, double last_step)
    {
        double const* x = V.x + static_cast<long long>(b) * V.rows;
        double* dv = V.dv + static_cast<long long>(b) * V.dv_len;
        bool const tr = mode == MODE_TR;
        {
            double const* hist = V.c_hist + static_cast<long long>(b) * V.nC;
            double const* prevg = V.c_prevg + static_cast<long long>(b) * V.nC;
            for(int i = tm.tid(); i < V.nC; i += tm.size())
            {
                dv[V.dv_cg + i] = tr ? prevg[i] : 0.0;
                dv[V.dv_ci + i] = tr ? hist[i] : 0.0;
            }
        }
        if(!tr || !(last_step > 0.0))
        {
            for(int i = tm.tid(); i < V.nL; i += tm.size())
            {
                dv[V.dv_lr + i] = 0.0;
                dv[V.dv_lu + i] = 0.0;
            }
        }
        {
            double const* par = V.vac_par + static_cast<long long>(b) * V.nVac * 3;
            for(int i = tm.tid(); i < V.nVac; i += tm.size())
            {
                double e = 0.0;
                if(tr) e = par[3 * i] * sin(par[3 * i + 1] * t + par[3 * i + 2]);
                else if(mode == MODE_TROP)
                    e = par[3 * i] * sin(par[3 * i + 1] * 0.0 + par[3 * i + 2]);
                dv[V.dv_vac + i] = e;
            }
        }
        {
            double* udl = V.d_udlast + static_cast<long long>(b) * V.nD;
            double* geqs = V.d_geq + static_cast<long long>(b) * V.nD;
            double const* hist = V.d_hist + static_cast<long long>(b) * V.nD;
            double const* prevg = V.d_prevg + static_cast<long long>(b) * V.nD;
            double const* par = V.d_par + static_cast<long long>(b) * V.nD * DP_NCOL;
            for(int i = tm.tid(); i < V.nD; i += tm.size())
            {
                double const* p = par + i * DP_NCOL;
                double const Ute = p[DP_UTE], Uter = p[DP_UTER], Bv_eff = p[DP_BV_EFF];
                bool const Bv_set = p[DP_BV_SET] != 0.0;
                double Ud = volt(x, V.d_a[i]) - volt(x, V.d_c[i]);
                Ud = vlimit(Ud, udl[i], Ute, p[DP_UTH], Bv_eff, Bv_set);
                udl[i] = Ud;
                double Id, geq;
                if(Bv_set && Ud < -Bv_eff)
                {
                    double const e = limexp(-(Bv_eff + Ud) / Ute);
                    Id = -p[DP_IS_EFF] * e;
                    geq = p[DP_IS_EFF] * e / Ute;
                }
                else
                {
                    double e = limexp(Ud / Ute);
                    geq = p[DP_IS_EFF] * e / Ute;
                    Id = p[DP_IS_EFF] * (e - 1.0);
                    e = limexp(Ud / Uter);
                    geq += p[DP_ISR_EFF] * e / Uter;
                    Id += p[DP_ISR_EFF] * (e - 1.0);
                }
                geqs[i] = geq;
                double const Ieq = Id - Ud * geq;
                double g = geq, ie = Ieq;
                if(tr && p[DP_TT_STAMP] != 0.0 && prevg[i] != 0.0)
                {
                    // PN_junction.h:478-503: the diode conductance and the diffusion-cap companion are two
                    // consecutive += on the same four cells; summed here in the same order
                    g = geq + prevg[i];
                    ie = Ieq + hist[i];
                }
                dv[V.dv_dg + i] = g;
                dv[V.dv_di + i] = ie;
            }
        }
    }

    // ------------------------------------------------------------------------------------------------
    // MNA assembly: every A slot / RHS row gathers its contributions in model order (deterministic)
    // ------------------------------------------------------------------------------------------------
    template <class Team>
    PE_DEV void stamp(Team const& tm, DevView const& V, int b)
    {
        double const* dv = V.dv + static_cast<long long>(b) * V.dv_len;
        double* a = V.aval + static_cast<long long>(b) * V.nnzA;
        double* rhs = V.rhs + static_cast<long long>(b) * V.rows;
        for(int s = tm.tid(); s < V.nnzA; s += tm.size())
        {
            double acc = 0.0;
            for(int e = V.a_ptr[s]; e < V.a_ptr[s + 1]; ++e)
            {
                int const src = V.a_src[e];
                double const v = dv[src >> 1];
                acc = (src & 1) ? acc - v : acc + v;
            }
            a[s] = acc;
        }
        for(int r = tm.tid(); r < V.rows; r += tm.size())
        {
            double acc = 0.0;
            for(int e = V.b_ptr[r]; e < V.b_ptr[r + 1]; ++e)
            {
                int const src = V.b_src[e];
                double const v = dv[src >> 1];
                acc = (src & 1) ? acc - v : acc + v;
            }
            rhs[r] = acc;
        }
    }

    // ------------------------------------------------------------------------------------------------
    // multifrontal LU: one dense front.  F is m x m column major (LDS when it fits, else global scratch).
    // Returns false (uniformly over the team) on a zero / non-finite pivot.
    // ------------------------------------------------------------------------------------------------
    template <class Team>
    PE_DEV bool factor_front(Team const& tm, DevView const& V, int b, int s, double* F)
    {
        int const p = V.f_p[s], u = V.f_u[s], m = p + u;
        double const* a = V.aval + static_cast<long long>(b) * V.nnzA;
        double* arena = V.arena + static_cast<long long>(b) * V.arena_doubles;
        double* fac = V.factor + static_cast<long long>(b) * V.factor_doubles;
        int const T = tm.size(), t0 = tm.tid();

        for(int i = t0; i < m * m; i += T) F[i] = 0.0;
        tm.sync();
        for(int e = V.f_asm_ptr[s] + t0; e < V.f_asm_ptr[s + 1]; e += T)
        {
            int const pos = V.asm_pos[e];
            F[(pos >> 16) + (pos & 0xffff) * m] = a[V.asm_slot[e]];
        }
        tm.sync();
        for(int ch = V.f_child_ptr[s]; ch < V.f_child_ptr[s + 1]; ++ch)
        {
            int const c = V.f_child[ch];
            int const uc = V.f_u[c];
            double const* Sc = arena + V.f_sptr[c];
            int const* rel = V.f_rel + V.f_rows_ptr[c];
            for(int idx = t0; idx < uc * uc; idx += T)
            {
                int const j = idx / uc, i = idx - j * uc;
                F[rel[i] + rel[j] * m] += Sc[idx];
            }
            tm.sync();
        }
        bool ok = true;
        for(int k = 0; k < p; ++k)
        {
            double const piv = F[k + k * m];
            if(piv == 0.0 || !(fabs(piv) <= 1.7976931348623157e308))
            {
                ok = false;
                break;  // uniform: every thread reads the same pivot
            }
            double const inv = 1.0 / piv;
            int const nr = m - k - 1;
            for(int idx = t0; idx < nr * nr; idx += T)
            {
                int const jj = idx / nr, ii = idx - jj * nr;
                int const i = k + 1 + ii, j = k + 1 + jj;
                F[i + j * m] -= (F[i + k * m] * inv) * F[k + j * m];
            }
            tm.sync();
        }
        if(!ok) return false;
        // panels out: L (m x p, unit lower part scaled here; upper part + diagonal = U11), U12 (p x u), S (u x u)
        double* Lp = fac + V.f_lptr[s];
        for(int idx = t0; idx < m * p; idx += T)
        {
            int const k = idx / m, i = idx - k * m;
            double v = F[i + k * m];
            if(i > k) v *= 1.0 / F[k + k * m];
            Lp[idx] = v;
        }
        double* Up = fac + V.f_uptr[s];
        for(int idx = t0; idx < p * u; idx += T)
        {
            int const j = idx / p, k = idx - j * p;
            Up[idx] = F[k + (p + j) * m];
        }
        double* Ss = arena + V.f_sptr[s];
        for(int idx = t0; idx < u * u; idx += T)
        {
            int const j = idx / u, i = idx - j * u;
            Ss[idx] = F[(p + i) + (p + j) * m];
        }
        tm.sync();
        return true;
    }

    template <class Team>
    PE_DEV bool factor_all(Team const& tm, DevView const& V, int b, double* lds_front)
    {
        double* big = V.bigfront ? V.bigfront + static_cast<long long>(b) * V.bigfront_doubles : nullptr;
        for(int s = 0; s < V.nfronts; ++s)
        {
            int const m = V.f_p[s] + V.f_u[s];
            double* F = (m <= V.lds_front_cap) ? lds_front : big;
            if(!factor_front(tm, V, b, s, F)) return false;
        }
        return true;
    }

    // ------------------------------------------------------------------------------------------------
    // triangular solves.  yl: team-shared scratch of at least max_m doubles.
    // ------------------------------------------------------------------------------------------------
    template <class Team>
    PE_DEV void solve_all(Team const& tm, DevView const& V, int b, double* yl)
    {
        int const T = tm.size(), t0 = tm.tid();
        double const* rhs = V.rhs + static_cast<long long>(b) * V.rows;
        double* w = V.w + static_cast<long long>(b) * V.rows;
        double* x = V.x + static_cast<long long>(b) * V.rows;
        double const* fac = V.factor + static_cast<long long>(b) * V.factor_doubles;
        for(int k = t0; k < V.rows; k += T) w[k] = rhs[V.row_src[k]];
        tm.sync();
        // forward: L y = b, fronts in postorder
        for(int s = 0; s < V.nfronts; ++s)
        {
            int const c0 = V.f_col0[s], p = V.f_p[s], u = V.f_u[s], m = p + u;
            double const* Lp = fac + V.f_lptr[s];
            int const* rows = V.f_rows + V.f_rows_ptr[s];
            for(int i = t0; i < p; i += T) yl[i] = w[c0 + i];
            tm.sync();
            for(int k = 0; k + 1 < p; ++k)
            {
                double const yk = yl[k];
                for(int i = k + 1 + t0; i < p; i += T) yl[i] -= Lp[i + k * m] * yk;
                tm.sync();
            }
            for(int i = t0; i < p; i += T) w[c0 + i] = yl[i];
            for(int i = t0; i < u; i += T)
            {
                double acc = 0.0;
                for(int k = 0; k < p; ++k) acc += Lp[(p + i) + k * m] * yl[k];
                w[rows[i]] -= acc;
            }
            tm.sync();
        }
        // backward: U x = y, fronts in reverse postorder
        for(int s = V.nfronts - 1; s >= 0; --s)
        {
            int const c0 = V.f_col0[s], p = V.f_p[s], u = V.f_u[s], m = p + u;
            double const* Lp = fac + V.f_lptr[s];
            double const* Up = fac + V.f_uptr[s];
            int const* rows = V.f_rows + V.f_rows_ptr[s];
            for(int j = t0; j < u; j += T) yl[p + j] = w[rows[j]];
            tm.sync();
            for(int k = t0; k < p; k += T)
            {
                double acc = w[c0 + k];
                for(int j = 0; j < u; ++j) acc -= Up[k + j * p] * yl[p + j];
                yl[k] = acc;
            }
            tm.sync();
            for(int k = p - 1; k >= 0; --k)
            {
                double const xk = yl[k] / Lp[k + k * m];
                for(int i = t0; i < k; i += T) yl[i] -= Lp[i + k * m] * xk;
                tm.sync();
                if(t0 == 0) yl[k] = xk;
            }
            tm.sync();
            for(int k = t0; k < p; k += T) w[c0 + k] = yl[k];
            tm.sync();
        }
        for(int k = t0; k < V.rows; k += T) x[V.col_src[k]] = w[k];
        tm.sync();
    }

    // per-thread part of the Newton convergence test; the team reduces the returned flag with OR
    template <class Team>
    PE_DEV int newton_violations(Team const& tm, DevView const& V, int b)
    {
        double const* x = V.x + static_cast<long long>(b) * V.rows;
        double const* xp = V.xprev + static_cast<long long>(b) * V.rows;
        int bad = 0;
        for(int r = tm.tid(); r < V.rows; r += tm.size())
        {
            bool const node = r < V.n_nodes;
            double const atol = node ? V.v_abstol : V.i_abstol;
            double const rtol = node ? V.v_reltol : V.i_reltol;
            double const tol = atol + rtol * fmax(fabs(x[r]), fabs(xp[r]));
            // NaN-safe: a non-finite iterate counts as a violation
            if(!(fabs(x[r] - xp[r]) <= tol)) bad = 1;
        }
        return bad;
    }
    // ------------------------------------------------------------------------------------------------
    // one solve point = circult::solve (circuit.h:892-985).  Returns the number of solve_once-equivalents
    // (>= 1) or a negative status.  `reuse_factor`: linear circuit whose matrix is unchanged since the last
    // factorisation (same dt, same mode): stamp + triangular solves only.
    // ------------------------------------------------------------------------------------------------
    template <class Team>
    PE_DEV int solve_point(Team const& tm, DevView const& V, int b, int mode, double t, double last_step, bool reuse_factor, double* lds_front,
                           double* yl)
    {
        double* x = V.x + static_cast<long long>(b) * V.rows;
        double* xp = V.xprev + static_cast<long long>(b) * V.rows;
        int const iters = V.nonlinear ? V.max_newton : 1;
        for(int it = 0; it < iters; ++it)
        {
            if(V.nonlinear)
                for(int r = tm.tid(); r < V.rows; r += tm.size()) xp[r] = x[r];
            eval_devices(tm, V, b, mode, t, last_step);
            tm.sync();
            stamp(tm, V, b);
            tm.sync();
            if(!reuse_factor)
            {
                if(!factor_all(tm, V, b, lds_front)) return ST_SINGULAR;
            }
            solve_all(tm, V, b, yl);
            int nonfinite = 0;
            for(int r = tm.tid(); r < V.rows; r += tm.size())
                if(!(fabs(x[r]) <= 1.7976931348623157e308)) nonfinite = 1;
            if(tm.sync_or(nonfinite)) return ST_SINGULAR;
            if(!V.nonlinear) return 1;
            if(!tm.sync_or(newton_violations(tm, V, b))) return it + 1;
        }
        return ST_NO_CONVERGENCE;
    }

    // TR loop of circult::analyze (circuit.h:242-254) for `nsteps` steps of one instance
    template <class Team>
    PE_DEV void tr_steps(Team const& tm, DevView const& V, int b, double dt, int nsteps, bool reuse_factor, double* lds_front, double* yl)
    {
        if(V.status[b] != ST_OK) return;
        double t = V.t_now[b];
        long long steps = 0, iters = 0;
        int st = ST_OK;
        for(int sidx = 0; sidx < nsteps; ++sidx)
        {
            companion_update(tm, V, b, dt);
            tm.sync();
            double const prev = t;
            t = prev + dt;
            int const it = solve_point(tm, V, b, MODE_TR, t, dt, reuse_factor && sidx > 0 ? true : reuse_factor, lds_front, yl);
            if(b == 0 && tm.tid() == 0 && V.trace)
            {
                int const pos = *V.trace_len;
                if(pos < V.trace_cap) V.trace[pos] = it;
                *V.trace_len = pos + 1;
            }
            if(it < 0)
            {
                t = prev;
                st = it;
                break;
            }
            ++steps;
            iters += it;
        }
        tm.sync();
        if(tm.tid() == 0)
        {
            V.t_now[b] = t;
            V.last_step[b] = dt;
            V.status[b] = st;
            V.n_steps[b] += steps;
            V.n_iters[b] += iters;
        }
    }

    // OP / DC / TROP point of circult::analyze (circuit.h:183-191, 257-266)
    template <class Team>
    PE_DEV void dc_point(Team const& tm, DevView const& V, int b, int mode, double* lds_front, double* yl)
    {
        if(V.status[b] != ST_OK) return;
        int const it = solve_point(tm, V, b, mode, V.t_now[b], V.last_step[b], false, lds_front, yl);
        tm.sync();
        if(tm.tid() == 0)
        {
            if(b == 0 && V.trace)
            {
                int const pos = *V.trace_len;
                if(pos < V.trace_cap) V.trace[pos] = it;
                *V.trace_len = pos + 1;
            }
            if(it < 0) V.status[b] = it;
            else
                V.n_iters[b] += it;
        }
    }
}  // namespace pe
